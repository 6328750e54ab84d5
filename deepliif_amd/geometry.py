"""Layer geometry: turns one Conv2d / ConvTranspose2d of the reference networks into the descriptors the HIP gather-GEMM,
weight-pack and weight-gradient kernels consume (include/deepliif_hip.h).  Pure Python, no device work -- the formulas are
unit-tested on CPU against torch's own convolutions (tests/test_geometry.py).

Conventions (reference layouts, networks.py): Conv2d weight OIHW = src[A=out][B=in][kh][kw];
ConvTranspose2d weight IOHW = src[A=in][B=out][kh][kw].
"""
from __future__ import annotations

import os

from dataclasses import dataclass, field
from typing import List, Tuple

from . import _lib as L


def cpad(c: int) -> int:
    """Padded channel count of an engine tensor: the next power of two, at least 8."""
    p = 8
    while p < c:
        p *= 2
    return p


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


@dataclass
class GatherPlan:
    """One gather-GEMM problem family (forward or data-gradient of a layer), independent of N/H/W."""
    n_phase: int
    phase_off: List[Tuple[int, int]]            # (oh, ow) per phase
    phase_taps: List[List[Tuple[int, int, int, int]]]   # per phase: (dh, dw, kh, kw)
    out_step: int
    in_step: int
    pad_mode: int
    row_is_a: bool                              # packed row = src dim A ?
    rows_real: int                              # output channels (real)
    cc_real: int                                # contracted channels (real)

    # derived
    rows_pad: int = 0
    cc_pad: int = 0
    kbase: List[int] = field(default_factory=list)
    kstride: int = 0
    stack_kw: int = 0                           # >0: packed row = a*stack_kw + kw (narrow-Cout path)

    def finish(self):
        self.rows_pad = round_up(cpad(self.rows_real), 128)
        self.cc_pad = cpad(self.cc_real)
        self.kbase = []
        k = 0
        for taps in self.phase_taps:
            self.kbase.append(k)
            k += round_up(len(taps) * self.cc_pad, 64)
        self.kstride = k
        assert sum(len(t) for t in self.phase_taps) <= L.MAX_TAPS
        return self

    def taps_flat(self):
        return [t for ph in self.phase_taps for t in ph]

    def tap_begin(self):
        b = [0]
        for ph in self.phase_taps:
            b.append(b[-1] + len(ph))
        return b + [b[-1]] * (L.MAX_PHASES + 1 - len(b))


@dataclass
class ConvSpec:
    """Static description of one convolution layer of the reference networks."""
    kind: str          # 'conv' | 'convT'
    cin: int
    cout: int
    k: int
    stride: int
    pad: int
    pad_mode: int = L.PAD_ZERO      # reflect = an explicit nn.ReflectionPad2d(pad) in front of a padding=0 Conv2d
    out_pad: int = 0

    def out_hw(self, h, w):
        if self.kind == 'conv':
            return (h + 2 * self.pad - self.k) // self.stride + 1, (w + 2 * self.pad - self.k) // self.stride + 1
        return ((h - 1) * self.stride - 2 * self.pad + self.k + self.out_pad,
                (w - 1) * self.stride - 2 * self.pad + self.k + self.out_pad)

    # ---- forward: y = layer(x)
    def forward_plan(self) -> GatherPlan:
        k, p, s = self.k, self.pad, self.stride
        if self.kind == 'conv':
            taps = [(kh - p, kw - p, kh, kw) for kh in range(k) for kw in range(k)]
            return GatherPlan(1, [(0, 0)], [taps], 1, s, self.pad_mode, True, self.cout, self.cin).finish()
        assert s == 2, 'ConvTranspose2d on this path is always stride 2 (networks.py:426-430, 584-600)'
        offs, phases = [], []
        for ph in range(2):
            for pw in range(2):
                taps = [((ph + p - kh) // 2, (pw + p - kw) // 2, kh, kw)
                        for kh in range(k) if (ph + p - kh) % 2 == 0
                        for kw in range(k) if (pw + p - kw) % 2 == 0]
                offs.append((ph, pw))
                phases.append(taps)
        return GatherPlan(4, offs, phases, 2, 1, L.PAD_ZERO, False, self.cout, self.cin).finish()

    # ---- narrow-Cout forward (dl_shift_sum): rows = (co, kw), vertical taps only
    def is_narrow(self) -> bool:
        return self.kind == 'conv' and self.stride == 1 and self.cout <= 4 and self.k >= 5 and self.cout * self.k <= 32

    def narrow_forward_plan(self) -> GatherPlan:
        k, p = self.k, self.pad
        taps = [(kh - p, 0, kh, 0) for kh in range(k)]
        plan = GatherPlan(1, [(0, 0)], [taps], 1, 1, self.pad_mode, True, self.cout * k, self.cin)
        plan.stack_kw = k
        return plan.finish()

    # ---- data gradient: dx = layer^T(dy)
    def dgrad_plan(self) -> GatherPlan:
        k, p, s = self.k, self.pad, self.stride
        if self.kind == 'conv':
            if self.pad_mode != L.PAD_ZERO:
                # nn.ReflectionPad2d(p) + Conv2d(padding=0): this plan is the gradient with respect to the EXPLICITLY PADDED input
                # (extent (H+2p) x (W+2p), pad-0 conv: dxp[a,b] = sum dy[a-kh, b-kw] W); dl_reflect_fold then adds the mirrored
                # borders back onto the interior (engine.conv).  Same packed image as the zero-padding plan, other tap offsets.
                assert s == 1, 'reflection-padded convs of the reference networks are all stride 1 (networks.py:386-388, 438-440, 478-498)'
                taps = [(-kh, -kw, kh, kw) for kh in range(k) for kw in range(k)]
                return GatherPlan(1, [(0, 0)], [taps], 1, 1, L.PAD_ZERO, False, self.cin, self.cout).finish()
            if s == 1:
                taps = [(p - kh, p - kw, kh, kw) for kh in range(k) for kw in range(k)]
                return GatherPlan(1, [(0, 0)], [taps], 1, 1, L.PAD_ZERO, False, self.cin, self.cout).finish()
            assert s == 2
            offs, phases = [], []
            for ph in range(2):
                for pw in range(2):
                    taps = [((ph + p - kh) // 2, (pw + p - kw) // 2, kh, kw)
                            for kh in range(k) if (ph + p - kh) % 2 == 0
                            for kw in range(k) if (pw + p - kw) % 2 == 0]
                    offs.append((ph, pw))
                    phases.append(taps)
            return GatherPlan(4, offs, phases, 2, 1, L.PAD_ZERO, False, self.cin, self.cout).finish()
        # ConvTranspose2d: dx[ci][hi] = sum dy[co][hi*2 - p + kh] * Wt[ci][co][kh]  -> a stride-2 conv over dy
        taps = [(kh - p, kw - p, kh, kw) for kh in range(k) for kw in range(k)]
        return GatherPlan(1, [(0, 0)], [taps], 1, s, L.PAD_ZERO, True, self.cin, self.cout).finish()


def fill_pack_desc(plan: GatherPlan, A: int, B: int, k: int) -> L.PackDesc:
    d = L.PackDesc()
    d.A, d.B, d.KH, d.KW = A, B, k, k
    d.row_is_a = 1 if plan.row_is_a else 0
    d.rows_real, d.rows_pad = plan.rows_real, plan.rows_pad
    d.Cc, d.Cc_pad = plan.cc_real, plan.cc_pad
    d.n_phase = plan.n_phase
    for i, v in enumerate(plan.tap_begin()):
        d.phase_tap_begin[i] = v
    for i, v in enumerate(plan.kbase):
        d.phase_kbase[i] = v
    for i, (_, _, kh, kw) in enumerate(plan.taps_flat()):
        d.tap_kh[i], d.tap_kw[i] = kh, kw
    d.kstride = plan.kstride
    d.stack_kw = 1 if plan.stack_kw else 0
    if plan.stack_kw:
        d.KW = plan.stack_kw
    return d


def conv_tile(co_pad: int):
    """(BM pixels, BN channels) the kernel dispatch picks (conv_gemm.hip dispatch_tile)."""
    if co_pad <= 16:
        return 256, 16
    if co_pad <= 64:
        return 128, 64
    return 128, 128


def choose_splitk(plan: GatherPlan, n: int, hq: int, wq: int, co_pad: int, target_blocks: int = 512, max_split: int = 32) -> int:
    bm, bn = conv_tile(co_pad)
    blocks = ((n * hq * wq + bm - 1) // bm) * ((co_pad + bn - 1) // bn) * plan.n_phase
    nk64 = min(round_up(len(t) * plan.cc_pad, 64) // 64 for t in plan.phase_taps)
    if co_pad % 256 == 0 and plan.cc_pad >= 64:
        # few pixels, long K (PatchGAN 512->512 at 31 x 31: 61 x 4 tiles of 128 x 128 leave the K loop bound by the bytes staged per MFMA):
        # 256 x 256 tiles stage half the bytes per MFMA; split K so that every CU gets exactly one workgroup.  Pays only while each
        # partial still runs a long loop -- the partials cost a 60 MB slab round trip (tools/splitk_probe.py: 139 -> 89 us at 32 steps
        # per partial, 78 -> 64 us at 16, a loss at 2 partials).  Mirrors big_tile_fills_gpu() in csrc/conv_gemm.hip.
        t256 = ((n * hq * wq + 255) // 256) * (co_pad // 256) * plan.n_phase
        sk = 256 // max(t256, 1)
        if t256 < 224 and sk >= 2 and t256 * sk >= 224 and nk64 // sk >= 24:
            return sk
    if blocks >= 256:
        return 1
    sk = min(max_split, nk64, (target_blocks + blocks - 1) // blocks)
    return max(1, sk)


def fill_conv_desc(plan: GatherPlan, n: int, hi: int, wi: int, in_pstride: int, ho: int, wo: int, out_cp: int, out_pstride: int,
                   hq: int, wq: int, dtype: int, prec: int, act: int, in_act: int, bias_n: int, splitk: int, raw_out: int = 0) -> L.ConvDesc:
    """the dl_conv_desc of one launch.  Filling ~70 ctypes fields (9-49 taps) from Python costs 10-24 us -- about a tenth of the host time of a training step
    (r05, tools/host_profile.py) -- and a layer launches with the same geometry every step: the filled descriptor is kept on the plan and every call gets its own
    COPY (0.5 us; callers set in_split / splitk on it)."""
    key = (n, hi, wi, in_pstride, ho, wo, out_cp, out_pstride, hq, wq, dtype, prec, act, in_act, bias_n, splitk, raw_out)
    cache = plan.__dict__.setdefault('_desc_cache', {})
    d = cache.get(key)
    if d is None:
        if len(cache) > 64:
            cache.clear()
        d = cache[key] = _fill_conv_desc(plan, *key)
    return L.ConvDesc.from_buffer_copy(d)


def _fill_conv_desc(plan: GatherPlan, n: int, hi: int, wi: int, in_pstride: int, ho: int, wo: int, out_cp: int, out_pstride: int,
                    hq: int, wq: int, dtype: int, prec: int, act: int, in_act: int, bias_n: int, splitk: int, raw_out: int = 0) -> L.ConvDesc:
    d = L.ConvDesc()
    d.N, d.Hi, d.Wi, d.Ci, d.in_pstride = n, hi, wi, plan.cc_pad, in_pstride
    d.Ho, d.Wo, d.Co, d.out_pstride = ho, wo, out_cp, out_pstride
    d.Hq, d.Wq, d.out_step, d.in_step, d.n_phase = hq, wq, plan.out_step, plan.in_step, plan.n_phase
    for i, (oh, ow) in enumerate(plan.phase_off):
        d.phase_oh[i], d.phase_ow[i] = oh, ow
    for i, v in enumerate(plan.tap_begin()):
        d.phase_tap_begin[i] = v
    for i, v in enumerate(plan.kbase):
        d.phase_kbase[i] = v
    for i, (dh, dw, _, _) in enumerate(plan.taps_flat()):
        d.tap_dh[i], d.tap_dw[i] = dh, dw
    d.pad_mode, d.w_kstride, d.w_rows = plan.pad_mode, plan.kstride, plan.rows_pad
    d.act, d.in_dtype, d.out_dtype, d.prec, d.splitk, d.in_act, d.bias_n = act, dtype, dtype, prec, splitk, in_act, bias_n
    d.raw_out = raw_out
    d.ci_real = plan.cc_real
    return d


NUM_CUS = 256      # MI355X
# experiment switches (r05): share of the chip one weight-gradient launch is sized for -- a smaller share means fewer split-K partials (slab bytes ~ share x 67 MB
# per launch whatever the layer), more K steps per workgroup, and relies on the other branch streams to fill the remaining CUs
WGRAD_FILL = float(os.environ.get('DL_WGRAD_FILL', '1.0'))               # one layer per launch (choose_wgrad_splitk, fast path)
WGRAD_BATCH_FILL = float(os.environ.get('DL_WGRAD_BATCH_FILL', '0.3333'))    # a layer inside a batch (choose_wgrad_batch_splitk)


def wgrad_fast_path(cap: int, j: int, bf16: bool, no_act: bool, zero_pad: bool, strict: bool = False, strict_act_ok: bool = True) -> bool:
    """mirror of the dispatch predicates in wgrad.hip: the direct-to-LDS 8-wave kernels -- wgrad_glds_kernel (bf16 policy, no staged
    activation) and wgrad_glds_x3_kernel (strict policy = fp32 storage + split-bf16 x3; relu / lrelu on an operand are applied while the
    tile is split in LDS, csrc/wgrad_x3.h)"""
    shape_ok = zero_pad and j >= 256 and cap % 128 == 0
    return shape_ok and ((bf16 and no_act) or (strict and strict_act_ok))


def choose_wgrad_splitk(cap: int, j: int, ptot: int, fast: bool = False, target_blocks: int = 1024, max_split: int = 256) -> int:
    if fast:
        # 8-wave kernel: one workgroup per CU (248 VGPRs) -> the grid must not exceed one round of 256 CUs by a few blocks
        ba = 256 if cap % 256 == 0 else 128
        tiles = (cap // ba) * ((j + 255) // 256)
        cus = max(1, int(NUM_CUS * WGRAD_FILL))
        if tiles >= cus:
            return 1
        return max(1, min(cus // tiles, ptot // 128, max_split))
    ba = 16 if cap <= 16 else (64 if cap <= 64 else 128)
    tiles = ((cap + ba - 1) // ba) * ((j + 127) // 128)
    sk = min(max_split, (target_blocks + tiles - 1) // tiles, max(1, ptot // 64))
    return max(1, sk)


def wgrad_batch_shape(kh: int, kw: int, step: int, wp: int, hp: int, hq: int, cap: int, cbp: int) -> bool:
    """the layers ops.HipBackend queues for the batched weight gradient: the ResnetBlock conv (networks.py:467-513) -- 3x3, stride 1, image rows of 128 pixels --
    i.e. the one shape a network repeats (18 times in a Resnet-9).  Every other layer occurs once per network and pass: a batch of one would only
    under-fill the chip with the batch's small split-K."""
    return kh == 3 and kw == 3 and step == 1 and wp == 128 and hp == hq and cap % 128 == 0 and cbp % 128 == 0


def choose_wgrad_batch_splitk(tiles: int, ksteps: int) -> int:
    """split-K of a layer inside a batch: its tiles x splitk workgroups take about a THIRD of the chip, whatever the batch size -- the result of a layer
    then does not depend on how many layers share its launch (a flush in the middle of a network, branch streams or not, leaves every gradient
    bit-identical), three layers fill the chip, and 18 make ~6 rounds of it, so every workgroup's slab store hides behind the next round's main loops.
    ResnetBlock conv at batch 8: 12 tiles x 7 row ranges (w4 kernel; r04: 9 tiles x 28 pixel ranges, four times the slab bytes)."""
    return max(1, min((int(NUM_CUS * WGRAD_BATCH_FILL) + tiles // 2) // tiles, max(1, ksteps // 16)))


WGRAD_C4_PARTS = 512


def wgrad_c4_ok(cap: int, ca: int, cbp: int, cb: int, k: int, step: int, pad: int, pad_mode: int, hp: int, wp: int, hq: int, wq: int, bf16: bool,
                no_act: bool, stacked: bool) -> bool:
    """mirror of wgrad_c4_form (csrc/wgrad_c4.h): the 7x7 weight gradient with one 64-channel and one <= 4-channel operand (Resnet stem / head)"""
    if not (bf16 and no_act) or stacked or k != 7 or step != 1 or pad != 3 or pad_mode != L.PAD_ZERO:
        return False
    if (hp, wp) != (hq, wq) or hp % 4 or wp % 64:
        return False
    return (cap == 64 and cbp == 8 and cb <= 4) or (cap == 8 and ca <= 4 and cbp == 64)
