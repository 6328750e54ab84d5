"""Execution engine: NHWC activations, a reverse-mode tape and the differentiable building blocks of the three reference
networks.  This replaces torch.autograd + ATen for the hot path: every forward/backward node below launches the
hand-written gfx950 kernels through deepliif_amd.ops (C ABI), on torch's current HIP stream.

Precision policies:
  'bf16' : activations bf16, one bf16 MFMA pass, fp32 accumulate / statistics / losses / master weights  (throughput mode)
  'fp32' : activations fp32, split-bf16 x3 MFMA (fp32-class accuracy)                                   (strict parity mode)
"""
from __future__ import annotations

import os
import weakref

from dataclasses import dataclass
from typing import Callable, List, Optional

import torch

from . import _lib as L
from . import ops
from .geometry import ConvSpec, cpad


@dataclass(frozen=True)
class Precision:
    name: str
    dtype: torch.dtype
    prec: int

    @property
    def is16(self) -> bool:
        """activations stored in the library's 16-bit type, one MFMA pass"""
        return self.dtype in (torch.bfloat16, torch.float16)

    @property
    def half(self) -> str:
        """which library serves this policy (ops.half_mode): 'fp16' -> libdeepliif_hip_f16.so, everything else -> libdeepliif_hip.so"""
        return 'fp16' if self.dtype == torch.float16 else 'bf16'

    @staticmethod
    def get(name: str) -> 'Precision':
        if name in ('bf16', 'bfloat16'):
            return Precision('bf16', torch.bfloat16, L.PREC_BF16)
        if name in ('fp16', 'float16', 'half'):
            # INFERENCE ONLY (Ctx refuses a tape): IEEE half storage + MFMA operands, 11 significand bits instead of bf16's 8 at the same matrix rate --
            # a generator's forward lands 7x nearer to the fp32 reference (profiles/r05/fp16_policy_experiment.json).  Not for training: 89-100 % of this
            # model's conv-output gradients are below half's smallest normal number.
            return Precision('fp16', torch.float16, L.PREC_BF16)
        if name in ('fp32', 'float32'):
            return Precision('fp32', torch.float32, L.PREC_BF16X3)
        if name == 'fp32_bf16mma':
            return Precision('fp32_bf16mma', torch.float32, L.PREC_BF16)
        raise ValueError(f'unknown precision {name!r} (bf16 | fp16 | fp32 | fp32_bf16mma)')


class Act:
    """An activation in engine layout: t = [N, H, W, Cp] view (bf16/fp32), C real channels; padded channels hold zeros."""
    __slots__ = ('t', 'C', 'grad', 'needs_grad', 'bias_grad', 'bias_done', 'stats', 'bn_ctx', 'grad_stats', 'norm_only', 'split', 'grad_split',
                 'split_backward_ok', 'grad_values_stored', 'values_stored')

    def __init__(self, t: torch.Tensor, C: int, needs_grad: bool = False):
        self.t = t
        self.C = C
        self.grad: Optional[torch.Tensor] = None
        self.needs_grad = needs_grad
        self.bias_grad: Optional[torch.Tensor] = None     # conv output: where the producer's bias gradient accumulates
        self.bias_done = False                            # set when a consumer's backward already added sum(dy) to it
        self.stats = None                                 # conv output: (chunks, workspace token) of fused norm statistics
        self.norm_only = False                            # conv output handed straight to ONE norm_act and to nothing else (conv(stats=True))
        self.bn_ctx = None                                # norm_act output z = act(norm(y)): (y tensor, statistics, act) for a consumer conv's backward
        self.grad_stats = None                            # (chunks, workspace token): the ONLY contribution to .grad also left the norm-backward reductions
        # strict policy: SPLIT COPIES ([8 bf16 hi | 8 bf16 lo] per group of 8 channels, same shape / bytes as the fp32 tensor) written by the
        # norm kernel that produced the tensor, so that the convolutions consuming it skip their in-kernel hi / lo split (csrc/conv_x3.h)
        self.split: Optional[torch.Tensor] = None         # of .t
        self.grad_split: Optional[torch.Tensor] = None    # of .grad, valid only while .grad is that ONE contribution
        # conv output: callable(dy-like tensor, bias_done) -> bool, "my whole backward can run from the SPLIT copy of dL/dy" (conv() sets it); a
        # norm_act behind a norm_only conv output then skips the fp32 store of dL/dy (csrc/norm.hip: dy == NULL) and clears grad_values_stored
        self.split_backward_ok = None
        self.grad_values_stored = True                    # False: .grad only carries the geometry, the values live in .grad_split
        self.values_stored = True                         # False: .t only carries the geometry, the values live in .split (norm_act(sole_reader=...))

    @property
    def shape(self):
        return self.t.shape

    def detach(self) -> 'Act':
        return Act(self.t, self.C, False)

    def add_grad(self, g: torch.Tensor):
        """accumulate a gradient contribution (first one is adopted, later ones are added with the axpby kernel)"""
        if self.grad is None:
            self.grad = g
        else:
            assert self.grad_values_stored, 'a split-only gradient cannot take a second contribution (norm_only promise broken)'
            ops.impl().axpby(1.0, self.grad, 1.0, g, self.grad)
            self.grad_stats = None          # reductions fused into the first contribution's producer no longer describe the sum
            self.grad_split = None          # ... nor does its split copy


_STREAM_OBJECTS: dict = {}         # raw HIP stream handle -> a torch Stream object of it (two objects with one handle are the same stream)


class Tape:
    """Reverse-mode tape: forward ops append closures, backward() runs them last-to-first.

    Parameter bookkeeping for the data-parallel exchange (distributed.GradExchanger): every recorded op that will add to a parameter's
    gradient announces it with use(); its backward closure calls done() after the gradient has been accumulated.  When the LAST
    outstanding use of a parameter is done, its gradient is final for this pass and `on_final(param)` fires -- a network that runs
    several times on one tape (a discriminator on real and fake pairs) therefore reports a weight only once, after both passes."""

    def __init__(self, streams: bool = False):
        self.nodes: List[Callable[[], None]] = []
        self.uses = {}
        self.on_final: Optional[Callable[[torch.nn.Parameter], None]] = None
        # branch streams (models.BaseModel._branch): every node remembers the HIP stream it was recorded on and runs its backward there -- a
        # branch's forward, its losses and its backward stay in one in-order stream, different branches overlap on the GPU
        self.node_streams: Optional[list] = [] if streams else None

    def record(self, fn: Callable[[], None]):
        self.nodes.append(fn)
        if self.node_streams is not None:
            # (the Stream OBJECT of the current stream, looked up by its raw handle: torch.cuda.current_stream() costs ~7 us per call, ops._current_stream_handle)
            h = (torch._C._cuda_getDevice() if ops._FAST_STREAM and torch.cuda.is_initialized() else -1, ops._current_stream_handle())
            s = _STREAM_OBJECTS.get(h)
            if s is None or h[0] < 0:
                s = _STREAM_OBJECTS[h] = torch.cuda.current_stream()
            self.node_streams.append(s)

    def use(self, *params):
        for p in params:
            if p is not None and p.requires_grad:
                self.uses[id(p)] = self.uses.get(id(p), 0) + 1

    def done(self, *params):
        for p in params:
            if p is None or id(p) not in self.uses:
                continue
            self.uses[id(p)] -= 1
            if self.uses[id(p)] == 0:
                del self.uses[id(p)]
                if self.on_final is not None:
                    self.on_final(p)

    def backward(self):
        nodes, self.nodes = self.nodes, []
        # the weight gradients of the pass leave their split-K slabs side by side and are reduced in batches (ops.HipBackend.wgrad_flush)
        be = ops.impl()
        begin = getattr(be, 'wgrad_defer_begin', None)
        if begin is not None:
            begin()
        streams, self.node_streams = self.node_streams, ([] if self.node_streams is not None else None)
        try:
            if streams is None:
                for fn in reversed(nodes):
                    fn()
            else:
                home = torch.cuda.current_stream()
                cur = home
                try:
                    for fn, s in zip(reversed(nodes), reversed(streams)):
                        if s != cur:
                            torch.cuda.set_stream(s)
                            cur = s
                        fn()
                finally:
                    torch.cuda.set_stream(home)
        finally:
            if begin is not None:
                be.wgrad_defer_end()


def settle_gc():
    """Called once a model / a set of inference nets has been built.  A training step creates a few thousand short-lived Python objects (tape
    closures, Act wrappers) that reference each other, so the cyclic collector runs: ~10 generation-0 passes per step (cheap) and, every ~20 steps,
    a FULL collection that walks every tracked object of the process -- with ten networks' module trees, parameters and torch's own import graph that
    took 75 ms on the benched step, during which no kernel is issued (tools/step_jitter.py on MI355X: median 101.75 ms, every 21st step 175-182 ms,
    mean 105.6 ms).  gc.freeze() moves everything that exists NOW (the long-lived model) into the permanent generation, which full collections skip:
    they drop below 1 ms (mean 102.25 ms, max 103.75 ms over 60 steps).  Nothing is ever leaked by this -- frozen objects are still freed by reference
    counting; only cycles among objects that existed at this moment stay until the next call (which unfreezes first) or exit.  It is a process-wide change of the
    interpreter's collector made by a library constructor: INTEGRATION.md section 1 lists it; DEEPLIIF_AMD_GC_FREEZE=0 leaves the collector alone."""
    if os.environ.get('DEEPLIIF_AMD_GC_FREEZE', '1') == '0':
        return
    import gc
    # unfreeze first: whatever an EARLIER call froze and the host application has dropped since (a previous model of a sweep, a test's
    # fixture) becomes collectable again -- cycles included -- before the objects alive NOW are moved to the permanent generation
    gc.unfreeze()
    gc.collect()            # every call: a model built later (a DeepLIIFKD student after its teacher) is settled as well
    gc.freeze()


def new_act(n, h, w, c, prec: Precision, device, zero=False) -> Act:
    cp = cpad(c)
    t = (torch.zeros if zero else torch.empty)((n, h, w, cp), dtype=prec.dtype, device=device)
    return Act(t, c)


# A/B switch (DL_CONV_PREACT=0 disables): apply a conv's input activation in a separate pass so the conv can take the direct-to-LDS path
_PREACT = os.environ.get('DL_CONV_PREACT', '1') != '0'
# A/B switch (DL_WGRAD_PREACT=0 disables): the activation materialised by DL_CONV_PREACT is kept for the layer's weight gradient, which then takes the direct-to-LDS kernel too
_WGRAD_PREACT = os.environ.get('DL_WGRAD_PREACT', '1') != '0'
# A/B switch (DL_CONVT4=0 disables): narrow-Cout ConvTranspose2d(4, 2, 1) at inference as one 1x1 GEMM + a 2x2 gather-sum (conv() below)
_CONVT4 = os.environ.get('DL_CONVT4', '1') != '0'
# A/B switch (DL_CONVT4_TRAIN=0): that route only without a tape, as in rounds 4-5 (the training forward of the layer then runs the 4-phase gather GEMM: 337 us per UNet)
_CONVT4_TRAIN = os.environ.get('DL_CONVT4_TRAIN', '1') != '0'


def empty_like_act(a: torch.Tensor) -> torch.Tensor:
    return torch.empty(a.shape, dtype=a.dtype, device=a.device)


# ---------------------------------------------------------------------------------------------------------------
# layers (parameter holders are the nn.Modules in networks.py; these classes own geometry + packed weight images)
# ---------------------------------------------------------------------------------------------------------------
# nn.Parameter (by identity) -> the ConvLayer bound to it: lets an optimizer find the GEMM images of the weights it updates
LAYER_OF_WEIGHT: 'weakref.WeakValueDictionary[int, ConvLayer]' = weakref.WeakValueDictionary()


class ConvLayer:
    """One Conv2d / ConvTranspose2d of the reference networks bound to its nn.Parameter(s)."""

    def __init__(self, spec: ConvSpec, weight: torch.nn.Parameter, bias: Optional[torch.nn.Parameter]):
        self.spec = spec
        self.weight = weight
        self.bias = bias
        self.fwd_plan = spec.forward_plan()
        self._dgrad_plan = None
        self.packed_fwd: Optional[ops.PackedWeights] = None
        self.packed_dgrad: Optional[ops.PackedWeights] = None
        self.fwd_key = None
        self.dgrad_key = None
        LAYER_OF_WEIGHT[id(weight)] = self
        # narrow-Cout layers (the 7x7, 64 -> 3 ResnetGenerator head): kernel columns folded into the GEMM rows (dl_shift_sum)
        self.narrow = spec.is_narrow()
        if self.narrow:
            self.fwd_plan = spec.narrow_forward_plan()

    @property
    def dgrad_plan(self):
        if self._dgrad_plan is None:
            self._dgrad_plan = self.spec.dgrad_plan()
        return self._dgrad_plan

    def ensure_packed(self, prec: Precision, need_dgrad: bool):
        """(Re)build the bf16 GEMM images when the fp32 master weights changed: torch-side in-place updates bump
        `_version`; the fused Adam kernel writes through raw pointers and bumps `_dl_epoch` instead (optim.py)."""
        w = self.weight
        key = (w._version, w.data_ptr(), getattr(w, '_dl_epoch', 0), prec.prec, str(w.device), prec.half)
        with_lo = prec.prec == L.PREC_BF16X3
        be = ops.impl()
        if self.fwd_key != key:
            if self.packed_fwd is None or self.packed_fwd.hi.device != w.device or (with_lo and self.packed_fwd.lo is None):
                self.packed_fwd = ops.PackedWeights(self.fwd_plan, w.device, with_lo)
            be.pack_weights(self.packed_fwd, w.detach())
            self.fwd_key = key
        if need_dgrad and self.dgrad_key != key:
            if self.packed_dgrad is None or self.packed_dgrad.hi.device != w.device or (with_lo and self.packed_dgrad.lo is None):
                self.packed_dgrad = ops.PackedWeights(self.dgrad_plan, w.device, with_lo)
            be.pack_weights(self.packed_dgrad, w.detach())
            self.dgrad_key = key

    def ensure_packed_taps(self, prec: Precision):
        """Narrow-Cout ConvTranspose2d(k=4, s=2, p=1) at inference (conv() below): the weights as ONE 1x1 GEMM image with a row per
        (ky, kx, co) -- W'[(ky*4+kx)*Cout + co][ci] = W[ci][co][ky][kx] -- so the input is staged once instead of once per (phase, tap)."""
        w = self.weight
        key = (w._version, w.data_ptr(), getattr(w, '_dl_epoch', 0), prec.prec, str(w.device), prec.half)
        if getattr(self, 'taps_key', None) != key:
            spec = self.spec
            if getattr(self, 'packed_taps', None) is None or self.packed_taps.hi.device != w.device:
                self.taps_spec = ConvSpec('conv', spec.cin, 16 * spec.cout, 1, 1, 0)
                self.packed_taps = ops.PackedWeights(self.taps_spec.forward_plan(), w.device, prec.prec == L.PREC_BF16X3)
            wt = w.detach().permute(2, 3, 1, 0).reshape(16 * spec.cout, spec.cin, 1, 1).contiguous()
            ops.impl().pack_weights(self.packed_taps, wt)
            self.taps_key = key
        return self.packed_taps


class PackBatch:
    """Repacks, in ONE launch, every GEMM image that already exists for the conv weights of a parameter set (called by the fused
    Adam right after it changed them).  ConvLayer.ensure_packed stays the source of truth: images that do not exist yet, or whose
    weights were changed some other way (load_state_dict, a torch optimizer), are (re)built lazily there as before."""

    def __init__(self, params):
        self.layers = [LAYER_OF_WEIGHT[id(p)] for p in params if id(p) in LAYER_OF_WEIGHT]
        self.table, self.sig = None, None

    def run(self):
        be = ops.impl()
        jobs, stamps = [], []
        for layer in self.layers:
            w = layer.weight
            for packed, attr in ((layer.packed_fwd, 'fwd_key'), (layer.packed_dgrad, 'dgrad_key')):
                key = getattr(layer, attr)
                if packed is None or key is None or key[1] != w.data_ptr() or packed.hi.device != w.device:
                    continue            # never packed / storage moved: the lazy path handles it
                jobs.append((packed, w.detach()))
                stamps.append((layer, attr, (w._version, w.data_ptr(), getattr(w, '_dl_epoch', 0), key[3], key[4], key[5])))
        if not jobs:
            return 0
        sig = tuple((pk.hi.data_ptr(), pk.lo.data_ptr() if pk.lo is not None else 0, w.data_ptr()) for pk, w in jobs)
        if sig != self.sig:
            self.table, self.sig = be.pack_batch_build(jobs), sig
        be.pack_batch_run(self.table, len(jobs))
        for layer, attr, key in stamps:
            setattr(layer, attr, key)
        return len(jobs)


class NormLayer:
    """BatchNorm2d-on-batch-statistics (affine, optional running-stat tracking) or InstanceNorm2d (no affine)."""

    def __init__(self, kind: str, C: int, module: Optional[torch.nn.Module]):
        self.kind = kind
        self.C = C
        self.module = module        # nn.BatchNorm2d holding weight/bias/running_* (None for instance norm)

    @property
    def scope(self):
        return L.NORM_BATCH if self.kind == 'batch' else L.NORM_INSTANCE


# ---------------------------------------------------------------------------------------------------------------
# differentiable ops
# ---------------------------------------------------------------------------------------------------------------
class Ctx:
    """Per-call execution context."""

    def __init__(self, prec: Precision, tape: Optional[Tape], training: bool, per_sample_norm: bool = False):
        if prec.half == 'fp16' and (tape is not None or training):
            raise ValueError("precision 'fp16' is an inference policy: gradients of this model underflow IEEE half (train with 'bf16' or 'fp32')")
        if prec.half != ops.half_format():
            raise RuntimeError(f"precision {prec.name!r} needs the {prec.half} library: run the forward inside ops.half_mode({prec.half!r}) "
                               f"(the networks' forward() and inference.py do)")
        self.prec = prec
        self.tape = tape
        self.training = training            # BatchNorm running-stat updates (module.training and tracking enabled)
        # batched inference must normalise per sample to reproduce the reference's one-tile-per-forward outputs
        # (SURVEY 0 #5): BatchNorm on batch statistics with N=1 == InstanceNorm + affine
        self.per_sample_norm = per_sample_norm


def conv_reads_split_only(ctx: Ctx, layer: 'ConvLayer', x_t: torch.Tensor, in_act: int = L.ACT_NONE) -> bool:
    """Will conv(ctx, x, layer, in_act=in_act) -- forward AND backward -- read x only through its split copy?  Mirrors the branches of conv()."""
    be = ops.impl()
    spec = layer.spec
    if not (getattr(be, 'supports_split', False) and ops._SPLIT_ONLY_GRAD) or in_act != L.ACT_NONE or layer.narrow or ctx.prec.prec != L.PREC_BF16X3:
        return False
    if not be.conv_takes_split(x_t, ctx.prec.prec, L.ACT_NONE, spec.pad_mode):
        return False
    if layer.weight.requires_grad and ctx.tape is not None:
        n, hi, wi, _ = x_t.shape
        ho, wo = spec.out_hw(hi, wi)
        g_like = torch.empty((n, ho, wo, cpad(spec.cout)), dtype=x_t.dtype, device='meta')
        if spec.kind == 'conv':
            return be.wgrad_takes_split(g_like, x_t, layer.weight.grad, spec.k, spec.pad_mode, ctx.prec.prec)
        return be.wgrad_takes_split(x_t, g_like, layer.weight.grad, spec.k, L.PAD_ZERO, ctx.prec.prec)
    return True


def conv(ctx: Ctx, x: Act, layer: ConvLayer, act: int = L.ACT_NONE, in_act: int = L.ACT_NONE, out: Optional[torch.Tensor] = None,
         stats: bool = False) -> Act:
    """y = act(conv(in_act(x)) + bias).  `out` may be a channel-slice view of a concat buffer.
    stats: the caller feeds y straight into ONE norm_act and into nothing else -- let the conv epilogue produce the norm statistics when
    it can, and let that norm's backward add sum(dy) to this conv's bias gradient (dy is then the only gradient y ever receives)."""
    be = ops.impl()
    spec = layer.spec
    n, hi, wi, _ = x.t.shape
    ho, wo = spec.out_hw(hi, wi)
    x_pre = None            # in_act(x) materialised by the forward pass and kept for the weight gradient (below)
    w_needs = layer.weight.requires_grad and ctx.tape is not None
    x_needs = x.needs_grad and ctx.tape is not None
    layer.ensure_packed(ctx.prec, need_dgrad=x_needs)
    if out is None:
        out = torch.empty((n, ho, wo, cpad(spec.cout)), dtype=ctx.prec.dtype, device=x.t.device)
    hq, wq = (ho, wo) if spec.kind == 'conv' else (hi, wi)
    if spec.kind == 'convT':
        assert (ho, wo) == (2 * hi, 2 * wi)
    assert x.values_stored or not layer.narrow, 'a split-only activation reached the narrow-Cout path'
    if layer.narrow and in_act == L.ACT_NONE and ((ctx.prec.prec == L.PREC_BF16 and ctx.prec.is16 and x.t.dtype == ctx.prec.dtype) or (ctx.prec.prec == L.PREC_BF16X3 and x.t.dtype == torch.float32)) and \
            be.conv_narrow_supported(x.t, x.t.shape[3], spec.cout, spec.k, spec.pad, spec.pad_mode, act):
        # one kernel: every input row staged once, all kernel rows at once, kernel-column sum from LDS (conv_small.hip)
        be.conv_narrow_forward(layer.packed_fwd, x.t, out, spec.cout, spec.k, spec.pad, layer.bias.detach() if layer.bias is not None else None, act)
        nch = 0
    elif (_CONVT4 and spec.kind == 'convT' and spec.k == 4 and spec.stride == 2 and spec.pad == 1 and spec.cout <= 4 and (_CONVT4_TRAIN or not (x_needs or w_needs))
          and ctx.prec.prec == L.PREC_BF16 and ctx.prec.is16 and x.t.dtype == ctx.prec.dtype and getattr(be, 'convt4_gather', None) is not None):
        # UnetGenerator's outermost up-convolution to 3 channels (networks.py:573-576), inference and (r06) the training forward: one 1x1 GEMM over the input with a row per
        # (ky, kx, co), then the 2x2 gather-sum + bias + tanh (dl_convt4_gather).  The 4-phase gather GEMM stages every input pixel 16 times for
        # 3 useful columns: 327 us at 8 x 256^2 x 128, 5.9 % of the inference batch.
        xin = x.t
        if in_act != L.ACT_NONE:
            xin = empty_like_act(x.t)
            be.act_forward(in_act, x.t, xin)
            if w_needs and _WGRAD_PREACT:
                x_pre = xin                    # kept for the weight gradient (see the general branch below)
        T = torch.empty((n, hi, wi, cpad(16 * spec.cout)), dtype=torch.float32, device=x.t.device)
        be.conv_forward(layer.ensure_packed_taps(ctx.prec), xin, T, hi, wi, None, L.ACT_NONE, L.ACT_NONE, ctx.prec.prec, raw_out=True)
        be.convt4_gather(T, spec.cout, layer.bias.detach() if layer.bias is not None else None, act, out)
        del T, xin
        nch = 0
    elif layer.narrow:
        # T[n,h,w,(co,kw)] by the gather GEMM (vertical taps), then y = act(bias + sum_kw T[.., w+kw-pad, (co,kw)])
        T = torch.empty((n, ho, wo, cpad(spec.cout * spec.k)), dtype=torch.float32, device=x.t.device)
        be.conv_forward(layer.packed_fwd, x.t, T, ho, wo, None, L.ACT_NONE, in_act, ctx.prec.prec, raw_out=True)
        be.shift_sum(T, spec.cout, spec.k, spec.pad, spec.pad_mode, layer.bias.detach() if layer.bias is not None else None, act, out)
        del T
        nch = 0
    else:
        xin, fwd_in_act = x.t, in_act
        if in_act != L.ACT_NONE and _PREACT and ctx.prec.prec == L.PREC_BF16 and ctx.prec.is16 and x.t.dtype == ctx.prec.dtype:
            # The direct-to-LDS conv kernels cannot transform while staging (the DMA bypasses the registers), so a conv with an
            # input activation falls back to the register-staged kernel.  Materialise in_act(x) once instead (one elementwise
            # pass over the input) and run the fast kernel on it; backward still masks with the raw x (kept as it is).
            xin = empty_like_act(x.t)
            be.act_forward(in_act, x.t, xin)
            fwd_in_act = L.ACT_NONE
            if w_needs and _WGRAD_PREACT:
                # ... and the weight gradient has the same problem (an operand with a staged activation goes to the register-staged wgrad_kernel: 105 us per UNet
                # layer, 7 % of the 18-network step): keep in_act(x) for it -- one more activation-sized tensor per UNet conv, alive until its backward
                x_pre = xin
        x_split = x.split is not None and fwd_in_act == L.ACT_NONE and xin is x.t and getattr(be, 'supports_split', False) and \
            be.conv_takes_split(x.t, ctx.prec.prec, fwd_in_act, spec.pad_mode)
        assert x.values_stored or x_split, 'a split-only activation reached a convolution that needs its fp32 values'
        nch = be.conv_forward(layer.packed_fwd, x.split if x_split else xin, out, hq, wq, layer.bias.detach() if layer.bias is not None else None, act,
                              fwd_in_act, ctx.prec.prec, want_stats=stats and act == L.ACT_NONE, **({'in_split': True} if x_split else {}))
        del xin
    y = Act(out, spec.cout, x_needs or w_needs)
    y.norm_only = bool(stats)       # the caller's promise (see the docstring); norm_act's bias-gradient fusion relies on it
    if nch:
        y.stats = (nch, be.norm_ws_token())
    if not (x_needs or w_needs):
        return y
    if w_needs and act == L.ACT_NONE and layer.bias is not None and layer.bias.requires_grad:
        y.bias_grad = layer.bias.grad           # a following norm_act folds sum(dy) into its backward pass

    if w_needs:
        ctx.tape.use(layer.weight, layer.bias)

    def split_backward_ok(g_like: torch.Tensor, bias_done: bool) -> bool:
        """True if every consumer of dL/dy in backward_body() below reads the split copy (mirrors its branches one by one)"""
        if not (getattr(be, 'supports_split', False) and ops._SPLIT_ONLY_GRAD) or act != L.ACT_NONE or layer.narrow:
            return False
        if w_needs:
            if spec.kind == 'conv':
                if not be.wgrad_takes_split(g_like, x.t, layer.weight.grad, spec.k, spec.pad_mode, ctx.prec.prec):
                    return False
            elif not be.wgrad_takes_split(x.t, g_like, layer.weight.grad, spec.k, L.PAD_ZERO, ctx.prec.prec):
                return False
            if layer.bias is not None and layer.bias.requires_grad and not bias_done:
                return False
        if x_needs:
            if spec.kind == 'conv' and spec.pad_mode == L.PAD_REFLECT:
                return False
            if not be.conv_takes_split(g_like, ctx.prec.prec, L.ACT_NONE, L.PAD_ZERO):
                return False
        return True
    y.split_backward_ok = split_backward_ok

    def backward():
        try:
            backward_body()
        finally:
            if w_needs:
                ctx.tape.done(layer.weight, layer.bias)      # (also when no gradient arrived: this use contributes nothing more)

    def backward_body():
        g = y.grad
        gs = y.grad_split if (y.norm_only and getattr(be, 'supports_split', False)) else None       # split copy of g (norm_act's backward wrote it)
        stored = y.grad_values_stored
        y.grad, y.grad_split, y.grad_values_stored = None, None, True
        if g is None:
            return
        if not stored:
            # g only carries the geometry (the norm backward skipped its fp32 store): every branch below must take gs
            assert gs is not None and split_backward_ok(g, y.bias_done), 'split-only gradient reached a consumer that needs the fp32 values'
        if act != L.ACT_NONE:                       # epilogue activation: derivative from the saved output
            gp = empty_like_act(g)
            be.act_backward(act, g, y.t, gp)
            g, gs = gp, None
        xs = x.split if (x.split is not None and in_act == L.ACT_NONE and getattr(be, 'supports_split', False)) else None
        if w_needs and not x.values_stored:
            assert xs is not None and conv_reads_split_only(ctx, layer, x.t, in_act), 'split-only activation reached a weight gradient that needs fp32 values'
        if w_needs:
            if layer.narrow and spec.pad_mode == L.PAD_ZERO and getattr(be, 'wgrad_c4_applies', None) is not None and \
                    be.wgrad_c4_applies(g, x.t, layer.weight.grad, spec.k, 1, spec.pad, L.PAD_ZERO, L.ACT_NONE, in_act, ctx.prec.prec):
                # 7x7, 64 -> 3: one persistent kernel over (dL/dy: 4-channel patch, x: 64-channel tile), csrc/wgrad_c4.h
                be.conv_wgrad(g, x.t, layer.weight.grad, spec.k, 1, spec.pad, L.PAD_ZERO, L.ACT_NONE, in_act, ctx.prec.prec, True)
            elif layer.narrow and spec.pad_mode == L.PAD_ZERO:
                # D[.., (co,kw)] = dy shifted by kw; the weight gradient becomes a KH x 1 problem with Cout*KW rows
                D = torch.empty((n, ho, wo, cpad(spec.cout * spec.k)), dtype=g.dtype, device=g.device)
                be.shift_stack(g, spec.cout, spec.k, spec.pad, D)
                be.conv_wgrad(D, x.t, layer.weight.grad, spec.k, 1, spec.pad, L.PAD_ZERO, L.ACT_NONE, in_act, ctx.prec.prec, True, stack_kw=spec.k)
                del D
            elif spec.kind == 'conv':
                xw, xw_act = (x_pre, L.ACT_NONE) if x_pre is not None else (x.t, in_act)       # (x_pre: bf16 policy only, so no split copies are in play)
                sp = (gs is not None or xs is not None) and be.wgrad_takes_split(g, x.t, layer.weight.grad, spec.k, spec.pad_mode, ctx.prec.prec)
                be.conv_wgrad(gs if (sp and gs is not None) else g, xs if (sp and xs is not None) else xw, layer.weight.grad, spec.k, spec.stride, spec.pad,
                              spec.pad_mode, L.ACT_NONE, xw_act, ctx.prec.prec, True, **({'p_split': gs is not None, 'q_split': xs is not None} if sp else {}))
            else:
                xw, xw_act = (x_pre, L.ACT_NONE) if x_pre is not None else (x.t, in_act)
                sp = (gs is not None or xs is not None) and be.wgrad_takes_split(x.t, g, layer.weight.grad, spec.k, L.PAD_ZERO, ctx.prec.prec)
                be.conv_wgrad(xs if (sp and xs is not None) else xw, gs if (sp and gs is not None) else g, layer.weight.grad, spec.k, spec.stride, spec.pad,
                              L.PAD_ZERO, xw_act, L.ACT_NONE, ctx.prec.prec, True, **({'p_split': xs is not None, 'q_split': gs is not None} if sp else {}))
            if layer.bias is not None and layer.bias.requires_grad and not y.bias_done:
                be.channel_sum(g, spec.cout, layer.bias.grad, True)
        if x_needs and spec.kind == 'conv' and spec.pad_mode == L.PAD_REFLECT:
            # gradient w.r.t. the explicitly reflection-padded input, then fold the mirrored borders back (dl_reflect_fold)
            hp, wp = hi + 2 * spec.pad, wi + 2 * spec.pad
            dxp = torch.empty((n, hp, wp, x.t.shape[3]), dtype=g.dtype, device=g.device)
            be.conv_forward(layer.packed_dgrad, g, dxp, hp, wp, None, L.ACT_NONE, L.ACT_NONE, ctx.prec.prec)
            dx = torch.empty((n, hi, wi, x.t.shape[3]), dtype=g.dtype, device=g.device)
            be.reflect_fold(dxp, dx, spec.pad)
            del dxp
            if in_act != L.ACT_NONE:
                be.act_backward(in_act, dx, x.t, dx)
            x.add_grad(dx)
        elif x_needs:
            if spec.kind == 'conv' and spec.stride == 2:
                # four sub-pixel phases over a ceil(hi/2) x ceil(wi/2) grid; for odd sizes the kernels drop the outputs of the
                # odd phases that fall outside dx (torch accepts any tile size, so does this path)
                dq = ((hi + 1) // 2, (wi + 1) // 2)
            else:
                dq = (hi, wi)
            # x = act(norm(y')) of the layer in front and nothing else has contributed to its gradient yet: the store epilogue of this
            # data gradient also produces that norm's backward reductions (one pass over y' and dx saved); valid only while dx stays
            # the sole contribution (Act.add_grad drops it otherwise)
            fuse = x.bn_ctx if (x.grad is None and in_act == L.ACT_NONE) else None
            g_split = gs is not None and be.conv_takes_split(g, ctx.prec.prec, L.ACT_NONE, L.PAD_ZERO)
            # x already carries a gradient (a residual block's skip connection delivered it first, networks.py:509-513): the ResnetBlock-shape kernel adds it in
            # its store epilogue -- in place, every thread reads its 16 bytes before it writes them -- instead of a separate pass over both tensors (axpby)
            if x.grad is not None and x.grad_values_stored and in_act == L.ACT_NONE and not g_split and getattr(be, 'conv_forward_add', None) is not None \
                    and tuple(x.grad.shape) == (n, hi, wi, x.t.shape[3]) and be.conv_forward_add(layer.packed_dgrad, g, x.grad, x.grad, dq[0], dq[1], ctx.prec.prec):
                x.grad_stats = None          # (as Act.add_grad: what described the first contribution alone no longer describes the sum)
                x.grad_split = None
                return
            dx = torch.empty((n, hi, wi, x.t.shape[3]), dtype=g.dtype, device=g.device)
            nch = be.conv_forward(layer.packed_dgrad, gs if g_split else g, dx, dq[0], dq[1], None, L.ACT_NONE, L.ACT_NONE, ctx.prec.prec, bn=fuse,
                                  **({'in_split': True} if g_split else {}))
            if in_act != L.ACT_NONE:                # relu / lrelu keep the sign: mask from the un-activated input
                be.act_backward(in_act, dx, x.t, dx)
            x.add_grad(dx)
            if fuse is not None and nch:
                x.grad_stats = (nch, be.norm_ws_token())

    ctx.tape.record(backward)
    return y


def norm_act(ctx: Ctx, y: Act, norm: Optional[NormLayer], act: int = L.ACT_NONE, residual: Optional[Act] = None,
             out: Optional[torch.Tensor] = None, sole_reader: Optional['ConvLayer'] = None) -> Act:
    """z = act(norm(y)) (+ residual).  norm None = identity norm (norm='none').
    sole_reader: the caller's promise that the result is read by conv(ctx, z, sole_reader) and by NOTHING else (no residual use, no dropout, no
    concat, not returned): under the strict policy the fp32 values are then not stored when that conv reads the split copy everywhere."""
    be = ops.impl()
    own_out = out is None
    if out is None:
        out = empty_like_act(y.t)
    needs = ctx.tape is not None and (y.needs_grad or (residual is not None and residual.needs_grad))
    if norm is None:
        be.act_forward(act, y.t, out)
        if residual is not None:
            be.axpby(1.0, out, 1.0, residual.t, out)
        z = Act(out, y.C, needs)
        if needs:
            def backward_plain():
                g = z.grad
                z.grad = None
                if g is None:
                    return
                if residual is not None and residual.needs_grad:
                    residual.add_grad(g)
                if y.needs_grad:
                    dy = empty_like_act(g)
                    # out = act(y) + res : act' from act(y) = out - res is not available; recompute from y's sign (relu family)
                    be.act_backward(act, g, y.t if act in (L.ACT_RELU, L.ACT_LRELU) else out, dy)
                    y.add_grad(dy)
            ctx.tape.record(backward_plain)
        return z

    m = norm.module
    gamma = m.weight.detach() if m is not None else None
    beta = m.bias.detach() if m is not None else None
    scope = L.NORM_INSTANCE if ctx.per_sample_norm else norm.scope
    rm = rv = None
    momentum = -1.0
    if m is not None and ctx.training and m.training and m.track_running_stats and m.running_mean is not None and scope == L.NORM_BATCH:
        rm, rv, momentum = m.running_mean, m.running_var, (m.momentum if m.momentum is not None else 0.1)
        m.num_batches_tracked += 1
    ext = y.stats[0] if (y.stats is not None and y.stats[1] == be.norm_ws_token()) else 0
    # strict policy, stand-alone output: the kernel also writes the split copy its consumers (conv forward / weight gradient) read instead
    # of splitting the fp32 values themselves for every tap (one extra 4-byte store per element here; csrc/conv_x3.h)
    zs = None
    if own_out and ctx.prec.prec == L.PREC_BF16X3 and out.dtype == torch.float32 and getattr(be, 'supports_split', False):
        zs = torch.empty(out.shape, dtype=torch.float32, device=out.device)
    split_only = zs is not None and sole_reader is not None and conv_reads_split_only(ctx, sole_reader, out)
    kwf = {}
    if zs is not None:
        kwf['z_split'] = zs
    if split_only:
        kwf['store_z'] = False
    stats = be.norm_forward(y.t, out, norm.C, scope, act, gamma, beta, rm, rv, momentum, residual.t if residual is not None else None,
                            ext_nchunks=ext, **kwf)
    z = Act(out, y.C, needs)
    z.split = zs
    z.values_stored = not split_only
    if not needs:
        return z
    if residual is None and act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU) and y.t.dtype == torch.bfloat16 and y.needs_grad:
        z.bn_ctx = (y.t, stats, act)

    track_affine = m is not None and m.weight.requires_grad
    if track_affine:
        ctx.tape.use(m.weight, m.bias)

    def backward():
        try:
            backward_body()
        finally:
            if track_affine:
                ctx.tape.done(m.weight, m.bias)

    def backward_body():
        g = z.grad
        z.grad = None
        gs, z.grad_stats = z.grad_stats, None
        if g is None:
            return
        ext_b = gs[0] if (gs is not None and gs[1] == be.norm_ws_token()) else 0
        if residual is not None and residual.needs_grad:
            residual.add_grad(g)
        affine = m is not None and m.weight.requires_grad
        if y.needs_grad or affine:
            dy = empty_like_act(g)
            # dy is y's ONLY gradient contribution: promised by the producer (conv(stats=True) -> norm_only), not inferred from the moment's
            # y.grad (a consumer recorded EARLIER on the tape would contribute later and its share of the bias gradient would be lost)
            fuse_bias = y.bias_grad is not None and y.norm_only and y.grad is None and not y.bias_done
            # dy is the ONLY gradient the conv output y ever receives (norm_only): its producer's data / weight gradient kernels can read a split copy
            dys = None
            if y.norm_only and y.needs_grad and y.grad is None and ctx.prec.prec == L.PREC_BF16X3 and dy.dtype == torch.float32 and \
                    getattr(be, 'supports_split', False):
                dys = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
            # ... and when its whole backward can run from that copy, the fp32 values are not stored at all (one 4-byte store per element less)
            split_only = dys is not None and y.split_backward_ok is not None and y.split_backward_ok(dy, fuse_bias or y.bias_done)
            kw = {}
            if dys is not None:
                kw['dy_split'] = dys
            if split_only:
                kw['store_dy'] = False
            be.norm_backward(g, y.t, dy, stats, norm.C, scope, act, gamma, m.weight.grad if affine else None, m.bias.grad if affine else None,
                             y.bias_grad if fuse_bias else None, ext_nchunks=ext_b, **kw)
            if fuse_bias:
                y.bias_done = True
            if y.needs_grad:
                y.add_grad(dy)
                y.grad_split = dys
                y.grad_values_stored = not split_only

    ctx.tape.record(backward)
    return z


_DROPOUT_COUNTER = [0]


def _rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def dropout(ctx: Ctx, x: Act, p: float, in_place: bool = False) -> Act:
    """nn.Dropout(p) in training mode.  The mask is a function of a per-call seed (torch's CPU RNG seeds the stream, so
    torch.manual_seed() makes runs repeatable); backward re-applies the same mask to the gradient."""
    be = ops.impl()
    _DROPOUT_COUNTER[0] += 1
    # the rank is mixed in: replicas seeded identically (parameters are broadcast anyway) must not drop the same units on different tiles
    seed = (int(torch.initial_seed()) * 1000003 + 7919 * _rank() + _DROPOUT_COUNTER[0]) & (2 ** 63 - 1)
    assert not (in_place and ctx.tape is not None and x.needs_grad), 'in-place dropout would let the gradient through unmasked: use in_place=False when training'
    out = x.t if in_place else empty_like_act(x.t)
    be.dropout(x.t, out, p, seed)
    needs = ctx.tape is not None and x.needs_grad
    z = x if in_place else Act(out, x.C, needs)
    if needs and not in_place:
        def backward():
            g = z.grad
            z.grad = None
            if g is None:
                return
            dx = empty_like_act(g)
            be.dropout(g, dx, p, seed)
            x.add_grad(dx)
        ctx.tape.record(backward)
    return z


def concat_channels(ctx: Ctx, parts: List[Act], out: Optional[torch.Tensor] = None) -> Act:
    """torch.cat(parts, 1) for small channel counts (D inputs: cat(cond, image), DeepLIIF_model.py:223).
    out: an already ZEROED [n, h, w, cpad(sum C)] tensor to fill instead of a new one (a batch slice of a larger buffer: pair_batch below)"""
    be = ops.impl()
    ctot = sum(p.C for p in parts)
    n, h, w, _ = parts[0].t.shape
    if out is None:
        out = torch.zeros((n, h, w, cpad(ctot)), dtype=parts[0].t.dtype, device=parts[0].t.device)
    else:
        assert tuple(out.shape) == (n, h, w, cpad(ctot)) and out.dtype == parts[0].t.dtype
    c0 = 0
    for p in parts:
        be.copy_channels(p.t, 0, out, c0, p.C)
        c0 += p.C
    needs = ctx.tape is not None and any(p.needs_grad for p in parts)
    z = Act(out, ctot, needs)
    if needs:
        def backward():
            g = z.grad
            z.grad = None
            if g is None:
                return
            c = 0
            for p in parts:
                if p.needs_grad:
                    gp = torch.zeros(p.t.shape, dtype=g.dtype, device=g.device)
                    be.copy_channels(g, c, gp, 0, p.C)
                    p.add_grad(gp)
                c += p.C
        ctx.tape.record(backward)
    return z


def act_op(ctx: Ctx, x: Act, act: int) -> Act:
    """a stand-alone activation layer (nn.Sigmoid of the attention gate, att_unet.py:100-104): y = act(x)"""
    be = ops.impl()
    out = empty_like_act(x.t)
    be.act_forward(act, x.t, out)
    needs = ctx.tape is not None and x.needs_grad
    y = Act(out, x.C, needs)
    if needs:
        def backward():
            g = y.grad
            y.grad = None
            if g is None:
                return
            dx = empty_like_act(g)
            be.act_backward(act, g, y.t, dx)
            x.add_grad(dx)
        ctx.tape.record(backward)
    return y


def gate(ctx: Ctx, x: Act, psi: Act, out: Optional[torch.Tensor] = None) -> Act:
    """Attention_block.forward's last line (att_unet.py:113-115): x * psi, psi a ONE-channel map broadcast over the channels of x.
    `out` may be the first half of a concat buffer (torch.cat((x_gated, d), dim=1) is then zero-copy, like the UNet skips)."""
    be = ops.impl()
    assert psi.C == 1 and psi.t.shape[:3] == x.t.shape[:3]
    if out is None:
        out = empty_like_act(x.t)
    be.gate_forward(x.t, psi.t, out)
    needs = ctx.tape is not None and (x.needs_grad or psi.needs_grad)
    z = Act(out, x.C, needs)
    if needs:
        def backward():
            g = z.grad
            z.grad = None
            if g is None:
                return
            dx = empty_like_act(x.t) if x.needs_grad else None
            dpsi = empty_like_act(psi.t)
            be.gate_backward(g, x.t, psi.t, dx, dpsi)
            if dx is not None:
                x.add_grad(dx)
            if psi.needs_grad:
                psi.add_grad(dpsi)
        ctx.tape.record(backward)
    return z


def weighted_sum(ctx: Ctx, parts: List[Act], weights: List[float]) -> Act:
    """stack([w_i * x_i]).sum(0)  (DeepLIIF_model.py:203, 258-262)."""
    be = ops.impl()
    out = empty_like_act(parts[0].t)
    be.axpby(weights[0], parts[0].t, 0.0, None, out)
    for p, w in zip(parts[1:], weights[1:]):
        be.axpby(1.0, out, w, p.t, out)
    needs = ctx.tape is not None and any(p.needs_grad for p in parts)
    z = Act(out, parts[0].C, needs)
    if needs:
        def backward():
            g = z.grad
            z.grad = None
            if g is None:
                return
            for p, w in zip(parts, weights):
                if p.needs_grad:
                    gp = empty_like_act(g)
                    be.axpby(w, g, 0.0, None, gp)
                    p.add_grad(gp)
        ctx.tape.record(backward)
    return z


def maxpool2(ctx: Ctx, x: Act) -> Act:
    """nn.MaxPool2d(kernel_size=2, stride=2) (torchvision VGG19 features, networks.py:698-731)."""
    be = ops.impl()
    n, h, w, cp = x.t.shape
    out = torch.empty((n, h // 2, w // 2, cp), dtype=x.t.dtype, device=x.t.device)
    be.maxpool2_forward(x.t, out)
    needs = ctx.tape is not None and x.needs_grad
    y = Act(out, x.C, needs)
    if needs:
        def backward():
            g = y.grad
            y.grad = None
            if g is None:
                return
            dx = empty_like_act(x.t)
            be.maxpool2_backward(x.t, g, dx)
            x.add_grad(dx)
        ctx.tape.record(backward)
    return y


def loss_op(ctx: Ctx, kind: int, x: Act, target: Optional[Act], target_const: float, weight: float, loss_out: torch.Tensor,
            out_scale: float = 1.0, accumulate: bool = False) -> None:
    """loss_out[0] (+)= out_scale * mean loss (unweighted by `weight`, as the reference logs it); if x needs grad, d(weight*loss)/dx is queued."""
    be = ops.impl()
    needs = ctx.tape is not None and x.needs_grad
    grad = empty_like_act(x.t) if needs else None
    be.loss(kind, x.t, target.t if target is not None else None, target_const, x.C, loss_out, grad, weight, out_scale, accumulate)
    if needs:
        def backward():
            x.add_grad(grad)
        ctx.tape.record(backward)


def loss_op_halves(ctx: Ctx, kind: int, x: Act, consts, weight: float, loss_outs) -> None:
    """Two loss_op calls on the two batch halves of x -- x[:n/2] against consts[0] into loss_outs[0], x[n/2:] against consts[1] into loss_outs[1] -- with ONE
    gradient tensor for x: the discriminator pass over cat(fake batch, real batch) (models.DeepLIIFModel.backward_D).  Each half is a mean over its own
    elements, exactly what two separate passes compute."""
    be = ops.impl()
    n = x.t.shape[0]
    assert n % 2 == 0
    needs = ctx.tape is not None and x.needs_grad
    grad = empty_like_act(x.t) if needs else None
    for half, (c, lo) in enumerate(zip(consts, loss_outs)):
        sl = slice(half * (n // 2), (half + 1) * (n // 2))
        be.loss(kind, x.t[sl], None, c, x.C, lo, grad[sl] if needs else None, weight, 1.0, False)
    if needs:
        def backward():
            x.add_grad(grad)
        ctx.tape.record(backward)


def upsample2(ctx: Ctx, x: Act) -> Act:
    """nn.Upsample(scale_factor=2, mode='nearest') (ResnetGenerator --upsample resize_conv, networks.py:409-411)"""
    be = ops.impl()
    n, h, w, cp = x.t.shape
    out = torch.empty((n, 2 * h, 2 * w, cp), dtype=x.t.dtype, device=x.t.device)
    be.upsample2(x.t, out)
    needs = ctx.tape is not None and x.needs_grad
    z = Act(out, x.C, needs)
    if needs:
        def backward():
            g = z.grad
            z.grad = None
            if g is None:
                return
            dx = empty_like_act(x.t)
            be.upsample2(g, dx, backward=True)
            x.add_grad(dx)
        ctx.tape.record(backward)
    return z


def kldiv_op(ctx: Ctx, x: Act, teacher: Act, weight: float, loss_out: torch.Tensor) -> None:
    """DeepLIIFKD_model.py:313-336: loss_out[0] = KLDivLoss(batchmean)(LogSoftmax(x.view(1,1,-1)), Softmax(teacher.view(1,1,-1))) (unweighted, as the
    reference logs it); if x needs grad, d(weight * loss)/dx = weight * (softmax(x) - softmax(teacher)) is queued.  The teacher gets no gradient."""
    be = ops.impl()
    needs = ctx.tape is not None and x.needs_grad
    grad = empty_like_act(x.t) if needs else None
    be.kldiv(x.t, teacher.t, x.C, loss_out, grad, weight)
    if needs:
        def backward():
            x.add_grad(grad)
        ctx.tape.record(backward)


def to_engine(x_nchw: torch.Tensor, prec: Precision) -> Act:
    """NCHW fp32 (the reference's tensors) -> engine NHWC with padded channels."""
    x = x_nchw.detach().contiguous().float()
    n, c, h, w = x.shape
    a = new_act(n, h, w, c, prec, x.device)
    ops.impl().nchw_to_nhwc(x, a.t, 0, a.t.shape[3])
    return a


def from_engine(a: Act) -> torch.Tensor:
    n, h, w, _ = a.t.shape
    out = torch.empty((n, a.C, h, w), dtype=torch.float32, device=a.t.device)
    ops.impl().nhwc_to_nchw(a.t, 0, out)
    return out
