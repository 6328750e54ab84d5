"""Tensor-level wrappers over the C ABI (include/deepliif_hip.h).

Every function takes torch tensors that live on the GPU (PyTorch is only the allocator / stream owner here), extracts
raw pointers + strides, and launches the hand-written gfx950 kernels on torch's *current* HIP stream.  There is no CPU or
PyTorch-op fallback: a non-CUDA tensor raises.

Engine tensors are NHWC views `[N, H, W, C]` with unit channel stride; `stride(2)` is the pixel stride, so a tensor may be
a channel slice of a wider (concat) buffer.

`_impl` is the single dispatch seam: the product always uses HipBackend; the CPU test-suite swaps in an emulation to
exercise the host logic without a GPU (tests/fake_backend.py).
"""
from __future__ import annotations

import ctypes as C
import os
import contextlib
import threading
from typing import Optional

import torch

from . import _lib as L
from .geometry import (GatherPlan, NUM_CUS, WGRAD_C4_PARTS, choose_splitk, choose_wgrad_batch_splitk, choose_wgrad_splitk, fill_conv_desc, fill_pack_desc,
                       wgrad_batch_shape, wgrad_c4_ok, wgrad_fast_path)


def dl_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.DL_F32
    if t.dtype == torch.bfloat16 or t.dtype == torch.float16:      # "the 16-bit type of the library": HipBackend._need_cuda checks that it is THIS library's
        return L.DL_BF16
    raise TypeError(f'engine tensors are fp32 or bf16 / fp16, got {t.dtype}')


def pstride(t: torch.Tensor) -> int:
    assert t.dim() == 4 and t.stride(3) == 1, 'NHWC view with unit channel stride expected'
    ps = t.stride(2)
    assert t.stride(1) == t.shape[2] * ps and t.stride(0) == t.shape[1] * t.shape[2] * ps, 'pixels must be densely strided'
    return ps


_SHARED_SCRATCH = os.environ.get('DL_SHARED_SCRATCH', '0') == '1'
# DL_BNSTATS=1: the data gradient's store epilogue also produces the following norm backward's reductions (dl_conv_forward_bnstats).  OFF by
# default: same-box A/B of the training step (r02) 104.9-105.0 ms with it vs 103.8-104.0 without -- the y tile read sits exposed in the
# epilogue of a 1-workgroup-per-CU kernel (+37 us per fused ResnetBlock launch) and costs what the saved pass (49 us) was worth.
_BNSTATS = os.environ.get('DL_BNSTATS', '0') == '1'
_CONV_ADD = os.environ.get('DL_CONV_ADD', '1') != '0'      # A/B switch (Python side only): 0 = a second gradient contribution is always added by dl_axpby
_NO_WGRAD_C4 = os.environ.get('DL_NO_WGRAD_C4', '0') == '1'
_NO_X3_GLDS = os.environ.get('DL_NO_X3_GLDS') is not None           # A/B switch: the strict policy on the round-1 register-staged kernels (csrc reads the same variable)
_X3_ACTS = (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU)
_SPLIT_ONLY_GRAD = os.environ.get('DL_NO_SPLIT_ONLY_GRAD', '0') != '1'     # A/B switch: 1 = the norm backward always stores the fp32 gradient next to its split copy
_NO_C4_X3 = 'DL_NO_C4_X3' in os.environ              # A/B switch: the strict 7x7 stem / head on the general x3 kernels (csrc/conv_x3.h, wgrad_x3.h)
_WGRAD_DEFER = os.environ.get('DL_WGRAD_DEFER', '1') != '0'            # A/B switch: 0 = every weight gradient reduces its slabs right behind the split-K kernel (rounds 1-3)
# slab arena of the deferred reduction, per scratch state (thread / branch stream): grown on demand, never below this.  r04 reserved 4 GB per state whatever the
# model (~16 GB with three branch streams, also for an ngf = 8 fixture); since the batched weight gradient (below) a Resnet-9 pass at batch 8 writes
# ~0.9 GB of slabs (r04: 1.9 GB), and an arena that is too small only costs an extra reduction launch
_WGRAD_ARENA_MB = int(os.environ.get('DL_WGRAD_ARENA_MB', '256'))
# DL_WGRAD_BATCH=0: A/B switch -- every weight gradient launches its own split-K kernel as in round 4.  Default: inside a backward pass the layers of the
# ResnetBlock shape (the only shape a network repeats: 18 of a Resnet-9's 23 convolutions) are QUEUED with their operands and computed by one launch per
# network (dl_conv_wgrad_multi) at the network's tape marker / the end of the pass
_WGRAD_BATCH = os.environ.get('DL_WGRAD_BATCH', '1') != '0'
_NO_NARROW_ROLL = os.environ.get('DL_NO_NARROW_ROLL', '0') == '1'      # A/B switch: 1 = head forward through dl_conv_forward(raw) + dl_shift_sum (round 1)
_SHARED_STATE: dict = {}


# The raw handle of torch's current HIP stream, asked from torch's C layer directly: torch.cuda.current_stream() builds a Stream object and resolves the device
# index through is_available() / os.environ on every call -- ~7 us, several times per launch, 4-5 ms of host time per training step (r05, tools/host_profile.py).
# (tests that replace torch.cuda.current_stream by a recorder switch the shortcut off)
_FAST_STREAM = hasattr(torch._C, '_cuda_getCurrentRawStream') and hasattr(torch._C, '_cuda_getDevice')


def _current_stream_handle(dev_index=None) -> int:
    if _FAST_STREAM and torch.cuda.is_initialized():
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice() if dev_index is None else dev_index)
    return torch.cuda.current_stream(dev_index).cuda_stream if dev_index is not None else torch.cuda.current_stream().cuda_stream


class Workspace:
    """Grow-only fp32 scratch buffers, one per purpose AND PER THREAD.  Within a thread all kernels run in stream order, so a buffer
    can be reused by the next call of the same kind.  Across threads nothing may be shared: one C call launches e.g. the split-K conv
    and then its slab reduction, and another thread (the reference drives different nets from dask worker threads,
    deepliif/models/__init__.py:283-334; ctypes releases the GIL) could launch its own conv into the same slab in between."""

    def __init__(self):
        self._tls = threading.local()
        # HIP streams registered as a model's BRANCH streams -- process-wide: a model caches its streams once, so a thread other than the
        # one that ran the first forward (a worker-thread trainer, validation then training) must recognise them too, or its three concurrently
        # running branches would share one slab arena and one statistics workspace
        self._branch_ids = set()
        self._branch_lock = threading.Lock()

    def _thread_state(self):
        # DL_SHARED_SCRATCH=1 restores the process-wide buffers: ONLY for demonstrating the hazard (tests/test_gpu_networks.py,
        # test_inference_seam_is_thread_safe fails with it)
        return _SHARED_STATE if _SHARED_SCRATCH else self._tls.__dict__

    def _state(self):
        """scratch state of the calling thread -- or, while one of a model's BRANCH streams is current (branch_streams_on), of that stream: two
        streams of one thread run concurrently on the GPU, so they may not share a slab, a statistics workspace or a slab arena"""
        st = self._thread_state()
        ids = self._branch_ids
        # (while models.StepGraph captures -- one stream by construction -- the capture stream keeps the thread's state whatever its handle: torch hands out
        # POOLED streams, 32 per device, so the capture stream can carry the handle of a branch stream some earlier model registered)
        if ids and not st.get('capturing'):
            h = _current_stream_handle()
            if h in ids:                       # (any other stream -- torch's default stream, a graph-capture stream -- keeps the thread's own state, as before)
                sub = st.setdefault('streams', {}).get(h)
                if sub is None:
                    sub = st['streams'][h] = {'stream_obj': torch.cuda.current_stream()}
                st = sub
        if 'bufs' not in st:
            st['bufs'], st['norm_ws_token'] = {}, 0
        return st

    def branch_streams_on(self, streams):
        """register the branch streams of a model (models.BaseModel._branch_streams): work launched while one of them is current gets its own scratch state"""
        with self._branch_lock:
            self._branch_ids = self._branch_ids | {s.cuda_stream for s in streams}          # (replaced, never mutated: readers need no lock)

    def forget_branch_streams(self):
        """drop every registration and this thread's per-stream scratch states (tests; a model that is gone leaves its streams' states behind)"""
        with self._branch_lock:
            self._branch_ids = set()
        self._thread_state().pop('streams', None)

    def stream_states(self):
        """every per-stream state of this thread (empty unless branch_streams_on() was called)"""
        return list(self._thread_state().get('streams', {}).values())

    # Counts every REPLACEMENT of a scratch buffer or slab arena that already existed (process-wide, monotonic).  A captured step (models.StepGraph) has the
    # addresses of these buffers baked into its kernel nodes; when a later eager step needs more scratch, the old buffer goes back to the allocator and
    # every replay would write into freed memory: StepGraph compares this counter and re-captures.
    realloc_generation = 0

    def get(self, name: str, nfloats: int, device) -> torch.Tensor:
        bufs = self._state()['bufs']
        b = bufs.get(name)
        if b is None or b.numel() < nfloats or b.device != device:
            if b is not None:
                Workspace.realloc_generation += 1
            b = torch.empty(max(int(nfloats), 1024), dtype=torch.float32, device=device)
            bufs[name] = b
        return b

    # the 'norm_ws' buffer carries conv-epilogue statistics to the following norm: a per-thread token says whether they are still there
    def norm_token(self) -> int:
        return self._state()['norm_ws_token']

    def bump_norm_token(self):
        self._state()['norm_ws_token'] += 1


WS = Workspace()


class PackedWeights:

    def __init__(self, plan: GatherPlan, device, with_lo: bool):
        n = plan.rows_pad * plan.kstride
        self.plan = plan
        # raw 16-bit words: bfloat16 or IEEE half, whichever library packs them (dl_pack_weights of libdeepliif_hip.so / libdeepliif_hip_f16.so)
        self.hi = torch.empty(n, dtype=torch.int16, device=device)
        self.lo = torch.empty(n, dtype=torch.int16, device=device) if with_lo else None


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_LAUNCH = threading.local()


def _stream(t: Optional[torch.Tensor] = None):
    """torch's current HIP stream OF THE OPERANDS' DEVICE (not of whatever device happens to be current): nets live on
    cuda:gpu_ids[0] (base_model.py:38), and a launch on another device's stream with these pointers would fault or silently rely on
    peer access.  The launching thread's current device is switched to the operands' device if it differs (the reference's CLI does
    torch.cuda.set_device(gpu_ids[0]) once, cli.py:250-256; the model classes here do the same in BaseModel.__init__)."""
    dev = t.device if t is not None else getattr(_LAUNCH, 'dev', None)
    if dev is None:
        return C.c_void_p(_current_stream_handle())
    if dev.index is not None and dev.index != (torch._C._cuda_getDevice() if _FAST_STREAM else torch.cuda.current_device()):
        torch.cuda.set_device(dev)
    return C.c_void_p(_current_stream_handle(dev.index))


# ---- the 16-bit format of the calling thread's engine work: 'bf16' (libdeepliif_hip.so) unless an fp16 inference forward is running (half_mode);
# impl() hands out the backend bound to that library, _need_cuda refuses 16-bit tensors of the OTHER format (their bits would be misread silently)
_HALF = threading.local()
H16_DTYPE = {'bf16': torch.bfloat16, 'fp16': torch.float16}


def half_format() -> str:
    return getattr(_HALF, 'name', 'bf16')


@contextlib.contextmanager
def half_mode(name: str):
    """engine work of this thread inside the block runs on the library of that 16-bit format (engine.Precision.half)"""
    if name not in H16_DTYPE:
        raise ValueError(f'unknown 16-bit format {name!r} (bf16 | fp16)')
    prev = half_format()
    _HALF.name = name
    try:
        yield
    finally:
        _HALF.name = prev


def _need_cuda(*ts):
    dev = None
    other = torch.float16 if half_format() == 'bf16' else torch.bfloat16
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise L.HipLibraryError('deepliif_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback')
        if t.dtype == other:
            raise L.HipLibraryError(f'a {t.dtype} tensor reached the {half_format()} library: engine tensors of the two 16-bit formats cannot be mixed '
                                    f'(ops.half_mode / engine.Precision)')
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise L.HipLibraryError(f'operands of one kernel launch live on different devices ({dev} and {t.device})')
    _LAUNCH.dev = dev


class HipBackend:
    """The product path: ctypes calls into libdeepliif_hip.so (half = 'bf16') or libdeepliif_hip_f16.so (half = 'fp16', inference only)."""

    def __init__(self, half: str = 'bf16'):
        self.half = half
        self.lib = L.load(half)
        self._last_conv_desc = None

    def check(self, rc, what):
        L.check(rc, what, self.lib)

    @property
    def last_conv_kernel(self) -> str:
        """name of the kernel the last dl_conv_forward of this backend dispatched to (dl_conv_kernel_name)"""
        d = self._last_conv_desc
        return '' if d is None else self.lib.dl_conv_kernel_name(C.byref(d)).decode()

    def norm_ws_token(self) -> int:
        return WS.norm_token()           # bumped whenever this thread's 'norm_ws' workspace is overwritten

    # ---- weights
    def pack_weights(self, packed: PackedWeights, src: torch.Tensor):
        _need_cuda(src, packed.hi)
        assert src.dtype == torch.float32 and src.is_contiguous()
        d = fill_pack_desc(packed.plan, src.shape[0], src.shape[1], src.shape[2])
        d.KH = src.shape[2]
        self.check(self.lib.dl_pack_weights(C.byref(d), _ptr(src), _ptr(packed.hi), _ptr(packed.lo), _stream()), 'dl_pack_weights')

    def pack_batch_build(self, jobs):
        """jobs: [(PackedWeights, fp32 master weight)] -> device table of dl_pack_job records (built once, see dl_pack_weights_batch)"""
        jb = int(self.lib.dl_pack_job_bytes())
        host = (C.c_ubyte * (jb * len(jobs)))()
        base = C.addressof(host)
        for i, (packed, src) in enumerate(jobs):
            _need_cuda(src, packed.hi)
            assert src.dtype == torch.float32 and src.is_contiguous()
            d = fill_pack_desc(packed.plan, src.shape[0], src.shape[1], src.shape[2])
            d.KH = src.shape[2]
            self.check(self.lib.dl_pack_job_fill(C.byref(d), _ptr(src), _ptr(packed.hi), _ptr(packed.lo), C.c_void_p(base + i * jb)),
                    'dl_pack_job_fill')
        nblocks = int(self.lib.dl_pack_batch_blocks(C.c_void_p(base), len(jobs), None))
        if nblocks < 0:
            self.check(nblocks, 'dl_pack_batch_blocks')
        tab = (C.c_int32 * (2 * max(nblocks, 1)))()
        self.check(min(int(self.lib.dl_pack_batch_blocks(C.c_void_p(base), len(jobs), C.c_void_p(C.addressof(tab)))), 0), 'dl_pack_batch_blocks')
        dev = jobs[0][1].device
        return (torch.frombuffer(host, dtype=torch.uint8).clone().to(dev), torch.frombuffer(tab, dtype=torch.int32).clone().to(dev), nblocks)

    def pack_batch_run(self, table, count: int):
        jobs_dev, tab_dev, nblocks = table
        _need_cuda(jobs_dev, tab_dev)
        self.check(self.lib.dl_pack_weights_batch(_ptr(jobs_dev), _ptr(tab_dev), nblocks, _stream()), 'dl_pack_weights_batch')

    # ---- convolution forward / data-gradient (gather GEMM)
    def conv_forward(self, packed: PackedWeights, x: torch.Tensor, out: torch.Tensor, hq: int, wq: int, bias: Optional[torch.Tensor],
                     act: int, in_act: int, prec: int, splitk: Optional[int] = None, raw_out: bool = False, want_stats: bool = False,
                     bn=None, in_split: bool = False):
        """raw_out: `out` is an fp32 [N,Ho,Wo,Co] tensor that receives the raw accumulators (narrow-Cout path).
        want_stats: ask the kernel to also leave the per-(image, channel) partial sums of `out` at the start of the shared
        normalisation workspace; returns the chunk count to hand to norm_forward(ext_nchunks=...) -- 0 when the dispatch for
        this layer cannot produce them (the caller then runs the stand-alone statistics pass).
        bn = (y, stats, act): `out` is dz for the layer z = act(norm(y)) with statistics `stats` (norm_forward's result) and this conv is its
        only contribution: let the store epilogue also produce the reductions of that norm's backward (dl_conv_forward_bnstats); returns
        the chunk count for norm_backward(ext_nchunks=...), 0 when the dispatch cannot (the norm then makes its own pass)."""
        _need_cuda(x, out, bias)
        if raw_out:
            plan = packed.plan
            n, hi, wi, cp = x.shape
            _, ho, wo, cop = out.shape
            assert out.dtype == torch.float32 and out.is_contiguous()
            d = fill_conv_desc(plan, n, hi, wi, pstride(x), ho, wo, cop, cop, hq, wq, dl_dtype(x), prec, L.ACT_NONE, in_act, 0, 1, 1)
            self.check(self.lib.dl_conv_forward(C.byref(d), _ptr(x), _ptr(packed.hi), _ptr(packed.lo), None, None, _ptr(out), None, _stream()),
                    'dl_conv_forward(raw)')
            return 0
        plan = packed.plan
        n, hi, wi, cp = x.shape
        assert cp == plan.cc_pad, (cp, plan.cc_pad)
        _, ho, wo, cop = out.shape
        auto_split = splitk is None
        if auto_split:
            splitk = choose_splitk(plan, n, hq, wq, cop)
        d = fill_conv_desc(plan, n, hi, wi, pstride(x), ho, wo, cop, pstride(out), hq, wq, dl_dtype(x), prec, act, in_act,
                           0 if bias is None else bias.numel(), splitk)
        d.in_split = 1 if in_split else 0           # x is the producer-written split copy of the fp32 activations (norm_forward(z_split=...))
        assert dl_dtype(out) == d.in_dtype
        if auto_split and splitk > 1 and plan.cc_real <= 4:
            # the 4-channel patch kernel (csrc/conv_c4.h) has no split-K form and needs none (its K is 7 steps): prefer it when it applies
            d.splitk = 1
            if self.lib.dl_conv_kernel_name(C.byref(d)).decode().startswith('conv_c4_patch'):
                splitk = 1
            else:
                d.splitk = splitk
        self._last_conv_desc = d          # diagnostic (bench.py roofline label): the kernel name is asked from the library only when somebody reads last_conv_kernel
        slab = WS.get('conv_slab', splitk * n * ho * wo * cop, x.device) if splitk > 1 else None
        nch, part = 0, None
        if want_stats:
            nch = int(self.lib.dl_conv_stats_chunks(C.byref(d)))
            if nch > 0:
                nd = self._norm_desc(out, cop, L.NORM_BATCH, L.ACT_NONE, -1.0, 8, 8)     # only N/H/W/Cp matter for the size
                nd.ext_nchunks = nch
                part = WS.get('norm_ws', self.lib.dl_norm_ws_floats(C.byref(nd)), x.device)
                WS.bump_norm_token()
        if bn is not None and bias is None and splitk == 1 and not want_stats and _BNSTATS:
            by, bstats, bact = bn
            nch = int(self.lib.dl_conv_bnstats_chunks(C.byref(d)))
            if nch > 0 and by.dtype == out.dtype == torch.bfloat16 and by.shape == out.shape:
                _need_cuda(by, bstats)
                nd = self._norm_desc(out, cop, L.NORM_BATCH, L.ACT_NONE, -1.0, 8, 8)     # only N/H/W/Cp matter for the size
                nd.ext_nchunks = nch
                part = WS.get('norm_ws', self.lib.dl_norm_ws_floats(C.byref(nd)), x.device)
                WS.bump_norm_token()
                b = L.ConvBnStats(by.data_ptr(), pstride(by), bact, bstats[0].data_ptr(), bstats[1].data_ptr(), bstats[2].data_ptr(),
                                  bstats[3].data_ptr())
                self.check(self.lib.dl_conv_forward_bnstats(C.byref(d), _ptr(x), _ptr(packed.hi), _ptr(packed.lo), _ptr(out), _ptr(part),
                                                         C.byref(b), _stream()), 'dl_conv_forward_bnstats')
                return nch
            nch = 0
        self.check(self.lib.dl_conv_forward(C.byref(d), _ptr(x), _ptr(packed.hi), _ptr(packed.lo), _ptr(bias), _ptr(out), _ptr(slab),
                                         _ptr(part), _stream()), 'dl_conv_forward')
        return nch

    def conv_forward_add(self, packed: PackedWeights, x: torch.Tensor, addend: torch.Tensor, out: torch.Tensor, hq: int, wq: int, prec: int) -> bool:
        """out = conv(x) + addend in the conv's store epilogue (dl_conv_forward_add; `out` may be `addend`).  Returns False -- nothing launched -- when the kernel
        the dispatch picks for this layer has no fused form: the caller then runs conv_forward + axpby as before."""
        if not _CONV_ADD or x.dtype != torch.bfloat16 or addend.dtype != torch.bfloat16 or addend.shape != out.shape:
            return False
        _need_cuda(x, addend, out)
        plan = packed.plan
        n, hi, wi, cp = x.shape
        _, ho, wo, cop = out.shape
        d = fill_conv_desc(plan, n, hi, wi, pstride(x), ho, wo, cop, pstride(out), hq, wq, dl_dtype(x), prec, L.ACT_NONE, L.ACT_NONE, 0, 1)
        if not self.lib.dl_conv_add_supported(C.byref(d)):
            return False
        self._last_conv_desc = d
        self.check(self.lib.dl_conv_forward_add(C.byref(d), _ptr(x), _ptr(packed.hi), _ptr(packed.lo), _ptr(addend), pstride(addend), _ptr(out), _stream()),
                'dl_conv_forward_add')
        return True

    # ---- weight gradient
    def conv_wgrad(self, P: torch.Tensor, Q: torch.Tensor, grad: torch.Tensor, k: int, step: int, pad: int, pad_mode: int,
                   p_act: int, q_act: int, prec: int, accumulate: bool, splitk: Optional[int] = None, stack_kw: int = 0, p_split: bool = False,
                   q_split: bool = False):
        """stack_kw > 0: P is a dl_shift_stack image (channel = a*stack_kw + kw); vertical taps only (KH = k, KW = 1)"""
        _need_cuda(P, Q, grad)
        assert grad.dtype == torch.float32 and grad.is_contiguous()
        d = L.WgradDesc()
        d.N, d.Hp, d.Wp, d.CAp = P.shape
        d.p_pstride = pstride(P)
        _, d.Hq, d.Wq, d.CBp = Q.shape
        d.q_pstride = pstride(Q)
        d.KH = d.KW = k
        d.step, d.pad, d.pad_mode = step, pad, pad_mode
        d.CA, d.CB = grad.shape[0], grad.shape[1]
        d.pad_w, d.stack_kw = -1, 0
        if stack_kw:
            d.KW, d.pad_w, d.stack_kw, d.CA = 1, 0, stack_kw, grad.shape[0] * stack_kw
        d.dtype, d.prec = dl_dtype(P), prec
        d.accumulate = 1 if accumulate else 0
        d.p_act, d.q_act = p_act, q_act
        d.p_split, d.q_split = (1 if p_split else 0), (1 if q_split else 0)
        j = d.KH * d.KW * d.CBp
        strict = d.dtype == L.DL_F32 and prec == L.PREC_BF16X3 and not _NO_X3_GLDS
        fast = wgrad_fast_path(d.CAp, j, d.dtype == L.DL_BF16 and prec == L.PREC_BF16, p_act == L.ACT_NONE and q_act == L.ACT_NONE,
                               pad_mode == L.PAD_ZERO, strict, p_act in _X3_ACTS and q_act in _X3_ACTS)
        # which kernel will run (the library decides: csrc/wgrad.hip wgrad_kernel_of), with its tile count -- split-K is sized from that
        d.splitk = 1
        tiles, ksteps, kname = L.i32(), L.i32(), C.c_char_p()
        multi = int(self.lib.dl_wgrad_plan(C.byref(d), C.byref(tiles), C.byref(ksteps), C.byref(kname)))
        if multi < 0:
            self.check(multi, 'dl_wgrad_plan')
        w4 = kname.value == b'wgrad_w4_kernel'
        deferring = _WGRAD_DEFER and WS._thread_state().get('defer_depth', 0) > 0
        # batched: the w4 kernel's layers only.  Measured r05 (profiles/r05/wgrad_first_look.txt, 18 layers): bf16 w4 129.6 -> 114.3 us per layer; the strict
        # glds_x3 kernel 465.5 -> 473.2 (its 3 x MFMA work already hides the slab store; split-K 9 instead of 28 only lengthens the tail)
        if deferring and _WGRAD_BATCH and splitk is None and multi == 1 and w4 and wgrad_batch_shape(d.KH, d.KW, d.step, d.Wp, d.Hp, d.Hq, d.CAp, d.CBp):
            self._wgrad_queued(WS._state(), d, P, Q, grad, tiles.value, ksteps.value)
            return
        if splitk is not None:
            d.splitk = splitk
        elif w4:
            d.splitk = max(1, min(NUM_CUS // tiles.value, ksteps.value))       # 12 tiles x 21 row ranges = 252 workgroups: one round of the chip
        else:
            d.splitk = choose_wgrad_splitk(d.CAp, j, d.N * d.Hp * d.Wp, fast)
        if splitk is None and not _NO_WGRAD_C4 and self.wgrad_c4_applies(P, Q, grad, k, step, pad, pad_mode, p_act, q_act, prec, stack_kw):
            d.splitk = WGRAD_C4_PARTS          # one partial result per persistent workgroup (csrc/wgrad_c4.h)
        nslab = int(self.lib.dl_wgrad_slab_floats(C.byref(d)))
        if deferring and self.lib.dl_conv_wgrad_deferrable(C.byref(d)):
            self._wgrad_deferred(WS._state(), d, P, Q, grad, nslab)
            return
        slab = WS.get('wgrad_slab', nslab, P.device)
        self.check(self.lib.dl_conv_wgrad(C.byref(d), _ptr(P), _ptr(Q), _ptr(grad), _ptr(slab), _stream()), 'dl_conv_wgrad')

    # ---- deferred slab reduction (include/deepliif_hip.h: dl_conv_wgrad_slabs / dl_wgrad_reduce_batch)
    # Inside Tape.backward() every general-path weight gradient only writes its split-K slabs, each into its own region of a per-thread arena;
    # ONE batched launch reduces them when (a) the pass ends, (b) a parameter comes up a second time (a discriminator that ran on real and
    # fake pairs: the second accumulation must see the first), (c) the arena is full, or (d) the data-parallel exchange is about to put
    # gradients on the wire (distributed.GradExchanger._launch).  Results are bit-identical to the immediate reduction.
    def wgrad_defer_begin(self):
        st = WS._thread_state()
        st['defer_depth'] = st.get('defer_depth', 0) + 1

    def wgrad_defer_end(self):
        st = WS._thread_state()
        try:
            if st.get('defer_depth', 0) == 1:
                self.wgrad_flush_all()
        finally:
            st['defer_depth'] = max(0, st.get('defer_depth', 0) - 1)

    def wgrad_abandon(self):
        """forget everything queued / pending on this thread WITHOUT launching (models.StepGraph: a capture that failed half-way recorded weight gradients whose
        kernels never ran; reducing their slabs later would add garbage to the gradients)"""
        st = WS._thread_state()
        for sub in [st] + WS.stream_states():
            sub['defer_pending'], sub['defer_queue'], sub['defer_blocks'], sub['defer_off'] = [], [], 0, 0
            sub['defer_grads'] = set()
        st['defer_depth'] = 0

    def wgrad_flush_all(self):
        """wgrad_flush() on every stream of this thread that has slabs pending (each batch is launched on the stream that wrote its slabs)"""
        self.wgrad_flush()
        for sub in WS.stream_states():
            if sub.get('defer_pending') or sub.get('defer_queue'):
                with torch.cuda.stream(sub['stream_obj']):
                    self.wgrad_flush()

    def _arena_region(self, st, need: int, device):
        """`need` floats of this state's slab arena (flushes what is pending when it is full; grows it when it is too small)"""
        arena = st.get('defer_arena')
        if arena is not None and st.get('defer_off', 0) + need > arena.numel():
            self._reduce_pending(st)
        if arena is None or arena.numel() < need or arena.device != device:
            self._reduce_pending(st)
            if arena is not None:
                Workspace.realloc_generation += 1          # (see Workspace.realloc_generation: a captured graph holds the old arena's address)
            arena = torch.empty(max(need, _WGRAD_ARENA_MB * (1 << 18)), dtype=torch.float32, device=device)
            st['defer_arena'], st['defer_off'] = arena, 0
        off = st.get('defer_off', 0)
        st['defer_off'] = off + need
        return arena[off:off + need]

    def _wgrad_queued(self, st, d, P, Q, grad, tiles, ksteps):
        """batched weight gradient: keep the operands (the references keep dL/dy and x alive until the launch) and compute at the next flush"""
        gp = grad.data_ptr()
        if gp in st.setdefault('defer_grads', set()):
            self.wgrad_flush()
        st['defer_grads'].add(gp)
        d.splitk = choose_wgrad_batch_splitk(tiles, ksteps)
        queue = st.setdefault('defer_queue', [])
        queue.append((bytes(d), d, P, Q, grad))
        # a launch takes at most WGRAD_MULTI_MAX layers anyway: once that many are queued (a tape that never reaches a network marker -- CycleGAN, KD, direct
        # Tape users -- would otherwise keep the operands of EVERY layer of the pass alive, 2 x 67 MB each at batch 8), compute them now.  A layer's split-K
        # does not depend on how many layers share its launch, so the bits are the same.
        if len(queue) >= L.WGRAD_MULTI_MAX:
            self._launch_queued(st)

    def _launch_queued(self, st):
        """one dl_conv_wgrad_multi per group of same-descriptor layers (<= WGRAD_MULTI_MAX each); their reductions join the pending list"""
        queue = st.get('defer_queue')
        if not queue:
            return
        st['defer_queue'] = []
        groups = {}
        for item in queue:
            groups.setdefault(item[0], []).append(item)
        for items in groups.values():
            d = items[0][1]
            per_layer = int(self.lib.dl_wgrad_slab_floats(C.byref(d)))           # the library places layer l's slabs at slab + l * per_layer
            for i0 in range(0, len(items), L.WGRAD_MULTI_MAX):
                chunk = items[i0:i0 + L.WGRAD_MULTI_MAX]
                n = len(chunk)
                slab = self._arena_region(st, (n * per_layer + 63) // 64 * 64, chunk[0][2].device)
                arr = C.c_void_p * n
                Ps, Qs, Gs = arr(*[it[2].data_ptr() for it in chunk]), arr(*[it[3].data_ptr() for it in chunk]), arr(*[it[4].data_ptr() for it in chunk])
                ents = (L.WgradReduceEntry * n)()
                _LAUNCH.dev = chunk[0][2].device
                self.check(self.lib.dl_conv_wgrad_multi(C.byref(d), n, Ps, Qs, Gs, _ptr(slab), ents, _stream()), 'dl_conv_wgrad_multi')
                pend = st.setdefault('defer_pending', [])
                for l in range(n):
                    e = L.WgradReduceEntry.from_buffer_copy(ents[l])
                    e.block0 = st.get('defer_blocks', 0)
                    st['defer_blocks'] = e.block0 + e.nblocks
                    pend.append(e)

    def _wgrad_deferred(self, st, d, P, Q, grad, nslab):
        gp = grad.data_ptr()
        need = (nslab + 63) // 64 * 64
        if gp in st.setdefault('defer_grads', set()):
            self.wgrad_flush()
        slab = self._arena_region(st, need, P.device)
        e = L.WgradReduceEntry()
        self.check(self.lib.dl_conv_wgrad_slabs(C.byref(d), _ptr(P), _ptr(Q), _ptr(grad), _ptr(slab), C.byref(e), _stream()), 'dl_conv_wgrad_slabs')
        e.block0 = st.get('defer_blocks', 0)
        st['defer_blocks'] = e.block0 + e.nblocks
        st['defer_grads'].add(gp)
        st.setdefault('defer_pending', []).append(e)       # (looked up AFTER the flushes above: they start a new list)

    def wgrad_flush(self):
        """compute every queued weight gradient and reduce every pending slab set of this thread / branch stream (no-op when there is nothing)"""
        st = WS._state()
        self._launch_queued(st)
        self._reduce_pending(st)
        st['defer_grads'] = set()

    def _reduce_pending(self, st):
        pend = st.get('defer_pending')
        if not pend:
            st['defer_off'] = 0                  # whatever wrote into the arena has been reduced by a launch earlier in stream order
            return
        arena = st['defer_arena']
        tab = (L.WgradReduceEntry * len(pend))(*pend)
        key = bytes(tab)
        cache = st.setdefault('defer_tables', {})
        dev_tab = cache.get(key)
        if dev_tab is None:
            # static shapes: the same tables come back every step (the arena and the flat gradient buffer do not move), so after the first
            # step nothing is copied -- which is also what lets models.StepGraph capture the pass
            if len(cache) >= 512:
                cache.clear()
            dev_tab = torch.frombuffer(bytearray(key), dtype=torch.uint8).to(arena.device)
            cache[key] = dev_tab
        total = st['defer_blocks']
        st['defer_pending'], st['defer_blocks'], st['defer_off'] = [], 0, 0
        _LAUNCH.dev = arena.device
        self.check(self.lib.dl_wgrad_reduce_batch(_ptr(dev_tab), len(pend), total, _stream()), 'dl_wgrad_reduce_batch')

    supports_split = os.environ.get('DL_NO_SPLIT_COPY') is None          # A/B switch: DL_NO_SPLIT_COPY=1 keeps every hi / lo split inside the conv kernels

    def conv_takes_split(self, x, prec, in_act, pad_mode) -> bool:
        """can dl_conv_forward read the split copy of x (strict policy on the direct-to-LDS kernels, csrc/conv_x3.h: x3_glds_applies)?"""
        return (self.supports_split and not _NO_X3_GLDS and x.dtype == torch.float32 and prec == L.PREC_BF16X3 and in_act == L.ACT_NONE)

    def wgrad_takes_split(self, P, Q, grad, k, pad_mode, prec, stack_kw=0) -> bool:
        """mirror of the fast3 predicate in wgrad.hip: strict direct-to-LDS weight gradient"""
        if not self.supports_split or _NO_X3_GLDS or stack_kw or P.dtype != torch.float32 or prec != L.PREC_BF16X3 or pad_mode != L.PAD_ZERO:
            return False
        cap, j, ptot = P.shape[3], k * k * Q.shape[3], P.shape[0] * P.shape[1] * P.shape[2]
        return cap % 128 == 0 and j >= 256 and ptot >= 32 * choose_wgrad_splitk(cap, j, ptot, True)

    def wgrad_c4_applies(self, P, Q, grad, k, step, pad, pad_mode, p_act, q_act, prec, stack_kw=0) -> bool:
        return not _NO_WGRAD_C4 and wgrad_c4_ok(P.shape[3], grad.shape[0], Q.shape[3], grad.shape[1], k, step, pad, pad_mode, P.shape[1], P.shape[2], Q.shape[1], Q.shape[2],
                           (P.dtype == torch.bfloat16 and prec == L.PREC_BF16) or (P.dtype == torch.float32 and prec == L.PREC_BF16X3 and not _NO_C4_X3),
                           p_act == L.ACT_NONE and q_act == L.ACT_NONE, bool(stack_kw))

    # ---- normalisation
    def _norm_desc(self, y, C_real, scope, act, momentum, z_ps, r_ps):
        d = L.NormDesc()
        d.N, d.H, d.W, d.Cp = y.shape
        d.C = C_real
        d.y_pstride, d.z_pstride, d.r_pstride = pstride(y), z_ps, r_ps
        d.dtype, d.scope, d.act = dl_dtype(y), scope, act
        d.eps, d.momentum = 1e-5, momentum
        return d

    def norm_forward(self, y, z, C_real, scope, act, gamma, beta, running_mean, running_var, momentum, residual, ext_nchunks=0, z_split=None, store_z=True):
        """ext_nchunks > 0: the convolution that produced y already left its partial sums in the 'norm_ws' workspace.
        z_split (fp32 policy): dense buffer shaped like z that receives the split copy of z ([8 bf16 hi | 8 bf16 lo] per 8 channels);
        store_z=False (needs z_split): `z` only gives the geometry, its fp32 values are NOT written"""
        _need_cuda(y, z, gamma, beta, residual, z_split)
        assert store_z or z_split is not None
        assert z_split is None or (z_split.is_contiguous() and z_split.dtype == torch.float32 and z_split.shape == z.shape)
        d = self._norm_desc(y, C_real, scope, act, momentum, pstride(z), pstride(residual) if residual is not None else 8)
        d.ext_nchunks = ext_nchunks
        if not ext_nchunks:
            WS.bump_norm_token()            # the stand-alone statistics pass overwrites the shared workspace
        stats = torch.empty(4, y.shape[0], y.shape[3], dtype=torch.float32, device=y.device)   # mean, rstd, scale, shift
        ws = WS.get('norm_ws', self.lib.dl_norm_ws_floats(C.byref(d)), y.device)
        self.check(self.lib.dl_norm_forward(C.byref(d), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                         _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), _ptr(residual), _ptr(z) if store_z else None,
                                         _ptr(ws), _ptr(z_split), _stream()), 'dl_norm_forward')
        return stats

    def norm_backward(self, dz, y, dy, stats, C_real, scope, act, gamma, dgamma, dbeta, dy_chansum=None, ext_nchunks=0, dy_split=None, store_dy=True):
        """ext_nchunks > 0: the conv that produced dz already left the reductions in the 'norm_ws' workspace (conv_forward(bn=...)).
        dy_split: see norm_forward(z_split).  store_dy=False (needs dy_split): `dy` only gives the geometry, its fp32 values are NOT written."""
        _need_cuda(dz, y, dy, dy_chansum, dy_split)
        assert store_dy or dy_split is not None
        assert dy_split is None or (dy_split.is_contiguous() and dy_split.dtype == torch.float32 and dy_split.shape == dy.shape)
        d = self._norm_desc(y, C_real, scope, act, -1.0, pstride(dz), pstride(dy))
        d.ext_nchunks = ext_nchunks
        WS.bump_norm_token()
        ws = WS.get('norm_ws', self.lib.dl_norm_ws_floats(C.byref(d)), y.device)
        self.check(self.lib.dl_norm_backward(C.byref(d), _ptr(dz), _ptr(y), _ptr(gamma), _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]),
                                          _ptr(stats[3]), _ptr(dy) if store_dy else None, _ptr(dgamma), _ptr(dbeta), 1, _ptr(dy_chansum), _ptr(ws), _ptr(dy_split),
                                          _stream()),
                'dl_norm_backward')

    # ---- elementwise
    def act_forward(self, act, x, y):
        _need_cuda(x, y)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        self.check(self.lib.dl_act_forward(act, dl_dtype(x), _ptr(x), pstride(x), _ptr(y), pstride(y), npix, x.shape[3], _stream()), 'dl_act_forward')

    def act_backward(self, act, dy, y, dx):
        _need_cuda(dy, y, dx)
        npix = y.shape[0] * y.shape[1] * y.shape[2]
        self.check(self.lib.dl_act_backward(act, dl_dtype(y), _ptr(dy), pstride(dy), _ptr(y), pstride(y), _ptr(dx), pstride(dx), npix,
                                         y.shape[3], _stream()), 'dl_act_backward')

    def dropout(self, x, y, p, seed):
        _need_cuda(x, y)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        self.check(self.lib.dl_dropout(dl_dtype(x), _ptr(x), pstride(x), _ptr(y), pstride(y), npix, x.shape[3], float(p), int(seed) & (2 ** 64 - 1),
                                    _stream()), 'dl_dropout')

    def axpby(self, alpha, a, beta, b, out):
        _need_cuda(a, b, out)
        npix = a.shape[0] * a.shape[1] * a.shape[2]
        self.check(self.lib.dl_axpby(dl_dtype(a), float(alpha), _ptr(a), pstride(a), float(beta), _ptr(b), pstride(b) if b is not None else 8,
                                  _ptr(out), pstride(out), npix, a.shape[3], _stream()), 'dl_axpby')

    def gate_forward(self, x, psi, out):
        """out = x * psi[..., :1] (attention gate, att_unet.py:108-115)"""
        _need_cuda(x, psi, out)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        self.check(self.lib.dl_gate_forward(dl_dtype(x), _ptr(x), pstride(x), _ptr(psi), pstride(psi), _ptr(out), pstride(out), npix, x.shape[3], _stream()),
                'dl_gate_forward')

    def gate_backward(self, g, x, psi, dx, dpsi):
        """dx = g * psi[..., :1] (dx may be None); dpsi[..., 0] = sum_c g * x, dpsi[..., 1:] = 0"""
        _need_cuda(g, x, psi, dx, dpsi)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        self.check(self.lib.dl_gate_backward(dl_dtype(x), _ptr(g), pstride(g), _ptr(x), pstride(x), _ptr(psi), pstride(psi), _ptr(dx),
                                          pstride(dx) if dx is not None else 8, _ptr(dpsi), pstride(dpsi), npix, x.shape[3], _stream()), 'dl_gate_backward')

    def copy_channels(self, src, s_c0, dst, d_c0, nch, accumulate=False):
        _need_cuda(src, dst)
        npix = src.shape[0] * src.shape[1] * src.shape[2]
        self.check(self.lib.dl_copy_channels(dl_dtype(src), _ptr(src), pstride(src), s_c0, _ptr(dst), pstride(dst), d_c0, npix, nch,
                                          1 if accumulate else 0, _stream()), 'dl_copy_channels')

    def channel_sum(self, x, C_real, out, accumulate):
        _need_cuda(x, out)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        ws = WS.get('csum_ws', 256 * x.shape[3], x.device)
        self.check(self.lib.dl_channel_sum(dl_dtype(x), _ptr(x), pstride(x), npix, x.shape[3], C_real, _ptr(out), 1 if accumulate else 0,
                                        _ptr(ws), _stream()), 'dl_channel_sum')

    def nchw_to_nhwc(self, src, dst, c0, zero_pad_to):
        _need_cuda(src, dst)
        assert src.dtype == torch.float32 and src.is_contiguous()
        n, c, h, w = src.shape
        self.check(self.lib.dl_nchw_to_nhwc(_ptr(src), n, c, h, w, dl_dtype(dst), _ptr(dst), pstride(dst), c0, zero_pad_to, _stream()),
                'dl_nchw_to_nhwc')

    def nhwc_to_nchw(self, src, c0, dst):
        _need_cuda(src, dst)
        assert dst.dtype == torch.float32 and dst.is_contiguous()
        n, c, h, w = dst.shape
        self.check(self.lib.dl_nhwc_to_nchw(dl_dtype(src), _ptr(src), pstride(src), c0, _ptr(dst), n, c, h, w, _stream()), 'dl_nhwc_to_nchw')

    # ---- narrow-Cout forward in one kernel (rolling input rows, csrc/conv_small.hip)
    def conv_narrow_supported(self, x, cin_p, cout, k, pad, pad_mode, act=L.ACT_NONE) -> bool:
        return act in (L.ACT_NONE, L.ACT_TANH) and bool(self.lib.dl_conv_narrow_supported(dl_dtype(x), cin_p, pstride(x), cout, k, k, pad, pad_mode)) and not _NO_NARROW_ROLL

    def conv_narrow_forward(self, packed, x, out, cout, k, pad, bias, act):
        _need_cuda(x, out, bias, packed.hi)
        n, h, w, cp = x.shape
        if x.dtype == torch.float32:            # strict policy: fp32 rows split while the fragments are read, hi weights in registers, lo weights in LDS
            assert packed.lo is not None and out.dtype == torch.float32
            self.check(self.lib.dl_conv_narrow_forward_x3(_ptr(x), n, h, w, cp, pstride(x), _ptr(packed.hi), _ptr(packed.lo), packed.plan.kstride, cout, k, k, pad,
                                                       _ptr(bias), act, _ptr(out), pstride(out), out.shape[3], _stream()), 'dl_conv_narrow_forward_x3')
            return
        self.check(self.lib.dl_conv_narrow_forward(_ptr(x), n, h, w, cp, pstride(x), _ptr(packed.hi), packed.plan.kstride, cout, k, k, pad, _ptr(bias), act,
                                                _ptr(out), pstride(out), out.shape[3], _stream()), 'dl_conv_narrow_forward')

    # ---- narrow-Cout helpers
    def shift_sum(self, T, cout, kw, pad, pad_mode, bias, act, out):
        _need_cuda(T, out, bias)
        n, h, w, tc = T.shape
        assert T.dtype == torch.float32 and T.is_contiguous()
        self.check(self.lib.dl_shift_sum(_ptr(T), n, h, w, tc, cout, kw, pad, pad_mode, _ptr(bias), act, dl_dtype(out), _ptr(out), pstride(out),
                                      out.shape[3], _stream()), 'dl_shift_sum')

    def convt4_gather(self, T, cout, bias, act, out):
        """second half of the narrow-Cout ConvTranspose2d(k=4, s=2, p=1): T = raw 1x1 GEMM [N,H,W,>=16*cout] fp32 -> out [N,2H,2W,Cp]"""
        _need_cuda(T, out, bias)
        n, h, w, tc = T.shape
        assert T.dtype == torch.float32 and T.is_contiguous() and tuple(out.shape[:3]) == (n, 2 * h, 2 * w)
        self.check(self.lib.dl_convt4_gather(_ptr(T), n, h, w, tc, cout, _ptr(bias), act, dl_dtype(out), _ptr(out), pstride(out), out.shape[3], _stream()),
                'dl_convt4_gather')

    def shift_stack(self, dy, cout, kw, pad, D):
        _need_cuda(dy, D)
        n, h, w, _ = dy.shape
        assert D.is_contiguous() and D.dtype == dy.dtype
        self.check(self.lib.dl_shift_stack(dl_dtype(dy), _ptr(dy), pstride(dy), n, h, w, cout, kw, pad, L.PAD_ZERO, _ptr(D), D.shape[3], _stream()),
                'dl_shift_stack')

    def reflect_fold(self, src, dst, pad):
        """dst [N,H,W,Cp] <- gradient of nn.ReflectionPad2d(pad) applied to src [N,H+2p,W+2p,Cp] (mirrored borders added back)"""
        _need_cuda(src, dst)
        n, h, w, cp = dst.shape
        assert src.shape == (n, h + 2 * pad, w + 2 * pad, cp) and src.dtype == dst.dtype
        self.check(self.lib.dl_reflect_fold(dl_dtype(src), _ptr(src), pstride(src), _ptr(dst), pstride(dst), n, h, w, pad, cp, _stream()), 'dl_reflect_fold')

    # ---- tiles: uint8 [H, W, 3] images <-> engine tile batches (crop + transform, is_empty statistic, tensor2im + stitch)
    def tile_gather(self, images, H0, W0, origins, tile, pad, pad_rgb, lut, out):
        _need_cuda(origins, lut, out, *images)
        assert origins.dtype == torch.int32 and origins.is_contiguous() and lut.dtype == torch.float32 and lut.numel() == 256
        n = len(images)
        ptrs = (C.c_void_p * n)(*[im.data_ptr() for im in images])
        strides = (C.c_int64 * n)(*[im.stride(0) for im in images])
        self.check(self.lib.dl_tile_gather_u8(ptrs, strides, n, H0, W0, _ptr(origins), origins.shape[0], tile, pad, pad_rgb, _ptr(lut), dl_dtype(out),
                                           _ptr(out), pstride(out), out.shape[3], _stream(out)), 'dl_tile_gather_u8')

    def tile_gray_stats(self, image, H0, W0, origins, tile, pad, pad_rgb, stats):
        _need_cuda(image, origins, stats)
        assert stats.dtype == torch.int64 and stats.is_contiguous() and stats.shape == (origins.shape[0], 3)
        self.check(self.lib.dl_tile_gray_stats_u8(_ptr(image), image.stride(0), H0, W0, _ptr(origins), origins.shape[0], tile, pad, pad_rgb, _ptr(stats),
                                               _stream(image)), 'dl_tile_gray_stats_u8')

    def tile_paste(self, tiles, tile, rects, dst):
        _need_cuda(tiles, rects, dst)
        assert rects.dtype == torch.int32 and rects.is_contiguous() and rects.shape[1] == 8
        assert dst.dtype == torch.uint8 and dst.stride(2) == 1 and dst.stride(1) == 3
        if tiles is None:
            dt, tp, ps = L.DL_F32, None, 8
        else:
            dt, tp, ps = dl_dtype(tiles), _ptr(tiles), pstride(tiles)
        self.check(self.lib.dl_tile_paste_u8(dt, tp, ps, tile, _ptr(rects), rects.shape[0], _ptr(dst), dst.stride(0), _stream(dst)), 'dl_tile_paste_u8')

    # ---- losses
    def loss(self, kind, x, target, target_const, C_real, loss_out, grad, grad_scale, out_scale=1.0, accumulate=False):
        """loss_out[0] = (accumulate ? loss_out[0] : 0) + out_scale * mean loss; grad = grad_scale * d(mean loss)/dx"""
        _need_cuda(x, target, loss_out, grad)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        ws = WS.get('loss_ws', self.lib.dl_loss_ws_floats(), x.device)
        self.check(self.lib.dl_loss_acc(kind, dl_dtype(x), _ptr(x), pstride(x), _ptr(target), pstride(target) if target is not None else 8,
                                     float(target_const), npix, C_real, x.shape[3], _ptr(loss_out), float(out_scale), 1 if accumulate else 0, _ptr(grad),
                                     pstride(grad) if grad is not None else 8, float(grad_scale), _ptr(ws), _stream()), 'dl_loss_acc')

    def upsample2(self, src, dst, backward=False):
        """nn.Upsample(scale_factor=2, mode='nearest') (backward=False: src [N,H,W,Cp] -> dst [N,2H,2W,Cp]) or its gradient (2 x 2 block sums)"""
        _need_cuda(src, dst)
        small = dst if backward else src
        n, h, w, cp = small.shape
        big = src if backward else dst
        assert tuple(big.shape) == (n, 2 * h, 2 * w, cp), (big.shape, small.shape)
        self.check(self.lib.dl_upsample2_nearest(dl_dtype(src), 1 if backward else 0, _ptr(src), pstride(src), _ptr(dst), pstride(dst), n, h, w, cp, _stream()),
                'dl_upsample2_nearest')

    def kldiv(self, x, t, C_real, loss_out, grad, grad_scale, out_scale=1.0, accumulate=False):
        """loss_out[0] (+)= out_scale * KL(softmax(t) || softmax(x)) over ALL real elements (DeepLIIFKD_model.py:313-336); grad = grad_scale * (softmax(x) - softmax(t))"""
        _need_cuda(x, t, loss_out, grad)
        assert x.shape == t.shape and x.dtype == t.dtype
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        ws = WS.get('kldiv_ws', self.lib.dl_kldiv_ws_floats(), x.device)
        self.check(self.lib.dl_kldiv(dl_dtype(x), _ptr(x), pstride(x), _ptr(t), pstride(t), npix, C_real, x.shape[3], _ptr(loss_out), float(out_scale),
                                  1 if accumulate else 0, _ptr(grad), pstride(grad) if grad is not None else 8, float(grad_scale), _ptr(ws), _stream()), 'dl_kldiv')

    # ---- 2x2 max pooling (VGG19 features)
    def maxpool2_forward(self, x, y):
        _need_cuda(x, y)
        n, h, w, cp = x.shape
        assert tuple(y.shape) == (n, h // 2, w // 2, cp)
        self.check(self.lib.dl_maxpool2_forward(dl_dtype(x), _ptr(x), pstride(x), _ptr(y), pstride(y), n, h, w, cp, _stream()), 'dl_maxpool2_forward')

    def maxpool2_backward(self, x, dy, dx):
        _need_cuda(x, dy, dx)
        n, h, w, cp = x.shape
        self.check(self.lib.dl_maxpool2_backward(dl_dtype(x), _ptr(x), pstride(x), _ptr(dy), pstride(dy), _ptr(dx), pstride(dx), n, h, w, cp, _stream()),
                'dl_maxpool2_backward')

    # ---- optimiser
    def adam_step(self, p, g, m, v, lr, b1, b2, eps, step, gscale):
        _need_cuda(p, g, m, v)
        self.check(self.lib.dl_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(b1), float(b2), float(eps), int(step),
                                      float(gscale), _stream()), 'dl_adam_step')


    def adam_hyper(self, lr, b1, b2, eps, step, gscale, hyper_host: torch.Tensor):
        """fill the (pinned) host tensor with the scalars dl_adam_step would use at this step (graph mode, optim.FusedAdam.prepare_step)"""
        assert hyper_host.device.type == 'cpu' and hyper_host.dtype == torch.float32 and hyper_host.numel() >= 8
        self.check(self.lib.dl_adam_hyper(float(lr), float(b1), float(b2), float(eps), int(step), float(gscale), C.c_void_p(hyper_host.data_ptr())), 'dl_adam_hyper')

    def adam_step_dev(self, p, g, m, v, hyper_dev):
        _need_cuda(p, g, m, v, hyper_dev)
        self.check(self.lib.dl_adam_step_dev(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(hyper_dev), _stream()), 'dl_adam_step_dev')


_impl = None          # the bf16 backend (tests replace it with the CPU emulation, which serves both formats)
_impl_f16 = None


def impl():
    global _impl, _impl_f16
    if half_format() == 'fp16' and not getattr(_impl, 'emulates_any_half', False):
        if _impl_f16 is None:
            _impl_f16 = HipBackend('fp16')        # raises if libdeepliif_hip_f16.so is missing
        return _impl_f16
    if _impl is None:
        _impl = HipBackend()        # raises if the .so is missing
    return _impl
