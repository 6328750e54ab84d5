"""Data-parallel gradient exchange: one process per GPU (torchrun env contract: RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT), torch.distributed backend 'nccl' (= RCCL over xGMI on ROCm) on GPUs, 'gloo' in CPU tests.

The reference wraps each of its 18 networks in DistributedDataParallel (networks.py:131-134) -> 18 reducers with 25 MB
buckets, all-reduce overlapped with the two backward passes.  Here every optimizer owns ONE flat fp32 gradient buffer
(optim.FlatParams) in which the parameters of one network are contiguous, so the exchange is one sum-all-reduce per NETWORK slice
(28-45 MB for the 64-wide nets; xGMI is point-to-point: few large messages beat many small ones), launched asynchronously the
moment that network's backward has finished -- the tape carries a marker in front of each network's first forward node, which
reverse mode reaches after the network's last weight gradient (models.py) -- and waited for right before the optimizer step.
RCCL runs the collective on its own stream behind an event on the compute stream, so it overlaps the backward kernels of the
networks that are still to come.  The 1/world_size averaging is folded into the Adam kernel (grad_scale).  BatchNorm statistics
stay per-rank, exactly like the reference's un-synchronised DDP (broadcast_buffers=False, no SyncBatchNorm).

Initial parameters are broadcast from rank 0 once (DistributedDataParallel does the same at construction): replicas that were
not seeded identically would otherwise average gradients of different weights for ever without any error.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

BUCKET_ELEMS = 8 * 1024 * 1024       # 32 MB fp32: size at which the final tail of a PROGRESSIVE slice goes on the wire (one all-reduce call per bucket)
MSG_ELEMS = 64 * 1024 * 1024         # 256 MB fp32: the largest single all-reduce call of everything else -- a whole-network message (Resnet-9 45 MB,
                                     # NLayerD 28 MB) stays ONE call (the 32 MB bound used to cut the 45 MB Resnet-9 slice into two calls)
SPLIT_ELEMS = 16 * 1024 * 1024       # network slices above 64 MB (the 67 M-parameter UNet-512 generators: 268 MB) are exchanged progressively:
                                     # a bucket goes on the wire as soon as >= 32 MB of the network's gradient tail is final (Tape.on_final),
                                     # instead of the whole slice at the network's marker.  The network whose backward ends the pass -- the FIRST one
                                     # registered with an optimizer: reverse mode reaches it last -- is progressive too, in two halves, whatever its
                                     # size: as one message it would start only when the pass is over and be exposed in full (VERDICT r3 #8a)
OVERLAP = os.environ.get('DL_DP_OVERLAP', '1') != '0'        # A/B switch: 0 = one blocking exchange after the whole backward (round 1)
GRAD_BF16 = os.environ.get('DL_DP_GRAD_BF16', '0') == '1'    # opt-in: the gradients go on the wire as bf16 -- every rank rounds its fp32 slice to bf16 (nearest even), the
                                                             # collective sums the bf16 values, the sum is widened back into the fp32 gradient buffer: half the bytes
                                                             # per step (5 G + 5 D: 368 -> 184 MB) for one extra rounding of the SUMMED gradient (2^-9 relative, the
                                                             # size of the bf16 policy's own activation rounding).  Deterministic; off by default: the fp32 exchange
                                                             # is the one held bit-identical to the single-process run (tests/test_distributed_gloo.py)
FORCE = os.environ.get('DL_DP_FORCE', '0') == '1'            # run the exchange path with ONE rank too (all-reduce over a 1-rank group = identity):
                                                             # exercises RCCL's stream ordering against the ctypes launches on a single GPU


def active() -> bool:
    return world_size() > 1 or (FORCE and dist.is_available() and dist.is_initialized())


def init_process_group_from_env(backend: str = None):
    """Idempotent; returns (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    if (world > 1 or FORCE) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class GradExchanger:
    """Sum-all-reduce of an optimizer's flat gradient across ranks; averaging is applied inside the optimizer step.

    register_net(opt, params)  once: this contiguous run of the optimizer's parameters is one network
    begin(optimizer)           start of a backward pass whose gradients belong to `optimizer`
    param_final(p)             (Tape.on_final) one parameter's gradient is final: networks above 64 MB send their final tail in 32 MB buckets
    ready(params)              the gradients of this contiguous run of parameters are final: exchange what has not gone out yet (asynchronously)
    finish(optimizer)          exchange whatever was not announced, wait for everything, set optimizer.dp_scale
    all_reduce(optimizer)      begin + finish without announcements (one blocking exchange)"""

    def __init__(self):
        self.handles: List = []
        self.done: List[Tuple[int, int]] = []
        self.current = None
        self._slices: Dict[int, Tuple[int, int]] = {}
        self._synced = set()
        self.launch_log: List[Tuple[int, int]] = []          # (start, end) of every range exchanged early in the last pass (tests / diagnostics)
        self._big: Dict[int, Tuple[int, int]] = {}           # id(param) -> slice (start, end) of its network, progressive networks only
        self._bucket: Dict[Tuple[int, int], int] = {}        # progressive slice -> its bucket size in elements
        self._first_of: Dict[int, Tuple[int, int]] = {}      # id(optimizer) -> slice of the first network registered with it
        self.profile = False                                 # bench.py: record device-side events around the waits of finish()
        self.exposed: List[Tuple[str, object, object]] = []  # (optimizer tag, event before the waits, event after) per finish() while profiling
        self.pass_log: List[dict] = []                       # per finish(): bytes and all-reduce calls of the pass
        self._param_range: Dict[int, Tuple[int, int]] = {}   # id(param) -> its own (start, end) in the flat buffer
        self._progress: Dict[Tuple[int, int], dict] = {}     # per big slice and pass: final-but-unsent intervals, send watermark
        self._wire: List = []                                # DL_DP_GRAD_BF16: (bf16 wire copy, fp32 destination) of every message of the pass

    def register_net(self, optimizer, params):
        """tell the exchanger which contiguous run of `optimizer`'s flat parameters is one network (models call this once per network)"""
        flat = getattr(optimizer, 'flat', None)
        params = [p for p in params]
        if flat is None or not params:
            return
        s, e = flat.slice_of(params)
        self._slices[id(params[0])] = (s, e)
        first = id(optimizer) not in self._first_of
        if first:
            self._first_of[id(optimizer)] = (s, e)
        if e - s > SPLIT_ELEMS or (first and len(params) > 1):
            self._bucket[(s, e)] = BUCKET_ELEMS if e - s > SPLIT_ELEMS else max(1, (e - s + 1) // 2)
            for p in params:
                ps, pe = flat.slice_of([p])
                self._big[id(p)] = (s, e)
                self._param_range[id(p)] = (ps, pe)

    # -- one-time parameter synchronisation ---------------------------------------------------------------------
    def sync_parameters(self, optimizer, modules=()):
        """broadcast the flat parameters (and BatchNorm buffers of `modules`) from rank 0, once per optimizer"""
        if not active() or id(optimizer) in self._synced:
            return
        flat = getattr(optimizer, 'flat', None)
        if flat is not None:
            dist.broadcast(flat.data, src=0)
            flat.bump_epoch()                    # packed weight images must be rebuilt from the broadcast values
        for m in modules:
            for b in m.buffers():
                if b.is_floating_point():
                    dist.broadcast(b, src=0)
        self._synced.add(id(optimizer))

    # -- per-pass protocol ---------------------------------------------------------------------------------------
    def begin(self, optimizer):
        self.current = optimizer if (active() and OVERLAP and getattr(optimizer, 'flat', None) is not None) else None
        self.handles, self.done, self.launch_log = [], [], []
        self._progress = {}
        self._calls, self._elems = 0, 0
        self._wire = []

    def param_final(self, p):
        """Tape.on_final: the gradient of `p` is complete for this pass.  Inside a network above SPLIT_ELEMS the flat order of the parameters
        is the network's forward order (ResnetGenerator: a Sequential; UnetGenerator: down path outermost -> innermost, then the up path back
        out), so reverse mode finalises a slice from its END: the final run [w, sent) behind the watermark is launched in buckets once it
        reaches BUCKET_ELEMS.  A parameter that becomes final out of order just waits until the watermark reaches it (or for ready())."""
        opt = self.current
        if opt is None or id(p) not in self._big:
            return
        sl = self._big[id(p)]
        st = self._progress.setdefault(sl, {'final': {}, 'w': sl[1], 'sent': sl[1]})
        a, b = self._param_range[id(p)]
        st['final'][b] = a
        while st['w'] in st['final']:
            st['w'] = st['final'].pop(st['w'])
        if st['sent'] - st['w'] >= self._bucket.get(sl, BUCKET_ELEMS):
            self._launch(opt.flat.grad, st['w'], st['sent'])
            self.launch_log.append((st['w'], st['sent']))
            st['sent'] = st['w']

    def ready(self, params):
        opt = self.current
        if opt is None:
            return
        key = id(params[0])
        if key not in self._slices:
            self._slices[key] = opt.flat.slice_of(params)
        s, e = self._slices[key]
        st = self._progress.get((s, e))
        if st is not None:
            e = st['sent']                     # the tail of this slice already went out in buckets
            st['w'] = st['sent'] = s
        if e > s:
            self._launch(opt.flat.grad, s, e)
            self.launch_log.append((s, e))

    def _launch(self, g, s, e):
        if g.is_cuda:
            from . import ops
            flush = getattr(ops.impl(), 'wgrad_flush_all', None)
            if flush is not None:
                flush()                        # pending split-K slabs (ops.HipBackend: deferred reduction) become gradients before anything goes on the wire
        for b in range(s, e, MSG_ELEMS):
            piece = g[b:min(b + MSG_ELEMS, e)]
            if GRAD_BF16:
                wire = piece.to(torch.bfloat16)              # round-to-nearest-even copy on the compute stream; kept alive until finish() has widened it back
                self._wire.append((wire, piece))
                piece = wire
            self.handles.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True))
            self._calls = getattr(self, '_calls', 0) + 1
        self._elems = getattr(self, '_elems', 0) + (e - s)
        self.done.append((s, e))

    def finish(self, optimizer):
        ws = world_size()
        flat = getattr(optimizer, 'flat', None)
        if flat is None:
            raise RuntimeError('every optimizer on this path owns a FlatParams set (optim.FusedAdam / optim.flat_optimizer)')
        if not active():
            optimizer.dp_scale = 1.0
            self.current = None
            return
        pos = 0
        for s, e in sorted(self.done):           # the ranges nobody announced (all of it when overlap is off)
            if s > pos:
                self._launch(flat.grad, pos, s)
            pos = max(pos, e)
        if pos < flat.numel:
            self._launch(flat.grad, pos, flat.numel)
        # h.wait() on an RCCL work object makes the COMPUTE stream wait (the host returns at once): events on that stream before and after
        # the waits bracket exactly the time the step sits exposed behind the exchange
        ev = None
        if self.profile and flat.grad.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for h in self.handles:
            h.wait()
        for wire, piece in getattr(self, '_wire', []):       # DL_DP_GRAD_BF16: the summed bf16 values back into the fp32 gradient buffer (behind the waits, stream-ordered)
            piece.copy_(wire)
        self._wire = []
        if ev is not None:
            ev[1].record()
            self.exposed.append((getattr(optimizer, 'dp_tag', 'opt'), ev[0], ev[1]))
        self.pass_log.append({'tag': getattr(optimizer, 'dp_tag', 'opt'), 'bytes': (2 if GRAD_BF16 else 4) * getattr(self, '_elems', 0), 'calls': getattr(self, '_calls', 0),
                              'early_ranges': len(self.launch_log)})
        if len(self.pass_log) > 64:
            del self.pass_log[:-64]
        self.handles, self.current = [], None
        optimizer.dp_scale = 1.0 / ws

    def exposed_ms(self) -> Dict[str, float]:
        """mean device-side wait per finish(), by optimizer tag (call after a synchronize; clears the list)"""
        out: Dict[str, List[float]] = {}
        for tag, a, b in self.exposed:
            out.setdefault(tag, []).append(a.elapsed_time(b))
        self.exposed = []
        return {k: sum(v) / len(v) for k, v in out.items()}

    def all_reduce(self, optimizer):
        self.begin(None)
        self.finish(optimizer)
