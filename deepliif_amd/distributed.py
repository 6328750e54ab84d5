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

BUCKET_ELEMS = 64 * 1024 * 1024      # 256 MB fp32: upper bound for one all-reduce call
OVERLAP = os.environ.get('DL_DP_OVERLAP', '1') != '0'        # A/B switch: 0 = one blocking exchange after the whole backward (round 1)


def init_process_group_from_env(backend: str = None):
    """Idempotent; returns (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world > 1 and not dist.is_initialized():
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

    begin(optimizer)           start of a backward pass whose gradients belong to `optimizer`
    ready(params)              the gradients of this contiguous run of parameters are final: exchange them now (asynchronously)
    finish(optimizer)          exchange whatever was not announced, wait for everything, set optimizer.dp_scale
    all_reduce(optimizer)      begin + finish without announcements (one blocking exchange)"""

    def __init__(self):
        self.handles: List = []
        self.done: List[Tuple[int, int]] = []
        self.current = None
        self._slices: Dict[int, Tuple[int, int]] = {}
        self._synced = set()
        self.launch_log: List[Tuple[int, int]] = []          # (start, end) of every slice exchanged early in the last pass (tests / diagnostics)

    # -- one-time parameter synchronisation ---------------------------------------------------------------------
    def sync_parameters(self, optimizer, modules=()):
        """broadcast the flat parameters (and BatchNorm buffers of `modules`) from rank 0, once per optimizer"""
        if world_size() == 1 or id(optimizer) in self._synced:
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
        self.current = optimizer if (world_size() > 1 and OVERLAP and getattr(optimizer, 'flat', None) is not None) else None
        self.handles, self.done, self.launch_log = [], [], []

    def ready(self, params):
        opt = self.current
        if opt is None:
            return
        key = id(params[0])
        if key not in self._slices:
            self._slices[key] = opt.flat.slice_of(params)
        s, e = self._slices[key]
        self._launch(opt.flat.grad, s, e)
        self.launch_log.append((s, e))

    def _launch(self, g, s, e):
        for b in range(s, e, BUCKET_ELEMS):
            self.handles.append(dist.all_reduce(g[b:min(b + BUCKET_ELEMS, e)], op=dist.ReduceOp.SUM, async_op=True))
        self.done.append((s, e))

    def finish(self, optimizer):
        ws = world_size()
        flat = getattr(optimizer, 'flat', None)
        if flat is None:
            raise RuntimeError('every optimizer on this path owns a FlatParams set (optim.FusedAdam / optim.flat_optimizer)')
        if ws == 1:
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
        for h in self.handles:
            h.wait()
        self.handles, self.current = [], None
        optimizer.dp_scale = 1.0 / ws

    def all_reduce(self, optimizer):
        self.begin(None)
        self.finish(optimizer)
