"""Data-parallel gradient exchange: one process per GPU (torchrun env contract: RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT), torch.distributed backend 'nccl' (= RCCL over xGMI on ROCm) on GPUs, 'gloo' in CPU tests.

The reference wraps each of its 18 networks in DistributedDataParallel (networks.py:131-134) -> 18 reducers with 25 MB
buckets.  Here every optimizer owns ONE flat fp32 gradient buffer (optim.FlatParams), so the exchange is a handful of large
sum-all-reduces over contiguous slices (xGMI is point-to-point: few large messages beat many small ones); the 1/world_size
averaging is folded into the Adam kernel (grad_scale).  BatchNorm statistics stay per-rank, exactly like the reference's
un-synchronised DDP (broadcast_buffers=False, no SyncBatchNorm).
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.distributed as dist

BUCKET_ELEMS = 64 * 1024 * 1024      # 256 MB fp32 per all-reduce call


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


class GradExchanger:
    """Sum-all-reduce of an optimizer's gradients across ranks; averaging is applied inside the optimizer step."""

    def __init__(self):
        self.handles: List = []

    def all_reduce(self, optimizer):
        ws = world_size()
        flat = getattr(optimizer, 'flat', None)
        if ws == 1:
            if flat is not None:
                optimizer.grad_scale = 1.0
            return
        if flat is not None:
            g = flat.grad
            handles = []
            for s in range(0, g.numel(), BUCKET_ELEMS):
                handles.append(dist.all_reduce(g[s:s + BUCKET_ELEMS], op=dist.ReduceOp.SUM, async_op=True))
            for h in handles:
                h.wait()
            optimizer.grad_scale = 1.0 / ws
        else:       # a torch.optim optimizer chosen with --optimizer: average parameter-wise
            for group in optimizer.param_groups:
                for p in group['params']:
                    if p.grad is not None:
                        dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                        p.grad.div_(ws)
