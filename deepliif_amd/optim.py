"""Flat parameter sets and the fused Adam step.

All parameters of one optimizer (the generator set / the discriminator set, DeepLIIF_model.py:128-147) live in ONE flat
fp32 buffer with a matching flat gradient buffer: the optimizer step is a single HBM-streaming kernel (dl_adam_step) and
the data-parallel gradient exchange operates on contiguous slices (deepliif_amd.distributed).  nn.Parameter objects stay
valid: their .data / .grad become views into the flat buffers, so state_dict()/load_state_dict() and any torch.optim
optimizer keep working on them.
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch

from . import ops


class FlatParams:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        assert self.params, 'empty parameter set'
        dev = self.params[0].device
        assert all(p.device == dev and p.dtype == torch.float32 for p in self.params), 'one device, fp32 master weights'
        self.offsets = []
        n = 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4        # keep every tensor 16-byte aligned
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            view = self.data[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[off:off + p.numel()].view(p.shape)
            p._dl_epoch = getattr(p, '_dl_epoch', 0)

    def attached(self) -> bool:
        """False if something (e.g. module.to()) re-allocated a parameter away from the flat buffer."""
        base = self.data.data_ptr()
        return all(p.data_ptr() == base + 4 * off and p.grad is not None and p.grad.data_ptr() == self.grad.data_ptr() + 4 * off
                   for p, off in zip(self.params, self.offsets))

    def zero_grad(self):
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):       # re-attach views a caller may have dropped (zero_grad(set_to_none))
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)

    def bump_epoch(self):
        for p in self.params:
            p._dl_epoch += 1

    def slice_of(self, params: Iterable[torch.nn.Parameter]):
        """(start, end) element range of the flat buffers covered by a contiguous run of this set's parameters."""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), 'parameters of one network must be contiguous in the flat set'
        end = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.params) else self.numel
        return self.offsets[idx[0]], end


_PACK_BATCH = os.environ.get('DL_PACK_BATCH', '1') != '0'
_HYPER_RING = 4            # pinned staging slots of the graph-mode Adam scalars (FusedAdam.enable_graph_mode)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0, amsgrad=False) semantics on a FlatParams set, one kernel per
    step.  A torch lr_scheduler drives param_groups[0]['lr'] as usual (networks.py:55-81, base_model.py:132-141)."""

    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.flat = FlatParams(params)
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self.step_count = 0
        self.dp_scale = 1.0          # set to 1/world_size by the data-parallel driver (sum all-reduce)
        self._pack_batch = None        # engine.PackBatch over this set's conv weights, built at the first step
        # graph mode (models.StepGraph): the per-step scalars live in device memory -- step() launches dl_adam_step_dev, which a captured hipGraph can
        # replay, and prepare_step() (outside the graph) advances the step counter and refreshes them
        self.hyper_dev = None
        self._hyper_ring = None
        self._prepared = False

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()

    # ---- optimizer state in torch.optim.Adam's OWN checkpoint layout, so a file written here loads into torch.optim.Adam over the same
    # parameter list (and the other way round).  The reference never saves optimizer state (base_model.py:190-208 stores the nets only:
    # --continue-train restarts Adam's moments from zero); BaseModel.save_optimizers / load_optimizers add it.
    def state_dict(self):
        g = self.param_groups[0]
        state = {}
        if self.step_count > 0:
            for i, (p, off) in enumerate(zip(self.flat.params, self.flat.offsets)):
                state[i] = {'step': torch.tensor(float(self.step_count)),
                            'exp_avg': self.exp_avg[off:off + p.numel()].view(p.shape).detach().cpu().clone(),
                            'exp_avg_sq': self.exp_avg_sq[off:off + p.numel()].view(p.shape).detach().cpu().clone()}
        group = {k: v for k, v in g.items() if k != 'params'}
        group.update(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        group['params'] = list(range(len(self.flat.params)))
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.flat.params):
            raise ValueError('optimizer state does not match this parameter set (one group over %d parameters expected)' % len(self.flat.params))
        if groups[0].get('weight_decay', 0) or groups[0].get('amsgrad', False) or groups[0].get('maximize', False):
            raise ValueError('FusedAdam implements plain Adam: weight_decay / amsgrad / maximize state cannot be resumed')
        g = self.param_groups[0]
        for k in ('lr', 'betas', 'eps', 'initial_lr'):
            if k in groups[0]:
                g[k] = tuple(groups[0][k]) if k == 'betas' else groups[0][k]
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_count = 0
        steps = set()
        for i, (p, off) in enumerate(zip(self.flat.params, self.flat.offsets)):
            st = sd['state'].get(i, sd['state'].get(str(i)))
            if st is None:
                continue
            if tuple(st['exp_avg'].shape) != tuple(p.shape):
                raise ValueError(f'optimizer state of parameter {i}: shape {tuple(st["exp_avg"].shape)} != {tuple(p.shape)}')
            self.exp_avg[off:off + p.numel()].view(p.shape).copy_(st['exp_avg'])
            self.exp_avg_sq[off:off + p.numel()].view(p.shape).copy_(st['exp_avg_sq'])
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ ({sorted(steps)}): one fused step counter cannot resume that')
        self.step_count = steps.pop() if steps else 0

    def enable_graph_mode(self):
        """per-step scalars from device memory (see __init__); idempotent"""
        if self.hyper_dev is None:
            self.hyper_dev = torch.zeros(8, dtype=torch.float32, device=self.flat.data.device)
            # A RING of pinned staging buffers, each with the event of its last copy: the asynchronous H2D copy reads the pinned memory when
            # the copy EXECUTES on the stream, not when it is enqueued, and a replayed step never synchronises -- with one buffer the host, running
            # a step or more ahead, overwrote step N's scalars with step N+1's before step N's copy had run (a scheduler's new learning rate and
            # the bias corrections landed a step early, nondeterministically).  A slot is rewritten only after its previous copy has completed.
            cuda = self.flat.data.is_cuda
            self._hyper_ring = [(torch.zeros(8, dtype=torch.float32).pin_memory() if cuda else torch.zeros(8, dtype=torch.float32),
                                 torch.cuda.Event() if cuda else None) for _ in range(_HYPER_RING)]
            self._hyper_slot, self._hyper_used = 0, [False] * _HYPER_RING

    def prepare_step(self):
        """graph mode, OUTSIDE the captured region, once per step and BEFORE it runs: advance Adam's step counter and put this step's scalars
        (learning rate of the scheduler, bias corrections, 1 / world size) where the captured dl_adam_step_dev reads them"""
        assert self.hyper_dev is not None, 'enable_graph_mode() first'
        g = self.param_groups[0]
        self.step_count += 1
        slot = self._hyper_slot
        host, ev = self._hyper_ring[slot]
        if ev is not None and self._hyper_used[slot]:
            ev.synchronize()                   # the copy that last read this slot has executed (a no-op unless the host is _HYPER_RING steps ahead)
        ops.impl().adam_hyper(g['lr'], g['betas'][0], g['betas'][1], g['eps'], self.step_count, self.dp_scale, host)
        self.hyper_dev.copy_(host, non_blocking=True)
        if ev is not None:
            ev.record()
            self._hyper_used[slot] = True
        self._hyper_slot = (slot + 1) % _HYPER_RING
        self._prepared = True

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        if not self.flat.attached():
            raise RuntimeError('parameters were moved after the optimizer was built; rebuild the optimizer (FlatParams lost its views)')
        g = self.param_groups[0]
        if self.hyper_dev is not None and getattr(self, 'step_eager_once', False):
            # graph mode, but THIS step runs outside the captured graph (models.StepGraph: a batch of another shape): scalar-argument kernel
            self.step_eager_once = False
            self.step_count += 1
            ops.impl().adam_step(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                                 self.step_count, self.dp_scale)
        elif self.hyper_dev is not None:
            assert self._prepared, 'graph mode: prepare_step() must run before every step (models.StepGraph does)'
            self._prepared = False
            ops.impl().adam_step_dev(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.hyper_dev)
        else:
            self.step_count += 1
            ops.impl().adam_step(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                                 self.step_count, self.dp_scale)
        self.flat.bump_epoch()         # packed bf16 weight images are stale now (engine.ConvLayer.ensure_packed)
        # ... so rebuild the ones that exist in ONE launch instead of one launch per image at their next use (DL_PACK_BATCH=0
        # restores the lazy per-image path).  Measured in round 1 (rocprof, 3 steps x 2 optimizers): 6 batched launches x 302 us
        # = 1.81 ms replace 645 single-image launches x 6.2 us = 4.0 ms, i.e. -0.73 ms/step
        # (profiles/r01/bench_train_kernel_stats_v17_packbatch.csv).  The first version of the batched kernel -- per-element decode,
        # 2-byte stores, 32 blocks per image regardless of size -- took 766 us per launch and LOST 0.6 ms per 3 steps
        # (profiles/r01/bench_train_kernel_stats_v16_packbatch.csv): chunked 16-byte stores and a size-proportional block table fixed it.
        if not _PACK_BATCH:
            return
        if self._pack_batch is None:
            from . import engine            # (engine does not import optim: no cycle, but keep the import local to the hot path's owner)
            self._pack_batch = engine.PackBatch(self.flat.params)
        self._pack_batch.run()


def flat_optimizer(cls):
    """`--optimizer <name>` other than adam (networks.py:46-53 accepts any torch.optim class): the update rule stays torch.optim's
    (ATen -- NOT an MI355X kernel, not the hot path), but the parameter set is flat like FusedAdam's so that the engine's gradient
    writes (`p.grad` views of one buffer), zero_grad and the data-parallel exchange work unchanged."""

    class Flat(cls):
        def __init__(self, params, **kw):
            params = list(params)
            self.flat = FlatParams(params)
            self.dp_scale = 1.0
            super().__init__(params, **kw)

        def zero_grad(self, set_to_none: bool = False):
            self.flat.zero_grad()

        @torch.no_grad()
        def step(self, closure=None):
            if not self.flat.attached():
                raise RuntimeError('parameters were moved after the optimizer was built; rebuild the optimizer (FlatParams lost its views)')
            if self.dp_scale != 1.0:
                self.flat.grad.mul_(self.dp_scale)
            out = super().step(closure)
            self.flat.bump_epoch()
            return out

    Flat.__name__ = 'Flat' + cls.__name__
    return Flat
