"""The gradient-exchange path on REAL hardware with one rank: gpurun boxes have a single MI355X, so RCCL over xGMI itself is the driver's to
run -- but everything around the collective can be exercised here: torch.distributed's 'nccl' backend (= RCCL) with a 1-rank process group,
asynchronous all-reduces launched from the backward tape on RCCL's own stream while the ctypes-launched HIP kernels keep running on
torch's current stream, the waits before the optimizer step, the 1/world scaling.  A 1-rank all-reduce is the identity, so the step with the
exchange forced on (DL_DP_FORCE) must be BIT-IDENTICAL to the plain single-process step; a stream-ordering bug (an all-reduce reading a
gradient slice before the kernels that produce it have finished, or Adam running before the collective is done) shows up as a difference."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(steps, precision):
    from deepliif_amd import models as M
    from golden_util import seeded_uniform
    from test_gpu_networks import make_opt
    torch.manual_seed(3)
    opt = make_opt(2, True, 'batch', 'unet_64', 16, precision)
    model = M.create_model(opt)
    model.setup(opt)
    A = seeded_uniform((2, 3, 64, 64), 22)
    B = [seeded_uniform((2, 3, 64, 64), 23 + i) for i in range(3)]
    losses, logs = [], []
    for _ in range(steps):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        losses.append(dict(model.get_current_losses()))
        logs.append(list(model.exchange.launch_log))
    torch.cuda.synchronize()
    flat = torch.cat([o.flat.data.clone() for o in model.optimizers])
    return flat, losses, logs, model


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
@pytest.mark.parametrize('progressive', [False, True], ids=['one-message-per-network', 'progressive-buckets'])
def test_one_rank_rccl_exchange_is_bit_identical_to_no_exchange(monkeypatch, precision, progressive):
    import torch.distributed as dist
    from deepliif_amd import distributed as D
    ref_flat, ref_losses, ref_logs, _ = _run(3, precision)
    assert all(len(l) == 0 for l in ref_logs)                       # no process group: nothing was exchanged
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    try:
        monkeypatch.setattr(D, 'FORCE', True)
        if progressive:
            monkeypatch.setattr(D, 'SPLIT_ELEMS', 50000)
            monkeypatch.setattr(D, 'BUCKET_ELEMS', 100000)
        assert D.active()
        flat, losses, logs, model = _run(3, precision)
        assert torch.equal(flat, ref_flat), 'the step with the (identity) all-reduces in flight differs from the plain step'
        assert losses == ref_losses
        # the exchange really ran: every generator slice of the last backward_G was launched from the tape (before finish())
        g = model.optimizer_G.flat
        slices = [g.slice_of(list(getattr(model, 'net' + n).parameters())) for n in model.model_names_g + model.model_names_gs]
        covered = sum(b - a for a, b in logs[-1])
        assert covered == sum(e - s for s, e in slices), (logs[-1], slices)
        if progressive:
            assert len(logs[-1]) > len(slices)
        assert model.optimizer_G.dp_scale == 1.0 and model.optimizer_D.dp_scale == 1.0
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_exchange_with_bf16_gradients_on_the_wire(monkeypatch):
    """DL_DP_GRAD_BF16 on RCCL with one rank: the collective is the identity, so the step equals the plain step with every gradient rounded to bf16 right
    before Adam -- checked against exactly that (the flat gradient buffers of a plain model rounded by hand), stream ordering of the rounding copy, the
    all-reduce on RCCL's stream and the widening copy included."""
    import torch.distributed as dist
    from deepliif_amd import distributed as D
    from deepliif_amd import optim
    # reference: plain model, gradients rounded to bf16 in front of every optimizer step
    orig_step = optim.FusedAdam.step

    def rounding_step(self, *a, **k):
        self.flat.grad.copy_(self.flat.grad.to(torch.bfloat16))
        return orig_step(self, *a, **k)
    monkeypatch.setattr(optim.FusedAdam, 'step', rounding_step)
    ref_flat, ref_losses, _, _ = _run(3, 'bf16')
    monkeypatch.setattr(optim.FusedAdam, 'step', orig_step)
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    try:
        monkeypatch.setattr(D, 'FORCE', True)
        monkeypatch.setattr(D, 'GRAD_BF16', True)
        flat, losses, logs, model = _run(3, 'bf16')
        assert torch.equal(flat, ref_flat), 'bf16 wire format: the step differs from "round every gradient, then step"'
        assert losses == ref_losses
        assert model.exchange.pass_log[-1]['bytes'] == 2 * model.optimizer_G.flat.numel
    finally:
        dist.destroy_process_group()
