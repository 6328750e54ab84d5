"""Branch streams (DL_STREAMS=N, models.BaseModel._branch_streams): the independent (G_i, D_i) branches of a DeepLIIF training step spread over N HIP
streams -- every branch's forward, losses and backward stay in ONE in-order stream (engine.Tape runs a node's backward on the stream it was recorded on),
scratch buffers / statistics workspaces / slab arenas are per stream (ops.Workspace), phases fork from and join into the main stream around the
optimizer steps.  Nothing about a branch's arithmetic or summation order changes, so N streams must be BIT-identical to one: every loss of every step
and the final parameters, bf16 and strict policy, N = 2, 3 and 5 (one stream per branch); the same with the data-parallel exchange in flight (one-rank
RCCL group: every network's all-reduce is issued from a tape node of its branch, behind THAT stream); and the models whose branches are NOT independent decline."""
import pytest
import torch

from deepliif_amd import models as M
from deepliif_amd import ops
from test_gpu_graph import _batches, _build, _flat

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(autouse=True)
def _reset():
    ops._impl = None
    ops.WS.forget_branch_streams()       # (models of earlier test files leave their branch streams' scratch states behind)
    yield
    ops.WS.forget_branch_streams()


def _run(model, batches):
    losses = []
    for b in batches:
        model.set_input({k: ([t.to(DEV) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else (v.to(DEV) if torch.is_tensor(v) else v)) for k, v in b.items()})
        model.optimize_parameters()
        torch.cuda.synchronize()
        losses.append(dict(model.get_current_losses()))
    return losses, _flat(model), [getattr(model, k).clone() for k in sorted(vars(model)) if k.startswith('fake_B_') and torch.is_tensor(getattr(model, k))]


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
@pytest.mark.parametrize('nstreams', [2, 3, 5])
def test_branch_streams_are_bit_identical_to_one_stream(nstreams, precision, monkeypatch):
    batches = _batches('train', 2, 64, 4, 5)
    monkeypatch.setattr(M, '_N_STREAMS', 1)
    ref_model = _build('train', precision)
    assert ref_model._branch_streams() is None
    ref = _run(ref_model, batches)
    monkeypatch.setattr(M, '_N_STREAMS', nstreams)
    model = _build('train', precision)
    got = _run(model, batches)
    assert model._streams is not None and len(model._streams) == nstreams
    used = [s for s in ops.WS.stream_states() if s.get('bufs')]
    assert len(used) == nstreams                      # every branch stream worked out of its own scratch state (the main stream keeps the thread's)
    assert got[0] == ref[0]
    assert torch.equal(got[1], ref[1])
    for a, b in zip(got[2], ref[2]):
        assert torch.equal(a, b)


def test_switches_keep_dependent_models_on_one_stream(monkeypatch):
    monkeypatch.setattr(M, '_N_STREAMS', 3)
    monkeypatch.setattr(M, '_SEG_STREAMS', False)
    monkeypatch.setattr(M, '_EXT_STREAMS', False)
    seg = _build('train18', 'bf16')                   # segmentation generators read the other branches' outputs: DL_STREAMS_SEG=0 keeps round 3's single stream
    assert seg.branch_parallel is False and seg._branch_streams() is None
    ext = _build('ext', 'bf16')
    assert not ext.branch_parallel and ext._branch_streams() is None
    sg = M.StepGraph(_build('train', 'bf16'))
    assert sg.why_eager and 'streams' in sg.why_eager


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
@pytest.mark.parametrize('nstreams', [2, 3])
def test_ext_model_chains_on_streams_are_bit_identical(nstreams, precision, monkeypatch):
    """DeepLIIFExt (round 5): chain i = G_i -> GS_i (+ D_i, DS_i) on stream i % N; fake_1 feeds EVERY seg generator (DeepLIIFExt_model.py:173), so the
    concatenations and -- in backward -- the accumulation of its gradient stay on the main stream between a recorded join and a recorded fork.  2 G + 2 GS
    + 2 D + 2 DS at fixture width, both policies: losses, parameters and images bit-identical to one stream."""
    batches = _batches('ext', 2, 64, 3, 2)
    monkeypatch.setattr(M, '_N_STREAMS', 1)
    ref = _run(_build('ext', precision), batches)
    monkeypatch.setattr(M, '_N_STREAMS', nstreams)
    monkeypatch.setattr(M, '_EXT_STREAMS', True)
    model = _build('ext', precision)
    assert model.branch_parallel
    got = _run(model, batches)
    assert model._streams is not None and len(model._streams) == nstreams
    assert got[0] == ref[0]
    assert torch.equal(got[1], ref[1])
    for a, b in zip(got[2], ref[2]):
        assert torch.equal(a, b)


def test_branch_streams_under_the_gradient_exchange(monkeypatch):
    """DL_STREAMS=3 with a (one-rank, identity) RCCL exchange forced on: the all-reduce of network i must be ordered behind the stream that produced its
    gradients, finish() behind the join -- a race shows up as a difference from the plain single-stream step"""
    import torch.distributed as dist
    from deepliif_amd import distributed as D
    from test_gpu_distributed import _free_port
    batches = _batches('train', 2, 64, 3, 5)
    monkeypatch.setattr(M, '_N_STREAMS', 1)
    ref = _run(_build('train', 'bf16'), batches)
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    try:
        monkeypatch.setattr(D, 'FORCE', True)
        monkeypatch.setattr(M, '_N_STREAMS', 3)
        assert D.active()
        model = _build('train', 'bf16')
        got = _run(model, batches)
        assert model._streams is not None and len(model._streams) == 3
        assert len(model.exchange.launch_log) >= 5          # every generator's slice went out from the tape of the last backward_G
        assert got[0] == ref[0]
        assert torch.equal(got[1], ref[1])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
@pytest.mark.parametrize('nstreams', [3, 5])
def test_seg_model_chains_on_streams_are_bit_identical(nstreams, precision, monkeypatch):
    """DL_STREAMS_SEG (default on since round 5): the model WITH segmentation generators -- chain i = G_i -> GS_i (+ D_i) on its own stream, GS_0 on the next one, the seg
    discriminators and the summed seg image on the main stream; the join in front of the weighted sum is a tape node whose backward makes every branch
    wait for the seg image's gradient.  4 G + 5 GS + 4 D + 5 DS at fixture width, BatchNorm, both policies: bit-identical to one stream."""
    batches = _batches('train18', 2, 64, 3, 5)
    monkeypatch.setattr(M, '_N_STREAMS', 1)
    ref = _run(_build('train18', precision), batches)
    monkeypatch.setattr(M, '_N_STREAMS', nstreams)
    monkeypatch.setattr(M, '_SEG_STREAMS', True)
    model = _build('train18', precision)
    assert model.seg_gen and model.branch_parallel
    got = _run(model, batches)
    assert model._streams is not None and len(model._streams) == nstreams
    assert got[0] == ref[0]
    assert torch.equal(got[1], ref[1])
    for a, b in zip(got[2], ref[2]):
        assert torch.equal(a, b)
