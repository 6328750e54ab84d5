"""optimize_parameters() replayed from a captured hipGraph (models.StepGraph) against the eager step: same seeded weights, a DIFFERENT
batch every step, a learning-rate change in between -- every loss of every step and the final parameters must be bit-identical, for the bf16 and the
strict policy, DeepLIIF (5 G + 5 D and the 18-net form at fixture width) and DeepLIIFExt."""
import argparse
import types

import pytest
import torch

import bench
from deepliif_amd import models as M

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(autouse=True)
def _one_stream(monkeypatch):
    """StepGraph captures the single-stream step (with the branches on several streams, DL_STREAMS > 1, it declines: test_gpu_streams.py)"""
    monkeypatch.setattr(M, '_N_STREAMS', 1)


def _batches(kind, n, size, count, m):
    g = torch.Generator().manual_seed(77)
    out = []
    for _ in range(count):
        b = {'A': torch.rand(n, 3, size, size, generator=g) * 2 - 1, 'B': [torch.rand(n, 3, size, size, generator=g) * 2 - 1 for _ in range(m)], 'A_paths': ['x']}
        if kind == 'ext':
            b['BS'] = [torch.rand(n, 3, size, size, generator=g) * 2 - 1 for _ in range(2)]
        out.append(b)
    return out


def _build(kind, precision):
    args = argparse.Namespace(ngf=8, norm='instance', precision=precision, batch=2, size=64)
    torch.manual_seed(0)
    if kind == 'train':
        opt = bench.make_opt(args, 0)
    elif kind == 'train18':
        opt = bench.make_opt(args, 0, M=4, seg_gen=True)
        opt.net_gs, opt.norm = 'unet_64', 'batch'
    else:
        opt = bench.make_opt(args, 0, M=2, seg_gen=True)
        opt.model, opt.net_ds, opt.net_gs = 'DeepLIIFExt', 'n_layers', 'unet_64'
        opt.loss_G_weights = opt.loss_D_weights = opt.seg_weights = [0.5, 0.5]
    model = M.create_model(opt)
    model.setup(opt)
    return model


def _flat(model):
    return torch.cat([o.flat.data.clone() for o in model.optimizers])


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
@pytest.mark.parametrize('kind', ['train', 'train18', 'ext'])
def test_replayed_step_is_bit_identical_to_the_eager_step(kind, precision):
    m = {'train': 5, 'train18': 5, 'ext': 2}[kind]
    batches = _batches(kind, 2, 64, 6, m)
    eager, graphed = _build(kind, precision), _build(kind, precision)
    assert torch.equal(_flat(eager), _flat(graphed))
    sg = M.StepGraph(graphed, warmup=2)
    assert sg.why_eager is None
    for i, b in enumerate(batches):
        if i == 4:                                   # a scheduler step between two replays: the learning rate reaches the graph through device memory
            for mdl in (eager, graphed):
                for o in mdl.optimizers:
                    o.param_groups[0]['lr'] *= 0.5
        eager.set_input({k: ([t.to(DEV) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else (v.to(DEV) if torch.is_tensor(v) else v)) for k, v in b.items()})
        eager.optimize_parameters()
        sg.step(b)
        torch.cuda.synchronize()
        le, lg = eager.get_current_losses(), graphed.get_current_losses()
        assert le.keys() == lg.keys() and all(le[k] == lg[k] for k in le), (i, {k: (le[k], lg[k]) for k in le if le[k] != lg[k]})
    assert sg.graph is not None and sg.calls == 6, sg.why_eager
    if not torch.equal(_flat(eager), _flat(graphed)):
        bad = []
        for name in eager.model_names:
            for (k, a), (_, b) in zip(getattr(eager, 'net' + name).named_parameters(), getattr(graphed, 'net' + name).named_parameters()):
                if not torch.equal(a, b):
                    bad.append((name, k, float((a - b).abs().max()), float(a.abs().max())))
        raise AssertionError(f'parameters differ after 2 eager + 1 captured + 3 replayed steps: {bad[:12]} ({len(bad)} tensors)')
    assert all(o.step_count == 6 for o in graphed.optimizers)


def test_step_graph_declines_what_it_cannot_capture():
    args = argparse.Namespace(ngf=8, norm='instance', precision='bf16', batch=2, size=64)
    torch.manual_seed(0)
    opt = bench.make_opt(args, 0)
    opt.no_dropout = False                           # Dropout(0.5) in the ResnetBlocks, training mode
    model = M.create_model(opt)
    model.setup(opt)
    sg = M.StepGraph(model)
    assert sg.why_eager and 'dropout' in sg.why_eager
    b = _batches('train', 2, 64, 1, 5)[0]
    sg.step(b)                                       # still trains, eagerly
    torch.cuda.synchronize()
    assert all(v == v for v in model.get_current_losses().values())


def test_replays_without_a_synchronize_in_between():
    """the Adam scalars of step N are staged in pinned memory and copied asynchronously; the host runs ahead of the GPU when nothing
    synchronises, so the staging buffer of step N must not be rewritten before its copy has executed (optim.FusedAdam: ring of pinned slots guarded
    by events).  Six replays back to back, a learning-rate change in the middle, ONE synchronize at the end -- against the eager run, bit for bit"""
    batches = _batches('train', 2, 64, 10, 5)
    eager, graphed = _build('train', 'bf16'), _build('train', 'bf16')
    sg = M.StepGraph(graphed, warmup=2)
    on_dev = [{k: ([t.to(DEV) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else (v.to(DEV) if torch.is_tensor(v) else v)) for k, v in b.items()}
              for b in batches]
    torch.cuda.synchronize()
    for i, b in enumerate(on_dev):
        if i in (5, 7):
            for o in graphed.optimizers:
                o.param_groups[0]['lr'] *= 0.5
        sg.step(b)                                   # steps 0-1 eager, 2 captured, 3.. replayed: no synchronize from here on
    torch.cuda.synchronize()
    for i, b in enumerate(on_dev):
        if i in (5, 7):
            for o in eager.optimizers:
                o.param_groups[0]['lr'] *= 0.5
        eager.set_input(b)
        eager.optimize_parameters()
    torch.cuda.synchronize()
    assert sg.graph is not None and sg.calls == 10
    assert torch.equal(_flat(eager), _flat(graphed))


def test_a_batch_of_another_shape_runs_eagerly_and_the_graph_survives():
    """the last batch of an epoch may be smaller (the reference's loaders do not drop it); StepGraph must not copy it into the captured
    tensors (a remainder of 1 would broadcast silently)"""
    big, small = _batches('train', 2, 64, 5, 5), _batches('train', 1, 64, 1, 5)
    order = big[:4] + small + big[4:]
    eager, graphed = _build('train', 'bf16'), _build('train', 'bf16')
    sg = M.StepGraph(graphed, warmup=2)
    for b in order:
        eager.set_input(b)
        eager.optimize_parameters()
        sg.step(b)
        torch.cuda.synchronize()
        le, lg = eager.get_current_losses(), graphed.get_current_losses()
        assert all(le[k] == lg[k] for k in le)
    assert sg.graph is not None and sg.eager_steps == 1
    assert torch.equal(_flat(eager), _flat(graphed)) and all(o.step_count == 6 for o in graphed.optimizers)


def test_a_larger_batch_drops_the_graph_and_captures_again(monkeypatch):
    """an odd batch that needs MORE scratch than the captured one makes the grow-only scratch buffers / the slab arena reallocate; the captured graph
    then holds freed addresses.  StepGraph watches ops.Workspace.realloc_generation, drops the graph and captures again on the next fitting batches -- every step
    stays bit-identical to the eager model.  (Whether a larger batch really outgrows a buffer depends on what ran earlier in the process -- the buffers are
    grow-only per thread -- so the odd step here asks for twice the floats of every workspace it touches: the reallocation is certain.)"""
    from deepliif_amd import ops
    small, big = _batches('train', 1, 64, 8, 5), _batches('train', 3, 96, 1, 5)
    order = small[:4] + big + small[4:]
    eager, graphed = _build('train', 'bf16'), _build('train', 'bf16')
    sg = M.StepGraph(graphed, warmup=2)
    orig_get = ops.Workspace.get
    for i, b in enumerate(order):
        eager.set_input(b)
        eager.optimize_parameters()
        if i == 4:
            assert sg.graph is not None
            gen0 = ops.Workspace.realloc_generation
            monkeypatch.setattr(ops.Workspace, 'get', lambda self, name, n, dev: orig_get(self, name, 2 * int(n) + 4096, dev))
        sg.step(b)
        torch.cuda.synchronize()
        if i == 4:
            monkeypatch.setattr(ops.Workspace, 'get', orig_get)
            assert ops.Workspace.realloc_generation > gen0
            assert sg.graph is None and sg.recaptures == 1
        le, lg = eager.get_current_losses(), graphed.get_current_losses()
        assert all(le[k] == lg[k] for k in le), (i, le, lg)
    assert sg.graph is not None and sg.eager_steps == 1
    assert torch.equal(_flat(eager), _flat(graphed)) and all(o.step_count == len(order) for o in graphed.optimizers)
