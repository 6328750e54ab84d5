"""Round 5: the one-wave-per-SIMD weight gradient of the ResnetBlock shape (csrc/wgrad_w4.h) and the batched launch (dl_conv_wgrad_multi,
ops.HipBackend._wgrad_queued).  Parity against an fp64 restatement of the reference's conv backward-weight (networks.py:467-513 reached from
DeepLIIF_model.py:332,429: torch's conv2d weight gradient) on the SAME bf16-rounded operands; the batch against single launches bit for bit."""
import pytest
import torch
import torch.nn.functional as F

from deepliif_amd import _lib as L
from deepliif_amd import ops
from deepliif_amd.geometry import choose_wgrad_batch_splitk

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ops(n, h, ca, cb, seed, dtype=torch.bfloat16, w=128):
    g = torch.Generator().manual_seed(seed)
    P = (torch.randn(n, h, w, ca, generator=g) * 0.5).to(DEV).to(dtype)
    Q = torch.randn(n, h, w, cb, generator=g).to(DEV).to(dtype)
    return P, Q


def _reference(P, Q):
    """dW[a][b][kh][kw] = sum_p P[p][a] * Q[p + (kh-1, kw-1)][b], zero padding 1 -- fp64 on the GPU (torch's own conv backward-weight)"""
    x = Q.double().permute(0, 3, 1, 2).contiguous()
    dy = P.double().permute(0, 3, 1, 2).contiguous()
    return torch.nn.grad.conv2d_weight(x, (P.shape[3], Q.shape[3], 3, 3), dy, stride=1, padding=1)


@pytest.fixture
def be():
    ops._impl = None
    b = ops.impl()
    yield b
    st = ops.WS._state()
    assert st.get('defer_depth', 0) == 0 and not st.get('defer_pending') and not st.get('defer_queue')


def _plan(be, P, Q):
    d = L.WgradDesc()
    d.N, d.Hp, d.Wp, d.CAp = P.shape
    d.p_pstride = P.shape[3]
    _, d.Hq, d.Wq, d.CBp = Q.shape
    d.q_pstride = Q.shape[3]
    d.KH = d.KW = 3
    d.step, d.pad, d.pad_mode = 1, 1, L.PAD_ZERO
    d.CA, d.CB = P.shape[3], Q.shape[3]
    d.pad_w = -1
    d.dtype, d.prec = (L.DL_BF16, L.PREC_BF16) if P.dtype == torch.bfloat16 else (L.DL_F32, L.PREC_BF16X3)
    d.splitk = 1
    import ctypes as C
    t, k, nm = L.i32(), L.i32(), C.c_char_p()
    rc = be.lib.dl_wgrad_plan(C.byref(d), C.byref(t), C.byref(k), C.byref(nm))
    return rc, t.value, k.value, nm.value.decode()


@pytest.mark.parametrize('shape', [(2, 16, 256, 256), (1, 6, 128, 256), (3, 5, 256, 128), (1, 2, 128, 128)])
@pytest.mark.parametrize('splitk', [None, 1, 3, 7])
def test_w4_weight_gradient_against_fp64(be, shape, splitk):
    n, h, ca, cb = shape
    P, Q = _ops(n, h, ca, cb, 11)
    rc, tiles, ksteps, name = _plan(be, P, Q)
    assert (rc, name, tiles, ksteps) == (1, 'wgrad_w4_kernel', (ca // 128) * (cb // 128) * 3, n * h)
    if splitk is not None and splitk > n * h:
        pytest.skip('more row ranges than rows')
    g0 = torch.randn(ca, cb, 3, 3, device=DEV)
    grad = g0.clone()
    be.conv_wgrad(P, Q, grad, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, True, splitk=splitk)
    ref = _reference(P, Q)
    err = float(((grad - g0).double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err                     # exact bf16 products, fp32 accumulation over <= 2048 x 16 pixels per partial, fixed-order fp32 combine
    # not accumulating overwrites
    grad2 = torch.full_like(g0, 7.0)
    be.conv_wgrad(P, Q, grad2, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, False, splitk=splitk)
    assert torch.equal(grad2, grad - g0) or float((grad2 - (grad - g0)).abs().max()) < 1e-5 * float(ref.abs().max())
    # run-to-run determinism
    grad3 = torch.zeros_like(g0)
    be.conv_wgrad(P, Q, grad3, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, False, splitk=splitk)
    assert torch.equal(grad2, grad3)


def test_w4_channel_slice_operands(be):
    """operands that are channel slices of wider buffers (pixel stride > channels): the kernel addresses pixels by pstride"""
    n, h = 2, 8
    g = torch.Generator().manual_seed(5)
    Pw = torch.randn(n, h, 128, 384, generator=g).to(DEV).to(torch.bfloat16)
    Qw = torch.randn(n, h, 128, 512, generator=g).to(DEV).to(torch.bfloat16)
    P, Q = Pw[..., 128:384], Qw[..., 256:384]
    grad = torch.zeros(256, 128, 3, 3, device=DEV)
    be.conv_wgrad(P, Q, grad, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, False)
    ref = _reference(P.contiguous(), Q.contiguous())
    assert float((grad.double() - ref).abs().max() / ref.abs().max()) < 2e-6


def test_strict_layers_are_not_queued(be, monkeypatch):
    """the batch is the w4 kernel's (ops.HipBackend.conv_wgrad: measured, the strict glds_x3 kernel gains nothing from it): a strict ResnetBlock layer inside a
    pass launches its split-K kernel at once and only its reduction is deferred"""
    monkeypatch.setattr(ops, '_WGRAD_BATCH', True)
    monkeypatch.setattr(ops, '_WGRAD_DEFER', True)
    P, Q = _ops(2, 16, 256, 256, 7, torch.float32)
    g = torch.zeros(256, 256, 3, 3, device=DEV)
    ref = torch.zeros_like(g)
    be.conv_wgrad(P, Q, ref, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16X3, False)
    be.wgrad_defer_begin()
    be.conv_wgrad(P, Q, g, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16X3, False)
    st = ops.WS._state()
    assert not st.get('defer_queue') and len(st['defer_pending']) == 1
    be.wgrad_defer_end()
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    r64 = _reference(P, Q)
    assert float((g.double() - r64).abs().max() / r64.abs().max()) < 1e-5


@pytest.mark.parametrize('precision', ['bf16'])
def test_batched_launch_is_bit_identical_to_single_launches(be, precision, monkeypatch):
    dtype, prec = (torch.bfloat16, L.PREC_BF16) if precision == 'bf16' else (torch.float32, L.PREC_BF16X3)
    n, h, ca, cb, layers = 2, 16, 256, 256, 5
    data = [_ops(n, h, ca, cb, 40 + i, dtype) for i in range(layers)]
    rc, tiles, ksteps, name = _plan(be, *data[0])
    assert rc == 1
    sk = choose_wgrad_batch_splitk(tiles, ksteps)
    g0 = [torch.randn(ca, cb, 3, 3, device=DEV) for _ in range(layers)]
    ref = [g.clone() for g in g0]
    for (P, Q), g in zip(data, ref):
        be.conv_wgrad(P, Q, g, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec, True, splitk=sk)
    monkeypatch.setattr(ops, '_WGRAD_BATCH', True)
    monkeypatch.setattr(ops, '_WGRAD_DEFER', True)
    got = [g.clone() for g in g0]
    be.wgrad_defer_begin()
    for (P, Q), g in zip(data, got):
        be.conv_wgrad(P, Q, g, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec, True)
    st = ops.WS._state()
    assert len(st['defer_queue']) == layers and torch.equal(got[0], g0[0])          # nothing has run yet
    del data[1:]                                                                     # the queue keeps the operands alive
    be.wgrad_defer_end()
    torch.cuda.synchronize()
    for a, b in zip(got, ref):
        assert torch.equal(a, b), float((a - b).abs().max())
    if precision == 'fp32':
        r64 = _reference(*_ops(n, h, ca, cb, 40, dtype))
        assert float(((got[0] - g0[0]).double() - r64).abs().max() / r64.abs().max()) < 1e-5


def _model(precision):
    import argparse
    import bench
    from deepliif_amd import models as M
    args = argparse.Namespace(ngf=32, norm='instance', precision=precision, batch=1, size=512)       # 4 * ngf = 128 channels at 128-pixel rows: the w4 / batch shape
    torch.manual_seed(0)
    opt = bench.make_opt(args, 0, M=2)
    model = M.create_model(opt)
    model.setup(opt)
    return model


def _step_batches(count, m):
    g = torch.Generator().manual_seed(77)
    return [{'A': torch.rand(1, 3, 512, 512, generator=g) * 2 - 1, 'B': [torch.rand(1, 3, 512, 512, generator=g) * 2 - 1 for _ in range(m)], 'A_paths': ['x']}
            for _ in range(count)]


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
def test_training_step_with_and_without_the_batch(precision, monkeypatch):
    """whole DeepLIIF steps (2 Resnet generators + 2 discriminators, 128 channels at 128-pixel rows: the ResnetBlock layers take the batch, under the
    default branch streams).  The batched pass differs from the per-layer pass only by the split-K factor of those layers -- same losses to fp32 summation
    noise -- and two batched runs are bit-identical"""
    outs = {}
    for tag, batch in (('off', False), ('on', True), ('on2', True)):
        monkeypatch.setattr(ops, '_WGRAD_BATCH', batch)
        m = _model(precision)
        losses = []
        for b in _step_batches(2, 2):
            m.set_input(b)
            m.optimize_parameters()
            losses.append([float(v) for v in m.get_current_losses().values()])
        outs[tag] = (losses, torch.cat([o.flat.data.clone() for o in m.optimizers]))
        del m
    assert outs['on'][0] == outs['on2'][0] and torch.equal(outs['on'][1], outs['on2'][1])
    for a, b in zip(outs['on'][0], outs['off'][0]):
        for x, y in zip(a, b):
            assert abs(x - y) <= 2e-3 * max(1.0, abs(y)), (x, y)
    d = (outs['on'][1] - outs['off'][1]).abs().max()
    assert float(d) < 1e-3, float(d)            # Adam steps of 2e-4: a handful of sign-level flips at most


@pytest.mark.parametrize('progressive', [False, True], ids=['one-message-per-network', 'progressive-buckets'])
def test_batched_weight_gradient_under_the_gradient_exchange(progressive, monkeypatch):
    """The queued weight gradients of a network must have RUN before its slice goes on the wire: the tape marker flushes them and GradExchanger._launch flushes
    again (progressive buckets cut a network's batch in the middle -- a layer's split-K does not depend on the size of its batch, so nothing may change).
    One-rank RCCL group, exchange forced on (an identity all-reduce), three branch streams, the w4 / batch shape: bit-identical to the plain step."""
    import torch.distributed as dist
    from deepliif_amd import distributed as D
    from test_gpu_distributed import _free_port
    monkeypatch.setattr(ops, '_WGRAD_BATCH', True)

    def run():
        m = _model('bf16')
        for b in _step_batches(2, 2):
            m.set_input(b)
            m.optimize_parameters()
        torch.cuda.synchronize()
        return torch.cat([o.flat.data.clone() for o in m.optimizers]), [float(v) for v in m.get_current_losses().values()], m
    ref_flat, ref_losses, _ = run()
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    try:
        monkeypatch.setattr(D, 'FORCE', True)
        if progressive:
            monkeypatch.setattr(D, 'SPLIT_ELEMS', 200000)
            monkeypatch.setattr(D, 'BUCKET_ELEMS', 400000)
        assert D.active()
        flat, losses, m = run()
        assert len(m.exchange.launch_log) >= 2                       # slices left from the tape of the last backward_G
        assert losses == ref_losses
        assert torch.equal(flat, ref_flat)
    finally:
        dist.destroy_process_group()
