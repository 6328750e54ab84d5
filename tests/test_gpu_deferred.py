"""Deferred reduction of the weight gradient's split-K slabs (include/deepliif_hip.h: dl_conv_wgrad_slabs / dl_wgrad_reduce_batch, ops.HipBackend.wgrad_flush):
inside Tape.backward() the slabs of many layers lie side by side in an arena and ONE launch reduces them.  Per element the summation order is the
immediate kernel's, so everything here is BIT-identical to DL_WGRAD_DEFER=0: single layers (bf16 and strict, both operand roles, stacked kernel columns),
a gradient that is accumulated twice in one pass (flush before the second use), an arena too small for the pass (wrap-around flushes), and whole
training steps of DeepLIIF / DeepLIIFExt (losses of every step and the final parameters)."""
import pytest
import torch

from deepliif_amd import _lib as L
from deepliif_amd import ops
from test_gpu_graph import _batches, _build, _flat

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# (Cout_p, Cin_p, k, stride, N, H): P = dL/dy on the output grid, Q = x
LAYERS = [(256, 256, 3, 1, 2, 64), (128, 64, 3, 2, 2, 64), (64, 128, 4, 2, 1, 64), (512, 256, 4, 1, 1, 32), (64, 8, 7, 1, 1, 64), (16, 64, 3, 1, 2, 32)]


def _pad(layer):
    _, _, k, stride, _, _ = layer
    return (k - 1) // 2 if (stride == 1 and k != 4) else 1


def _operands(layer, dtype, seed):
    cout, cin, k, stride, n, h = layer
    g = torch.Generator().manual_seed(seed)
    ho = (h + 2 * _pad(layer) - k) // stride + 1
    P = (torch.randn(n, ho, ho, cout, generator=g) * 0.5).to(DEV).to(dtype)
    Q = torch.randn(n, h, h, cin, generator=g).to(DEV).to(dtype)
    grad0 = torch.randn(cout, cin, k, k, generator=g).to(DEV)
    return P, Q, grad0


def _wgrad(be, layer, P, Q, grad, prec):
    _, _, k, stride, _, _ = layer
    be.conv_wgrad(P, Q, grad, k, stride, _pad(layer), L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec, True)


@pytest.fixture
def be():
    ops._impl = None
    b = ops.impl()
    st = ops.WS._state()
    assert st.get('defer_depth', 0) == 0 and not st.get('defer_pending')
    yield b
    assert st.get('defer_depth', 0) == 0 and not st.get('defer_pending')


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
def test_batched_reduction_equals_the_immediate_one(be, precision, monkeypatch):
    dtype, prec = (torch.bfloat16, L.PREC_BF16) if precision == 'bf16' else (torch.float32, L.PREC_BF16X3)
    ops_ = [_operands(l, dtype, 100 + i) for i, l in enumerate(LAYERS)]
    monkeypatch.setattr(ops, '_WGRAD_DEFER', False)
    ref = []
    for l, (P, Q, g0) in zip(LAYERS, ops_):
        g = g0.clone()
        _wgrad(be, l, P, Q, g, prec)
        ref.append(g)
    monkeypatch.setattr(ops, '_WGRAD_DEFER', True)
    got = [g0.clone() for _, _, g0 in ops_]
    be.wgrad_defer_begin()
    for l, (P, Q, _), g in zip(LAYERS, ops_, got):
        _wgrad(be, l, P, Q, g, prec)
    pending = len(ops.WS._state()['defer_pending'])
    assert pending >= 4                                  # the narrow persistent forms (if any of these shapes take them) reduce in place
    assert not torch.equal(got[0], ref[0])               # nothing has been reduced yet
    be.wgrad_defer_end()
    torch.cuda.synchronize()
    for l, a, b in zip(LAYERS, got, ref):
        assert torch.equal(a, b), (l, float((a - b).abs().max()))


def test_stacked_kernel_columns_through_the_batch(be, monkeypatch):
    """the narrow head's weight gradient (engine.conv: dl_shift_stack image, KH x 1 taps, grad[a][b][kh][kw] scattered by the reduction)"""
    g = torch.Generator().manual_seed(7)
    P = torch.randn(2, 64, 64, 24, generator=g).to(DEV).to(torch.bfloat16)          # 3 output channels x 7 kernel columns, padded to 24
    Q = torch.randn(2, 64, 64, 64, generator=g).to(DEV).to(torch.bfloat16)
    g0 = torch.randn(3, 64, 7, 7, generator=g).to(DEV)
    out = []
    for defer in (False, True):
        monkeypatch.setattr(ops, '_WGRAD_DEFER', defer)
        grad = g0.clone()
        be.wgrad_defer_begin()
        be.conv_wgrad(P, Q, grad, 7, 1, 3, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, True, stack_kw=7)
        assert len(ops.WS._state().get('defer_pending') or []) == (1 if defer else 0)
        be.wgrad_defer_end()
        out.append(grad)
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1]) and not torch.equal(out[0], g0)


def test_second_use_of_a_gradient_and_a_small_arena(be, monkeypatch):
    l = LAYERS[0]
    prec = L.PREC_BF16
    sets = [_operands(l, torch.bfloat16, 200 + i) for i in range(5)]
    monkeypatch.setattr(ops, '_WGRAD_DEFER', False)
    shared_ref = sets[0][2].clone()
    _wgrad(be, l, sets[0][0], sets[0][1], shared_ref, prec)
    _wgrad(be, l, sets[1][0], sets[1][1], shared_ref, prec)
    refs = []
    for P, Q, g0 in sets[2:]:
        g = g0.clone()
        _wgrad(be, l, P, Q, g, prec)
        refs.append(g)
    monkeypatch.setattr(ops, '_WGRAD_DEFER', True)
    st = ops.WS._state()
    # an arena that holds ONE slab set of this layer (+ a little): every further layer wraps around after a flush
    shared = sets[0][2].clone()
    be.wgrad_defer_begin()
    _wgrad(be, l, sets[0][0], sets[0][1], shared, prec)
    one = st['defer_off']
    be.wgrad_defer_end()
    monkeypatch.setitem(st, 'defer_arena', torch.empty(one + 4096, dtype=torch.float32, device=DEV))
    shared = sets[0][2].clone()
    got = [g0.clone() for _, _, g0 in sets[2:]]
    flushes = []
    real_reduce = be._reduce_pending            # (a second use goes through wgrad_flush, a full arena straight to the batched reduction: both end here)
    monkeypatch.setattr(be, '_reduce_pending', lambda s: (flushes.append(len(s.get('defer_pending') or [])), real_reduce(s))[1])
    be.wgrad_defer_begin()
    _wgrad(be, l, sets[0][0], sets[0][1], shared, prec)
    _wgrad(be, l, sets[1][0], sets[1][1], shared, prec)          # same gradient again: the first accumulation is flushed before this one is queued
    for (P, Q, _), g in zip(sets[2:], got):
        _wgrad(be, l, P, Q, g, prec)
    be.wgrad_defer_end()
    torch.cuda.synchronize()
    assert [f for f in flushes if f] == [1] * 5, flushes
    assert torch.equal(shared, shared_ref)
    for a, b in zip(got, refs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('kind,precision', [('train', 'bf16'), ('train', 'fp32'), ('ext', 'bf16'), ('train18', 'fp32')])
def test_training_steps_do_not_depend_on_the_deferral(kind, precision, monkeypatch):
    m = {'train': 5, 'train18': 5, 'ext': 2}[kind]
    batches = _batches(kind, 2, 64, 3, m)
    results = []
    for defer in (False, True):
        ops._impl = None
        monkeypatch.setattr(ops, '_WGRAD_DEFER', defer)
        model = _build(kind, precision)
        be = ops.impl()
        batched = []
        real_flush = be.wgrad_flush
        monkeypatch.setattr(be, 'wgrad_flush', lambda: (batched.append(len(ops.WS._state().get('defer_pending') or [])), real_flush())[1])
        losses = []
        for b in batches:
            model.set_input({k: ([t.to(DEV) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else (v.to(DEV) if torch.is_tensor(v) else v)) for k, v in b.items()})
            model.optimize_parameters()
            torch.cuda.synchronize()
            losses.append(dict(model.get_current_losses()))
        results.append((losses, _flat(model), sum(batched)))
    (l0, p0, n0), (l1, p1, n1) = results
    assert l0 == l1
    assert torch.equal(p0, p1)
    assert n0 == 0 and n1 >= 3 * 20          # the deferred run really went through the batched reduction: every general-path layer of every pass
