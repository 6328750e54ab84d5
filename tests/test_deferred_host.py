"""Host logic of the deferred slab reduction (ops.HipBackend.conv_wgrad / _wgrad_deferred / wgrad_flush) on a GPU-less machine: the library is replaced by a
recorder, so what is under test is the BOOK-KEEPING around dl_conv_wgrad_slabs / dl_wgrad_reduce_batch -- every queued layer reaches exactly one batch,
also across a flush in the middle of a call (second use of a gradient, arena wrap-around: the bug class that drops entries), slab regions of one batch do
not overlap, first-block numbers are the running sum, tables are cached by content, nothing is deferred outside a pass.  Bit-identity on the GPU:
tests/test_gpu_deferred.py."""
import ctypes as C

import pytest
import torch

from deepliif_amd import _lib as L
from deepliif_amd import ops


class StubLib:
    def __init__(self):
        self.slab_calls, self.batches, self.immediate = [], [], []

    def dl_wgrad_slab_floats(self, d):
        d = d._obj
        return d.splitk * d.CAp * d.KH * d.KW * d.CBp

    def dl_conv_wgrad_deferrable(self, d):
        return 1

    def dl_conv_wgrad(self, d, P, Q, grad, slab, stream):
        self.immediate.append(grad.value)
        return 0

    def dl_conv_wgrad_slabs(self, d, P, Q, grad, slab, entry, stream):
        d, e = d._obj, entry._obj
        e.slab, e.grad = slab.value, grad.value
        e.splitk, e.CAp, e.CBp, e.J = d.splitk, d.CAp, d.CBp, d.KH * d.KW * d.CBp
        e.nblocks = 7 + d.CAp // 8                           # any positive number that depends on the layer only
        self.slab_calls.append((slab.value, self.dl_wgrad_slab_floats(C.byref(d)), grad.value))
        return 0

    def dl_wgrad_reduce_batch(self, table, count, total, stream):
        self.batches.append((table.value, count, total))
        return 0

    # round 5: the kernel plan and the batched launch.  The stub calls a layer "batchable" when it has 128-pixel rows and 128-multiple channel counts,
    # i.e. when geometry.wgrad_batch_shape() also says yes
    def dl_wgrad_plan(self, d, tiles, ksteps, name):
        d = d._obj
        big = d.Wp == 128 and d.CAp % 128 == 0 and d.CBp % 128 == 0
        tiles._obj.value = 12 if big else 1
        ksteps._obj.value = d.N * d.Hp if big else max(1, d.N * d.Hp * d.Wp // 32)
        name._obj.value = b'wgrad_w4_kernel' if big else b'wgrad_kernel'
        return 1 if big else 0

    def dl_conv_wgrad_multi(self, d, n, P, Q, grad, slab, entries, stream):
        d = d._obj
        per = self.dl_wgrad_slab_floats(C.byref(d))
        self.multi = getattr(self, 'multi', [])
        self.multi.append((d.splitk, n, [P[i] for i in range(n)], [grad[i] for i in range(n)], slab.value))
        for l in range(n):
            e = entries[l]
            e.slab, e.grad = slab.value + 4 * l * per, grad[l]
            e.splitk, e.CAp, e.CBp, e.J = d.splitk, d.CAp, d.CBp, d.KH * d.KW * d.CBp
            e.nblocks = 5
        return 0


@pytest.fixture
def be(monkeypatch):
    monkeypatch.setattr(ops, 'WS', ops.Workspace())
    monkeypatch.setattr(ops, '_need_cuda', lambda *ts: None)
    monkeypatch.setattr(ops, '_stream', lambda t=None: None)
    monkeypatch.setattr(ops, '_WGRAD_DEFER', True)
    b = ops.HipBackend.__new__(ops.HipBackend)
    b.lib = StubLib()
    b.wgrad_c4_applies = lambda *a, **k: False
    return b


def _layer(cout=16, cin=8, hw=8):
    P = torch.zeros(1, hw, hw, cout, dtype=torch.bfloat16)
    Q = torch.zeros(1, hw, hw, cin, dtype=torch.bfloat16)
    return P, Q, torch.zeros(cout, cin, 3, 3)


def _wgrad(be, P, Q, g):
    be.conv_wgrad(P, Q, g, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, True, splitk=2)


def _entries(be, batch_index):
    """decode the table of batch `batch_index` from the cached device tensor (a CPU tensor here)"""
    tabs = list(ops.WS._state()['defer_tables'].values())
    ptr, count, total = be.lib.batches[batch_index]
    t = next(t for t in tabs if t.data_ptr() == ptr)
    arr = (L.WgradReduceEntry * count).from_buffer_copy(bytes(t.numpy().tobytes()))
    return list(arr), total


def test_nothing_is_deferred_outside_a_pass(be):
    P, Q, g = _layer()
    _wgrad(be, P, Q, g)
    assert be.lib.immediate == [g.data_ptr()] and not be.lib.slab_calls and not be.lib.batches


def test_every_queued_layer_reaches_exactly_one_batch(be):
    layers = [_layer(16, 8), _layer(32, 8), _layer(16, 16), _layer(8, 8)]
    be.wgrad_defer_begin()
    for P, Q, g in layers:
        _wgrad(be, P, Q, g)
    assert not be.lib.batches and len(ops.WS._state()['defer_pending']) == 4
    be.wgrad_defer_end()
    assert len(be.lib.batches) == 1 and not be.lib.immediate
    entries, total = _entries(be, 0)
    assert [e.grad for e in entries] == [g.data_ptr() for _, _, g in layers]
    assert [e.block0 for e in entries] == [0, 9, 9 + 11, 9 + 11 + 9] and total == 9 + 11 + 9 + 8
    spans = sorted((s, s + 4 * n) for s, n, _ in be.lib.slab_calls)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))            # slab regions of one batch do not overlap
    assert not ops.WS._state()['defer_pending'] and ops.WS._thread_state()['defer_depth'] == 0
    # the same pass again: same arena, same gradients -> the cached table is used, nothing new is uploaded
    n_tables = len(ops.WS._state()['defer_tables'])
    be.wgrad_defer_begin()
    for P, Q, g in layers:
        _wgrad(be, P, Q, g)
    be.wgrad_defer_end()
    assert len(ops.WS._state()['defer_tables']) == n_tables and be.lib.batches[1] == be.lib.batches[0]


def test_second_use_and_wrap_around_flush_without_losing_the_new_entry(be, monkeypatch):
    P, Q, g_shared = _layer(16, 8)
    others = [_layer(16, 8) for _ in range(3)]
    one = int(be.lib.dl_wgrad_slab_floats(C.byref(_desc_of(P, Q, g_shared))))
    monkeypatch.setattr(ops, '_WGRAD_ARENA_MB', 0)                         # the arena is sized by the first request ...
    be.wgrad_defer_begin()
    _wgrad(be, P, Q, g_shared)
    arena = ops.WS._state()['defer_arena']
    assert arena.numel() == (one + 63) // 64 * 64                          # ... i.e. it holds exactly ONE slab set of this layer
    _wgrad(be, P, Q, g_shared)                                             # same gradient again: flush first, then queue
    for Pk, Qk, gk in others:
        _wgrad(be, Pk, Qk, gk)                                             # arena full every time: flush, wrap to offset 0, queue
    be.wgrad_defer_end()
    got = []
    for k in range(len(be.lib.batches)):
        entries, total = _entries(be, k)
        assert [e.block0 for e in entries][0] == 0 and total == sum(e.nblocks for e in entries)
        got += [e.grad for e in entries]
    assert got == [g_shared.data_ptr(), g_shared.data_ptr()] + [g.data_ptr() for _, _, g in others]      # all five, in order, none dropped
    assert len(be.lib.batches) == 5
    assert all(s == arena.data_ptr() for s, _, _ in be.lib.slab_calls)     # every slab set started at the arena's base after its flush


def _desc_of(P, Q, grad):
    d = L.WgradDesc()
    d.N, d.Hp, d.Wp, d.CAp = P.shape
    _, d.Hq, d.Wq, d.CBp = Q.shape
    d.KH = d.KW = 3
    d.splitk = 2
    return d


# ---- round 5: the batched weight gradient (ops.HipBackend._wgrad_queued / _launch_queued) ---------------------------------------------------------
def _big_layer(n=1, h=4):
    P = torch.zeros(n, h, 128, 128, dtype=torch.bfloat16)
    Q = torch.zeros(n, h, 128, 128, dtype=torch.bfloat16)
    return P, Q, torch.zeros(128, 128, 3, 3)


def _wgrad_auto(be, P, Q, g):
    be.conv_wgrad(P, Q, g, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, True)


def test_repeated_shape_is_queued_and_computed_by_one_launch(be, monkeypatch):
    monkeypatch.setattr(ops, '_WGRAD_BATCH', True)
    big = [_big_layer() for _ in range(5)]
    small = _layer(16, 8)
    be.wgrad_defer_begin()
    for P, Q, g in big[:3]:
        _wgrad_auto(be, P, Q, g)
    _wgrad_auto(be, *small)                                  # an ordinary layer in between: its own split-K kernel at once, reduction pending
    for P, Q, g in big[3:]:
        _wgrad_auto(be, P, Q, g)
    st = ops.WS._state()
    assert len(st['defer_queue']) == 5 and len(st['defer_pending']) == 1 and not getattr(be.lib, 'multi', None)
    assert all(it[2] is P and it[3] is Q for it, (P, Q, _) in zip(st['defer_queue'], big))        # the operands are kept alive by the queue
    be.wgrad_defer_end()
    assert len(be.lib.multi) == 1 and len(be.lib.batches) == 1
    splitk, n, Ps, Gs, slab0 = be.lib.multi[0]
    assert n == 5 and Gs == [g.data_ptr() for _, _, g in big] and Ps == [P.data_ptr() for P, _, _ in big]
    from deepliif_amd.geometry import choose_wgrad_batch_splitk
    assert splitk == choose_wgrad_batch_splitk(12, 4)        # a function of the layer alone: not of how many layers share the launch
    entries, total = _entries(be, 0)
    assert [e.grad for e in entries] == [small[2].data_ptr()] + Gs and total == sum(e.nblocks for e in entries)
    assert [e.block0 for e in entries] == [sum(x.nblocks for x in entries[:i]) for i in range(len(entries))]
    spans = sorted([(e.slab, e.slab + 4 * e.splitk * e.CAp * e.J) for e in entries])
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))            # no two slab sets of the batch overlap
    assert not st['defer_queue'] and not st['defer_pending']


def test_batch_is_cut_at_the_launch_limit_and_at_a_second_use(be, monkeypatch):
    monkeypatch.setattr(ops, '_WGRAD_BATCH', True)
    layers = [_big_layer() for _ in range(L.WGRAD_MULTI_MAX + 3)]
    be.wgrad_defer_begin()
    for P, Q, g in layers:
        _wgrad_auto(be, P, Q, g)
    _wgrad_auto(be, *layers[0])                              # the first gradient again (a discriminator's second pass): everything queued runs first
    assert [m[1] for m in be.lib.multi] == [L.WGRAD_MULTI_MAX, 3] and len(ops.WS._state()['defer_queue']) == 1
    be.wgrad_defer_end()
    assert [m[1] for m in be.lib.multi] == [L.WGRAD_MULTI_MAX, 3, 1]
    got = []
    for k in range(len(be.lib.batches)):
        entries, _ = _entries(be, k)
        got += [e.grad for e in entries]
    assert got == [g.data_ptr() for _, _, g in layers] + [layers[0][2].data_ptr()]


def test_batching_can_be_switched_off(be, monkeypatch):
    monkeypatch.setattr(ops, '_WGRAD_BATCH', False)
    P, Q, g = _big_layer()
    be.wgrad_defer_begin()
    _wgrad_auto(be, P, Q, g)
    assert len(be.lib.slab_calls) == 1 and not ops.WS._state().get('defer_queue')
    be.wgrad_defer_end()
    assert len(be.lib.batches) == 1 and not getattr(be.lib, 'multi', None)
