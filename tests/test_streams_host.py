"""Host logic of the branch streams (models.BaseModel._branch_streams, engine.Tape(streams=True), ops.Workspace) without a GPU: torch.cuda's stream calls are
replaced by a recorder, so what is under test is the BOOK-KEEPING -- every tape node runs its backward under the stream it was recorded on, in reverse order,
and the home stream is restored also when a node raises; scratch state is per REGISTERED branch stream only (torch's default stream and a graph-capture
stream keep the thread's own state); the deferred slab reduction counts its nesting per thread, not per stream.  The GPU side (bit-identity of N streams
with one) is tests/test_gpu_streams.py."""
import pytest
import torch

from deepliif_amd import engine as E
from deepliif_amd import ops


class FakeStream:
    def __init__(self, sid):
        self.cuda_stream = sid

    def __eq__(self, other):
        return isinstance(other, FakeStream) and other.cuda_stream == self.cuda_stream

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self.cuda_stream)

    def __repr__(self):
        return f'S{self.cuda_stream}'


@pytest.fixture
def fake_cuda(monkeypatch):
    state = {'cur': FakeStream(0), 'log': []}
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda device=None: state['cur'])
    monkeypatch.setattr(ops, '_FAST_STREAM', False)            # (ops asks torch's C layer for the raw stream handle; here the recorder above must answer)

    def set_stream(s):
        state['log'].append(s)
        state['cur'] = s
    monkeypatch.setattr(torch.cuda, 'set_stream', set_stream)
    return state


class _NoDefer:
    """a backend without the deferred reduction: Tape.backward() must not need it"""


def test_tape_runs_every_node_on_the_stream_it_was_recorded_on(fake_cuda, monkeypatch):
    monkeypatch.setattr(ops, '_impl', _NoDefer())
    tape = E.Tape(streams=True)
    ran = []
    plan = [0, 1, 1, 2, 0, 2, 2, 1]                   # stream of node k
    for k, sid in enumerate(plan):
        fake_cuda['cur'] = FakeStream(sid)
        tape.record(lambda k=k: ran.append((k, fake_cuda['cur'].cuda_stream)))
    fake_cuda['cur'] = FakeStream(0)
    fake_cuda['log'].clear()
    tape.backward()
    assert ran == [(k, plan[k]) for k in reversed(range(len(plan)))]
    assert fake_cuda['cur'] == FakeStream(0)                          # home again
    # one switch per CHANGE of stream (reverse order: 1, 2, 2, 0, 2, 1, 1, 0), then home
    assert [s.cuda_stream for s in fake_cuda['log']] == [1, 2, 0, 2, 1, 0, 0]
    assert tape.nodes == [] and tape.node_streams == []               # the tape is reusable


def test_tape_restores_the_home_stream_when_a_node_raises(fake_cuda, monkeypatch):
    monkeypatch.setattr(ops, '_impl', _NoDefer())
    tape = E.Tape(streams=True)
    fake_cuda['cur'] = FakeStream(3)
    tape.record(lambda: (_ for _ in ()).throw(RuntimeError('boom')))
    fake_cuda['cur'] = FakeStream(0)
    with pytest.raises(RuntimeError, match='boom'):
        tape.backward()
    assert fake_cuda['cur'] == FakeStream(0)


def test_a_plain_tape_never_touches_the_streams(fake_cuda, monkeypatch):
    monkeypatch.setattr(ops, '_impl', _NoDefer())
    tape = E.Tape()
    ran = []
    tape.record(lambda: ran.append(1))
    tape.record(lambda: ran.append(2))
    tape.backward()
    assert ran == [2, 1] and fake_cuda['log'] == [] and tape.node_streams is None


def test_scratch_state_is_per_registered_branch_stream_only(fake_cuda):
    ws = ops.Workspace()
    fake_cuda['cur'] = FakeStream(0)
    base = ws._state()
    assert base is ws._thread_state()                                 # nothing registered: the thread's own state, no stream lookup needed
    ws.branch_streams_on([FakeStream(11), FakeStream(12)])
    assert ws._state() is base                                        # torch's default stream keeps it
    fake_cuda['cur'] = FakeStream(99)                                 # e.g. the side stream torch.cuda.graph captures on
    assert ws._state() is base
    fake_cuda['cur'] = FakeStream(11)
    a = ws._state()
    fake_cuda['cur'] = FakeStream(12)
    b = ws._state()
    assert a is not base and b is not base and a is not b
    assert a['stream_obj'] == FakeStream(11) and a['bufs'] == {} and a['norm_ws_token'] == 0
    fake_cuda['cur'] = FakeStream(11)
    assert ws._state() is a
    ws.bump_norm_token()
    assert a['norm_ws_token'] == 1 and b['norm_ws_token'] == 0 and base['norm_ws_token'] == 0
    assert sorted(s['stream_obj'].cuda_stream for s in ws.stream_states()) == [11, 12]


def test_deferral_nesting_is_counted_per_thread(fake_cuda, monkeypatch):
    """wgrad_defer_begin() on the main stream must switch the deferral on for the branch streams too, and the end of the OUTER pass flushes every stream"""
    monkeypatch.setattr(ops, 'WS', ops.Workspace())
    be = ops.HipBackend.__new__(ops.HipBackend)                       # no library needed for the book-keeping
    flushed = []
    monkeypatch.setattr(be, 'wgrad_flush', lambda: flushed.append(fake_cuda['cur'].cuda_stream), raising=False)

    class _Ctx:
        def __init__(self, s):
            self.s = s

        def __enter__(self):
            self.prev, fake_cuda['cur'] = fake_cuda['cur'], self.s

        def __exit__(self, *a):
            fake_cuda['cur'] = self.prev
    monkeypatch.setattr(torch.cuda, 'stream', lambda s: _Ctx(s))
    ops.WS.branch_streams_on([FakeStream(21), FakeStream(22)])
    be.wgrad_defer_begin()
    be.wgrad_defer_begin()                                            # a nested pass (a tape inside a tape node)
    fake_cuda['cur'] = FakeStream(21)
    assert ops.WS._thread_state()['defer_depth'] == 2                 # seen from a branch stream
    ops.WS._state()['defer_pending'] = ['x']                          # something is pending on stream 21 only
    fake_cuda['cur'] = FakeStream(0)
    be.wgrad_defer_end()
    assert flushed == []                                              # the inner end does not flush
    be.wgrad_defer_end()
    assert flushed == [0, 21]                                         # the main state, then every stream with pending slabs -- under THAT stream
    assert ops.WS._thread_state()['defer_depth'] == 0


class _NullStream:
    """stands in for torch.cuda.Stream on a GPU-less machine: waits are no-ops (everything runs in program order on the CPU)"""
    cuda_stream = -1

    def wait_stream(self, other):
        self.waited = getattr(self, 'waited', 0) + 1


@pytest.mark.parametrize('seg_only', [False, True])
@pytest.mark.parametrize('names', [['IHC', 'Hema', 'DAPI', 'Lap2', 'Marker'], []])
def test_inference_dag_on_streams_keeps_keys_order_and_values(seg_only, names, monkeypatch):
    """inference._run_deepliif_dag_on_streams restructures run_dask's DAG (chain i = G_i -> GS_i) -- on the emulated backend, with stand-in streams, its result
    dict must equal the one-stream DAG's: same keys, same ORDER (callers index by position), same values; seg_only drops the same entries."""
    import contextlib
    import types

    import fake_backend
    from deepliif_amd import inference as I
    fake_backend.install()
    try:
        opt = types.SimpleNamespace(model='DeepLIIF', modalities_no=4, seg_gen=True, mod_id_seg='S', input_id=0, input_nc=3, output_nc=3, ngf=8, norm='batch', padding='zero',
                                    net_g='resnet_9blocks', net_gs='unet_32', input_no=1, scale_size=32, modalities_names=names, gpu_ids=[])
        torch.manual_seed(3)
        nets = I.build_generators(opt, torch.device('cpu'), 'fp32')
        x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(4)) * 2 - 1
        sw = [0.25, 0.15, 0.25, 0.1, 0.25]
        ref = I.run_generators(x, nets, opt, seg_only=seg_only, seg_weights=sw)
        streams = [_NullStream() for _ in range(3)]
        monkeypatch.setattr(I, '_infer_streams', lambda device: streams)
        monkeypatch.setattr(torch.cuda, 'current_stream', lambda device=None: _NullStream())
        monkeypatch.setattr(torch.cuda, 'stream', lambda s: contextlib.nullcontext())
        got = I.run_generators(x, nets, opt, seg_only=seg_only, seg_weights=sw)
        assert all(s.waited == 1 for s in streams)                       # forked once; the join is the main stream's wait
        assert list(got.keys()) == list(ref.keys())
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
        # zero-weight seg branches are skipped on both routes
        sw0 = [0.5, 0.0, 0.5, 0.0, 0.0]
        monkeypatch.setattr(I, '_infer_streams', lambda device: None)
        ref0 = I.run_generators(x, nets, opt, seg_only=True, seg_weights=sw0)
        monkeypatch.setattr(I, '_infer_streams', lambda device: streams)
        got0 = I.run_generators(x, nets, opt, seg_only=True, seg_weights=sw0)
        assert list(got0.keys()) == list(ref0.keys()) and all(torch.equal(got0[k], ref0[k]) for k in ref0)
    finally:
        fake_backend.uninstall()


class _LogStream:
    """a stand-in stream that logs who waits for whom; kernels 'run' in program order on the CPU, so results cannot depend on it"""

    def __init__(self, sid, log):
        self.cuda_stream, self.log = sid, log

    def wait_stream(self, other):
        self.log.append((self.cuda_stream, 'waits', other.cuda_stream))

    def __eq__(self, other):
        return isinstance(other, _LogStream) and other.cuda_stream == self.cuda_stream

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self.cuda_stream)


@pytest.mark.parametrize('seg_gen,modalities_no', [(False, 2), (True, 2)])
def test_training_step_on_stand_in_streams_matches_the_plain_step(seg_gen, modalities_no, monkeypatch):
    """A whole optimize_parameters() on the emulated backend with the branch-stream book-keeping switched on over stand-in streams: (a) losses, images and
    parameters equal the plain step (the restructured loops compute the same thing in the same order per branch); (b) every phase forks before its first
    branch op and joins before the optimizer step; (c) with segmentation generators the join in front of the weighted seg sum is ON THE TAPE: in backward
    every branch waits for the main stream after the weighted sum's backward and before the seg generators' backward."""
    import contextlib
    import types

    import fake_backend
    from deepliif_amd import models as M
    fake_backend.install()
    try:
        n = modalities_no + 1
        opt = types.SimpleNamespace(
            model='DeepLIIF', name='t', checkpoints_dir='/tmp/dl_amd_test', gpu_ids=[0], is_train=True, phase='train', continue_train=False, modalities_no=modalities_no,
            seg_gen=seg_gen, modalities_names=[], input_nc=3, input_no=1, output_nc=3, ngf=8, ndf=8, net_g='resnet_9blocks', net_gs='unet_32', net_d='n_layers', n_layers_D=2,
            norm='batch', no_dropout=True, init_type='normal', init_gain=0.02, padding='zero', upsample='convtranspose', gan_mode='vanilla', gan_mode_s='lsgan', optimizer='adam',
            lr_g=2e-4, lr_d=2e-4, beta1=0.5, lr_policy='linear', n_epochs=100, n_epochs_decay=100, epoch_count=0, seg_weights=[1.0 / n] * n, loss_G_weights=[1.0 / n] * n,
            loss_D_weights=[1.0 / n] * n, lambda_L1=100.0, verbose=False, epoch='latest', load_iter=0, precision='fp32')

        class CpuModel(M.DeepLIIFModel):
            def _device_from_opt(self, o):
                return torch.device('cpu')

            def _net_gpu_ids(self):
                return []

        g = torch.Generator().manual_seed(8)
        batch = {'A': torch.rand(2, 3, 32, 32, generator=g) * 2 - 1, 'B': [torch.rand(2, 3, 32, 32, generator=g) * 2 - 1 for _ in range(n if seg_gen else modalities_no)], 'A_paths': ['x']}

        def run(with_streams):
            torch.manual_seed(0)
            model = CpuModel(opt)
            model.setup(opt)
            log = []
            if with_streams:
                main = _LogStream(0, log)
                model._streams = [_LogStream(10 + k, log) for k in range(3)]
                model.branch_parallel = True
                state = {'cur': main}
                monkeypatch.setattr(torch.cuda, 'current_stream', lambda device=None: state['cur'])
                monkeypatch.setattr(torch.cuda, 'set_stream', lambda s: state.__setitem__('cur', s))

                @contextlib.contextmanager
                def on(s):
                    prev, state['cur'] = state['cur'], s
                    try:
                        yield
                    finally:
                        state['cur'] = prev
                monkeypatch.setattr(torch.cuda, 'stream', on)
                orig_ws = M.E.weighted_sum

                def ws(ctx, parts, weights):
                    log.append(('weighted_sum', 'on', state['cur'].cuda_stream))
                    out = orig_ws(ctx, parts, weights)
                    if ctx.tape is not None:
                        ctx.tape.record(lambda: log.append(('weighted_sum backward', 'on', state['cur'].cuda_stream)))
                    return out
                monkeypatch.setattr(M.E, 'weighted_sum', ws)
            else:
                model._streams = None
            for _ in range(2):
                model.set_input(batch)
                model.optimize_parameters()
            return dict(model.get_current_losses()), torch.cat([o.flat.data.clone() for o in model.optimizers]), model.fake_B_1.clone(), log

        ref = run(False)
        got = run(True)
        assert got[0] == ref[0] and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
        log = got[3]
        forks = [k for k, e in enumerate(log) if e == (10, 'waits', 0)]
        joins = [k for k, e in enumerate(log) if e == (0, 'waits', 10)]
        assert len(forks) >= 6 and len(joins) >= 4                     # forward + backward_D + backward_G, two steps
        if seg_gen:
            join3 = [(0, 'waits', 10), (0, 'waits', 11), (0, 'waits', 12)]
            fork3 = [(10, 'waits', 0), (11, 'waits', 0), (12, 'waits', 0)]
            # forward (once per step): the main stream joins the branches right before the weighted seg sum, which runs on the main stream
            fwd = [k for k, e in enumerate(log) if e == ('weighted_sum', 'on', 0) and log[k - 3:k] == join3]
            assert len(fwd) == 2, fwd
            # backward of the generator tape (once per step): the weighted sum's backward on the main stream, THEN every branch waits for the main stream
            bwd = [k for k, e in enumerate(log) if e == ('weighted_sum backward', 'on', 0) and log[k + 1:k + 4] == fork3]
            assert len(bwd) == 2, bwd
            assert fwd[0] < bwd[0] < fwd[1] < bwd[1]
    finally:
        fake_backend.uninstall()


def test_branch_streams_are_recognised_from_another_thread(fake_cuda):
    """a model registers its branch streams once (first forward); a trainer that later runs on a worker thread must get per-stream scratch too"""
    import threading
    ws = ops.Workspace()
    ws.branch_streams_on([FakeStream(31), FakeStream(32)])
    seen = {}

    def worker():
        fake_cuda['cur'] = FakeStream(31)
        a = ws._state()
        fake_cuda['cur'] = FakeStream(32)
        b = ws._state()
        seen['ok'] = a is not b and a is not ws._thread_state() and a['stream_obj'] == FakeStream(31)
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert seen['ok']
    ws.forget_branch_streams()
    fake_cuda['cur'] = FakeStream(31)
    assert ws._state() is ws._thread_state()
