"""Host logic of the DeepLIIFModel drop-in (deepliif_amd/models.py) on CPU with the emulated ops backend: the two-step
optimize_parameters() trajectory must follow the oracle (which is pinned to the reference by tests/test_oracle_golden.py) --
losses, generated images and updated weights -- for the translation-only and the full seg-generator graphs."""
import os
import types

import pytest
import numpy as np
import torch

import fake_backend
from deepliif_amd import models as M
from deepliif_amd import networks as N
from golden_util import seeded_uniform
from oracle import deepliif_oracle as O


@pytest.fixture(autouse=True)
def _fake():
    fake_backend.install()
    yield
    fake_backend.uninstall()


def make_opt(modalities_no, seg_gen, norm, net_gs='unet_64', nf=8, precision='fp32'):
    n = modalities_no + 1
    w = [0.25, 0.15, 0.25, 0.1, 0.25] if modalities_no == 4 else [1.0 / n] * n
    lw = [0.2] * 5 if modalities_no == 4 else [1.0 / n] * n
    return types.SimpleNamespace(
        model='DeepLIIF', name='t', checkpoints_dir='/tmp/dl_amd_test', gpu_ids=[0], is_train=True, phase='train', continue_train=False,
        modalities_no=modalities_no, seg_gen=seg_gen, modalities_names=[], input_nc=3, input_no=1, output_nc=3, ngf=nf, ndf=nf,
        net_g='resnet_9blocks', net_gs=net_gs, net_d='n_layers', n_layers_D=4, norm=norm, no_dropout=True, init_type='normal', init_gain=0.02,
        padding='zero', upsample='convtranspose', gan_mode='vanilla', gan_mode_s='lsgan', optimizer='adam', lr_g=2e-4, lr_d=2e-4, beta1=0.5,
        lr_policy='linear', n_epochs=100, n_epochs_decay=100, epoch_count=0, seg_weights=w, loss_G_weights=lw, loss_D_weights=lw,
        lambda_L1=100.0, verbose=False, epoch='latest', load_iter=0, precision=precision)


class CpuModel(M.DeepLIIFModel):
    """DeepLIIFModel with device placement redirected to the CPU (emulated backend only; the product raises without a GPU)."""

    def _device_from_opt(self, opt):
        return torch.device('cpu')

    def _net_gpu_ids(self):
        return []


@pytest.mark.parametrize('modalities_no,seg_gen,norm,padding', [(1, False, 'batch', 'zero'), (2, True, 'instance', 'zero'), (4, True, 'batch', 'zero'),
                                                                 (2, True, 'instance', 'reflect'), (1, False, 'batch', 'reflect')])
def test_two_steps_follow_oracle(modalities_no, seg_gen, norm, padding):
    """padding='reflect' (cli --padding reflect): the ResnetGenerator convs sit behind nn.ReflectionPad2d; training needs their
    data gradient (pad-0 plan over the padded extent + dl_reflect_fold) and the reflect-gather weight gradient"""
    torch.manual_seed(0)
    opt = make_opt(modalities_no, seg_gen, norm)
    opt.padding = padding
    model = CpuModel(opt)
    model.setup(opt)
    # oracle with identical weights
    cfg = O.OracleConfig(modalities_no=modalities_no, seg_gen=seg_gen, norm=norm, padding=padding, net_gs='unet_64', ngf=8, ndf=8)
    S = str(model.mod_id_seg)
    nets = {}
    for n in model.model_names:
        sd = {k: v.detach().clone() for k, v in getattr(model, 'net' + n).state_dict().items()}
        nets[n.replace(S, 'S', 1) if (len(n) > 2 and n[1] == S) else n] = sd
    om = O.OracleDeepLIIF(cfg, nets)
    size, batch = 64, 2
    nB = modalities_no + (1 if seg_gen else 0)
    A = seeded_uniform((batch, 3, size, size), 22)
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(nB)]
    for step in range(2):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        om.set_input({'A': A, 'B': B})
        om.optimize_parameters()
        got, exp = model.get_current_losses(), om.current_losses()
        tol = 5e-4 if step == 0 else 5e-3
        for k, v in got.items():
            ko = k.replace('_' + S, '_S') if k.endswith('_' + S) else k
            assert abs(v - exp[ko]) <= tol * max(1.0, abs(exp[ko])), (step, k, v, exp[ko])
        for i in range(modalities_no):
            a, b = getattr(model, f'fake_B_{i + 1}'), om.fake_B[i].detach()
            assert float((a - b).abs().max() / b.abs().max()) < (5e-4 if step == 0 else 3e-2)
        if seg_gen:
            a, b = getattr(model, f'fake_B_{S}'), om.fake_seg.detach()
            assert float((a - b).abs().max() / b.abs().max()) < (5e-4 if step == 0 else 3e-2)
        # weights after the step: see tests/test_oracle_golden.py for why 1e-3 of |w| (10% of the Adam update)
        for n in model.model_names:
            sd = getattr(model, 'net' + n).state_dict()
            so = nets[n.replace(S, 'S', 1) if (len(n) > 2 and n[1] == S) else n]
            a = torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()])
            b = torch.cat([v.detach().reshape(-1).float() for v in so.values() if v.is_floating_point()])
            assert float((a - b).norm() / b.norm()) < 1e-3, (step, n)


class CpuExtModel(M.DeepLIIFExtModel):
    def _device_from_opt(self, opt):
        return torch.device('cpu')

    def _net_gpu_ids(self):
        return []


def test_deepliif_ext_two_steps_follow_oracle():
    torch.manual_seed(0)
    opt = make_opt(2, True, 'batch')
    opt.model, opt.net_ds = 'DeepLIIFExt', 'n_layers'
    opt.loss_G_weights = opt.loss_D_weights = opt.seg_weights = [0.5, 0.5]
    model = CpuExtModel(opt)
    model.setup(opt)
    cfg = O.OracleConfig(modalities_no=2, seg_gen=True, norm='batch', padding='zero', net_gs='unet_64', ngf=8, ndf=8,
                         loss_G_weights=[0.5, 0.5], loss_D_weights=[0.5, 0.5])
    nets = {n: {k: v.detach().clone() for k, v in net.state_dict().items()} for n, net in model._nets()}
    om = O.OracleDeepLIIFExt(cfg, nets)
    A = seeded_uniform((1, 3, 64, 64), 22)
    B = [seeded_uniform((1, 3, 64, 64), 23 + i) for i in range(2)]
    BS = [seeded_uniform((1, 3, 64, 64), 43 + i) for i in range(2)]
    for step in range(2):
        model.set_input({'A': A, 'B': B, 'BS': BS, 'A_paths': ['x']})
        model.optimize_parameters()
        om.set_input({'A': A, 'B': B, 'BS': BS})
        om.optimize_parameters()
        got, exp = model.get_current_losses(), om.current_losses()
        tol = 5e-4 if step == 0 else 5e-3
        for k, v in got.items():
            if '_VGG_' in k:        # lambda_feat = 0 here: the term is not evaluated and is reported as NaN, never as a plausible 0.0
                assert v != v, (k, v)
                continue
            assert abs(v - exp[k]) <= tol * max(1.0, abs(exp[k])), (step, k, v, exp[k])
        for i in range(2):
            for a, b in ((model.fake_B[i], om.fake_B[i].detach()), (model.fake_BS[i], om.fake_BS[i].detach())):
                assert float((a - b).abs().max() / b.abs().max()) < (5e-4 if step == 0 else 3e-2)


class CpuSDGModel(M.SDGModel):
    def _device_from_opt(self, opt):
        return torch.device('cpu')

    def _net_gpu_ids(self):
        return []


def test_sdg_two_steps_follow_oracle_and_reference_loss_names():
    torch.manual_seed(0)
    opt = make_opt(2, False, 'instance')
    opt.model, opt.input_no = 'SDG', 2
    opt.loss_G_weights = opt.loss_D_weights = opt.seg_weights = [0.5, 0.5]
    model = CpuSDGModel(opt)
    model.setup(opt)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'step_sdg_m2_in2_instance.npz'))
    assert list(model.loss_names) == [str(n) for n in z['loss_names']], 'loss_names of the reference SDGModel, in its order'
    assert [n for n in model.model_names] == [str(n) for n in z['model_names']]
    cfg = O.OracleConfig(modalities_no=2, seg_gen=False, norm='instance', padding='zero', ngf=8, ndf=8,
                         loss_G_weights=[0.5, 0.5], loss_D_weights=[0.5, 0.5])
    nets = {n: {k: v.detach().clone() for k, v in net.state_dict().items()} for n, net in model._nets()}
    assert nets['G_1']['model.1.weight'].shape[1] == 6 and nets['D_1']['model.0.weight'].shape[1] == 9      # input_nc * input_no (+ output_nc)
    om = O.OracleSDG(cfg, nets)
    A = [seeded_uniform((1, 3, 64, 64), 22 + 100 * k) for k in range(2)]
    B = [seeded_uniform((1, 3, 64, 64), 23 + i) for i in range(2)]
    for step in range(2):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        om.set_input({'A': A, 'B': B})
        om.optimize_parameters()
        got, exp = model.get_current_losses(), om.current_losses()
        tol = 5e-4 if step == 0 else 5e-3
        for k, v in got.items():
            if '_VGG_' in k:        # lambda_feat = 0 here: the term is not evaluated and is reported as NaN, never as a plausible 0.0
                assert v != v, (k, v)
                continue
            assert abs(v - exp[k]) <= tol * max(1.0, abs(exp[k])), (step, k, v, exp[k])
        for i in range(2):
            a, b = model.fake_B[i], om.fake_B[i].detach()
            assert float((a - b).abs().max() / b.abs().max()) < (5e-4 if step == 0 else 3e-2)


def test_fused_adam_repacks_existing_images_in_one_batch():
    """After FusedAdam.step the GEMM images that already exist are rebuilt by ONE batched call and ensure_packed finds them
    current (no per-layer repack at the next forward); images that never existed are left to the lazy path."""
    import fake_backend
    from deepliif_amd import engine as E, ops, optim, networks
    fb = fake_backend.FakeBackend()
    old, old_flag = ops._impl, optim._PACK_BATCH
    ops._impl = fb
    optim._PACK_BATCH = True            # (the default; DL_PACK_BATCH=0 in the environment would turn the hook off)
    try:
        torch.manual_seed(0)
        net = networks.define_D(6, 8, 'n_layers', n_layers_D=2, norm='instance', init_type='normal', init_gain=0.02, gpu_ids=[])
        net.train()
        opt = optim.FusedAdam(net.parameters(), lr=1e-3, betas=(0.5, 0.999))
        prec = E.Precision.get('fp32')
        x = torch.randn(1, 6, 32, 32)

        def fwd_bwd():                              # the engine's own reverse mode (explicit tape), not torch.autograd
            opt.zero_grad()                         # keeps p.grad attached to the optimizer's flat gradient buffer
            tape = E.Tape()
            ctx = E.Ctx(prec, tape, training=True)
            xa = E.to_engine(x, prec)
            xa.needs_grad = True                    # so the data-gradient images are built as well
            ya = net.run(ctx, xa)
            ya.grad = torch.ones_like(ya.t)
            tape.backward()

        fwd_bwd()                                   # builds forward (+ data-gradient) images lazily
        n_images = fb.calls.get('pack', 0)
        assert n_images > 0
        opt.step()
        assert fb.calls.get('pack_batch', 0) == 1 and fb.calls.get('pack_batch_build', 0) == 1
        assert fb.calls['pack'] == 2 * n_images    # every existing image repacked exactly once, inside the batch
        fwd_bwd()
        assert fb.calls['pack'] == 2 * n_images, 'ensure_packed repacked an image the batch had already rebuilt'
        opt.step()
        assert fb.calls['pack_batch'] == 2 and fb.calls['pack_batch_build'] == 1, 'the job table must be reused'
        # weights changed behind the optimizer's back (load_state_dict): the lazy path must still notice
        net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
        before = fb.calls['pack']
        fwd_bwd()
        assert fb.calls['pack'] > before
    finally:
        ops._impl, optim._PACK_BATCH = old, old_flag


def test_non_adam_optimizer_runs_on_the_flat_parameter_set():
    """--optimizer sgd (the reference's tests/test_cli_train.py exercises it): torch.optim's own update rule on a FlatParams set, so the
    engine's gradient writes, zero_grad and the exchange keep working (p.grad used to be None after zero_grad)"""
    torch.manual_seed(0)
    opt = make_opt(1, False, 'batch')
    opt.optimizer = 'sgd'
    model = CpuModel(opt)
    model.setup(opt)
    assert type(model.optimizer_G).__name__ == 'FlatSGD' and model.optimizer_G.flat.attached()
    before = torch.cat([p.detach().reshape(-1).clone() for p in model.netG1.parameters()])
    A = seeded_uniform((2, 3, 64, 64), 1)
    for _ in range(2):
        model.set_input({'A': A, 'B': [seeded_uniform((2, 3, 64, 64), 2)], 'A_paths': ['x']})
        model.optimize_parameters()
    after = torch.cat([p.detach().reshape(-1) for p in model.netG1.parameters()])
    assert torch.isfinite(after).all() and not torch.equal(before, after)
    assert all(p.grad is not None and p.grad.data_ptr() != 0 for p in model.netG1.parameters())


def test_single_process_multi_gpu_training_is_refused():
    opt = make_opt(1, False, 'batch')
    opt.gpu_ids = [0, 1]
    with pytest.raises(NotImplementedError, match='one process per GPU'):
        M.DeepLIIFModel(opt)


def _vgg_file(tmp_path):
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vgg_cases.npz'))
    path = os.path.join(str(tmp_path), 'vgg19.pth')
    sd = O.random_vgg19_state_dict(torch.Generator().manual_seed(int(z['vgg_seed'])))
    sd['classifier.0.weight'] = torch.zeros(2, 2)          # a torchvision file carries classifier weights too: ignored
    torch.save(sd, path)
    return z, path


def test_vgg_loss_module_matches_reference_vector(tmp_path):
    """networks.VGGLoss on the engine (conv+ReLU kernels, 2x2 max pooling, weighted L1 terms) vs the reference's VGGLoss value and d/dx"""
    from deepliif_amd import engine as E
    z, path = _vgg_file(tmp_path)
    crit = N.VGGLoss(path, torch.device('cpu'), 'fp32')
    prec = E.Precision.get('fp32')
    for tag in ('s64', 's48x80'):
        shape = tuple(int(v) for v in z[f'{tag}/shape'])
        tape = E.Tape()
        ctx = E.Ctx(prec, tape, training=True)
        x = E.to_engine(seeded_uniform(shape, 71), prec)
        x.needs_grad = True
        y = E.to_engine(seeded_uniform(shape, 72), prec)
        out = torch.zeros(1)
        crit.run(ctx, x, y, 1.0, out)
        tape.backward()
        assert abs(float(out) - float(z[f'{tag}/loss'])) <= 1e-4 * float(z[f'{tag}/loss'])
        dx = E.from_engine(E.Act(x.grad, 3))
        assert float((dx - torch.from_numpy(z[f'{tag}/dx'])).abs().max() / np.abs(z[f'{tag}/dx']).max()) < 1e-4


def test_default_objective_with_vgg_follows_reference_trajectory(tmp_path):
    """lambda_feat = 100 (what the reference's Options sets for every training run) with the weights supplied as a file"""
    z, path = _vgg_file(tmp_path)
    torch.manual_seed(0)
    opt = make_opt(2, True, 'batch')
    opt.lambda_feat, opt.vgg_weights = 100, path
    opt.loss_G_weights = z['step/loss_G_weights'].tolist()
    model = CpuModel(opt)
    model.setup(opt)
    for n, seed in zip(z['step/model_names'], z['step/net_seeds']):
        n = str(n)
        arch, cin, pad = ('n_layers', 6, 'zero') if n.startswith('D') else (('resnet_9blocks', 3, 'zero') if n in ('G1', 'G2') else ('unet_64', 3, 'reflect'))
        getattr(model, 'net' + n).load_state_dict(O.random_state_dict(arch, cin, 3, 8, 'batch', pad, 4, generator=torch.Generator().manual_seed(int(seed))))
    A = seeded_uniform((2, 3, 64, 64), 22)
    B = [seeded_uniform((2, 3, 64, 64), 23 + i) for i in range(3)]
    for s in range(2):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        got = model.get_current_losses()
        tol = 5e-4 if s == 0 else 5e-3
        for k, exp in zip(z['step/loss_names'], z[f'step{s}/losses']):
            assert abs(got[str(k)] - exp) <= tol * max(1.0, abs(exp)), (s, k, got[str(k)], exp)
        vg = [float(getattr(model, f'loss_G_VGG_{i + 1}')) for i in range(2)]
        assert np.allclose(vg, z[f'step{s}/vgg'], rtol=tol)
        for i in range(2):
            e = (getattr(model, f'fake_B_{i + 1}')[:, :, ::2, ::2] - torch.from_numpy(z[f'step{s}/fake_B_{i + 1}'])).abs().max()
            assert float(e) < (1e-3 if s == 0 else 5e-2)


def test_lambda_feat_without_weights_is_an_error_not_a_silent_change_of_objective(monkeypatch):
    monkeypatch.delenv('DEEPLIIF_VGG19_WEIGHTS', raising=False)
    monkeypatch.delenv('DEEPLIIF_AMD_ALLOW_NO_VGG', raising=False)
    opt = make_opt(1, False, 'batch')
    opt.lambda_feat = 100
    with pytest.raises(RuntimeError, match='VGG19'):
        CpuModel(opt)
    opt.allow_no_vgg = True
    assert CpuModel(opt).criterionVGG is None


def test_paired_discriminator_batch_equals_the_two_calls(monkeypatch):
    """DeepLIIFModel.backward_D with DL_D_PAIR_BATCH (InstanceNorm discriminators see cat(fake pairs, real pairs) as one batch of 2N) against the reference's
    two calls per discriminator: the same four loss kinds and the same discriminator gradients (the emulation computes in fp32: differences are summation order)"""
    res = {}
    for paired in (False, True):
        monkeypatch.setattr(M, '_D_PAIR_BATCH', paired)
        torch.manual_seed(0)
        opt = make_opt(2, True, 'instance')
        model = CpuModel(opt)
        model.setup(opt)
        A = seeded_uniform((2, 3, 64, 64), 22)
        B = [seeded_uniform((2, 3, 64, 64), 23 + i) for i in range(3)]
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.forward()
        for o in model.optimizers:
            o.zero_grad()
        model.backward_D()
        losses = {k: float(v) for k, v in model.get_current_losses().items() if k.startswith('D_')}
        grads = torch.cat([p.grad.reshape(-1) for n in model.model_names if n.startswith('D') for p in getattr(model, 'net' + n).parameters()])
        res[paired] = (losses, grads.clone())
    assert res[False][0].keys() == res[True][0].keys() and len(res[True][0]) == 6
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) <= 1e-6 * max(1.0, abs(v)), (k, v, res[True][0][k])
    g0, g1 = res[False][1], res[True][1]
    assert float((g0 - g1).norm() / g0.norm()) < 1e-5 and float(g0.norm()) > 0
    # BatchNorm discriminators keep the two calls (their statistics are per call): the switch must not change anything there
    monkeypatch.setattr(M, '_D_PAIR_BATCH', True)
    torch.manual_seed(0)
    opt = make_opt(1, False, 'batch')
    model = CpuModel(opt)
    model.setup(opt)
    calls = []
    d = model.netD1
    orig = d.run
    d.run = lambda ctx, x: (calls.append(x.t.shape[0]), orig(ctx, x))[1]
    model.set_input({'A': seeded_uniform((2, 3, 64, 64), 22), 'B': [seeded_uniform((2, 3, 64, 64), 23)], 'A_paths': ['x']})
    model.forward()
    model.backward_D()
    assert calls == [2, 2]
