"""Pin the CPU oracle (oracle/deepliif_oracle.py) against fixtures produced by the REFERENCE itself
(tests/golden/make_golden.py, run in the build container).  fp32 vs fp32 on the same torch build: tolerance 1e-4
relative to the tensor's scale (both sides run the same ATen convs; differences come from norm/loss restatement)."""
import os

import numpy as np
import pytest
import torch

from golden_util import digest, digest_close, seeded_uniform
from oracle import deepliif_oracle as O

G = os.path.join(os.path.dirname(__file__), 'golden')
RTOL = 1e-4


def rel_err(a, b, floor=1e-30):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(floor))


def _cases(npz):
    return sorted({k.split('/')[0] for k in npz.files if '/meta' in k})


NETS = np.load(os.path.join(G, 'nets_small.npz'))


def _run(arch, sd, x, norm, pad, update_running=False):
    if arch == 'n_layers':
        return O.nlayer_discriminator(sd, x, norm, 4, update_running)
    return O.run_generator(arch, sd, x, norm, pad, update_running)


@pytest.mark.parametrize('tag', _cases(NETS))
def test_network_forward_backward(tag):
    arch, cin, nf, norm, pad, wseed, xseed, xshape = NETS[f'{tag}/meta']
    sd = O.random_state_dict(arch, int(cin), 3, int(nf), norm, pad, 4, generator=torch.Generator().manual_seed(int(wseed)))
    flat = torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()])
    ok, msg = digest_close(flat, NETS[f'{tag}/w_digest'], 1e-9)
    assert ok, 'seeded weights differ from the ones the fixture was made with: ' + msg
    x = seeded_uniform(eval(xshape), int(xseed)).requires_grad_(True)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running' not in k}
    y = _run(arch, sd, x, norm, pad, update_running=(norm == 'batch'))
    assert rel_err(y.detach(), NETS[f'{tag}/y']) < RTOL
    r = torch.randn(y.shape, generator=torch.Generator().manual_seed(99))
    grads = torch.autograd.grad((y * r).sum(), [x] + list(params.values()))
    assert rel_err(grads[0], NETS[f'{tag}/dx']) < RTOL
    # a conv bias in front of an affine-less InstanceNorm has an exactly-zero gradient (only fp noise is left), so
    # errors are measured against the largest parameter gradient of the network, not per tensor
    gscale = max(float(g.abs().max()) for g in grads[1:])
    for (k, _), g in zip(params.items(), grads[1:]):
        exp = NETS[f'{tag}/dw/{k}']
        if exp.shape == tuple(g.shape):
            assert rel_err(g, exp, floor=0.05 * gscale) < 5 * RTOL, k
        else:
            ok, msg = digest_close(g, exp, 5 * RTOL)
            assert ok, f'{k}: {msg}'
    # BatchNorm running statistics after one training-mode forward
    for k in NETS.files:
        if k.startswith(f'{tag}/sd_after/'):
            name = k[len(f'{tag}/sd_after/'):]
            assert rel_err(sd[name].detach(), NETS[k]) < RTOL, name
    # eval-mode forward == batch statistics (disable_batchnorm_tracking_stats)
    with torch.no_grad():
        y2 = _run(arch, sd, x.detach(), norm, pad)
    assert rel_err(y2, NETS[f'{tag}/y_eval']) < RTOL


def test_unet512_full_depth():
    z = np.load(os.path.join(G, 'unet512_ngf8.npz'))
    arch, cin, nf, norm, pad, wseed, xseed, xshape = z['meta']
    sd = O.random_state_dict(arch, int(cin), 3, int(nf), norm, pad, generator=torch.Generator().manual_seed(int(wseed)))
    x = seeded_uniform(eval(xshape), int(xseed))
    with torch.no_grad():
        y = O.unet_generator(sd, x, norm, 9)
    assert rel_err(y[:, :, ::8, ::8], z['y_strided']) < RTOL
    ok, msg = digest_close(y, z['y_digest'], RTOL)
    assert ok, msg


def build_oracle_model(z, nf):
    mod_no, seg_gen, norm, padding, net_gs, size, nf_, batch, steps = z['meta']
    cfg = O.OracleConfig(modalities_no=int(mod_no), seg_gen=(seg_gen == 'True'), net_g='resnet_9blocks', net_gs=net_gs,
                         norm=norm, padding=padding, ngf=int(nf_), ndf=int(nf_))
    nets = {}
    g_names, gs_names, d_names, ds_names = cfg.names(str(z['mod_id_seg']))
    for name, seed in zip(z['model_names'], z['net_seeds']):
        name = str(name)
        if name.startswith('D'):
            arch, pad, cin = 'n_layers', 'zero', 6
        elif name in g_names:
            arch, pad, cin = cfg.net_g, padding, 3
        else:
            arch, pad, cin = net_gs, 'reflect', 3
        nets[name] = O.random_state_dict(arch, cin, 3, int(nf_), norm, pad, 4, generator=torch.Generator().manual_seed(int(seed)))
    return cfg, nets, int(size), int(batch), int(steps)


@pytest.mark.parametrize('tag', ['m1_noseg_batch', 'm5_noseg_instance', 'm4_seg_batch', 'm2_seg_instance_reflect'])
def test_two_step_trajectory(tag):
    z = np.load(os.path.join(G, f'step_{tag}.npz'))
    cfg, nets, size, batch, steps = build_oracle_model(z, 8)
    mid = str(z['mod_id_seg'])
    # the oracle names its nets with mod_id_seg 'S'; map the fixture's names
    ren = {n: n.replace(mid, 'S', 1) if (len(n) > 2 and n[1] == mid[0] and mid != 'None') else n for n in nets}
    nets = {ren[n]: sd for n, sd in nets.items()}
    model = O.OracleDeepLIIF(cfg, nets)
    nB = cfg.modalities_no + (1 if cfg.seg_gen else 0)
    A = seeded_uniform((batch, 3, size, size), 22)
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(nB)]
    for s in range(steps):
        model.set_input({'A': A, 'B': B})
        model.optimize_parameters()
        got = model.current_losses()
        # step 0 is a pure function of the inputs; later steps inherit the noise-determined part of the Adam update
        # (see the weight check below), which perturbs losses / outputs at the 1e-3 level
        tol = 2e-4 if s == 0 else 3e-3
        for name, exp in zip(z['loss_names'], z[f'step{s}/losses']):
            name = str(name).replace('_' + mid, '_S') if mid != 'None' else str(name)
            assert abs(got[name] - exp) <= tol * max(1.0, abs(exp)), (s, name, got[name], exp)
        for i in range(cfg.modalities_no):
            assert rel_err(model.fake_B[i].detach()[:, :, ::2, ::2], z[f'step{s}/fake_B_{i + 1}']) < (tol if s == 0 else 2e-2)
        if cfg.seg_gen:
            assert rel_err(model.fake_seg.detach()[:, :, ::2, ::2], z[f'step{s}/fake_B_S']) < (tol if s == 0 else 2e-2)
        for n_fix in z['model_names']:
            sd = nets[ren[str(n_fix)]]
            flat = torch.cat([v.detach().reshape(-1).float() for v in sd.values() if v.is_floating_point()])
            # Adam's first steps move every weight by ~lr*sign(g): |dw| ~ 1% of |w| here, and elements whose gradient is
            # at fp32-noise level take a noise-determined +-lr step (e.g. every conv bias in front of an InstanceNorm: its
            # true gradient is exactly 0).  1e-3 of |w| = 10% of the update norm; a wrong lr / bias correction / sign is >= 100%.
            ok, msg = digest_close(flat, z[f'step{s}/w_digest/{n_fix}'], 1e-3)
            assert ok, f'step {s} weights of {n_fix}: {msg}'


def test_inference_dag():
    """run_dask equivalent: G_i(tile), GS_0(tile), GS_i(G_i(tile)), seg = sum w_i seg_i  (models/__init__.py:293-338)."""
    z = np.load(os.path.join(G, 'inference_small.npz'))
    nets = {}
    for name, seed in zip(z['net_names'], z['net_seeds']):
        name = str(name)
        seg = len(name) > 2
        arch = 'unet_64' if seg else 'resnet_9blocks'
        nets[name] = O.random_state_dict(arch, 3, 3, 8, 'batch', 'reflect' if seg else 'zero', generator=torch.Generator().manual_seed(int(seed)))
    tiles = seeded_uniform((3, 3, 64, 64), 32)
    w = z['seg_weights']
    with torch.no_grad():
        for t in range(3):
            x = tiles[t:t + 1]
            gens = {f'G{i}': O.resnet_generator(nets[f'G{i}'], x, 'batch', 'zero') for i in range(1, 5)}
            segs = {'GS0': O.unet_generator(nets['GS0'], x, 'batch', 6)}
            for i in range(1, 5):
                segs[f'GS{i}'] = O.unet_generator(nets[f'GS{i}'], gens[f'G{i}'], 'batch', 6)
            seg = sum(segs[f'GS{i}'] * float(w[i]) for i in range(5))
            for k, v in {**gens, **segs, 'GS': seg}.items():
                assert rel_err(v, z[f'tile{t}/{k}']) < RTOL, (t, k)


def test_deepliif_ext_two_step_trajectory():
    """DeepLIIFExtModel (DeepLIIFExt_model.py): list-valued nets, 9-channel seg generators, 12-channel seg discriminators."""
    z = np.load(os.path.join(G, 'step_ext_m2_batch.npz'))
    M, norm, size, nf, batch, steps = z['meta']
    M, size, nf, batch = int(M), int(size), int(nf), int(batch)
    cfg = O.OracleConfig(modalities_no=M, seg_gen=True, norm=norm, padding='zero', net_gs='unet_64', ngf=nf, ndf=nf,
                         loss_G_weights=[1.0 / M] * M, loss_D_weights=[1.0 / M] * M)
    spec = {'G': ('resnet_9blocks', 3, 'zero'), 'GS': ('unet_64', 9, 'reflect'), 'D': ('n_layers', 6, 'zero'), 'DS': ('n_layers', 12, 'zero')}
    nets = {}
    for name, seed in zip(z['model_names'], z['net_seeds']):
        arch, cin, pad = spec[str(name).split('_')[0]]
        nets[str(name)] = O.random_state_dict(arch, cin, 3, nf, norm, pad, 4, generator=torch.Generator().manual_seed(int(seed)))
    om = O.OracleDeepLIIFExt(cfg, nets)
    A = seeded_uniform((batch, 3, size, size), 22)
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(M)]
    BS = [seeded_uniform((batch, 3, size, size), 43 + i) for i in range(M)]
    for s in range(int(steps)):
        om.set_input({'A': A, 'B': B, 'BS': BS})
        om.optimize_parameters()
        got = om.current_losses()
        tol = 2e-4 if s == 0 else 3e-3
        for name, exp in zip(z['loss_names'], z[f'step{s}/losses']):
            assert abs(got[str(name)] - exp) <= tol * max(1.0, abs(exp)), (s, name, got[str(name)], exp)
        for i in range(M):
            assert rel_err(om.fake_B[i].detach()[:, :, ::2, ::2], z[f'step{s}/fake_B_{i + 1}']) < (tol if s == 0 else 2e-2)
            assert rel_err(om.fake_BS[i].detach()[:, :, ::2, ::2], z[f'step{s}/fake_BS_{i + 1}']) < (tol if s == 0 else 2e-2)
        for n in z['model_names']:
            flat = torch.cat([v.detach().reshape(-1).float() for v in nets[str(n)].values() if v.is_floating_point()])
            ok, msg = digest_close(flat, z[f'step{s}/w_digest/{n}'], 1e-3)
            assert ok, f'step {s} weights of {n}: {msg}'


def test_sdg_two_step_trajectory():
    """SDGModel (SDG_model.py): two input modalities concatenated on the channel axis -> 6-channel generators, 9-channel
    discriminators; the reference's loss_names (with the zeroed G_VGG_i) and 2-step trajectory."""
    z = np.load(os.path.join(G, 'step_sdg_m2_in2_instance.npz'))
    M, input_no, norm, size, nf, batch, steps = z['meta']
    M, input_no, size, nf, batch = int(M), int(input_no), int(size), int(nf), int(batch)
    cfg = O.OracleConfig(modalities_no=M, seg_gen=False, norm=norm, padding='zero', ngf=nf, ndf=nf,
                         loss_G_weights=[1.0 / M] * M, loss_D_weights=[1.0 / M] * M)
    spec = {'G': ('resnet_9blocks', 3 * input_no, 'zero'), 'D': ('n_layers', 3 * input_no + 3, 'zero')}
    nets = {}
    for name, seed in zip(z['model_names'], z['net_seeds']):
        arch, cin, pad = spec[str(name).split('_')[0]]
        nets[str(name)] = O.random_state_dict(arch, cin, 3, nf, norm, pad, 4, generator=torch.Generator().manual_seed(int(seed)))
    om = O.OracleSDG(cfg, nets)
    A = [seeded_uniform((batch, 3, size, size), 22 + 100 * k) for k in range(input_no)]
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(M)]
    for s in range(int(steps)):
        om.set_input({'A': A, 'B': B})
        om.optimize_parameters()
        got = om.current_losses()
        assert set(got) == {str(n) for n in z['loss_names']}, 'loss names of the seam'
        tol = 2e-4 if s == 0 else 3e-3
        for name, exp in zip(z['loss_names'], z[f'step{s}/losses']):
            assert abs(got[str(name)] - exp) <= tol * max(1.0, abs(exp)), (s, name, got[str(name)], exp)
        for i in range(M):
            assert rel_err(om.fake_B[i].detach()[:, :, ::2, ::2], z[f'step{s}/fake_B_{i + 1}']) < (tol if s == 0 else 2e-2)
        for n in z['model_names']:
            flat = torch.cat([v.detach().reshape(-1).float() for v in nets[str(n)].values() if v.is_floating_point()])
            ok, msg = digest_close(flat, z[f'step{s}/w_digest/{n}'], 1e-3)
            assert ok, f'step {s} weights of {n}: {msg}'


# ---------------------------------------------------------------------------------------------------------------------------
# VGG19 perceptual term (tests/golden/make_golden_vgg.py: the reference's VGGLoss on a seeded stand-in for the downloaded weights)
# ---------------------------------------------------------------------------------------------------------------------------
VGG = np.load(os.path.join(G, 'vgg_cases.npz'))


def _vgg_sd():
    return O.random_vgg19_state_dict(torch.Generator().manual_seed(int(VGG['vgg_seed'])))


@pytest.mark.parametrize('tag', ['s64', 's48x80'])
def test_vgg_loss_value_gradient_and_features(tag):
    sd = _vgg_sd()
    shape = tuple(int(v) for v in VGG[f'{tag}/shape'])
    x = seeded_uniform(shape, 71).requires_grad_(True)
    y = seeded_uniform(shape, 72)
    loss = O.vgg_loss(sd, x, y)
    loss.backward()
    assert abs(float(loss) - float(VGG[f'{tag}/loss'])) <= RTOL * float(VGG[f'{tag}/loss'])
    assert rel_err(x.grad, VGG[f'{tag}/dx']) < RTOL
    for i, f in enumerate(O.vgg19_features(sd, x.detach())):
        assert list(f.shape) == VGG[f'{tag}/feat{i}_shape'].tolist()
        ok, msg = digest_close(f, VGG[f'{tag}/feat{i}_digest'], RTOL)
        assert ok, (i, msg)


def test_step_with_vgg_term_follows_reference():
    """the reference's DEFAULT training objective (lambda_feat = 100): GAN + SmoothL1 + VGG for the modalities, GAN + SmoothL1 for seg"""
    names = [str(n) for n in VGG['step/model_names']]
    cfg = O.OracleConfig(modalities_no=2, seg_gen=True, norm='batch', padding='zero', net_gs='unet_64', ngf=8, ndf=8, seg_weights=[1 / 3] * 3,
                         loss_G_weights=VGG['step/loss_G_weights'].tolist(), loss_D_weights=[1 / 3] * 3, lambda_feat=100.0)
    nets = {}
    for n, seed in zip(names, VGG['step/net_seeds']):
        arch, cin, pad = ('n_layers', 6, 'zero') if n.startswith('D') else (('resnet_9blocks', 3, 'zero') if n in ('G1', 'G2') else ('unet_64', 3, 'reflect'))
        nets[n] = O.random_state_dict(arch, cin, 3, 8, 'batch', pad, 4, generator=torch.Generator().manual_seed(int(seed)))
    om = O.OracleDeepLIIF(cfg, nets, vgg_sd=_vgg_sd())
    A = seeded_uniform((2, 3, 64, 64), 22)
    B = [seeded_uniform((2, 3, 64, 64), 23 + i) for i in range(3)]
    for s in range(2):
        om.set_input({'A': A, 'B': B})
        om.optimize_parameters()
        got = om.current_losses()
        tol = RTOL if s == 0 else 2e-3
        for k, exp in zip(VGG['step/loss_names'], VGG[f'step{s}/losses']):
            assert abs(got[str(k)] - exp) <= tol * max(1.0, abs(exp)), (s, k, got[str(k)], exp)
        assert np.allclose(om.vgg_losses, VGG[f'step{s}/vgg'], rtol=tol)
        for i in range(2):
            assert rel_err(om.fake_B[i].detach()[:, :, ::2, ::2], VGG[f'step{s}/fake_B_{i + 1}']) < (RTOL if s == 0 else 2e-2)


# ---------------------------------------------------------------------------------------------------------------------------
# attention U-Net (unet_512_attention, att_unet.py): oracle restatement vs the reference-generated fixture
# ---------------------------------------------------------------------------------------------------------------------------
import att_util  # noqa: E402


@pytest.mark.parametrize('tag', att_util.TAGS)
def test_attention_unet_forward_backward(tag):
    cin, sd, x, r = att_util.case(tag)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running' not in k}
    xo = x.clone().requires_grad_(True)
    y = O.att_unet_generator(sd, xo, update_running=True)
    (y * r).sum().backward()
    with torch.no_grad():
        y_eval = O.att_unet_generator({k: v.detach() for k, v in sd.items()}, x)
    running = {k: v for k, v in sd.items() if 'running_' in k}
    att_util.check_against_fixture(tag, y.detach(), xo.grad, {k: p.grad for k, p in params.items()}, running, y_eval, RTOL)
