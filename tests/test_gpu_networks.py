"""GPU parity of the engine-backed networks and of the full DeepLIIF training step against the CPU oracle
(oracle/deepliif_oracle.py, pinned to the reference by tests/test_oracle_golden.py), through the product path:
deepliif_amd.networks / models -> engine -> ops.HipBackend -> libdeepliif_hip.so.

Tolerances.
  Forward outputs (max abs error relative to max |expected|):
    precision 'fp32' (strict parity mode, split-bf16 x3 MFMA): 1e-3 = the north-star bar (measured 2e-5 .. 9e-5)
    precision 'bf16' (throughput mode): bf16 storage + single-pass bf16 MFMA cannot meet 1e-3 through ~24 stacked
      conv+norm layers (each layer adds ~2^-9 relative noise); asserted bound 6e-2 (measured 6e-3 .. 4e-2).
  Gradients of whole networks: a 9-block ResNet with ReLU + norm layers is ill-conditioned in reverse mode -- perturbing
    the REFERENCE arithmetic itself (the fp32 CPU oracle) by 1.5e-5 relative noise after every conv moves its own input
    gradient by 3e-2 (relative L2) and by 25-40% for bf16-sized noise (ReLU masks flip).  A fixed 1e-3 bound on network
    gradients is therefore not a property any implementation has; the bound asserted here is
        err_engine <= max(floor, 4 x the oracle's own sensitivity to noise of the engine's per-layer rounding size)
    with floor 1e-3 (fp32) / 3e-2 (bf16).  Per-kernel gradient parity (tests/test_gpu_kernels.py) is tight (1e-4), and
    the end-to-end training-step test below checks losses / images / updated weights against the reference trajectory.
  Every measured value is written to gpurun_out/parity_errors.json (copied to profiles/ for DESIGN.md).
"""
import json
import os
import types

import numpy as np
import pytest
import torch

from deepliif_amd import _lib as L
from deepliif_amd import engine as E
from deepliif_amd import models as M
from deepliif_amd import networks as N
from deepliif_amd import ops
from golden_util import digest_close, seeded_uniform
from oracle import deepliif_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
G = os.path.join(os.path.dirname(__file__), 'golden')
ERRLOG = {}
LOSS_FLOOR = 0.25         # absolute floor of the relative loss errors: the GAN / L1 losses of the trajectories are 0.5 ... 20; only the lsgan
                          # D_fake_S terms are smaller (0.02 at step 0, 0.11 after the first update) and are judged against the floor


@pytest.fixture(autouse=True)
def _real_backend():
    ops._impl = None
    yield
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_errors.json', 'w') as f:
        json.dump(ERRLOG, f, indent=1, sort_keys=True)


def rel(a, b, floor=1e-30):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(floor))


with open(os.path.join(G, 'step_noise_floor.json')) as _f:
    NOISE_FLOOR = json.load(_f)
TOL_OUT = {'fp32': 1e-3, 'bf16': 6e-2}
GRAD_FLOOR = {'fp32': 1e-3, 'bf16': 3e-2}
LAYER_NOISE = {'fp32': 1.5e-5, 'bf16': 4e-3}      # relative rounding noise per conv output of each precision policy


def l2(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


class conv_noise:
    """Context manager: every F.conv2d / F.conv_transpose2d output gets eps-relative gaussian noise (oracle sensitivity)."""

    def __init__(self, eps, seed):
        self.eps, self.g = eps, torch.Generator().manual_seed(seed)

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.c, self.ct = F, F.conv2d, F.conv_transpose2d

        def nz(y):
            return y + self.eps * y.detach().abs().mean() * torch.randn(y.shape, generator=self.g)
        F.conv2d = lambda *a, **k: nz(self.c(*a, **k))
        F.conv_transpose2d = lambda *a, **k: nz(self.ct(*a, **k))

    def __exit__(self, *exc):
        self.F.conv2d, self.F.conv_transpose2d = self.c, self.ct

CASES = [
    ('resnet_9blocks', 3, 8, 'batch', 'zero', (2, 3, 32, 32)),
    ('resnet_9blocks', 3, 16, 'instance', 'zero', (2, 3, 64, 48)),
    ('resnet_2blocks', 3, 8, 'batch', 'reflect', (1, 3, 32, 32)),
    ('resnet_9blocks', 3, 8, 'instance', 'reflect', (2, 3, 40, 24)),    # reflect-padded training graph, H != W
    ('resnet_9blocks', 3, 8, 'instance', 'zero', (1, 3, 72, 104)),      # batch 1, H != W, not a multiple of any tile size
    ('n_layers', 6, 8, 'instance', 'zero', (3, 6, 100, 76)),           # odd batch, sizes that leave odd feature maps in the PatchGAN
    ('unet_32', 3, 8, 'batch', 'zero', (2, 3, 32, 32)),
    ('unet_64', 9, 8, 'instance', 'zero', (1, 9, 64, 64)),
    ('unet_512', 3, 8, 'batch', 'zero', (1, 3, 512, 512)),
    ('n_layers', 6, 8, 'batch', 'zero', (2, 6, 64, 64)),
    ('n_layers', 12, 16, 'instance', 'zero', (1, 12, 128, 128)),
]


def build(arch, cin, nf, norm, pad):
    if arch == 'n_layers':
        return N.define_D(cin, nf, 'n_layers', 4, norm, 'normal', 0.02, [0])
    return N.define_G(cin, 3, nf, arch, norm, False, 'normal', 0.02, [0], pad)


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('arch,cin,nf,norm,pad,shape', CASES, ids=lambda v: str(v).replace(' ', ''))
def test_network_forward_backward(arch, cin, nf, norm, pad, shape, precname):
    sd = O.random_state_dict(arch, cin, 3, nf, norm, pad, 4, generator=torch.Generator().manual_seed(5))
    net = build(arch, cin, nf, norm, pad)
    net.load_state_dict(sd, strict=True)
    net.train()
    x = seeded_uniform(shape, 6)
    prec = E.Precision.get(precname)
    train = True                        # (reflect padding included: pad-0 data-gradient plan over the padded extent + dl_reflect_fold)
    tape = E.Tape() if train else None
    ctx = E.Ctx(prec, tape, training=train)
    xa = E.to_engine(x.to(DEV), prec)
    xa.needs_grad = train
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    ya = net.run(ctx, xa)
    y = E.from_engine(ya)
    # oracle (CPU fp32), clean and with per-layer noise of this precision's size
    def oracle(noise_seed=None):
        sdo = {k: v.clone() for k, v in sd.items()}
        params = {k: v.requires_grad_(True) for k, v in sdo.items() if v.is_floating_point() and 'running' not in k}
        xo = x.clone().requires_grad_(True)

        def fwd():
            if arch == 'n_layers':
                return O.nlayer_discriminator(sdo, xo, norm, 4)
            return O.run_generator(arch, sdo, xo, norm, pad)
        if noise_seed is None:
            yo = fwd()
        else:
            with conv_noise(LAYER_NOISE[precname], noise_seed):
                yo = fwd()
        r = torch.randn(yo.shape, generator=torch.Generator().manual_seed(7))
        grads = torch.autograd.grad((yo * r).sum(), [xo] + list(params.values())) if train else None
        return yo.detach(), r, grads, list(params.keys())

    yo, r, grads, keys = oracle()
    e_out = rel(y, yo)
    tag = f'{arch}-{cin}-{nf}-{norm}-{pad}-{precname}'
    ERRLOG[tag + '/y'] = e_out
    assert e_out < TOL_OUT[precname]
    if not train:
        return
    ya.grad = E.to_engine(r.to(DEV), prec).t
    tape.backward()
    dx = E.from_engine(E.Act(xa.grad, xa.C))
    named = dict(net.named_parameters())
    dw_engine = torch.cat([named[k].grad.reshape(-1).cpu() for k in keys])
    dw_oracle = torch.cat([g.reshape(-1) for g in grads[1:]])
    e_dx, e_dw = l2(dx, grads[0]), l2(dw_engine, dw_oracle)
    # the oracle's own gradient sensitivity to noise of this size (worst of the noise draws)
    # On small feature maps the sensitivity is QUANTISED by single ReLU mask flips: for resnet_9blocks / ngf 8 / 2x3x40x24 twelve
    # draws of fp32-sized noise gave dx changes of 4.5e-5 (no flip), 1.3e-3, 9.5e-3 (4 of 12 draws), 1.2e-2 -- two draws can
    # easily land on the low values while the GPU arithmetic flips the 9.5e-3 unit.  Eight draws for inputs below 256 x 256
    # (cheap there); large inputs average over many units and keep two.
    s_dx = s_dw = 0.0
    for seed in (range(1, 9) if shape[2] * shape[3] < 256 * 256 else (1, 2)):
        _, _, gn, _ = oracle(seed)
        s_dx = max(s_dx, l2(gn[0], grads[0]))
        s_dw = max(s_dw, l2(torch.cat([g.reshape(-1) for g in gn[1:]]), dw_oracle))
    ERRLOG[tag + '/dx_l2'], ERRLOG[tag + '/dw_l2'] = e_dx, e_dw
    ERRLOG[tag + '/oracle_sensitivity_dx_l2'], ERRLOG[tag + '/oracle_sensitivity_dw_l2'] = s_dx, s_dw
    assert e_dx <= max(GRAD_FLOOR[precname], 4 * s_dx), (e_dx, s_dx)
    assert e_dw <= max(GRAD_FLOOR[precname], 4 * s_dw), (e_dw, s_dw)


def test_inference_golden_fixture_from_reference():
    """run_dask on the fixture produced by the REFERENCE's run_dask (tests/golden/inference_small.npz): three tiles in
    one batch must each equal the reference's single-tile outputs (per-sample normalisation)."""
    from deepliif_amd import inference as I
    z = np.load(os.path.join(G, 'inference_small.npz'))
    opt = types.SimpleNamespace(model='DeepLIIF', modalities_no=4, seg_gen=True, mod_id_seg='S', input_id=0, input_nc=3, output_nc=3, ngf=8,
                                norm='batch', padding='zero', net_g='resnet_9blocks', net_gs='unet_64', input_no=1,
                                modalities_names=['IHC', 'Hema', 'DAPI', 'Lap2', 'Marker'], gpu_ids=[0])
    nets = I.build_generators(opt, torch.device('cuda', 0), 'fp32')
    for name, seed in zip(z['net_names'], z['net_seeds']):
        name = str(name)
        seg = len(name) > 2
        sd = O.random_state_dict('unet_64' if seg else 'resnet_9blocks', 3, 3, 8, 'batch', 'reflect' if seg else 'zero',
                                 generator=torch.Generator().manual_seed(int(seed)))
        nets[name].load_state_dict(sd)
    tiles = seeded_uniform((3, 3, 64, 64), 32)
    res = I.run_dask(tiles.to(DEV), nets=nets, opt=opt, seg_weights=[float(w) for w in z['seg_weights']], output_tensor=True)
    assert list(res.keys()) == [str(k) for k in z['keys']]
    for t in range(3):
        for k in res:
            e = rel(res[k][t:t + 1], torch.from_numpy(z[f'tile{t}/{k}']))
            ERRLOG[f'inference/{k}/tile{t}'] = e
            assert e < 1e-3, (t, k, e)


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
def test_inference_dag_replays_from_a_captured_hip_graph(precname):
    """All launches of the engine go to torch's current HIP stream through ctypes, so torch.cuda.graph captures a whole generator DAG
    (DESIGN 5): the replayed graph must reproduce the eager outputs bit for bit, also after the input buffer has been overwritten in place."""
    from deepliif_amd import inference as I
    torch.manual_seed(11)
    opt = types.SimpleNamespace(model='DeepLIIF', modalities_no=2, seg_gen=True, mod_id_seg='S', input_id=0, input_nc=3, output_nc=3, ngf=8,
                                norm='batch', padding='zero', net_g='resnet_9blocks', net_gs='unet_64', input_no=1,
                                modalities_names=['input1', 'mod1', 'mod2'], gpu_ids=[0])
    nets = I.build_generators(opt, torch.device('cuda', 0), precname)
    prec = E.Precision.get(precname)
    x = E.to_engine(seeded_uniform((4, 3, 64, 64), 5).to(DEV), prec)
    x2 = E.to_engine(seeded_uniform((4, 3, 64, 64), 6).to(DEV), prec)
    run = lambda: I.run_generators_engine(x, nets, opt)
    eager1 = {k: v.t.clone() for k, v in run().items()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):             # warm-up on a side stream: grow-only scratch buffers and packed weight images exist before the capture
        run()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    g.replay()
    torch.cuda.synchronize()
    for k in eager1:
        assert torch.equal(out[k].t, eager1[k]), k
    keep = x.t.clone()
    x.t.copy_(x2.t)                           # new tiles in the captured input buffer
    g.replay()
    torch.cuda.synchronize()
    replay2 = {k: v.t.clone() for k, v in out.items()}
    eager2 = {k: v.t.clone() for k, v in run().items()}
    torch.cuda.synchronize()
    for k in eager2:
        assert torch.equal(replay2[k], eager2[k]), k
        assert not torch.equal(replay2[k], eager1[k]), k
    x.t.copy_(keep)


def make_opt(modalities_no, seg_gen, norm, net_gs, nf, precision):
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


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('tag', ['m1_noseg_batch', 'm5_noseg_instance', 'm4_seg_batch', 'm2_seg_instance_reflect'])
def test_training_step_golden_fixture_from_reference(tag, precname):
    """Two optimize_parameters() steps against the trajectory recorded from the REFERENCE DeepLIIFModel
    (tests/golden/step_*.npz): same seeded weights, same batch."""
    z = np.load(os.path.join(G, f'step_{tag}.npz'))
    mod_no, seg_gen, norm, padding, net_gs, size, nf, batch, steps = z['meta']
    opt = make_opt(int(mod_no), seg_gen == 'True', norm, net_gs, int(nf), precname)
    opt.padding = str(padding)              # 'reflect' for the m2_seg_instance_reflect trajectory (cli --padding reflect)
    model = M.create_model(opt)
    model.setup(opt)
    S_fix, S = str(z['mod_id_seg']), str(model.mod_id_seg)
    for name, seed in zip(z['model_names'], z['net_seeds']):
        name = str(name)
        mine = name.replace(S_fix, S, 1) if (len(name) > 2 and name[1] == S_fix[0]) else name
        if name.startswith('D'):
            arch, pad, cin = 'n_layers', 'zero', 6
        elif len(name) == 2:
            arch, pad, cin = 'resnet_9blocks', padding, 3
        else:
            arch, pad, cin = net_gs, 'reflect', 3
        sd = O.random_state_dict(arch, cin, 3, int(nf), norm, pad, 4, generator=torch.Generator().manual_seed(int(seed)))
        getattr(model, 'net' + mine).load_state_dict(sd)
    size, batch = int(size), int(batch)
    nB = int(mod_no) + (1 if seg_gen == 'True' else 0)
    A = seeded_uniform((batch, 3, size, size), 22)
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(nB)]
    # step 0 is a pure function of the inputs; step 1 inherits the first Adam update, which is +-lr*sign(g) per weight and
    # therefore amplifies gradient noise (see the module docstring): looser bounds there
    ltol = {'fp32': (1e-3, 5e-3), 'bf16': (3e-2, 6e-2)}[precname]
    otol = {'fp32': (1e-3, 8e-2), 'bf16': (6e-2, 3e-1)}[precname]
    for s in range(int(steps)):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        got = model.get_current_losses()
        for name, exp in zip(z['loss_names'], z[f'step{s}/losses']):
            name = str(name)
            mine = name[:-len(S_fix)] + S if name.endswith('_' + S_fix) else name
            err = abs(got[mine] - exp) / max(abs(exp), LOSS_FLOOR)       # true relative error; values below the floor (lsgan D_fake_S) are judged against it
            ERRLOG[f'step/{tag}/{precname}/s{s}/{mine}'] = err
            assert err <= ltol[min(s, 1)], (s, mine, got[mine], exp)
        # image bound = max(the hand-set bound, 1.5 x the NOISE FLOOR of this trajectory): the deviation of the oracle itself when
        # every conv output carries rounding noise of this policy's size (tests/golden/step_noise_floor.json, max over 6 draws,
        # generated by tests/golden/make_noise_floor.py).  After the first Adam update the floor differs a lot between
        # trajectories (bf16 seg output: 0.23 for the batch-norm fixture, 0.34 for the instance-norm / reflect one); the
        # engine's error is one more draw from that long-tailed distribution, hence the factor.
        def bound(key):
            return max(otol[min(s, 1)], 1.5 * NOISE_FLOOR[f'{tag}/{precname}'][f's{s}/{key}'])
        for i in range(int(mod_no)):
            e = rel(getattr(model, f'fake_B_{i + 1}')[:, :, ::2, ::2], torch.from_numpy(z[f'step{s}/fake_B_{i + 1}']))
            ERRLOG[f'step/{tag}/{precname}/s{s}/fake_B_{i + 1}'] = e
            assert e < bound(f'fake_B_{i + 1}'), (s, i, e, bound(f'fake_B_{i + 1}'))
        if seg_gen == 'True':
            e = rel(getattr(model, f'fake_B_{S}')[:, :, ::2, ::2], torch.from_numpy(z[f'step{s}/fake_B_S']))
            ERRLOG[f'step/{tag}/{precname}/s{s}/fake_B_S'] = e
            assert e < bound('fake_B_S'), (s, e, bound('fake_B_S'))
        if precname == 'fp32':
            for name in z['model_names']:
                name = str(name)
                mine = name.replace(S_fix, S, 1) if (len(name) > 2 and name[1] == S_fix[0]) else name
                sd = getattr(model, 'net' + mine).state_dict()
                flat = torch.cat([v.reshape(-1).float().cpu() for v in sd.values() if v.is_floating_point()])
                # |dw| after one Adam step ~ 1% of |w| (every weight moves by exactly lr); 8e-3 of |w| = 80% of the update norm: catches a
                # wrong learning rate, bias correction or update direction (>= 100%), tolerates the sign flips of noise-level gradients
                # (a few % of the weights; the reflect case measured 6.4e-3 once the stem moved to the patch kernel -- a different summation order -- and < 6e-3 before)
                ok, msg = digest_close(flat, z[f'step{s}/w_digest/{name}'], 8e-3)
                assert ok, f'step {s} weights of {name}: {msg}'


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
def test_deepliif_ext_step_golden_fixture_from_reference(precname):
    """DeepLIIFExtModel (9-channel seg generators, 12-channel seg discriminators) against the reference trajectory."""
    z = np.load(os.path.join(G, 'step_ext_m2_batch.npz'))
    Mn, norm, size, nf, batch, steps = z['meta']
    Mn, size, nf, batch = int(Mn), int(size), int(nf), int(batch)
    opt = make_opt(Mn, True, norm, 'unet_64', nf, precname)
    opt.model, opt.net_ds = 'DeepLIIFExt', 'n_layers'
    opt.loss_G_weights = opt.loss_D_weights = opt.seg_weights = [1.0 / Mn] * Mn
    model = M.create_model(opt)
    model.setup(opt)
    spec = {'G': ('resnet_9blocks', 3, 'zero'), 'GS': ('unet_64', 9, 'reflect'), 'D': ('n_layers', 6, 'zero'), 'DS': ('n_layers', 12, 'zero')}
    for name, seed in zip(z['model_names'], z['net_seeds']):
        arch, cin, pad = spec[str(name).split('_')[0]]
        model._net(str(name)).load_state_dict(O.random_state_dict(arch, cin, 3, nf, norm, pad, 4, generator=torch.Generator().manual_seed(int(seed))))
    A = seeded_uniform((batch, 3, size, size), 22)
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(Mn)]
    BS = [seeded_uniform((batch, 3, size, size), 43 + i) for i in range(Mn)]
    ltol = {'fp32': (1e-3, 5e-3), 'bf16': (3e-2, 6e-2)}[precname]
    otol = {'fp32': (1e-3, 8e-2), 'bf16': (6e-2, 3e-1)}[precname]
    for s in range(int(steps)):
        model.set_input({'A': A, 'B': B, 'BS': BS, 'A_paths': ['x']})
        model.optimize_parameters()
        got = model.get_current_losses()
        for name, exp in zip(z['loss_names'], z[f'step{s}/losses']):
            err = abs(got[str(name)] - exp) / max(abs(exp), LOSS_FLOOR)
            ERRLOG[f'step_ext/{precname}/s{s}/{name}'] = err
            assert err <= ltol[min(s, 1)], (s, name, got[str(name)], exp)
        for i in range(Mn):
            for fam, t in (('fake_B', model.fake_B[i]), ('fake_BS', model.fake_BS[i])):
                e = rel(t[:, :, ::2, ::2], torch.from_numpy(z[f'step{s}/{fam}_{i + 1}']))
                ERRLOG[f'step_ext/{precname}/s{s}/{fam}_{i + 1}'] = e
                assert e < otol[min(s, 1)]


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
def test_sdg_step_golden_fixture_from_reference(precname):
    """SDGModel (input modalities concatenated: 6-channel generators, 9-channel discriminators) against the reference trajectory,
    including the reference's loss_names (the fixture's reference ran with the VGG term zeroed; here lambda_feat = 0 reports G_VGG_i as NaN)."""
    z = np.load(os.path.join(G, 'step_sdg_m2_in2_instance.npz'))
    Mn, input_no, norm, size, nf, batch, steps = z['meta']
    Mn, input_no, size, nf, batch = int(Mn), int(input_no), int(size), int(nf), int(batch)
    opt = make_opt(Mn, False, norm, 'unet_64', nf, precname)
    opt.model, opt.input_no = 'SDG', input_no
    opt.loss_G_weights = opt.loss_D_weights = opt.seg_weights = [1.0 / Mn] * Mn
    model = M.create_model(opt)
    model.setup(opt)
    assert list(model.loss_names) == [str(n) for n in z['loss_names']]
    spec = {'G': ('resnet_9blocks', 3 * input_no, 'zero'), 'D': ('n_layers', 3 * input_no + 3, 'zero')}
    for name, seed in zip(z['model_names'], z['net_seeds']):
        arch, cin, pad = spec[str(name).split('_')[0]]
        model._net(str(name)).load_state_dict(O.random_state_dict(arch, cin, 3, nf, norm, pad, 4, generator=torch.Generator().manual_seed(int(seed))))
    A = [seeded_uniform((batch, 3, size, size), 22 + 100 * k) for k in range(input_no)]
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(Mn)]
    ltol = {'fp32': (1e-3, 5e-3), 'bf16': (3e-2, 6e-2)}[precname]
    otol = {'fp32': (1e-3, 8e-2), 'bf16': (6e-2, 3e-1)}[precname]
    for s in range(int(steps)):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        got = model.get_current_losses()
        for name, exp in zip(z['loss_names'], z[f'step{s}/losses']):
            if '_VGG_' in str(name):      # not evaluated (lambda_feat = 0; the fixture's reference had it zeroed): reported as NaN, not as 0.0
                assert got[str(name)] != got[str(name)]
                continue
            err = abs(got[str(name)] - exp) / max(abs(exp), LOSS_FLOOR)
            ERRLOG[f'step_sdg/{precname}/s{s}/{name}'] = err
            assert err <= ltol[min(s, 1)], (s, name, got[str(name)], exp)
        for i in range(Mn):
            e = rel(model.fake_B[i][:, :, ::2, ::2], torch.from_numpy(z[f'step{s}/fake_B_{i + 1}']))
            ERRLOG[f'step_sdg/{precname}/s{s}/fake_B_{i + 1}'] = e
            assert e < otol[min(s, 1)]


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
def test_inference_seam_is_thread_safe(precname):
    """The reference drives different generators concurrently from dask worker threads (deepliif/models/__init__.py:283-334,
    SURVEY 8b) through net(tensor).  Small tiles put every conv on the split-K path, whose scratch slab is filled by one launch and
    reduced by the next: with process-wide scratch a second thread could overwrite it in between.  Outputs of concurrent calls
    must be bit-identical to the serial ones -- different nets in parallel, and the same net from two threads."""
    import threading
    torch.manual_seed(3)
    nets = [N.define_G(3, 3, 16, 'resnet_9blocks', 'instance', False, 'normal', 0.02, [0], 'zero'),
            N.define_G(3, 3, 16, 'unet_64', 'batch', False, 'normal', 0.02, [0], 'zero'),
            N.define_G(9, 3, 8, 'unet_64', 'instance', False, 'normal', 0.02, [0], 'zero'),
            N.define_G(3, 3, 8, 'resnet_9blocks', 'batch', False, 'normal', 0.02, [0], 'reflect')]
    for net in nets:
        net.precision = precname
        net.eval()
    xs = [seeded_uniform((2, 9 if i == 2 else 3, 64, 64), 70 + i).to(DEV) for i in range(len(nets))]
    K = 12
    with torch.no_grad():
        serial = [[net(x).clone() for _ in range(K)] for net, x in zip(nets, xs)]
    for i in range(len(nets)):
        for k in range(1, K):
            assert torch.equal(serial[i][k], serial[i][0]), 'serial runs must be deterministic to begin with'
    results, errors = {}, []
    start = threading.Barrier(len(nets) + 2)

    def worker(tag, net, x):
        try:
            start.wait()
            with torch.no_grad():
                results[tag] = [net(x).clone() for _ in range(K)]
        except Exception as e:          # surface the failure in the main thread
            errors.append((tag, repr(e)))

    jobs = [(f'net{i}', nets[i], xs[i]) for i in range(len(nets))] + [('net0/b', nets[0], xs[0]), ('net1/b', nets[1], xs[1])]
    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for tag, outs in results.items():
        ref = serial[int(tag[3])][0]
        bad = [k for k, o in enumerate(outs) if not torch.equal(o, ref)]
        assert not bad, f'{tag}: {len(bad)} of {K} concurrent results differ from the serial result (first at iteration {bad[0]})'


# ---------------------------------------------------------------------------------------------------------------------------
# VGG19 perceptual term: reference vectors from tests/golden/make_golden_vgg.py
# ---------------------------------------------------------------------------------------------------------------------------
def _vgg_file(tmp_path):
    z = np.load(os.path.join(G, 'vgg_cases.npz'))
    path = os.path.join(str(tmp_path), 'vgg19.pth')
    torch.save(O.random_vgg19_state_dict(torch.Generator().manual_seed(int(z['vgg_seed']))), path)
    return z, path


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
def test_vgg_loss_matches_reference_vector(tmp_path, precname):
    from deepliif_amd import networks as N
    z, path = _vgg_file(tmp_path)
    crit = N.VGGLoss(path, torch.device(DEV), precname)
    prec = E.Precision.get(precname)
    for tag in ('s64', 's48x80'):
        shape = tuple(int(v) for v in z[f'{tag}/shape'])
        tape = E.Tape()
        ctx = E.Ctx(prec, tape, training=True)
        x = E.to_engine(seeded_uniform(shape, 71).to(DEV), prec)
        x.needs_grad = True
        y = E.to_engine(seeded_uniform(shape, 72).to(DEV), prec)
        out = torch.zeros(1, device=DEV)
        crit.run(ctx, x, y, 1.0, out)
        tape.backward()
        torch.cuda.synchronize()
        lerr = abs(float(out) - float(z[f'{tag}/loss'])) / float(z[f'{tag}/loss'])
        dx = E.from_engine(E.Act(x.grad, 3)).cpu()
        ref = torch.from_numpy(z[f'{tag}/dx'])
        gmax = float((dx - ref).abs().max() / ref.abs().max())
        gerr = float((dx - ref).norm() / ref.norm())
        ERRLOG[f'vgg/{precname}/{tag}/loss'] = lerr
        ERRLOG[f'vgg/{precname}/{tag}/dx_l2'] = gerr
        ERRLOG[f'vgg/{precname}/{tag}/dx_maxabs'] = gmax
        assert lerr < (1e-3 if precname == 'fp32' else 3e-2), (tag, lerr)
        # the loss is a sum of |a - b|: its gradient is a field of SIGNS routed through 13 ReLU masks and 4 arg-max choices, so 1e-5 of
        # arithmetic noise flips a handful of discrete decisions (deep features carry the largest weights: 1/numel of a 512 x 4 x 4 map).
        # Flips move isolated entries of dx by whole units -> judged in L2; the value of the loss and the training trajectory (next test)
        # are held to the usual 1e-3.
        assert gerr < (5e-2 if precname == 'fp32' else 4.5e-1), (tag, gerr, gmax)


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
def test_default_objective_with_vgg_follows_reference_trajectory(tmp_path, precname):
    """the reference's default training objective (lambda_feat = 100): two optimize_parameters() steps vs the reference trajectory"""
    z, path = _vgg_file(tmp_path)
    torch.manual_seed(0)
    opt = make_opt(2, True, 'batch', 'unet_64', 8, precname)
    opt.lambda_feat, opt.vgg_weights = 100, path
    opt.loss_G_weights = z['step/loss_G_weights'].tolist()
    model = M.create_model(opt)
    model.setup(opt)
    for n, seed in zip(z['step/model_names'], z['step/net_seeds']):
        n = str(n)
        arch, cin, pad = ('n_layers', 6, 'zero') if n.startswith('D') else (('resnet_9blocks', 3, 'zero') if n in ('G1', 'G2') else ('unet_64', 3, 'reflect'))
        getattr(model, 'net' + n).load_state_dict(O.random_state_dict(arch, cin, 3, 8, 'batch', pad, 4, generator=torch.Generator().manual_seed(int(seed))))
    A = seeded_uniform((2, 3, 64, 64), 22)
    B = [seeded_uniform((2, 3, 64, 64), 23 + i) for i in range(3)]
    ltol = {'fp32': (1e-3, 5e-3), 'bf16': (3e-2, 6e-2)}[precname]
    for s in range(2):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        got = model.get_current_losses()
        for k, exp in zip(z['step/loss_names'], z[f'step{s}/losses']):
            err = abs(got[str(k)] - exp) / max(abs(exp), LOSS_FLOOR)
            ERRLOG[f'step_vgg/{precname}/s{s}/{k}'] = err
            assert err <= ltol[s], (s, k, got[str(k)], exp)
        for i in range(2):
            err = abs(float(getattr(model, f'loss_G_VGG_{i + 1}')) - z[f'step{s}/vgg'][i]) / z[f'step{s}/vgg'][i]
            ERRLOG[f'step_vgg/{precname}/s{s}/G_VGG_{i + 1}'] = err
            assert err <= ltol[s], (s, i, err)


@pytest.mark.parametrize('tag', ['m1_noseg_batch', 'm5_noseg_instance', 'm4_seg_batch', 'm2_seg_instance_reflect'])
def test_policy_variant_fp32_storage_bf16_products(tag):
    """Policy `fp32_bf16mma` (fp32 activations / statistics / residual stream everywhere, ONE bf16 MFMA pass per product) on the
    reference trajectories, step 0 (a pure function of inputs and weights).  Its error is what any "bf16 MFMA + wider storage" mix
    can reach at best -- every such mix rounds the same products -- so it decides whether a policy between `bf16` and `fp32`
    (split-bf16 x3) could meet the 1e-3 bar: the measured values go to profiles/parity_errors_r02.json under 'policy/...'."""
    z = np.load(os.path.join(G, f'step_{tag}.npz'))
    mod_no, seg_gen, norm, padding, net_gs, size, nf, batch, steps = z['meta']
    opt = make_opt(int(mod_no), seg_gen == 'True', norm, net_gs, int(nf), 'fp32_bf16mma')
    opt.padding = str(padding)
    model = M.create_model(opt)
    model.setup(opt)
    S_fix, S = str(z['mod_id_seg']), str(model.mod_id_seg)
    for name, seed in zip(z['model_names'], z['net_seeds']):
        name = str(name)
        mine = name.replace(S_fix, S, 1) if (len(name) > 2 and name[1] == S_fix[0]) else name
        if name.startswith('D'):
            arch, pad, cin = 'n_layers', 'zero', 6
        elif len(name) == 2:
            arch, pad, cin = 'resnet_9blocks', padding, 3
        else:
            arch, pad, cin = net_gs, 'reflect', 3
        sd = O.random_state_dict(arch, cin, 3, int(nf), norm, pad, 4, generator=torch.Generator().manual_seed(int(seed)))
        getattr(model, 'net' + mine).load_state_dict(sd)
    size, batch = int(size), int(batch)
    nB = int(mod_no) + (1 if seg_gen == 'True' else 0)
    A = seeded_uniform((batch, 3, size, size), 22)
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(nB)]
    model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
    model.optimize_parameters()
    got = model.get_current_losses()
    worst_loss = worst_img = 0.0
    for name, exp in zip(z['loss_names'], z['step0/losses']):
        name = str(name)
        mine = name[:-len(S_fix)] + S if name.endswith('_' + S_fix) else name
        worst_loss = max(worst_loss, abs(got[mine] - exp) / max(abs(exp), LOSS_FLOOR))
    for i in range(int(mod_no)):
        worst_img = max(worst_img, rel(getattr(model, f'fake_B_{i + 1}')[:, :, ::2, ::2], torch.from_numpy(z[f'step0/fake_B_{i + 1}'])))
    if seg_gen == 'True':
        worst_img = max(worst_img, rel(getattr(model, f'fake_B_{S}')[:, :, ::2, ::2], torch.from_numpy(z['step0/fake_B_S'])))
    ERRLOG[f'policy/fp32_bf16mma/{tag}/s0/losses_max'] = worst_loss
    ERRLOG[f'policy/fp32_bf16mma/{tag}/s0/images_max'] = worst_img
    # same class as the bf16 policy (the products are what is rounded), far outside the strict 1e-3 bar that `fp32` meets
    assert worst_loss <= 3e-2 and worst_img <= 6e-2, (worst_loss, worst_img)


@pytest.mark.parametrize('norm', ['instance', 'batch'])
def test_benched_configuration_full_size_step(norm):
    """The configuration bench.py times (BASELINE configs[2] per GPU: 5x Resnet-9 G + 5x NLayerD, 512x512, batch 8, ngf 64; `--norm instance` is
    bench.py's default and BASELINE.json's wording, `batch` the reference CLI's default -- SURVEY 0 #2) under test
    itself: one optimize_parameters() of the bf16 policy and one of the strict fp32 policy from the same seeded weights and batch --
    every loss finite on both, the bf16 losses within the bf16 step-0 bound of the strict ones (the bound the fixture-size
    trajectories assert against the reference, tests above), and a second bf16 step still finite after the Adam update."""
    import argparse
    import bench
    args = argparse.Namespace(ngf=64, norm=norm, precision='bf16', batch=8, size=512)
    g = torch.Generator().manual_seed(4321)
    batch = {'A': (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).to(DEV),
             'B': [(torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).to(DEV) for _ in range(5)], 'A_paths': ['synthetic']}
    losses = {}
    for prec in ('bf16', 'fp32'):
        torch.manual_seed(0)
        args.precision = prec
        opt = bench.make_opt(args, 0)
        model = M.create_model(opt)
        model.setup(opt)
        model.set_input(batch)
        model.optimize_parameters()
        torch.cuda.synchronize()
        losses[prec] = dict(model.get_current_losses())
        assert len(losses[prec]) >= 20 and all(np.isfinite(v) for v in losses[prec].values()), losses[prec]
        if prec == 'bf16':
            model.set_input(batch)
            model.optimize_parameters()
            second = model.get_current_losses()
            assert all(np.isfinite(v) for v in second.values()), second
            for i in range(1, 6):
                img = getattr(model, f'fake_B_{i}')
                assert tuple(img.shape) == (8, 3, 512, 512) and bool(torch.isfinite(img).all()) and float(img.abs().max()) <= 1.0
        del model
        torch.cuda.empty_cache()
    worst = max(abs(losses['bf16'][k] - losses['fp32'][k]) / max(1.0, abs(losses['fp32'][k])) for k in losses['fp32'])
    ERRLOG[f'fullsize/train_5g5d_512_b8_{norm}/bf16_vs_fp32_losses_max_rel'] = worst
    assert worst <= 3e-2, (worst, losses)


def _resnet_units(net):
    """(name, torch modules of the unit, residual?, bound engine layers) for every conv -> norm -> act unit of a ResnetGenerator, in order."""
    m, b = net.model, net._layers()
    units = [('stem', [m[0], m[1], m[2], m[3]], False, b['stem'], L.ACT_RELU)]
    idx = 4
    for i in range(2):
        units.append((f'down{i}', [m[idx], m[idx + 1], m[idx + 2]], False, b['down'][i], L.ACT_RELU))
        idx += 3
    for k, (ent, blk) in enumerate(b['blocks']):
        cb = blk.conv_block
        units.append((f'block{k}.a', [cb[blk.idx['conv0']], cb[blk.idx['norm0']], cb[blk.idx['norm0'] + 1]], False, ent[0], L.ACT_RELU))
        units.append((f'block{k}.b', [cb[blk.idx['conv1']], cb[blk.idx['norm1']]], True, ent[1], L.ACT_NONE))
        idx += 1
    for i in range(2):
        units.append((f'up{i}', [m[idx], m[idx + 1], m[idx + 2]], False, b['up'][i], L.ACT_RELU))
        idx += 3
    units.append(('head', [m[idx], m[idx + 1], m[idx + 2]], False, (b['head'], None), L.ACT_TANH))
    return units


def _nlayer_units(net):
    """the same for an NLayerDiscriminator: conv + LeakyReLU, three conv -> norm -> LeakyReLU units, the 1-channel output conv"""
    m, b = net.model, net._layers()
    units = [('first', [m[0], m[1]], False, (b['first'], None), L.ACT_LRELU)]
    idx = 2
    for i, (c, nl) in enumerate(b['mid']):
        units.append((f'mid{i}', [m[idx], m[idx + 1], m[idx + 2]], False, (c, nl), L.ACT_LRELU))
        idx += 3
    units.append(('last', [m[idx]], False, (b['last'], None), L.ACT_NONE))
    return units


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('tag', ['in3_n1', 'in9_n2'])
def test_attention_unet_against_reference_fixture(tag, precname):
    """`--net-gs unet_512_attention` (AttU_Net, att_unet.py:117-199) on the GPU against the reference-generated fixture: outputs (training-mode
    and inference-mode forward) at 1e-3 for the strict policy, BatchNorm running statistics, and the gradients at the bound the fixture
    can carry (att_util.GRAD_TOL: the reference's own fp32 backward is 1e-2 ... 4e-2 from an fp64 evaluation at random initialisation)."""
    import att_util
    cin, sd, x, r = att_util.case(tag)
    net = N.define_G(cin, 3, 64, 'unet_512_attention', 'batch', False, 'normal', 0.02, [0])
    net.load_state_dict(sd, strict=True)
    net.set_precision(precname).train()
    prec = E.Precision.get(precname)
    tape = E.Tape()
    ctx = E.Ctx(prec, tape, training=True)
    xa = E.to_engine(x.to(DEV), prec)
    xa.needs_grad = True
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    ya = net.run(ctx, xa)
    y = E.from_engine(ya)
    ya.grad = E.to_engine(r.to(DEV), prec).t
    tape.backward()
    dx = E.from_engine(E.Act(xa.grad, xa.C))
    running = {k: v.clone() for k, v in net.state_dict().items() if 'running_' in k}
    net.eval()
    net.batched_per_sample_norm = False          # the fixture's inference forward is ONE reference forward over the whole batch
    with torch.no_grad():
        y_eval = net(x.to(DEV))
    if precname == 'fp32':
        errs = att_util.check_against_fixture(tag, y.cpu(), dx.cpu(), {k: p.grad.cpu() for k, p in net.named_parameters()}, {k: v.cpu() for k, v in running.items()},
                                              y_eval.cpu(), 1e-3)
    else:
        errs = {'y': att_util.rel(y.cpu()[:, :, ::8, ::8], att_util.Z[f'{tag}/y_strided']), 'y_eval': att_util.rel(y_eval.cpu()[:, :, ::8, ::8], att_util.Z[f'{tag}/y_eval_strided'])}
        assert errs['y'] < 8e-2 and errs['y_eval'] < 8e-2, errs
    for k, v in errs.items():
        ERRLOG[f'att_unet/{tag}/{precname}/{k}'] = v


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('level,shape', [(2, (2, 16, 24)), (4, (2, 8, 8)), (8, (3, 2, 2))], ids=lambda v: str(v).replace(' ', ''))
def test_teacher_forced_attention_block_gradients(level, shape, precname):
    """One Attention_block (att_unet.py:88-115) at a fixed tolerance: the fp32 torch teacher's g, x and upstream gradient go in, the engine's
    three 1x1 conv + BatchNorm units, relu(g1 + x1) (residual form of the norm kernel + input activation of the psi conv), the sigmoid and the
    gate x * psi must give the block output, dg, dx and every parameter gradient: 1e-3 for the strict policy."""
    import torch.nn.functional as F
    f, fi = O.ATT_GATE[level]
    n, hh, ww = shape
    net = N.define_G(3, 3, 64, 'unet_512_attention', 'batch', False, 'normal', 0.02, [0])
    net.set_precision(precname).train()
    blk = getattr(net, f'Att{level}')
    gen = torch.Generator().manual_seed(31 + level)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * (2.0 / m.weight.shape[1]) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
            elif isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1 + 0.1 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))
    tsd = {k: v.detach().cpu().clone() for k, v in blk.state_dict().items()}
    g = seeded_uniform((n, f, hh, ww), 41)
    x = seeded_uniform((n, f, hh, ww), 42)
    up = torch.randn((n, f, hh, ww), generator=torch.Generator().manual_seed(43))
    names = [k for k in tsd if tsd[k].is_floating_point() and 'running' not in k]

    def teacher(gg, xx, params):
        def bn(pre, t):
            m, v = t.mean((0, 2, 3), keepdim=True), t.var((0, 2, 3), unbiased=False, keepdim=True)
            return (t - m) / torch.sqrt(v + 1e-5) * params[pre + '.weight'].view(1, -1, 1, 1) + params[pre + '.bias'].view(1, -1, 1, 1)
        g1 = bn('W_g.1', F.conv2d(gg, params['W_g.0.weight'], params['W_g.0.bias']))
        x1 = bn('W_x.1', F.conv2d(xx, params['W_x.0.weight'], params['W_x.0.bias']))
        psi = torch.sigmoid(bn('psi.1', F.conv2d(torch.relu(g1 + x1), params['psi.0.weight'], params['psi.0.bias'])))
        return xx * psi

    tp = {k: tsd[k].clone().requires_grad_(True) for k in names}
    gr, xr = g.clone().requires_grad_(True), x.clone().requires_grad_(True)
    out_t = teacher(gr, xr, tp)
    ref = torch.autograd.grad(out_t, [gr, xr] + [tp[k] for k in names], up)
    prec = E.Precision.get(precname)
    a = net._layers()['atts'][level]
    params = dict(blk.named_parameters())
    for p in params.values():
        p.grad = torch.zeros_like(p)
    tape = E.Tape()
    ctx = E.Ctx(prec, tape, training=True)
    ga, xa = E.to_engine(g.to(DEV), prec), E.to_engine(x.to(DEV), prec)
    ga.needs_grad = xa.needs_grad = True
    g1 = E.norm_act(ctx, E.conv(ctx, ga, a['wg'][0], stats=True), a['wg'][1], L.ACT_NONE)
    s_ = E.norm_act(ctx, E.conv(ctx, xa, a['wx'][0], stats=True), a['wx'][1], L.ACT_NONE, residual=g1)
    p_ = E.act_op(ctx, E.norm_act(ctx, E.conv(ctx, s_, a['psi'][0], in_act=L.ACT_RELU, stats=True), a['psi'][1], L.ACT_NONE), L.ACT_SIGMOID)
    oa = E.gate(ctx, xa, p_)
    errs = {'y': rel(E.from_engine(oa), out_t.detach())}
    oa.grad = E.to_engine(up.to(DEV), prec).t
    tape.backward()
    errs['dg'] = l2(E.from_engine(E.Act(ga.grad, ga.C)), ref[0])
    errs['dx'] = l2(E.from_engine(E.Act(xa.grad, xa.C)), ref[1])
    scale = max(float(t.abs().max()) for t in ref[2:])
    errs['dparams'] = max(float((params[k].grad.cpu() - rp).abs().max()) / scale for k, rp in zip(names, ref[2:]))
    tol = {'fp32': 1e-3, 'bf16': 1.5e-1}[precname]        # bf16: the ReLU mask of relu(g1 + x1) flips where |g1 + x1| is below the bf16 step (see the Resnet test)
    for k, v in errs.items():
        ERRLOG[f'teacher_forced/attention{level}/{precname}/{k}'] = v
        assert v <= tol, (k, v)


def test_strict_step_with_and_without_split_copies_is_bit_identical(monkeypatch):
    """DL_NO_SPLIT_COPY A/B on the whole training step (strict policy): with the split copies the norm kernels write and the convolutions read,
    two optimize_parameters() steps give bit-identical weights and losses to the step that splits inside every conv kernel."""
    def run(split):
        monkeypatch.setattr(ops.HipBackend, 'supports_split', split)
        torch.manual_seed(3)
        opt = make_opt(2, True, 'batch', 'unet_64', 32, 'fp32')
        model = M.create_model(opt)
        model.setup(opt)
        A = seeded_uniform((2, 3, 128, 128), 22)
        B = [seeded_uniform((2, 3, 128, 128), 23 + i) for i in range(3)]
        losses = []
        for _ in range(2):
            model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
            model.optimize_parameters()
            losses.append(dict(model.get_current_losses()))
        torch.cuda.synchronize()
        return torch.cat([o.flat.data.clone() for o in model.optimizers]), losses
    w1, l1 = run(True)
    w0, l0 = run(False)
    assert l1 == l0 and torch.equal(w1, w0)
    # ... and with the split copies on, skipping the fp32 store of the norm backward where every consumer reads the copy (engine.conv:
    # split_backward_ok) changes nothing either, while it does skip stores (otherwise this would test nothing)
    calls, fcalls = [], []
    orig, forig = ops.HipBackend.norm_backward, ops.HipBackend.norm_forward
    monkeypatch.setattr(ops.HipBackend, 'norm_backward', lambda self, *a, **k: (calls.append(k.get('store_dy', True)), orig(self, *a, **k))[1])
    monkeypatch.setattr(ops.HipBackend, 'norm_forward', lambda self, *a, **k: (fcalls.append(k.get('store_z', True)), forig(self, *a, **k))[1])
    w2, l2 = run(True)
    assert calls.count(False) > 0 and calls.count(True) > 0
    assert fcalls.count(False) > 0 and fcalls.count(True) > 0        # the ResnetBlocks' inner activations (norm_act(sole_reader=...)) keep only their split copy
    monkeypatch.setattr(ops, '_SPLIT_ONLY_GRAD', False)
    calls.clear(); fcalls.clear()
    w3, l3 = run(True)
    assert calls.count(False) == 0 and fcalls.count(False) == 0
    assert l2 == l1 and torch.equal(w2, w1) and l3 == l1 and torch.equal(w3, w1)


def _unet_chain(net):
    chain, blk = [], net.model
    while blk is not None:
        chain.append(blk)
        blk = None if blk.innermost else blk.model[blk.pos['sub']]
    return chain


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('arch,norm,shape', [('unet_64', 'batch', (2, 3, 64, 64)), ('unet_64', 'instance', (1, 9, 64, 128)), ('unet_512', 'batch', (1, 3, 512, 512))],
                         ids=lambda v: str(v).replace(' ', ''))
def test_teacher_forced_unet_level_gradients(arch, norm, shape, precname):
    """The same fixed-tolerance backward check for UnetGenerator (networks.py:548-615), one UnetSkipConnectionBlock half at a time: every down unit
    (LeakyReLU -> Conv2d k4 s2 p1 -> norm; the outermost one without activation, the innermost one without norm -- it ends on a 1 x 1 map for
    unet_512 at 512 x 512 / unet_64 at 64 x 64) and every up unit (ReLU -> ConvTranspose2d k4 s2 p1 -> norm over the CONCATENATED [skip | up]
    tensor; the outermost one ends in Tanh) gets the fp32 teacher's input and upstream gradient and must reproduce its output, the gradient
    with respect to its whole input (both concat halves) and the parameter gradients of the conv / transposed conv / norm.  This pins the
    stride-2 k4 forward / data gradient / weight gradient, the 4-phase transposed convolution, the activation applied while the input is
    staged (in_act) and its mask in backward: 1e-3 for the strict policy."""
    import copy
    import torch.nn.functional as F
    nf, cin = 8, shape[1]
    sd = O.random_state_dict(arch, cin, 3, nf, norm, 'zero', 4, generator=torch.Generator().manual_seed(5))
    net = build(arch, cin, nf, norm, 'zero')
    net.load_state_dict(sd, strict=True)
    net.train()
    prec = E.Precision.get(precname)
    lv = net._layers()
    D = len(lv)
    tnet = copy.deepcopy(net).cpu().float().train()
    tchain = _unet_chain(tnet)
    chain = _unet_chain(net)

    def mods(blk, which):
        conv = blk.model[blk.pos['down' if which == 'down' else 'up']]
        key = 'downnorm' if which == 'down' else 'upnorm'
        return conv, (blk.model[blk.pos[key]] if key in blk.pos else None)

    def t_down(d, h):
        conv, nm = mods(tchain[d], 'down')
        z = conv(h if d == 0 else F.leaky_relu(h, 0.2))
        return nm(z) if nm is not None else z

    def t_up(d, v):
        conv, nm = mods(tchain[d], 'up')
        z = conv(F.relu(v))
        return torch.tanh(z) if d == 0 else nm(z)

    x = seeded_uniform(shape, 6)
    h = x.clone().requires_grad_(True)
    units, outs = [], []
    for d in range(D):
        hin = h
        h = t_down(d, hin)
        h.retain_grad()
        outs.append(h)
        units.append(('down', d, hin, h))
    u_in = h
    for d in range(D - 1, 0, -1):
        u = t_up(d, u_in)
        u.retain_grad()
        units.append(('up', d, u_in, u))
        u_in = torch.cat([outs[d - 1], u], 1)
        u_in.retain_grad()
    y = t_up(0, u_in)
    y.retain_grad()
    units.append(('up', 0, u_in, y))
    yo = O.run_generator(arch, {k: v.clone() for k, v in sd.items()}, x, norm, 'zero')
    assert rel(y, yo) < 1e-5                                   # the teacher IS the pinned oracle
    (y * torch.randn(y.shape, generator=torch.Generator().manual_seed(7))).sum().backward()
    tol = {'fp32': 1e-3, 'bf16': 6e-2}[precname]
    worst = {}
    for which, d, xin, yout in units:
        g = yout.grad.detach()
        tconv, tnorm = mods(tchain[d], which)
        tparams = list(tconv.parameters()) + (list(tnorm.parameters()) if tnorm is not None else [])
        xr = xin.detach().clone().requires_grad_(True)
        ref = torch.autograd.grad(t_down(d, xr) if which == 'down' else t_up(d, xr), [xr] + tparams, g)
        ref_dx, ref_dp = ref[0], ref[1:]
        conv_m, norm_m = mods(chain[d], which)
        params = list(conv_m.parameters()) + (list(norm_m.parameters()) if norm_m is not None else [])
        for p in params:
            p.grad = torch.zeros_like(p)
        tape = E.Tape()
        ctx = E.Ctx(prec, tape, training=True)
        xa = E.to_engine(xin.detach().to(DEV), prec)
        xa.needs_grad = True
        l = lv[d]
        if which == 'down':
            in_act = L.ACT_NONE if d == 0 else L.ACT_LRELU
            nl = l['downnorm'] if l['has_downnorm'] else None
            ya = E.norm_act(ctx, E.conv(ctx, xa, l['down'], in_act=in_act, stats=True), nl, L.ACT_NONE) if nl is not None else E.conv(ctx, xa, l['down'], in_act=in_act)
        elif d == 0:
            ya = E.conv(ctx, xa, l['up'], act=L.ACT_TANH, in_act=L.ACT_RELU)
        else:
            ya = E.norm_act(ctx, E.conv(ctx, xa, l['up'], in_act=L.ACT_RELU, stats=True), l['upnorm'], L.ACT_NONE)
        errs = {'y': rel(E.from_engine(ya), yout.detach())}
        ya.grad = E.to_engine(g.to(DEV), prec).t
        tape.backward()
        errs['dx'] = l2(E.from_engine(E.Act(xa.grad, xa.C)), ref_dx)
        scale = max(float(t.abs().max()) for t in ref_dp)
        errs['dparams'] = max(float((p.grad.cpu() - rp).abs().max()) / scale for p, rp in zip(params, ref_dp))
        for k, v in errs.items():
            worst[k] = max(worst.get(k, 0.0), v)
            assert v <= tol, (which, d, k, v, tuple(xin.shape))
    for k, v in worst.items():
        ERRLOG[f'teacher_forced/{arch}-{norm}/{precname}/{k}'] = v


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('norm', ['instance', 'batch'])
@pytest.mark.parametrize('arch', ['resnet_9blocks', 'n_layers'])
def test_teacher_forced_layer_gradients(arch, norm, precname):
    """Whole-backward parity at a FIXED tolerance, without the conditioning of a 9-block reverse pass: a plain-torch fp32 teacher (the same
    module tree, equal to the pinned oracle on the output) runs forward + backward once and keeps every unit's input and the gradient
    arriving at its output; each conv -> norm -> activation (+ residual) unit of the engine then gets the TEACHER's input and the TEACHER's
    upstream gradient and must reproduce that unit's dx, d(residual) and parameter gradients: 1e-3 for the strict policy."""
    import copy
    import torch.nn.functional as F
    is_d = arch == 'n_layers'
    nf, cin, shape = 8, (6 if is_d else 3), ((2, 6, 64, 80) if is_d else (2, 3, 64, 48))
    sd = O.random_state_dict(arch, cin, 3, nf, norm, 'zero', 4, generator=torch.Generator().manual_seed(5))
    net = build(arch, cin, nf, norm, 'zero')
    net.load_state_dict(sd, strict=True)
    net.train()
    prec = E.Precision.get(precname)
    make_units = _nlayer_units if is_d else _resnet_units
    units = make_units(net)
    # ---- teacher: the same torch modules on CPU in fp32 (ReLU(True) is in-place in the tree: use functional copies)
    tnet = copy.deepcopy(net).cpu().float().train()
    tunits = make_units(tnet)

    def run_unit(mods, x, res):
        h = x
        for mod in mods:
            h = F.relu(h) if isinstance(mod, torch.nn.ReLU) else (F.leaky_relu(h, 0.2) if isinstance(mod, torch.nn.LeakyReLU) else mod(h))
        return h + res if res is not None else h

    x = seeded_uniform(shape, 6)
    h = x.clone().requires_grad_(True)
    keep, skip = [], None
    for name, mods, has_res, _, _ in tunits:
        if name.endswith('.a'):
            skip = h
        xin = h
        h = run_unit(mods, xin, skip if has_res else None)
        h.retain_grad()
        keep.append((xin, skip if has_res else None, h))
    sdo = {k: v.clone() for k, v in sd.items()}
    yo = O.nlayer_discriminator(sdo, x, norm, 4) if is_d else O.run_generator(arch, sdo, x, norm, 'zero')
    assert rel(h, yo) < 1e-5                                   # the teacher IS the pinned oracle
    r = torch.randn(h.shape, generator=torch.Generator().manual_seed(7))
    (h * r).sum().backward()
    # bf16: one unit, no compounding -- but the unit stores its pre-activation y in bf16 (|error| ~ 4e-3 of a unit-variance value), which
    # flips the ReLU mask of the ~0.3 % of elements with |xhat| below that; every flipped element enters or leaves dn at full size:
    # |d dn| / |dn| ~ sqrt(0.003 / 0.5) = 8 %.  Measured 3e-2 ... 1.2e-1 (dx, dw of the units with a ReLU); the strict policy is at 5e-6.
    # Units WITHOUT a ReLU (the second half of every ResnetBlock, the tanh head) show the bf16 arithmetic itself: <= 6e-3.
    tol = {'fp32': 1e-3, 'bf16': 4e-1}[precname]
    tol_smooth = {'fp32': 1e-3, 'bf16': 3e-2}[precname]
    worst = {}
    for (name, mods, has_res, (conv, nl), act), (tname, tmods, _, _, _), (xin, res, yout) in zip(units, tunits, keep):
        g = yout.grad.detach()
        # reference gradients of this unit alone, from the teacher's input and upstream gradient
        xr = xin.detach().clone().requires_grad_(True)
        rr = res.detach().clone().requires_grad_(True) if res is not None else None
        tparams = [p for mod in tmods for p in mod.parameters()]
        ref = torch.autograd.grad(run_unit(tmods, xr, rr), [xr] + ([rr] if rr is not None else []) + tparams, g)
        ref_dx, ref_dres, ref_dp = ref[0], (ref[1] if rr is not None else None), ref[(2 if rr is not None else 1):]
        # engine: the same unit on the teacher's tensors
        params = [p for mod in mods for p in mod.parameters()]
        for p in params:
            p.grad = torch.zeros_like(p)
        tape = E.Tape()
        ctx = E.Ctx(prec, tape, training=True)
        xa = E.to_engine(xin.detach().to(DEV), prec)
        xa.needs_grad = True
        ra = None
        if res is not None:
            ra = E.to_engine(res.detach().to(DEV), prec)
            ra.needs_grad = True
        if nl is None:                      # conv with an epilogue activation only (generator head, discriminator first / last layer)
            ya = E.conv(ctx, xa, conv, act=act)
        else:
            ya = E.norm_act(ctx, E.conv(ctx, xa, conv, stats=nl is not None), nl, act, residual=ra)
        e_y = rel(E.from_engine(ya), yout.detach())
        ya.grad = E.to_engine(g.to(DEV), prec).t
        tape.backward()
        errs = {'y': e_y, 'dx': l2(E.from_engine(E.Act(xa.grad, xa.C)), ref_dx)}
        if ra is not None:
            errs['dres'] = l2(E.from_engine(E.Act(ra.grad, ra.C)), ref_dres)
        scale = max(float(t.abs().max()) for t in ref_dp)
        for p, rp in zip(params, ref_dp):
            # a conv bias in front of InstanceNorm has an exactly-zero true gradient: judge every parameter against the unit's largest
            errs.setdefault('dparams', 0.0)
            errs['dparams'] = max(errs['dparams'], float((p.grad.cpu() - rp).abs().max()) / scale)
            if os.environ.get('DL_TEST_VERBOSE'):
                print(name, tuple(p.shape), 'max|ref| %.3e  max|err| %.3e  unit scale %.3e' % (float(rp.abs().max()), float((p.grad.cpu() - rp).abs().max()), scale))
        for k, v in errs.items():
            kk = k if act in (L.ACT_RELU, L.ACT_LRELU) else k + '_units_without_relu'
            worst[kk] = max(worst.get(kk, 0.0), v)
            assert v <= (tol if act in (L.ACT_RELU, L.ACT_LRELU) else tol_smooth), (name, k, v)
    for k, v in worst.items():
        ERRLOG[f'teacher_forced/{arch}-{norm}/{precname}/{k}'] = v
