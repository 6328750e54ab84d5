"""Generate the committed golden fixtures from the REFERENCE implementation.

Runs only in the build container (needs /root/reference).  Usage:
    python tests/golden/make_golden.py
Writes tests/golden/*.npz -- data only: seeds, small inputs, expected outputs.  The reference is imported through
tests/golden/_ref_import.py (third-party stubs + VGG loss zeroed).  Dropout is off in every vector
(no_dropout=True): dropout RNG cannot be matched across implementations.

To keep the fixtures small, network weights are not stored: they are drawn from a seeded torch.Generator by
oracle.deepliif_oracle.random_state_dict (same N(0,.02)/N(1,.02)/0 distributions as init_weights) and loaded into the
reference modules with load_state_dict(strict=True); a test regenerates them from the seed and checks the stored
weight checksum first.  Large gradients are stored as (l2 norm, 4 seeded random projections), see golden_util.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402
from golden_util import digest, seeded_uniform  # noqa: E402
from oracle import deepliif_oracle as O  # noqa: E402

models, networks = _ref_import.import_reference()
from deepliif.options import Options  # noqa: E402
from deepliif.util import disable_batchnorm_tracking_stats  # noqa: E402

torch.set_num_threads(8)


def load_seeded(net, arch, input_nc, nf, norm, pad, seed, n_layers=4):
    sd = O.random_state_dict(arch, input_nc, 3, nf, norm, pad, n_layers, generator=torch.Generator().manual_seed(seed))
    net.load_state_dict(sd, strict=True)       # also proves oracle.layer_table == the reference's key set / shapes
    return sd


def net_case(tag, net, meta, x_seed, x_shape, out):
    """forward, d/dx and d/dw of L = sum(y * r) for one reference network (training mode), then the eval forward."""
    arch, input_nc, nf, norm, pad, w_seed = meta
    sd = load_seeded(net, arch, input_nc, nf, norm, pad, w_seed)
    out[f'{tag}/meta'] = np.array([arch, str(input_nc), str(nf), norm, pad, str(w_seed), str(x_seed), str(tuple(x_shape))])
    out[f'{tag}/w_digest'] = digest(torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()]))
    net.train()
    x = seeded_uniform(x_shape, x_seed).requires_grad_(True)
    y = net(x)
    r = torch.randn(y.shape, generator=torch.Generator().manual_seed(99))
    (y * r).sum().backward()
    out[f'{tag}/y'] = y.detach().numpy()
    out[f'{tag}/dx'] = x.grad.numpy()
    for k, p in net.named_parameters():
        out[f'{tag}/dw/{k}'] = p.grad.numpy() if p.numel() <= 2048 else digest(p.grad)
    for k, v in net.state_dict().items():      # BatchNorm running stats after one training-mode forward
        if 'running_' in k:
            out[f'{tag}/sd_after/{k}'] = v.numpy()
    # eval-mode forward the way the reference serves it: BN on batch stats (util/__init__.py:743-755)
    net.eval()
    disable_batchnorm_tracking_stats(net)
    with torch.no_grad():
        out[f'{tag}/y_eval'] = net(x.detach()).numpy()


def make_nets_small():
    out = {}
    seed = 100
    for norm in ('batch', 'instance'):
        for pad in ('zero', 'reflect'):
            # the named 9-block generator once; 2-block variants (define_G parses any resnet_<n>blocks) elsewhere
            arch = 'resnet_9blocks' if (norm, pad) == ('batch', 'zero') else 'resnet_2blocks'
            net = networks.define_G(3, 3, 8, arch, norm, False, 'normal', 0.02, [], pad)
            net_case(f'{arch}_{norm}_{pad}', net, (arch, 3, 8, norm, pad, seed), seed + 1, (2, 3, 32, 32), out)
            seed += 2
        net = networks.define_G(3, 3, 8, 'unet_32', norm, False, 'normal', 0.02, [])
        net_case(f'unet_32_{norm}', net, ('unet_32', 3, 8, norm, 'zero', seed), seed + 1, (2, 3, 32, 32), out)
        seed += 2
        net = networks.define_D(6, 8, 'n_layers', 4, norm, 'normal', 0.02, [])
        net_case(f'n_layers_{norm}', net, ('n_layers', 6, 8, norm, 'zero', seed), seed + 1, (2, 6, 64, 64), out)
        seed += 2
    # DeepLIIFExt channel counts: GS in 9 ch, DS in 12 ch (DeepLIIFExt_model.py:85,97)
    net = networks.define_G(9, 3, 8, 'unet_32', 'batch', False, 'normal', 0.02, [])
    net_case('unet_32_in9_batch', net, ('unet_32', 9, 8, 'batch', 'zero', seed), seed + 1, (1, 9, 32, 32), out)
    seed += 2
    net = networks.define_D(12, 8, 'n_layers', 4, 'batch', 'normal', 0.02, [])
    net_case('n_layers_in12_batch', net, ('n_layers', 12, 8, 'batch', 'zero', seed), seed + 1, (1, 12, 64, 64), out)
    np.savez_compressed(os.path.join(HERE, 'nets_small.npz'), **out)
    print('nets_small.npz', len(out), 'arrays')


def make_unet512():
    """Full-depth unet_512 (9 downs, needs 512x512) with ngf=8; eval forward; output stored strided."""
    out = {}
    net = networks.define_G(3, 3, 8, 'unet_512', 'batch', False, 'normal', 0.02, [])
    sd = load_seeded(net, 'unet_512', 3, 8, 'batch', 'zero', 11)
    out['meta'] = np.array(['unet_512', '3', '8', 'batch', 'zero', '11', '12', str((1, 3, 512, 512))])
    out['w_digest'] = digest(torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()]))
    x = seeded_uniform((1, 3, 512, 512), 12)
    net.eval()
    disable_batchnorm_tracking_stats(net)
    with torch.no_grad():
        y = net(x)
    out['y_strided'] = y[:, :, ::8, ::8].numpy()
    out['y_digest'] = digest(y)
    np.savez_compressed(os.path.join(HERE, 'unet512_ngf8.npz'), **out)
    print('unet512_ngf8.npz')


def make_seeded_init():
    """Per-key checksums of full-size networks built by the reference under torch.manual_seed(0): pins the RNG
    consumption order of define_G / define_D / init_weights (networks.py:84-139) for seeded-parity runs."""
    out = {}
    cases = [('resnet_9blocks_batch', lambda: networks.define_G(3, 3, 64, 'resnet_9blocks', 'batch', False, 'normal', 0.02, [], 'zero')),
             ('resnet_9blocks_instance', lambda: networks.define_G(3, 3, 64, 'resnet_9blocks', 'instance', False, 'normal', 0.02, [], 'zero')),
             ('unet_512_batch', lambda: networks.define_G(3, 3, 64, 'unet_512', 'batch', False, 'normal', 0.02, [])),
             ('n_layers_batch', lambda: networks.define_D(6, 64, 'n_layers', 4, 'batch', 'normal', 0.02, []))]
    for tag, fn in cases:
        torch.manual_seed(0)
        net = fn()
        sd = net.state_dict()
        out[f'{tag}/keys'] = np.array(list(sd.keys()))
        out[f'{tag}/shapes'] = np.array([str(tuple(v.shape)) for v in sd.values()])
        out[f'{tag}/sums'] = np.array([v.double().sum().item() for v in sd.values()])
        out[f'{tag}/abs_sums'] = np.array([v.double().abs().sum().item() for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, 'seeded_init.npz'), **out)
    print('seeded_init.npz')


def base_params(modalities_no, seg_gen, norm, padding, net_gs, nf):
    n = modalities_no + 1
    w = [0.25, 0.15, 0.25, 0.1, 0.25] if modalities_no == 4 else [1.0 / n] * n
    lw = [0.2] * 5 if modalities_no == 4 else [1.0 / n] * n
    return dict(
        model='DeepLIIF', name='golden', checkpoints_dir='/tmp/golden_ckpt', gpu_ids=[], phase='train', preprocess='none',
        remote_transfer_cmd=None, continue_train=False, modalities_no=modalities_no, seg_gen=seg_gen,
        modalities_names=[], input_nc=3, input_no=1, output_nc=3, ngf=nf, ndf=nf, net_g='resnet_9blocks',
        net_gs=net_gs, net_d='n_layers', norm=norm, no_dropout=True, init_type='normal', init_gain=0.02,
        padding=padding, upsample='convtranspose', gan_mode='vanilla', gan_mode_s='lsgan', optimizer='adam',
        lr_g=2e-4, lr_d=2e-4, beta1=0.5, lr_policy='linear', n_epochs=100, n_epochs_decay=100, epoch_count=0,
        seg_weights=w, loss_G_weights=lw, loss_D_weights=lw, verbose=False, epoch='latest', load_iter=0)


def seed_model_nets(model, opt, nf, base_seed):
    """Load seeded weights into every network of a reference DeepLIIFModel; returns {name: seed}."""
    seeds = {}
    for j, n in enumerate(model.model_names):
        net = getattr(model, 'net' + n)
        if n.startswith('D'):
            arch, pad = 'n_layers', 'zero'
            cin = 6
        elif n in model.model_names_g:
            arch, pad = opt.netG[model.model_names_g.index(n)], opt.padding
            cin = 3
        else:
            arch, pad = opt.net_gs[model.model_names_gs.index(n)], 'reflect'     # define_G default (DeepLIIF_model.py:98-99)
            cin = 3
        load_seeded(net, arch, cin, nf, opt.norm, pad, base_seed + j)
        seeds[n] = base_seed + j
    return seeds


def make_step(tag, modalities_no, seg_gen, norm, padding, net_gs, size, nf=8, batch=2, steps=2):
    out = {}
    opt = Options(d_params=base_params(modalities_no, seg_gen, norm, padding, net_gs, nf))
    model = models.create_model(opt)
    model.setup(opt)
    seeds = seed_model_nets(model, opt, nf, 500)
    nB = modalities_no + (1 if seg_gen else 0)
    A = seeded_uniform((batch, 3, size, size), 22)
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(nB)]
    out['meta'] = np.array([str(modalities_no), str(seg_gen), norm, padding, net_gs, str(size), str(nf), str(batch), str(steps)])
    out['model_names'] = np.array(model.model_names)
    out['net_seeds'] = np.array([seeds[n] for n in model.model_names])
    out['loss_names'] = np.array(model.loss_names)
    out['mod_id_seg'] = np.array(str(model.mod_id_seg))
    for s in range(steps):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        losses = model.get_current_losses()
        out[f'step{s}/losses'] = np.array([losses[k] for k in model.loss_names], dtype=np.float64)
        for i in range(modalities_no):
            out[f'step{s}/fake_B_{i + 1}'] = getattr(model, f'fake_B_{i + 1}').detach().numpy()[:, :, ::2, ::2]
            out[f'step{s}/fake_B_{i + 1}_digest'] = digest(getattr(model, f'fake_B_{i + 1}').detach())
        if seg_gen:
            fs = getattr(model, f'fake_B_{model.mod_id_seg}').detach()
            out[f'step{s}/fake_B_S'] = fs.numpy()[:, :, ::2, ::2]
            out[f'step{s}/fake_B_S_digest'] = digest(fs)
        for n in model.model_names:
            sd = getattr(model, 'net' + n).state_dict()
            flat = torch.cat([v.reshape(-1).float() for k, v in sd.items() if v.is_floating_point()])
            out[f'step{s}/w_digest/{n}'] = digest(flat)
    np.savez_compressed(os.path.join(HERE, f'step_{tag}.npz'), **out)
    print(f'step_{tag}.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')


def make_step_sdg(tag, modalities_no, input_no, norm, size, nf=8, batch=1, steps=2):
    """SDGModel trajectory (SDG_model.py): input_no modalities concatenated on the channel axis, one translation generator +
    discriminator per output modality, GAN + SmoothL1 (+ VGG, zeroed by _ref_import like everywhere else)."""
    out = {}
    p = base_params(modalities_no, False, norm, 'zero', 'unet_64', nf)
    p.update(model='SDG', input_no=input_no, seg_weights=[1.0 / modalities_no] * modalities_no, lambda_feat=100.0,
             loss_G_weights=[1.0 / modalities_no] * modalities_no, loss_D_weights=[1.0 / modalities_no] * modalities_no)
    opt = Options(d_params=p)
    from deepliif.models.SDG_model import SDGModel
    model = SDGModel(opt)
    model.setup(opt)
    seeds = {}
    for j, n in enumerate(model.model_names):
        kind, idx = n.split('_')
        net = getattr(model, 'net' + kind)[int(idx) - 1]
        arch, cin, pad = {'G': ('resnet_9blocks', 3 * input_no, 'zero'), 'D': ('n_layers', 3 * input_no + 3, 'zero')}[kind]
        load_seeded(net, arch, cin, nf, norm, pad, 1300 + j)
        seeds[n] = 1300 + j
    A = [seeded_uniform((batch, 3, size, size), 22 + 100 * k) for k in range(input_no)]
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(modalities_no)]
    out['meta'] = np.array([str(modalities_no), str(input_no), norm, str(size), str(nf), str(batch), str(steps)])
    out['model_names'] = np.array(model.model_names)
    out['net_seeds'] = np.array([seeds[n] for n in model.model_names])
    out['loss_names'] = np.array(model.loss_names)
    for s in range(steps):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        losses = model.get_current_losses()
        out[f'step{s}/losses'] = np.array([losses[k] for k in model.loss_names], dtype=np.float64)
        for i in range(modalities_no):
            out[f'step{s}/fake_B_{i + 1}'] = model.fake_B[i].detach().numpy()[:, :, ::2, ::2]
        for n in model.model_names:
            kind, idx = n.split('_')
            sd = getattr(model, 'net' + kind)[int(idx) - 1].state_dict()
            out[f'step{s}/w_digest/{n}'] = digest(torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()]))
    np.savez_compressed(os.path.join(HERE, f'step_sdg_{tag}.npz'), **out)
    print(f'step_sdg_{tag}.npz', out['loss_names'])


def make_step_ext(tag, modalities_no, norm, size, nf=8, batch=1, steps=2):
    """DeepLIIFExtModel trajectory (DeepLIIFExt_model.py): list-valued nets, GS input 9 ch, DS input 12 ch."""
    out = {}
    p = base_params(modalities_no, True, norm, 'zero', 'unet_64', nf)
    p.update(model='DeepLIIFExt', net_ds='n_layers', seg_weights=[1.0 / modalities_no] * modalities_no,
             loss_G_weights=[1.0 / modalities_no] * modalities_no, loss_D_weights=[1.0 / modalities_no] * modalities_no)
    opt = Options(d_params=p)
    from deepliif.models.DeepLIIFExt_model import DeepLIIFExtModel
    model = DeepLIIFExtModel(opt)
    model.setup(opt)
    seeds = {}
    for j, n in enumerate(model.model_names):
        kind, idx = n.split('_')
        net = getattr(model, 'net' + kind)[int(idx) - 1]
        arch, cin, pad = {'G': ('resnet_9blocks', 3, 'zero'), 'GS': ('unet_64', 9, 'reflect'), 'D': ('n_layers', 6, 'zero'),
                          'DS': ('n_layers', 12, 'zero')}[kind]
        load_seeded(net, arch, cin, nf, norm, pad, 900 + j)
        seeds[n] = 900 + j
    A = seeded_uniform((batch, 3, size, size), 22)
    B = [seeded_uniform((batch, 3, size, size), 23 + i) for i in range(modalities_no)]
    BS = [seeded_uniform((batch, 3, size, size), 43 + i) for i in range(modalities_no)]
    out['meta'] = np.array([str(modalities_no), norm, str(size), str(nf), str(batch), str(steps)])
    out['model_names'] = np.array(model.model_names)
    out['net_seeds'] = np.array([seeds[n] for n in model.model_names])
    out['loss_names'] = np.array(model.loss_names)
    for s in range(steps):
        model.set_input({'A': A, 'B': B, 'BS': BS, 'A_paths': ['x']})
        model.optimize_parameters()
        losses = model.get_current_losses()
        out[f'step{s}/losses'] = np.array([losses[k] for k in model.loss_names], dtype=np.float64)
        for i in range(modalities_no):
            out[f'step{s}/fake_B_{i + 1}'] = model.fake_B[i].detach().numpy()[:, :, ::2, ::2]
            out[f'step{s}/fake_BS_{i + 1}'] = model.fake_BS[i].detach().numpy()[:, :, ::2, ::2]
        for n in model.model_names:
            kind, idx = n.split('_')
            sd = getattr(model, 'net' + kind)[int(idx) - 1].state_dict()
            out[f'step{s}/w_digest/{n}'] = digest(torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()]))
    np.savez_compressed(os.path.join(HERE, f'step_ext_{tag}.npz'), **out)
    print(f'step_ext_{tag}.npz')


def make_inference():
    """run_dask(tensor, nets, opt, use_dask=False, output_tensor=True): the 2-stage generator DAG + weighted seg sum
    (deepliif/models/__init__.py:293-361), one tile per call."""
    out = {}
    p = base_params(4, True, 'batch', 'zero', 'unet_64', 8)
    opt = Options(d_params=p)
    model = models.create_model(opt)
    seeds = seed_model_nets(model, opt, 8, 700)
    nets = {}
    for n in model.model_names:
        if n.startswith('G'):
            net = getattr(model, 'net' + n)
            net.eval()
            disable_batchnorm_tracking_stats(net)
            nets[n] = net
    out['net_names'] = np.array(list(nets.keys()))
    out['net_seeds'] = np.array([seeds[n] for n in nets])
    opt.mod_id_seg = model.mod_id_seg
    opt.input_id = int(model.input_id)
    opt.modalities_names = ['IHC', 'Hema', 'DAPI', 'Lap2', 'Marker']
    tiles = seeded_uniform((3, 3, 64, 64), 32)
    seg_weights = [0.25, 0.15, 0.25, 0.1, 0.25]
    out['seg_weights'] = np.array(seg_weights)
    for t in range(tiles.shape[0]):
        res = models.run_dask(tiles[t:t + 1], nets=nets, opt=opt, seg_weights=seg_weights, use_dask=False, output_tensor=True)
        for k, v in res.items():
            out[f'tile{t}/{k}'] = v.numpy()
    out['keys'] = np.array(list(res.keys()))
    np.savez_compressed(os.path.join(HERE, 'inference_small.npz'), **out)
    print('inference_small.npz', list(res.keys()))


if __name__ == '__main__':
    os.makedirs('/tmp/golden_ckpt/golden', exist_ok=True)
    if sys.argv[1:] == ['sdg']:             # regenerate only the SDG trajectory (the other fixtures are left as committed)
        make_step_sdg('m2_in2_instance', 2, 2, 'instance', 64)
        sys.exit(0)
    make_nets_small()
    make_unet512()
    make_seeded_init()
    make_step('m1_noseg_batch', 1, False, 'batch', 'zero', 'unet_64', 64, batch=1)
    make_step('m5_noseg_instance', 5, False, 'instance', 'zero', 'unet_64', 64, batch=1)
    make_step('m4_seg_batch', 4, True, 'batch', 'zero', 'unet_64', 64, batch=2)
    make_step('m2_seg_instance_reflect', 2, True, 'instance', 'reflect', 'unet_64', 64, batch=1)
    make_inference()
    make_step_ext('m2_batch', 2, 'batch', 64)
    make_step_sdg('m2_in2_instance', 2, 2, 'instance', 64)
