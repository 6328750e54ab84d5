"""Golden fixtures for the remaining model zoo, generated from the REFERENCE classes:
    step_kd_m2.npz        deepliif/models/DeepLIIFKD_model.py: 2 optimize_parameters() steps of a student distilled from the 'dl_m2' teacher directory
    step_cyclegan_m2.npz  deepliif/models/CycleGAN_model.py: 2 steps with pool_size = 2 and batch 2, so that the image pools fill in step 0 and
                          draw from Python's `random` stream in step 1 (whose discriminator losses then depend on the draws)
    step_options_m1.npz   DeepLIIF_model.py with the CLI-exposed non-default options --upsample resize_conv --net-d pixel --gan-mode wgangp
Runs only in the build container (needs /root/reference):  python tests/golden/make_golden_zoo.py [kd] [cyclegan] [options]
Data only (seeds, sub-sampled expected images, losses, weight digests); network weights are regenerated from seeds (make_golden.py).  The teacher
directory (tests/seam_util.build_kd_teacher_dir) is the 'dl_m2' training options file of the seam fixtures + seeded weights at ngf = 64: the
reference's test-mode Options force ngf = 64 (options/__init__.py:75) and DeepLIIFKD_model.py:107-112 gives the caller no way to override it.  The VGG terms are zeroed as in every other fixture (_ref_import.py)."""
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402
from golden_util import digest, seeded_uniform  # noqa: E402
from oracle import deepliif_oracle as O  # noqa: E402

models, networks = _ref_import.import_reference()
from deepliif.options import Options  # noqa: E402
import make_golden as MG  # noqa: E402  (base_params, load_seeded)
import seam_util  # noqa: E402

torch.set_num_threads(8)


def weight_digests(model, out, s):
    for n in model.model_names:
        net = getattr(model, 'net' + n) if '_' not in n else getattr(model, 'net' + n.split('_')[0])[int(n.split('_')[1]) - 1]
        sd = net.state_dict()
        out[f'step{s}/w_digest/{n}'] = digest(torch.cat([v.reshape(-1).float() for k, v in sd.items() if v.is_floating_point()]))


def make_kd(tag='kd_m2', size=64, nf=8, batch=2, steps=2):
    out = {}
    tmp = tempfile.mkdtemp()
    tdir = seam_util.build_kd_teacher_dir(tmp)
    p = MG.base_params(2, True, 'batch', 'zero', 'unet_64', nf)
    p.update(model='DeepLIIFKD', model_dir_teacher=tdir)
    opt = Options(d_params=p)
    model = models.create_model(opt)
    model.setup(opt)
    seeds = MG.seed_model_nets(model, opt, nf, 700)
    A = seeded_uniform((batch, 3, size, size), 42)
    B = [seeded_uniform((batch, 3, size, size), 43 + i) for i in range(3)]
    S = str(model.mod_id_seg)
    out['meta'] = np.array(['2', 'batch', 'zero', 'unet_64', str(size), str(nf), str(batch), str(steps), 'dl_m2@ngf64'])
    out['teacher_digest'] = np.stack([digest(torch.cat([v.reshape(-1).float() for v in torch.load(os.path.join(tdir, f'latest_net_{n}.pth')).values() if v.is_floating_point()])) for n, _, _ in seam_util.KD_TEACHER_NETS])
    out['model_names'] = np.array(model.model_names)
    out['net_seeds'] = np.array([seeds[n] for n in model.model_names])
    out['loss_names'] = np.array(model.loss_names)
    out['mod_id_seg'] = np.array(S)
    out['teacher_mapping'] = np.array([f'{k}={v}' for k, v in model.d_mapping_model_name.items()])
    extra = [f'G_KLDiv_{S}0']                                   # computed and part of loss_G, but not in loss_names
    out['extra_loss_names'] = np.array(extra)
    for s in range(steps):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        losses = model.get_current_losses()
        out[f'step{s}/losses'] = np.array([losses[k] for k in model.loss_names], dtype=np.float64)
        out[f'step{s}/extra_losses'] = np.array([float(getattr(model, 'loss_' + k)) for k in extra], dtype=np.float64)
        for i in range(2):
            out[f'step{s}/fake_B_{i + 1}'] = getattr(model, f'fake_B_{i + 1}').detach().numpy()[:, :, ::2, ::2]
        out[f'step{s}/fake_B_S'] = getattr(model, f'fake_B_{S}').detach().numpy()[:, :, ::2, ::2]
        if s == 0:
            for i in range(2):
                out[f'teacher/fake_B_{i + 1}'] = getattr(model, f'fake_B_{i + 1}_teacher').detach().numpy()[:, :, ::2, ::2]
            for i in range(3):
                out[f'teacher/fake_B_S_{i}'] = getattr(model, f'fake_B_{S}_{i}_teacher').detach().numpy()[:, :, ::2, ::2]
            out['teacher/fake_B_S'] = getattr(model, f'fake_B_{S}_teacher').detach().numpy()[:, :, ::2, ::2]
        weight_digests(model, out, s)
    np.savez_compressed(os.path.join(HERE, f'step_{tag}.npz'), **out)
    print(f'step_{tag}.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw', dict(zip(model.loss_names, out['step0/losses'])))


def make_cyclegan(tag='cyclegan_m2', size=64, nf=8, batch=2, steps=2, pool_size=2):
    out = {}
    p = MG.base_params(2, False, 'batch', 'zero', 'unet_64', nf)
    p.update(model='CycleGAN', gan_mode='lsgan', pool_size=pool_size, BtoA=False, label_smoothing=0, lambda_identity=0)
    opt = Options(d_params=p)
    model = models.create_model(opt)
    model.setup(opt)
    seeds = {}
    for j, n in enumerate(model.model_names):
        kind, idx = n.split('_')
        net = getattr(model, 'net' + kind)[int(idx) - 1]
        if kind.startswith('D'):
            MG.load_seeded(net, 'n_layers', 3, nf, 'batch', 'zero', 900 + j)
        else:
            MG.load_seeded(net, 'resnet_9blocks', 3, nf, 'batch', 'zero', 900 + j)
        seeds[n] = 900 + j
    A = seeded_uniform((batch, 3, size, size), 52)
    Bs = [seeded_uniform((batch, 3, size, size), 53 + i) for i in range(2)]
    out['meta'] = np.array(['2', 'batch', 'zero', 'resnet_9blocks', str(size), str(nf), str(batch), str(steps), str(pool_size), 'lsgan', '1234'])
    out['model_names'] = np.array(model.model_names)
    out['net_seeds'] = np.array([seeds[n] for n in model.model_names])
    out['loss_names'] = np.array(model.loss_names)
    random.seed(1234)                       # util/image_pool.py draws from Python's global `random`
    for s in range(steps):
        model.set_input({'A': A, 'Bs': Bs, 'A_paths': ['x']})
        model.optimize_parameters()
        losses = model.get_current_losses()
        out[f'step{s}/losses'] = np.array([losses[k] for k in model.loss_names], dtype=np.float64)
        for fam in ('fake_Bs', 'rec_As', 'fake_As', 'rec_Bs'):
            for i in range(2):
                out[f'step{s}/{fam}_{i + 1}'] = getattr(model, fam)[i].detach().numpy()[:, :, ::2, ::2]
        weight_digests(model, out, s)
    out['random_after'] = np.array([random.random()])          # the stream position after the run: the pools consumed the same draws
    np.savez_compressed(os.path.join(HERE, f'step_{tag}.npz'), **out)
    print(f'step_{tag}.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw', dict(zip(model.loss_names, out['step0/losses'])))


def make_options(tag='options_m1', size=64, nf=8, batch=2, steps=2):
    """DeepLIIF with the CLI-exposed non-default options --upsample resize_conv, --net-d pixel, --gan-mode wgangp (cli.py:103, 176-182)"""
    out = {}
    p = MG.base_params(1, False, 'batch', 'zero', 'unet_64', nf)
    p.update(upsample='resize_conv', net_d='pixel', gan_mode='wgangp')
    opt = Options(d_params=p)
    model = models.create_model(opt)
    model.setup(opt)
    assert type(model.netD1).__name__ == 'PixelDiscriminator' and model.criterionGAN_mod.gan_mode == 'wgangp'
    seeds = {}
    for j, n in enumerate(model.model_names):
        net = getattr(model, 'net' + n)
        arch, cin = ('pixel', 6) if n.startswith('D') else ('resnet_9blocks:resize_conv', 3)
        MG.load_seeded(net, arch, cin, nf, 'batch', 'zero', 600 + j)
        seeds[n] = 600 + j
    A = seeded_uniform((batch, 3, size, size), 62)
    B = [seeded_uniform((batch, 3, size, size), 63)]
    out['meta'] = np.array(['1', 'batch', 'zero', 'resnet_9blocks', str(size), str(nf), str(batch), str(steps), 'resize_conv', 'pixel', 'wgangp'])
    out['model_names'] = np.array(model.model_names)
    out['net_seeds'] = np.array([seeds[n] for n in model.model_names])
    out['loss_names'] = np.array(model.loss_names)
    for s in range(steps):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        losses = model.get_current_losses()
        out[f'step{s}/losses'] = np.array([losses[k] for k in model.loss_names], dtype=np.float64)
        out[f'step{s}/fake_B_1'] = model.fake_B_1.detach().numpy()[:, :, ::2, ::2]
        weight_digests(model, out, s)
    np.savez_compressed(os.path.join(HERE, f'step_{tag}.npz'), **out)
    print(f'step_{tag}.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw', dict(zip(model.loss_names, out['step0/losses'])))


if __name__ == '__main__':
    which = sys.argv[1:] or ['kd', 'cyclegan', 'options']
    if 'options' in which:
        make_options()
    if 'kd' in which:
        make_kd()
    if 'cyclegan' in which:
        make_cyclegan()
