"""Golden vectors for the checkpoint-directory / inference seam, produced by the REFERENCE (build container only).

    python tests/golden/make_golden_seam.py      -> tests/golden/seam_cases.npz + tests/golden/seam_train_opt_*.txt

What the reference does here (all on CPU, eager mode):
  * Options(d_params) -> print_options(save=True): the 'train_opt.txt' sidecar a training run leaves next to its checkpoints
    (deepliif/options/__init__.py:198-217) -- committed as TEXT (option values, i.e. data);
  * BaseModel.save_networks('latest') (base_model.py:190-212): the key list / shapes of every '<epoch>_net_<name>.pth';
  * Options(path_file=..., mode='test') + init_nets(dir, eager_mode=True, opt) (models/__init__.py:158-219) + run_dask(PIL) and
    inference(PIL, tile_size, overlap, ...) (:258-579) -> uint8 result images;
  * get_scheduler (networks.py:55-81) learning-rate sequences for the four policies.
Weights are not stored: the test rebuilds the checkpoint files from the same seeds (oracle.random_state_dict, checked against the
stored digests), exactly like the other fixtures.

torchvision is not installed here.  The reference's transform() (deepliif/data/__init__.py:133-138) needs four of its classes; this
script installs minimal stand-ins for Compose / Lambda / ToTensor / Normalize that restate their documented behaviour (uint8 HWC ->
float CHW / 255; (x - mean) / std).  Everything downstream of that (resize, tiling, networks, tensor2im, stitching, naming) is the
reference's own code.
"""
import os
import shutil
import sys

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402
from golden_util import digest, synth_image  # noqa: E402
from oracle import deepliif_oracle as O  # noqa: E402

models, networks = _ref_import.import_reference()


class _Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class _Lambda:
    def __init__(self, f):
        self.f = f

    def __call__(self, x):
        return self.f(x)


class _ToTensor:
    def __call__(self, img):
        a = np.asarray(img.convert('RGB') if img.mode != 'RGB' else img)
        return torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).float().div(255)


class _Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


tv = sys.modules['torchvision.transforms']
tv.Compose, tv.Lambda, tv.ToTensor, tv.Normalize = _Compose, _Lambda, _ToTensor, _Normalize
import deepliif.data as _ref_data  # noqa: E402
_ref_data.transforms = tv          # the package bound `torchvision.transforms` through the parent stub's attribute fallback at import time

from deepliif.options import Options, print_options  # noqa: E402

torch.set_num_threads(8)
CKPT = '/tmp/golden_seam_ckpt'


def d_params(model, name, M, seg_gen, net_gs, nf, input_no=1):
    n = M + 1
    return dict(
        model=model, name=name, checkpoints_dir=CKPT, gpu_ids=[], phase='train', preprocess='none', remote_transfer_cmd=None,
        continue_train=False, modalities_no=M, seg_gen=seg_gen, seg_no=(1 if seg_gen else 0) if model == 'DeepLIIF' else (M if seg_gen else 0),
        modalities_names=[], input_nc=3, input_no=input_no, output_nc=3, ngf=nf, ndf=nf,
        net_g='resnet_9blocks', net_gs=net_gs, net_d='n_layers', net_ds='n_layers', norm='batch', no_dropout=True, init_type='normal', init_gain=0.02,
        padding='zero', upsample='convtranspose', gan_mode='vanilla', gan_mode_s='lsgan', optimizer='adam', lr_g=2e-4, lr_d=2e-4, beta1=0.5,
        lr_policy='linear', n_epochs=100, n_epochs_decay=100, epoch_count=0, scale_size=64,
        seg_weights=[1.0 / n] * n if model == 'DeepLIIF' else [1.0 / M] * M, loss_G_weights=[1.0 / n] * n if model == 'DeepLIIF' else [1.0 / M] * M,
        loss_D_weights=[1.0 / n] * n if model == 'DeepLIIF' else [1.0 / M] * M, verbose=False, epoch='latest', load_iter=0)


def net_of(model, name):
    if '_' in name:
        kind, idx = name.split('_')
        return getattr(model, 'net' + kind)[int(idx) - 1]
    return getattr(model, 'net' + name)


def arch_of(model, opt, name):
    """(arch, input channels, padding) of network `name` -- mirrors how each reference model class calls define_G / define_D"""
    if opt.model == 'DeepLIIF':
        if name.startswith('D'):
            return 'n_layers', 6, 'zero'
        if name in model.model_names_g:
            return opt.netG[model.model_names_g.index(name)], 3, opt.padding
        return opt.net_gs[model.model_names_gs.index(name)], 3, 'reflect'
    kind = name.split('_')[0]
    if kind == 'G':
        return (opt.net_g if isinstance(opt.net_g, str) else opt.net_g[0]), 3 * (opt.input_no if opt.model == 'SDG' else 1), opt.padding
    if kind == 'GS':
        return (opt.net_gs if isinstance(opt.net_gs, str) else opt.net_gs[0]), 9, 'reflect'
    if kind == 'D':
        return 'n_layers', 3 * (opt.input_no if opt.model == 'SDG' else 1) + 3, 'zero'
    return 'n_layers', 12, 'zero'


def make_checkpoint(out, tag, model_name, M, seg_gen, net_gs, nf, seed0, input_no=1):
    """train-mode reference model with seeded weights -> save_networks('latest') + train_opt.txt in CKPT/<tag>"""
    shutil.rmtree(os.path.join(CKPT, tag), ignore_errors=True)
    os.makedirs(os.path.join(CKPT, tag), exist_ok=True)
    opt = Options(d_params=d_params(model_name, tag, M, seg_gen, net_gs, nf, input_no))
    model = models.create_model(opt)
    model.setup(opt)
    names, seeds = [], []
    for j, n in enumerate(model.model_names):
        arch, cin, pad = arch_of(model, opt, n)
        sd = O.random_state_dict(arch, cin, 3, nf, 'batch', pad, 4, generator=torch.Generator().manual_seed(seed0 + j))
        net_of(model, n).load_state_dict(sd, strict=True)
        names.append(n)
        seeds.append(seed0 + j)
    model.save_networks('latest')
    print_options(opt, save=True)                       # -> CKPT/<tag>/train_opt.txt
    shutil.copy(os.path.join(CKPT, tag, 'train_opt.txt'), os.path.join(HERE, f'seam_train_opt_{tag}.txt'))
    out[f'{tag}/model_names'] = np.array(names)
    out[f'{tag}/net_seeds'] = np.array(seeds)
    out[f'{tag}/net_arch'] = np.array(['|'.join(str(v) for v in arch_of(model, opt, n)) for n in names])
    for n in names:
        sd = torch.load(os.path.join(CKPT, tag, f'latest_net_{n}.pth'), map_location='cpu')
        out[f'{tag}/sd_keys/{n}'] = np.array(list(sd.keys()))
        out[f'{tag}/sd_shapes/{n}'] = np.array(['x'.join(str(d) for d in v.shape) for v in sd.values()])
        out[f'{tag}/sd_digest/{n}'] = digest(torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()]))
    out[f'{tag}/files'] = np.array(sorted(os.listdir(os.path.join(CKPT, tag))))
    return opt


def pil_u8(d):
    return {k: np.asarray(v) for k, v in d.items()}


def main():
    out = {}
    # ---- DeepLIIF, 2 modalities + seg, ngf 8: checkpoint dir -> test-mode Options -> init_nets(eager) -> run_dask / inference
    tag = 'dl_m2'
    make_checkpoint(out, tag, 'DeepLIIF', 2, True, 'unet_64', 8, 900)
    mdir = os.path.join(CKPT, tag)
    opt = models.get_opt(mdir)                          # Options(path_file=train_opt.txt, mode='test'): forces ngf = 64 (options/__init__.py:75)
    out[f'{tag}/test_opt'] = np.array([f'{k}={getattr(opt, k)!r}' for k in ('model', 'modalities_no', 'seg_gen', 'mod_id_seg', 'input_id', 'modalities_names',
                                                                            'background_colors', 'scale_size', 'seg_weights', 'input_no', 'norm', 'ngf', 'phase',
                                                                            'is_train', 'seg_no', 'padding', 'net_g', 'net_gs')])
    opt.ngf = 8                                         # the checkpoint was trained at ngf 8: the caller overrides, as `deepliif test` users must
    img = Image.fromarray(synth_image(150, 100, 31))
    tile = img.crop((0, 0, 64, 64))
    res = models.run_dask(tile, model_path=mdir, eager_mode=True, opt=opt, use_dask=False)
    for k, v in pil_u8(res).items():
        out[f'{tag}/run_dask/{k}'] = v
    out[f'{tag}/run_dask_keys'] = np.array(list(res.keys()))
    res = models.run_dask(tile, model_path=mdir, eager_mode=True, opt=opt, use_dask=False, seg_only=True, seg_weights=[0.5, 0.0, 0.5])
    out[f'{tag}/run_dask_segonly_keys'] = np.array(list(res.keys()))
    for k, v in pil_u8(res).items():
        out[f'{tag}/run_dask_segonly/{k}'] = v
    a = np.asarray(img).copy()
    a[:, :70] = 252                                     # an empty strip: run_wrapper's constant tiles
    img2 = Image.fromarray(a)
    for name, kw in (('inf', {}), ('inf_seginter', dict(return_seg_intermediate=True)), ('inf_modonly', dict(mod_only=True)),
                     ('inf_segonly', dict(seg_only=True))):
        r = models.inference(img2, 64, 4, mdir, eager_mode=True, opt=opt, **kw)
        out[f'{tag}/{name}_keys'] = np.array(list(r.keys()))
        for k, v in pil_u8(r).items():
            out[f'{tag}/{name}/{k}'] = v
    # transform() and tensor_to_pil on their own
    from deepliif.data import transform
    from deepliif.util.util import tensor_to_pil
    out[f'{tag}/transform_in'] = np.asarray(tile)
    ts = transform(tile)
    out[f'{tag}/transform_out'] = ts.numpy()
    odd = img.crop((0, 0, 70, 61))                      # sides that are no multiple of 4: bicubic resize to 72 x 60 first
    out[f'{tag}/transform_odd_in'] = np.asarray(odd)
    out[f'{tag}/transform_odd_out'] = transform(odd).numpy()
    out[f'{tag}/t2p'] = np.asarray(tensor_to_pil(ts * 0.731))

    # ---- DeepLIIFExt (2 modalities + 2 seg generators with 9-channel input) and SDG (2 input modalities): run_dask / inference
    tag = 'ext_m2'
    make_checkpoint(out, tag, 'DeepLIIFExt', 2, True, 'unet_64', 8, 940)
    mdir = os.path.join(CKPT, tag)
    opt = models.get_opt(mdir)
    opt.ngf = 8
    out[f'{tag}/test_opt'] = np.array([f'{k}={getattr(opt, k)!r}' for k in ('model', 'modalities_no', 'seg_gen', 'modalities_names', 'background_colors',
                                                                            'scale_size', 'seg_weights', 'input_no', 'seg_no')])
    res = models.run_dask(tile, model_path=mdir, eager_mode=True, opt=opt, use_dask=False)
    out[f'{tag}/run_dask_keys'] = np.array(list(res.keys()))
    for k, v in pil_u8(res).items():
        out[f'{tag}/run_dask/{k}'] = v
    r = models.inference(img2, 64, 4, mdir, eager_mode=True, opt=opt)
    out[f'{tag}/inf_keys'] = np.array(list(r.keys()))
    for k, v in pil_u8(r).items():
        out[f'{tag}/inf/{k}'] = v

    tag = 'sdg_m2_in2'
    make_checkpoint(out, tag, 'SDG', 2, False, 'unet_64', 8, 970, input_no=2)
    mdir = os.path.join(CKPT, tag)
    opt = models.get_opt(mdir)
    opt.ngf = 8
    out[f'{tag}/test_opt'] = np.array([f'{k}={getattr(opt, k)!r}' for k in ('model', 'modalities_no', 'seg_gen', 'modalities_names', 'scale_size', 'input_no', 'seg_no')])
    wide = Image.fromarray(np.concatenate([np.asarray(img2), synth_image(150, 100, 32)], axis=1))      # two input modalities side by side
    r = models.inference(wide, 64, 4, mdir, eager_mode=True, opt=opt)
    out[f'{tag}/inf_keys'] = np.array(list(r.keys()))
    for k, v in pil_u8(r).items():
        out[f'{tag}/inf/{k}'] = v

    # ---- learning-rate schedules (networks.py:55-81), 8 epochs each
    import types
    for policy, extra in (('linear', dict(n_epochs=3, n_epochs_decay=4, epoch_count=1)), ('step', dict(lr_decay_iters=3)), ('cosine', dict(n_epochs=6)),
                          ('plateau', {})):
        o = types.SimpleNamespace(lr_policy=policy, n_epochs=100, n_epochs_decay=100, epoch_count=0, lr_decay_iters=50)
        o.__dict__.update(extra)
        p = torch.nn.Parameter(torch.zeros(1))
        optim = torch.optim.Adam([p], lr=2e-4, betas=(0.5, 0.999))
        sch = networks.get_scheduler(optim, o)
        lrs = [optim.param_groups[0]['lr']]
        for e in range(8):
            optim.step()
            if policy == 'plateau':
                sch.step(1.0 if e < 1 else 1.0 + 0.01 * e)      # a metric that stops improving
            else:
                sch.step()
            lrs.append(optim.param_groups[0]['lr'])
        out[f'sched/{policy}'] = np.array(lrs, dtype=np.float64)
        out[f'sched/{policy}_args'] = np.array([f'{k}={v}' for k, v in sorted(o.__dict__.items())])
    np.savez_compressed(os.path.join(HERE, 'seam_cases.npz'), **out)
    print('wrote seam_cases.npz', os.path.getsize(os.path.join(HERE, 'seam_cases.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
