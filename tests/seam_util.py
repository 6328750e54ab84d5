"""Shared by the CPU (emulated backend) and GPU seam tests: rebuild the reference's checkpoint directories from the seeds stored in
tests/golden/seam_cases.npz (weights are never stored; tests/golden/make_golden_seam.py), and compare uint8 result images."""
import os
import shutil

import numpy as np
import torch

from golden_util import digest_close
from oracle import deepliif_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
Z = np.load(os.path.join(G, 'seam_cases.npz'))


def build_checkpoint_dir(tmp_path, tag, nf=8):
    """<tmp>/<tag>/latest_net_<name>.pth + train_opt.txt exactly as the reference's save_networks / print_options left them"""
    d = os.path.join(str(tmp_path), tag)
    os.makedirs(d, exist_ok=True)
    shutil.copy(os.path.join(G, f'seam_train_opt_{tag}.txt'), os.path.join(d, 'train_opt.txt'))
    for name, seed, arch in zip(Z[f'{tag}/model_names'].tolist(), Z[f'{tag}/net_seeds'].tolist(), Z[f'{tag}/net_arch'].tolist()):
        a, cin, pad = arch.split('|')
        sd = O.random_state_dict(a, int(cin), 3, nf, 'batch', pad, 4, generator=torch.Generator().manual_seed(int(seed)))
        assert list(sd.keys()) == Z[f'{tag}/sd_keys/{name}'].tolist()
        assert ['x'.join(str(x) for x in v.shape) for v in sd.values()] == Z[f'{tag}/sd_shapes/{name}'].tolist()
        ok, msg = digest_close(torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()]), Z[f'{tag}/sd_digest/{name}'], 1e-12)
        assert ok, (name, msg)
        torch.save(sd, os.path.join(d, f'latest_net_{name}.pth'))
    assert sorted(os.listdir(d)) == Z[f'{tag}/files'].tolist()
    return d


# the DeepLIIFKD teacher of tests/golden/step_kd_m2.npz: (network, architecture, padding of define_G), seeds 800 + index; ngf = 64 because the
# reference's test-mode Options force it (options/__init__.py:75) and DeepLIIFKD_model.py:107-112 cannot override it
KD_TEACHER_NETS = (('G1', 'resnet_9blocks', 'zero'), ('G2', 'resnet_9blocks', 'zero'), ('GS0', 'unet_64', 'reflect'), ('GS1', 'unet_64', 'reflect'),
                   ('GS2', 'unet_64', 'reflect'))


def build_kd_teacher_dir(tmp_path):
    """<tmp>/kd_teacher: train_opt.txt of the 'dl_m2' seam case (DeepLIIF, 2 modalities + seg, unet_64 seg generators, BatchNorm) + seeded
    generator checkpoints at ngf 64"""
    d = os.path.join(str(tmp_path), 'kd_teacher')
    os.makedirs(d, exist_ok=True)
    shutil.copy(os.path.join(G, 'seam_train_opt_dl_m2.txt'), os.path.join(d, 'train_opt.txt'))
    for j, (name, arch, pad) in enumerate(KD_TEACHER_NETS):
        sd = O.random_state_dict(arch, 3, 3, 64, 'batch', pad, 4, generator=torch.Generator().manual_seed(800 + j))
        torch.save(sd, os.path.join(d, f'latest_net_{name}.pth'))
    return d


def close_u8(got, exp, max_mismatch):
    """uint8 images produced through float -> uint8 TRUNCATION: a 1e-5 difference in the float flips a pixel that sits on an integer
    boundary, so equality is 'never more than one step apart, and only in a small fraction of the pixels'"""
    got, exp = np.asarray(got), np.asarray(exp)
    assert got.shape == exp.shape, (got.shape, exp.shape)
    d = np.abs(got.astype(int) - exp.astype(int))
    assert d.max() <= 1, int(d.max())
    frac = float((d != 0).mean())
    assert frac <= max_mismatch, frac
    return frac


class _Holder(torch.nn.Module):
    def forward(self, x):
        return x


def serialize_checkpoint_dir(mdir, out_dir, drop_running_stats=True):
    """The on-disk form `deepliif serialize` (cli.py:770-830) leaves behind -- `<name>.pt` TorchScript modules + train_opt.txt, NO .pth files --
    built WITHOUT the reference: every `<epoch>_net_<name>.pth` state_dict is wrapped in a module tree with the same parameter paths and
    traced (identity forward).  What the engine relies on is exactly what this reproduces: torch.jit.load(f).state_dict() carries the
    reference's keys, minus the BatchNorm running statistics that disable_batchnorm_tracking_stats nulls before tracing
    (util/__init__.py:743-755).  (The reference's own traced files cannot be committed as fixtures: a TorchScript archive embeds the traced
    module's source.)  tests/test_reference_seam.py checks the same loader against files the reference's tracer really wrote."""
    os.makedirs(out_dir, exist_ok=True)
    shutil.copy(os.path.join(mdir, 'train_opt.txt'), os.path.join(out_dir, 'train_opt.txt'))
    for f in sorted(os.listdir(mdir)):
        if not f.endswith('.pth'):
            continue
        name = f[len('latest_net_'):-len('.pth')]
        if not name.startswith('G'):            # serialize traces what init_nets returns: the generators
            continue
        sd = torch.load(os.path.join(mdir, f), map_location='cpu')
        root = _Holder()
        for k, v in sd.items():
            leaf = k.rsplit('.', 1)[-1]
            if drop_running_stats and leaf in ('running_mean', 'running_var'):
                continue
            mod = root
            for part in k.split('.')[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, _Holder())
                mod = mod._modules[part]
            if leaf in ('running_mean', 'running_var', 'num_batches_tracked'):
                mod.register_buffer(leaf, v.clone())
            else:
                mod.register_parameter(leaf, torch.nn.Parameter(v.clone(), requires_grad=False))
        torch.jit.trace(root, torch.zeros(1)).save(os.path.join(out_dir, f'{name}.pt'))
    return out_dir
