"""Golden vectors for the VGG19 perceptual term, produced by the REFERENCE's own VGGLoss / DeepLIIFModel (build container only).

    python tests/golden/make_golden_vgg.py      -> tests/golden/vgg_cases.npz

The reference builds Vgg19 from torchvision.models.vgg19(pretrained=True).features (networks.py:698-731): a download, and torchvision is
not installed here.  This script supplies a stand-in `vgg19` whose `.features` is torchvision's documented configuration 'E' (3x3 convs
+ ReLU(inplace) + 2x2 max pooling) carrying SEEDED RANDOM weights (oracle.random_vgg19_state_dict: torchvision's own kaiming init);
slicing, the five L1 terms and their weights, and the place of the term in loss_G are the reference's code.  What the vectors pin is
therefore the arithmetic of the term, not the pretrained weights (which must be supplied as a file at run time anyway).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402
from golden_util import digest, seeded_uniform  # noqa: E402
from oracle import deepliif_oracle as O  # noqa: E402

_ref_import.install_stubs()
import deepliif.models as models  # noqa: E402
from deepliif.models import networks  # noqa: E402
from deepliif.options import Options  # noqa: E402

VGG_SEED = 4321


def fake_vgg19(pretrained=True):
    layers, cin = [], 3
    for v in O.VGG19_CFG:
        if v == 'M':
            layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [torch.nn.Conv2d(cin, v, kernel_size=3, padding=1), torch.nn.ReLU(inplace=True)]
            cin = v
    net = types.SimpleNamespace(features=torch.nn.Sequential(*layers))
    sd = O.random_vgg19_state_dict(torch.Generator().manual_seed(VGG_SEED))
    missing = net.features.load_state_dict({k[len('features.'):]: v for k, v in sd.items()}, strict=False)
    assert all(k.split('.')[0] in ('30', '32', '34') for k in missing.missing_keys), missing      # conv5_2..4 lie beyond features[0:30]
    return net


networks.models = types.SimpleNamespace(vgg19=fake_vgg19)
torch.set_num_threads(8)


def main():
    out = {'vgg_seed': np.array(VGG_SEED)}
    # ---- the loss on its own: value and d/dx
    crit = networks.VGGLoss()
    for tag, shape in (('s64', (2, 3, 64, 64)), ('s48x80', (1, 3, 48, 80))):
        x = seeded_uniform(shape, 71).requires_grad_(True)
        y = seeded_uniform(shape, 72)
        loss = crit(x, y)
        loss.backward()
        out[f'{tag}/shape'] = np.array(shape)
        out[f'{tag}/loss'] = np.array(float(loss))
        out[f'{tag}/dx'] = x.grad.numpy()
        feats = crit.vgg(x.detach())
        for i, f in enumerate(feats):
            out[f'{tag}/feat{i}_digest'] = digest(f)
            out[f'{tag}/feat{i}_shape'] = np.array(f.shape)
    # ---- the full default objective: DeepLIIF, 2 modalities + seg, lambda_feat = 100 (what Options sets in train mode), 2 steps
    n = 3
    p = dict(model='DeepLIIF', name='vgg', checkpoints_dir='/tmp/golden_ckpt', gpu_ids=[], phase='train', preprocess='none', remote_transfer_cmd=None,
             continue_train=False, modalities_no=2, seg_gen=True, modalities_names=[], input_nc=3, input_no=1, output_nc=3, ngf=8, ndf=8,
             net_g='resnet_9blocks', net_gs='unet_64', net_d='n_layers', norm='batch', no_dropout=True, init_type='normal', init_gain=0.02, padding='zero',
             upsample='convtranspose', gan_mode='vanilla', gan_mode_s='lsgan', optimizer='adam', lr_g=2e-4, lr_d=2e-4, beta1=0.5, lr_policy='linear',
             n_epochs=100, n_epochs_decay=100, epoch_count=0, seg_weights=[1.0 / n] * n, loss_G_weights=[0.5, 0.3, 0.2], loss_D_weights=[1.0 / n] * n,
             verbose=False, epoch='latest', load_iter=0)
    opt = Options(d_params=p)
    assert opt.lambda_feat == 100
    os.makedirs('/tmp/golden_ckpt/vgg', exist_ok=True)
    model = models.create_model(opt)
    model.setup(opt)
    seeds = {}
    for j, name in enumerate(model.model_names):
        net = getattr(model, 'net' + name)
        if name.startswith('D'):
            arch, cin, pad = 'n_layers', 6, 'zero'
        elif name in model.model_names_g:
            arch, cin, pad = 'resnet_9blocks', 3, 'zero'
        else:
            arch, cin, pad = 'unet_64', 3, 'reflect'
        net.load_state_dict(O.random_state_dict(arch, cin, 3, 8, 'batch', pad, 4, generator=torch.Generator().manual_seed(600 + j)), strict=True)
        seeds[name] = 600 + j
    A = seeded_uniform((2, 3, 64, 64), 22)
    B = [seeded_uniform((2, 3, 64, 64), 23 + i) for i in range(3)]
    out['step/model_names'] = np.array(model.model_names)
    out['step/net_seeds'] = np.array([seeds[k] for k in model.model_names])
    out['step/loss_names'] = np.array(model.loss_names)
    out['step/loss_G_weights'] = np.array(p['loss_G_weights'])
    for s in range(2):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        losses = model.get_current_losses()
        out[f'step{s}/losses'] = np.array([losses[k] for k in model.loss_names], dtype=np.float64)
        out[f'step{s}/vgg'] = np.array([float(getattr(model, f'loss_G_VGG_{i + 1}')) for i in range(2)])
        for i in range(2):
            out[f'step{s}/fake_B_{i + 1}'] = getattr(model, f'fake_B_{i + 1}').detach().numpy()[:, :, ::2, ::2]
        out[f'step{s}/fake_B_S'] = getattr(model, f'fake_B_{model.mod_id_seg}').detach().numpy()[:, :, ::2, ::2]
        for k in model.model_names:
            sd = getattr(model, 'net' + k).state_dict()
            out[f'step{s}/w_digest/{k}'] = digest(torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()]))
    np.savez_compressed(os.path.join(HERE, 'vgg_cases.npz'), **out)
    print('wrote vgg_cases.npz', os.path.getsize(os.path.join(HERE, 'vgg_cases.npz')) // 1024, 'KiB', {k: float(v) for k, v in losses.items()},
          out['step1/vgg'])


if __name__ == '__main__':
    main()
