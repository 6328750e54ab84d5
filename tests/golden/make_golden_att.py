"""Golden vectors for the attention U-Net (`--net-gs unet_512_attention`: deepliif/models/att_unet.py:117-199 through networks.define_G,
networks.py:189-190) from the REFERENCE.  Build container only (needs /root/reference):  python tests/golden/make_golden_att.py

Writes tests/golden/att_unet.npz -- data only.  The network has fixed widths (64 ... 512, 56.6 M parameters), so nothing of it is stored:
weights are drawn from a seeded generator by oracle.random_state_dict (which also proves the oracle's key / shape table against the
reference's state_dict through load_state_dict(strict=True)); outputs and input gradients are stored strided + as digests, parameter gradients
and BatchNorm running statistics as digests (golden_util.digest).  Smallest legal input: 256 x 256 (eight stride-2 levels, x8 is 1 x 1); the batch-1 case uses 512 x 512, the size the net is named after: at 256 x 256 with N = 1
the innermost BatchNorms would normalise 4 values per channel, where the backward pass amplifies rounding differences between any two
implementations (ATen vs the oracle: 2e-2 on dx) and pins nothing."""
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
from deepliif.util import disable_batchnorm_tracking_stats  # noqa: E402

torch.set_num_threads(8)


def main():
    out = {}
    for tag, cin, shape, wseed, xseed in (('in3_n1', 3, (1, 3, 512, 512), 301, 302), ('in9_n2', 9, (2, 9, 512, 512), 303, 304)):
        net = networks.define_G(cin, 3, 64, 'unet_512_attention', 'batch', True, 'normal', 0.02, [])      # use_dropout / norm / ngf are ignored by define_G
        sd = O.random_state_dict('unet_512_attention', cin, 3, 64, 'batch', 'zero', 4, generator=torch.Generator().manual_seed(wseed))
        net.load_state_dict(sd, strict=True)
        out[f'{tag}/meta'] = np.array([str(cin), str(wseed), str(xseed), str(tuple(shape))])
        out[f'{tag}/sd_keys'] = np.array(list(sd.keys()))
        out[f'{tag}/w_digest'] = digest(torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()]))
        net.train()
        x = seeded_uniform(shape, xseed).requires_grad_(True)
        y = net(x)
        r = torch.randn(y.shape, generator=torch.Generator().manual_seed(99))
        (y * r).sum().backward()
        out[f'{tag}/y_strided'] = y.detach()[:, :, ::8, ::8].numpy()
        out[f'{tag}/y_digest'] = digest(y)
        out[f'{tag}/dx_strided'] = x.grad[:, :, ::8, ::8].numpy()
        out[f'{tag}/dx_digest'] = digest(x.grad)
        names = []
        for k, p in net.named_parameters():
            names.append(k)
            out[f'{tag}/dw/{k}'] = digest(p.grad)
        out[f'{tag}/param_names'] = np.array(names)
        for k, v in net.state_dict().items():
            if 'running_' in k:
                out[f'{tag}/sd_after/{k}'] = digest(v) if v.numel() > 64 else v.numpy()
        net.eval()
        disable_batchnorm_tracking_stats(net)
        with torch.no_grad():
            ye = net(x.detach())
        out[f'{tag}/y_eval_strided'] = ye[:, :, ::8, ::8].numpy()
        out[f'{tag}/y_eval_digest'] = digest(ye)
    np.savez_compressed(os.path.join(HERE, 'att_unet.npz'), **out)
    print('att_unet.npz', len(out), 'arrays', os.path.getsize(os.path.join(HERE, 'att_unet.npz')) // 1024, 'KB')


if __name__ == '__main__':
    main()
