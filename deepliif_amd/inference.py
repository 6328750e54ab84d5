"""Inference seam: init_nets() / run_dask() counterparts of deepliif/models/__init__.py:158-219 and :258-361.

The reference runs ONE tile per forward, spreads the generator groups over GPUs (`init_nets` chunker) and fans the per-net
forwards out on dask threads.  On MI355X all nine generators fit one GPU many times over (288 GB), so `nets` live on one
device, tiles are batched (per-sample normalisation keeps every tile's output identical to the reference's single-tile
forward), the seg generators consume the translation generators' outputs in engine layout (no NCHW round
trip), and tile-level data parallelism across GPUs is plain sharding of the tile list (no collective).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from . import engine as E
from . import networks
from .models import _get, init_input_and_mod_id


def read_model_params(path):
    """Minimal reader of the reference's '<phase>_opt.txt' (deepliif/options/__init__.py:8-36): 'key: value' lines."""
    import ast
    import re
    params = {}
    with open(path) as f:
        for line in f:
            if ':' not in line:
                continue
            key = line.split(':')[0].strip()
            val = ':'.join(line.split(':')[1:])
            for d in re.findall(r'\[default.+?\]', val):
                val = val.replace(d, '')
            val = val.strip()
            try:
                params[key] = ast.literal_eval(val)
            except Exception:
                params[key] = val
    return params


class _Opt:
    def __init__(self, d):
        self.__dict__.update(d)


def get_opt(model_dir, mode='test'):
    """deepliif/models/__init__.py:53-68 (test-mode defaults of deepliif/options/__init__.py:69-180 that the path reads)."""
    p = read_model_params(os.path.join(model_dir, 'train_opt.txt'))
    opt = _Opt(p)
    opt.is_train = False
    opt.phase = 'test'
    opt.input_nc, opt.output_nc, opt.ngf = 3, 3, _get(opt, 'ngf', 64)
    opt.norm = _get(opt, 'norm', 'batch')
    opt.no_dropout = True
    opt.input_no = _get(opt, 'input_no', 1)
    if not hasattr(opt, 'modalities_no') and hasattr(opt, 'targets_no'):
        opt.modalities_no = opt.targets_no - 1
    opt.seg_gen = _get(opt, 'seg_gen', True)
    opt.padding = _get(opt, 'padding', 'zero')
    if not hasattr(opt, 'net_g') and hasattr(opt, 'netG'):
        opt.net_g = opt.netG
    opt.net_g = _get(opt, 'net_g', 'resnet_9blocks')
    opt.net_gs = _get(opt, 'net_gs', 'unet_512')
    files = os.listdir(model_dir)
    seg_names = [f[:-4].split('_')[2][1:] for f in files if f.endswith('.pth') and 'net_G' in f]
    if not hasattr(opt, 'mod_id_seg') or opt.mod_id_seg is None:
        longest = max(seg_names, key=len) if seg_names else 'S0'
        opt.mod_id_seg = longest[0] if opt.seg_gen else None
    opt.input_id = 0 if any(n[1:] == '0' for n in seg_names if len(n) > 1) or not seg_names else 1
    if opt.modalities_no == 4 and not hasattr(opt, 'modalities_names'):
        opt.modalities_names = ['IHC', 'Hema', 'DAPI', 'Lap2', 'Marker']
        opt.seg_weights = [0.5, 0, 0, 0, 0.5]
    if not _get(opt, 'modalities_names', None):
        opt.modalities_names = [f'input{i + 1}' for i in range(opt.input_no)] + [f'mod{i + 1}' for i in range(opt.modalities_no)]
    opt.gpu_ids = list(range(torch.cuda.device_count()))
    return opt


def generator_names(opt):
    M, S, off = opt.modalities_no, _get(opt, 'mod_id_seg', 'S'), int(_get(opt, 'input_id', 0))
    g = [f'G{i + 1}' for i in range(M)]
    gs = [f'G{S}{off + i}' for i in range(M + 1)] if opt.seg_gen else []
    return g, gs


def build_generators(opt, device, precision: Optional[str] = None) -> 'OrderedDict[str, torch.nn.Module]':
    """All generators of a DeepLIIF model on ONE device, eval mode, BatchNorm on batch statistics."""
    g, gs = generator_names(opt)
    net_g = opt.net_g if isinstance(opt.net_g, (list, tuple)) else [opt.net_g] * len(g)
    net_gs = opt.net_gs if isinstance(opt.net_gs, (list, tuple)) else [opt.net_gs] * len(gs)
    ids = [device.index if device.index is not None else 0] if device.type == 'cuda' else []
    nets = OrderedDict()
    cin = opt.input_nc * _get(opt, 'input_no', 1)
    for n, arch in zip(g, net_g):
        nets[n] = networks.define_G(cin, opt.output_nc, opt.ngf, arch, opt.norm, False, 'normal', 0.02, ids, opt.padding)
    for n, arch in zip(gs, net_gs):
        nets[n] = networks.define_G(cin, opt.output_nc, opt.ngf, arch, opt.norm, False, 'normal', 0.02, ids)
    for net in nets.values():
        net.eval()
        if precision:
            net.set_precision(precision)
    return nets


_NETS_CACHE: Dict = {}


def init_nets(model_dir, eager_mode=False, opt=None, phase='test'):
    """deepliif/models/__init__.py:158-219.  Returns {name: net}; every net is callable on a [N,3,H,W] tensor.
    TorchScript '<name>.pt' files are CUDA/ATen graphs and are not loaded here: the '<epoch>_net_<name>.pth' state_dicts
    (the reference's own checkpoint format, base_model.py:190-212) are the interchange."""
    key = (model_dir, phase)
    if key in _NETS_CACHE and opt is None:
        return _NETS_CACHE[key]
    if opt is None:
        opt = get_opt(model_dir, mode=phase)
    device = torch.device('cuda', opt.gpu_ids[0] if _get(opt, 'gpu_ids', None) else 0)
    nets = build_generators(opt, device, _get(opt, 'precision', None))
    epoch = _get(opt, 'epoch', 'latest')
    for n, net in nets.items():
        path = os.path.join(model_dir, f'{epoch}_net_{n}.pth')
        sd = torch.load(path, map_location='cpu')
        if hasattr(sd, '_metadata'):
            del sd._metadata
        net.load_state_dict(sd)
    _NETS_CACHE[key] = nets
    return nets


def transform(img, scale_size=None) -> torch.Tensor:
    """deepliif/data/__init__.py:133-138: PIL RGB -> [1,3,H,W] in [-1,1] (H, W rounded to multiples of 4, bicubic)."""
    from PIL import Image
    if scale_size:
        img = img.resize((scale_size, scale_size))
    w, h = img.size
    w4, h4 = int(round(w / 4) * 4), int(round(h / 4) * 4)
    if (w4, h4) != (w, h):
        img = img.resize((w4, h4), Image.BICUBIC)
    a = np.asarray(img.convert('RGB'), dtype=np.float32) / 255.0
    return ((torch.from_numpy(a).permute(2, 0, 1) - 0.5) / 0.5).unsqueeze(0)


def tensor_to_pil(t: torch.Tensor):
    """deepliif/util/util.py:117-139: first image of the batch, (x+1)/2*255 truncated to uint8."""
    from PIL import Image
    a = t[0].detach().float().cpu().numpy()
    if a.shape[0] == 1:
        a = np.tile(a, (3, 1, 1))
    return Image.fromarray(((np.transpose(a, (1, 2, 0)) + 1) / 2.0 * 255.0).astype(np.uint8))


def run_generators(ts: torch.Tensor, nets, opt, seg_only=False, mod_only=False, seg_weights=None) -> 'OrderedDict[str, torch.Tensor]':
    """The DeepLIIF branch of run_dask (deepliif/models/__init__.py:293-361) on a batch of tiles [N,3,H,W]:
    G_i(tile); GS_0(tile); GS_i(G_i(tile)); seg = sum_k w_k * seg_k.  Returns name -> [N,3,H,W] fp32 tensors."""
    M, S, off = opt.modalities_no, _get(opt, 'mod_id_seg', 'S'), int(_get(opt, 'input_id', 0))
    first = next(iter(nets.values()))
    device = next(first.parameters()).device
    prec = E.Precision.get(first.precision)
    ctx = E.Ctx(prec, None, training=False, per_sample_norm=True)
    x = E.to_engine(ts.to(device), prec)
    seg_map = OrderedDict((f'G{i + 1}', f'G{S}{off + i + 1}') for i in range(M))
    weights = None
    if opt.seg_gen:
        if seg_weights is None:
            weights = {f'G{S}{off + i}': 1 / (M + 1) for i in range(M + 1)}
        else:
            weights = {f'G{S}{off + i}': w for i, w in enumerate(seg_weights)}
        if seg_only:
            seg_map = OrderedDict((k, v) for k, v in seg_map.items() if weights[v] != 0)
    gens = OrderedDict((k, nets[k].run(ctx, x)) for k in seg_map)
    names = _get(opt, 'modalities_names', [])
    if 'Marker' in names:
        km = f'G{names.index("Marker")}'
        if km not in gens and km in nets:
            gens[km] = nets[km].run(ctx, x)
    res = OrderedDict((k, E.from_engine(v)) for k, v in gens.items())
    if not opt.seg_gen or mod_only:
        return res
    segs = OrderedDict((v, nets[v].run(ctx, gens[k])) for k, v in seg_map.items())
    base = f'G{S}{off}'
    if weights[base] != 0:
        segs[base] = nets[base].run(ctx, x)
    keys = list(segs.keys())
    seg = E.weighted_sum(ctx, [segs[k] for k in keys], [float(weights[k]) for k in keys])
    if seg_only and M > 0:
        last = f'G{M}'
        res = OrderedDict([(last, res[last])] if last in res else [])
    else:
        res.update((k, E.from_engine(v)) for k, v in segs.items())
    res[f'G{S}'] = E.from_engine(seg)
    return res


def run_dask(img, model_path=None, nets=None, eager_mode=False, opt=None, seg_only=False, mod_only=False, seg_weights=None, use_dask=True,
             output_tensor=False):
    """Same call shape as the reference's run_dask (`use_dask` is accepted and ignored: branch concurrency is the engine's
    business).  `img` is a PIL image or a [N,3,H,W] tensor; PIL in -> dict of PIL images of the first tile."""
    assert model_path is not None or nets is not None, 'Provide either the model path or the networks object.'
    if nets is None:
        nets = init_nets(os.getenv('DEEPLIIF_MODEL_DIR', model_path), eager_mode, opt)
    if opt is None:
        opt = get_opt(os.getenv('DEEPLIIF_MODEL_DIR', model_path))
    if _get(opt, 'model', 'DeepLIIF') not in ('DeepLIIF', 'DeepLIIFKD'):
        raise NotImplementedError(f'run_dask for model {opt.model} is not on the MI355X hot path yet')
    ts = img if isinstance(img, torch.Tensor) else transform(img, _get(opt, 'scale_size', None))
    res = run_generators(ts, nets, opt, seg_only, mod_only, seg_weights)
    if output_tensor:
        return res
    return {k: tensor_to_pil(v) for k, v in res.items()}
