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
from . import ops
from .models import _get


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


def _seg_names_in_dir(model_dir):
    """(mod_id_seg, input_id) sniffed from the generator file names like the reference does (deepliif/util/util.py:208-240):
    'latest_net_GS0.pth' -> names 'S0', ...; the seg id is the first character of the longest name, input id '0' iff a '...0' file exists"""
    files = os.listdir(model_dir)
    names = [f[:-4].split('_')[2][1:] for f in files if f.endswith('.pth') and 'net_G' in f]
    if not names:
        names = [f[1:-3] for f in files if f.endswith('.pt') and f.startswith('G')]
    if not names:
        raise Exception('Cannot find any model file ending with .pt or .pth in directory', model_dir)
    return max(names, key=len)[0], ('0' if '0' in [n[1:] for n in names] else '1')


def get_opt(model_dir, mode='test'):
    """deepliif/models/__init__.py:53-68 -> Options(path_file=..., mode='test') (deepliif/options/__init__.py:38-180): parse
    '<phase>_opt.txt' and back-fill the test-mode defaults the inference path reads."""
    path = os.path.join(model_dir, 'test_opt.txt')
    if mode == 'train' or not os.path.exists(path):
        path = os.path.join(model_dir, 'train_opt.txt')
    opt = _Opt(read_model_params(path))
    opt.optimizer = _get(opt, 'optimizer', 'adam')
    opt.model = _get(opt, 'model', 'DeepLIIF')
    if mode == 'train':
        opt.is_train = True
        return opt
    opt.is_train, opt.phase, opt.continue_train = False, 'test', False
    opt.input_nc, opt.output_nc, opt.ngf = 3, 3, 64                      # options/__init__.py:73-75 (forced in test mode)
    opt.norm = _get(opt, 'norm', 'batch')
    opt.no_dropout = True
    if not hasattr(opt, 'modalities_no') and hasattr(opt, 'targets_no'):
        opt.modalities_no = opt.targets_no - 1
    if opt.model in ('DeepLIIF', 'DeepLIIFKD'):
        sniff = None
        if not hasattr(opt, 'mod_id_seg') or opt.mod_id_seg is None:
            sniff = _seg_names_in_dir(model_dir)
            opt.mod_id_seg = sniff[0]
        opt.input_id = int((sniff or _seg_names_in_dir(model_dir))[1])
        if hasattr(opt, 'seg_gen') and opt.seg_gen is False:
            opt.mod_id_seg = None
        if opt.modalities_no == 4 and not hasattr(opt, 'modalities_names'):
            opt.modalities_names = ['IHC', 'Hema', 'DAPI', 'Lap2', 'Marker']
            opt.seg_weights = [0.5, 0, 0, 0, 0.5]
        if not _get(opt, 'modalities_names', None):
            opt.modalities_names = [f'input{i + 1}' for i in range(_get(opt, 'input_no', 1))] + [f'mod{i + 1}' for i in range(opt.modalities_no)]
    else:
        opt.modalities_names = [f'mod{i}' for i in range(opt.modalities_no + 1)]
    if not hasattr(opt, 'background_colors'):
        opt.background_colors = ([(201, 211, 208), (10, 10, 10), (0, 0, 0), (10, 10, 10)] if opt.model in ('DeepLIIF', 'DeepLIIFKD')
                                 else [(10, 10, 10)] * opt.modalities_no)
    opt.checkpoints_dir, opt.name = os.path.dirname(os.path.abspath(model_dir)), os.path.basename(os.path.abspath(model_dir))
    if isinstance(_get(opt, 'gpu_ids', ()), int):
        opt.gpu_ids = (opt.gpu_ids,)
    if not hasattr(opt, 'seg_no'):
        if opt.model == 'DeepLIIF':
            opt.seg_no, opt.seg_gen = 1, True
        elif opt.model == 'DeepLIIFExt':
            opt.seg_no = opt.modalities_no if opt.seg_gen else 0
        elif opt.model == 'SDG':
            opt.seg_no, opt.seg_gen = 0, False
        else:
            raise Exception(f'seg_gen cannot be automatically determined for {opt.model}')
    if opt.model == 'SDG':
        opt.seg_gen = False
    opt.input_no = _get(opt, 'input_no', 1)
    if not hasattr(opt, 'scale_size'):
        opt.scale_size = {'DeepLIIF': 512, 'SDG': 512, 'DeepLIIFExt': 1024}[opt.model]
    if not hasattr(opt, 'seg_weights'):
        opt.seg_weights = [0.25, 0.15, 0.25, 0.1, 0.25] if opt.model == 'DeepLIIF' else [1 / opt.modalities_no] * opt.modalities_no
    opt.upsample = _get(opt, 'upsample', 'convtranspose')
    opt.padding = _get(opt, 'padding', 'zero')
    if not hasattr(opt, 'net_g') and hasattr(opt, 'netG'):
        opt.net_g = opt.netG
    opt.net_g = _get(opt, 'net_g', 'resnet_9blocks')
    opt.net_gs = _get(opt, 'net_gs', 'unet_512')
    opt.use_dp = False
    opt.gpu_ids = list(range(torch.cuda.device_count()))
    return opt


def generator_names(opt):
    """(translation generators, seg generators) in the reference's naming (deepliif/models/__init__.py:172-198)"""
    M = opt.modalities_no
    if _get(opt, 'model', 'DeepLIIF') in ('DeepLIIFExt', 'SDG'):
        return [f'G_{i + 1}' for i in range(M)], ([f'GS_{i + 1}' for i in range(M)] if opt.seg_gen else [])
    if _get(opt, 'model', 'DeepLIIF') == 'CycleGAN':          # one direction only (models/__init__.py:193-197)
        return [f'{"GB" if _get(opt, "BtoA", False) else "GA"}_{i + 1}' for i in range(M)], []
    S, off = _get(opt, 'mod_id_seg', 'S'), int(_get(opt, 'input_id', 0))
    g = [f'G{i + 1}' for i in range(M)]
    gs = [f'G{S}{off + i}' for i in range(M + 1)] if opt.seg_gen else []
    return g, gs


def build_generators(opt, device, precision: Optional[str] = None) -> 'OrderedDict[str, torch.nn.Module]':
    """All generators of a model on ONE device, eval mode, BatchNorm on batch statistics (288 GB of HBM hold every generator many
    times over: the reference's per-group GPU placement, models/__init__.py:201-211, has no counterpart)."""
    g, gs = generator_names(opt)
    net_g = opt.net_g if isinstance(opt.net_g, (list, tuple)) else [opt.net_g] * len(g)
    net_gs = opt.net_gs if isinstance(opt.net_gs, (list, tuple)) else [opt.net_gs] * len(gs)
    ids = [device.index if device.index is not None else 0] if device.type == 'cuda' else []
    nets = OrderedDict()
    model = _get(opt, 'model', 'DeepLIIF')
    cin = opt.input_nc * (_get(opt, 'input_no', 1) if model != 'DeepLIIFExt' else 1)
    cin_s = opt.input_nc * 3 if model in ('DeepLIIFExt', 'SDG') else cin          # GS_i(cat(A, fake_1, fake_i)), DeepLIIFExt_model.py:85,173
    cout = opt.output_nc
    if model == 'CycleGAN' and _get(opt, 'BtoA', False):       # GB_i: B -> A (CycleGAN_model.py:82-85)
        cin, cout = opt.output_nc, opt.input_nc
    for n, arch in zip(g, net_g):
        nets[n] = networks.define_G(cin, cout, opt.ngf, arch, opt.norm, False, 'normal', 0.02, ids, opt.padding, _get(opt, 'upsample', 'convtranspose'))
    for n, arch in zip(gs, net_gs):
        nets[n] = networks.define_G(cin_s, opt.output_nc, opt.ngf, arch, opt.norm, False, 'normal', 0.02, ids)
    for net in nets.values():
        net.eval()
        if precision:
            net.set_precision(precision)
    return nets


_NETS_CACHE: Dict = {}


def _device_for(opt) -> torch.device:
    """every generator lives on ONE GPU: the first of opt.gpu_ids (the reference spreads net groups over them, models/__init__.py:201-211)"""
    if not torch.cuda.is_available():
        from . import _lib
        raise _lib.HipLibraryError('deepliif_amd inference runs on MI355X only (no CPU fallback): no GPU is visible')
    ids = _get(opt, 'gpu_ids', None)
    return torch.device('cuda', ids[0] if ids else 0)


_BN_STAT_KEYS = ('running_mean', 'running_var', 'num_batches_tracked')


def load_generator_weights(net, model_dir, name, epoch='latest', eager_mode=False):
    """Fill `net` from a reference model directory.  The reference knows two on-disk forms (deepliif/models/__init__.py:117-156, 216-219):
      * eager_mode=True : `<epoch>_net_<name>.pth`, the state_dict save_networks() writes (base_model.py:190-212);
      * eager_mode=False: `<name>.pt`, the TorchScript module `deepliif serialize` traces from it (cli.py:770-830) -- the DEFAULT inference
        route and the form the published models are distributed in.
    The engine does not execute TorchScript graphs (they are ATen programs); a `.pt` file is used as a WEIGHT CONTAINER: its
    state_dict() carries the same keys as the `.pth` file, minus the BatchNorm running statistics, which the reference nulls before
    tracing (disable_batchnorm_tracking_stats, util/__init__.py:743-755) and which the per-sample-statistics inference path never reads.
    eager_mode=False falls back to the `.pth` file when no `.pt` exists (a training directory that was never serialized)."""
    pth = os.path.join(model_dir, f'{epoch}_net_{name}.pth')
    pt = os.path.join(model_dir, f'{name}.pt')
    if not eager_mode and os.path.exists(pt):
        sd = torch.jit.load(pt, map_location='cpu').state_dict()
        src = pt
    elif os.path.exists(pth):
        sd = torch.load(pth, map_location='cpu')
        src = pth
    else:
        raise FileNotFoundError(f'no weights for network {name!r} in {model_dir}: neither {os.path.basename(pth)}' +
                                ('' if eager_mode else f' nor {os.path.basename(pt)}') + ' exists' +
                                (f' ({os.path.basename(pt)} does: pass eager_mode=False to read it)' if eager_mode and os.path.exists(pt) else ''))
    if hasattr(sd, '_metadata'):
        del sd._metadata
    res = net.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys if k.rsplit('.', 1)[-1] not in _BN_STAT_KEYS]
    if missing or res.unexpected_keys:
        raise RuntimeError(f'{src}: state_dict does not match network {name!r}: missing {missing}, unexpected {list(res.unexpected_keys)}')
    return src


def _nets_cache_key(model_dir, phase, eager_mode, opt):
    """the reference's lru_cache keys on (model_dir, eager_mode, opt, phase) (models/__init__.py:157); opt objects here are plain namespaces, so
    the fields that change what gets built / loaded stand in for them"""
    fields = tuple((k, repr(_get(opt, k, None))) for k in ('model', 'precision', 'epoch', 'norm', 'net_g', 'net_gs', 'ngf', 'padding', 'modalities_no', 'seg_gen',
                                                           'input_no', 'gpu_ids', 'mod_id_seg', 'input_id', 'BtoA', 'upsample', 'input_nc', 'output_nc'))
    return (os.path.abspath(model_dir), phase, bool(eager_mode), fields)


def init_nets(model_dir, eager_mode=False, opt=None, phase='test'):
    """deepliif/models/__init__.py:158-219.  Returns {name: net}; every net is callable on a [N,C,H,W] tensor.
    eager_mode=False (the reference's default) reads the serialized `<name>.pt` files, eager_mode=True the `<epoch>_net_<name>.pth`
    checkpoints -- see load_generator_weights().  Nets are process-lifetime singletons per (directory, phase, eager_mode, relevant
    option fields), like the reference's lru_cache."""
    if opt is None:
        opt = get_opt(model_dir, mode=phase)
    key = _nets_cache_key(model_dir, phase, eager_mode, opt)
    if key in _NETS_CACHE:
        return _NETS_CACHE[key]
    nets = build_generators(opt, _device_for(opt), _get(opt, 'precision', None))
    epoch = _get(opt, 'epoch', 'latest')
    for n, net in nets.items():
        load_generator_weights(net, model_dir, n, epoch, eager_mode)
    _NETS_CACHE[key] = nets
    E.settle_gc()
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


def _seg_weight_map(opt, seg_weights):
    M, S, off = opt.modalities_no, _get(opt, 'mod_id_seg', 'S'), int(_get(opt, 'input_id', 0))
    if seg_weights is None:
        return {f'G{S}{off + i}': 1 / (M + 1) for i in range(M + 1)}
    return {f'G{S}{off + i}': w for i, w in enumerate(seg_weights)}


def _wrapper_flags(opt, seg_only, mod_only):
    """run_wrapper forwards seg_only / mod_only to the generator DAG only for DeepLIIF / DeepLIIFKD; for DeepLIIFExt / SDG it calls
    run_fn(tile, model_path, None, eager_mode, opt) (deepliif/models/__init__.py:446-452), so their tiled inference always produces GS_i"""
    if _get(opt, 'model', 'DeepLIIF') in ('DeepLIIFExt', 'SDG', 'CycleGAN'):
        return False, False
    return seg_only, mod_only


# ---- fp16 policy: IEEE half tops out at 65 504.  The forward values of these networks sit orders of magnitude below that (normalised activations), but
# a checkpoint with unusually large weights could overflow a conv output; the following norm turns the inf into NaN for the rest of the net.  Every fp16 batch
# therefore ORs "some output holds a NaN" into a device-side flag (no host sync on the launch path); the flag is read -- and the error raised -- where the host
# waits for the GPU anyway: at the end of infer_region / inference(), in run_dask's PIL path, and at the start of the next run_generators call.
_FP16_NAN: Dict[torch.device, torch.Tensor] = {}


def _fp16_note(res):
    for a in res.values():
        t = a.t if isinstance(a, E.Act) else a
        if t.dtype != torch.float16:
            continue
        bad = torch.isnan(t).any()
        flag = _FP16_NAN.get(t.device)
        if flag is None:
            _FP16_NAN[t.device] = bad.clone()
        else:
            flag.logical_or_(bad)


def fp16_check():
    """Raise if any fp16 inference batch since the last call produced a NaN (half overflow).  One device-to-host read per device that ran fp16 batches."""
    for dev, flag in list(_FP16_NAN.items()):
        del _FP16_NAN[dev]
        if bool(flag.item()):
            raise FloatingPointError("precision 'fp16': an output of the generators holds NaN -- a value left IEEE half's range (65 504) inside a network. "
                                     "Serve this checkpoint with precision='bf16' (same speed, fp32 range) or 'fp32'.")


def run_generators_engine(x: E.Act, nets, opt, seg_only=False, mod_only=False, seg_weights=None, per_sample_norm=True) -> 'OrderedDict[str, E.Act]':
    """_run_generators_engine + the fp16 policy's overflow note (above)"""
    res = _run_generators_engine(x, nets, opt, seg_only, mod_only, seg_weights, per_sample_norm)
    if x.t.dtype == torch.float16:
        _fp16_note(res)
    return res


def _run_generators_engine(x: E.Act, nets, opt, seg_only=False, mod_only=False, seg_weights=None, per_sample_norm=True) -> 'OrderedDict[str, E.Act]':
    """The generator DAG of run_dask (deepliif/models/__init__.py:293-388) on a batch of tiles in ENGINE layout, outputs in engine
    layout, keys and key ORDER as the reference's result dict.
      DeepLIIF / DeepLIIFKD (:293-361): G_i(tile); GS_0(tile); GS_i(G_i(tile)); seg = sum_k w_k * seg_k
      DeepLIIFExt / SDG     (:362-388): G_i(tile); GS_i(cat(tile, G_1(tile), G_i(tile)))
      CycleGAN              (:362-372): GA_i(tile)  (GB_i with opt.BtoA)
    per_sample_norm: every tile normalised with its own statistics (= the reference's one-tile-at-a-time calls); False = the statistics of the
    whole batch, which is what a batched call of the reference's nets computes (the DeepLIIFKD teacher on a training batch)."""
    first = next(iter(nets.values()))
    prec = E.Precision.get(first.precision)
    ctx = E.Ctx(prec, None, training=False, per_sample_norm=per_sample_norm)
    model = _get(opt, 'model', 'DeepLIIF')
    M = opt.modalities_no
    if model == 'CycleGAN':
        return OrderedDict((k, nets[k].run(ctx, x)) for k in generator_names(opt)[0])
    if model in ('DeepLIIFExt', 'SDG'):
        names = [f'G_{i}' for i in range(1, M + 1)]
        gens = OrderedDict((k, nets[k].run(ctx, x)) for k in names)
        res = OrderedDict(gens)
        if mod_only or not opt.seg_gen:
            return res
        for i, k in enumerate(names, start=1):
            res[f'GS_{i}'] = nets[f'GS_{i}'].run(ctx, E.concat_channels(ctx, [x, gens[names[0]], gens[k]]))
        return res
    if model not in ('DeepLIIF', 'DeepLIIFKD'):
        raise NotImplementedError(f'run_dask for model {model} is not on the MI355X hot path')
    S, off = _get(opt, 'mod_id_seg', 'S'), int(_get(opt, 'input_id', 0))
    seg_map = OrderedDict((f'G{i + 1}', f'G{S}{off + i + 1}') for i in range(M))
    weights = None
    if opt.seg_gen:
        weights = _seg_weight_map(opt, seg_weights)
        if seg_only:
            seg_map = OrderedDict((k, v) for k, v in seg_map.items() if weights[v] != 0)
    streams = _infer_streams(x.t.device)
    if streams and opt.seg_gen and not mod_only and len(seg_map) > 1:
        return _run_deepliif_dag_on_streams(ctx, x, nets, opt, seg_map, weights, seg_only, streams)
    gens = OrderedDict((k, nets[k].run(ctx, x)) for k in seg_map)
    names = _get(opt, 'modalities_names', [])
    if 'Marker' in names:
        km = f'G{names.index("Marker")}'
        if km not in gens and km in nets:
            gens[km] = nets[km].run(ctx, x)
    if not opt.seg_gen or mod_only:
        return OrderedDict(gens)
    segs = OrderedDict((v, nets[v].run(ctx, gens[k])) for k, v in seg_map.items())
    base = f'G{S}{off}'
    if weights[base] != 0:
        segs[base] = nets[base].run(ctx, x)
    keys = list(segs.keys())
    seg = E.weighted_sum(ctx, [segs[k] for k in keys], [float(weights[k]) for k in keys])
    if seg_only and M > 0:
        last = f'G{M}'
        res = OrderedDict([(last, gens[last])] if last in gens else [])
    else:
        res = OrderedDict(gens)
        res.update(segs)
    res[f'G{S}'] = seg
    return res


_INFER_STREAMS = max(1, int(os.environ.get('DL_INFER_STREAMS', '3')))


def _infer_streams(device):
    """DL_INFER_STREAMS=N (default 3 since round 5; 1 = everything on the current stream): the independent chains G_i -> GS_i of the DeepLIIF inference DAG on N HIP streams of the calling thread
    (the training step's branch streams, models.BaseModel._branch_streams, applied to run_dask's DAG)"""
    if _INFER_STREAMS <= 1 or device.type != 'cuda':
        return None
    st = ops.WS._thread_state()
    key = ('infer_streams', device.index)
    if key not in st:
        st[key] = [torch.cuda.Stream(device) for _ in range(_INFER_STREAMS)]
        ops.WS.branch_streams_on(st[key])
    return st[key]


def _run_deepliif_dag_on_streams(ctx, x, nets, opt, seg_map, weights, seg_only, streams):
    """the DeepLIIF branch of run_generators_engine with chain i (G_i(tile) -> GS_i(G_i(tile))) on stream i mod N and GS_0(tile) on the next one; the weighted
    sum and everything after it on the calling stream, behind a join.  Same kernels on the same operands: results are bit-identical to the one-stream DAG."""
    M = opt.modalities_no
    S, off = _get(opt, 'mod_id_seg', 'S'), int(_get(opt, 'input_id', 0))
    main = torch.cuda.current_stream(x.t.device)
    for s in streams:
        s.wait_stream(main)
    gens, segs = OrderedDict(), OrderedDict()
    i = -1
    for i, (k, v) in enumerate(seg_map.items()):
        with torch.cuda.stream(streams[i % len(streams)]):
            gens[k] = nets[k].run(ctx, x)
            segs[v] = nets[v].run(ctx, gens[k])
    names = _get(opt, 'modalities_names', [])
    if 'Marker' in names:
        km = f'G{names.index("Marker")}'
        if km not in gens and km in nets:
            with torch.cuda.stream(streams[(i + 1) % len(streams)]):
                gens[km] = nets[km].run(ctx, x)
            i += 1
    base = f'G{S}{off}'
    if weights[base] != 0:
        with torch.cuda.stream(streams[(i + 1) % len(streams)]):
            segs[base] = nets[base].run(ctx, x)
    for s in streams:
        main.wait_stream(s)
    keys = list(segs.keys())
    seg = E.weighted_sum(ctx, [segs[k] for k in keys], [float(weights[k]) for k in keys])
    if seg_only and M > 0:
        last = f'G{M}'
        res = OrderedDict([(last, gens[last])] if last in gens else [])
    else:
        res = OrderedDict(gens)
        res.update(segs)
    res[f'G{S}'] = seg
    return res


def run_generators(ts: torch.Tensor, nets, opt, seg_only=False, mod_only=False, seg_weights=None) -> 'OrderedDict[str, torch.Tensor]':
    """run_generators_engine on a [N, C, H, W] fp32 tensor; returns name -> [N, 3, H, W] fp32 tensors (run_dask(output_tensor=True))."""
    first = next(iter(nets.values()))
    device = next(first.parameters()).device
    prec = E.Precision.get(first.precision)
    if _FP16_NAN:
        fp16_check()              # the batches of earlier calls have long finished: no stall
    with ops.half_mode(prec.half):
        x = E.to_engine(ts.to(device), prec)
        return OrderedDict((k, E.from_engine(v)) for k, v in run_generators_engine(x, nets, opt, seg_only, mod_only, seg_weights).items())


def run_dask(img, model_path=None, nets=None, eager_mode=False, opt=None, seg_only=False, mod_only=False, seg_weights=None, use_dask=True,
             output_tensor=False):
    """Same call shape as the reference's run_dask (deepliif/models/__init__.py:258-388; `use_dask` is accepted and ignored: branch
    concurrency is the engine's business).  `img` is a PIL image, a list of PIL images (multi-input models, :276-279) or a
    [N, C, H, W] tensor; PIL in -> dict of PIL images of the first tile."""
    assert model_path is not None or nets is not None, 'Provide either the model path or the networks object.'
    if nets is None:
        nets = init_nets(os.getenv('DEEPLIIF_MODEL_DIR', model_path), eager_mode, opt)
    if opt is None:
        opt = get_opt(os.getenv('DEEPLIIF_MODEL_DIR', model_path))
    if isinstance(img, torch.Tensor):
        ts = img
    elif _get(opt, 'input_no', 1) > 1 or _get(opt, 'model', 'DeepLIIF') == 'SDG':
        ts = torch.cat([transform(i, _get(opt, 'scale_size', None)) for i in img], dim=1)
    else:
        ts = transform(img, _get(opt, 'scale_size', None))
    res = run_generators(ts, nets, opt, seg_only, mod_only, seg_weights)
    if output_tensor:
        return res
    fp16_check()
    return {k: tensor_to_pil(v) for k, v in res.items()}


# -------------------------------------------------------------------------------------------------------------
# tiled inference: inference() / infer_region()   (deepliif/models/__init__.py:399-579)
# -------------------------------------------------------------------------------------------------------------
def empty_tile_colors(opt, seg_only=False, mod_only=False) -> 'OrderedDict[str, tuple]':
    """What run_wrapper returns for a tile that is_empty (deepliif/models/__init__.py:399-461): key -> constant colour."""
    model, M = _get(opt, 'model', 'DeepLIIF'), opt.modalities_no
    black = (0, 0, 0)
    if model in ('DeepLIIFExt', 'SDG'):
        res = OrderedDict((f'G_{i}', black) for i in range(1, M + 1))
        res.update((f'GS_{i}', black) for i in range(1, M + 1))
        return res
    if model == 'CycleGAN':                                                                 # models/__init__.py:453-459: black tiles under the net names
        return OrderedDict((k, black) for k in generator_names(opt)[0])
    if model not in ('DeepLIIF', 'DeepLIIFKD'):
        raise NotImplementedError(f'run_wrapper is not implemented for model {model}')
    S, bg = _get(opt, 'mod_id_seg', 'S'), _get(opt, 'background_colors', None)
    if bg is None:
        bg = [(201, 211, 208), (10, 10, 10), (0, 0, 0), (10, 10, 10)]                      # options/__init__.py:118-122
    if seg_only:
        res = OrderedDict()
        if M >= 1:
            res[f'G{M}'] = tuple(bg[-1])
        res[f'G{S}'] = black
    elif mod_only or not opt.seg_gen:
        res = OrderedDict((f'G{i + 1}', tuple(bg[i])) for i in range(M))
    else:
        res = OrderedDict((f'G{i + 1}', tuple(bg[i])) for i in range(M))
        res[f'G{S}'] = black
        first = 1 if int(_get(opt, 'input_id', 0)) == 1 else 0
        res.update((f'G{S}{first + i}', black) for i in range(M + 1))
    res.pop('G0', None)
    return res


def infer_region(images, tile_size, overlap_size, nets, opt, seg_only=False, mod_only=False, seg_weights=None, batch_size=8, rank=0, world=1,
                 limit_tiles=None):
    """Tile loop of inference() (deepliif/models/__init__.py:496-500) for uint8 RGB image(s) [H, W, 3] resident in HBM, entirely on
    the GPU: crop + transform (dl_tile_gather_u8), is_empty (dl_tile_gray_stats_u8), the generator DAG on batches of `batch_size`
    tiles with per-sample normalisation, tensor2im + stitch (dl_tile_paste_u8).
    Tile-parallel over `world` ranks (BASELINE configs[4]): rank r owns a contiguous band of tile rows (tiling.split_rows) and
    returns ({key: uint8 [band rows, W, 3]}, (y0, y1)); the bands of all ranks concatenate to the full result images.
    limit_tiles (measurement only): stop after that many non-empty tiles of the band."""
    from .tiling import RegionTiler, TilePlan, split_rows
    scale = _get(opt, 'scale_size', tile_size)
    if tile_size != scale:
        raise NotImplementedError(f'infer_region runs tiles at the network resolution (tile_size == scale_size == {scale}); '
                                  f'inference() resamples other tile sizes with PIL on the host')
    first = next(iter(nets.values()))
    prec = E.Precision.get(first.precision)
    h, w = int(images[0].shape[0]), int(images[0].shape[1])
    n_rows = len(TilePlan(w, h, tile_size, overlap_size).ys)
    with ops.half_mode(prec.half):
        tiler = RegionTiler(images, tile_size, overlap_size, rows=split_rows(n_rows, world)[rank])
        if len(tiler) == 0:
            return {}, tiler.band
        empty = tiler.empty_mask()
        ids = np.array(tiler.tile_ids)
        seg_only, mod_only = _wrapper_flags(opt, seg_only, mod_only)
        colors = empty_tile_colors(opt, seg_only, mod_only)
        if empty.any():
            for k, c in colors.items():
                tiler.paste(k, None, ids[empty].tolist(), const_rgb=c)
        work = ids[~empty].tolist()
        if limit_tiles is not None:
            work = work[:limit_tiles]
        cp = E.cpad(3 * len(images))
        for s in range(0, len(work), batch_size):
            chunk = work[s:s + batch_size]
            x = E.Act(tiler.gather(chunk, prec.dtype, cp), 3 * len(images))
            for k, a in run_generators_engine(x, nets, opt, seg_only, mod_only, seg_weights).items():
                tiler.paste(k, a.t, chunk)
        fp16_check()
        return tiler.results(), tiler.band


def gather_bands(local: Dict[str, torch.Tensor], band, height: int, width: int, keys: List[str], rank: int, world: int):
    """Concatenate the per-rank result bands on rank 0 (point-to-point sends of contiguous uint8 rows; every rank knows every band
    from the plan, so no size exchange).  Returns {key: uint8 [height, width, 3]} on rank 0, None elsewhere."""
    import torch.distributed as dist
    if world == 1:
        return local
    dev = next(iter(local.values())).device if local else torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    bands = [None] * world
    dist.all_gather_object(bands, (int(band[0]), int(band[1])))
    out = None
    if rank == 0:
        out = {k: torch.zeros((height, width, 3), dtype=torch.uint8, device=dev) for k in keys}
    for k in keys:
        for r in range(world):
            y0, y1 = bands[r]
            if y1 <= y0:
                continue
            if rank == 0:
                if r == 0:
                    out[k][y0:y1] = local[k] if k in local else 0
                else:
                    dist.recv(out[k][y0:y1], src=r)
            elif r == rank:
                dist.send(local[k].contiguous() if k in local else torch.zeros((y1 - y0, width, 3), dtype=torch.uint8, device=dev), dst=0)
    return out


def _to_u8_device(img, device) -> torch.Tensor:
    a = np.asarray(img.convert('RGB') if img.mode != 'RGB' else img)
    return torch.from_numpy(np.array(a, dtype=np.uint8, order="C")).to(device)


def _result_names(opt, results, seg_only, mod_only, return_seg_intermediate):
    """Result-key -> caller-facing name mapping of inference() (deepliif/models/__init__.py:502-579)."""
    model, M = _get(opt, 'model', 'DeepLIIF'), opt.modalities_no
    if model == 'DeepLIIFExt':
        out = OrderedDict((f'mod{i}', f'G_{i}') for i in range(1, M + 1))
        if opt.seg_gen:
            out.update((f'Seg{i}', f'GS_{i}') for i in range(1, M + 1))
        return out
    if model == 'SDG':
        return OrderedDict((f'mod{i}', f'G_{i}') for i in range(1, M + 1))
    if model not in ('DeepLIIF', 'DeepLIIFKD'):
        return OrderedDict((k, k) for k in results)
    S, input_no = _get(opt, 'mod_id_seg', 'S'), _get(opt, 'input_no', 1)
    names = list(_get(opt, 'modalities_names', []))
    mods = [f'mod{i + 1}' for i in range(M)]
    if mods != names[input_no:]:
        mods = [f'mod{i + 1}-{n}' for i, n in enumerate(names[input_no:])]
    ids = OrderedDict((n, f'G{i + 1}') for i, n in enumerate(mods))
    seg_ids = OrderedDict()
    if opt.seg_gen:
        segm = [f'mod{i}' for i in range(M + 1)]
        if segm != names:
            segm = [f'mod{i}-{n}' for i, n in enumerate(names)]
        first = 0 if f'G{S}0' in results else 1
        seg_ids = OrderedDict((n, f'G{S}{first + i}') for i, n in enumerate(segm))
    if not mod_only and opt.seg_gen:
        ids['Seg'] = f'G{S}'
    if seg_only:
        out = OrderedDict([('Seg', ids['Seg'])])
        marker = next((n for n in ids if n.endswith('Marker')), None)          # find_marker_key (models/__init__.py)
        if marker is not None:
            out[marker] = ids[marker]
        return out
    out = OrderedDict(ids)
    if opt.seg_gen and return_seg_intermediate:
        out.update((f'{n}_s', k) for n, k in seg_ids.items())
    return out


def inference(img, tile_size, overlap_size, model_path, use_torchserve=False, eager_mode=False, color_dapi=False, color_marker=False, opt=None,
              return_seg_intermediate=False, seg_only=False, mod_only=False, seg_weights=None, opt_args={}, nets=None, batch_size=8, rank=0, world=1):
    """Drop-in for deepliif.models.inference (deepliif/models/__init__.py:464-579): PIL image in, dict name -> PIL image out.
    The tile loop runs on the GPU in batches (infer_region); `nets` / `batch_size` are extensions (default: init_nets(model_path)).
    rank / world (extension, BASELINE configs[4]): `world` ranks call this with the SAME image; each infers its band of tile rows, rank 0 gathers the
    bands (gather_bands) and returns the dict, the other ranks return None.  Everything else -- seg_gen guard, input_no / SDG split, result names --
    is this one code path whatever the world size.  With tile_size != scale_size (PIL resampling on the host) rank 0 does the whole image."""
    from PIL import Image
    if use_torchserve:
        raise NotImplementedError('the TorchServe client route is not part of the MI355X engine (use the in-process engine)')
    if not opt:
        opt = get_opt(model_path)
    for k, v in opt_args.items():
        setattr(opt, k, v)
    if hasattr(opt, 'seg_gen') and opt.seg_gen is False and (seg_only or return_seg_intermediate):
        seg_only = return_seg_intermediate = False
        print('option seg_gen is False, disabled seg_only and return_seg_intermediate')
    if nets is None:
        nets = init_nets(os.getenv('DEEPLIIF_MODEL_DIR', model_path), eager_mode, opt)
    device = next(next(iter(nets.values())).parameters()).device
    input_no = _get(opt, 'input_no', 1)
    if input_no > 1 or _get(opt, 'model', 'DeepLIIF') == 'SDG':
        w, h = int(img.width / input_no), img.height
        origs = [img.crop((w * i, 0, w * (i + 1), h)) for i in range(input_no)]
    else:
        origs = [img]
    scale = _get(opt, 'scale_size', tile_size)
    if tile_size == scale:
        bands, band = infer_region([_to_u8_device(o, device) for o in origs], tile_size, overlap_size, nets, opt, seg_only, mod_only, seg_weights,
                                   batch_size, rank=rank, world=world)
        if world > 1:
            keys = sorted(empty_tile_colors(opt, *_wrapper_flags(opt, seg_only, mod_only)))
            bands = gather_bands(bands, band, origs[0].height, origs[0].width, keys, rank, world)
            if bands is None:
                return None
        results = {k: Image.fromarray(v.cpu().numpy()) for k, v in bands.items()}
    else:
        if rank != 0:
            return None
        results = _inference_resampled(origs, tile_size, overlap_size, nets, opt, seg_only, mod_only, seg_weights, batch_size, scale)
    names = _result_names(opt, results, seg_only, mod_only, return_seg_intermediate)
    return {n: results[k] for n, k in names.items()}


def _inference_resampled(origs, tile_size, overlap_size, nets, opt, seg_only, mod_only, seg_weights, batch_size, scale):
    """tile_size != scale_size: the reference resamples every tile to the network resolution and every result tile back with PIL
    (run_dask :276-280, InferenceTiler.stitch :291-292).  Resampling is image-format plumbing and stays PIL on the host; the tiles
    still go through the generators in batches."""
    from PIL import Image
    from .tiling import TilePlan, gray_stats_empty
    w, h = origs[0].size
    plan = TilePlan(w, h, tile_size, overlap_size)
    device = next(next(iter(nets.values())).parameters()).device
    srcs = [np.asarray(o.convert('RGB')) for o in origs]
    if (plan.image_width, plan.image_height) != (w, h):             # mirror extension (util/__init__.py:196-211)
        def ext(a):
            while a.shape[1] < plan.image_width:
                a = np.concatenate([a, a[:, ::-1]], axis=1)
            while a.shape[0] < plan.image_height:
                a = np.concatenate([a, a[::-1]], axis=0)
            return a[:plan.image_height, :plan.image_width]
        srcs = [ext(a) for a in srcs]
    rects, origins = plan.paste_rects(), plan.origins
    seg_only, mod_only = _wrapper_flags(opt, seg_only, mod_only)
    colors = empty_tile_colors(opt, seg_only, mod_only)
    out: Dict[str, np.ndarray] = {}

    def put(key, tile_u8, t):
        if key not in out:
            out[key] = np.zeros((plan.image_height, plan.image_width, 3), dtype=np.uint8)
        l, tp, rw, rh, px, py = rects[t]
        out[key][py:py + rh, px:px + rw] = tile_u8[tp:tp + rh, l:l + rw]

    def gray_empty(a):
        g = ((a[..., 0].astype(np.uint32) * 19595 + a[..., 1].astype(np.uint32) * 38470 + a[..., 2].astype(np.uint32) * 7471 + 0x8000) >> 16).astype(np.int64)
        v = g[(g != 0) & (g != 255)]
        return bool(gray_stats_empty(np.array([[v.size, v.sum(), (v * v).sum()]], dtype=np.int64))[0])

    pending = []
    for t, (x, y) in enumerate(origins):
        crops = [a[y:y + plan.patch_size, x:x + plan.patch_size] for a in srcs]
        if all(gray_empty(c) for c in crops):
            for k, c in colors.items():
                put(k, np.broadcast_to(np.array(c, dtype=np.uint8), (tile_size, tile_size, 3)), t)
            continue
        pending.append((t, torch.cat([transform(Image.fromarray(c), scale) for c in crops], dim=1)))
    for s in range(0, len(pending), batch_size):
        chunk = pending[s:s + batch_size]
        res = run_generators(torch.cat([ts for _, ts in chunk]).to(device), nets, opt, seg_only, mod_only, seg_weights)
        for k, v in res.items():
            for i, (t, _) in enumerate(chunk):
                tile = tensor_to_pil(v[i:i + 1]).resize((tile_size, tile_size))
                put(k, np.asarray(tile), t)
    fp16_check()
    return {k: Image.fromarray(v[:h, :w]) for k, v in out.items()}


def find_marker_key(dictionary):
    """deepliif/models/__init__.py:950-954"""
    for key in dictionary:
        if key.endswith('Marker'):
            return key
    return None


def postprocess(orig, images, tile_size, model, seg_thresh=120, size_thresh='default', marker_thresh=None, size_thresh_upper=None):
    """Drop-in for deepliif.models.postprocess (deepliif/models/__init__.py:582-610): the stitched Seg (+ Marker) images -> overlay / refined
    images and the scoring dictionary, through the GPU post-processing (deepliif_amd/postprocessing.py) instead of the numba loops."""
    from PIL import Image
    from .postprocessing import compute_final_results
    if model in ('DeepLIIF', 'DeepLIIFKD'):
        resolution = '40x' if tile_size > 384 else ('20x' if tile_size > 192 else '10x')
        mk = find_marker_key(images)
        overlay, refined, scoring = compute_final_results(orig, images['Seg'], images.get(mk) if mk is not None else None, resolution,
                                                          size_thresh, marker_thresh, size_thresh_upper, seg_thresh)
        return {'SegOverlaid': Image.fromarray(overlay), 'SegRefined': Image.fromarray(refined)}, scoring
    if model in ('DeepLIIFExt', 'SDG'):
        resolution = '40x' if tile_size > 768 else ('20x' if tile_size > 384 else '10x')
        processed, scoring = {}, {}
        for name in list(images.keys()):
            if 'Seg' in name:
                overlay, refined, score = compute_final_results(orig, images[name], None, resolution, size_thresh, marker_thresh, size_thresh_upper,
                                                                seg_thresh)
                processed[name + '_Overlaid'], processed[name + '_Refined'] = Image.fromarray(overlay), Image.fromarray(refined)
                scoring[name] = score
        return processed, scoring
    raise Exception(f'postprocess() not implemented for model {model}')


def infer_modalities(img, tile_size, model_dir, eager_mode=False, color_dapi=False, color_marker=False, opt=None, return_seg_intermediate=False,
                     seg_only=False, mod_only=False, seg_weights=None, nets=None, batch_size=8, rank=0, world=1):
    """Drop-in for deepliif.models.infer_modalities (deepliif/models/__init__.py:613-660): inference() with overlap tile_size // 16, then
    postprocess() when the model has a segmentation branch.  -> (images, scoring).  rank / world: see inference(); ranks other than 0 get (None, None)."""
    if opt is None:
        opt = get_opt(model_dir)
        opt.use_dp = False
    images = inference(img, tile_size=tile_size, overlap_size=tile_size // 16, model_path=model_dir, eager_mode=eager_mode, color_dapi=color_dapi,
                       color_marker=color_marker, opt=opt, return_seg_intermediate=return_seg_intermediate, seg_only=seg_only, mod_only=mod_only,
                       seg_weights=seg_weights, nets=nets, batch_size=batch_size, rank=rank, world=world)
    if images is None:
        return None, None
    if not hasattr(opt, 'seg_gen') or opt.seg_gen:
        if not mod_only:
            post_images, scoring = postprocess(img, images, tile_size, opt.model)
            images = {**images, **post_images}
            if seg_only:
                for name in [k for k in images.keys() if 'Seg' not in k]:
                    del images[name]
            return images, scoring
        return images, None
    return images, None
