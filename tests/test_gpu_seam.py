"""The checkpoint-directory / inference seam on the MI355X against the REFERENCE's own outputs for the same directories
(tests/golden/make_golden_seam.py): init_nets(model_dir) -> run_dask(PIL) / inference(PIL) -> uint8 images, for DeepLIIF, DeepLIIFExt
and SDG.  fp32 policy: float outputs agree to ~1e-5, so the truncated uint8 bytes differ by at most one step in a fraction of a
percent of the pixels (an exact-equality test would only measure how many float values sit on an integer boundary)."""
import numpy as np
import pytest
import torch
from PIL import Image

from golden_util import synth_image
from seam_util import Z, build_checkpoint_dir, close_u8, serialize_checkpoint_dir

pytestmark = pytest.mark.gpu


def _opt(mdir, precision):
    from deepliif_amd import inference as I
    I._NETS_CACHE.clear()
    opt = I.get_opt(mdir)
    opt.ngf, opt.precision, opt.gpu_ids = 8, precision, [0]
    return opt


def _images():
    img = Image.fromarray(synth_image(150, 100, 31))
    a = np.asarray(img).copy()
    a[:, :70] = 252
    return img, Image.fromarray(a)


def test_deepliif_checkpoint_dir_to_pil_bytes(tmp_path):
    from deepliif_amd import inference as I
    mdir = build_checkpoint_dir(tmp_path, 'dl_m2')
    opt = _opt(mdir, 'fp32')
    img, img2 = _images()
    tile = img.crop((0, 0, 64, 64))
    nets = I.init_nets(mdir, eager_mode=True, opt=opt)
    assert list(nets) == ['G1', 'G2', 'GS0', 'GS1', 'GS2'] and all(next(n.parameters()).is_cuda for n in nets.values())
    res = I.run_dask(tile, model_path=mdir, eager_mode=True, opt=opt)
    assert list(res) == Z['dl_m2/run_dask_keys'].tolist()
    worst = max(close_u8(v, Z[f'dl_m2/run_dask/{k}'], 0.005) for k, v in res.items())
    for name, kw in (('inf', {}), ('inf_seginter', dict(return_seg_intermediate=True)), ('inf_modonly', dict(mod_only=True)), ('inf_segonly', dict(seg_only=True))):
        r = I.inference(img2, 64, 4, mdir, eager_mode=True, opt=opt, **kw)
        assert list(r) == Z[f'dl_m2/{name}_keys'].tolist(), name
        for k, v in r.items():
            exp = Z[f'dl_m2/{name}/{k}']
            worst = max(worst, close_u8(v, exp, 0.005))
            assert np.array_equal(np.asarray(v)[:, :60], exp[:, :60]), (name, k)
    print('fp32 policy: worst fraction of pixels one uint8 step away from the reference:', worst)


@pytest.mark.parametrize('tag', ['ext_m2', 'sdg_m2_in2'])
def test_ext_and_sdg_checkpoint_dir_to_pil_bytes(tmp_path, tag):
    from deepliif_amd import inference as I
    mdir = build_checkpoint_dir(tmp_path, tag)
    opt = _opt(mdir, 'fp32')
    img, img2 = _images()
    if tag == 'ext_m2':
        res = I.run_dask(img.crop((0, 0, 64, 64)), model_path=mdir, eager_mode=True, opt=opt)
        assert list(res) == Z[f'{tag}/run_dask_keys'].tolist()
        for k, v in res.items():
            close_u8(v, Z[f'{tag}/run_dask/{k}'], 0.005)
        src = img2
    else:
        src = Image.fromarray(np.concatenate([np.asarray(img2), synth_image(150, 100, 32)], axis=1))
    r = I.inference(src, 64, 4, mdir, eager_mode=True, opt=opt)
    assert list(r) == Z[f'{tag}/inf_keys'].tolist()
    for k, v in r.items():
        close_u8(v, Z[f'{tag}/inf/{k}'], 0.005)


def test_bf16_policy_stays_within_its_documented_distance(tmp_path):
    """the throughput policy on the same directory: bf16 activations put the images a few uint8 steps away (DESIGN.md precision table)"""
    from deepliif_amd import inference as I
    mdir = build_checkpoint_dir(tmp_path, 'dl_m2')
    opt = _opt(mdir, 'bf16')
    _, img2 = _images()
    r = I.inference(img2, 64, 4, mdir, eager_mode=True, opt=opt)
    for k, v in r.items():
        d = np.abs(np.asarray(v).astype(int) - Z[f'dl_m2/inf/{k}'].astype(int))
        assert d.max() <= 12 and d.mean() < 1.5, (k, int(d.max()), float(d.mean()))


def test_model_on_second_argument_device_uses_that_devices_stream():
    """kernels must launch on the stream of the tensors' device, whatever the current device is (one visible GPU here: the
    check is that selecting the device explicitly, from a thread whose current stream is a side stream, still works)"""
    from deepliif_amd import inference as I
    import types
    opt = types.SimpleNamespace(model='DeepLIIF', modalities_no=1, seg_gen=False, mod_id_seg=None, input_id=0, input_nc=3, output_nc=3, ngf=8,
                                norm='batch', padding='zero', net_g='resnet_9blocks', net_gs='unet_64', input_no=1, modalities_names=['input1', 'mod1'], gpu_ids=[0])
    torch.manual_seed(0)
    nets = I.build_generators(opt, torch.device('cuda', 0), 'fp32')
    x = torch.rand(2, 3, 64, 64) * 2 - 1
    ref = nets['G1'](x.cuda())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        out = nets['G1'](x.cuda())
    side.synchronize()
    assert torch.equal(out, ref)


def test_infer_modalities_adds_postprocessed_images_and_scoring(tmp_path):
    """infer_modalities = inference() + postprocess() (deepliif/models/__init__.py:582-660): the glue picks Seg / Marker, derives the resolution
    from the tile size and returns SegOverlaid / SegRefined + the scoring dictionary; the post-processing of the INFERRED images must equal the
    pinned oracle's post-processing of the same images byte for byte."""
    from deepliif_amd import inference as I
    from oracle import postprocess_oracle as PO
    mdir = build_checkpoint_dir(tmp_path, 'dl_m2')
    opt = _opt(mdir, 'fp32')
    _, img2 = _images()
    images, scoring = I.infer_modalities(img2, 64, mdir, eager_mode=True, opt=opt)
    assert 'SegOverlaid' in images and 'SegRefined' in images and images['SegOverlaid'].size == img2.size
    mk = I.find_marker_key(images)
    o_overlay, o_refined, o_scoring = PO.compute_final_results(np.asarray(img2), np.asarray(images['Seg']), np.asarray(images[mk]) if mk else None, '10x')
    assert np.array_equal(np.asarray(images['SegOverlaid']), o_overlay) and np.array_equal(np.asarray(images['SegRefined']), o_refined)
    assert scoring == o_scoring
    only_seg, _ = I.infer_modalities(img2, 64, mdir, eager_mode=True, opt=opt, seg_only=True)
    assert all('Seg' in k for k in only_seg)
    mods, none = I.infer_modalities(img2, 64, mdir, eager_mode=True, opt=opt, mod_only=True)
    assert none is None and 'SegOverlaid' not in mods


def test_serialized_model_directory_default_route_on_the_gpu(tmp_path):
    """the reference's DEFAULT inference route (eager_mode=False, models/__init__.py:216-219) on a directory that holds only `<name>.pt` TorchScript
    files + train_opt.txt: used as weight containers, same uint8 bytes as the checkpoint directory they were serialized from"""
    import os
    from deepliif_amd import inference as I
    mdir = build_checkpoint_dir(tmp_path, 'dl_m2')
    sdir = serialize_checkpoint_dir(mdir, str(tmp_path / 'serialized'))
    assert not [f for f in os.listdir(sdir) if f.endswith('.pth')]
    img, img2 = _images()
    r_pt = I.inference(img2, 64, 4, sdir, opt=_opt(sdir, 'fp32'))
    r_pth = I.inference(img2, 64, 4, mdir, eager_mode=True, opt=_opt(mdir, 'fp32'))
    assert list(r_pt) == Z['dl_m2/inf_keys'].tolist()
    for k, v in r_pt.items():
        close_u8(v, Z[f'dl_m2/inf/{k}'], 0.005)
        assert np.array_equal(np.asarray(v), np.asarray(r_pth[k])), k


@pytest.mark.parametrize('kind', ['resnet', 'unet'])
def test_torchserve_handler_contract(kind):
    """model-server/resnet.py:4-14, unet.py:4-13: the served classes construct the generators with EXACTLY these keyword arguments (no-argument
    subclasses), TorchServe loads a state_dict into them, and net_handler.py:7-16 puts the module in train() mode and feeds it
    torch.load(BytesIO(body)).to(device).  In that mode the reference's module tree has Dropout(0.5) active and BatchNorm on batch statistics with
    running-statistic updates; with use_dropout=False the train()-mode output of one tile equals the eval()-mode output (statistics of that tile)."""
    from io import BytesIO
    from deepliif_amd.networks import ResnetGenerator, UnetGenerator, get_norm_layer
    from oracle import deepliif_oracle as O

    class Resnet(ResnetGenerator):
        def __init__(self, use_dropout=True):
            super(Resnet, self).__init__(input_nc=3, output_nc=3, ngf=64, norm_layer=get_norm_layer(norm_type='batch'), use_dropout=use_dropout, n_blocks=9,
                                         padding_type='zero')

    class Unet(UnetGenerator):
        def __init__(self, use_dropout=True):
            super(Unet, self).__init__(input_nc=3, output_nc=3, num_downs=9, ngf=64, norm_layer=get_norm_layer(norm_type='batch'), use_dropout=use_dropout)

    cls, arch, size = (Resnet, 'resnet_9blocks', 128) if kind == 'resnet' else (Unet, 'unet_512', 512)
    sd = O.random_state_dict(arch, 3, 3, 64, 'batch', 'zero', 4, generator=torch.Generator().manual_seed(11))
    body = BytesIO()
    torch.save(torch.rand(1, 3, size, size, generator=torch.Generator().manual_seed(12)) * 2 - 1, body)

    def serve(net):                                      # NetHandler.preprocess + BaseHandler.inference
        net.train()
        with torch.no_grad():
            return net(torch.load(BytesIO(body.getvalue())).to('cuda'))

    net = cls().set_precision('fp32')
    # with use_dropout=True the ResnetBlock's Sequential holds a Dropout at index 3 (networks.py:490-494): its second conv / norm are
    # conv_block.4 / .5 in a served checkpoint, .3 / .4 in the dropout-free state_dict the oracle generates
    sd_drop = {(k.replace('conv_block.4.', 'conv_block.5.').replace('conv_block.3.', 'conv_block.4.') if kind == 'resnet' else k): v for k, v in sd.items()}
    net.load_state_dict(sd_drop, strict=True)
    net.to('cuda')
    bn = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    rm0 = [m.running_mean.clone() for m in bn]
    y1, y2 = serve(net), serve(net)
    assert y1.shape == (1, 3, size, size) and y1.device.type == 'cuda' and bool(torch.isfinite(y1).all()) and float(y1.abs().max()) <= 1.0
    assert not torch.equal(y1, y2), 'Dropout(0.5) must be active in train() mode (use_dropout=True)'
    assert all(int(m.num_batches_tracked) == 2 for m in bn)
    moved = sum(float((m.running_mean - r).abs().max()) > 0 for m, r in zip(bn, rm0))
    assert moved >= len(bn) - 1, 'BatchNorm running statistics are updated in train() mode (%d of %d moved)' % (moved, len(bn))   # a 1x1 map has mean == input: may not move
    # without dropout: train()-mode serving of ONE tile == eval()-mode inference of that tile (both normalise with the tile's own statistics),
    # and both equal the CPU oracle's forward of the same state_dict
    net2 = cls(use_dropout=False).set_precision('fp32')
    net2.load_state_dict(sd, strict=True)
    net2.to('cuda')
    yt = serve(net2)
    net2.eval()
    with torch.no_grad():
        ye = net2(torch.load(BytesIO(body.getvalue())).to('cuda'))
    assert torch.equal(yt, ye)
    exp = O.run_generator(arch, sd, torch.load(BytesIO(body.getvalue())), 'batch', 'zero')
    err = float((yt.cpu() - exp).abs().max() / exp.abs().max())
    assert err < 1e-3, err
