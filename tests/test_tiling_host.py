"""Host side of the tiled inference path on CPU (emulated ops backend): the closed-form TilePlan geometry, RegionTiler's
crop / is_empty / stitch calls and the whole infer_region() loop against the oracle (tiler_oracle = the reference's sequential
InferenceTiler semantics, pinned by test_oracle_tiler.py; deepliif_oracle = the reference networks).  Bit-exact: integer work."""
import types

import numpy as np
import pytest
import torch

import fake_backend
from deepliif_amd import inference as I
from deepliif_amd import tiling as TL
from golden_util import synth_image
from oracle import deepliif_oracle as O
from oracle import tiler_oracle as T


@pytest.fixture(autouse=True)
def _fake():
    fake_backend.install()
    yield
    fake_backend.uninstall()


GEOMS = [(300, 200, 128, 8, 0), (257, 130, 128, 16, 0), (100, 70, 128, 8, 0), (300, 90, 128, 8, 0), (300, 200, 128, 8, 4), (128, 128, 128, 8, 0),
         (129, 128, 128, 0, 0), (1381, 949, 512, 32, 0), (640, 512, 512, 32, 0), (513, 700, 256, 16, 0), (97, 333, 64, 4, 2), (5000, 3337, 512, 32, 0)]


@pytest.mark.parametrize('w,h,tile,overlap,pad', GEOMS)
def test_plan_reproduces_reference_iteration_and_paste_order(w, h, tile, overlap, pad):
    """unique origins + one disjoint rectangle per tile == the reference's visit sequence + up to nine ordered pastes per visit"""
    g = T.TilerGeometry(w, h, tile, overlap, pad)
    p = TL.TilePlan(w, h, tile, overlap, pad)
    org = p.origins
    assert [tuple(org[i]) for i in p.visit_order()] == g.positions()
    # ownership map: the reference's last writer per pixel vs the plan's single rectangle per tile
    owner_ref = -np.ones((g.image_height, g.image_width), dtype=np.int64)
    local_ref = np.zeros((g.image_height, g.image_width, 2), dtype=np.int64)
    order = p.visit_order()
    for (x, y), t in zip(g.positions(), order):
        for (l, tp, r, b), (px, py) in g.pastes(x, y):
            owner_ref[py:py + b - tp, px:px + r - l] = t
            yy, xx = np.mgrid[tp:b, l:r]
            local_ref[py:py + b - tp, px:px + r - l, 0] = xx
            local_ref[py:py + b - tp, px:px + r - l, 1] = yy
    owner = -np.ones_like(owner_ref)
    local = np.zeros_like(local_ref)
    for t, (l, tp, rw, rh, px, py) in enumerate(p.paste_rects()):
        assert (owner[py:py + rh, px:px + rw] == -1).all(), 'rectangles must be disjoint'
        owner[py:py + rh, px:px + rw] = t
        yy, xx = np.mgrid[tp:tp + rh, l:l + rw]
        local[py:py + rh, px:px + rw, 0] = xx
        local[py:py + rh, px:px + rw, 1] = yy
    assert (owner >= 0).all(), 'the rectangles cover the image'
    assert np.array_equal(owner, owner_ref)
    assert np.array_equal(local, local_ref)


def test_plan_argument_errors():
    for args in ((100, 100, 0, 0, 0), (100, 100, 64, -1, 0), (100, 100, 64, 0, -1), (300, 300, 64, 32, 0)):
        with pytest.raises(ValueError):
            TL.TilePlan(*args)


def _fake_out(x):
    """two 'network outputs' per tile batch (engine layout in, engine layout out) that keep values inside [-1, 1]"""
    return {'A': -x, 'B': x * 0.5 + 0.25}


@pytest.mark.parametrize('w,h,tile,overlap,pad,world', [(300, 200, 128, 8, 0, 1), (300, 200, 128, 8, 4, 1), (100, 70, 128, 8, 0, 1), (513, 700, 128, 8, 0, 3),
                                                         (300, 200, 128, 8, 0, 5)])
def test_region_tiler_matches_oracle(w, h, tile, overlap, pad, world):
    img = synth_image(w, h, 3)
    g = T.TilerGeometry(w, h, tile, overlap, pad)
    ext = T.mirror_to_patch(img, g.patch_size)
    pos = g.positions()
    ref_tiles = [T.extract_tile(ext, g, x, y) for x, y in pos]
    ref_in = [T.transform(t) for t in ref_tiles]                                      # [1, 3, T, T]
    ref_out = [{k: T.tensor2im(np.transpose(v.numpy(), (0, 3, 1, 2))) for k, v in _fake_out(torch.from_numpy(np.transpose(x, (0, 2, 3, 1)))).items()}
               for x in ref_in]
    expect = T.stitch(g, pos, ref_out)
    n_rows = len(TL.TilePlan(w, h, tile, overlap, pad).ys)
    full = {}
    for r, rows in enumerate(TL.split_rows(n_rows, world)):
        rt = TL.RegionTiler([torch.from_numpy(img)], tile, overlap, pad, rows=rows)
        if len(rt) == 0:
            continue
        x = rt.gather(rt.tile_ids, torch.float32, 8)
        # crop + transform, against the oracle's tile for the same origin
        by_origin = {p_: t for p_, t in zip(pos, ref_in)}
        for i, tid in enumerate(rt.tile_ids):
            o = tuple(rt.plan.origins[tid])
            assert np.array_equal(x[i, :, :, :3].numpy(), np.transpose(by_origin[o][0], (1, 2, 0)))
            assert (x[i, :, :, 3:] == 0).all()
        for k, v in _fake_out(x).items():
            rt.paste(k, v.contiguous(), rt.tile_ids)
        for k, band in rt.results().items():
            full.setdefault(k, np.zeros_like(expect[k]))[rt.band[0]:rt.band[1]] = band.numpy()
    for k in expect:
        assert np.array_equal(full[k], expect[k]), k


def test_empty_mask_and_constant_paste():
    rng = np.random.RandomState(0)
    img = np.full((200, 300, 3), 230, dtype=np.uint8)                       # flat background: empty
    img[64:200, 150:300] = rng.randint(0, 256, (136, 150, 3))               # textured corner: not empty
    rt = TL.RegionTiler([torch.from_numpy(img)], 128, 8)
    g = T.TilerGeometry(300, 200, 128, 8)
    expect = [T.is_empty(T.extract_tile(img, g, int(x), int(y))) for x, y in rt.plan.origins]
    assert rt.empty_mask().tolist() == expect and any(expect) and not all(expect)
    ids = [i for i, e in enumerate(expect) if e]
    rt.paste('K', None, ids, const_rgb=(201, 211, 208))
    res = rt.results()['K'].numpy()
    rects = rt.plan.paste_rects()
    for i in ids:
        l, t, w, h, px, py = rects[i]
        assert (res[py:py + h, px:px + w] == [201, 211, 208]).all()
    # two input modalities: a tile is empty only if ALL of them are (models/__init__.py:393-394)
    rt2 = TL.RegionTiler([torch.from_numpy(img), torch.from_numpy(np.ascontiguousarray(img[::-1, ::-1]))], 128, 8)
    e2 = [T.is_empty([T.extract_tile(img, g, int(x), int(y)), T.extract_tile(np.ascontiguousarray(img[::-1, ::-1]), g, int(x), int(y))]) for x, y in rt2.plan.origins]
    assert rt2.empty_mask().tolist() == e2


def _small_opt(model='DeepLIIF', M=2, seg_gen=True):
    return types.SimpleNamespace(model=model, modalities_no=M, seg_gen=seg_gen, mod_id_seg='S', input_id=0, input_nc=3, output_nc=3, ngf=8,
                                 norm='batch', padding='zero', net_g='resnet_9blocks', net_gs='unet_32', input_no=1, scale_size=64,
                                 modalities_names=['input1', 'mod1', 'mod2'], background_colors=[(201, 211, 208), (10, 10, 10)], gpu_ids=[])


@pytest.mark.parametrize('world', [1, 2])
def test_infer_region_against_oracle_networks(world):
    """the whole loop: crop -> is_empty -> 4 generators + weighted seg sum on batches of 3 tiles (per-sample norm) -> uint8 -> stitch,
    against per-tile N=1 oracle forwards + the oracle tiler.  fp32 emulation: pixels may differ where the float value sits within
    1e-3 of an integer (truncation), nowhere else."""
    torch.manual_seed(0)
    opt = _small_opt()
    nets = I.build_generators(opt, torch.device('cpu'), 'fp32')
    img = synth_image(150, 100, 9)
    img[:, :50] = 255                                        # an empty strip on the left
    seg_w = [0.5, 0.2, 0.3]
    got = {}
    for r in range(world):
        bands, band = I.infer_region([torch.from_numpy(img)], 64, 4, nets, opt, seg_weights=seg_w, batch_size=3, rank=r, world=world)
        for k, v in bands.items():
            got.setdefault(k, np.zeros((100, 150, 3), dtype=np.uint8))[band[0]:band[1]] = v.numpy()
    # oracle
    g = T.TilerGeometry(150, 100, 64, 4)
    pos = g.positions()
    sds = {k: {kk: vv.detach().clone() for kk, vv in n.state_dict().items()} for k, n in nets.items()}
    colors = I.empty_tile_colors(opt)
    tiles_u8, floats = [], []
    for x, y in pos:
        t = T.extract_tile(img, g, x, y)
        if T.is_empty(t):
            tiles_u8.append({k: np.broadcast_to(np.array(c, dtype=np.uint8), (64, 64, 3)) for k, c in colors.items()})
            floats.append(None)
            continue
        ts = torch.from_numpy(T.transform(t))
        f = {}
        for i in (1, 2):
            f[f'G{i}'] = O.run_generator('resnet_9blocks', sds[f'G{i}'], ts, 'batch', 'zero')
        segs = {'GS1': O.run_generator('unet_32', sds['GS1'], f['G1'], 'batch', 'zero'),
                'GS2': O.run_generator('unet_32', sds['GS2'], f['G2'], 'batch', 'zero'),
                'GS0': O.run_generator('unet_32', sds['GS0'], ts, 'batch', 'zero')}
        f.update(segs)
        f['GS'] = seg_w[1] * segs['GS1'] + seg_w[2] * segs['GS2'] + seg_w[0] * segs['GS0']
        floats.append({k: v.numpy() for k, v in f.items()})
        tiles_u8.append({k: T.tensor2im(v.numpy()) for k, v in f.items()})
    expect = T.stitch(g, pos, tiles_u8)
    assert set(got) == set(expect)
    for k in expect:
        diff = got[k].astype(int) - expect[k].astype(int)
        assert np.abs(diff).max() <= 1, k
        assert (diff != 0).mean() < 0.02, (k, (diff != 0).mean())
