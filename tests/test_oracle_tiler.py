"""Pin the tiling / stitching oracle (oracle/tiler_oracle.py) against vectors produced by the REFERENCE's own InferenceTiler,
image_variance_gray / is_empty and tensor2im (tests/golden/make_golden_tiler.py).  Integer and byte work: bit-exact."""
import hashlib
import os

import numpy as np
import pytest

from golden_util import synth_image, tiler_result_tiles
from oracle import tiler_oracle as T

Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'tiler_cases.npz'))
PIXEL_CASES = sorted({k.split('/')[0] for k in Z.files if k.endswith('/geom')})
COORD_CASES = sorted({k.split('/')[0] for k in Z.files if k.endswith('/paste_ops')})


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('tag', PIXEL_CASES)
def test_tiles_and_stitched_images_match_reference(tag):
    w, h, tile, overlap, pad, seed = (int(v) for v in Z[f'{tag}/meta'])
    g = T.TilerGeometry(w, h, tile, overlap, pad)
    geom = [g.image_width, g.image_height, g.patch_size, g.center_width, g.center_height, g.c0x, g.c0y, g.c1x, g.c1y, g.c2x, g.c2y,
            g.c3x, g.c3y, g.p1x, g.p1y, g.p2x, g.p2y]
    assert geom == Z[f'{tag}/geom'].tolist()
    pos = g.positions()
    assert pos == [tuple(p) for p in Z[f'{tag}/positions'].tolist()]
    img = T.mirror_to_patch(synth_image(w, h, seed), g.patch_size)
    assert img.shape[:2] == (g.image_height, g.image_width)
    tiles = [T.extract_tile(img, g, x, y) for x, y in pos]
    assert [sha(t) for t in tiles[:3]] == Z[f'{tag}/tile_sha'].tolist()
    res = T.stitch(g, pos, [tiler_result_tiles(t) for t in tiles])
    for k in ('A', 'B'):
        assert list(res[k].shape) == Z[f'{tag}/res_shape/{k}'].tolist()
        assert sha(res[k]) == str(Z[f'{tag}/res_sha/{k}'])
        if f'{tag}/res/{k}' in Z.files:
            assert np.array_equal(res[k], Z[f'{tag}/res/{k}'])


@pytest.mark.parametrize('tag', COORD_CASES)
def test_wsi_scale_coordinates_and_paste_rectangles(tag):
    w, h, tile, overlap = (int(v) for v in Z[f'{tag}/meta'][:4])
    g = T.TilerGeometry(w, h, tile, overlap)
    pos = g.positions()
    assert np.array_equal(np.array(pos, dtype=np.int32), Z[f'{tag}/positions'])
    ops, counts = [], []
    for x, y in pos:
        p = g.pastes(x, y)
        counts.append(len(p))
        ops += [(px, py, r - l, b - t) for (l, t, r, b), (px, py) in p]
    assert np.array_equal(np.array(counts, dtype=np.int8), Z[f'{tag}/paste_count'])
    assert np.array_equal(np.array(ops, dtype=np.int32), Z[f'{tag}/paste_ops'])


def test_gray_conversion_variance_and_is_empty():
    names = Z['empty/names'].tolist()
    for i, n in enumerate(names):
        t = Z['empty/tiles'][i]
        assert np.array_equal(T.rgb_to_gray(t), Z['empty/gray'][i]), n
        assert T.image_variance_gray(t) == pytest.approx(float(Z['empty/var'][i]), rel=1e-12, abs=1e-12), n
        assert T.is_empty(t) == bool(Z['empty/is_empty'][i]), n
        cnt, s1, s2 = T.gray_sums(t)           # the exact integer form the GPU kernel evaluates
        assert (cnt == 0 or cnt * s2 - s1 * s1 < 9 * cnt * cnt) == bool(Z['empty/is_empty'][i]), n
    assert T.is_empty([Z['empty/tiles'][names.index('white')], Z['empty/tiles'][names.index('noise')]]) is False


def test_tensor2im_truncation():
    assert np.array_equal(T.tensor2im(Z['t2i/in']), Z['t2i/out'])
    assert np.array_equal(T.tensor2im(Z['t2i/in_gray']), Z['t2i/out_gray'])


def test_invalid_arguments_raise_like_the_reference():
    for args in ((100, 100, 0, 0, 0), (100, 100, 64, -1, 0), (100, 100, 64, 0, -1), (300, 300, 64, 32, 0), (300, 300, 64, 16, 16)):
        with pytest.raises(ValueError):
            T.TilerGeometry(*args)
