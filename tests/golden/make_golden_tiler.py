"""Golden vectors for the tiling / stitching side of the inference path, produced by the REFERENCE (build container only).

    python tests/golden/make_golden_tiler.py      -> tests/golden/tiler_cases.npz

Runs the reference's own InferenceTiler (deepliif/util/__init__.py:129-331) on PIL images, its image_variance_gray /
is_empty (util/__init__.py:478-486, models/__init__.py:391-396) and tensor2im (util/util.py:117-135).  Data only:
image sizes, seeds, tile coordinates, paste rectangles, uint8 pixels (small cases in full, large ones as SHA-256).
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402
from golden_util import synth_image, tiler_result_tiles as result_tiles  # noqa: E402

_ref_import.install_stubs()
import torch  # noqa: E402
from deepliif.util import InferenceTiler, image_variance_gray  # noqa: E402
from deepliif.util.util import tensor2im  # noqa: E402
import deepliif.models as ref_models  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(out, tag, w, h, tile, overlap, pad, seed, store_full):
    img = synth_image(w, h, seed)
    tiler = InferenceTiler(Image.fromarray(img), tile, overlap, pad)
    pos, first_tiles = [], []
    for t in tiler:
        tn = np.asarray(t)
        pos.append((tiler.x, tiler.y))
        if len(first_tiles) < 3:
            first_tiles.append(tn)
        tiler.stitch({k: Image.fromarray(v) for k, v in result_tiles(tn).items()})
    res = tiler.results()
    out[f'{tag}/meta'] = np.array([w, h, tile, overlap, pad, seed], dtype=np.int64)
    out[f'{tag}/positions'] = np.array(pos, dtype=np.int64)
    out[f'{tag}/geom'] = np.array([tiler.image_width, tiler.image_height, tiler.patch_size, tiler.center_width, tiler.center_height,
                                   tiler.c0x, tiler.c0y, tiler.c1x, tiler.c1y, tiler.c2x, tiler.c2y, tiler.c3x, tiler.c3y,
                                   tiler.p1x, tiler.p1y, tiler.p2x, tiler.p2y], dtype=np.int64)
    out[f'{tag}/tile_sha'] = np.array([sha(t) for t in first_tiles])
    for k, im in res.items():
        a = np.asarray(im)
        out[f'{tag}/res_sha/{k}'] = np.array(sha(a))
        out[f'{tag}/res_shape/{k}'] = np.array(a.shape, dtype=np.int64)
        if store_full:
            out[f'{tag}/res/{k}'] = a
    print(tag, 'tiles', len(pos), 'result', {k: np.asarray(v).shape for k, v in res.items()})


class PasteRecorder:
    """stands in for a result image: records (paste position, pasted size) instead of holding 1.2 GB of pixels"""

    def __init__(self):
        self.ops = []

    def paste(self, im, box):
        self.ops.append((box[0], box[1], im.size[0], im.size[1]))


def coords_case(out, tag, w, h, tile, overlap):
    tiler = InferenceTiler(Image.new('L', (w, h)), tile, overlap)
    dummy = Image.new('RGB', (tile, tile))
    rec = PasteRecorder()
    tiler.res['k'] = rec
    pos, nops = [], []
    for _ in tiler:
        pos.append((tiler.x, tiler.y))
        before = len(rec.ops)
        tiler.stitch({'k': dummy})
        nops.append(len(rec.ops) - before)
    out[f'{tag}/meta'] = np.array([w, h, tile, overlap, 0, 0], dtype=np.int64)
    out[f'{tag}/positions'] = np.array(pos, dtype=np.int32)
    out[f'{tag}/paste_ops'] = np.array(rec.ops, dtype=np.int32)          # (x, y, w, h) in paste order
    out[f'{tag}/paste_count'] = np.array(nops, dtype=np.int8)
    print(tag, 'tiles', len(pos), 'pastes', len(rec.ops))


def main():
    out = {}
    run_case(out, 'one512', 512, 512, 512, 32, 0, 1, False)                 # a single tile: overlap collapses to 0
    run_case(out, 'w300h200_t128_o8', 300, 200, 128, 8, 0, 2, True)
    run_case(out, 'w257h130_t128_o16', 257, 130, 128, 16, 0, 3, True)        # one pixel more than two centres wide
    run_case(out, 'w100h70_t128_o8', 100, 70, 128, 8, 0, 4, True)            # smaller than a patch: mirrored
    run_case(out, 'w300h90_t128_o8', 300, 90, 128, 8, 0, 5, True)            # only the height is mirrored
    run_case(out, 'w300h200_t128_o8_p4', 300, 200, 128, 8, 4, 6, True)       # solid padding around every tile
    run_case(out, 'w1381h949_t512_o32', 1381, 949, 512, 32, 0, 7, False)     # SURVEY 8(c) size
    coords_case(out, 'w20000h20000_t512_o32', 20000, 20000, 512, 32)         # BASELINE configs[4] geometry
    coords_case(out, 'w20000h13337_t512_o32', 20000, 13337, 512, 32)

    # ---- is_empty / image_variance_gray (threshold 9)
    rng = np.random.RandomState(11)
    tiles = {
        'noise': rng.randint(0, 256, (64, 64, 3)),
        'white': np.full((64, 64, 3), 255),
        'black': np.zeros((64, 64, 3)),
        'flat200': np.full((64, 64, 3), 200),
        'lowvar': 200 + rng.randint(-2, 3, (64, 64, 3)),
        'near9a': 120 + rng.randint(-7, 8, (64, 64, 3)),
        'near9b': 120 + rng.randint(-8, 9, (64, 64, 3)),
        'whiteblobs': np.where(rng.rand(64, 64, 1) < 0.7, 255, 90 + rng.randint(-3, 4, (64, 64, 3))),
    }
    names = sorted(tiles)
    out['empty/names'] = np.array(names)
    out['empty/tiles'] = np.stack([np.clip(tiles[n], 0, 255).astype(np.uint8) for n in names])
    out['empty/var'] = np.array([float(image_variance_gray(Image.fromarray(np.clip(tiles[n], 0, 255).astype(np.uint8)))) for n in names])
    out['empty/is_empty'] = np.array([bool(ref_models.is_empty(Image.fromarray(np.clip(tiles[n], 0, 255).astype(np.uint8)))) for n in names])
    out['empty/gray'] = np.stack([np.asarray(Image.fromarray(np.clip(tiles[n], 0, 255).astype(np.uint8)).convert('L')) for n in names])
    print('is_empty', dict(zip(names, out['empty/is_empty'])), out['empty/var'])

    # ---- tensor2im
    g = torch.Generator().manual_seed(5)
    t = torch.rand(2, 3, 24, 24, generator=g) * 2 - 1
    t[0, 0, 0, :6] = torch.tensor([-1.0, 1.0, 0.0, 0.999999, -0.999999, 0.00392157])
    out['t2i/in'] = t.numpy()
    out['t2i/out'] = tensor2im(t)
    t1 = torch.rand(1, 1, 8, 8, generator=g) * 2 - 1
    out['t2i/in_gray'] = t1.numpy()
    out['t2i/out_gray'] = tensor2im(t1)
    np.savez_compressed(os.path.join(HERE, 'tiler_cases.npz'), **out)
    print('wrote tiler_cases.npz', os.path.getsize(os.path.join(HERE, 'tiler_cases.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
