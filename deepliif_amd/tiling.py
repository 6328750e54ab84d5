"""Tile geometry + GPU-side crop / is_empty / stitch for the inference path (BASELINE configs[4]).

The reference walks a PIL image one tile at a time (deepliif/util/__init__.py:129-331 InferenceTiler; the loop is
deepliif/models/__init__.py:496-498) and resolves overlapping pastes by paste ORDER.  Here the whole plan is integer geometry
computed up front, in closed form:

  * tile origins are separable: xs x ys, each axis = range(0, extent, centre) with the last origin clamped to extent - patch
    (:258-265); clamping makes the reference visit the last column / row more than once -- identical tiles, visited once here;
  * the nine conditional pastes of stitch() (:293-320) are one rectangle per tile: along an axis the tile at origin o owns
    [o + (0 if o == 0 else overlap), o + (patch if o == extent - patch else patch - overlap));
  * "later paste wins" along an axis means the tile with the LARGEST origin covering a pixel owns it, so tile i's effective
    interval ends where tile i+1's begins.  The effective rectangles are disjoint -> one order-independent GPU launch.

TilePlan holds that geometry (pinned against the reference's own iteration and paste sequence through the oracle,
tests/test_tiling_host.py).  RegionTiler runs it on a uint8 RGB image resident in HBM: dl_tile_gather_u8 (crop + transform()),
dl_tile_gray_stats_u8 (is_empty), dl_tile_paste_u8 (tensor2im + stitch) -- include/deepliif_hip.h.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops

EMPTY_VARIANCE_THRESHOLD = 9          # deepliif/models/__init__.py:392


def _axis_origins(extent: int, patch: int, centre: int) -> List[int]:
    """unique tile origins along one axis, increasing (range(0, extent, centre), clamped: util/__init__.py:258-265)"""
    out: List[int] = []
    for o in range(0, extent, centre):
        if o + patch > extent:
            o = extent - patch
        if not out or o != out[-1]:
            out.append(o)
    return out


def _axis_intervals(origins: Sequence[int], extent: int, patch: int, overlap: int) -> List[Tuple[int, int]]:
    """effective [lo, hi) each tile owns along the axis after "the later paste wins" (module docstring)"""
    lo = [o + (0 if o == 0 else overlap) for o in origins]
    hi = [o + (patch if o == extent - patch else patch - overlap) for o in origins]
    return [(lo[i], min(hi[i], lo[i + 1]) if i + 1 < len(origins) else hi[i]) for i in range(len(origins))]


class TilePlan:
    """Integer geometry of InferenceTiler(orig, tile_size, overlap_size, pad_size) for a width x height image."""

    def __init__(self, width: int, height: int, tile_size: int, overlap_size: int = 0, pad_size: int = 0):
        if tile_size <= 0:
            raise ValueError('InferenceTiler input tile_size must be positive and non-zero')
        if overlap_size < 0:
            raise ValueError('InferenceTiler input overlap_size must be positive or zero')
        if pad_size < 0:
            raise ValueError('InferenceTiler input pad_size must be positive or zero')
        self.orig_width, self.orig_height = int(width), int(height)
        self.tile_size, self.pad_size = int(tile_size), int(pad_size)
        self.patch_size = patch = tile_size - 2 * pad_size
        self.image_width, self.image_height = max(width, patch), max(height, patch)      # mirrored out to a patch (:196-211)
        self.overlap_width = 0 if patch >= self.image_width else overlap_size
        self.overlap_height = 0 if patch >= self.image_height else overlap_size
        self.center_width = patch - 2 * self.overlap_width
        self.center_height = patch - 2 * self.overlap_height
        if self.center_width <= 0 or self.center_height <= 0:
            raise ValueError('InferenceTiler combined overlap_size and pad_size are too large')
        self.xs = _axis_origins(self.image_width, patch, self.center_width)
        self.ys = _axis_origins(self.image_height, patch, self.center_height)
        self._ix = _axis_intervals(self.xs, self.image_width, patch, self.overlap_width)
        self._iy = _axis_intervals(self.ys, self.image_height, patch, self.overlap_height)

    def __len__(self) -> int:
        return len(self.xs) * len(self.ys)

    @property
    def origins(self) -> np.ndarray:
        """int32 [n_tiles, 2] = (x, y), row-major (the reference's iteration order without the repeated visits)"""
        return np.array([(x, y) for y in self.ys for x in self.xs], dtype=np.int32).reshape(-1, 2)

    def visit_order(self) -> List[int]:
        """tile index for every tile the reference's __iter__ yields, repeats included"""
        def axis(extent, patch, centre, origins):
            idx = []
            for o in range(0, extent, centre):
                if o + patch > extent:
                    o = extent - patch
                idx.append(origins.index(o))
            return idx
        ax = axis(self.image_width, self.patch_size, self.center_width, self.xs)
        ay = axis(self.image_height, self.patch_size, self.center_height, self.ys)
        return [j * len(self.xs) + i for j in ay for i in ax]

    def paste_rects(self) -> np.ndarray:
        """int32 [n_tiles, 6] = (l, t, w, h, px, py): window of the result TILE (pad included in l, t) and where it lands in the
        result image; disjoint, covering the image exactly once."""
        rects = []
        for j, y in enumerate(self.ys):
            y0, y1 = self._iy[j]
            for i, x in enumerate(self.xs):
                x0, x1 = self._ix[i]
                rects.append((x0 - x + self.pad_size, y0 - y + self.pad_size, x1 - x0, y1 - y0, x0, y0))
        return np.array(rects, dtype=np.int32).reshape(-1, 6)


def transform_lut() -> np.ndarray:
    """transform() per byte value (deepliif/data/__init__.py:133-138: ToTensor = v / 255 in fp32, Normalize = (x - 0.5) / 0.5)"""
    v = np.arange(256, dtype=np.float32) / np.float32(255.0)
    return ((v - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def gray_stats_empty(stats: np.ndarray) -> np.ndarray:
    """is_empty from {count, sum, sumsq} rows (exact): variance < 9 <=> n*s2 - s1^2 < 9*n^2; no pixel in 1..254 -> variance 0"""
    n, s1, s2 = (stats[:, k].astype(object) for k in range(3))         # Python ints: no overflow whatever the tile size
    return np.array([bool(a == 0 or a * c - b * b < EMPTY_VARIANCE_THRESHOLD * a * a) for a, b, c in zip(n, s1, s2)], dtype=bool)


def _rgb_word(c) -> int:
    return int(c[0]) | (int(c[1]) << 8) | (int(c[2]) << 16)


class RegionTiler:
    """TilePlan bound to uint8 RGB image(s) [H, W, 3] in HBM (several = the input modalities of a multi-input model).

    `rows=(j0, j1)` restricts the tiler to the tile rows j0..j1-1 of the plan: the rectangles those tiles own form ONE contiguous
    horizontal band of the result, so tile-parallel inference over R ranks is "every rank takes a band, the bands are concatenated"
    -- no overlap to resolve, no collective on the data path (BASELINE configs[4]; deepliif_amd.inference.infer_region)."""

    def __init__(self, images: Sequence[torch.Tensor], tile_size: int, overlap_size: int = 0, pad_size: int = 0, pad_color=(255, 255, 255),
                 rows: Optional[Tuple[int, int]] = None):
        assert images, 'at least one image'
        for im in images:
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or im.stride(2) != 1 or im.stride(1) != 3:
                raise TypeError('RegionTiler images are uint8 [H, W, 3] tensors with densely packed pixels')
            if im.shape != images[0].shape:
                raise ValueError('InferenceTiler input images do not have the same size.')
        self.images = list(images)
        self.device = images[0].device
        h, w = int(images[0].shape[0]), int(images[0].shape[1])
        self.plan = p = TilePlan(w, h, tile_size, overlap_size, pad_size)
        j0, j1 = rows if rows is not None else (0, len(p.ys))
        assert 0 <= j0 <= j1 <= len(p.ys)
        self.rows = (j0, j1)
        nx = len(p.xs)
        self.tile_ids = list(range(j0 * nx, j1 * nx))                 # global tile indices this tiler owns
        # the band of result rows those tiles own (clipped to the un-mirrored image height)
        self.band = (min(p._iy[j0][0], p.orig_height), min(p._iy[j1 - 1][1], p.orig_height)) if j1 > j0 else (0, 0)
        self.pad_rgb = _rgb_word(pad_color)
        self.origins = torch.from_numpy(p.origins).to(self.device)
        self.lut = torch.from_numpy(transform_lut()).to(self.device)
        self._rects = p.paste_rects()
        self._results: Dict[str, torch.Tensor] = {}

    def __len__(self):
        return len(self.tile_ids)

    def _origins_of(self, tile_ids) -> torch.Tensor:
        idx = torch.as_tensor(list(tile_ids), dtype=torch.long, device=self.device)
        return self.origins.index_select(0, idx).contiguous()

    def empty_mask(self) -> np.ndarray:
        """is_empty for every tile of this tiler, in tile_ids order (all input images must be empty: models/__init__.py:393-394)"""
        be = ops.impl()
        p = self.plan
        empty = np.ones(len(self.tile_ids), dtype=bool)
        if not self.tile_ids:
            return empty
        org = self._origins_of(self.tile_ids)
        for im in self.images:
            stats = torch.empty((len(self.tile_ids), 3), dtype=torch.int64, device=self.device)
            be.tile_gray_stats(im, p.orig_height, p.orig_width, org, p.tile_size, p.pad_size, self.pad_rgb, stats)
            empty &= gray_stats_empty(stats.cpu().numpy())
        return empty

    def gather(self, tile_ids: Sequence[int], dtype: torch.dtype, cp: int) -> torch.Tensor:
        """engine tile batch [len(tile_ids), tile, tile, cp] in [-1, 1] (crop + transform)"""
        p = self.plan
        out = torch.empty((len(tile_ids), p.tile_size, p.tile_size, cp), dtype=dtype, device=self.device)
        ops.impl().tile_gather(self.images, p.orig_height, p.orig_width, self._origins_of(tile_ids), p.tile_size, p.pad_size, self.pad_rgb, self.lut, out)
        return out

    def result(self, key: str) -> torch.Tensor:
        """this tiler's band of result image `key`: uint8 [band height (mirror extension included), image_width, 3]"""
        if key not in self._results:          # Image.new('RGB', ...) is black (util/__init__.py:289-290)
            p = self.plan
            j0, j1 = self.rows
            y0, y1 = (p._iy[j0][0], p._iy[j1 - 1][1]) if j1 > j0 else (0, 0)
            self._results[key] = torch.zeros((y1 - y0, p.image_width, 3), dtype=torch.uint8, device=self.device)
        return self._results[key]

    def paste(self, key: str, tiles: Optional[torch.Tensor], tile_ids: Sequence[int], const_rgb=None):
        """stitch tiles[i] (engine layout, values in [-1, 1]) as tile tile_ids[i] of result image `key`; tiles None pastes the
        constant colour instead (empty tiles)."""
        if not len(tile_ids):
            return
        p = self.plan
        rec = np.zeros((len(tile_ids), 8), dtype=np.int32)
        rec[:, 0] = -1 if tiles is None else np.arange(len(tile_ids))
        rec[:, 1:7] = self._rects[list(tile_ids)]
        rec[:, 6] -= p._iy[self.rows[0]][0]                            # band-local row
        rec[:, 7] = 0 if const_rgb is None else _rgb_word(const_rgb)
        rec = rec[(rec[:, 3] > 0) & (rec[:, 4] > 0)]
        if len(rec):
            ops.impl().tile_paste(tiles, p.tile_size, torch.from_numpy(np.ascontiguousarray(rec)).to(self.device), self.result(key))

    def results(self) -> Dict[str, torch.Tensor]:
        """result bands cropped back to the original image (util/__init__.py:322-331): rows band[0]..band[1] of the full result"""
        p = self.plan
        h = self.band[1] - self.band[0]
        return {k: v[:h, :p.orig_width] for k, v in self._results.items()}


def split_rows(n_rows: int, world: int) -> List[Tuple[int, int]]:
    """contiguous, near-equal bands of tile rows, one per rank (ranks beyond the row count get an empty band)"""
    base, extra = divmod(n_rows, world)
    out, j = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((j, j + n))
        j += n
    return out
