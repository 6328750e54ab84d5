"""Whole-slide driver: the region loop of deepliif.models.infer_results_for_wsi (deepliif/models/__init__.py:663-727) as a schedule over
"regions x ranks".

The reference reads a slide in regions of at most `region_size` (20 000) pixels a side -- start_x outer loop, start_y inner loop -- runs
infer_modalities() on every region as a stand-alone image (its own InferenceTiler: tiles never straddle regions), pastes the region's
result images into slide-sized uint8 canvases and adds up the regions' cell counts (num_pos / num_neg; percent_pos from the sums).  One
device, one region after the other.

Here (BASELINE configs[4] asks for 8 GPUs): the regions are independent, and inside a region the tiles are independent
(inference.infer_region: bands of tile rows, disjoint paste rectangles).  plan_slide() turns (slide size, region size, tile size, world) into
a deterministic assignment with no collective on the data path:
  * at least as many regions as ranks (a 100 k x 80 k slide = 20 regions on 8 GPUs): whole regions go to ranks, longest-processing-time
    first on the tile count (regions at the right / bottom edge are smaller) -- a rank reads, infers, post-processes and counts cells of its
    regions alone;
  * fewer regions than ranks (one 20 k x 20 k region on 8 GPUs): every region is split into bands of tile rows over ALL ranks
    (tiling.split_rows); the bands of a region are reassembled (inference.gather_bands) before its post-processing, which needs the whole
    region.
File I/O stays outside (bioformats / tiff writers are out of scope): the caller passes read_region(x, y, w, h) -> uint8 [h, w, 3]
and receives per-region results to paste / write; paste_into() does the reference's canvas paste."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

from .tiling import TilePlan, split_rows


def region_grid(size_x: int, size_y: int, region_size: int = 20000) -> List[Tuple[int, int, int, int]]:
    """(x, y, w, h) of every region in the reference's visiting order (models/__init__.py:690-715: start_x outer, start_y inner)."""
    if size_x <= 0 or size_y <= 0 or region_size <= 0:
        raise ValueError(f'empty slide or region ({size_x} x {size_y}, region {region_size})')
    out = []
    for x in range(0, size_x, region_size):
        for y in range(0, size_y, region_size):
            out.append((x, y, min(region_size, size_x - x), min(region_size, size_y - y)))
    return out


@dataclass(frozen=True)
class RegionJob:
    index: int                       # position in region_grid() order
    xywh: Tuple[int, int, int, int]
    n_tiles: int
    rank: int                        # infer_region(rank=..., world=...) arguments of this job on the calling rank
    world: int


@dataclass
class SlidePlan:
    regions: List[Tuple[int, int, int, int]]
    mode: str                        # 'regions' (whole regions per rank) | 'bands' (every region split over all ranks)
    jobs: List[List[RegionJob]]      # jobs[r] = what rank r runs, in region order
    tiles_per_rank: List[int]


def plan_slide(size_x: int, size_y: int, tile_size: int, world: int = 1, region_size: int = 20000, overlap_size: Optional[int] = None) -> SlidePlan:
    overlap = tile_size // 16 if overlap_size is None else overlap_size          # infer_modalities' overlap (models/__init__.py:632)
    regions = region_grid(size_x, size_y, region_size)
    plans = [TilePlan(w, h, tile_size, overlap) for (_, _, w, h) in regions]
    counts = [len(p.ys) * len(p.xs) for p in plans]
    jobs: List[List[RegionJob]] = [[] for _ in range(world)]
    load = [0] * world
    if len(regions) >= world:
        # longest processing time first; ties broken by region order, then by rank: deterministic on every rank
        for i in sorted(range(len(regions)), key=lambda i: (-counts[i], i)):
            r = min(range(world), key=lambda r: (load[r], r))
            jobs[r].append(RegionJob(i, regions[i], counts[i], 0, 1))
            load[r] += counts[i]
        for r in range(world):
            jobs[r].sort(key=lambda j: j.index)
        return SlidePlan(regions, 'regions', jobs, load)
    for i, (reg, p) in enumerate(zip(regions, plans)):
        for r, (j0, j1) in enumerate(split_rows(len(p.ys), world)):
            n = (j1 - j0) * len(p.xs)
            jobs[r].append(RegionJob(i, reg, n, r, world))
            load[r] += n
    return SlidePlan(regions, 'bands', jobs, load)


def add_scoring(total: Optional[dict], part: Optional[dict]) -> Optional[dict]:
    """models/__init__.py:697-706: cell counts add up over the regions"""
    if part is None or 'num_pos' not in part:          # DeepLIIFExt / SDG key their counts by Seg image (models/__init__.py:604-609): nothing to add up
        return total
    if total is None:
        return {'num_pos': part['num_pos'], 'num_neg': part['num_neg']}
    total['num_pos'] += part['num_pos']
    total['num_neg'] += part['num_neg']
    return total


def finish_scoring(total: Optional[dict]) -> Optional[dict]:
    """models/__init__.py:721-725"""
    if total is None:
        return None
    total['num_total'] = total['num_pos'] + total['num_neg']
    total['percent_pos'] = round(total['num_pos'] / total['num_total'] * 100, 1) if total['num_pos'] > 0 else 0
    return total


def paste_into(canvases: Dict[str, 'object'], xywh: Tuple[int, int, int, int], images: Dict[str, 'object'], size_x: int, size_y: int):
    """models/__init__.py:708-713: results[name][y : y + h, x : x + w] = region image (numpy canvases, created on first use)"""
    import numpy as np
    x, y, w, h = xywh
    for name, img in images.items():
        if name not in canvases:
            canvases[name] = np.zeros((size_y, size_x, 3), dtype=np.uint8)
        canvases[name][y:y + h, x:x + w] = np.asarray(img)
    return canvases


def infer_slide(read_region: Callable[[int, int, int, int], 'object'], size_x: int, size_y: int, tile_size: int, model_dir: Optional[str] = None, *,
                nets=None, opt=None, region_size: int = 20000, rank: int = 0, world: int = 1, eager_mode: bool = False, seg_weights=None,
                seg_only: bool = False, return_seg_intermediate: bool = False, color_dapi: bool = False, color_marker: bool = False, batch_size: int = 8,
                on_region: Optional[Callable[[Tuple[int, int, int, int], dict, Optional[dict]], None]] = None):
    """Run this rank's share of a slide.  'regions' mode: every job is a whole region -> infer_modalities() (tiling, generators, uint8 stitch and
    post-processing on the GPU) -> on_region(xywh, images, scoring) and the running cell counts.  'bands' mode: every region is inferred in bands
    by all ranks, gathered on rank 0 (inference.gather_bands) and post-processed there.
    Returns (plan, scoring of THIS rank's regions or None); the caller sums the scorings over ranks (add_scoring) and calls finish_scoring()."""
    import numpy as np
    from PIL import Image
    from . import inference as I
    plan = plan_slide(size_x, size_y, tile_size, world, region_size)
    total = None
    if opt is None and model_dir is not None:
        opt = I.get_opt(model_dir)
    for job in plan.jobs[rank]:
        x, y, w, h = job.xywh
        region = read_region(x, y, w, h)
        if plan.mode == 'regions':
            img = region if isinstance(region, Image.Image) else Image.fromarray(np.asarray(region, dtype=np.uint8))
            images, scoring = I.infer_modalities(img, tile_size, model_dir, eager_mode=eager_mode, color_dapi=color_dapi, color_marker=color_marker, opt=opt,
                                                 return_seg_intermediate=return_seg_intermediate, seg_only=seg_only, seg_weights=seg_weights, nets=nets,
                                                 batch_size=batch_size)
            total = add_scoring(total, scoring)
            if on_region is not None:
                on_region(job.xywh, images, scoring)
            continue
        # bands: every rank infers its tile rows of this region; rank 0 reassembles and post-processes -- through the SAME infer_modalities() as the
        # 'regions' mode (scale_size resampling, input_no / SDG split, seg_gen guard and the seg_only clean-up do not depend on the world size)
        import torch
        if torch.is_tensor(region):
            region = region.cpu().numpy()
        img = region if isinstance(region, Image.Image) else Image.fromarray(np.ascontiguousarray(np.asarray(region, dtype=np.uint8)))
        images, scoring = I.infer_modalities(img, tile_size, model_dir, eager_mode=eager_mode, color_dapi=color_dapi, color_marker=color_marker, opt=opt,
                                             return_seg_intermediate=return_seg_intermediate, seg_only=seg_only, seg_weights=seg_weights, nets=nets,
                                             batch_size=batch_size, rank=job.rank, world=job.world)
        if images is None:
            continue
        total = add_scoring(total, scoring)
        if on_region is not None:
            on_region(job.xywh, images, scoring)
    return plan, total
