"""The whole-slide region loop (deepliif_amd/wsi.py; reference: deepliif/models/__init__.py:663-727) on CPU: region order and geometry, the
regions x ranks schedule, and -- through the emulated ops backend -- that running a slide as scheduled regions on 1, 2 or 3 ranks gives the
same uint8 canvases and cell counts as the reference's sequential loop of stand-alone infer_modalities() calls."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from deepliif_amd import wsi as W

HERE = os.path.dirname(os.path.abspath(__file__))


def test_region_grid_follows_the_reference_loop():
    # the reference's loop, literally (models/__init__.py:688-716)
    def ref(size_x, size_y, region_size):
        out, start_x, start_y = [], 0, 0
        while start_x < size_x:
            while start_y < size_y:
                out.append((start_x, start_y, min(region_size, size_x - start_x), min(region_size, size_y - start_y)))
                start_y += region_size
            start_y = 0
            start_x += region_size
        return out
    for sx, sy, rs in ((100000, 80000, 20000), (20000, 20000, 20000), (20001, 19999, 20000), (150, 230, 100), (7, 7, 3)):
        assert W.region_grid(sx, sy, rs) == ref(sx, sy, rs)
    with pytest.raises(ValueError):
        W.region_grid(0, 10, 5)


@pytest.mark.parametrize('size,world', [((100000, 80000), 8), ((100000, 80000), 3), ((47000, 20000), 2), ((20000, 20000), 8), ((30000, 20000), 8), ((20000, 13337), 1)])
def test_plan_covers_every_tile_once_and_balances(size, world):
    sx, sy = size
    p = W.plan_slide(sx, sy, 512, world)
    assert p.regions == W.region_grid(sx, sy)
    assert W.plan_slide(sx, sy, 512, world).jobs == p.jobs                       # deterministic: every rank computes the same plan
    total = sum(j.n_tiles for r in range(world) for j in p.jobs[r])
    from deepliif_amd.tiling import TilePlan
    per_region = [len(TilePlan(w, h, 512, 32).ys) * len(TilePlan(w, h, 512, 32).xs) for (_, _, w, h) in p.regions]
    assert total == sum(per_region) and p.tiles_per_rank == [sum(j.n_tiles for j in p.jobs[r]) for r in range(world)]
    if len(p.regions) >= world:
        assert p.mode == 'regions'
        seen = sorted(j.index for r in range(world) for j in p.jobs[r])
        assert seen == list(range(len(p.regions))) and all(j.world == 1 and j.rank == 0 for r in range(world) for j in p.jobs[r])
        # longest-processing-time bound: no rank exceeds the mean by more than one (largest) region
        assert max(p.tiles_per_rank) <= total / world + max(per_region)
    else:
        assert p.mode == 'bands'
        for i in range(len(p.regions)):
            parts = [j for r in range(world) for j in p.jobs[r] if j.index == i]
            assert len(parts) == world and sorted(j.rank for j in parts) == list(range(world)) and sum(j.n_tiles for j in parts) == per_region[i]
        assert max(p.tiles_per_rank) - min(p.tiles_per_rank) <= len(p.regions) * len(TilePlan(p.regions[0][2], p.regions[0][3], 512, 32).xs)


def _setup():
    import fake_backend
    from deepliif_amd import inference as I
    from golden_util import synth_image
    fake_backend.install()
    I._device_for = lambda opt: torch.device('cpu')
    torch.manual_seed(0)
    opt = types.SimpleNamespace(model='DeepLIIF', modalities_no=1, seg_gen=True, mod_id_seg='S', input_id=0, input_nc=3, output_nc=3, ngf=8,
                                norm='batch', padding='zero', net_g='resnet_9blocks', net_gs='unet_32', input_no=1, scale_size=64,
                                modalities_names=['IHC', 'Marker'], background_colors=[(201, 211, 208)], gpu_ids=[])
    nets = I.build_generators(opt, torch.device('cpu'), 'fp32')
    # the GPU post-processing has no emulation in the fake ops backend: the CPU oracle's compute_final_results stands in for it here
    # (tests/test_gpu_tiles.py / test_gpu_post.py hold the GPU path itself to the reference's bytes)
    from deepliif_amd import postprocessing as PP
    from oracle import postprocess_oracle as PO
    PP.compute_final_results = lambda orig, seg, marker, resolution, size_thresh='default', marker_thresh=None, size_thresh_upper=None, seg_thresh=120: \
        PO.compute_final_results(np.asarray(orig), np.asarray(seg), None if marker is None else np.asarray(marker), resolution, size_thresh, marker_thresh,
                                 size_thresh_upper, seg_thresh)
    slide = synth_image(230, 150, 17)               # 150 rows x 230 columns
    slide[:40] = 251                                # is_empty tiles at the top
    return I, opt, nets, slide


def _reference_loop(I, opt, nets, slide, region_size, tile_size=64, **kw):
    """infer_results_for_wsi restated: sequential regions, stand-alone infer_modalities per region, canvas paste, summed counts"""
    from PIL import Image
    h, w = slide.shape[:2]
    canv, total = {}, None
    for (x, y, rw, rh) in W.region_grid(w, h, region_size):
        images, scoring = I.infer_modalities(Image.fromarray(slide[y:y + rh, x:x + rw]), tile_size, None, opt=opt, nets=nets, batch_size=3, **kw)
        total = W.add_scoring(total, scoring)
        W.paste_into(canv, (x, y, rw, rh), images, w, h)
    return canv, W.finish_scoring(total)


def _run_ranks(world, region_size):
    I, opt, nets, slide = _setup()
    h, w = slide.shape[:2]
    canv, total = {}, None
    for rank in range(world):
        def on_region(xywh, images, scoring):
            W.paste_into(canv, xywh, images, w, h)
        plan, part = W.infer_slide(lambda x, y, rw, rh: slide[y:y + rh, x:x + rw], w, h, 64, None, nets=nets, opt=opt, region_size=region_size, rank=rank, world=world,
                                   batch_size=3, on_region=on_region)
        assert plan.mode == 'regions'
        total = W.add_scoring(total, part)
    ref_canv, ref_total = _reference_loop(I, opt, nets, slide, region_size)
    import fake_backend
    fake_backend.uninstall()
    return canv, W.finish_scoring(total), ref_canv, ref_total


@pytest.mark.parametrize('world', [1, 2, 4])
def test_scheduled_regions_equal_the_sequential_region_loop(world):
    canv, total, ref_canv, ref_total = _run_ranks(world, 100)          # 230 x 150 slide, regions of 100: 3 x 2 = 6 regions
    assert sorted(canv) == sorted(ref_canv) and 'Seg' in canv and 'SegRefined' in canv
    for k in ref_canv:
        assert np.array_equal(canv[k], ref_canv[k]), k
    assert total == ref_total and total['num_total'] == total['num_pos'] + total['num_neg']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


BAND_VARIANTS = {'plain': (64, {}), 'seg_only': (64, dict(seg_only=True)), 'resampled': (48, {})}      # tile 48 != scale_size 64: the PIL-resampling route


def _band_worker(rank, world, port, out, variant='plain'):
    tile_size, kw = BAND_VARIANTS[variant]
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, 'golden')):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from deepliif_amd import distributed as D
    D.init_process_group_from_env('gloo')
    I, opt, nets, slide = _setup()
    h, w = slide.shape[:2]
    canv = {}
    plan, total = W.infer_slide(lambda x, y, rw, rh: slide[y:y + rh, x:x + rw], w, h, tile_size, None, nets=nets, opt=opt, region_size=200, rank=rank, world=world,
                                batch_size=3, on_region=lambda xywh, images, scoring: W.paste_into(canv, xywh, images, w, h), **kw)
    assert plan.mode == 'bands' and len(plan.regions) == 2
    if rank == 0:
        torch.save({'canv': canv, 'total': W.finish_scoring(total)}, os.path.join(out, 'bands.pt'))
    else:
        assert not canv and total is None                      # the bands were sent to rank 0; post-processing and counting happen there
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('variant', ['plain', 'seg_only', 'resampled'])
def test_fewer_regions_than_ranks_splits_every_region_into_bands(tmp_path, variant):
    """the 'bands' mode must give the images, keys and counts of the 'regions' mode / the reference loop also with seg_only
    (non-Seg images dropped after post-processing) and with tile_size != scale_size (host resampling) -- both go through infer_modalities()"""
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)
    tile_size, kw = BAND_VARIANTS[variant]
    mp.spawn(_band_worker, args=(3, _free_port(), str(tmp_path), variant), nprocs=3, join=True)          # 2 regions (200 + 30 columns) on 3 ranks
    got = torch.load(tmp_path / 'bands.pt', weights_only=False)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        os.environ.pop(k, None)
    I, opt, nets, slide = _setup()
    ref_canv, ref_total = _reference_loop(I, opt, nets, slide, 200, tile_size, **kw)
    import fake_backend
    fake_backend.uninstall()
    assert sorted(got['canv']) == sorted(ref_canv)
    if variant == 'seg_only':
        assert all('Seg' in k for k in ref_canv)
    for k in ref_canv:
        assert np.array_equal(got['canv'][k], ref_canv[k]), k
    assert got['total'] == ref_total
