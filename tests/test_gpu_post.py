"""GPU post-processing (deepliif_amd/postprocessing.py -> dl_pp_cells / dl_pp_finish) against the REFERENCE's results
(tests/golden/post_cases.npz) and, at sizes the reference's pure-Python loops cannot reach here, against the pinned oracle.
Integer / byte work: everything is compared bit-exactly."""
import os

import numpy as np
import pytest
import torch

from deepliif_amd import ops
from deepliif_amd import postprocessing as PP
from golden_util import synth_cells
from oracle import postprocess_oracle as PO

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'post_cases.npz'))
NAMES = [str(n) for n in Z['names']]


@pytest.fixture(autouse=True)
def _real_backend():
    ops._impl = None
    yield


@pytest.mark.parametrize('name', NAMES)
def test_cell_mapping_against_reference_fixture(name):
    kw = eval(str(Z[f'{name}/kwargs']))
    use_od = kw.get('od_thresh_lower') is not None or kw.get('od_thresh_upper') is not None
    large = PP.calculate_large_noise_thresh(kw.get('large_noise_thresh', None), kw['resolution'])
    cm = PP.get_cells_info(Z[f'{name}/seg'], Z[f'{name}/orig'] if use_od else Z[f'{name}/marker'], kw['resolution'],
                           kw.get('noise_thresh', PP.DEFAULT_NOISE_THRESH), kw.get('seg_thresh', PP.DEFAULT_SEG_THRESH), large, use_od=use_od)
    assert np.array_equal(cm.mask.cpu().numpy(), Z[f'{name}/mask_after_mapping'])
    got = np.array([[int(v) for v in c] for c in cm.cells], dtype=np.int64).reshape(-1, 7)
    assert np.array_equal(got, Z[f'{name}/cells'])
    assert cm.defaults['size_thresh'] == int(Z[f'{name}/default_size_thresh'])
    assert cm.defaults.get('marker_thresh', -1) == int(Z[f'{name}/default_marker_thresh'])
    # the label image names every cell by its first pixel in raster order
    lab = cm.label.cpu().numpy()
    for c in cm.cells:
        assert lab[c[4], c[3]] == c[4] * lab.shape[1] + c[3]


@pytest.mark.parametrize('name', NAMES)
def test_final_results_against_reference_fixture(name):
    kw = eval(str(Z[f'{name}/kwargs']))
    overlay, refined, scoring = PP.compute_final_results(Z[f'{name}/orig'], Z[f'{name}/seg'], Z[f'{name}/marker'], **kw)
    assert np.array_equal(overlay, Z[f'{name}/overlay'])
    assert np.array_equal(refined, Z[f'{name}/refined'])
    assert scoring == eval(str(Z[f'{name}/scoring']))


@pytest.mark.parametrize('h,w,ncell,seed,kw', [
    (512, 512, 400, 11, dict(resolution='40x')),
    (777, 1000, 900, 12, dict(resolution='20x', marker_thresh='default', large_noise_thresh='default')),
    (300, 2048, 700, 13, dict(resolution='40x', od_thresh_lower=15, od_thresh_upper=160)),
    (1, 37, 0, 14, dict(resolution='40x')),                       # a single row: every pixel is a border pixel
    (61, 1, 0, 15, dict(resolution='10x', size_thresh=None)),
])
def test_larger_images_against_the_oracle(h, w, ncell, seed, kw):
    orig, seg, marker = synth_cells(h, w, ncell, seed)
    overlay, refined, scoring = PP.compute_final_results(torch.from_numpy(orig).cuda(), torch.from_numpy(seg).cuda(), torch.from_numpy(marker).cuda(), **kw)
    import time
    t0 = time.perf_counter()
    o_overlay, o_refined, o_scoring = PO.compute_final_results(orig, seg, marker, **kw)
    print(f'oracle (scipy labelling + numpy) on {h}x{w}: {time.perf_counter() - t0:.3f} s')
    assert scoring == o_scoring
    assert np.array_equal(overlay, o_overlay) and np.array_equal(refined, o_refined)


def test_pil_inputs_and_tensor_outputs():
    from PIL import Image
    name = 'default_40x'
    kw = eval(str(Z[f'{name}/kwargs']))
    pil = [Image.fromarray(Z[f'{name}/{k}']) for k in ('orig', 'seg', 'marker')]
    overlay, refined, scoring = PP.compute_final_results(*pil, return_tensors=True, **kw)
    assert overlay.is_cuda and refined.is_cuda and overlay.dtype == torch.uint8
    assert np.array_equal(overlay.cpu().numpy(), Z[f'{name}/overlay']) and np.array_equal(refined.cpu().numpy(), Z[f'{name}/refined'])
    assert scoring == eval(str(Z[f'{name}/scoring']))


@pytest.mark.parametrize('name', [n for n in NAMES if n.startswith('sample_')])
def test_known_answers_of_the_reference_sample_dataset(name):
    """The reference's own recorded cell counts (Datasets/Sample_Dataset/val/metrics.json) through the GPU path"""
    kw = eval(str(Z[f'{name}/kwargs']))
    want = eval(str(Z[f'{name}/metrics_json']))
    _, _, scoring = PP.compute_final_results(Z[f'{name}/orig'], Z[f'{name}/seg'], Z[f'{name}/marker'], **kw)
    assert {k: scoring[k] for k in ('num_total', 'num_pos', 'num_neg', 'percent_pos')} == {k: want[k] for k in ('num_total', 'num_pos', 'num_neg', 'percent_pos')}


@pytest.mark.parametrize('version', [3, 4, 5, 6])
@pytest.mark.parametrize('name', NAMES)
def test_compute_cell_results_against_reference_fixture(name, version):
    """compute_cell_results end to end (GPU cell mapping + host-side boundaries / encoding) vs the reference's own output"""
    kw = eval(str(Z[f'{name}/kwargs']))
    ckw = {k: kw[k] for k in ('seg_thresh', 'noise_thresh', 'large_noise_thresh') if k in kw}
    got = PP.compute_cell_results(Z[f'{name}/seg'], Z[f'{name}/orig'] if version >= 5 else Z[f'{name}/marker'], kw['resolution'], version=version, **ckw)
    assert got == eval(str(Z[f'{name}/cell_results_v{version}']))
