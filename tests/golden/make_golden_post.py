"""Fixtures for the post-processing row: the REFERENCE's deepliif/postprocessing.py run in this container
(numba stubbed to the identity decorator -> the pure-Python loops) on small synthetic seg / marker / original images.

  python tests/golden/make_golden_post.py        ->  tests/golden/post_cases.npz

numba widens every integer operation to 64 bits, the stubbed pure-Python run would instead wrap numpy's uint8 / uint16 scalars
(seg[y, x, 0] + seg[y, x, 2] overflows at 256, the optical-density sum of a cell at 65536).  To reproduce what the jitted reference
computes, the harness hands the images over as int64 arrays and converts create_od_image's uint16 result to int64 (a wrapper around
the reference's own function) -- values and comparisons are unchanged, only the wrap-around is avoided.

Stored per case: the three uint8 input images and the options, and what the reference returned: the label mask and cell list of
get_cells_info (postprocessing.py:311-362), the default thresholds, and overlay / refined / scoring of compute_final_results
(:1223-1304).  Inputs come from golden_util.synth_cells (seeded), so the GPU tests regenerate nothing -- they read this file.
Two more cases are the reference's own test data: the two validation images of Datasets/Sample_Dataset/val with the cell counts recorded in
its metrics.json (a known-answer test the reference ships; the images are data files, stored here as arrays)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import install_stubs          # noqa: E402
from golden_util import synth_cells            # noqa: E402

CASES = [
    # name, (H, W), seed, n_cells, kwargs of compute_final_results
    ('default_40x', (120, 150), 1, 28, dict(resolution='40x')),
    ('marker_default_20x', (96, 130), 2, 22, dict(resolution='20x', marker_thresh='default')),
    ('fixed_thresholds', (100, 100), 3, 20, dict(resolution='40x', size_thresh=30, marker_thresh=120, size_thresh_upper=400, noise_thresh=8)),
    ('large_noise_10x', (90, 160), 4, 26, dict(resolution='10x', large_noise_thresh='default', size_thresh=None)),
    ('optical_density', (110, 120), 5, 24, dict(resolution='40x', od_thresh_lower=20, od_thresh_upper=140)),
    ('crowded_touching', (128, 128), 6, 60, dict(resolution='40x', seg_thresh=100)),
    ('empty', (64, 80), 7, 0, dict(resolution='40x')),
    ('few_cells', (64, 64), 8, 4, dict(resolution='40x', noise_thresh=2)),
]


def cell_results(P, out, name, orig, seg, marker, kw, wide):
    """compute_cell_results (postprocessing.py:1136-1220) for data versions 3 (dicts), 4 (base-92 strings), 5 / 6 (optical density from the
    original image): stored as repr() strings"""
    ckw = {k: kw[k] for k in ('seg_thresh', 'noise_thresh', 'large_noise_thresh') if k in kw}
    for version in (3, 4, 5, 6):
        mk = orig.copy() if version >= 5 else wide(marker)
        res = P.compute_cell_results(wide(seg), mk, kw['resolution'], version=version, **ckw)
        out[f'{name}/cell_results_v{version}'] = np.array(repr(res))


def main():
    install_stubs()
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_postprocessing', '/root/reference/deepliif/postprocessing.py')
    P = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(P)
    _od = P.create_od_image
    P.create_od_image = lambda o: _od(o).astype(np.int64)          # numba: uint16 + uint16 -> int64 (see the module docstring)
    wide = lambda a: a.astype(np.int64)
    out = {'names': np.array([c[0] for c in CASES])}
    for name, (h, w), seed, ncell, kw in CASES:
        orig, seg, marker = synth_cells(h, w, ncell, seed)
        out[f'{name}/orig'], out[f'{name}/seg'], out[f'{name}/marker'] = orig, seg, marker
        out[f'{name}/kwargs'] = np.array(repr(kw))
        # stage results (get_cells_info with the same thresholds compute_final_results derives)
        noise = kw.get('noise_thresh', P.DEFAULT_NOISE_THRESH)
        segt = kw.get('seg_thresh', P.DEFAULT_SEG_THRESH)
        large = P.calculate_large_noise_thresh(kw.get('large_noise_thresh', None), kw['resolution'])
        use_od = kw.get('od_thresh_lower') is not None or kw.get('od_thresh_upper') is not None
        mask, cells, defaults = P.get_cells_info(wide(seg), wide(orig if use_od else marker), kw['resolution'], noise, segt, large, use_od=use_od)
        out[f'{name}/mask_after_mapping'] = np.asarray(mask)
        out[f'{name}/cells'] = np.array([[int(v) for v in c] for c in cells], dtype=np.int64).reshape(-1, 7)
        out[f'{name}/default_size_thresh'] = np.int64(defaults['size_thresh'])
        out[f'{name}/default_marker_thresh'] = np.int64(defaults.get('marker_thresh', -1))
        overlay, refined, scoring = P.compute_final_results(orig.copy(), wide(seg), (wide(marker) if not use_od else marker.copy()), **kw)
        out[f'{name}/overlay'], out[f'{name}/refined'] = np.asarray(overlay), np.asarray(refined)
        out[f'{name}/scoring'] = np.array(repr(scoring))
        cell_results(P, out, name, orig, seg, marker, kw, wide)
        print(name, 'cells', len(cells), 'defaults', defaults, 'scoring', scoring)
    # ---- the reference's own known answers: Datasets/Sample_Dataset/val/{Lung1,Bladder1}.png are 6 tiles side by side (IHC, Hema, DAPI,
    # Lap2, Marker, Seg) and val/metrics.json holds the cell counts its postprocess produced for them (prob_thresh = seg_thresh = 150)
    import json
    from PIL import Image
    val = '/root/reference/Datasets/Sample_Dataset/val'
    metrics = json.load(open(os.path.join(val, 'metrics.json')))
    names = list(out['names'])
    for n in ('Lung1', 'Bladder1'):
        im = np.asarray(Image.open(os.path.join(val, n + '.png')).convert('RGB'))
        orig, marker, seg = (np.ascontiguousarray(im[:, i * 512:(i + 1) * 512]) for i in (0, 4, 5))
        name = 'sample_' + n
        kw = dict(resolution='40x', size_thresh=metrics[n]['size_thresh'], seg_thresh=metrics[n]['prob_thresh'])
        out[f'{name}/orig'], out[f'{name}/seg'], out[f'{name}/marker'] = orig, seg, marker
        out[f'{name}/kwargs'] = np.array(repr(kw))
        large = P.calculate_large_noise_thresh(None, '40x')
        mask, cells, defaults = P.get_cells_info(wide(seg), wide(marker), '40x', P.DEFAULT_NOISE_THRESH, kw['seg_thresh'], large, use_od=False)
        out[f'{name}/mask_after_mapping'] = np.asarray(mask)
        out[f'{name}/cells'] = np.array([[int(v) for v in c] for c in cells], dtype=np.int64).reshape(-1, 7)
        out[f'{name}/default_size_thresh'] = np.int64(defaults['size_thresh'])
        out[f'{name}/default_marker_thresh'] = np.int64(defaults.get('marker_thresh', -1))
        overlay, refined, scoring = P.compute_final_results(orig.copy(), wide(seg), wide(marker), **kw)
        out[f'{name}/overlay'], out[f'{name}/refined'] = np.asarray(overlay), np.asarray(refined)
        out[f'{name}/scoring'] = np.array(repr(scoring))
        out[f'{name}/metrics_json'] = np.array(repr(metrics[n]))
        cell_results(P, out, name, orig, seg, marker, kw, wide)
        for k in ('num_total', 'num_pos', 'num_neg', 'percent_pos'):
            assert scoring[k] == metrics[n][k], (n, k, scoring[k], metrics[n][k])          # today's reference code reproduces its recorded counts
        names.append(name)
        print(name, 'cells', len(cells), 'scoring', scoring, '== metrics.json', metrics[n])
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'post_cases.npz'), **out)


if __name__ == '__main__':
    main()
