"""Post-processing row (f2): compute_final_results on the GPU (inputs resident in HBM) on synthetic images.

  python tools/post_probe.py            -> gpurun_out/post_probe.json
Per size: whole-call time (two C-ABI calls + the host-side cell-list arithmetic + one device->host copy of the cell table), the GPU part alone
(HIP events around dl_pp_cells and dl_pp_finish), algorithmic bytes = 9 B/pixel read (seg, marker, orig) + 6 B/pixel written (overlay,
refined).  The CPU side of the comparison (the pinned oracle: scipy.ndimage.label + numpy) is timed by the test suite, not here -- only
tests/, smoke() and bench.py's cpu_baseline leg may execute oracle/ (tests/test_gpu_post.py::test_larger_images_against_the_oracle prints it with -s)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
from deepliif_amd import postprocessing as PP
from golden_util import synth_cells

out = []
SIZES = ((512, 512, 300), (2048, 2048, 5000)) + (((8192, 8192, 60000),) if os.environ.get('POST_PROBE_BIG') else ())
for (h, w, ncell) in SIZES:
    t0 = time.perf_counter()
    orig, seg, marker = synth_cells(512, 512, 300, 21)
    if h > 512:                                    # tile the 512^2 image (the generator is O(cells x pixels)): blobs stay cell-sized, the count scales with the area
        r = h // 512
        orig, seg, marker = (np.ascontiguousarray(np.tile(a, (r, r, 1))) for a in (orig, seg, marker))
    d = [torch.from_numpy(a).cuda() for a in (orig, seg, marker)]
    kw = dict(resolution='40x', marker_thresh='default')
    res = PP.compute_final_results(*d, return_tensors=True, **kw)          # warm-up
    torch.cuda.synchronize()
    reps = 5 if h <= 2048 else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        res = PP.compute_final_results(*d, return_tensors=True, **kw)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / reps
    # GPU part alone
    large = PP.calculate_large_noise_thresh(None, '40x')
    s, e, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    s.record()
    cm = PP.get_cells_info(d[1], d[2], '40x', PP.DEFAULT_NOISE_THRESH, PP.DEFAULT_SEG_THRESH, large)
    e.record()
    torch.cuda.synchronize()
    t_cells_call = s.elapsed_time(e) * 1e-3
    row = {'size': [h, w], 'cells': res[2]['num_total'], 'components': int(len(cm.keep)), 'gpu_call_s': t_all, 'cells_stage_incl_host_s': t_cells_call,
           'algorithmic_MB': 15 * h * w / 1e6, 'algorithmic_GBs_whole_call': 15 * h * w / t_all / 1e9}
    print(json.dumps(row), flush=True)
    out.append(row)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/post_probe.json', 'w'), indent=1)
