timeout 600 python -m pytest tests/test_gpu_post.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python tools/post_probe.py 2>&1 | tail -3
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pp_prof -o pp -- python -c "
import sys, os; sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/tests/golden')
import torch, numpy as np
from deepliif_amd import postprocessing as PP
from golden_util import synth_cells
o, s, m = synth_cells(2048, 2048, 5000, 21)
d = [torch.from_numpy(a).cuda() for a in (o, s, m)]
for _ in range(3): PP.compute_final_results(*d, resolution='40x', marker_thresh='default', return_tensors=True)
torch.cuda.synchronize()
" > /dev/null 2>&1)
cp gpurun_out/pp_prof/pp_kernel_stats.csv gpurun_out/post_kernel_stats.csv; rm -rf gpurun_out/pp_prof
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/post_kernel_stats.csv')))
print('GPU total per call: %.1f us' % (sum(float(r['TotalDurationNs']) for r in rows if 'pp_' in r['Name'] or 'rocprim' in r['Name'] or 'rocclr' in r['Name']) / 3e3))
for r in rows[:8]:
    print('%-60s %5s calls %9.1f us avg' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
