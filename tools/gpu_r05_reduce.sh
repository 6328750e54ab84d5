#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_wgrad_batch.py tests/test_gpu_kernels.py -q -k "wgrad or deferred or batch or exchange or reduction or training" 2>&1 | tail -4
(cd /tmp && DL_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_red -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check --no-other-workloads > /dev/null 2>&1)
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_red/bench_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows if 'probe_mfma' not in r['Name'])
print('one-stream kernel ms per step', round(tot / 4e6, 2))
for r in rows:
    if 'wgrad' in r['Name']:
        print('%-80s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
rm -rf gpurun_out/prof_red
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-baseline-n8 --no-graph --no-other-workloads --no-timer-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], 'strict', d['strict_parity']['value'])"
} > gpurun_out/r05_reduce.txt 2>&1
cat gpurun_out/r05_reduce.txt
