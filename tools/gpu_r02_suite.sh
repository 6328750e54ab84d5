#!/bin/bash
# round-2 verification on ONE box: every GPU test, then the default bench line (with the strict-parity leg) and the wsi workload
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout=300 2>&1 | tail -15 > gpurun_out/tests_$TAG.log; echo "tests rc=${PIPESTATUS[0]}"; cat gpurun_out/tests_$TAG.log
cp gpurun_out/parity_errors.json gpurun_out/parity_errors_$TAG.json 2>/dev/null
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
timeout 300 python bench.py --workload wsi --steps 254 --warmup 2 > gpurun_out/bench_wsi_$TAG.json 2> gpurun_out/bench_wsi_$TAG.err; echo "wsi rc=$?"
python - <<PY
import json
for f in ('bench_$TAG', 'bench_wsi_$TAG'):
    try:
        d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['steps'], (d.get('strict_parity') or {}).get('value'), (d.get('strict_parity') or {}).get('headline_vs_strict'))
    except Exception as e:
        print(f, 'failed', e); print(open(f'gpurun_out/{f}.err').read()[-1500:])
PY
# per-kernel statistics of the same bench command (kernel trace only; counters are collected in separate passes)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict > $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$TAG.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$TAG.err); echo "rocprof rc=$?"
cp gpurun_out/prof_$TAG/bench_kernel_stats.csv gpurun_out/bench_kernel_stats_$TAG.csv 2>/dev/null
rm -rf gpurun_out/prof_$TAG
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/bench_kernel_stats_$TAG.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print('%-70s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
