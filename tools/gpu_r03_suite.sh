#!/bin/bash
# round-3 verification on ONE box: every GPU test, default bench line (+ strict leg), strict kernel profile, strict layer budget
TAG=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -15 > gpurun_out/tests_$TAG.log; echo "tests rc=${PIPESTATUS[0]}"; cat gpurun_out/tests_$TAG.log
cp gpurun_out/parity_errors.json gpurun_out/parity_errors_$TAG.json 2>/dev/null
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], 'roofline', d['roofline'] and d['roofline']['frac'], 'strict', (d.get('strict_parity') or {}).get('value'), (d.get('strict_parity') or {}).get('ms_per_step'))
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-strict > $GRAFT_REPO_ROOT/gpurun_out/bench_prof_strict_$TAG.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof_strict_$TAG.err); echo "rocprof rc=$?"
cp gpurun_out/prof_$TAG/bench_kernel_stats.csv gpurun_out/bench_train_kernel_stats_strict_$TAG.csv 2>/dev/null
rm -rf gpurun_out/prof_$TAG
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/bench_train_kernel_stats_strict_$TAG.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per 4 steps', tot / 1e6)
for r in rows[:22]:
    print('%-90s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
timeout 600 python tools/layer_budget.py strict_$TAG fp32 2>&1 | tail -24
