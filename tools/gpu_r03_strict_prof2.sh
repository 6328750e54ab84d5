#!/bin/bash
TAG=${1:-r03b}
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-timer-check > $GRAFT_REPO_ROOT/gpurun_out/bench_prof_strict_$TAG.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof_strict_$TAG.err); echo "rocprof rc=$?"
cp gpurun_out/prof_$TAG/bench_kernel_stats.csv gpurun_out/bench_train_kernel_stats_strict_$TAG.csv 2>/dev/null
rm -rf gpurun_out/prof_$TAG
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/bench_train_kernel_stats_strict_$TAG.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per 4 steps', tot / 1e6)
for r in rows[:26]:
    print('%-90s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
