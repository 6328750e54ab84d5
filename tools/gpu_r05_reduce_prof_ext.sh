#!/bin/bash
# round 5, run 21: rocprofv3 stats of the 18-net step (one stream) after the tiled slab reduction -- the reduce kernel's average against 253 us before
export TMPDIR=/tmp
mkdir -p gpurun_out
W=ext
(cd /tmp && DL_STREAMS=1 timeout 50 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$W -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-graph --no-timer-check --no-other-workloads > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$W.err); echo "rocprof $W rc=$?"
cp gpurun_out/prof_$W/bench_kernel_stats.csv gpurun_out/bench_${W}_kernel_stats_r05b.csv 2>/dev/null
rm -rf gpurun_out/prof_$W
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/bench_${W}_kernel_stats_r05b.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('$W: total kernel ms per 4 steps', round(tot / 1e6, 1))
for r in rows[:12]:
    print('%-86s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
