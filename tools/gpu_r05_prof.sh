#!/bin/bash
# round 5 mid-round profile: rocprofv3 stats of the one-stream step (both policies), stream-count sweep, layer budget
TAG=${1:-r05b}
export TMPDIR=/tmp
mkdir -p gpurun_out
for P in bf16 fp32; do
  (cd /tmp && DL_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$P -o bench -- python $GRAFT_REPO_ROOT/bench.py --precision $P --steps 3 --warmup 1 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check --no-other-workloads > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$P.err); echo "rocprof $P rc=$?"
  cp gpurun_out/prof_$P/bench_kernel_stats.csv gpurun_out/bench_train_kernel_stats_${P}_$TAG.csv 2>/dev/null
  rm -rf gpurun_out/prof_$P
  python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/bench_train_kernel_stats_${P}_$TAG.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('$P: total kernel ms per 4 steps', round(tot / 1e6, 1))
for r in rows[:22]:
    print('%-86s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
done
for ns in 2 3 4 5; do
  echo "== DL_STREAMS=$ns"; DL_STREAMS=$ns timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')})"
done
timeout 600 python tools/layer_budget.py $TAG bf16 2>&1 | tail -26
