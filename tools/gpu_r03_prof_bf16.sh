#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default (bf16) bench line: per-kernel averages next to the event-timed roofline average
TAG=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check > $GRAFT_REPO_ROOT/gpurun_out/bench_prof_bf16_$TAG.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof_bf16_$TAG.err); echo "rocprof rc=$?"
cp gpurun_out/prof_bf16_$TAG/bench_kernel_stats.csv gpurun_out/bench_train_kernel_stats_$TAG.csv 2>/dev/null
rm -rf gpurun_out/prof_bf16_$TAG
python - <<PY
import csv, json
rows = list(csv.DictReader(open('gpurun_out/bench_train_kernel_stats_$TAG.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot / 4e6)
for r in rows[:16]:
    print('%-96s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:96], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
d = json.loads(open('gpurun_out/bench_prof_bf16_$TAG.json').read().strip().splitlines()[-1])
print('bench under rocprof', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
PY
