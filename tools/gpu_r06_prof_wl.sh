#!/bin/bash
# one-stream rocprofv3 kernel table of another workload: tools/gpu_r06_prof_wl.sh <workload> [precision]
export TMPDIR=/tmp
mkdir -p gpurun_out
WL=${1:-train18}; P=${2:-bf16}
rm -rf gpurun_out/prof_wl
(cd /tmp && DL_STREAMS=1 DL_INFER_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_wl -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --precision $P --steps 3 --warmup 1 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check --no-other-workloads > /dev/null 2>&1)
cp gpurun_out/prof_wl/bench_kernel_stats.csv gpurun_out/bench_${WL}_kernel_stats_${P}_r06.csv
rm -rf gpurun_out/prof_wl
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/bench_${WL}_kernel_stats_${P}_r06.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows if 'probe_mfma' not in r['Name'])
print('$WL $P: total kernel ms per 4 steps', round(tot / 1e6, 1))
for r in rows[:36]:
    print('%-92s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:92], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
