export TMPDIR=/tmp
true
cd /tmp && DL_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_add -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check --no-other-workloads > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_add/bench_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per 4 steps', round(tot / 1e6, 1))
for r in rows[:30]:
    print('%-80s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
for r in rows:
    if 'axpby' in r['Name']: print('AXPBY', r['Calls'], r['AverageNs'])
PY
cp gpurun_out/prof_add/bench_kernel_stats.csv gpurun_out/bench_train_kernel_stats_bf16_r06a.csv; rm -rf gpurun_out/prof_add
