export TMPDIR=/tmp
mkdir -p gpurun_out
for P in bf16 fp16; do
  rm -rf gpurun_out/prof_inf
  (cd /tmp && DL_INFER_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_inf -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload infer --precision $P --steps 5 --warmup 2 --no-cpu-baseline --no-timer-check > /dev/null 2>&1)
  cp gpurun_out/prof_inf/bench_kernel_stats.csv gpurun_out/bench_infer_kernel_stats_${P}_r06.csv
  python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/bench_infer_kernel_stats_${P}_r06.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('$P: total kernel ms', round(tot / 1e6, 1))
for r in rows[:12]:
    print('  %-80s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
done
rm -rf gpurun_out/prof_inf
