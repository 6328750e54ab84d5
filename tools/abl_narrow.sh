# scratch driver for the probe of the moment (rewritten per experiment)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wsi_prof -o wsi -- python $GRAFT_REPO_ROOT/bench.py --workload wsi --region 8192 --steps 40 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/bench_wsi_prof.json 2> /dev/null)
cp gpurun_out/wsi_prof/wsi_kernel_stats.csv gpurun_out/wsi_kernel_stats.csv; rm -rf gpurun_out/wsi_prof
python - <<PY
import csv, json
rows = list(csv.DictReader(open('gpurun_out/wsi_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
tile = sum(float(r['TotalDurationNs']) for r in rows if 'tile_' in r['Name'])
print('tile kernels share %.2f %%' % (100 * tile / tot))
for r in rows[:10]:
    print('%-64s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
for r in rows:
    if 'tile_' in r['Name']: print('%-64s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print(open('gpurun_out/bench_wsi_prof.json').read()[:300])
PY
