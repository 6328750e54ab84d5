#!/bin/bash
# round 6: (a) parity of conv_s2d / conv_s2u after the compile-time ReLU, their isolated times;
#          (b) non-temporal loads / stores in the norm passes: libdeepliif_hip_nt{1,2,3,4,7}.so (norm.hip built with -DDL_NORM_NT=<bits>) against the shipped library,
#              kernel averages from rocprofv3 (one stream) + whole steps
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/nt_${1:-a}.txt
rm -f $O
echo "== parity s2d/s2u" >> $O
true
echo "== layers" >> $O
true
for v in base 3 5 7 11 13 15 base 3; do
  if [ $v = base ]; then unset DEEPLIIF_AMD_LIB; else export DEEPLIIF_AMD_LIB=$GRAFT_REPO_ROOT/deepliif_amd/libdeepliif_hip_nt$v.so; fi
  rm -rf gpurun_out/prof_nt
  (cd /tmp && DL_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_nt -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check --no-other-workloads > /dev/null 2>&1)
  echo "== NT=$v kernel averages (one stream, us)" >> $O
  python - >> $O <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_nt/bench_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
norm = sum(float(r['TotalDurationNs']) for r in rows if 'norm_' in r['Name'])
print('total kernel ms / 4 steps %.1f   norm family %.1f' % (tot / 1e6, norm / 1e6))
for r in rows:
    if 'norm_' in r['Name'] and float(r['TotalDurationNs']) / tot > 0.004:
        print('  %-70s %5s calls %8.1f us avg' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/nt_bench.json
  python -c "
import json; d=json.loads(open('gpurun_out/nt_bench.json').read()); r=d['roofline']; print('NT=$v tiles/s', d['value'], 'ms', d['ms_per_step'], 'one-stream', r.get('one_stream_ms_per_step'))" >> $O
done
rm -rf gpurun_out/prof_nt
cat $O
