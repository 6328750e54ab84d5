#!/bin/bash
# same-box A/B of the training step between the shipped library and variant builds: tools/gpu_r06_libab.sh <tag> <variant .so name> [<variant> ...]
# KSHOW=substr,substr: also list these kernels when they are not among the first 14
# (variants live next to the shipped library, e.g. deepliif_amd/libdeepliif_hip_w4nt.so; selected through DEEPLIIF_AMD_LIB)
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=$1; shift
O=gpurun_out/libab_$TAG.txt
rm -f $O
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then unset DEEPLIIF_AMD_LIB; else export DEEPLIIF_AMD_LIB=$GRAFT_REPO_ROOT/deepliif_amd/$v; fi
  rm -rf gpurun_out/prof_ab
  (cd /tmp && DL_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ab -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check --no-other-workloads > /dev/null 2>&1)
  echo "== $v (round $rep): kernel averages, one stream, us" >> $O
  python - >> $O <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_ab/bench_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows if 'probe_mfma' not in r['Name'])
print('total kernel ms / step %.2f' % (tot / 4e6))
import os
show = [t for t in os.environ.get('KSHOW', '').split(',') if t]
for i, r in enumerate(rows):
    if i < 14 or any(t in r['Name'] for t in show):
        print('  %-70s %5s calls %8.1f us avg' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/libab_bench.json
  python -c "
import json; d=json.loads(open('gpurun_out/libab_bench.json').read()); r=d['roofline']; print('$v tiles/s', d['value'], 'ms', d['ms_per_step'], 'one-stream', r.get('one_stream_ms_per_step'), 'w4 us', r.get('avg_launch_us'))" >> $O
done
done
rm -rf gpurun_out/prof_ab
cat $O
