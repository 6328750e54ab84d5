#!/bin/bash
# round 6: conv_s2d_kernel (csrc/conv_s2d.hip) -- parity, isolated launches with / without it (two alternations), whole steps
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/s2d_${1:-a}.txt
rm -f $O
echo "== parity" >> $O
timeout 900 python -m pytest tests/test_gpu_s2d.py -m gpu -q -x ${KSEL:+-k "$KSEL"} 2>&1 | tail -15 >> $O
for rep in 1 2; do
  for v in 0 1; do
    echo "== layers DL_CONV_S2D=$v (round $rep)" >> $O
    DL_CONV_S2D=$v timeout 300 python tools/s2d_time.py 2>&1 | grep -v '^{' | tail -4 >> $O
  done
done
if [ "$2" != "nosteps" ]; then
echo "== whole steps (DL_CONV_S2D)" >> $O
for v in 0 1 0 1; do
  DL_CONV_S2D=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/s2d_bench_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/s2d_bench_$v.json').read()); r=d['roofline']; print('S2D=$v', d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('one_stream_ms_per_step'))" >> $O
done
fi
cat $O
