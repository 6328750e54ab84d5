#!/bin/bash
# w4 DMA variants (DL_W4_VAR: 1 = global_load_lds, 5 = weights by buffer_load lds, 13 = weights + activations by buffer_load lds): parity, isolated launches, whole steps
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/w4_${1:-c}.txt
rm -f $O
KS='(big_tiles or fused_norm_statistics) and bf16 and (conv256-256k3s1n8 or conv192-256 or conv64-256k3s1n16)'
for v in 5 13; do
  echo "== parity (eligible cases) DL_W4_VAR=$v" >> $O
  DL_W4_VAR=$v timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "$KS" 2>&1 | tail -2 >> $O
done
for rep in 1 2; do
  echo "== isolated launches, round $rep" >> $O
  for v in 1 5 13; do DL_W4_VAR=$v timeout 120 python tools/conv_time.py bf16 fwd,dgrad 2>/dev/null | tail -1 >> $O; done
done
echo "== whole steps (DL_W4_VAR)" >> $O
for v in 1 5 13 1 5 13; do
  DL_W4_VAR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check 2>/dev/null | tail -1 > gpurun_out/w4c_bench_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/w4c_bench_$v.json').read()); r=d['roofline']; print('VAR=$v', d['value'], d['ms_per_step'], r['avg_launch_us'], r.get('median_launch_us'), r['frac'], r.get('sustained'))" >> $O
done
cat $O
