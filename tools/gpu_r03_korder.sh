#!/bin/bash
# A/B of the K order of the strict 8-phase kernel (DL_X3_KORDER=1: channel-chunk-major): isolated launch, parity, whole strict step
mkdir -p gpurun_out
for v in 0 1 0 1; do
  echo "== DL_X3_KORDER=$v" >> gpurun_out/korder.txt
  DL_X3_KORDER=$v python tools/conv_time.py fp32 fwd,dgrad 2>/dev/null | tail -1 >> gpurun_out/korder.txt
done
DL_X3_KORDER=1 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "big_tiles or fused_norm_statistics or split_copy" 2>&1 | tail -2 >> gpurun_out/korder.txt
for v in 0 1 0 1; do
  DL_X3_KORDER=$v python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check 2>/dev/null | tail -1 > gpurun_out/korder_bench_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/korder_bench_$v.json').read()); print('DL_X3_KORDER=$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])" >> gpurun_out/korder.txt
done
cat gpurun_out/korder.txt
