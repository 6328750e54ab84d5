#!/bin/bash
# A/B of the K order of the bf16 8-phase kernel (DL_8PH_KORDER=1: channel-chunk-major): isolated launch, parity, whole step (headline policy)
mkdir -p gpurun_out
rm -f gpurun_out/korder8.txt
for v in 0 1 0 1; do
  echo "== DL_8PH_KORDER=$v" >> gpurun_out/korder8.txt
  DL_8PH_KORDER=$v python tools/conv_time.py bf16 fwd,dgrad 2>/dev/null | tail -1 >> gpurun_out/korder8.txt
done
DL_8PH_KORDER=1 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "big_tiles or fused_norm_statistics" 2>&1 | tail -2 >> gpurun_out/korder8.txt
for v in 0 1 0 1; do
  DL_8PH_KORDER=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check 2>/dev/null | tail -1 > gpurun_out/korder8_bench_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/korder8_bench_$v.json').read()); print('DL_8PH_KORDER=$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])" >> gpurun_out/korder8.txt
done
cat gpurun_out/korder8.txt
