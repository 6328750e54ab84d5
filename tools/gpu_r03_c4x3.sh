#!/bin/bash
# strict stem / head kernels (conv_c4_patch_x3_kernel, wgrad_c4_x3_kernel): parity tests, layer timings, strict step A/B
TAG=${1:-c4x3}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "c4 or fused_norm_statistics" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python tools/layer_budget.py strict_$TAG fp32 2>&1 | grep -E "stem|head"
for v in 1 0; do
  if [ $v = 1 ]; then export DL_NO_C4_X3=1; else unset DL_NO_C4_X3; fi
  timeout 300 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check 2>/dev/null | tail -1 > gpurun_out/bench_strict_$TAG_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/bench_strict_$TAG_$v.json').read()); print('DL_NO_C4_X3=$v', d['value'], d['ms_per_step'])"
done
