#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -x -k "big_tiles or fused_norm" 2>&1 | tail -8
for v in 1 0; do
  export DL_CONV_8PH=$v
  echo "=== 8ph=$v"
  timeout 300 python tools/microbench.py 2>/dev/null | grep "bf16" | grep -E "res3x3" | cut -c1-200
done
