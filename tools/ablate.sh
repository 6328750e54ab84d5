#!/bin/bash
export TMPDIR=/tmp
for v in 0 3; do
  if [ $v = 0 ]; then unset DL_CONV_ABLATE; else export DL_CONV_ABLATE=$v; fi
  echo "=== ablate=$v"
  timeout 300 python tools/microbench.py 2>/dev/null | grep "bf16" | grep -E "res3x3" | cut -c1-200
done
export DL_CONV_ABLATE=3
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -k "conv_forward" 2>&1 | tail -1
