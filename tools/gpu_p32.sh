#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -x -k "big_tiles or fused_norm" 2>&1 | tail -8
for cfg in "1 1 0" "1 0 0" "0 1 0" "1 1 1" "1 1 2" "1 1 3"; do
  set -- $cfg
  export DL_CONV_P32=$1 DL_CONV_KWR=$2
  if [ $3 = 0 ]; then unset DL_CONV_ABLATE; else export DL_CONV_ABLATE=$3; fi
  echo "=== p32=$1 kwr=$2 ablate=$3"
  timeout 300 python tools/microbench.py 2>/dev/null | grep "bf16" | grep -E "res3x3" | cut -c1-150
done
