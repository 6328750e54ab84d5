#!/bin/bash
# A/B: conv K-step order (DL_CONV_KORDER) and wgrad XCD grouping (DL_WGRAD_XCDGROUP) on the ResnetBlock layer shapes
export TMPDIR=/tmp
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  export DL_CONV_KORDER=$1 DL_WGRAD_XCDGROUP=$2
  echo "=== conv k_order=$1 wgrad xcd_group=$2"
  timeout 300 python tools/microbench.py 2>/dev/null | grep "bf16" | grep -E "res3x3|down2|up1" | cut -c1-220
done
unset DL_CONV_KORDER DL_WGRAD_XCDGROUP
timeout 800 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -x 2>&1 | tail -2
