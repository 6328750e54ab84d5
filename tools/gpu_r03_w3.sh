#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=300 -k "wgrad or big_tiles or conv_forward" 2>&1 | tail -8
timeout 120 python tools/conv_time.py fp32 2>&1 | tail -1
DL_WGRAD_X3=1 DL_X3_VAR=4 timeout 120 python tools/conv_time.py fp32 2>&1 | tail -1
DL_WGRAD_X3=2 DL_X3_VAR=1 timeout 120 python tools/conv_time.py fp32 2>&1 | tail -1
DL_X3_VAR=2 timeout 120 python tools/conv_time.py fp32 fwd 2>&1 | tail -1
timeout 120 python tools/conv_time.py bf16 2>&1 | tail -1
