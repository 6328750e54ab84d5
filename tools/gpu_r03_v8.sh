#!/bin/bash
export TMPDIR=/tmp
for v in 0 8 4 2; do DL_X3_VAR=$v timeout 120 python tools/conv_time.py fp32 fwd,dgrad 2>&1 | tail -1; done
DL_X3_VAR=8 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=300 -k "big_tiles" 2>&1 | tail -4
