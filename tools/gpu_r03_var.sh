#!/bin/bash
# timing of kernel variants of the strict 8-phase conv (one process per variant, same box)
export TMPDIR=/tmp
for v in 0 1 2 3; do DL_X3_VAR=$v timeout 120 python tools/conv_time.py fp32 fwd 2>&1 | tail -1; done
for v in 1 2 3 4; do DL_CONV_ABLATE=$v timeout 120 python tools/conv_time.py fp32 fwd 2>&1 | tail -1; done
DL_NO_X3_GLDS=1 timeout 120 python tools/conv_time.py fp32 fwd 2>&1 | tail -1
timeout 120 python tools/conv_time.py bf16 fwd 2>&1 | tail -1
