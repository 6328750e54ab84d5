#!/bin/bash
export TMPDIR=/tmp
for d in randn bf16 zero; do for v in 0 4 2; do TIME_DATA=$d DL_X3_VAR=$v timeout 120 python tools/conv_time.py fp32 fwd 2>&1 | tail -1; done; done
for d in randn zero; do TIME_DATA=$d timeout 120 python tools/conv_time.py bf16 fwd 2>&1 | tail -1; done
