#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -k "conv_forward" > gpurun_out/run5_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/run5_tests.log
timeout 300 python tools/microbench.py > gpurun_out/run5_microbench.log 2>&1; grep "bf16" gpurun_out/run5_microbench.log | cut -c1-260
