#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -k "narrow" > gpurun_out/run9_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/run9_tests.log
bash tools/gpu_bench.sh ${1:-v7}
