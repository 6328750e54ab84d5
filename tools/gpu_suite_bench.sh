#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py -m gpu -q --timeout=600 -x > gpurun_out/run7_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/run7_tests.log
bash tools/gpu_bench.sh ${1:-v3}
