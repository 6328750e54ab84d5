#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=900 > gpurun_out/run2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/run2_tests.log
tail -60 gpurun_out/run2_tests.log
