#!/bin/bash
# the complete GPU suite (no -x), log under gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=10 "$@" > gpurun_out/r05_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_gpu_tests.log
grep -E "passed|failed|^FAILED|^ERROR|pytest rc" gpurun_out/r05_gpu_tests.log | tail -40
