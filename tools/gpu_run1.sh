#!/bin/bash
# first GPU contact: probes + kernel parity; everything lands in gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=900 -k "probe" > gpurun_out/run1_probe.log 2>&1
echo "probe rc=$?" >> gpurun_out/run1_probe.log
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 > gpurun_out/run1_kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/run1_kernels.log
tail -40 gpurun_out/run1_kernels.log
timeout 600 python tools/microbench.py > gpurun_out/run1_microbench.log 2>&1
echo "microbench rc=$?" >> gpurun_out/run1_microbench.log
tail -30 gpurun_out/run1_microbench.log
