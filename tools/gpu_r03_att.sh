#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=300 -k "gate" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=600 -k "attention or teacher_forced_unet" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_seam.py tests/test_gpu_distributed.py -m gpu -q --timeout=600 2>&1 | grep -v "Warn\|warn" | tail -5
