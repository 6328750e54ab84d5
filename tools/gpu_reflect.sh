#!/bin/bash
# reflection-padded training on the GPU: fold kernel, network gradients, the reference's reflect trajectory
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=100 -k "reflect_fold" 2>&1 | tail -3
timeout 200 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=150 -k "reflect" 2>&1 | tail -6
