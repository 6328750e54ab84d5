#!/bin/bash
# round 3: strict-policy direct-to-LDS kernels: parity tests, then layer budget + strict bench line (+ A/B against the old path)
TAG=${1:-r03_x3}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=300 -k "conv" 2>&1 | tail -15 > gpurun_out/tests_k_$TAG.log; echo "kernel tests rc=${PIPESTATUS[0]}"; tail -8 gpurun_out/tests_k_$TAG.log
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q -x --timeout=600 2>&1 | tail -15 > gpurun_out/tests_n_$TAG.log; echo "network tests rc=${PIPESTATUS[0]}"; tail -8 gpurun_out/tests_n_$TAG.log
timeout 600 python tools/layer_budget.py $TAG fp32 2>&1 | tail -24
timeout 300 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_strict_$TAG.json 2> gpurun_out/bench_strict_$TAG.err; echo "bench rc=$?"
tail -1 gpurun_out/bench_strict_$TAG.json | cut -c1-300
DL_NO_X3_GLDS=1 timeout 300 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_strict_old_$TAG.json 2> gpurun_out/bench_strict_old_$TAG.err; echo "bench(old path) rc=$?"
tail -1 gpurun_out/bench_strict_old_$TAG.json | cut -c1-300
