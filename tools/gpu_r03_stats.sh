#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -k "fused_norm_statistics" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=600 -k "golden_fixture or teacher_forced_layer or network_forward" 2>&1 | tail -6
timeout 300 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-timer-check > gpurun_out/bench_stats.json 2> gpurun_out/bench_stats.err; echo "bench rc=$?"; tail -1 gpurun_out/bench_stats.json | cut -c1-200
DL_NO_X3_STATS=1 timeout 300 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-timer-check > gpurun_out/bench_nostats.json 2> gpurun_out/bench_nostats.err; echo "bench(no fused stats) rc=$?"; tail -1 gpurun_out/bench_nostats.json | cut -c1-200
