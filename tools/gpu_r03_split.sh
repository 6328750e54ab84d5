#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -k "split_copy" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=600 -k "split_copies or golden_fixture" 2>&1 | tail -8
timeout 300 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-timer-check > gpurun_out/bench_split.json 2> gpurun_out/bench_split.err; echo "bench rc=$?"; tail -1 gpurun_out/bench_split.json | cut -c1-200
DL_NO_SPLIT_COPY=1 timeout 300 python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-timer-check > gpurun_out/bench_nosplit.json 2> gpurun_out/bench_nosplit.err; echo "bench(no split copies) rc=$?"; tail -1 gpurun_out/bench_nosplit.json | cut -c1-200
