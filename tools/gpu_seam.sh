#!/bin/bash
# thread-safety of the inference seam (must FAIL with process-wide scratch, PASS with the default), SDG on the GPU, 18-net step
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "--- shared scratch (expected: FAIL)"
DL_SHARED_SCRATCH=1 timeout 120 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=100 -k "thread_safe" 2>&1 | tail -4
echo "--- thread-local scratch (expected: pass)"
timeout 120 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=100 -k "thread_safe or sdg" 2>&1 | tail -3
echo "--- real DeepLIIF 18-net step"
timeout 170 python bench.py --workload train18 --steps 3 --warmup 1 > gpurun_out/bench_train18.json 2> gpurun_out/bench_train18.err; echo "train18 rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/bench_train18.json').read()); print(d['value'], d['ms_per_step'], d['model_tflops'], d['roofline']['avg_launch_us'], '| peak mem GB', d.get('peak_mem_gb'))" 2>&1 | tail -1
tail -2 gpurun_out/bench_train18.err
