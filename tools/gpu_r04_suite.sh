#!/bin/bash
# Round-4 verification on ONE box: every GPU test (w4 kernel is the default), then the bench line with the strict leg.
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-a}
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/gpu_tests_r04$T.log
cat gpurun_out/gpu_tests_r04$T.log
cp gpurun_out/parity_errors_fullsize.json gpurun_out/parity_errors_fullsize_r04$T.json 2>/dev/null
cp gpurun_out/parity_errors.json gpurun_out/parity_errors_r04$T.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-baseline-n8 2>/dev/null | tail -1 > gpurun_out/bench_r04$T.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_r04$T.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['strict_parity'].get('value'), d['strict_parity'].get('ms_per_step'))"
