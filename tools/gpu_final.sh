#!/bin/bash
# bounded end-of-round verification: every command carries its own timeout (a hung rocprofv3 pass cost 15 GPU-minutes once)
TAG=${1:-final}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 330 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py -m gpu -q --timeout=120 2>&1 | tail -6 > gpurun_out/tests_$TAG.log; echo "tests rc=${PIPESTATUS[0]}"; cat gpurun_out/tests_$TAG.log
timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
python -c "import json; d=json.loads(open('gpurun_out/bench_$TAG.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['kernel'][:24])"
cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/prof_$TAG/*kernel_trace.csv
