#!/bin/bash
# bench + rocprof summary; args: tag
TAG=${1:-x}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/prof_$TAG/*kernel_trace.csv
