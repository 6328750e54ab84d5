#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --workload infer --steps 3 --warmup 1 > gpurun_out/bench_infer.json 2> gpurun_out/bench_infer.err; echo "bench rc=$?"; cat gpurun_out/bench_infer.json; tail -5 gpurun_out/bench_infer.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_infer -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload infer --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_infer.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/prof_infer/*kernel_trace.csv
