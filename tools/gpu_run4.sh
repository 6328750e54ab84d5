#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -x -k "norm or elementwise" > gpurun_out/run4_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/run4_tests.log
timeout 300 python tools/microbench.py > gpurun_out/run4_microbench.log 2>&1; grep norm gpurun_out/run4_microbench.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/run4_bench.json 2> gpurun_out/run4_bench.err; echo "bench rc=$?"; cat gpurun_out/run4_bench.json; tail -3 gpurun_out/run4_bench.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01b -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/run4_prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_r01b -type f | head; rm -f gpurun_out/prof_r01b/*kernel_trace.csv
