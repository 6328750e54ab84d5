#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/run3_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/run3_smoke.log; tail -3 gpurun_out/run3_smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/run3_bench.json 2> gpurun_out/run3_bench.err; echo "bench rc=$?"; cat gpurun_out/run3_bench.json; tail -5 gpurun_out/run3_bench.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/run3_prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; ls -la gpurun_out/prof_r01 | head; find gpurun_out/prof_r01 -name "*stats*" | head
