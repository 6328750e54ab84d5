#!/bin/bash
# diagnosis run: which of the deferred-reduction / one-rank RCCL tests dies, with full output
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_gpu_deferred.py -m gpu -v -x > gpurun_out/defer2_a.log 2>&1; echo "deferred rc=$?"
timeout 600 python -X faulthandler -m pytest tests/test_gpu_distributed.py -m gpu -v -x > gpurun_out/defer2_b.log 2>&1; echo "distributed rc=$?"
DL_WGRAD_DEFER=0 timeout 600 python -X faulthandler -m pytest tests/test_gpu_distributed.py -m gpu -v -x > gpurun_out/defer2_c.log 2>&1; echo "distributed (defer off) rc=$?"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-baseline-n8 --no-graph --no-timer-check --no-strict"
DL_DP_FORCE=1 timeout 600 $B > gpurun_out/defer2_dp.json 2> gpurun_out/defer2_dp.err; echo "dpforce rc=$?"
DL_DP_FORCE=1 DL_WGRAD_DEFER=0 timeout 600 $B > gpurun_out/defer2_dp0.json 2> gpurun_out/defer2_dp0.err; echo "dpforce (defer off) rc=$?"
for f in a b c; do echo "== $f"; grep -E "PASSED|FAILED|ERROR|Fatal|Error|error|fault|File \"/root" gpurun_out/defer2_$f.log | head -40; tail -5 gpurun_out/defer2_$f.log; done
tail -c 600 gpurun_out/defer2_dp.json; tail -15 gpurun_out/defer2_dp.err; tail -c 300 gpurun_out/defer2_dp0.json; tail -5 gpurun_out/defer2_dp0.err
