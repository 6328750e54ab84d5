#!/bin/bash
# round 5, extra evidence: PMC of the strict weight gradient with the inline-asm transposing reads, the contract line on BatchNorm
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== PMC strict weight gradient (split copies, one launch per layer)"
TIME_SPLIT=1 TIME_ONLY=deferred bash tools/gpu_pmc_any.sh wgrad_x3_r05 wgrad_glds_x3 python $GRAFT_REPO_ROOT/tools/wgrad_time.py fp32 18 2>&1 | tail -2
echo "== the same with the builtin reads (DL_WGRAD_TR_ASM=0)"
DL_WGRAD_TR_ASM=0 TIME_SPLIT=1 TIME_ONLY=deferred bash tools/gpu_pmc_any.sh wgrad_x3_builtin_r05 wgrad_glds_x3 python $GRAFT_REPO_ROOT/tools/wgrad_time.py fp32 18 2>&1 | tail -2
echo "== contract workload with --norm batch (the reference CLI's default norm)"
timeout 600 python bench.py --norm batch --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-baseline-n8 --no-graph --no-other-workloads > gpurun_out/bench_normbatch_r05.json 2>/dev/null
python -c "import json; d=json.loads(open('gpurun_out/bench_normbatch_r05.json').read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], 'strict', d['strict_parity']['value'])"
} > gpurun_out/r05_extra.txt 2>&1
cat gpurun_out/r05_extra.txt
