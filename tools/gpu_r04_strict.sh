#!/bin/bash
# strict-policy kernels of round 4: conv_narrow_roll_x3_kernel (head forward) and conv_s2f_x3_kernel (fused four-phase stride-2 tile) -- parity tests, the
# full-size oracle tests, the strict layer budget, and a same-box A/B of the strict step with each kernel switched off
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "narrow or strict_fused or split_copy" > gpurun_out/strict_tests.log 2>&1; echo "kernel tests rc=$?"
grep -E "passed|failed|Error|error|assert" gpurun_out/strict_tests.log | tail -8
timeout 900 python -m pytest tests/test_gpu_switches.py -m gpu -q -x -k "below_its_size_rule" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -4
B="python bench.py --precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-baseline-n8 --no-graph --no-timer-check --no-strict"
run() {
    tag=$1; shift
    env "$@" timeout 600 $B 2>gpurun_out/bench_strict_$tag.err | tail -1 > gpurun_out/bench_strict_$tag.json
    python - "$tag" <<'P'
import json, sys
d = json.loads(open(f'gpurun_out/bench_strict_{sys.argv[1]}.json').read())
print(sys.argv[1], d['value'], d['ms_per_step'])
P
}
run base DL_CONV_S2FX3=0 DL_NO_NARROW_X3=1
run new X=1
run nohead DL_NO_NARROW_X3=1
run nos2f DL_CONV_S2FX3=0
run base2 DL_CONV_S2FX3=0 DL_NO_NARROW_X3=1
run new2 X=1
LB_SPLIT=1 timeout 900 python tools/layer_budget.py r04s fp32 > gpurun_out/layer_budget_r04s.log 2>&1; grep -E "^G |^D |^sum" gpurun_out/layer_budget_r04s.log | head -20
LB_SPLIT=1 DL_CONV_S2FX3=0 DL_NO_NARROW_X3=1 timeout 900 python tools/layer_budget.py r04s_base fp32 > gpurun_out/layer_budget_r04s_base.log 2>&1; grep -E "^G |^D |^sum" gpurun_out/layer_budget_r04s_base.log | head -20
