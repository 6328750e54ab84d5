#!/bin/bash
# deferred reduction of the weight gradient's split-K slabs (dl_conv_wgrad_slabs / dl_wgrad_reduce_batch): bit-identity tests, then a same-box A/B of the
# training step over DL_WGRAD_DEFER / DL_WGRAD_ARENA_MB (bf16 + strict), and one DL_DP_FORCE=1 line (one-rank RCCL: the `exchange` block, flush before the wire)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_distributed.py -m gpu -q -x > gpurun_out/defer_tests.log 2>&1
grep -E 'passed|failed|Error|error' gpurun_out/defer_tests.log | tail -12
timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -q -x 2>&1 | tail -6 | tee -a gpurun_out/defer_tests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-baseline-n8 --no-graph --no-timer-check"
run() {   # tag, env...
    tag=$1; shift
    env "$@" timeout 600 $B 2>gpurun_out/bench_defer_$tag.err | tail -1 > gpurun_out/bench_defer_$tag.json
    python - "$tag" <<'P'
import json, sys
d = json.loads(open(f'gpurun_out/bench_defer_{sys.argv[1]}.json').read())
print(sys.argv[1], d['value'], d['ms_per_step'], 'strict', d.get('strict_parity', {}).get('value'), d.get('strict_parity', {}).get('ms_per_step'), 'exchange', d.get('exchange'))
P
}
run off DL_WGRAD_DEFER=0
run a2048 DL_WGRAD_ARENA_MB=2048
run a256 DL_WGRAD_ARENA_MB=256
run a12288 DL_WGRAD_ARENA_MB=12288
run off2 DL_WGRAD_DEFER=0
B="$B --no-strict"
run dpforce DL_DP_FORCE=1
tail -2 gpurun_out/bench_defer_dpforce.err
