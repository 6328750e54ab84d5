#!/bin/bash
# DL_INFER_STREAMS (opt-in): bit-identity test, then the inference DAG and the whole-slide workload with 1 / 2 / 3 / 5 streams on one box
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_infer_streams.py -m gpu -q -x > gpurun_out/infer_streams_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert" gpurun_out/infer_streams_tests.log | tail -6
for w in infer wsi; do
  for n in 1 3 5 2 1; do
    DL_INFER_STREAMS=$n timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-timer-check --no-graph 2>gpurun_out/bench_is_${w}_$n.err | tail -1 > gpurun_out/bench_is_${w}_$n.json
    python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/bench_is_${w}_$n.json').read())
    print('$w', 'streams $n', d['value'], d['ms_per_step'])
except Exception as e:
    print('$w $n failed', e); print(open('gpurun_out/bench_is_${w}_$n.err').read()[-500:])
PY
  done
done
