#!/bin/bash
# DL_STREAMS_SEG=1 (opt-in): branch streams for the model with segmentation generators -- bit-identity test, then the 18-net step with and without
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_streams.py -m gpu -q -x -k "seg_model" > gpurun_out/seg_streams_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert" gpurun_out/seg_streams_tests.log | tail -6
for v in 0 1 0 1; do
  DL_STREAMS_SEG=$v timeout 120 python bench.py --workload train18 --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-timer-check --no-graph 2>gpurun_out/bench_segs_$v.err | tail -1 > gpurun_out/bench_segs_$v.json
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/bench_segs_$v.json').read())
    print('train18 DL_STREAMS_SEG=$v', d['value'], d['ms_per_step'], 'streams', d['config'].get('streams'))
except Exception as e:
    print('failed', e); print(open('gpurun_out/bench_segs_$v.err').read()[-600:])
PY
done
