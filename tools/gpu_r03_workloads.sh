#!/bin/bash
# the other bench workloads (configs[1], configs[3], the real 18-net step) on ONE box, for the round's record
TAG=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in train18 ext infer wsi; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-timer-check > gpurun_out/bench_${w}_$TAG.json 2> gpurun_out/bench_${w}_$TAG.err; echo "$w rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/bench_${w}_$TAG.json').read().strip().splitlines()[-1])
    print('$w', d['value'], d['unit'], d['ms_per_step'], 'ms/step')
except Exception as e:
    print('$w failed', e); print(open('gpurun_out/bench_${w}_$TAG.err').read()[-800:])
PY
done
