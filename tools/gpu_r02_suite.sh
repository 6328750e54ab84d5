#!/bin/bash
# round-2 verification on ONE box: every GPU test, then the default bench line (with the strict-parity leg) and the wsi workload
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout=300 2>&1 | tail -15 > gpurun_out/tests_$TAG.log; echo "tests rc=${PIPESTATUS[0]}"; cat gpurun_out/tests_$TAG.log
cp gpurun_out/parity_errors.json gpurun_out/parity_errors_$TAG.json 2>/dev/null
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
timeout 300 python bench.py --workload wsi --steps 254 --warmup 2 > gpurun_out/bench_wsi_$TAG.json 2> gpurun_out/bench_wsi_$TAG.err; echo "wsi rc=$?"
python - <<PY
import json
for f in ('bench_$TAG', 'bench_wsi_$TAG'):
    try:
        d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['steps'], (d.get('strict_parity') or {}).get('value'), (d.get('strict_parity') or {}).get('headline_vs_strict'))
    except Exception as e:
        print(f, 'failed', e); print(open(f'gpurun_out/{f}.err').read()[-1500:])
PY
