#!/bin/bash
# round 5: the COMPLETE GPU suite under the new defaults (DL_STREAMS_SEG, DL_STREAMS_EXT, DL_INFER_STREAMS=3, batched weight gradient), then the contract line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r05_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_gpu_tests.log
tail -30 gpurun_out/r05_gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_contract.json 2> gpurun_out/r05_bench_contract.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_contract.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, 'frac', d['roofline']['frac'], 'strict', d['strict_parity']['value'] if d.get('strict_parity') else None)
print('cpu', d.get('cpu_baseline', {}).get('value'))
for k, v in (d.get('other_workloads') or {}).items():
    print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'streams', 'wall_s', 'error')}, (v.get('roofline') or {}).get('frac'), v.get('whole_slide', {}).get('tiles_per_s') if v.get('whole_slide') else '', (v.get('cpu_baseline') or {}).get('value'))
PY
