#!/bin/bash
# short end-of-round verification: every GPU test, smoke, the contract line exactly as the driver runs it (tools/gpu_r04_final.sh adds the profiles)
TAG=${1:-r04}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/gpu_tests_$TAG.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/gpu_tests_$TAG.log | tail -12
cp gpurun_out/parity_errors.json gpurun_out/parity_errors_main_$TAG.json 2>/dev/null
cp gpurun_out/parity_errors_fullsize.json gpurun_out/parity_errors_fullsize_$TAG.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_contract_$TAG.json 2> gpurun_out/bench_contract_$TAG.err; echo "contract bench rc=$?"
python - <<PY
import json
d = json.loads(open('gpurun_out/bench_contract_$TAG.json').read().strip().splitlines()[-1])
r = d['roofline']
print('bench', d['value'], d['ms_per_step'], 'streams', d['config'].get('streams'), 'kernel', r['kernel'][:24], r['avg_launch_us'], 'frac', r['frac'], 'one-stream ms', r.get('one_stream_ms_per_step'),
      'concurrent us', r.get('concurrent_avg_launch_us'), 'strict', d['strict_parity'].get('value'), d['strict_parity'].get('ms_per_step'), 'graph', d.get('graph_replay'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
