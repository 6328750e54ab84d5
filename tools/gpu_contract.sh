#!/bin/bash
# exactly what the driver runs at round end: smoke(), then the default bench line (with the cpu_baseline leg)
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
t1=$(date +%s); echo "smoke wall $((t1-t0)) s"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
t2=$(date +%s); echo "bench wall $((t2-t1)) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
need=['metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data','config','roofline','cpu_baseline']
print('missing keys:', [k for k in need if k not in d])
print({k:d[k] for k in ['value','ms_per_step','steps','warmup','n_gpus','dtype']})
print('roofline', {k:d['roofline'][k] for k in ['bound','achieved','peak','unit','frac','traffic']}, d['roofline']['kernel'][:60])
print('cpu_baseline', d['cpu_baseline'])
print('lines on stdout:', len(open('gpurun_out/bench_default.json').read().strip().splitlines()))
PY
tail -3 gpurun_out/bench_default.err
