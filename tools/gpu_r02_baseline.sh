#!/bin/bash
# round-2 baseline on ONE box: per-layer time budget, the bf16 bench line, the strict (fp32 = split-bf16x3) policy's step time
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python tools/layer_budget.py r02_base bf16 > gpurun_out/layer_budget_r02_base.log 2>&1; echo "budget rc=$?"
tail -32 gpurun_out/layer_budget_r02_base.log
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r02_base.json 2> gpurun_out/bench_r02_base.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp32 > gpurun_out/bench_r02_base_fp32.json 2> gpurun_out/bench_r02_base_fp32.err; echo "bench fp32 rc=$?"
python - <<'PY'
import json
for f in ('bench_r02_base', 'bench_r02_base_fp32'):
    try:
        d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'] if d.get('roofline') else None)
    except Exception as e:
        print(f, 'failed', e)
PY
