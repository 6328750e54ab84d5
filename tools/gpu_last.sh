#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 90 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], '| cpu:', d['cpu_baseline'])"
timeout 120 python bench.py --workload infer --steps 5 --warmup 2 > gpurun_out/bench_infer.json 2>/dev/null; echo "infer rc=$?"
python -c "import json; d=json.loads(open('gpurun_out/bench_infer.json').read()); print('infer', d['value'], d['ms_per_step'])"
for m in 0.5 1 2 4; do DL_NORM_GRID_MUL=$m timeout 60 python tools/norm_time.py 2>/dev/null | tail -1; done
