#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_seam.py -m gpu -q -x --timeout=600 2>&1 | grep -v "Warning\|warn" | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_new.json 2> gpurun_out/bench_new.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_new.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['launches_timed'], d['roofline'].get('timer_overhead'))
print('strict', d['strict_parity']['value'], d['strict_parity']['roofline'])
print(d['cpu_baseline']); print(d.get('cpu_baseline_n8'))
PY
