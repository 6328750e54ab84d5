#!/bin/bash
# round 6 mid-round regression on ONE box: every GPU test, smoke, a short bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-a}
timeout 2400 python -m pytest tests -m gpu -q --timeout=1200 > gpurun_out/gpu_tests_$TAG.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/gpu_tests_$TAG.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-other-workloads --no-graph 2>/dev/null | tail -1 > gpurun_out/bench_mid_$TAG.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_mid_$TAG.json').read()); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('one_stream_ms_per_step'), 'strict', d['strict_parity'].get('value'))"
