#!/bin/bash
# round 5, run 20 (last): the whole GPU suite on the final code (tiled slab reduction), then the contract line without the side legs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 680 python -m pytest tests -q -m gpu --durations=5 > gpurun_out/r05_gpu_tests_final2.log 2>&1; echo "pytest rc=$?" > gpurun_out/final_r05c.txt
grep -E "passed|failed" gpurun_out/r05_gpu_tests_final2.log | tail -3 >> gpurun_out/final_r05c.txt
timeout 110 python bench.py --no-cpu-baseline --no-other-workloads > gpurun_out/r05_bench_final2.json 2> gpurun_out/r05_bench_final2.err; echo "bench rc=$?" >> gpurun_out/final_r05c.txt
tail -1 gpurun_out/r05_bench_final2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, 'frac', d['roofline']['frac'], 'strict', (d.get('strict_parity') or {}).get('value'), 'graph', (d.get('graph_replay') or {}))" >> gpurun_out/final_r05c.txt 2>&1
cat gpurun_out/final_r05c.txt
