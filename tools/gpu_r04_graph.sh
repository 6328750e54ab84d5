#!/bin/bash
# hipGraph capture of the training step: bit-identity tests, then the default bench line (eager headline + graph_replay leg + strict leg)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -q 2>&1 | tail -8 > gpurun_out/graph_tests.log
cat gpurun_out/graph_tests.log
timeout 900 python -m pytest tests/test_gpu_switches.py -m gpu -q 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-baseline-n8 2>gpurun_out/bench_graph.err | tail -1 > gpurun_out/bench_graph.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_graph.json').read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('sustained'), d.get('graph_replay'), d['strict_parity'].get('value'))"
tail -3 gpurun_out/bench_graph.err
