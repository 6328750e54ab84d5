#!/bin/bash
# stream-count sweep of the workloads whose chains moved onto streams in round 5
mkdir -p gpurun_out; export TMPDIR=/tmp
{
B="--steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-graph --no-timer-check --no-other-workloads --no-wsi-whole"
for ns in 3 4 5; do
  echo "== infer DL_INFER_STREAMS=$ns"; DL_INFER_STREAMS=$ns timeout 300 python bench.py --workload infer $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')})"
done
for ns in 3 4 5; do
  echo "== train18 DL_STREAMS=$ns"; DL_STREAMS=$ns timeout 300 python bench.py --workload train18 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')})"
done
for ns in 2 3; do
  echo "== ext DL_STREAMS=$ns"; DL_STREAMS=$ns timeout 300 python bench.py --workload ext $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')})"
done
} > gpurun_out/r05_streams.txt 2>&1
cat gpurun_out/r05_streams.txt
