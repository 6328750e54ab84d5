#!/bin/bash
# same-box A/B: share of the chip a weight-gradient launch is sized for (fewer split-K partials under branch streams)
mkdir -p gpurun_out; export TMPDIR=/tmp
{
B="--steps 15 --warmup 4 --no-cpu-baseline --no-cpu-baseline-n8 --no-graph --no-other-workloads --no-timer-check"
for v in "" "DL_WGRAD_FILL=0.5" "DL_WGRAD_FILL=0.34" "DL_WGRAD_BATCH_FILL=0.5" "DL_WGRAD_BATCH_FILL=0.25" "DL_WGRAD_FILL=0.5 DL_WGRAD_BATCH_FILL=0.5" ""; do
  echo "== $v"; env $v timeout 600 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')}, 'one-stream', d['roofline'].get('one_stream_ms_per_step'), 'strict', d['strict_parity']['value'])"
done
} > gpurun_out/r05_fill.txt 2>&1
cat gpurun_out/r05_fill.txt
