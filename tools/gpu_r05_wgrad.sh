#!/bin/bash
# round 5, first GPU look: parity of the w4 weight gradient + the batched launch, then same-box A/Bs of every variant (each its own process)
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_wgrad_batch.py tests/test_gpu_deferred.py -q 2>&1 | tail -25
echo "== full-size step under the oracle"; timeout 900 python -m pytest tests/test_gpu_fullsize_step.py -q 2>&1 | tail -25
for p in bf16 fp32; do
  echo "== $p: r04 kernels (builtin transposing reads)";  DL_NO_WGRAD_W4=1 DL_WGRAD_TR_ASM=0 timeout 300 python tools/wgrad_time.py $p
  echo "== $p: r04 kernels, inline-asm transposing reads"; DL_NO_WGRAD_W4=1 timeout 300 python tools/wgrad_time.py $p
  echo "== $p: default"; timeout 300 python tools/wgrad_time.py $p
done
echo "== fp32 split copies, builtin"; TIME_SPLIT=1 DL_WGRAD_TR_ASM=0 timeout 300 python tools/wgrad_time.py fp32
echo "== fp32 split copies, asm"; TIME_SPLIT=1 timeout 300 python tools/wgrad_time.py fp32
for v in "DL_WGRAD_BATCH=0 DL_NO_WGRAD_W4=1 DL_WGRAD_TR_ASM=0" "DL_WGRAD_BATCH=0 DL_NO_WGRAD_W4=1" "DL_WGRAD_BATCH=0" ""; do
  echo "== bench bf16: $v"; env $v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')}, d.get('roofline',{}).get('frac'))"
done
for v in "DL_WGRAD_BATCH=0 DL_WGRAD_TR_ASM=0" "DL_WGRAD_BATCH=0" ""; do
  echo "== bench fp32: $v"; env $v timeout 600 python bench.py --steps 6 --warmup 2 --precision fp32 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')})"
done
} > gpurun_out/r05_wgrad.txt 2>&1
tail -60 gpurun_out/r05_wgrad.txt
