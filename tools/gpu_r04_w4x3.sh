#!/bin/bash
# conv_gemm_w4x3_kernel (strict policy, csrc/conv_w4x3.hip): parity, isolated launches on split copies against the 8-phase strict kernel, whole strict steps
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/w4x3_${1:-a}.txt
rm -f $O
echo "== parity" >> $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "strict_w4_kernel or split_copy_inputs" 2>&1 | tail -6 >> $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "full_size_forward or batch8" 2>&1 | tail -4 >> $O
for rep in 1 2; do
  echo "== isolated launches on split copies, round $rep (DL_CONV_W4X3=0 / 1)" >> $O
  for v in 0 1; do DL_CONV_W4X3=$v TIME_SPLIT=1 timeout 120 python tools/conv_time.py fp32 fwd,dgrad 2>/dev/null | tail -1 >> $O; done
done
echo "== ablations (timing only): 1 = no DMA, 3 = MFMA only, 4 = prologue + epilogue" >> $O
for v in 1 3 4; do DL_W4X3_ABLATE=$v TIME_SPLIT=1 timeout 120 python tools/conv_time.py fp32 fwd 2>/dev/null | tail -1 >> $O; done
echo "== whole strict steps (DL_CONV_W4X3)" >> $O
for v in 0 1 0 1; do
  DL_CONV_W4X3=$v timeout 300 python bench.py --precision fp32 --steps 8 --warmup 2 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check 2>/dev/null | tail -1 > gpurun_out/w4x3_bench_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/w4x3_bench_$v.json').read()); r=d['roofline']; print('W4X3=$v', d['value'], d['ms_per_step'], r['kernel'][:24], r['avg_launch_us'], r['frac'])" >> $O
done
cat $O
