#!/bin/bash
# Round 4, first look at conv_gemm_w4_kernel (csrc/conv_w4.hip): parity under the switch, isolated launches (two alternations, one box),
# timing-only ablations of both kernels on today's K order, then whole steps.  One gpurun call; every command has its own timeout.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/w4_${1:-a}.txt
rm -f $O
echo "== parity (DL_CONV_W4=1, SCHED=1 and 0)" >> $O
DL_CONV_W4=1 DL_W4_SCHED=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "(big_tiles or fused_norm_statistics) and bf16" 2>&1 | tail -3 >> $O
DL_CONV_W4=1 DL_W4_SCHED=0 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "(big_tiles or fused_norm_statistics) and bf16" 2>&1 | tail -3 >> $O
for rep in 1 2; do
  echo "== isolated launches, round $rep" >> $O
  timeout 120 python tools/conv_time.py bf16 fwd,dgrad 2>/dev/null | tail -1 >> $O
  DL_CONV_W4=1 DL_W4_SCHED=0 timeout 120 python tools/conv_time.py bf16 fwd,dgrad 2>/dev/null | tail -1 >> $O
  DL_CONV_W4=1 DL_W4_SCHED=1 timeout 120 python tools/conv_time.py bf16 fwd,dgrad 2>/dev/null | tail -1 >> $O
done
echo "== ablations (timing only): 8-phase 1 = no DMA, 2 = DMA only, 3 = MFMA only, 4 = prologue + epilogue; w4 1/2/3 the same" >> $O
for v in 1 2 3 4; do DL_CONV_ABLATE=$v timeout 120 python tools/conv_time.py bf16 fwd 2>/dev/null | tail -1 >> $O; done
for v in 1 2 3; do DL_CONV_W4=1 DL_W4_ABLATE=$v timeout 120 python tools/conv_time.py bf16 fwd 2>/dev/null | tail -1 >> $O; done
echo "== zero data (DVFS)" >> $O
TIME_DATA=zero timeout 120 python tools/conv_time.py bf16 fwd 2>/dev/null | tail -1 >> $O
TIME_DATA=zero DL_CONV_W4=1 DL_W4_SCHED=1 timeout 120 python tools/conv_time.py bf16 fwd 2>/dev/null | tail -1 >> $O
echo "== whole steps" >> $O
for v in 0 1 0 1; do
  DL_CONV_W4=$v DL_W4_SCHED=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check 2>/dev/null | tail -1 > gpurun_out/w4_bench_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/w4_bench_$v.json').read()); print('DL_CONV_W4=$v', d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline']['avg_launch_us'], d['roofline']['frac'])" >> $O
done
cat $O
