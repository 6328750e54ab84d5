#!/bin/bash
# Round 4: conv_gemm_w4_kernel (csrc/conv_w4.hip) against the 8-phase kernel on ONE box: parity under the switch, isolated launches (two
# alternations), schedule variants DL_W4_VAR=0..3, timing-only ablations, whole steps.  Every command has its own timeout.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/w4_${1:-a}.txt
rm -f $O
K='(big_tiles or fused_norm_statistics) and bf16'
KS='(big_tiles or fused_norm_statistics) and bf16 and (conv256-256k3s1n8 or conv192-256 or conv64-256k3s1n16)'
for v in 1 3; do
  echo "== parity DL_CONV_W4=1 DL_W4_VAR=$v" >> $O
  DL_CONV_W4=1 DL_W4_VAR=$v timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "$K" 2>&1 | tail -2 >> $O
done
for v in 0 2; do
  echo "== parity (eligible cases) DL_CONV_W4=1 DL_W4_VAR=$v" >> $O
  DL_CONV_W4=1 DL_W4_VAR=$v timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "$KS" 2>&1 | tail -2 >> $O
done
for rep in 1 2; do
  echo "== isolated launches, round $rep" >> $O
  timeout 120 python tools/conv_time.py bf16 fwd,dgrad 2>/dev/null | tail -1 >> $O
  for v in 0 1 2 3; do DL_CONV_W4=1 DL_W4_VAR=$v timeout 120 python tools/conv_time.py bf16 fwd,dgrad 2>/dev/null | tail -1 >> $O; done
done
echo "== ablations of w4 (timing only): 1 = no DMA, 2 = DMA only, 3 = MFMA only, 4 = prologue + epilogue, 5 = DMA never waited for" >> $O
for v in 1 2 3 4 5; do DL_CONV_W4=1 DL_W4_ABLATE=$v timeout 120 python tools/conv_time.py bf16 fwd 2>/dev/null | tail -1 >> $O; done
echo "== zero data (DVFS)" >> $O
TIME_DATA=zero timeout 120 python tools/conv_time.py bf16 fwd 2>/dev/null | tail -1 >> $O
for v in 1 3; do TIME_DATA=zero DL_CONV_W4=1 DL_W4_VAR=$v timeout 120 python tools/conv_time.py bf16 fwd 2>/dev/null | tail -1 >> $O; done
echo "== whole steps (DL_CONV_W4 / DL_W4_VAR)" >> $O
for cfg in "0 1" "1 1" "1 3" "0 1" "1 1" "1 3"; do
  set -- $cfg
  DL_CONV_W4=$1 DL_W4_VAR=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check 2>/dev/null | tail -1 > gpurun_out/w4_bench_$1_$2.json
  python -c "
import json; d=json.loads(open('gpurun_out/w4_bench_$1_$2.json').read()); print('DL_CONV_W4=$1 VAR=$2', d['value'], d['ms_per_step'], d['roofline'].get('kernel','')[:22], d['roofline']['avg_launch_us'], d['roofline']['frac'])" >> $O
done
cat $O
