#!/bin/bash
# fused four-phase tile (csrc/conv_s2f.hip): parity, isolated layers with / without it (two alternations), whole steps, sustained-MFMA clock probe
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/s2f_${1:-a}.txt
rm -f $O
echo "== parity (kernel tests bf16 + network tests)" >> $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "bf16 and (big_tiles or fused_norm_statistics or conv_forward_and_dgrad)" 2>&1 | tail -4 >> $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_networks.py -m gpu -q -k "full_size or benched or network_forward_backward" 2>&1 | tail -4 >> $O
for rep in 1 2; do
  for v in 0 1; do
    echo "== layers DL_CONV_S2F=$v (round $rep)" >> $O
    DL_CONV_S2F=$v timeout 300 python tools/s2f_time.py 2>/dev/null | grep -v '^{' >> $O
  done
done
echo "== whole steps (DL_CONV_S2F)" >> $O
for v in 0 1 0 1; do
  DL_CONV_S2F=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check 2>/dev/null | tail -1 > gpurun_out/s2f_bench_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/s2f_bench_$v.json').read()); r=d['roofline']; print('S2F=$v', d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'])" >> $O
done
echo "== sustained MFMA probe with clocks (rocm-smi)" >> $O
timeout 120 python tools/mfma_clock.py >> $O 2>&1
cat $O
