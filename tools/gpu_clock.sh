#!/bin/bash
# sclk / socket power while single kernels run back to back (evidence for the power cap): writes gpurun_out/clock_probe.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  echo "# tools/clock_probe.py: (sclk MHz, socket W) sampled every 0.2 s; 3x3 256->256 conv @ 8x128x128 bf16, random data"
  python tools/clock_probe.py fwd 3
  DL_CONV_ABLATE=3 python tools/clock_probe.py fwd 3      # MFMAs only (timing-only ablation)
  DL_CONV_ABLATE=2 python tools/clock_probe.py fwd 3      # DMA only
  DL_CONV_8PH=0 python tools/clock_probe.py fwd 3         # one-barrier kernel
  python tools/clock_probe.py wgrad 3
  python tools/clock_probe.py norm 3
} > gpurun_out/clock_probe.txt 2>/dev/null
cat gpurun_out/clock_probe.txt | cut -c1-260
