#!/bin/bash
# Round-5 PMC evidence for the dominant kernel (conv_gemm_w4_kernel, forward launches; the epilogue changed in r05) -> the file bench.py reads for roofline.traffic
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_pmc.sh fwd r05 > gpurun_out/pmc_r05_log.txt 2>&1
python tools/pmc_summarize.py r05 gpurun_out/pmc_dominant_conv256_r05.json
tail -3 gpurun_out/pmc_r05_log.txt
rm -rf gpurun_out/pmc_r05
