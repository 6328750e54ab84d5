#!/bin/bash
# Round-6 PMC evidence (counters only, FETCH_SIZE and WRITE_SIZE in separate passes): the dominant kernel conv_gemm_w4_kernel (forward launches; its epilogue
# changed in r06) -> profiles/r06/pmc_dominant_conv256.json (read by bench.py for roofline.traffic); conv_s2d_kernel on down1 forward and conv_s2u_kernel on up2
# forward -> HBM-side bytes against the algorithmic 402.8 MB ("every input byte staged once")
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_pmc.sh fwd r06 > gpurun_out/pmc_r06_log.txt 2>&1
python tools/pmc_summarize.py r06 gpurun_out/pmc_dominant_conv256_r06.json
tail -1 gpurun_out/pmc_r06_log.txt; rm -rf gpurun_out/pmc_r06
A=$((8*512*512*64*2 + 8*256*256*128*2 + 128*576*2))
bash tools/gpu_pmc.sh down1 r06d > gpurun_out/pmc_r06d_log.txt 2>&1
python tools/pmc_summarize.py r06d gpurun_out/pmc_s2d_down1_r06.json "3x3 s2 64->128 @ 8x512x512 -> 256x256 bf16, conv_s2d_kernel (tools/conv_only.py down1)" $A
tail -1 gpurun_out/pmc_r06d_log.txt; rm -rf gpurun_out/pmc_r06d
bash tools/gpu_pmc.sh up2 r06u > gpurun_out/pmc_r06u_log.txt 2>&1
python tools/pmc_summarize.py r06u gpurun_out/pmc_s2u_up2_r06.json "convT 3x3 s2 128->64 @ 8x256x256 -> 512x512 bf16, conv_s2u_kernel (tools/conv_only.py up2)" $A
tail -1 gpurun_out/pmc_r06u_log.txt; rm -rf gpurun_out/pmc_r06u
