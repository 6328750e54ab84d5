#!/bin/bash
# re-collect the PMC passes of both ResnetBlock kernels on the final round-3 defaults (channel-chunk-major K order)
bash tools/gpu_pmc.sh fwd r03 > gpurun_out/pmc_r03_log.txt 2>&1
python tools/pmc_summarize.py r03 gpurun_out/pmc_dominant_conv256.json
rm -rf gpurun_out/pmc_r03
bash tools/gpu_r03_pmc_strict.sh
