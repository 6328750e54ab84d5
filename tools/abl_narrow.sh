timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv" 2>&1 | tail -3
bash tools/gpu_pmc.sh fwd 8ph_r02 2>&1 | grep -E "BANK_CONFLICT|durations|MFMA_BUSY|FETCH|WRITE"
python tools/pmc_summarize.py 8ph_r02 gpurun_out/pmc_dominant_conv256_r02.json
for i in 1 2 3 4; do cp gpurun_out/pmc_8ph_r02/p$i/p_counter_collection.csv gpurun_out/conv8ph_r02_pass$i.csv 2>/dev/null; done
rm -rf gpurun_out/pmc_8ph_r02
