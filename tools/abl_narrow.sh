timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "c4_weight" 2>&1 | tail -3
for a in 0 1 2 4 7 8 16; do echo "ABL=$a"; DL_WC4_ABL=$a timeout 100 python tools/wc4_probe.py 2>&1 | grep "us/launch"; done
