timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=120 -k "narrow" 2>&1 | tail -3
for a in 0 2 5; do echo "ABL=$a"; DL_NARROW_ABL=$a timeout 100 python tools/narrow_probe.py 2>&1 | grep -E "us per" | head -1; done
