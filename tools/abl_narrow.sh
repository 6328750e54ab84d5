timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=120 -k "c4_patch" 2>&1 | grep -E "^E|passed|failed" | head -12
for a in 0 1 2 15; do echo "ABL=$a"; DL_C4_ABL=$a timeout 100 python tools/c4_probe.py 2>&1 | grep -E "stats True"; done
