timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "wgrad or weight_grad" 2>&1 | tail -2
for rep in 1 2; do for v in 0 1; do echo -n "DL_WGRAD_8PH=$v: "; DL_WGRAD_8PH=$v BLK_WGRAD=1 timeout 100 python tools/blk_probe.py 2>&1 | tail -1; done; done
