# scratch driver for the probe of the moment (rewritten per experiment)
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv" 2>&1 | tail -2
for v in 0 1; do echo "DL_CONV_PHASEFAST=$v"; DL_CONV_PHASEFAST=$v timeout 300 python tools/layer_budget.py r02i_$v 2>&1 | grep -E "^G down|^G up|^D c[1-4]|sum of" ; done
