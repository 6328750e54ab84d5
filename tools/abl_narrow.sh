# scratch driver for the probe of the moment (rewritten per experiment)
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "narrow_transposed" 2>&1 | tail -4
for v in 0 1; do echo -n "DL_CONVT4=$v: "; DL_CONVT4=$v timeout 200 python bench.py --workload infer --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
