#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 1 ]; then export DL_CONV_STAGGER=1; else unset DL_CONV_STAGGER; fi
  echo "=== stagger=$v"
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -k "conv_forward" 2>&1 | tail -1
  timeout 300 python tools/microbench.py 2>/dev/null | grep "bf16" | grep -E "res3x3|D 512|convT" | python -c "
import sys,json
for l in sys.stdin:
    name=l.split('{')[0]; d=json.loads('{'+l.split('{',1)[1])
    print(name, {k: round(v,3) for k,v in d.items() if 'ms' in k})
"
done
