#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -k "wgrad or norm" > gpurun_out/run8_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/run8_tests.log
timeout 300 python tools/microbench.py > gpurun_out/run8_microbench.log 2>&1; grep "bf16" gpurun_out/run8_microbench.log | grep -v norm | python -c "
import sys,json
for l in sys.stdin:
    name=l.split('{')[0]; d=json.loads('{'+l.split('{',1)[1])
    print(name, {k: round(v,3) for k,v in d.items() if 'ms' in k}, 'wgrad TF', round(d['wgrad_tflops']))
"
