#!/bin/bash
# same-box A/B of the inference workload: tools/ab_infer.sh VAR  (VAR=0 / VAR=1 alternately, twice) + UNet parity tests with VAR=1
export TMPDIR=/tmp
VAR=${1:-DL_CONV_PREACT}
for rep in 1 2; do
  for v in 0 1; do
    export $VAR=$v
    echo -n "$VAR=$v: "
    timeout 600 python bench.py --workload infer --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
export $VAR=1
timeout 800 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=600 -x 2>&1 | tail -2
