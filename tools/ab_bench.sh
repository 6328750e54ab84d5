#!/bin/bash
# same-box A/B of the whole training step: tools/ab_bench.sh VAR   (runs VAR=0 / VAR=1 alternately, twice)
export TMPDIR=/tmp
VAR=${1:-DL_CONV_8PH}
for rep in 1 2; do
  for v in 0 1; do
    export $VAR=$v
    echo -n "$VAR=$v: "
    timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
