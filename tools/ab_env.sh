#!/bin/bash
# same-box A/B of a HOST-side (Python) switch over workloads: tools/ab_env.sh VAR "wl1 wl2 ..."   (VAR=0 / VAR=1 alternately, three rounds)
export TMPDIR=/tmp
VAR=$1; WLS=${2:-train18}
for rep in 1 2 3; do
  for v in 0 1; do
    for wl in $WLS; do
      env $VAR=$v python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', '$wl', d['value'], d['ms_per_step'])"
    done
  done
done
