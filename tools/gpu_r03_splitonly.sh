#!/bin/bash
# the norm backward skips its fp32 store where the gradient's consumers all read the split copy (DL_NO_SPLIT_ONLY_GRAD=1 restores it): parity + strict step A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_networks.py -m gpu -x -q -k "split_copies or golden_fixture or teacher_forced_layer or benched or thread_safe or against_oracle" 2>&1 | tail -3
for v in 1 0 1 0; do
  DL_NO_SPLIT_ONLY_GRAD=$v python bench.py --precision fp32 --steps 8 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-timer-check 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('DL_NO_SPLIT_ONLY_GRAD=$v', d['value'], d['ms_per_step'])"
done
