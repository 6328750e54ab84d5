#!/bin/bash
# round 5, run 25 (last seconds of the budget): the tiled weight packing -- bit-exactness against the single-image kernel, timing against the chunk form, two bench lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/r05_pack.txt
timeout 40 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pack_weights_batch" 2>&1 | grep -E "passed|failed|rror" | tail -4 > $O
timeout 30 python tools/pack_time.py >> $O 2>&1; echo "pack_time rc=$?" >> $O
for w in train18 ext; do
  timeout 25 python bench.py --workload $w --no-cpu-baseline --no-other-workloads --no-strict --no-graph --no-timer-check --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])" >> $O 2>&1
done
cat $O
