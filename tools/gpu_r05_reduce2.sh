#!/bin/bash
# round 5, run 19: the tiled (coalesced-write) slab reduction -- targeted parity tests, then the workloads it matters for
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_deferred.py tests/test_gpu_wgrad_batch.py tests/test_gpu_networks.py -q -m gpu -k "wgrad or defer or batched or unet" -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/r05_reduce2.txt
for w in train18 ext train; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline --no-other-workloads --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), (d.get('strict_parity') or {}).get('value'))" >> gpurun_out/r05_reduce2.txt 2>&1
done
cat gpurun_out/r05_reduce2.txt
