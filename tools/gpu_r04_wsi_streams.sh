export TMPDIR=/tmp
mkdir -p gpurun_out
for n in 1 2 3 1 3 2; do
  DL_INFER_STREAMS=$n timeout 300 python bench.py --workload wsi --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-timer-check --no-graph 2>gpurun_out/bench_is2_wsi_$n.err | tail -1 > gpurun_out/bench_is2_wsi_$n.json
  python - <<PY
import json
d = json.loads(open('gpurun_out/bench_is2_wsi_$n.json').read())
print('wsi', 'streams $n', d['value'], d['ms_per_step'])
PY
done
