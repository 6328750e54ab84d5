#!/bin/bash
# round 6: the fp16 inference policy (libdeepliif_hip_f16.so) -- its GPU tests, the smoke, the infer / wsi workloads on both 16-bit formats
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/fp16_${1:-a}.txt
rm -f $O
echo "== tests/test_gpu_fp16.py" >> $O
timeout 2400 python -m pytest tests/test_gpu_fp16.py -m gpu -q -x 2>&1 | tail -25 >> $O
echo "== smoke" >> $O
timeout 600 python -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 >> $O
if [ "$2" != "nobench" ]; then
for p in bf16 fp16; do
  for wl in infer wsi; do
    echo "== bench --workload $wl --precision $p" >> $O
    timeout 900 python bench.py --workload $wl --precision $p --steps 5 --warmup 2 --no-cpu-baseline --no-timer-check 2>/dev/null | tail -1 > gpurun_out/fp16_bench_${wl}_${p}.json
    python -c "
import json; d=json.loads(open('gpurun_out/fp16_bench_${wl}_${p}.json').read()); r=d.get('roofline') or {}
print(d['value'], 'tiles/s', d['ms_per_step'], 'ms', d['dtype'], r.get('kernel'), r.get('avg_launch_us'), r.get('frac'))
print(json.dumps(d.get('policy_vs_strict'))); print(json.dumps(d.get('whole_slide')))" >> $O 2>&1
  done
done
fi
cat $O
