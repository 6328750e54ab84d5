#!/bin/bash
# Round-4 end-of-round verification on ONE box: every GPU test, smoke, the contract line exactly as the driver runs it (cpu_baseline legs included),
# rocprofv3 --kernel-trace --stats of the bf16 and of the strict step (on ONE stream, DL_STREAMS=1: per-kernel durations that mean the kernel, not its share of the
# GPU next to another branch), the bf16 layer budget, the other workloads.  Every command has its own timeout.
TAG=${1:-r04}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/gpu_tests_$TAG.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/gpu_tests_$TAG.log | tail -12
cp gpurun_out/parity_errors.json gpurun_out/parity_errors_main_$TAG.json 2>/dev/null
cp gpurun_out/parity_errors_fullsize.json gpurun_out/parity_errors_fullsize_$TAG.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_contract_$TAG.json 2> gpurun_out/bench_contract_$TAG.err; echo "contract bench rc=$?"
python - <<PY
import json
d = json.loads(open('gpurun_out/bench_contract_$TAG.json').read().strip().splitlines()[-1])
r = d['roofline']
print('bench', d['value'], d['ms_per_step'], 'kernel', r['kernel'][:24], r['avg_launch_us'], r.get('median_launch_us'), 'frac', r['frac'], 'sustained', r.get('sustained', {}).get('random_tflops'), r.get('sustained', {}).get('frac_of_sustained_random'),
      'strict', d['strict_parity'].get('value'), d['strict_parity'].get('ms_per_step'), 'graph', (d.get('graph_replay') or {}).get('ms_per_step'), (d.get('graph_replay') or {}).get('host_ms_per_step'),
      'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
for P in bf16 fp32; do
  (cd /tmp && DL_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$P -o bench -- python $GRAFT_REPO_ROOT/bench.py --precision $P --steps 3 --warmup 1 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$P.err); echo "rocprof $P rc=$?"
  cp gpurun_out/prof_$P/bench_kernel_stats.csv gpurun_out/bench_train_kernel_stats_${P}_$TAG.csv 2>/dev/null
  rm -rf gpurun_out/prof_$P
  python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/bench_train_kernel_stats_${P}_$TAG.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('$P: total kernel ms per 4 steps', round(tot / 1e6, 1))
for r in rows[:14]:
    print('%-86s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
done
timeout 600 python tools/layer_budget.py $TAG bf16 2>&1 | tail -26
for w in train18 ext infer wsi; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-timer-check --no-graph > gpurun_out/bench_${w}_$TAG.json 2> gpurun_out/bench_${w}_$TAG.err
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/bench_${w}_$TAG.json').read().strip().splitlines()[-1])
    print('$w', d['value'], d['unit'], d['ms_per_step'], 'ms/step')
except Exception as e:
    print('$w failed', e); print(open('gpurun_out/bench_${w}_$TAG.err').read()[-600:])
PY
done
