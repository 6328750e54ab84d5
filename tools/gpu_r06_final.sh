#!/bin/bash
# Round-6 end-of-round verification on ONE box: every GPU test, smoke, the contract line exactly as the driver runs it (cpu_baseline legs and other_workloads
# included), rocprofv3 --kernel-trace --stats of the bf16 and of the strict step (ONE stream, DL_STREAMS=1: per-kernel durations that mean the kernel, not its
# share of the GPU next to another branch), the layer budget, the on-spec batch-8 CPU baseline, the host-load probe.  Every command has its own timeout.
TAG=${1:-r06}
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
timeout 2700 python -m pytest tests -m gpu -q --timeout=1200 > gpurun_out/gpu_tests_$TAG.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/gpu_tests_$TAG.log | tail -12
cp gpurun_out/parity_errors.json gpurun_out/parity_errors_main_$TAG.json 2>/dev/null
cp gpurun_out/parity_errors_fullsize.json gpurun_out/parity_errors_fullsize_$TAG.json 2>/dev/null
cp gpurun_out/parity_errors_fullsize_step.json gpurun_out/parity_errors_fullsize_step_$TAG.json 2>/dev/null
cp gpurun_out/trajectory.json gpurun_out/trajectory_$TAG.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_contract_$TAG.json 2> gpurun_out/bench_contract_$TAG.err; echo "contract bench rc=$?"
python - <<PY
import json
d = json.loads(open('gpurun_out/bench_contract_$TAG.json').read().strip().splitlines()[-1])
r = d['roofline']
print('bench', d['value'], d['ms_per_step'], 'kernel', r['kernel'][:24], r['avg_launch_us'], r.get('median_launch_us'), 'frac', r['frac'], 'one-stream', r.get('one_stream_ms_per_step'), 'sustained', r.get('sustained', {}).get('random_tflops'),
      r.get('sustained', {}).get('frac_of_sustained_random'), 'strict', d['strict_parity'].get('value'), d['strict_parity'].get('ms_per_step'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
for k, v in (d.get('other_workloads') or {}).items():
    print(' ', k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'streams', 'wall_s', 'error')}, 'frac', (v.get('roofline') or {}).get('frac'), 'whole', (v.get('whole_slide') or {}).get('tiles_per_s'), 'cpu', (v.get('cpu_baseline') or {}).get('value'))
PY
for P in bf16 fp32; do
  (cd /tmp && DL_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$P -o bench -- python $GRAFT_REPO_ROOT/bench.py --precision $P --steps 3 --warmup 1 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check --no-other-workloads > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$P.err); echo "rocprof $P rc=$?"
  cp gpurun_out/prof_$P/bench_kernel_stats.csv gpurun_out/bench_train_kernel_stats_${P}_$TAG.csv 2>/dev/null
  rm -rf gpurun_out/prof_$P
  python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/bench_train_kernel_stats_${P}_$TAG.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('$P: total kernel ms per 4 steps', round(tot / 1e6, 1))
for r in rows[:16]:
    print('%-86s %6s calls %9.1f us avg %6.2f %%' % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
done
timeout 600 python tools/layer_budget.py $TAG bf16 2>&1 | tail -26
if [ "$3" != "nocpu" ]; then
timeout 1500 python bench.py --steps 3 --warmup 1 --no-strict --no-graph --no-timer-check --no-other-workloads --cpu-baseline-n8-full > gpurun_out/bench_n8full_$TAG.json 2>/dev/null; cat gpurun_out/cpu_baseline_n8_full.json
timeout 500 python tools/dp_host_load.py 2>&1 | tail -3
fi
