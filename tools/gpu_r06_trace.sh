#!/bin/bash
# per-dispatch view of the one-stream training step: rocprofv3 --kernel-trace, dispatches grouped by (kernel, grid, workgroup) -> launches, mean / min duration.
# WL=<workload> selects another bench workload (default train).  Tells apart the SHAPES a kernel template runs on (the --stats table averages over them).  tools/gpu_r06_trace.sh <tag> [substring ...]
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-a}; shift
rm -rf gpurun_out/prof_tr
(cd /tmp && DL_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tr -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload ${WL:-train} --steps 2 --warmup 1 --no-cpu-baseline --no-cpu-baseline-n8 --no-strict --no-graph --no-timer-check --no-other-workloads > /dev/null 2>&1)
python - "$@" > gpurun_out/trace_$TAG.txt <<'PY'
import csv, sys, collections
subs = sys.argv[1:] or ['norm_']
rows = list(csv.DictReader(open('gpurun_out/prof_tr/bench_kernel_trace.csv')))
g = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name']
    if not any(s in name for s in subs):
        continue
    key = (name[:64], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')))
    g[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0.0
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print('%-64s grid %8s x %-3s wg %-5s  %5d launches  mean %8.1f us  min %8.1f us  total %8.2f ms' % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v), sum(v) / 1e3))
print('total of the listed kernels over 3 steps: %.2f ms' % (tot / 1e3))
PY
rm -rf gpurun_out/prof_tr
cat gpurun_out/trace_$TAG.txt
