#!/bin/bash
which=$1; tag=$2
mkdir -p gpurun_out/pmc_$tag
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/q1 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/q2 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/q3 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob
agg=collections.defaultdict(list)
for p in sorted(glob.glob('gpurun_out/pmc_$tag/q*/p_counter_collection.csv')):
    for r in csv.DictReader(open(p)):
        n=r['Kernel_Name']
        if ('conv_gemm' in n or 'wgrad' in n) and 'reduce' not in n:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(f'{k:34s} n={len(v)} mean={sum(v)/len(v):.4g}')
PY
rm -rf gpurun_out/pmc_$tag/q*/p_kernel_trace.csv
