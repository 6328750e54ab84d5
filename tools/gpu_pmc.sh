#!/bin/bash
# PMC passes for conv_only.py <which>; out dir gpurun_out/pmc_<tag>
which=$1; tag=$2
mkdir -p gpurun_out/pmc_$tag
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/p1 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/p2 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/p3 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/p4 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob
agg=collections.defaultdict(list)
for p in sorted(glob.glob('gpurun_out/pmc_$tag/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(p)):
        n=r['Kernel_Name']
        if ('conv_' in n or 'wgrad' in n) and 'reduce' not in n and 'pack' not in n:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(f'{k:28s} n={len(v)} mean={sum(v)/len(v):.4g}')
for p in sorted(glob.glob('gpurun_out/pmc_$tag/p1/p_kernel_trace.csv')):
    d=[(float(r['End_Timestamp'])-float(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(p)) if ('conv_' in r['Kernel_Name'] or 'wgrad' in r['Kernel_Name']) and 'reduce' not in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
    print('durations us', [round(x,1) for x in d])
PY
rm -rf gpurun_out/pmc_$tag/p*/p_kernel_trace.csv
