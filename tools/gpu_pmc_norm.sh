#!/bin/bash
# counters for norm_bwd_apply / norm_partial<1> / norm_apply / axpby on the same tensors: is the wave time memory wait or VALU issue?
tag=${1:-norm}
mkdir -p gpurun_out/pmc_$tag
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/p1 -o p -- python $GRAFT_REPO_ROOT/tools/norm_only.py > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/p2 -o p -- python $GRAFT_REPO_ROOT/tools/norm_only.py > /dev/null 2>&1
# (a third pass with FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum in one --pmc list HUNG for 900 s in round 1: removed.
#  Collect FETCH_SIZE and WRITE_SIZE in separate passes as tools/gpu_pmc.sh does, each under its own timeout.)
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob
agg=collections.defaultdict(lambda: collections.defaultdict(list))
def short(n):
    for k in ('norm_bwd_apply','norm_apply','norm_partial','axpby','norm_chunk_sum','norm_bias_final','norm_fwd_finalize','norm_bwd_finalize'):
        if k in n: return k + ('<1>' if k=='norm_partial' and ', 1, ' in n else ('<0>' if k=='norm_partial' else ''))
    return None
for p in sorted(glob.glob('gpurun_out/pmc_$tag/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(p)):
        k=short(r['Kernel_Name'])
        if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
dur=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmc_$tag/p1/p_kernel_trace.csv')):
    k=short(r['Kernel_Name'])
    if k: dur[k].append((float(r['End_Timestamp'])-float(r['Start_Timestamp']))/1e3)
for k in ('axpby','norm_apply','norm_partial<1>','norm_bwd_apply'):
    c={n:sum(v)/len(v) for n,v in agg[k].items()}
    print(f"== {k}: {sum(dur[k])/max(len(dur[k]),1):.1f} us (profiled)")
    for n in sorted(c): print(f"   {n:24s} {c[n]:.4g}")
    if 'SQ_WAVE_CYCLES' in c:
        w=c['SQ_WAVE_CYCLES']
        print(f"   -> of wave cycles: waiting(any) {c.get('SQ_WAIT_ANY',0)/w:.2f}  wait_inst(any) {c.get('SQ_WAIT_INST_ANY',0)/w:.2f}  active_inst(any) {c.get('SQ_ACTIVE_INST_ANY',0)/w:.2f}  valu_active {c.get('SQ_ACTIVE_INST_VALU',0)/w:.2f}  vmem_active {c.get('SQ_ACTIVE_INST_VMEM',0)/w:.2f}")
    if 'FETCH_SIZE' in c: print(f"   -> fetch {2*c['FETCH_SIZE']*1024/1e6:.0f} MB (x2 corrected)  write {c.get('WRITE_SIZE',0)*1024/1e6:.0f} MB  L2 hit {c.get('TCC_HIT_sum',0)/max(c.get('TCC_HIT_sum',0)+c.get('TCC_MISS_sum',0),1):.2f}")
PY
rm -rf gpurun_out/pmc_$tag/p*/p_kernel_trace.csv
