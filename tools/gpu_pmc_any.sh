#!/bin/bash
# PMC passes over an arbitrary command: tools/gpu_pmc_any.sh <tag> <kernel-name-substring> <cmd...>   (counters only, no other trace domains)
tag=$1; pat=$2; shift 2
mkdir -p gpurun_out/pmc_$tag
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $R/gpurun_out/pmc_$tag/p1 -o p -- "$@" > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc_$tag/p2 -o p -- "$@" > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_$tag/p3 -o p -- "$@" > /dev/null 2>&1
cd $R
python - <<PY
import csv, collections, glob, json
agg=collections.defaultdict(list)
for p in sorted(glob.glob('gpurun_out/pmc_$tag/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(p)):
        if '$pat' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
c = {k: sum(v)/len(v) for k, v in agg.items()}
dur = []
for p in sorted(glob.glob('gpurun_out/pmc_$tag/p3/p_kernel_trace.csv')):
    dur = [(float(r['End_Timestamp'])-float(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(p)) if '$pat' in r['Kernel_Name']]
d = sum(dur)/max(len(dur),1)
out = {'tag': '$tag', 'kernel_substring': '$pat', 'launches': len(dur), 'mean_us_in_profiled_pass': round(d, 1), 'counters_mean_per_launch': c}
if 'GRBM_GUI_ACTIVE' in c and d:
    out['effective_clock_ghz'] = round(c['GRBM_GUI_ACTIVE'] / 8 / (d * 1e3), 3)
if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
    out['mfma_busy_frac'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] * 128), 4)
if 'SQ_WAVE_CYCLES' in c:
    for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
        if k in c: out[k + '_frac_of_wave_cycles'] = round(c[k] / c['SQ_WAVE_CYCLES'], 4)
json.dump(out, open('gpurun_out/pmc_$tag.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'counters_mean_per_launch'}))
print({k: float('%.4g' % v) for k, v in c.items()})
PY
rm -rf gpurun_out/pmc_$tag
