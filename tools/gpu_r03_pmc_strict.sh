#!/bin/bash
# PMC passes over the strict ResnetBlock conv (conv_gemm_8ph_x3_kernel, forward launches of tools/conv_time.py fp32 fwd): HBM-side traffic + MFMA busy.
# Counters only (no other trace domains), one --pmc set per pass.  Result: gpurun_out/pmc_strict_conv256.json (copied to profiles/r03/, read by bench.py)
mkdir -p gpurun_out/pmc_strict
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_strict/p1 -o p -- python $R/tools/conv_time.py fp32 fwd > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/pmc_strict/p2 -o p -- python $R/tools/conv_time.py fp32 fwd > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_strict/p3 -o p -- python $R/tools/conv_time.py fp32 fwd > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, collections, glob, json
agg, names = collections.defaultdict(list), collections.Counter()
for p in sorted(glob.glob('gpurun_out/pmc_strict/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(p)):
        n = r['Kernel_Name']
        if 'conv_gemm_8ph_x3' in n:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            names[n.split('(')[0].replace('void ', '').strip()] += 1
c = {k: sum(v) / len(v) for k, v in agg.items()}
dur = [(float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open('gpurun_out/pmc_strict/p1/p_kernel_trace.csv')) if 'conv_gemm_8ph_x3' in r['Kernel_Name']]
fetch_raw, write = c['FETCH_SIZE'] * 1024, c['WRITE_SIZE'] * 1024
res = {'kernel': names.most_common(1)[0][0], 'workload': '3x3 256->256 @ 8x128x128, fp32 storage + split-bf16 x3 (tools/conv_time.py fp32 fwd: in-kernel split of the fp32 input)',
       'launches': len(dur), 'mean_us_in_profiled_pass': round(sum(dur) / max(len(dur), 1), 1), 'counters_mean_per_launch': c,
       'fetch_bytes_raw': fetch_raw, 'fetch_bytes_corrected_x2': 2 * fetch_raw, 'write_bytes': write, 'traffic_bytes': 2 * fetch_raw + write,
       'algorithmic_bytes': 2 * 8 * 128 * 128 * 256 * 4 + 2 * 256 * 2304 * 2,
       'note': 'FETCH_SIZE / WRITE_SIZE in KB from separate rocprofv3 --pmc passes; gfx950 reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> x2 on the read side'}
if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
    res['mfma_util'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] * 128)
json.dump(res, open('gpurun_out/pmc_strict_conv256.json', 'w'), indent=1)
print({k: v for k, v in res.items() if k != 'counters_mean_per_launch'})
PY
rm -rf gpurun_out/pmc_strict
