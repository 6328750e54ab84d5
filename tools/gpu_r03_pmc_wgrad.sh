#!/bin/bash
# HBM-side traffic of the ResnetBlock weight-gradient launches (bf16 wgrad_glds_kernel<256>, strict wgrad_glds_x3_kernel<256>) from separate --pmc passes
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for prec in bf16 fp32; do
  cd /tmp
  timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcw_$prec/p1 -o p -- python $R/tools/conv_time.py $prec wgrad > /dev/null 2>&1
  timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/pmcw_$prec/p2 -o p -- python $R/tools/conv_time.py $prec wgrad > /dev/null 2>&1
  timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmcw_$prec/p3 -o p -- python $R/tools/conv_time.py $prec wgrad > /dev/null 2>&1
  cd $R
  python - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob('gpurun_out/pmcw_$prec/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(p)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        if 'wgrad' in n:
            agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for n, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    o = {'launches': len(next(iter(c.values()))), 'fetch_bytes_raw': m.get('FETCH_SIZE', 0) * 1024, 'write_bytes': m.get('WRITE_SIZE', 0) * 1024}
    o['traffic_bytes'] = 2 * o['fetch_bytes_raw'] + o['write_bytes']
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m:
        o['mfma_util'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] * 128)
    out[n] = o
el = 2 if '$prec' == 'bf16' else 4
out['algorithmic_bytes'] = {'operands': 2 * 8 * 128 * 128 * 256 * el, 'gradient': 256 * 2304 * 4, 'split_k_slabs_written_then_read': 'see WRITE_SIZE of the wgrad kernel / FETCH_SIZE of wgrad_reduce_kernel'}
json.dump(out, open('gpurun_out/pmc_wgrad_$prec.json', 'w'), indent=1)
print('$prec', json.dumps(out))
PY
  rm -rf gpurun_out/pmcw_$prec
done
