#!/bin/bash
# Round-4 PMC evidence (separate --pmc passes, no other trace domains): the dominant kernel (conv_gemm_w4_kernel, forward launches) -> the file
# bench.py reads for roofline.traffic; HBM-side bytes of the fused stride-2 tile on up2 (FETCH <= 1.5 x input) next to the 4-phase path.
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_pmc.sh fwd r04 > gpurun_out/pmc_r04_log.txt 2>&1
python tools/pmc_summarize.py r04 gpurun_out/pmc_dominant_conv256_r04.json
tail -3 gpurun_out/pmc_r04_log.txt
rm -rf gpurun_out/pmc_r04
bash tools/gpu_pmc_any.sh up2_s2f conv_s2f python $GRAFT_REPO_ROOT/tools/conv_only.py up2 2>&1 | tail -2
DL_CONV_S2F=0 bash tools/gpu_pmc_any.sh up2_4phase conv_gemm_glds python $GRAFT_REPO_ROOT/tools/conv_only.py up2 2>&1 | tail -2
# WRITE_SIZE of the fused tile in its own pass
cd /tmp
timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_up2w/p4 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py up2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
v=[float(r['Counter_Value']) for p in glob.glob('gpurun_out/pmc_up2w/p4/p_counter_collection.csv') for r in csv.DictReader(open(p)) if 'conv_s2f' in r['Kernel_Name'] and r['Counter_Name']=='WRITE_SIZE']
print('up2 s2f WRITE_SIZE KB mean', sum(v)/max(len(v),1))
open('gpurun_out/pmc_up2_s2f_write.txt','w').write('WRITE_SIZE_KB_mean %f over %d launches\n' % (sum(v)/max(len(v),1), len(v)))
PY
rm -rf gpurun_out/pmc_up2w
