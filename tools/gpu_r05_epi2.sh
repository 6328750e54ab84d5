#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== conv kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_switches.py tests/test_gpu_fullsize.py -q -k "conv or w4 or big or full_size_forward" 2>&1 | tail -3
echo "== conv fwd/dgrad, templated epilogue"
for d in randn halfzero; do TIME_DATA=$d timeout 200 python tools/conv_time.py bf16 fwd,fwdstats,dgrad 2>&1 | tail -1; done
echo "== host profile"; timeout 300 python tools/host_profile.py 5 2>&1 | head -4
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-baseline-n8 --no-graph --no-other-workloads --no-strict 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('one_stream_ms_per_step'))"
echo "== PMC wgrad_w4 (batched, 18 layers per launch)"
bash tools/gpu_pmc_any.sh wgrad_w4_r05 wgrad_w4_kernel python $GRAFT_REPO_ROOT/tools/wgrad_time.py bf16 18 2>&1 | tail -2
(cd /tmp && timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ww/p4 -o p -- python $GRAFT_REPO_ROOT/tools/wgrad_time.py bf16 18 > /dev/null 2>&1)
python - <<PY
import csv, glob
for pat in ('wgrad_w4_kernel', 'wgrad_reduce_batch_kernel'):
    v = [float(r['Counter_Value']) for p in glob.glob('gpurun_out/pmc_ww/p4/p_counter_collection.csv') for r in csv.DictReader(open(p)) if pat in r['Kernel_Name'] and r['Counter_Name'] == 'WRITE_SIZE']
    print(pat, 'WRITE_SIZE KB mean per launch', sum(v) / max(len(v), 1), 'launches', len(v))
    open('gpurun_out/pmc_wgrad_w4_write_r05.txt', 'a').write('%s WRITE_SIZE_KB_mean %f over %d launches\n' % (pat, sum(v) / max(len(v), 1), len(v)))
PY
rm -rf gpurun_out/pmc_ww
} > gpurun_out/r05_epi2.txt 2>&1
cat gpurun_out/r05_epi2.txt
