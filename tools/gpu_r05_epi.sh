#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== conv kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_switches.py tests/test_gpu_fullsize.py -q -k "conv or w4 or big or full_size_forward" 2>&1 | tail -4
echo "== conv fwd/dgrad with the bias through LDS"
for d in randn halfzero; do TIME_DATA=$d timeout 200 python tools/conv_time.py bf16 fwd,fwdstats,dgrad 2>&1 | tail -1; done
echo "== host profile"; timeout 300 python tools/host_profile.py 5 2>&1 | head -60
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-baseline-n8 --no-graph --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('one_stream_ms_per_step'), 'strict', d['strict_parity']['value'])"
} > gpurun_out/r05_epi.txt 2>&1
cat gpurun_out/r05_epi.txt
