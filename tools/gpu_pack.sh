#!/bin/bash
# batched weight repacking: bit-exactness test, then rocprof of the training step with the hook ON (per-kernel times are what counts)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -x -k "pack_weights_batch" 2>&1 | tail -2
DL_PACK_BATCH=1 timeout 600 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=600 -x -k "step" 2>&1 | tail -2
export DL_PACK_BATCH=1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_packb -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_packb.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/prof_packb/*kernel_trace.csv
grep -i "pack" gpurun_out/prof_packb/bench_kernel_stats.csv | cut -d, -f1-4
