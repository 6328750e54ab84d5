#!/bin/bash
# two experiments on one box: (1) padded slab stride of the weight gradient (DL_WGRAD_SLAB_PAD, default 1088 floats; 0 = unpadded), (2) branch streams
# (DL_STREAMS=N).  Bit-identity tests first, then the bf16 step in one process per setting.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_deferred.py -m gpu -q -x > gpurun_out/streams_tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/streams_tests.log | tail -8
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad" 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-baseline-n8 --no-graph --no-timer-check --no-strict"
run() {
    tag=$1; shift
    env "$@" timeout 600 $B 2>gpurun_out/bench_streams_$tag.err | tail -1 > gpurun_out/bench_streams_$tag.json
    python - "$tag" <<'P'
import json, sys
d = json.loads(open(f'gpurun_out/bench_streams_{sys.argv[1]}.json').read())
r = d['roofline']
print(sys.argv[1], d['value'], d['ms_per_step'], 'streams', d['config'].get('streams'), 'dominant us', r['avg_launch_us'], r.get('median_launch_us'))
P
}
run pad0 DL_WGRAD_SLAB_PAD=0
run pad1088 X=1
run pad4160 DL_WGRAD_SLAB_PAD=4160
run s2 DL_STREAMS=2
run s3 DL_STREAMS=3
run s5 DL_STREAMS=5
run pad0b DL_WGRAD_SLAB_PAD=0
run base2 X=1
