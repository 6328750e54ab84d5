#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
T="tests/test_gpu_zoo.py::test_deepliifkd_step_golden_fixture_from_reference"
for v in "" "DL_INFER_STREAMS=1" "DL_WGRAD_TR_ASM=0" "DL_WGRAD_ARENA_MB=4096" "DL_WGRAD_DEFER=0"; do
  echo "== $v"; env $v timeout 300 python -m pytest "$T" -q -x 2>&1 | grep -E "passed|failed|digest|Error" | tail -4
done
echo "== wgrad batch + graph tests"; timeout 600 python -m pytest tests/test_gpu_wgrad_batch.py tests/test_gpu_graph.py tests/test_gpu_streams.py -q 2>&1 | tail -6
echo "== conv fwd/dgrad asymmetry"
for d in randn halfzero; do TIME_DATA=$d timeout 200 python tools/conv_time.py bf16 fwd,fwdstats,dgrad 2>&1 | tail -1; done
echo "== serialize margin"; timeout 600 python tools/serialize_margin.py 2>&1 | tail -9
} > gpurun_out/r05_kd.txt 2>&1
cat gpurun_out/r05_kd.txt
