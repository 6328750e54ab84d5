#!/bin/bash
# every GPU test that reaches the inference DAG, with DL_INFER_STREAMS=3 (opt-in this round): evidence for switching it on by default
export TMPDIR=/tmp
mkdir -p gpurun_out
DL_INFER_STREAMS=3 timeout 330 python -m pytest tests/test_gpu_seam.py tests/test_gpu_tiles.py tests/test_gpu_infer_streams.py tests/test_gpu_networks.py tests/test_gpu_zoo.py -m gpu -q --timeout=300 \
  -k "seam or tiles or infer_streams or inference or dag or thread or kd or KD" > gpurun_out/infer_streams_suite.log 2>&1; echo "rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/infer_streams_suite.log | tail -12
