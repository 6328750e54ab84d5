#!/bin/bash
# residual gradient sum fused with the norm-backward reductions (DL_NO_SUM2_PARTIAL=1 = dl_axpby + stand-alone pass): parity + A/B of both policies
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "residual_sum_fused or norm_forward_backward" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_networks.py -m gpu -x -q -k "golden_fixture or split_copies or teacher_forced_layer or benched or against_oracle" 2>&1 | tail -2
for v in 1 0 1 0; do
  DL_NO_SUM2_PARTIAL=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-baseline-n8 --no-timer-check 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('DL_NO_SUM2_PARTIAL=$v bf16', d['value'], d['ms_per_step'], 'strict', d['strict_parity']['value'], d['strict_parity']['ms_per_step'])"
done
