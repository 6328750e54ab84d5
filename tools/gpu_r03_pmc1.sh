#!/bin/bash
DL_X3_VAR=0 bash tools/gpu_pmc_any.sh x3_fwd_v0 conv_gemm_8ph_x3 python $GRAFT_REPO_ROOT/tools/conv_time.py fp32 fwd
DL_X3_VAR=2 bash tools/gpu_pmc_any.sh x3_fwd_v2 conv_gemm_8ph_x3 python $GRAFT_REPO_ROOT/tools/conv_time.py fp32 fwd
DL_X3_VAR=4 bash tools/gpu_pmc_any.sh x3_fwd_v4 conv_gemm_8ph_x3 python $GRAFT_REPO_ROOT/tools/conv_time.py fp32 fwd
bash tools/gpu_pmc_any.sh bf16_fwd conv_gemm_8ph_kernel python $GRAFT_REPO_ROOT/tools/conv_time.py bf16 fwd
DL_WGRAD_X3=1 bash tools/gpu_pmc_any.sh x3_wgrad_1bar wgrad_glds_x3 python $GRAFT_REPO_ROOT/tools/conv_time.py fp32 wgrad
bash tools/gpu_pmc_any.sh x3_wgrad_4ph wgrad_4ph_x3 python $GRAFT_REPO_ROOT/tools/conv_time.py fp32 wgrad
