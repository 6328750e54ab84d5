#!/bin/bash
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters_list.txt 2>&1
for which in fwd wgrad; do
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc/${which}_p1 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > $GRAFT_REPO_ROOT/gpurun_out/pmc/${which}_p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $GRAFT_REPO_ROOT/gpurun_out/pmc/${which}_p2 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > $GRAFT_REPO_ROOT/gpurun_out/pmc/${which}_p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc/${which}_p3 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > $GRAFT_REPO_ROOT/gpurun_out/pmc/${which}_p3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc/${which}_p4 -o p -- python $GRAFT_REPO_ROOT/tools/conv_only.py $which > $GRAFT_REPO_ROOT/gpurun_out/pmc/${which}_p4.log 2>&1
done
cd $GRAFT_REPO_ROOT; find gpurun_out/pmc -name "*.csv" | head -30; du -sh gpurun_out/pmc
