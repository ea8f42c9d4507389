#!/bin/bash
# SQ counters of the split-fp16 convolution kernels over the ResNet-18 bench (GPU box); INFERA_SPLIT_PROBE / INFERA_LIB_PATH pass through.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-split_pmc}
ARGS="--workload resnet18 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-host-probe"
INFERA_PRECISION=f16x3 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    -d gpurun_out/$TAG/sq -o b -- python bench.py $ARGS > gpurun_out/$TAG.sq.log 2>&1
INFERA_PRECISION=f16x3 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 -d gpurun_out/$TAG/grbm -o b -- python bench.py $ARGS > gpurun_out/$TAG.grbm.log 2>&1
python tools/pmc_table.py $(find gpurun_out/$TAG -name "*.db") 2>&1 | grep -A16 "conv2d_split" | cut -c1-120
