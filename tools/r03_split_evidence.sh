#!/bin/bash
# Round-3 evidence for the split-fp16 convolution mode (GPU box): kernel trace + SQ / GRBM counter passes of the ResNet-18 bench with
# INFERA_PRECISION=f16x3, the bench line with the end-to-end leg, and the last pass's kernels in order.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export INFERA_PRECISION=f16x3
O=gpurun_out/r03_split
mkdir -p $O
ARGS="--workload resnet18 --precision f16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-host-probe"
rocprofv3 --kernel-trace --stats -d $O/trace -o b -- python bench.py $ARGS > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o b -- python bench.py $ARGS > $O/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 -d $O/pmc_grbm -o b -- python bench.py $ARGS > $O/pmc_grbm.log 2>&1
python tools/rocpd_summary.py $(find $O/trace -name "*.db") > gpurun_out/r03_split_resnet18_bench.txt 2>&1
python tools/trace_last_step.py $(find $O/trace -name "*.db") 2>&1 | grep -v copyBuffer | tail -27 > gpurun_out/r03_split_resnet18_last_pass.txt
python tools/pmc_table.py $(find $O/pmc_sq $O/pmc_grbm -name "*.db") > gpurun_out/r03_split_resnet18_pmc_table.txt 2>&1
python bench.py --workload resnet18 --precision f16x3 --steps 5 --warmup 2 --no-host-probe --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r03_split_line_resnet18.json
head -12 gpurun_out/r03_split_resnet18_bench.txt | cut -c1-60,100-160
grep -A1 "^conv2d" gpurun_out/r03_split_resnet18_pmc_table.txt | grep calls
python - <<PY
import json
d=json.load(open("gpurun_out/r03_split_line_resnet18.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["end_to_end"]["rows_per_s"], d["end_to_end"]["threads"], d["end_to_end"].get("vs_cpu_baseline"))
PY
