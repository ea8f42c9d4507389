#!/bin/bash
# PMC passes over the ResNet bench (runs on the GPU box via gpurun); one counter group per pass.
set -u
ROWS=${1:-512}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/convpmc
mkdir -p "$OUT"
CMD="python bench.py --workload resnet18 --rows $ROWS --steps 1 --warmup 1 --no-cpu-baseline"
i=0
while read -r grp; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp -d "$OUT/g$i" -o b -- $CMD > "$OUT/g$i.log" 2>&1
  echo "group $i rc=$?"
done <<'GRPS'
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
GRPS
find "$OUT" -name "*.db"
