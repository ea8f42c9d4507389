#!/bin/bash
# PMC passes over the ResNet bench (runs on the GPU box via gpurun); one counter group per pass.
# (TA_*/TCP_*/TD_* stall counters hang rocprofv3 on this pool -- not used.)
set -u
ROWS=${1:-512}
TAG=${2:-convpmc}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
CMD="python bench.py --workload resnet18 --rows $ROWS --steps 1 --warmup 1 --no-cpu-baseline"
i=0
while read -r grp; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp -d "$OUT/g$i" -o b -- $CMD > "$OUT/g$i.log" 2>&1
  echo "group $i rc=$?"
done <<'GRPS'
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
GRPS
find "$OUT" -name "*.db"
