#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/stempmc
mkdir -p "$OUT"
CMD="python bench.py --workload resnet18 --rows 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end"
i=0
while read -r grp; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp -d "$OUT/g$i" -o b -- $CMD > "$OUT/g$i.log" 2>&1
  echo "group $i rc=$?"
done <<'GRPS'
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL
GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_LDS_ADDR_CONFLICT
GRPS
python tools/pmc_table.py $(find "$OUT" -name "*.db") 2>&1 | grep -A26 "conv2d_patch_kernel\|conv2d_ws_kernel<2, 2, 8>" | head -80
