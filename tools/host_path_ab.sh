#!/bin/bash
# A/B of the host-path knobs on the C2 scan (10M rows, materialised columnar host table): how a call waits
# (INFERA_HOST_WAIT), result stores straight into pinned memory (INFERA_HOST_DIRECT_OUT), column-major chunk read by
# the fused kernel itself (INFERA_HOST_FUSED_TRANSPOSE).  Runs on the GPU box; output -> gpurun_out/<tag>/host_path_ab.txt
TAG=${1:-r02_ab}
THREADS=${2:-8,16,24,32,48}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
OUT=gpurun_out/$TAG/host_path_ab.txt
: > $OUT
for cfg in "spin 0 0" "block 0 0" "spin 1 0" "spin 1 1" "block 1 1"; do
  set -- $cfg
  echo "## INFERA_HOST_WAIT=$1 INFERA_HOST_DIRECT_OUT=$2 INFERA_HOST_FUSED_TRANSPOSE=$3" >> $OUT
  INFERA_HOST_WAIT=$1 INFERA_HOST_DIRECT_OUT=$2 INFERA_HOST_FUSED_TRANSPOSE=$3 python tools/host_scan_bench.py --rows 10000000 --threads $THREADS --reps 3 2>&1 | grep threads= >> $OUT
done
cat $OUT
