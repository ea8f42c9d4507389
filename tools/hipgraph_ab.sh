#!/bin/bash
# hipGraph replay per 2048-row chunk against direct stream enqueues, with CPU time per chunk as the yardstick (profiles/r04_hipgraph_ab.txt):
# C2 scan, 4 / 8 / 16 callers x 1 / 2 / 4 device slots on one GPU.   usage: gpurun -- bash tools/hipgraph_ab.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/hipgraph_ab.txt
: > $O
for hg in 0 1; do
  for slots in 1 2 4; do
    dev=$(python -c "print(','.join(['0']*$slots))")
    echo "=== INFERA_HIPGRAPH=$hg slots=$slots" >> $O
    INFERA_HIPGRAPH=$hg INFERA_DEVICES=$dev python tools/host_scan_bench.py --rows 6000000 --threads 4,8,16 --reps 3 --numa auto 2>&1 | grep -v "^devices=" >> $O
  done
done
