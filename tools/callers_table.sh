#!/bin/bash
# The rate-per-GPU-by-callers table of DESIGN.md 6.1 / INTEGRATION.md 2.4 on the current tree: C2 scan, staged and registered, two rounds.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/callers_table.txt}; : > $OUT
for rep in 1 2; do for flag in "" "--register"; do
  echo "=== round $rep ${flag:-staged}" >> $OUT
  python tools/host_scan_bench.py --rows 8000000 --threads 1,2,3,4,6,8,12,16 --reps 3 --numa auto $flag 2>&1 | grep -A1 "threads=" >> $OUT
done; done
awk '/=== round/{h=$2" "$3" "$4} /threads=/{t=$2; r=$3; getline l; match(l,/cpu_us_per_chunk.: [0-9.]+/); c=substr(l,RSTART+19,RLENGTH-19); printf "%-24s callers %2s  %7s M rows/s  %5s us CPU/chunk\n", h, t, r, c}' $OUT
