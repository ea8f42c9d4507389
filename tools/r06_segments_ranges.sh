#!/bin/bash
# (round 6) per-chunk timeline of the registered scan over DuckDB-shaped segments (pulling kernel for every chunk) at 1 / 4 callers, beside the
# contiguous registered table: roctx ranges (INFERA_PROFILE=1) + kernel + copy records, then the same scans unprofiled with their phase clocks.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r06_segments_ranges}
mkdir -p $OUT
for mode in segments rect; do
  for th in 1 4; do
    ( cd /tmp && rm -rf /tmp/p_rng && INFERA_PROFILE=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -d /tmp/p_rng -o t -- \
        python $OLDPWD/tools/r06_segments_scan.py $mode 1500000 $th 2 > $OLDPWD/$OUT/${mode}_$th.log 2>&1 )
    python tools/e2e_ranges.py $(find /tmp/p_rng -name "*.db" | head -1) "$mode, $th caller(s)" > $OUT/${mode}_$th.txt 2>&1
    grep -E "threads=|phases" $OUT/${mode}_$th.log >> $OUT/${mode}_$th.txt
  done
done
cat $OUT/*.txt > $OUT/all.txt
for mode in segments rect staged; do
  for th in 1 2 4 8; do python tools/r06_segments_scan.py $mode 6000000 $th 3 2>&1 | grep -E "threads=|phases" >> $OUT/rates.txt; done
done
cat $OUT/rates.txt
