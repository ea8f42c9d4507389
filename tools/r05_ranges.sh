#!/bin/bash
# Round 5: where a chunk's wall time goes, per caller count, staged and registered -- roctx ranges (INFERA_PROFILE=1) + kernel + copy records.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r05_ranges
mkdir -p $OUT
for mode in staged registered; do
  for th in 1 2 4; do
    flag=""; [ $mode = registered ] && flag="--register"
    ( cd /tmp && rm -rf /tmp/p_rng && INFERA_PROFILE=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -d /tmp/p_rng -o t -- \
        python $OLDPWD/tools/host_scan_bench.py --rows 1500000 --threads $th --reps 2 --numa auto $flag > $OLDPWD/$OUT/${mode}_$th.log 2>&1 )
    python tools/e2e_ranges.py $(find /tmp/p_rng -name "*.db" | head -1) "$mode, $th caller(s)" > $OUT/${mode}_$th.txt 2>&1
    grep "threads=" $OUT/${mode}_$th.log >> $OUT/${mode}_$th.txt
    grep "infera profile" $OUT/${mode}_$th.log >> $OUT/${mode}_$th.txt
  done
done
cat $OUT/*.txt > $OUT/all.txt
# the same scans without the profiler, for the rates
for flag in; do
  python tools/host_scan_bench.py --rows 4000000 --threads 1,2,4,8 --reps 3 --numa auto $flag 2>&1 | grep -v "^devices" >> $OUT/rates.txt
done
