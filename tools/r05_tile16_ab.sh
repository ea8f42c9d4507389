#!/bin/bash
# Round 5: the per-chunk fused-MLP kernel as 16-row tiles on 16x16x4 MFMAs (mlp3_tile16_kernel) vs 32-row tiles (mlp3_tile_kernel): parity tests,
# kernel durations from a trace, C2 scan rates staged and registered by caller count.  INFERA_MLP_TILE16_MAX_ROWS=0 keeps the 32-row kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_tile16; mkdir -p $OUT; rm -f $OUT/ab.txt
( timeout 900 python -m pytest tests/test_mlp_tile16_gpu.py tests/test_parity_gpu.py tests/test_mlp_jit_tile_gpu.py -q -x 2>&1 | tail -4 ) > $OUT/pytest.txt
for t16 in 0 4096; do
  ( cd /tmp && rm -rf /tmp/p_t16 && INFERA_MLP_TILE16_MAX_ROWS=$t16 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_t16 -o t -- python $OLDPWD/tools/host_scan_bench.py --rows 1000000 --threads 1 --reps 2 --numa auto > /dev/null 2>&1 )
  echo "=== kernel trace, INFERA_MLP_TILE16_MAX_ROWS=$t16 (one staged caller)" >> $OUT/ab.txt
  python tools/rocpd_summary.py $(find /tmp/p_t16 -name "*.db") 2>/dev/null | grep -E "mlp3_tile" | cut -c1-40,100-200 >> $OUT/ab.txt
done
for rep in 1 2; do
for t16 in 0 4096; do
  for flag in "" "--register"; do
    echo "=== round $rep INFERA_MLP_TILE16_MAX_ROWS=$t16 ${flag:-staged}" >> $OUT/ab.txt
    INFERA_MLP_TILE16_MAX_ROWS=$t16 python tools/host_scan_bench.py --rows 6000000 --threads 1,2,4,8,16 --reps 3 --numa auto $flag 2>&1 | grep -A1 "threads=" >> $OUT/ab.txt
  done
done
done
cat $OUT/pytest.txt; grep -E "===|mlp3_tile" $OUT/ab.txt | head -8
awk '/=== round/{h=$2" "$3" "$4" "$5} /threads=/{t=$2; r=$3; getline l; match(l,/cpu_us_per_chunk.: [0-9.]+/); c=substr(l,RSTART+19,RLENGTH-19); printf "%-55s callers %2s  %7s M rows/s  %5s us CPU/chunk\n", h, t, r, c}' $OUT/ab.txt
