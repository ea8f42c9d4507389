#!/bin/bash
# (round 6) C4's result stores: per-row 8-byte stores (mode 0) vs a tile's 32 x 10 results parked in LDS and written as 80 consecutive 16-byte
# pieces (mode 1), INFERA_DENSE16S_MODE, each with plain and with non-temporal table loads + result stores (INFERA_DENSE16S_NT), beside tools/ubench/hbm_stream on the same box.
# usage: tools/r06_c4_store_ab.sh [out]     (run through gpurun; two interleaved rounds)
OUT=${1:-gpurun_out/r06_c4_store_ab.txt}
mkdir -p "$(dirname "$OUT")"
{
  echo "== tools/ubench/hbm_stream 24 GiB =="
  [ -x tools/ubench/hbm_stream ] || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/hbm_stream tools/ubench/hbm_stream.hip
  tools/ubench/hbm_stream 24
  echo "== parity: C4 tests under each mode =="
  for m in "0 1" "1 0" "1 1"; do
    set -- $m
    echo "-- INFERA_DENSE16S_MODE=$1 INFERA_DENSE16S_NT=$2"
    INFERA_DENSE16S_MODE=$1 INFERA_DENSE16S_NT=$2 python -m pytest tests/test_parity_gpu.py -q -x -k "logreg or c4 or generic_dense or golden" 2>&1 | tail -2
  done
  echo "== bench.py --workload logreg (50M rows resident, 20 steps), two rounds =="
  for round in 1 2 3; do
    for m in "0 0" "1 0" "0 1" "1 1"; do
      set -- $m
      line=$(INFERA_DENSE16S_MODE=$1 INFERA_DENSE16S_NT=$2 python bench.py --workload logreg --no-cpu-baseline --no-end-to-end --steps 20 --warmup 3 2>/dev/null | tail -1)
      echo "round $round park $1 nt $2: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms_per_step %.4f kernel_ms %.4f achieved %.1f GB/s frac %.4f" % (d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"]))')"
    done
  done
} > "$OUT" 2>&1
tail -40 "$OUT"
