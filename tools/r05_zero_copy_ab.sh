#!/bin/bash
# Round 5: the zero-copy fetch -- 2-D copy (the runtime runs them one at a time) for the first R chunks in flight on a GPU, pulling kernel (one wave
# per column run) for the others; every chunk in place (MAX_INFLIGHT=0) vs surplus beyond N staged; by caller count.  Zero-copy parity tests first.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_zero_copy; mkdir -p $OUT; rm -f $OUT/ab.txt
( timeout 600 python -m pytest tests/test_zero_copy_gpu.py tests/test_gather_colmajor.py tests/test_duckdb_extension.py -q -x 2>&1 | tail -3 ) > $OUT/pytest.txt
run() {
  echo "=== $1 registered table" >> $OUT/ab.txt
  env $1 python tools/host_scan_bench.py --rows 6000000 --threads ${THREADS:-1,2,3,4,6,8,16} --reps 3 --numa auto --register 2>&1 | grep -A1 "threads=" >> $OUT/ab.txt
}
for cfg in "$@"; do run "$cfg"; done
cat $OUT/pytest.txt; grep -E "===|threads=" $OUT/ab.txt | cut -c1-100
