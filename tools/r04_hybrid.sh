#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_hybrid
mkdir -p $O
for k in 0 1 2 3 4 6 8; do
  echo "=== registered table, INFERA_ZERO_COPY_MAX_INFLIGHT=$k" >> $O/hybrid.txt
  INFERA_ZERO_COPY_MAX_INFLIGHT=$k timeout 300 python tools/host_scan_bench.py --rows 8000000 --threads 4,8,16,24 --reps 3 --numa auto --register 2>&1 | grep "^threads\|cpu_us\|zero-copy calls" >> $O/hybrid.txt
done
export TMPDIR=/tmp
bash tools/sanitizers_run.sh > $O/sanitizers.txt 2>&1
cp gpurun_out/tsan/asan_*.txt $O/
timeout 300 python -m pytest tests/test_zero_copy_gpu.py -m gpu -q 2>&1 | tail -3 > $O/pytest_zc.txt
