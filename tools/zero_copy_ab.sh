#!/bin/bash
# The zero-copy path against the staged path on one GPU (profiles/r04_zero_copy_*.txt): C2 scan over a registered table fetched by ONE 2-D copy
# per chunk (INFERA_ZERO_COPY_RECT=1, default) or by the pulling kernel (=0); table allocation as numpy leaves it, page-aligned, on huge pages;
# the number of hardware queues the HIP runtime creates.   usage: gpurun -- bash tools/zero_copy_ab.sh [tag]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-r04}
O=gpurun_out/${tag}_zero_copy
mkdir -p $O
python -m pytest tests/test_zero_copy_gpu.py -m gpu -x -q 2>&1 | tail -3 > $O/pytest.txt
for rect in 1 0; do
  for mode in "" "--align 4096" "--huge"; do
    echo "=== registered table, INFERA_ZERO_COPY_RECT=$rect, table: ${mode:-numpy default}" >> $O/rect_pages.txt
    INFERA_ZERO_COPY_RECT=$rect python tools/host_scan_bench.py --rows 8000000 --threads 1,2,4,8,16,24 --reps 3 --numa auto --register $mode 2>&1 | grep "^threads\|cpu_us\|^table\|registered" >> $O/rect_pages.txt
  done
done
echo "=== staged path (table not registered)" >> $O/rect_pages.txt
python tools/host_scan_bench.py --rows 8000000 --threads 1,2,4,8,16,24 --reps 3 --numa auto 2>&1 | grep "^threads\|cpu_us" >> $O/rect_pages.txt
for q in 4 8 16; do
  for rect in 0 1; do
    echo "=== GPU_MAX_HW_QUEUES=$q INFERA_ZERO_COPY_RECT=$rect registered table" >> $O/hw_queues.txt
    GPU_MAX_HW_QUEUES=$q INFERA_ZERO_COPY_RECT=$rect python tools/host_scan_bench.py --rows 8000000 --threads 2,8,16 --reps 3 --numa auto --register 2>&1 | grep "^threads" >> $O/hw_queues.txt
  done
done
