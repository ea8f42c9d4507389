#!/bin/bash
# round 4: what bounds the zero-copy plateau (41-42 GB/s with either fetch mechanism)?  Page granularity of the registered table:
# unaligned 8 KB runs touch three 4 KiB pages, page-aligned ones two; transparent huge pages: one 2 MiB page covers 256 chunks' worth of a column
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04_step4
mkdir -p $O
cat /sys/kernel/mm/transparent_hugepage/enabled > $O/thp.txt 2>&1
for rect in 0 1; do
  for mode in "" "--align 4096" "--huge"; do
    echo "=== INFERA_ZERO_COPY_RECT=$rect table: ${mode:-numpy default}" >> $O/zero_copy_pages.txt
    INFERA_ZERO_COPY_RECT=$rect timeout 300 python tools/host_scan_bench.py --rows 8000000 --threads 2,4,8,16 --reps 3 --numa auto --register $mode 2>&1 | grep "^threads\|cpu_us\|^table\|registered" >> $O/zero_copy_pages.txt
  done
done
for mode in "" "--align 4096" "--huge"; do
  echo "=== staged, table: ${mode:-numpy default}" >> $O/zero_copy_pages.txt
  timeout 300 python tools/host_scan_bench.py --rows 8000000 --threads 8,16,24 --reps 3 --numa auto $mode 2>&1 | grep "^threads\|cpu_us\|^table" >> $O/zero_copy_pages.txt
done
echo done > $O/done.txt
