#!/bin/bash
tag=${1:-r03}
out=gpurun_out/${tag}_zero_copy.txt
{
  python -m pytest tests/test_zero_copy_gpu.py -m gpu -x -q 2>&1 | tail -5
  echo "==== staged, 1 slot, numa auto ===="
  python tools/host_scan_bench.py --rows 10000000 --threads 1,2,4,8,16 --numa auto 2>&1 | grep -v "^$"
  echo "==== REGISTERED table (zero-copy), 1 slot, numa auto ===="
  python tools/host_scan_bench.py --rows 10000000 --threads 1,2,4,8,16,24 --numa auto --register 2>&1 | grep -v "^$"
  echo "==== REGISTERED table (zero-copy), DOUBLE columns ===="
  python tools/host_scan_bench.py --rows 6000000 --threads 2,4,8 --numa auto --register --double 2>&1 | grep -v "^$"
  echo "==== REGISTERED, C4 logreg ===="
  python tools/host_scan_bench.py --rows 10000000 --threads 2,4,8,16 --numa auto --register --workload logreg 2>&1 | grep -v "^$"
  echo "==== REGISTERED, 8 slots on one GPU, token kernels (host side only) ===="
  INFERA_HOST_PROBE_ELIDE_H2D=2 INFERA_DEVICES=0,0,0,0,0,0,0,0 INFERA_MAX_INFLIGHT=0 python tools/host_scan_bench.py --rows 20000000 --threads 8,16,32 --register 2>&1 | grep -v "^$"
} > $out 2>&1
cat $out | cut -c1-330
