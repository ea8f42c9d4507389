#!/bin/bash
# Round-3 host-path A/B, part 4: does thread migration cost the gather?  pinned scan workers vs L3-domain staging affinity.
tag=${1:-r03}
out=gpurun_out/${tag}_host_cpu_ab4.txt
{
  for cfg in "INFERA_HOST_CTX_AFFINITY=1" "INFERA_HOST_CTX_AFFINITY=2" "INFERA_HOST_CTX_AFFINITY=1 INFERA_BENCH_PIN=spread" "INFERA_HOST_CTX_AFFINITY=1 INFERA_BENCH_PIN=pack" "INFERA_HOST_CTX_AFFINITY=2 INFERA_HOST_GATHER=ntpf" "INFERA_HOST_CTX_AFFINITY=1"; do
    echo "==== 1 slot, numa auto, $cfg ===="
    env $cfg python tools/host_scan_bench.py --rows 10000000 --threads 1,2,4,8,12,16,24 --numa auto --reps 5 2>&1 | grep -v "^$"
  done
} > $out 2>&1
tail -3 $out
