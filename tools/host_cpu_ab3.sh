#!/bin/bash
# Round-3 host-path A/B, part 3: staging-context affinity, poll nap fractions, NT gather on top.  usage: tools/host_cpu_ab3.sh <tag>
tag=${1:-r03}
out=gpurun_out/${tag}_host_cpu_ab3.txt
{
  for cfg in "INFERA_HOST_CTX_AFFINITY=0" "INFERA_HOST_CTX_AFFINITY=1" "INFERA_HOST_CTX_AFFINITY=1 INFERA_HOST_POLL_FIRST=0.9 INFERA_HOST_POLL_NEXT=0.15" "INFERA_HOST_CTX_AFFINITY=1 INFERA_HOST_POLL_FIRST=0.6 INFERA_HOST_POLL_NEXT=0.2" "INFERA_HOST_CTX_AFFINITY=1 INFERA_HOST_GATHER=ntpf" "INFERA_HOST_CTX_AFFINITY=1 INFERA_HOST_CONTEXTS=12 INFERA_MAX_INFLIGHT=8" "INFERA_HOST_CTX_AFFINITY=0"; do
    echo "==== 1 slot, numa auto, $cfg ===="
    env $cfg python tools/host_scan_bench.py --rows 10000000 --threads 1,2,4,8,12,16,24 --numa auto --reps 5 2>&1 | grep -v "^$"
  done
  for cfg in "INFERA_HOST_CTX_AFFINITY=0" "INFERA_HOST_CTX_AFFINITY=1"; do
    echo "==== ELIDED H2D + token kernel, 8 slots on one GPU, unbound, $cfg ===="
    env $cfg INFERA_HOST_PROBE_ELIDE_H2D=2 INFERA_DEVICES=0,0,0,0,0,0,0,0 INFERA_MAX_INFLIGHT=0 python tools/host_scan_bench.py --rows 20000000 --threads 16,24,32,48 2>&1 | grep -v "^$"
  done
} > $out 2>&1
tail -3 $out
