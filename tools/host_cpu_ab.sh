#!/bin/bash
# Round-3 host-path A/B on one GPU box: CPU time per chunk (getrusage) by wait mode and gather mode, then the link-elided
# multi-slot probe.  usage: tools/host_cpu_ab.sh <tag>
tag=${1:-r03}
out=gpurun_out/${tag}_host_cpu_ab.txt
T="1,2,4,8,16,24,32"
{
  echo "== cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) =="
  for cfg in "" "INFERA_HOST_WAIT=poll" "INFERA_HOST_WAIT=spin" "INFERA_HOST_GATHER=nt" "INFERA_HOST_GATHER=ntpf" "INFERA_HOST_WAIT=poll INFERA_HOST_GATHER=ntpf"; do
    echo "==== 1 slot, numa auto, $cfg ===="
    env $cfg python tools/host_scan_bench.py --rows 10000000 --threads $T --numa auto 2>&1 | grep -v "^$"
  done
  echo "==== 1 slot, numa off, default ===="
  python tools/host_scan_bench.py --rows 10000000 --threads 8,16,32 2>&1 | grep -v "^$"
  for cfg in "" "INFERA_HOST_WAIT=poll"; do
    echo "==== ELIDED H2D, 1 slot, $cfg ===="
    env $cfg INFERA_HOST_PROBE_ELIDE_H2D=1 python tools/host_scan_bench.py --rows 10000000 --threads 4,8,16,24,32 --numa auto 2>&1 | grep -v "^$"
    echo "==== ELIDED H2D, 8 slots on one GPU, unbound, $cfg ===="
    env $cfg INFERA_HOST_PROBE_ELIDE_H2D=1 INFERA_DEVICES=0,0,0,0,0,0,0,0 python tools/host_scan_bench.py --rows 20000000 --threads 8,16,24,32,48,64 2>&1 | grep -v "^$"
  done
  for slots in "0,0" "0,0,0,0"; do
    echo "==== real H2D, slots $slots ===="
    INFERA_DEVICES=$slots python tools/host_scan_bench.py --rows 10000000 --threads 8,16,32,64 --numa auto 2>&1 | grep -v "^$"
  done
} > $out 2>&1
tail -3 $out
