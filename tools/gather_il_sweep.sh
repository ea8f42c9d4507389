#!/bin/bash
out=gpurun_out/${1:-r03}_gather_il_sweep.txt
{
for cfg in "INFERA_HOST_GATHER=memcpy" "INFERA_HOST_GATHER=il INFERA_GATHER_IL_STREAMS=2" "INFERA_HOST_GATHER=il INFERA_GATHER_IL_STREAMS=4" "INFERA_HOST_GATHER=il INFERA_GATHER_IL_STREAMS=8" "INFERA_HOST_GATHER=il INFERA_GATHER_IL_STREAMS=16" "INFERA_HOST_GATHER=il INFERA_GATHER_IL_STREAMS=8 INFERA_GATHER_IL_BYTES=256" "INFERA_HOST_GATHER=il INFERA_GATHER_IL_STREAMS=8 INFERA_GATHER_IL_BYTES=1024" "INFERA_HOST_GATHER=il INFERA_GATHER_IL_STREAMS=4 INFERA_GATHER_IL_BYTES=2048" "INFERA_HOST_GATHER=ilnt INFERA_GATHER_IL_STREAMS=4" "INFERA_HOST_GATHER=ilnt INFERA_GATHER_IL_STREAMS=8" "INFERA_HOST_GATHER=memcpy"; do
  echo "== $cfg"
  env $cfg python tools/host_scan_bench.py --rows 10000000 --threads 1,4,8,16,24 --numa auto 2>&1 | grep -E "^threads|cpu_us" | sed -E "s/threads= *([0-9]+) +([0-9.]+) M rows.*/T=\1 \2 M rows\/s/; s/.*(.gather.: [0-9.]+).*(.cpu_us_per_chunk.: [0-9.]+).*/      \1 \2/" | paste - -
done
} > $out 2>&1
cat $out
