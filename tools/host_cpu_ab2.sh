#!/bin/bash
# Round-3 host-path A/B, part 2: poll vs pollq, intra-call sub-pass split at low thread counts, link+kernel-elided 8-slot probe,
# CPU baseline variants.  usage: tools/host_cpu_ab2.sh <tag>
tag=${1:-r03}
out=gpurun_out/${tag}_host_cpu_ab2.txt
{
  for cfg in "INFERA_HOST_SPLIT=0" "INFERA_HOST_SPLIT=1" "INFERA_HOST_SPLIT=2" "INFERA_HOST_SPLIT=4" "INFERA_HOST_SPLIT=0 INFERA_HOST_WAIT=pollq" "INFERA_HOST_SPLIT=1 INFERA_HOST_WAIT=pollq"; do
    echo "==== 1 slot, numa auto, $cfg ===="
    env $cfg python tools/host_scan_bench.py --rows 10000000 --threads 1,2,3,4,6,8,16,24 --numa auto 2>&1 | grep -v "^$"
  done
  for cfg in "INFERA_HOST_PROBE_ELIDE_H2D=2 INFERA_HOST_SPLIT=0" "INFERA_HOST_PROBE_ELIDE_H2D=2 INFERA_HOST_SPLIT=0 INFERA_HOST_WAIT=pollq" "INFERA_HOST_PROBE_ELIDE_H2D=2 INFERA_HOST_SPLIT=0 INFERA_HOST_GATHER=ntpf" "INFERA_HOST_PROBE_ELIDE_H2D=2 INFERA_HOST_SPLIT=1"; do
    echo "==== ELIDED H2D + token kernel, 8 slots on one GPU, unbound, $cfg ===="
    env $cfg INFERA_DEVICES=0,0,0,0,0,0,0,0 INFERA_MAX_INFLIGHT=0 python tools/host_scan_bench.py --rows 20000000 --threads 8,16,24,32,48 2>&1 | grep -v "^$"
  done
  echo "==== CPU baseline variants (oracle) ===="
  python - <<'PY'
import os, tempfile
from infera_amd import onnx_writer as W, sqlmock
from oracle import oracle
p = W.write(os.path.join(tempfile.mkdtemp(), "m.onnx"), W.mlp((128, 256, 64, 1)))
m = oracle.Model(p)
rows = sqlmock.ROW_GROUP * 16
t = sqlmock.synth_table(rows, 128, 42, 16)
for b in (1, 0, 2):
    for th in (1, 8, 16, 32):
        s, _ = oracle.bench_scan_table(m, t, rows, 128, threads=th, boxed=b)
        print(f"boxed={b} threads={th:>2} {rows / s / 1e6:8.2f} M rows/s  {rows * 98432 / s / 1e9 / min(th, 16):7.1f} GFLOP/s per CPU (of min(threads,16))")
PY
} > $out 2>&1
tail -3 $out
