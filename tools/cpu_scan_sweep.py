#!/usr/bin/env python3
"""Thread sweep of the oracle's CPU scan baseline (oracle/infera_oracle.c orc_bench_scan) on this box."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infera_amd import onnx_writer  # noqa: E402
from oracle import oracle  # noqa: E402

tmp = tempfile.mkdtemp()
m = oracle.Model(onnx_writer.write(os.path.join(tmp, "mlp.onnx"), onnx_writer.mlp()))
print("cpu_count", os.cpu_count())
for th in [1, 2, 8, 32, 64, 128, 256]:
    if th > (os.cpu_count() or 1):
        break
    rows = 2048 * max(th * 4, 16)
    for boxed in (True, False):
        sec, _ = m.bench_scan(rows, 128, threads=th, boxed=boxed)
        print(f"threads={th:>3} boxed={int(boxed)} rows={rows:>8} {rows / sec / 1e3:>10.1f} k rows/s  ({rows / sec / th / 1e3:.1f} k/thread)")
