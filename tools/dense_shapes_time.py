#!/usr/bin/env python3
"""Device-resident throughput of MLPs that do NOT match the ahead-of-time fused chain: which kernel each takes and
what it reaches (TFLOP/s, rows/s)."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infera_amd import capi, onnx_writer as W  # noqa: E402

d = tempfile.mkdtemp()
dev = capi.device_ordinal(0)
shapes = [((128, 256, 64, 1), 4_000_000), ((128, 256, 64), 4_000_000), ((128, 512, 512, 10), 2_000_000), ((256, 256, 256, 256, 1), 2_000_000),
          ((64, 128, 1), 8_000_000), ((1024, 1024, 1024), 500_000), ((128, 96, 48, 3), 4_000_000), ((32, 64, 32, 1), 8_000_000),
          ((128, 100, 1), 4_000_000), ((100, 100, 100, 2), 4_000_000), ((30, 100, 50, 1), 4_000_000)]
for dims, rows in shapes:
    name = "m" + "x".join(map(str, dims))
    capi.load_model(name, W.write(f"{d}/{name}.onnx", W.mlp(dims)))
    plan = capi.get_plan(name)
    d_in, d_out = capi.DeviceBuffer(dev, rows * dims[0] * 4), capi.DeviceBuffer(dev, rows * dims[-1] * 4)
    capi.synth_fill(d_in, 42, 0, rows, dims[0])
    capi.predict_device(name, d_in, rows, dims[0], d_out)
    ms = capi.time_predict_device(name, d_in, rows, dims[0], d_out, 5) / 5
    flops = plan["plan"]["flops_per_row"] * rows
    byts = rows * 4 * (dims[0] + dims[-1])
    print(f"{'x'.join(map(str, dims)):<22} rows {rows:>9}  {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  {byts / ms / 1e9:6.2f} TB/s(in+out)  {rows / ms / 1e6:7.1f} G rows/s  "
          f"{plan.get('fused_kernel', ','.join(plan['exec']))[:60]}")
    capi.unload_model(name)
    del d_in, d_out
