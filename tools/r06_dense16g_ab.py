#!/usr/bin/env python3
"""(round 6) the generic-row-length narrow Dense kernel (tables of 8..128 columns) with and without the parked 16-byte result pieces +
non-temporal loads and stores, same process, same buffers, interleaved rounds (as tools/r06_c4_same_process_ab.py)."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from infera_amd import capi, onnx_writer  # noqa: E402

tmp = tempfile.mkdtemp()
dev = capi.device_ordinal(0)
for k, m, softmax, rows in ((30, 2, True, 100_000_000), (13, 3, True, 100_000_000), (100, 10, True, 50_000_000), (52, 16, False, 50_000_000)):
    name = f"m{k}_{m}"
    capi.load_model(name, onnx_writer.write(os.path.join(tmp, name + ".onnx"), onnx_writer.mlp((k, m), acts=[""], final_softmax=softmax)))
    kern = capi.get_plan(name)["dense_kernels"][0].split("<")[0]
    d_in, d_out = capi.DeviceBuffer(dev, rows * k * 4), capi.DeviceBuffer(dev, rows * m * 4)
    capi.synth_fill(d_in, 42, 0, rows, k)
    res = {}
    for r in range(4):
        for park, nt in ((0, 0), (1, 0), (0, 1), (1, 1)):
            os.environ["INFERA_DENSE16S_MODE"], os.environ["INFERA_DENSE16S_NT"] = str(park), str(nt)
            capi.predict_device(name, d_in, rows, k, d_out)
            res.setdefault((park, nt), []).append(capi.time_predict_device(name, d_in, rows, k, d_out, 10) / 10)
    byts = rows * (k + m) * 4
    print(f"{k} -> {m}{' + softmax' if softmax else ''}, {rows} rows ({kern}): " + "  ".join(
        f"park {p} nt {n}: {sorted(v)[len(v) // 2]:.3f} ms = {byts / sorted(v)[len(v) // 2] / 1e9:.2f} TB/s" for (p, n), v in res.items()), flush=True)
    d_in.free()
    d_out.free()
    capi.unload_model(name)
