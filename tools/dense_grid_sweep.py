"""Sweeps single Dense layers (K x M grid, with and without softmax) over a 2M-row table and lists the slowest shapes:
anything well under both the stream ceiling (~5 TB/s in+out) and the MFMA rate of the tiled kernels is a kernel-selection
gap.  usage (GPU box): python tools/dense_grid_sweep.py"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
from infera_amd import capi, onnx_writer as W
d = tempfile.mkdtemp(); dev = capi.device_ordinal(0)
rows = 2_000_000
res = []
for k in (3, 13, 30, 64, 100, 128, 200, 300, 512, 561, 1000, 2048):
    for m in (1, 3, 10, 16, 20, 32, 50, 100, 256, 1000):
        for sm in (False, True):
            if sm and m == 1:
                continue
            if k * m > 600_000:
                continue
            name = f"g{k}x{m}{int(sm)}"
            capi.load_model(name, W.write(f"{d}/{name}.onnx", W.mlp((k, m), final_softmax=sm)))
            plan = capi.get_plan(name)
            d_in, d_out = capi.DeviceBuffer(dev, rows * k * 4), capi.DeviceBuffer(dev, rows * m * 4)
            capi.synth_fill(d_in, 42, 0, rows, k)
            capi.predict_device(name, d_in, rows, k, d_out)
            ms = capi.time_predict_device(name, d_in, rows, k, d_out, 3) / 3
            tbs = rows * 4 * (k + m) / ms / 1e9
            tf = 2.0 * rows * k * m / ms / 1e9
            res.append((max(tbs / 5.0, tf / 100.0), k, m, sm, ms, tbs, tf, ",".join(plan["exec"])))
            capi.unload_model(name); del d_in, d_out
res.sort()
print("worst 25 of", len(res), "(score = max(TB/s / 5, TFLOP/s / 100))")
for sc, k, m, sm, ms, tbs, tf, ex in res[:25]:
    print(f"  {k:>5} x {m:<5} sm={int(sm)}  {ms:8.3f} ms  {tbs:5.2f} TB/s  {tf:6.1f} TFLOP/s  score {sc:.2f}  {ex}")
