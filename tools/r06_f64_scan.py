import os, sys, tempfile
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
from infera_amd import capi, onnx_writer, sqlharness
rows = 6_000_000
tmp = tempfile.mkdtemp()
capi.load_model("m", onnx_writer.write(os.path.join(tmp, "mlp.onnx"), onnx_writer.mlp((128, 256, 64, 1))))
t = sqlharness.synth_table(rows, 128, 42, 16, dtype=np.float64)
def run(label):
    out = []
    sqlharness.bench_scan_table("infera_predict", "m", t, 2048 * 300, 128, 4, 1)
    for th in (1, 2, 4, 8, 16):
        (secs, cs), ph = sqlharness.phase_breakdown(sqlharness.bench_scan_table, "infera_predict", "m", t, rows, 128, th, 3)
        med = sorted(secs)[1]
        out.append(f"{rows / med / 1e6:.1f} / {ph['cpu_us_per_chunk']:.0f}")
    print(f"| {label} | " + " | ".join(out) + " |", flush=True)
print("DOUBLE columns, C2, M rows/s / CPU us per chunk at 1 / 2 / 4 / 8 / 16 callers (link bound: 2 MiB per chunk -> ~53 M rows/s)")
run("staged (f64 -> f32 converted by the gather)")
capi.register_host_memory(t)
b = capi.zero_copy_calls()
run("registered (pulling kernel converts)")
print("zero-copy calls", capi.zero_copy_calls() - b)
capi.unregister_host_memory(t)
