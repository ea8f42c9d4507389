"""Latency of ONE infera_predict_from_blob call on ResNet-18 by batch size (the reference's binding calls it per row, batch 1).
usage (GPU box): python tools/blob_call_latency.py"""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from infera_amd import capi, onnx_writer as W, synth

d = tempfile.mkdtemp()
capi.load_model("r18", W.write(d + "/r18.onnx", W.resnet18(in_hw=224)))
for n in (1, 2, 4, 8, 16, 64):
    blob = synth.table(3, 0, n, 3 * 224 * 224).tobytes()
    for _ in range(3):
        capi.predict_from_blob("r18", blob)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        capi.predict_from_blob("r18", blob)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"images={n:>3}  median {1e3 * ts[len(ts) // 2]:8.3f} ms   p10 {1e3 * ts[2]:8.3f}   per image {1e3 * ts[len(ts) // 2] / n:7.3f} ms   INFERA_HIPGRAPH={os.environ.get('INFERA_HIPGRAPH', '0')}")

# where a 64-image call spends its time (library phase counters, us per chunk/pass)
for n in (16, 64, 256):
    blob = synth.table(3, 0, n, 3 * 224 * 224).tobytes()
    capi.predict_from_blob("r18", blob)
    before = capi.get_devices().get("host_phases", {})
    t0 = time.perf_counter()
    capi.predict_from_blob("r18", blob)
    dt = time.perf_counter() - t0
    after = capi.get_devices().get("host_phases", {})
    print(f"images={n}: {1e3 * dt:.2f} ms; phases before {before} after {after}")
