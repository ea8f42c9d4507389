"""Single-thread latency of one host-ABI call (infera_predict) by chunk size: the floor a DuckDB worker pays per vector.
usage (GPU box): python tools/call_latency.py"""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from infera_amd import capi, onnx_writer as W, synth
d = tempfile.mkdtemp()
models = {"13->1": (W.mlp((13, 1)), 13), "30->100->2": (W.mlp((30, 100, 2), final_softmax=True), 30), "C2 128->256->64->1": (W.mlp(), 128)}
L = capi.load_library()
for name, (blob, k) in models.items():
    capi.load_model("m", W.write(f"{d}/m.onnx", blob))
    for rows in (1, 64, 2048):
        x = synth.table(1, 0, rows, k)
        for _ in range(50):
            capi.predict("m", x)
        ts = []
        for _ in range(400):
            t0 = time.perf_counter()
            res = L.infera_predict(b"m", x.ctypes.data, rows, k)
            ts.append(time.perf_counter() - t0)
            L.infera_free_result(res)
        ts = np.sort(np.array(ts)) * 1e6
        print(f"{name:<20} rows={rows:<5} median {ts[200]:7.1f} us   p10 {ts[40]:7.1f}   p90 {ts[360]:7.1f}")
    capi.unload_model("m")
