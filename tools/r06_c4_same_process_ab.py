#!/usr/bin/env python3
"""(round 6) C4 kernel variants timed in ONE process on the SAME buffers (process-to-process the same kernel reads 4.9-5.4 ms on one box:
where the 25.6 GB table lands in HBM matters more than the variants): INFERA_DENSE16S_MODE (park the tile's results, 16-byte pieces) x
INFERA_DENSE16S_NT (non-temporal loads + stores), read per launch.  Interleaved rounds, HIP events around 10 back-to-back launches."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from infera_amd import capi, onnx_writer  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tmp = tempfile.mkdtemp()
capi.load_model("c4", onnx_writer.write(os.path.join(tmp, "logreg.onnx"), onnx_writer.logreg_softmax(128, 10)))
dev = capi.device_ordinal(0)
d_in, d_out = capi.DeviceBuffer(dev, rows * 128 * 4), capi.DeviceBuffer(dev, rows * 10 * 4)
capi.synth_fill(d_in, 42, 0, rows, 128)
configs = [(0, 0), (1, 0), (0, 1), (1, 1)]
ref = None
res = {c: [] for c in configs}
for r in range(rounds):
    for park, nt in configs:
        os.environ["INFERA_DENSE16S_MODE"], os.environ["INFERA_DENSE16S_NT"] = str(park), str(nt)
        capi.predict_device("c4", d_in, rows, 128, d_out)
        y = d_out.download((rows, 10))[:: max(1, rows // 4096)].copy()
        if ref is None:
            ref = y
        assert (y == ref).all(), (park, nt)  # every variant: the same bits
        ms = capi.time_predict_device("c4", d_in, rows, 128, d_out, 10) / 10
        res[(park, nt)].append(ms)
        print(f"round {r} park {park} nt {nt}: {ms:.4f} ms = {rows * 552 / ms / 1e9:.3f} TB/s", flush=True)
for c in configs:
    v = sorted(res[c])
    print(f"park {c[0]} nt {c[1]}: median {v[len(v) // 2]:.4f} ms = {rows * 552 / v[len(v) // 2] / 1e9:.3f} TB/s (min {v[0]:.4f}, max {v[-1]:.4f})")
