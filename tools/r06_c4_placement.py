#!/usr/bin/env python3
"""(round 6) why the SAME C4 kernel reads 4.7-5.4 ms per 50M rows from one process to the next on one box: is it where the 25.6 GB table (and the
2 GB of results) land in HBM?  One process; the buffers are freed and re-allocated between trials -- in the same order, in the opposite order,
with the table shifted inside a larger allocation, with a hole left by a freed allocation in front.  HIP events around 10 launches each."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from infera_amd import capi, onnx_writer  # noqa: E402

rows = 50_000_000
tmp = tempfile.mkdtemp()
capi.load_model("c4", onnx_writer.write(os.path.join(tmp, "logreg.onnx"), onnx_writer.logreg_softmax(128, 10)))
dev = capi.device_ordinal(0)
lib = capi.load_library()


class Shifted:  # a view `shift` bytes into a DeviceBuffer (same interface as far as predict_device / synth_fill need: .ptr)
    def __init__(self, buf, shift):
        self.buf, self.ptr, self.device, self.nbytes = buf, buf.ptr + shift, buf.device, buf.nbytes - shift


def trial(label, order="in_first", shift_in=0, shift_out=0, hole=0):
    h = capi.DeviceBuffer(dev, hole) if hole else None
    if order == "in_first":
        a = capi.DeviceBuffer(dev, rows * 512 + shift_in)
        b = capi.DeviceBuffer(dev, rows * 40 + shift_out)
    else:
        b = capi.DeviceBuffer(dev, rows * 40 + shift_out)
        a = capi.DeviceBuffer(dev, rows * 512 + shift_in)
    if h:
        h.free()
    d_in, d_out = Shifted(a, shift_in), Shifted(b, shift_out)
    capi.synth_fill(d_in, 42, 0, rows, 128)
    capi.predict_device("c4", d_in, rows, 128, d_out)
    ms = [capi.time_predict_device("c4", d_in, rows, 128, d_out, 10) / 10 for _ in range(3)]
    print(f"{label:<58} in @ {d_in.ptr:#x} out @ {d_out.ptr:#x}: {ms[0]:.4f} {ms[1]:.4f} {ms[2]:.4f} ms = {rows * 552 / min(ms) / 1e9:.3f} TB/s", flush=True)
    a.free()
    b.free()


for rep in range(3):
    trial(f"rep {rep}: table first, results second")
trial("results first, table second", order="out_first")
for s in (4096, 65536, 1 << 20, (1 << 21) + 4096, 1 << 30):
    trial(f"table shifted by {s} bytes", shift_in=s)
for s in (256, 4096, 1 << 20):
    trial(f"results shifted by {s} bytes", shift_out=s)
for hole in (1 << 30, 7 << 30):
    trial(f"a {hole >> 30} GiB allocation in front, freed before the run", hole=hole)
trial("again: table first, results second")
