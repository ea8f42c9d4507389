#!/usr/bin/env python3
"""PCIe-inclusive rate of the BLOB path (C5): T threads, each pushing batches of images through
infera_predict_from_blob (host bytes -> pinned staging -> H2D -> ResNet-18 -> D2H)."""
import argparse
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from infera_amd import capi, onnx_writer, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--calls", type=int, default=8)
ap.add_argument("--threads", default="1,2,4,8")
a = ap.parse_args()
tmp = tempfile.mkdtemp()
capi.load_model("rn", onnx_writer.write(os.path.join(tmp, "rn.onnx"), onnx_writer.resnet18()))
imgs = synth.table(1, 0, a.batch, 3 * 224 * 224)  # the "BLOB column": borrowed by the call, no Python-side copy
L = capi.load_library()


def call():
    res = L.infera_predict_from_blob(b"rn", imgs.ctypes.data, imgs.nbytes)
    assert res.status == 0, capi.last_error()
    L.infera_free_result(res)


call()  # warm
for t in [int(x) for x in a.threads.split(",")]:
    def work():
        for _ in range(a.calls):
            call()
    th = [threading.Thread(target=work) for _ in range(t)]
    t0 = time.perf_counter()
    [x.start() for x in th]
    [x.join() for x in th]
    sec = time.perf_counter() - t0
    n = t * a.calls * a.batch
    print(f"threads={t:>2} batch={a.batch}: {n / sec:>9.0f} img/s  ({n * 602112 / sec / 1e9:.1f} GB/s of image bytes)")
