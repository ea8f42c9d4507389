#!/usr/bin/env python3
"""Generates tests/golden/vectors.npz -- build-authored golden vectors (NOT reference-pinned).

The reference's own golden values (linear.onnx (1,2,3)->1.75 etc.) live in the test files as
literals, next to the reference test they come from.  The reference itself cannot be executed here
(Rust + un-vendored tract-onnx, no cargo), so there is nothing to import from /root/reference: these
vectors are the CPU oracle's outputs (oracle/infera_oracle.c, an ONNX-spec restatement) on seeded
synthetic inputs, frozen so that the oracle, the ONNX writer and the HIP path cannot drift together
unnoticed.  Inputs are not stored: they are regenerated from infera_amd.synth (seed, row0, rows).

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from infera_amd import onnx_writer as W, synth  # noqa: E402
from oracle import oracle  # noqa: E402

CASES = {
    # name: (model bytes, seed, row0, rows, cols)
    "mlp": (W.mlp((128, 256, 64, 1)), 42, 0, 4096, 128),
    "logreg": (W.logreg_softmax(128, 10), 42, 0, 4096, 128),
    "mlp_sigmoid": (W.mlp((16, 32, 32, 3), acts=["Sigmoid", "Tanh", ""]), 5, 10, 257, 16),
    "linear_dyn": (W.linear_dyn(), 3, 0, 1000, 3),
}


def main():
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, (blob, seed, row0, rows, cols) in CASES.items():
            path = W.write(os.path.join(td, name + ".onnx"), blob)
            x = synth.table(seed, row0, rows, cols)
            y = oracle.Model(path).predict(x)
            out[name + "_y"] = y
            out[name + "_meta"] = np.array([seed, row0, rows, cols], np.int64)
            out[name + "_sha256"] = np.frombuffer(hashlib.sha256(blob).digest(), np.uint8)
        # small conv net (ResNet-18 topology at width 8, 32x32 input, 10 classes) on 2 images
        blob = W.resnet18(classes=10, in_hw=32, width=8)
        path = W.write(os.path.join(td, "resnet_small.onnx"), blob)
        x = synth.table(11, 0, 2, 3 * 32 * 32)
        out["resnet_small_y"] = oracle.Model(path).predict_blob(x.tobytes())
        out["resnet_small_meta"] = np.array([11, 0, 2, 3 * 32 * 32], np.int64)
        out["resnet_small_sha256"] = np.frombuffer(hashlib.sha256(blob).digest(), np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "vectors.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
