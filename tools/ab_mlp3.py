#!/usr/bin/env python3
"""Within-process interleaved A/B of the fused-MLP kernel variants (build with `make PROBES=1`).

usage: python tools/ab_mlp3.py [--variants 0,1,2] [--rounds 5] [--iters 3] [--rows 10000000]
Prints per variant: median / min kernel ms per launch, TFLOP/s at the median, and max |diff| against
variant 0 on a sample (diagnostic variants are expected to differ).
"""
import argparse
import os
import statistics
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from infera_amd import capi, onnx_writer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,1,2,3,4,5,6,7,8")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000)
    a = ap.parse_args()
    variants = [int(v) for v in a.variants.split(",")]
    tmp = tempfile.mkdtemp()
    capi.load_model("ab", onnx_writer.write(os.path.join(tmp, "mlp.onnx"), onnx_writer.mlp()))
    dev = capi.device_ordinal(0)
    rows = a.rows
    d_in = capi.DeviceBuffer(dev, rows * 128 * 4)
    d_out = capi.DeviceBuffer(dev, rows * 4)
    capi.synth_fill(d_in, 42, 0, rows, 128)
    times = {v: [] for v in variants}
    sample = {}
    for v in variants:  # warm + correctness sample
        os.environ["INFERA_MLP3_VARIANT"] = str(v)
        capi.predict_device("ab", d_in, rows, 128, d_out)
        sample[v] = d_out.download((4096,))
    for _ in range(a.rounds):
        for v in variants:
            os.environ["INFERA_MLP3_VARIANT"] = str(v)
            times[v].append(capi.time_predict_device("ab", d_in, rows, 128, d_out, a.iters) / a.iters)
    print(f"{'variant':>7} {'median_ms':>10} {'min_ms':>10} {'TFLOP/s@med':>12} {'frac':>6} {'max|d| vs v0':>14}")
    for v in variants:
        med, mn = statistics.median(times[v]), min(times[v])
        tf = 98432.0 * rows / (med / 1e3) / 1e12
        d = float(np.max(np.abs(sample[v] - sample[variants[0]])))
        print(f"{v:>7} {med:>10.4f} {mn:>10.4f} {tf:>12.2f} {tf / 157.3:>6.3f} {d:>14.3e}")


if __name__ == "__main__":
    main()
