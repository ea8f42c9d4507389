"""INFERA_PRECISION=bf16x3 -- the OPTIONAL fast mode of the fused MLP (csrc/hip/mlp_bf16x3.hip): every fp32 product as
hi*hi + hi*lo + lo*hi on the bf16 matrix cores.  It is NOT the parity path (the 1e-4 bar of north_star is met by the
fp32 kernel; this mode is reported as non-parity precision).  Checked here: it is off by default, the plan labels it, its
error against the oracle is of the order 2^-16 of the operands' magnitude, and ragged / tiny inputs work.
The library reads INFERA_PRECISION once, so the mode runs in a child process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from infera_amd import capi, onnx_writer as W, synth
from oracle import oracle
out = {}
for d3 in (1, 3):
    path = W.write(os.path.join(%(tmp)r, "mlp%%d.onnx" %% d3), W.mlp((128, 256, 64, d3)))
    capi.load_model("m", path)
    plan = capi.get_plan("m")
    om = oracle.Model(path)
    rec = {"kernel": plan.get("fused_kernel"), "precision": plan.get("precision")}
    for rows in (1, 33, 4096 + 17):
        x = synth.table(42, 0, rows, 128)
        want = om.predict(x)
        got = capi.predict("m", x)                      # host ABI (row-major staging)
        cols = [np.ascontiguousarray(x[:, j]) for j in range(128)]
        got_c = capi.predict_columns("m", cols)         # columnar path (GPU transpose, then the same kernel)
        dev = capi.device_ordinal(0)
        d_in = capi.DeviceBuffer(dev, rows * 128 * 4); d_out = capi.DeviceBuffer(dev, rows * d3 * 4)
        capi.synth_fill(d_in, 42, 0, rows, 128)
        capi.predict_device("m", d_in, rows, 128, d_out)
        got_d = d_out.download((rows, d3))
        err = np.abs(got.astype(np.float64) - want.astype(np.float64))
        rec[str(rows)] = {"max_abs_err": float(err.max()), "max_rel_err": float((err / np.maximum(np.abs(want), 1e-3)).max()),
                          "scale": float(np.abs(want).max()), "same_paths": bool(np.array_equal(got, got_c) and np.array_equal(got, got_d)),
                          "within_parity_bar": bool(np.all(err <= 1e-4 * np.abs(want) + 1e-6))}
    out[str(d3)] = rec
    capi.unload_model("m")
print("RESULT " + json.dumps(out))
"""


def _run(tmp_path, precision):
    env = dict(os.environ)
    env.pop("INFERA_PRECISION", None)
    if precision:
        env["INFERA_PRECISION"] = precision
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tmp": str(tmp_path)}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


@pytest.mark.gpu
def test_bf16x3_mode_is_optional_labelled_and_close(gpu_api, tmp_path):
    default = _run(tmp_path, None)
    fast = _run(tmp_path, "bf16x3")
    for d3 in ("1", "3"):
        assert default[d3]["precision"] == "fp32" and "bf16x3" not in default[d3]["kernel"]
        assert fast[d3]["precision"].startswith("bf16x3") and "NOT parity" in fast[d3]["precision"] and "bf16x3" in fast[d3]["kernel"]
        for rows in ("1", "33", "4113"):
            d, f = default[d3][rows], fast[d3][rows]
            assert d["within_parity_bar"] and d["same_paths"]                 # the default path is the parity path
            assert f["same_paths"]                                            # host, columnar and device-resident entries agree bit for bit
            assert f["max_abs_err"] <= 3e-5 * max(f["scale"], 0.1), (d3, rows, f)  # ~2^-15 of the output scale
    print("bf16x3 error vs oracle:", json.dumps({k: {r: fast[k][r] for r in ("1", "33", "4113")} for k in fast}))
