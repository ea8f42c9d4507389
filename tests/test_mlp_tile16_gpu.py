"""The 16-row tile form of the fused MLP (mlp3_tile16_kernel, round 5): chunks of up to 4096 rows through the host ABI run on
v_mfma_f32_16x16x4_f32 blocks, longer ones on the 32-row tile kernel, device-resident scans on the split kernel -- and every sum keeps the
split kernel's order, so a chunk equals its slice of a resident scan BIT FOR BIT whichever of the three served it.  Row-major (infera_predict)
and column-major (infera_predict_columns) staging, ragged tile tails, one and three outputs; against the oracle within the parity tolerance."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from infera_amd import capi, onnx_writer, synth
out = {}
# (the third shape has no ahead-of-time instantiation: hipRTC compiles its kernels at load -- mlp_jit.cpp reads the same knob, ADVICE r5)
for name, dims in (("c2", (128, 256, 64, 1)), ("c2x3", (128, 256, 64, 3)), ("jit", (64, 128, 32, 2))):
    path = onnx_writer.write(os.path.join(%(tmp)r, name + ".onnx"), onnx_writer.mlp(dims))
    capi.load_model(name, path)
    h = hashlib.sha256()
    for rows in (1, 15, 16, 17, 31, 33, 2047, 2048, 4096, 4097, 6000):
        x = synth.table(5, 0, rows, dims[0])
        a = capi.predict(name, x)                                              # row-major staging
        b = capi.predict_columns(name, [np.ascontiguousarray(x[:, c]) for c in range(dims[0])])   # column-major staging
        assert a.shape == (rows, dims[-1]) and np.array_equal(a, b), (name, rows)
        h.update(a.tobytes())
    out[name] = h.hexdigest()
    out[name + "_exec"] = capi.get_plan(name)["exec"]
print("RESULT " + json.dumps(out))
"""


def run_child(tmp_path, tile16_rows):
    env = dict(os.environ, INFERA_MLP_TILE16_MAX_ROWS=str(tile16_rows))
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tmp": str(tmp_path)}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_tile16_equals_tile32_bit_for_bit(built, tmp_path):
    with16 = run_child(tmp_path, 4096)   # the shipped limit: chunks up to 4096 rows on 16-row tiles
    never = run_child(tmp_path, 0)       # 32-row tiles only (round 4's path)
    always = run_child(tmp_path, 1 << 20)
    assert with16["c2_exec"][0] == "mlp3_fused" and with16["jit_exec"][0] == "mlp3_fused"
    assert with16 == never == always


def test_tile16_chunk_equals_its_slice_of_a_resident_scan_and_the_oracle(built, tmp_path):
    from infera_amd import capi, onnx_writer, synth
    from oracle import oracle

    path = onnx_writer.write(os.path.join(str(tmp_path), "m.onnx"), onnx_writer.mlp())
    capi.load_model("t16", path)
    try:
        rows = 50_000
        dev = capi.device_ordinal(0)
        d_in, d_out = capi.DeviceBuffer(dev, rows * 128 * 4), capi.DeviceBuffer(dev, rows * 4)
        capi.synth_fill(d_in, 42, 0, rows, 128)
        capi.predict_device("t16", d_in, rows, 128, d_out)  # the split kernel
        scan = d_out.download((rows, 1))
        ref = oracle.Model(path)
        for r0, n in ((0, 2048), (2048 * 7 + 5, 2048), (rows - 100, 100), (12345, 1), (777, 16), (4096, 4000)):
            x = synth.table(42, r0, n, 128)
            got = capi.predict_columns("t16", [np.ascontiguousarray(x[:, c]) for c in range(128)])
            assert np.array_equal(got, scan[r0:r0 + n]), (r0, n)
            want = ref.predict(x)
            assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-6)
    finally:
        capi.unload_model("t16")
