"""Streamed chunks (round 5; csrc/hip/host_path.cpp run_streamed, mlp_device.inc mlp3_stream_kernel): on a quiet GPU a fused-MLP model's
DataChunk is not staged-then-copied -- the kernel is launched FIRST and consumes the chunk's columns out of pinned staging, sixteen at a time,
while the caller's thread is still gathering them.  Replaces the copy-in / run of engine.rs:139-145 for the shape DuckDB calls with.

Checked here: results are bit-identical to the staged path (itself bit-identical to the device-resident scan, tests/test_parity_gpu.py) for
every row count up to the streaming limit, for typed / constant columns, for the 3-output chain; longer chunks and busy GPUs fall back to
staging; concurrent callers mix both paths and agree; an exception half way through the gather (test hook) ends the kernel at once, reaches
the caller as an error, and the next chunk on that context works."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cols(x):
    return [np.ascontiguousarray(x[:, j]) for j in range(x.shape[1])]


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(128, 256, 64, 1), (128, 256, 64, 3)])
def test_gpu_streamed_chunks_equal_staged_chunks_bit_for_bit(gpu_api, tmp_path, dims):
    from oracle import oracle

    path = W.write(str(tmp_path / "m.onnx"), W.mlp(dims))
    gpu_api.load_model("sm", path)
    try:
        x = synth.table(3, 0, 4200, 128)
        want = gpu_api.predict("sm", x)  # row-major host path: staged (infera_predict has no columns to stream)
        ora = oracle.Model(path).predict(x[:2048])
        assert np.all(np.abs(want[:2048] - ora) <= 1e-4 * np.abs(ora) + 1e-6)
        before = gpu_api.get_devices()["streamed_calls"]
        streamed = 0
        for rows in (1, 2, 31, 32, 33, 63, 777, 2047, 2048, 4096):
            got = gpu_api.predict_columns("sm", _cols(x[:rows]))
            assert np.array_equal(got, want[:rows]), rows
            streamed += 1
        assert gpu_api.get_devices()["streamed_calls"] - before == streamed  # a lone caller: every chunk up to 4096 rows is streamed
        # an offset slice of a longer table (runs that do not start a page), and a chunk above the limit: staged, same bits
        got = gpu_api.predict_columns("sm", [np.ascontiguousarray(x[:, j])[77:77 + 2048] for j in range(128)])
        assert np.array_equal(got, want[77:77 + 2048])
        mid = gpu_api.get_devices()["streamed_calls"]
        got = gpu_api.predict_columns("sm", _cols(x[:4097]))
        assert np.array_equal(got, want[:4097]) and gpu_api.get_devices()["streamed_calls"] == mid
        # typed and constant columns: converted by the gather on their way into the staging the kernel is already reading
        y = x[:2048].copy()
        y[:, 5] = np.round(y[:, 5] * 1000)
        y[:, 9] = 0.25
        cols = _cols(y)
        cols[3] = cols[3].astype(np.float64)
        cols[5] = cols[5].astype(np.int32)
        cols[9] = np.array([0.25], np.float32)  # CONSTANT_VECTOR
        cols[17] = cols[17].astype(np.float64)
        assert np.array_equal(gpu_api.predict_columns("sm", cols, rows=2048), gpu_api.predict("sm", y))
    finally:
        gpu_api.unload_model("sm")


@pytest.mark.gpu
def test_gpu_concurrent_callers_mix_streamed_and_staged_chunks(gpu_api, tmp_path):
    """12 threads x 40 chunks: at most INFERA_STREAM_MAX_INFLIGHT (4) chunks stream at a time, the others are staged; every result matches."""
    path = W.write(str(tmp_path / "m.onnx"), W.mlp((128, 256, 64, 1)))
    gpu_api.load_model("smc", path)
    try:
        x = synth.table(5, 0, 2048 * 6, 128)
        want = gpu_api.predict("smc", x)
        chunks = [_cols(x[c * 2048:(c + 1) * 2048]) for c in range(6)]
        bad, before = [], gpu_api.get_devices()
        calls0 = sum(d["host_calls"] for d in before["devices"])

        def work(t):
            for i in range(40):
                c = (t + i) % 6
                if not np.array_equal(gpu_api.predict_columns("smc", chunks[c]), want[c * 2048:(c + 1) * 2048]):
                    bad.append((t, i))

        th = [threading.Thread(target=work, args=(t,)) for t in range(12)]
        [t.start() for t in th]
        [t.join() for t in th]
        after = gpu_api.get_devices()
        assert bad == []
        streamed = after["streamed_calls"] - before["streamed_calls"]
        assert 0 < streamed <= 12 * 40
        assert sum(d["host_calls"] for d in after["devices"]) - calls0 == 12 * 40  # a chunk handed back to the staged path is counted once
    finally:
        gpu_api.unload_model("smc")


CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from infera_amd import capi, onnx_writer as W, synth
capi.load_model("m", W.write(os.path.join(%(tmp)r, "m.onnx"), W.mlp((128, 256, 64, 1))))
x = synth.table(9, 0, 2048, 128)
cols = [np.ascontiguousarray(x[:, j]) for j in range(128)]
want = capi.predict("m", x)
out = {"errors": [], "ok": 0}
for i in range(6):
    try:
        got = capi.predict_columns("m", cols)
        out["ok"] += int(np.array_equal(got, want))
    except Exception as exc:
        out["errors"].append((i, str(exc)))
out["streamed"] = capi.get_devices()["streamed_calls"]
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
def test_gpu_a_gather_that_fails_half_way_aborts_the_waiting_kernel(gpu_api, tmp_path):
    """INFERA_STREAM_ABORT_INJECT=3 (test hook): the third streamed chunk's gather throws after half of its column groups.  The kernel that is
    already waiting for the other half is told to leave (abort flags), the call fails with the error text, and chunks four to six stream
    normally on the same staging context; with INFERA_STREAM_MAX_INFLIGHT=0 nothing is streamed at all."""
    env = dict(os.environ, INFERA_STREAM_ABORT_INJECT="3", INFERA_LOG_LEVEL="ERROR")
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tmp": str(tmp_path)}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert r["ok"] == 5 and len(r["errors"]) == 1 and r["errors"][0][0] == 2 and "injected gather failure" in r["errors"][0][1], r
    assert r["streamed"] == 5
    env = dict(os.environ, INFERA_STREAM_MAX_INFLIGHT="0")
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tmp": str(tmp_path)}], env=env, capture_output=True, text=True, timeout=600)
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert r["ok"] == 6 and r["errors"] == [] and r["streamed"] == 0, r
