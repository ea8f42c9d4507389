"""Every host-path variant produces the same bits: the default chain (column-major chunk read by the fused kernel, results
stored straight into pinned memory, blocking wait), the round-1 chain (GPU transpose + D2H copy + spinning wait), and the
hipGraph replay of either -- for the fused MLP (C2) and for a plan without those shortcuts (Dense+Softmax, C4)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json, hashlib
sys.path.insert(0, %(root)r)
import numpy as np
from infera_amd import capi, onnx_writer as W, synth
out = {}
for name, blob in (("mlp", W.mlp((128, 256, 64, 1))), ("logreg", W.logreg_softmax(128, 10))):
    capi.load_model(name, W.write(os.path.join(%(tmp)r, name + ".onnx"), blob))
    h = hashlib.sha256()
    for rows in (2048, 777, 1):
        x = synth.table(5, 100, rows, 128)
        cols = [np.ascontiguousarray(x[:, j]) for j in range(128)]
        a = capi.predict_columns(name, cols)      # columnar entry (what the DuckDB binding calls)
        b = capi.predict(name, x)                 # row-major entry
        assert np.array_equal(a, b)
        h.update(a.tobytes())
    out[name] = h.hexdigest()
print("RESULT " + json.dumps(out))
"""

VARIANTS = {
    "default": {},
    "round1_chain": {"INFERA_HOST_WAIT": "spin", "INFERA_HOST_DIRECT_OUT": "0", "INFERA_HOST_FUSED_TRANSPOSE": "0"},
    "hipgraph": {"INFERA_HIPGRAPH": "1"},
    "hipgraph_round1_chain": {"INFERA_HIPGRAPH": "1", "INFERA_HOST_DIRECT_OUT": "0", "INFERA_HOST_FUSED_TRANSPOSE": "0"},
}


@pytest.mark.gpu
def test_host_path_variants_are_bit_identical(gpu_api, tmp_path):
    got = {}
    for tag, extra in VARIANTS.items():
        env = {k: v for k, v in os.environ.items() if not k.startswith("INFERA_HOST_") and k != "INFERA_HIPGRAPH"}
        env.update(extra)
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tmp": str(tmp_path)}], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (tag, p.stderr[-2000:])
        got[tag] = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for tag in VARIANTS:
        assert got[tag] == got["default"], tag
