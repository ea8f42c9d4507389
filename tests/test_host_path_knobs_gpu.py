"""Both launch mechanisms of the host path produce the same bits: direct stream enqueues and the hipGraph replay per (model, rows) --
for the fused MLP (C2), for Dense+Softmax (C4), through the columnar and the row-major entry, and with the kernels that read
column-major chunks themselves switched off (GPU transpose in front)."""
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
    "direct": {"INFERA_HIPGRAPH": "0"},
    "hipgraph": {"INFERA_HIPGRAPH": "1"},
    "direct_transposed": {"INFERA_HIPGRAPH": "0", "INFERA_DENSE_XCM": "0", "INFERA_CHAIN_XCM": "0"},
    "hipgraph_transposed": {"INFERA_HIPGRAPH": "1", "INFERA_DENSE_XCM": "0", "INFERA_CHAIN_XCM": "0"},
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
