"""SURVEY.md 8(f) row 4 on the GPU: a model loaded THROUGH THE REMOTE-FETCH PATH -- `infera_load_model(name, "http://127.0.0.1:<port>/...")`
(reference infera/src/lib.rs:47-51 -> http.rs:179-294: download into the LRU disk cache, then the ordinary load) -- runs on the HIP
backend and matches the oracle: BASELINE's C2 MLP (fused MFMA kernel) and the reference's own `linear.onnx` fixture (golden 1.75).
A second load of the same URL must be served from the cache without a second request, and give the same bits.  Child process: the
library reads INFERA_CACHE_DIR once."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import http.server, os, sys, threading
import numpy as np
sys.path.insert(0, ROOT)
from infera_amd import capi, onnx_writer as W, synth
from oracle import oracle

mlp_path = W.write(os.path.join(os.environ["INFERA_CACHE_DIR"], "..", "mlp_src.onnx"), W.mlp((128, 256, 64, 1)))
FILES = {"/mlp.onnx": open(mlp_path, "rb").read(), "/linear.onnx": open(os.path.join(ROOT, "tests", "golden", "linear.onnx"), "rb").read()}
hits = {}

class H(http.server.BaseHTTPRequestHandler):
    protocol_version = "HTTP/1.1"
    def log_message(self, *a): pass
    def do_GET(self):
        hits[self.path] = hits.get(self.path, 0) + 1
        body = FILES.get(self.path)
        self.send_response(200 if body is not None else 404)
        body = body or b"nope"
        self.send_header("Content-Length", str(len(body)))
        self.send_header("Connection", "close")
        self.end_headers()
        self.wfile.write(body)

srv = http.server.ThreadingHTTPServer(("127.0.0.1", 0), H)
threading.Thread(target=srv.serve_forever, daemon=True).start()
base = f"http://127.0.0.1:{srv.server_address[1]}"
assert capi.device_count() >= 1, capi.get_devices()

# the reference's fixture through http: (1, 2, 3) -> 1.75 (test_core_functionality.test:48-56)
capi.load_model("lin", base + "/linear.onnx")
assert float(capi.predict("lin", np.array([[1, 2, 3]], np.float32))[0, 0]) == 1.75
capi.unload_model("lin")

# C2's MLP through http, on the fused HIP kernel, against the oracle
capi.load_model("mlp", base + "/mlp.onnx")
assert capi.get_plan("mlp")["exec"][0] == "mlp3_fused", capi.get_plan("mlp")["exec"]
x = synth.table(42, 0, 4096 + 17, 128)
got = capi.predict("mlp", x)
want = oracle.Model(mlp_path).predict(x)
assert got.shape == want.shape
assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-6), float(np.abs(got - want).max())
capi.unload_model("mlp")
assert hits["/mlp.onnx"] == 1

# second load: served from the disk cache (no ETag -> no revalidation request in the reference either, http.rs:208-215), same bits
capi.load_model("mlp2", base + "/mlp.onnx")
assert hits["/mlp.onnx"] == 1, hits
assert np.array_equal(capi.predict("mlp2", x), got)
info = capi.get_cache_info()
assert info["file_count"] >= 2 and info["total_size_bytes"] >= len(FILES["/mlp.onnx"]), info
capi.unload_model("mlp2")

# a 404 is an error with the reference's text class, and nothing is registered
try:
    capi.load_model("nope", base + "/missing.onnx")
    raise SystemExit("404 did not fail")
except capi.InferaError as exc:
    assert "HTTP request failed" in str(exc), str(exc)
assert capi.get_loaded_models() == []
print("REMOTE-GPU-OK")
'''


def _have_libcurl():
    import ctypes

    for n in ("libcurl.so.4", "libcurl-gnutls.so.4", "libcurl.so"):
        try:
            ctypes.CDLL(n)
            return True
        except OSError:
            pass
    return False


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["socket", "curl"])
def test_gpu_remote_model_runs_on_hip_and_matches_oracle(tmp_path, backend):
    if backend == "curl" and not _have_libcurl():
        pytest.skip("libcurl is not loadable here")
    env = dict(os.environ, INFERA_CACHE_DIR=str(tmp_path / "cache"), INFERA_HTTP_RETRY_ATTEMPTS="2", INFERA_HTTP_RETRY_DELAY="10",
               INFERA_HTTP_BACKEND=backend)
    os.makedirs(env["INFERA_CACHE_DIR"], exist_ok=True)
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "REMOTE-GPU-OK" in r.stdout, r.stdout + r.stderr
