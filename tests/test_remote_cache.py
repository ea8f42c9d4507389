"""Remote model fetch + LRU disk cache (reference infera/src/http.rs), over loopback.

Mirrors the reference's own tests (http.rs:346-628, written there against mockito / tiny_http servers on 127.0.0.1):
download + cache hit without a second request, ETag revalidation (304) and replacement (200 with a new ETag), servers
without ETag, a 5xx answer, a body shorter than its Content-Length, a dropped connection -- none of which may leave
a `.part` or final file behind -- plus LRU eviction under INFERA_CACHE_SIZE_LIMIT, chunked bodies, redirects, and the
C-ABI surface (infera_load_model on an http:// path, cache info / clear).  The library reads its configuration once,
so the scenario runs in a child process with its own INFERA_CACHE_DIR and short retry delays.
"""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, http.server, json, os, socket, sys, threading, time
sys.path.insert(0, ROOT)
from infera_amd import capi

GOLD = os.path.join(ROOT, "tests", "golden")
LINEAR = open(os.path.join(GOLD, "linear.onnx"), "rb").read()
IDENT = open(os.path.join(GOLD, "multi_output.onnx"), "rb").read()
CACHE = os.environ["INFERA_CACHE_DIR"]
hits = {}
import numpy as np
from infera_amd import onnx_writer as W
def _big():  # a valid model of exactly 4096 bytes (padding lives in producer_name)
    mk = lambda pad: W.model("Lin", [W.node("MatMul", ["X", "W"], ["Y"])], [W.tensor("W", np.ones((3, 1), np.float32))],
                             [W.value_info("X", ["N", 3])], [W.value_info("Y", ["N", 1])], producer="p" * pad)
    pad = 4096 - len(mk(0))
    while len(mk(pad)) != 4096: pad -= len(mk(pad)) - 4096
    return mk(pad)
BIG = _big()

class H(http.server.BaseHTTPRequestHandler):
    protocol_version = "HTTP/1.1"
    def log_message(self, *a): pass
    def do_GET(self):
        hits[self.path] = hits.get(self.path, 0) + 1
        inm = self.headers.get("If-None-Match")
        def send(status, body=b"", headers=()):
            self.send_response(status)
            for k, v in headers: self.send_header(k, v)
            if not any(k.lower() in ("content-length", "transfer-encoding") for k, _ in headers):
                self.send_header("Content-Length", str(len(body)))
            self.send_header("Connection", "close")
            self.end_headers()
            self.wfile.write(body)
        p = self.path
        if p == "/ok.onnx": send(200, LINEAR)
        elif p == "/etag304.onnx":
            send(304, headers=[("ETag", "tag1")]) if inm == "tag1" else send(200, LINEAR, [("ETag", "tag1")])
        elif p == "/etag200.onnx":
            send(200, IDENT, [("ETag", "tag2")]) if inm == "tag1" else send(200, LINEAR, [("ETag", "tag1")])
        elif p == "/err500.onnx": send(500, b"boom")
        elif p == "/short.onnx": send(200, b"incomplete data", [("Content-Length", "100")])
        elif p == "/drop.onnx":
            self.send_response(200); self.send_header("Content-Length", "1024"); self.end_headers()
            self.wfile.flush(); self.connection.shutdown(socket.SHUT_RDWR)
        elif p == "/chunked.onnx":
            self.send_response(200); self.send_header("Transfer-Encoding", "chunked"); self.send_header("Connection", "close"); self.end_headers()
            for i in range(0, len(LINEAR), 37):
                c = LINEAR[i:i + 37]
                self.wfile.write(b"%x\r\n" % len(c) + c + b"\r\n")
            self.wfile.write(b"0\r\n\r\n")
        elif p in ("/badchunk.onnx", "/emptychunk.onnx"):  # a chunk-size line that is not hex / is empty: truncated body, not "end of body"
            self.send_response(200); self.send_header("Transfer-Encoding", "chunked"); self.send_header("Connection", "close"); self.end_headers()
            self.wfile.write(b"%x\r\n" % 37 + LINEAR[:37] + b"\r\n")
            self.wfile.write(b"zz\r\n\r\n" if p == "/badchunk.onnx" else b"\r\n\r\n")
        elif p.startswith("/bigtag"): send(200, BIG, [("ETag", "t" + p[7])])
        elif p == "/redir.onnx": send(302, headers=[("Location", "/ok.onnx")])
        elif p.startswith("/big"): send(200, BIG)
        else: send(404, b"nope")

srv = http.server.ThreadingHTTPServer(("127.0.0.1", 0), H)
port = srv.server_address[1]
threading.Thread(target=srv.serve_forever, daemon=True).start()
base = f"http://127.0.0.1:{port}"
def key(url): return hashlib.sha256(url.encode()).hexdigest()
def cached(url): return os.path.join(CACHE, key(url) + ".onnx")
def leftovers(): return sorted(f for f in os.listdir(CACHE) if f.endswith(".part")) if os.path.isdir(CACHE) else []
out = {}

# success + cache hit without a second request (http.rs:461-483, :575-613)
u = base + "/ok.onnx"
capi.load_model("r1", u)
out["ok_info"] = capi.get_model_info("r1")["input_shape"]
out["ok_cached_equals"] = open(cached(u), "rb").read() == LINEAR
out["ok_no_etag_file"] = not os.path.exists(os.path.join(CACHE, key(u) + ".etag"))
capi.load_model("r1b", u)
out["ok_hits"] = hits["/ok.onnx"]

# ETag verified -> 304 (http.rs:485-528)
u = base + "/etag304.onnx"
capi.load_model("r2", u)
out["etag_file"] = open(os.path.join(CACHE, key(u) + ".etag")).read().strip()
capi.load_model("r2b", u)
out["etag304_hits"] = hits["/etag304.onnx"]
out["etag304_still_linear"] = open(cached(u), "rb").read() == LINEAR

# ETag changed -> 200 + new body + new tag (http.rs:530-573)
u = base + "/etag200.onnx"
capi.load_model("r3", u)
out["etag200_first"] = capi.get_model_info("r3")["input_shape"]
capi.load_model("r3", u)
out["etag200_second"] = capi.get_model_info("r3")["input_shape"]
out["etag200_tag"] = open(os.path.join(CACHE, key(u) + ".etag")).read().strip()

# failures leave nothing behind (http.rs:346-459)
for name in ("err500", "short", "drop", "missing", "badchunk", "emptychunk"):
    u = base + f"/{name}.onnx"
    try:
        capi.load_model("bad", u)
        out[name] = "loaded?!"
    except capi.InferaError as e:
        out[name] = str(e)
    out[name + "_clean"] = not os.path.exists(cached(u)) and leftovers() == []
out["err500_attempts"] = hits["/err500.onnx"]

# chunked body, redirect
capi.load_model("r4", base + "/chunked.onnx")
out["chunked_ok"] = open(cached(base + "/chunked.onnx"), "rb").read() == LINEAR
capi.load_model("r5", base + "/redir.onnx")
out["redir_ok"] = open(cached(base + "/redir.onnx"), "rb").read() == LINEAR

# a URL fragment (ADVICE r2): never sent to the server, the cache key is sha256 of the URL AS WRITTEN (the reference's key,
# http.rs:187-190), and a fragment that names no output of the model leaves output 0 served, as the reference would
u = base + "/ok.onnx#v2"
capi.load_model("frag", u)
out["frag_info"] = capi.get_model_info("frag")["input_shape"]
out["frag_cached_under_full_url"] = os.path.exists(cached(u))
out["frag_request_path_hits"] = hits.get("/ok.onnx#v2", 0)
out["frag_ok_hits"] = hits["/ok.onnx"]
# ... and one that DOES name an output selects it (multi_output.onnx has the single output "Y": index 0 and the name both work)
capi.load_model("frag_sel", base + "/etag200.onnx#0")
out["frag_sel_ok"] = capi.get_model_info("frag_sel")["loaded"]

# https is refused, not attempted
try:
    capi.load_model("tls", "https://127.0.0.1:1/x.onnx"); out["https"] = "loaded?!"
except capi.InferaError as e:
    out["https"] = str(e)

# cache info counts *.onnx files; LRU eviction under the 10000-byte limit (http.rs:98-120): 4096-byte files
out["info_before"] = capi.get_cache_info()
capi.clear_cache()
out["info_cleared"] = capi.get_cache_info()
urls = [base + f"/big{i}.onnx" for i in range(3)]
for i in (0, 1):                                  # two 4096-byte files fit under the 10000-byte limit
    capi.load_model(f"big{i}", urls[i])
    os.utime(cached(urls[i]), (1000 + i, 1000 + i))  # deterministic access order: big0 oldest
capi.load_model("big0again", urls[0])            # cache hit (no request) refreshes big0's access time -> big1 is now the oldest
out["present_before_evict"] = [os.path.exists(cached(u)) for u in urls]
capi.load_model("big2", urls[2])                 # 8192 + 4096 > 10000 -> evict least recently ACCESSED until it fits
out["present_after_evict"] = [os.path.exists(cached(u)) for u in urls]
out["big0_hits"] = hits["/big0.onnx"]
out["info_after"] = capi.get_cache_info()
# eviction takes the entry's .etag along (no orphan revalidation tags)
capi.clear_cache()
turls = [base + f"/bigtag{i}.onnx" for i in range(3)]
for i in range(3):
    capi.load_model(f"bt{i}", turls[i])
    os.utime(cached(turls[i]), (2000 + i, 2000 + i))
out["etag_files_after_evict"] = sorted(f for f in os.listdir(CACHE) if f.endswith(".etag"))
out["etag_expected"] = sorted(key(u) + ".etag" for u in turls[1:])
os.makedirs(os.path.join(CACHE, "subdir"), exist_ok=True)
open(os.path.join(CACHE, "subdir", "x.tmp"), "w").write("x")
capi.clear_cache()
out["cleared_listing"] = os.listdir(CACHE)
print("RESULT " + json.dumps(out))
'''


import pytest


def _have_libcurl():
    import ctypes

    for n in ("libcurl.so.4", "libcurl-gnutls.so.4", "libcurl.so"):
        try:
            ctypes.CDLL(n)
            return True
        except OSError:
            pass
    return False


@pytest.mark.parametrize("backend", ["socket", "curl"])
def test_remote_fetch_and_lru_cache(built, tmp_path, backend):
    """backend = socket: the built-in HTTP/1.1 client; curl: the dlopen'd libcurl path that serves https://."""
    if backend == "curl" and not _have_libcurl():
        pytest.skip("libcurl is not loadable here")
    cache = tmp_path / "cache"
    env = dict(os.environ, INFERA_CACHE_DIR=str(cache), INFERA_CACHE_SIZE_LIMIT="10000", INFERA_HTTP_RETRY_ATTEMPTS="2",
               INFERA_HTTP_RETRY_DELAY="10", INFERA_HTTP_TIMEOUT="5", INFERA_LOG_LEVEL="ERROR", INFERA_HTTP_BACKEND=backend)
    script = f"ROOT = {ROOT!r}\n" + textwrap.dedent(CHILD)
    p = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out["ok_info"] == [1, 3] and out["ok_cached_equals"] and out["ok_no_etag_file"]
    assert out["ok_hits"] == 1  # second load: cache hit, no request
    assert out["etag_file"] == "tag1" and out["etag304_hits"] == 2 and out["etag304_still_linear"]
    assert out["etag200_first"] == [1, 3] and out["etag200_second"] == [1, 4] and out["etag200_tag"] == "tag2"
    assert out["err500"].startswith("HTTP request failed: HTTP status server error (500") and out["err500_clean"]
    assert out["err500_attempts"] == 2  # INFERA_HTTP_RETRY_ATTEMPTS
    assert out["short"] == "IO error: unexpected end of file" and out["short_clean"]
    assert (out["drop"].startswith("IO error: ") or out["drop"].startswith("HTTP request failed: ")) and out["drop_clean"]
    assert out["missing"].startswith("HTTP request failed: HTTP status client error (404") and out["missing_clean"]
    assert out["chunked_ok"] and out["redir_ok"]
    assert out["frag_info"] == [1, 3] and out["frag_cached_under_full_url"] and out["frag_request_path_hits"] == 0 and out["frag_ok_hits"] == 3  # r1, the /redir.onnx target, and this one (a new cache key: its own fetch)
    assert out["frag_sel_ok"] is True
    for name in ("badchunk", "emptychunk"):  # a malformed chunk-size line is a failed download, never a cached truncated file
        assert (out[name].startswith("IO error: ") or out[name].startswith("HTTP request failed: ")) and out[name + "_clean"], out[name]
    assert out["etag_files_after_evict"] == out["etag_expected"]
    assert out["https"].startswith("HTTP request failed: ")  # refused by the socket client / connection refused through curl
    assert out["info_before"]["cache_dir"] == str(cache) and out["info_before"]["file_count"] >= 5
    assert out["info_before"]["size_limit_bytes"] == 10000
    assert out["info_cleared"]["file_count"] == 0 and out["info_cleared"]["total_size_bytes"] == 0
    assert out["present_before_evict"] == [True, True, False]
    assert out["present_after_evict"] == [True, False, True]  # big1 went: big0 had been re-accessed (and not re-fetched)
    assert out["big0_hits"] == 1
    assert out["info_after"]["file_count"] == 2 and out["info_after"]["total_size_bytes"] == 8192
    assert out["cleared_listing"] == []


def test_sha256_cache_key(built):
    """The cache key is sha256(url) in hex (http.rs:186-190); pin the digest implementation on FIPS 180-4 vectors
    through a file that a download creates -- here: directly against hashlib for a spread of lengths."""
    import ctypes
    import hashlib

    from infera_amd import capi

    L = capi.load_library()
    if not hasattr(L, "infera_hip_sha256_hex"):
        import pytest

        pytest.skip("helper not exported")
    L.infera_hip_sha256_hex.restype = ctypes.c_void_p
    L.infera_hip_sha256_hex.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    for n in (0, 1, 3, 55, 56, 57, 63, 64, 65, 119, 120, 1000):
        data = bytes((i * 7 + 3) & 0xFF or 1 for i in range(n))
        ptr = L.infera_hip_sha256_hex(data, n)
        got = ctypes.string_at(ptr).decode()
        L.infera_free(ctypes.c_char_p(ptr) if False else ctypes.cast(ptr, ctypes.c_char_p))
        assert got == hashlib.sha256(data).hexdigest(), n
