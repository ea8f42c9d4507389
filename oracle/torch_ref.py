"""PyTorch-CPU evaluations of the bench models.  TEST INFRASTRUCTURE ONLY (same rule as oracle.py: tests/ and
bench.py's cpu_baseline leg import this; nothing under infera_amd/ does).

Two uses:
  * tests/test_oracle_vs_torch.py cross-checks the C oracle against torch's operators in fp32 / fp64 -- an implementation
    that is independent of both this repository and the ONNX text the oracle was restated from;
  * bench.py times the same graphs on the host cores as the CPU leg this repository did NOT write: T worker
    threads, one intra-op thread each, 2048-row chunks -- the execution shape the reference gets from DuckDB's workers
    around a single-threaded Tract run (engine.rs:140-152), with oneDNN / MKL standing in for Tract's packed SIMD kernels.

Graphs are rebuilt from infera_amd.onnx_writer's weight stream, draw for draw, so they are the models the GPU path loads.
"""
from __future__ import annotations

import threading
import time

import numpy as np
import torch
import torch.nn.functional as F

from infera_amd import onnx_writer as W


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def mlp_layers(dims=(128, 256, 64, 1), dtype=torch.float32, seed: int = 1234):
    """[(weight [out, in], bias [out])] of onnx_writer.mlp(dims)."""
    ws = W._WeightStream(seed)
    out = []
    for k, m in zip(dims[:-1], dims[1:]):
        w, b = ws.take((k, m), k), ws.take((m,), k)
        out.append((_t(w.T, dtype), _t(b, dtype)))
    return out


def mlp_forward(layers, x, final_softmax: bool = False):
    h = x
    for i, (w, b) in enumerate(layers):
        h = F.linear(h, w, b)
        if i + 1 < len(layers):
            h = F.relu(h)
    return F.softmax(h, dim=1) if final_softmax else h


def resnet18_params(classes: int = 1000, width: int = 64, dtype=torch.float32, seed: int = 1234, fold_bn: bool = False):
    """Parameters of onnx_writer.resnet18 in draw order.  fold_bn: BatchNorm folded into the convolution (what any inference
    runtime does at load time; the timing leg uses it, the parity tests keep the separate operator)."""
    ws = W._WeightStream(seed)

    def conv_bn(cin, cout, k):
        w = ws.take((cout, cin, k, k), cin * k * k)
        scale = (1.0 + 0.1 * ws.take((cout,), 1)).astype(np.float32)
        beta = (0.1 * ws.take((cout,), 1)).astype(np.float32)
        mean = (0.1 * ws.take((cout,), 1)).astype(np.float32)
        var = (1.0 + 0.5 * np.abs(ws.take((cout,), 1))).astype(np.float32)
        if fold_bn:
            s = (scale.astype(np.float64) / np.sqrt(var.astype(np.float64) + 1e-5))
            wf = (w.astype(np.float64) * s[:, None, None, None]).astype(np.float32)
            bf = (beta.astype(np.float64) - mean.astype(np.float64) * s).astype(np.float32)
            return {"w": _t(wf, dtype), "b": _t(bf, dtype)}
        return {"w": _t(w, dtype), "bn": tuple(_t(a, dtype) for a in (mean, var, scale, beta))}

    p = {"stem": conv_bn(3, width, 7), "blocks": []}
    cin = width
    for stage, cout in enumerate([width, width * 2, width * 4, width * 8]):
        for blk in range(2):
            stride = 2 if (stage > 0 and blk == 0) else 1
            b = {"stride": stride, "c1": conv_bn(cin, cout, 3), "c2": conv_bn(cout, cout, 3)}
            if stride != 1 or cin != cout:
                b["ds"] = conv_bn(cin, cout, 1)
            p["blocks"].append(b)
            cin = cout
    p["fc_w"] = _t(ws.take((cin, classes), cin), dtype)
    p["fc_b"] = _t(ws.take((classes,), cin), dtype)
    return p


def _conv(x, c, stride, pad, relu):
    if "bn" in c:
        mean, var, scale, beta = c["bn"]
        y = F.batch_norm(F.conv2d(x, c["w"], None, stride, pad), mean, var, scale, beta, training=False, eps=1e-5)
    else:
        y = F.conv2d(x, c["w"], c["b"], stride, pad)
    return F.relu(y) if relu else y


def resnet18_forward(p, x):
    x = F.max_pool2d(_conv(x, p["stem"], 2, 3, True), 3, 2, 1)
    for b in p["blocks"]:
        y = _conv(_conv(x, b["c1"], b["stride"], 1, True), b["c2"], 1, 1, False)
        sc = _conv(x, b["ds"], b["stride"], 0, False) if "ds" in b else x
        x = F.relu(y + sc)
    return x.mean(dim=(2, 3)) @ p["fc_w"] + p["fc_b"]


# ---- timing legs (bench.py cpu_baseline.torch_cpu) ---------------------------------------------------------------------------

def _run_threads(threads: int, body) -> float:
    """`threads` Python threads pulling work items from a shared counter; torch releases the GIL inside its operators."""
    torch.set_num_threads(1)  # one intra-op thread per call: parallelism comes from the worker threads, as in the reference
    lock, nxt = threading.Lock(), [0]

    def worker():
        with torch.no_grad():
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if not body(i):
                    return

    th = [threading.Thread(target=worker) for _ in range(threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0


def scan_table(table: np.ndarray, rows: int, cols: int, row_group: int, threads: int, dims, final_softmax: bool = False, chunk: int = 2048):
    """Scans the first `rows` rows of the columnar host table (sqlharness.synth_table layout: row groups of `row_group` rows, one
    contiguous run per column inside a group) in `chunk`-row chunks: gather to a row-major [n, cols] tensor, Linear/ReLU chain.
    Returns (seconds, checksum)."""
    layers = mlp_layers(dims)
    chunks = []
    for g0 in range(0, rows, row_group):
        gr = min(row_group, rows - g0)
        seg = torch.from_numpy(table[g0 * cols: g0 * cols + cols * gr].reshape(cols, gr))
        chunks += [(seg, o, min(chunk, gr - o)) for o in range(0, gr, chunk)]
    sums = [0.0] * len(chunks)

    def body(i):
        if i >= len(chunks):
            return False
        seg, o, n = chunks[i]
        x = seg[:, o:o + n].t().contiguous()  # the gather: 128 column runs -> row-major features
        sums[i] = float(mlp_forward(layers, x, final_softmax).sum())
        return True

    sec = _run_threads(threads, body)
    return sec, float(sum(sums))


def scan_images(images: np.ndarray, rows: int, hw: int, threads: int, batch: int = 8, classes: int = 1000, width: int = 64):
    """`rows` inferences of the ResNet-18 topology over host images (cycling), `batch` images per call. Returns (seconds, checksum)."""
    p = resnet18_params(classes, width, fold_bn=True)
    imgs = torch.from_numpy(images.reshape(-1, 3, hw, hw))
    nimg = imgs.shape[0]
    ncalls = (rows + batch - 1) // batch
    sums = [0.0] * ncalls

    def body(i):
        if i >= ncalls:
            return False
        idx = [(i * batch + j) % nimg for j in range(min(batch, rows - i * batch))]
        sums[i] = float(resnet18_forward(p, imgs[idx]).sum())
        return True

    sec = _run_threads(threads, body)
    return sec, float(sum(sums))
