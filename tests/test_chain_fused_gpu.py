"""The load-time specialised chain kernel (csrc/hip/chain_device.inc): small MLPs over tables of any width, fused into one
pass -- against the oracle, with the layer-by-layer kernels (INFERA_FUSED_MLP=0 in a child process) as a second opinion."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth

RTOL, ATOL = 1e-4, 1e-6
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assert_close(got, want, rtol=RTOL, atol=ATOL):
    assert got.shape == want.shape, (got.shape, want.shape)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    bad = err > rtol * np.abs(want.astype(np.float64)) + atol
    assert not bad.any(), f"{bad.sum()} / {bad.size} out of tolerance; worst err {err.max():.3e}"


@pytest.fixture(scope="module")
def O(built):
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def api(built):
    from infera_amd import capi

    assert capi.device_count() >= 1, capi.get_devices()
    return capi


# dims, activations, final softmax
CHAINS = [
    ((4, 10, 3), None, True),                 # iris-sized MLPClassifier
    ((30, 100, 2), None, True),               # sklearn's default hidden layer on 30 features
    ((13, 64, 32, 1), None, False),
    ((20, 16, 1), ["Tanh", "Sigmoid"], False),
    ((30, 8, 1), ["LeakyRelu", ""], False),
    ((128, 128, 128, 16), ["Relu", "Sigmoid", ""], False),
    ((1, 5, 1), None, False),
    ((3, 3, 3, 3, 3), None, True),
    ((100, 100, 100, 10), None, True),
    ((77, 33, 17), ["Relu", "Relu"], False),  # 17 outputs: two output tiles, odd row length on the way out
    ((64, 48, 20), None, False),
    ((30, 100), None, False),                 # one wide layer behind a PadCols: fused to skip the padding pass
    ((100, 20), None, False),                 # 17..32 outputs over rows no aligned kernel reads: one-layer chain
    ((30, 100), None, True),                  # wide softmax heads: scores never leave the registers
    ((64, 50), None, True),
    ((13, 20, 128), None, True),
    ((50, 12, 12), None, True),
]


def _name(c):
    return "x".join(map(str, c[0])) + ("+sm" if c[2] else "")


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 33, 4099])
@pytest.mark.parametrize("case", CHAINS, ids=_name)
def test_gpu_chain_vs_oracle(api, O, tmp_path, case, rows):
    dims, acts, sm = case
    path = W.write(str(tmp_path / "c.onnx"), W.mlp(dims, acts=acts, final_softmax=sm, seed=77))
    x = synth.table(3, 0, rows, dims[0])
    api.load_model("chain", path)
    try:
        plan = api.get_plan("chain")
        if max(dims) < 100 or len(dims) <= 3:  # the widest chains exceed the LDS in one piece (or belong to the 32x32 kernel)
            assert plan["exec"][0] == "chain_fused" and set(plan["exec"][1:]) <= {"skipped"}, plan["exec"]
            assert plan["chain_kernels"][0].startswith("chain_kernel<" + "x".join(map(str, dims))), plan["chain_kernels"]
        else:
            assert {"chain_fused", "mlp3_fused"} & set(plan["exec"]), plan["exec"]
        got = api.predict("chain", x)
    finally:
        api.unload_model("chain")
    assert_close(got, O.Model(path).predict(x))


@pytest.mark.gpu
def test_gpu_chain_label_output_and_batch_invariance(api, O, tmp_path):
    """classifier head: MLP -> ArgMax as the served output; and the same rows in one big scan or many small ones"""
    ws = W._WeightStream(5)
    k, h, e = 30, 24, 5
    w1, b1, w2, b2 = ws.take((k, h), k), ws.take((h,), k), ws.take((h, e), h), ws.take((e,), h)
    nodes = [W.node("Gemm", ["X", "w1", "b1"], ["h"]), W.node("Relu", ["h"], ["a"]), W.node("Gemm", ["a", "w2", "b2"], ["s"]),
             W.node("ArgMax", ["s"], ["label"], [W.attr_i("axis", 1), W.attr_i("keepdims", 0)])]
    inits = [W.tensor("w1", w1), W.tensor("b1", b1), W.tensor("w2", w2), W.tensor("b2", b2)]
    path = W.write(str(tmp_path / "lab.onnx"),
                   W.model("lab", nodes, inits, [W.value_info("X", ["N", k])], [W.value_info("label", ["N"], W.INT64)]))
    x = synth.table(8, 0, 20000, k)
    raw = np.maximum(x.astype(np.float64) @ w1 + b1, 0) @ w2.astype(np.float64) + b2
    srt = np.sort(raw, axis=1)
    ok = (srt[:, -1] - srt[:, -2]) > 1e-4
    api.load_model("lab", path)
    try:
        plan = api.get_plan("lab")
        assert plan["exec"] == ["chain_fused", "skipped", "skipped"] and plan["chain_kernels"][0].endswith("+argmax> [hipRTC]"), plan
        whole = api.predict("lab", x).reshape(-1)
        pieces = np.concatenate([api.predict("lab", x[i:i + 777]).reshape(-1) for i in range(0, len(x), 777)])
    finally:
        api.unload_model("lab")
    assert np.array_equal(whole, pieces)
    assert np.array_equal(whole[ok], np.argmax(raw, axis=1)[ok].astype(np.float32))
    assert np.array_equal(whole[ok], O.Model(path).predict(x).reshape(-1)[ok])


@pytest.mark.gpu
def test_gpu_chain_agrees_with_layer_by_layer_kernels(api, tmp_path):
    """same model, same rows, INFERA_FUSED_MLP=0 in a child process: the ahead-of-time kernels, one layer per pass"""
    dims = (30, 100, 3)
    path = W.write(str(tmp_path / "c.onnx"), W.mlp(dims, final_softmax=True, seed=5))
    x = synth.table(4, 0, 5000, dims[0])
    api.load_model("c", path)
    try:
        fused = api.predict("c", x)
    finally:
        api.unload_model("c")
    np.save(tmp_path / "x.npy", x)
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); from infera_amd import capi; "
            "capi.load_model('c', %r); x = np.load(%r); y = capi.predict('c', x); np.save(%r, y); "
            "print(json.dumps(capi.get_plan('c')['exec']))") % (ROOT, path, str(tmp_path / "x.npy"), str(tmp_path / "y.npy"))
    env = dict(os.environ, INFERA_FUSED_MLP="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "chain_fused" not in json.loads(out.stdout.strip().splitlines()[-1])
    assert_close(fused, np.load(tmp_path / "y.npy"), rtol=2e-5, atol=1e-6)


# dims, final softmax: one model per streaming kernel (skinny / any-width 16x16x4 / 64-128-column 16x16x4 / chain)
UNALIGNED = [((13, 1), False), ((30, 2), True), ((30, 8), False), ((77, 5), True), ((100, 16), False), ((64, 10), True),
             ((128, 3), False), ((201, 1), False), ((300, 10), True), ((561, 6), True), ((1000, 2), False), ((4, 10, 3), True), ((30, 100, 2), True), ((50, 24, 12), False), ((30, 100), False)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", UNALIGNED, ids=lambda c: "x".join(map(str, c[0])))
@pytest.mark.parametrize("rows", [5, 1000, 4133])
def test_gpu_device_scan_from_unaligned_pointers(api, O, tmp_path, case, rows):
    """infera_hip_predict_device on a table that starts 4 / 8 / 12 bytes past a 16-byte boundary (a column range of a
    bigger allocation): the streaming kernels take their element-wise load path, results are the same"""
    dims, sm = case
    path = W.write(str(tmp_path / "u.onnx"), W.mlp(dims, final_softmax=sm, seed=11))
    x = synth.table(6, 0, rows, dims[0])
    want = O.Model(path).predict(x)
    dev = api.device_ordinal(0)
    api.load_model("u", path)
    try:
        for off_in, off_out in [(4, 0), (8, 4), (12, 12), (0, 8)]:
            d_in = api.DeviceBuffer(dev, x.nbytes + 16)
            d_out = api.DeviceBuffer(dev, want.nbytes + 16)
            d_in.upload(np.concatenate([np.zeros(off_in // 4, np.float32), x.ravel()]))
            r, c = api.predict_device("u", d_in, rows, dims[0], d_out, in_offset_bytes=off_in, out_offset_bytes=off_out)
            assert (r, c) == want.shape
            got = d_out.download(want.shape, offset_bytes=off_out)
            assert_close(got, want)
            d_in.free()
            d_out.free()
    finally:
        api.unload_model("u")


@pytest.mark.gpu
def test_gpu_jit_code_objects_are_cached_on_disk(tmp_path):
    """a second process loading the same chain shape finds the code object under INFERA_JIT_CACHE_DIR (no recompile);
    a truncated cache file is ignored and rewritten"""
    path = W.write(str(tmp_path / "c.onnx"), W.mlp((9, 20, 4), final_softmax=True, seed=3))
    cache = tmp_path / "jit"
    code = ("import sys, time, json; sys.path.insert(0, %r); from infera_amd import capi; t = time.perf_counter(); "
            "capi.load_model('c', %r); dt = time.perf_counter() - t; "
            "import numpy as np; y = capi.predict('c', np.ones((3, 9), np.float32)); "
            "print(json.dumps({'load_s': dt, 'exec': capi.get_plan('c')['exec'], 'y': y.tolist()}))") % (ROOT, path)
    env = dict(os.environ, INFERA_JIT_CACHE_DIR=str(cache))

    def run():
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    first = run()
    files = sorted(cache.glob("*.hsaco"))
    assert first["exec"][0] == "chain_fused" and len(files) == 1 and files[0].stat().st_size > 1000
    stamp = files[0].stat().st_mtime_ns
    second = run()
    assert second["y"] == first["y"] and second["exec"] == first["exec"]
    # found, not recompiled: the cache file was not written again.  (The load times are printed, not compared: both runs carry a process's
    # GPU start-up, whose run-to-run noise on a cold box is larger than the ~0.2 s compile the cache saves.)
    assert sorted(cache.glob("*.hsaco")) == files and files[0].stat().st_mtime_ns == stamp, (first["load_s"], second["load_s"])
    files[0].write_bytes(files[0].read_bytes()[:500])  # corrupt: must be ignored, recompiled and replaced
    third = run()
    assert third["y"] == first["y"] and sorted(cache.glob("*.hsaco"))[0].stat().st_size > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("k,m,sm", [(129, 1, False), (201, 1, False), (300, 10, True), (561, 6, True), (1000, 16, False), (1984, 3, True),
                                    (2048, 10, True), (2049, 3, False), (3000, 5, True), (4096, 16, False),
                                    (300, 20, True), (561, 32, False), (1000, 26, True), (2048, 17, True), (4099, 32, False),
                                    (561, 50, True), (300, 64, False), (2048, 64, True), (1001, 33, False)])
@pytest.mark.parametrize("rows", [1, 31, 32, 33, 4100])
def test_gpu_wide_tables_of_any_row_length(api, O, tmp_path, k, m, sm, rows):
    """dense_narrow16w_kernel: 64-column chunks row by row, K not a multiple of anything, partial last chunk and tile;
    from ~1500 columns on the weights live in the LDS one 1024-column window at a time"""
    path = W.write(str(tmp_path / "w.onnx"), W.mlp((k, m), final_softmax=sm, seed=13))
    x = synth.table(12, 0, rows, k)
    api.load_model("w", path)
    try:
        got = api.predict("w", x)
    finally:
        api.unload_model("w")
    assert_close(got, O.Model(path).predict(x))


# one model per compute kernel family: a poisoned row (NaN / Inf features) must stay that row's problem
ISOLATION = [((128, 256, 64, 1), False), ((128, 10), True), ((30, 100, 2), True), ((4, 10, 3), True), ((13, 1), False),
             ((30, 8), False), ((77, 5), True), ((300, 10), True), ((64, 96, 1), False), ((200, 160, 40), False), ((561, 6), False)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ISOLATION, ids=lambda c: "x".join(map(str, c[0])))
def test_gpu_rows_are_independent_even_when_one_is_poisoned(api, tmp_path, case):
    """table rows are independent records: NaN / Inf in one row may not leak into its neighbours through tile padding,
    zero-weight lanes or shared LDS staging (0 * NaN is NaN) -- every other row's result is bit-identical"""
    dims, sm = case
    path = W.write(str(tmp_path / "iso.onnx"), W.mlp(dims, final_softmax=sm, seed=19))
    rows = 2500
    x = synth.table(31, 0, rows, dims[0])
    bad = x.copy()
    poisoned = [0, 31, 32, 1000, 1777, rows - 1]
    for i, r in enumerate(poisoned):
        bad[r, (7 * i) % dims[0]] = [np.nan, np.inf, -np.inf][i % 3]
    api.load_model("iso", path)
    try:
        clean, dirty = api.predict("iso", x), api.predict("iso", bad)
    finally:
        api.unload_model("iso")
    keep = np.ones(rows, bool)
    keep[poisoned] = False
    assert np.isfinite(clean).all()
    assert np.array_equal(clean[keep], dirty[keep]), np.nonzero((clean != dirty).any(axis=1) & keep)[0][:10]


@pytest.mark.gpu
def test_gpu_images_are_independent_even_when_one_is_poisoned(api, tmp_path):
    blob, _ = W.zoo_ops_net()
    path = W.write(str(tmp_path / "iso_img.onnx"), blob)
    n = 40
    x = synth.table(32, 0, n, 3 * 16 * 16)
    bad = x.copy()
    bad[3, 100], bad[17, 5], bad[n - 1, 700] = np.nan, np.inf, -np.inf
    api.load_model("iso_img", path)
    try:
        clean, dirty = api.predict_from_blob("iso_img", x.tobytes()), api.predict_from_blob("iso_img", bad.tobytes())
    finally:
        api.unload_model("iso_img")
    keep = np.ones(n, bool)
    keep[[3, 17, n - 1]] = False
    assert np.isfinite(clean).all() and np.array_equal(clean[keep], dirty[keep])


@pytest.mark.gpu
def test_gpu_concurrent_load_predict_unload_of_jit_models(api, O, tmp_path):
    """eight threads load / run / unload models that need load-time compiled kernels (five shapes, shared code-object
    cache, per-thread streams): every result matches the oracle's, nothing deadlocks"""
    import threading

    shapes = [((9, 12, 3), True), ((30, 100, 2), True), ((17, 40), False), ((5, 8, 8, 1), False), ((32, 64, 32, 1), False)]
    models = []
    for i, (dims, sm) in enumerate(shapes):
        path = W.write(str(tmp_path / f"cc{i}.onnx"), W.mlp(dims, final_softmax=sm, seed=40 + i))
        x = synth.table(50 + i, 0, 777, dims[0])
        models.append((path, x, O.Model(path).predict(x)))
    errors = []

    def worker(t):
        try:
            for it in range(25):
                path, x, want = models[(t + it) % len(models)]
                name = f"cc_t{t}"
                api.load_model(name, path)
                got = api.predict(name, x)
                api.unload_model(name)
                assert_close(got, want)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=240)
    assert not any(th.is_alive() for th in threads), "a worker is stuck"
    assert not errors, errors[:3]
