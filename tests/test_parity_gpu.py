"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerance (BASELINE.json north_star): 1e-4 relative in fp32.  Written here as
|got - want| <= 1e-4 * |want| + 1e-6 -- the absolute floor only matters for outputs within 1e-2 of
zero, where the relative measure of an fp32 dot product of 128..256 O(1) terms is ill-conditioned.
Observed error is ~1e-6 (summation order is the only difference: exact-fp32 MFMA).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-6


def assert_close(got, want):
    assert got.shape == want.shape, (got.shape, want.shape)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    tol = RTOL * np.abs(want.astype(np.float64)) + ATOL
    bad = err > tol
    assert not bad.any(), f"{bad.sum()} / {bad.size} elements out of tolerance; worst err {err.max():.3e}"


@pytest.fixture(scope="module")
def api(built):
    from infera_amd import capi

    assert capi.device_count() >= 1, capi.get_devices()
    return capi


def test_reference_golden_linear(api, models):
    # test/sql/test_core_functionality.test:48-56, docs/examples/e3_integration_and_errors.sql:14-24
    api.load_model("linear", models["linear"])
    assert api.predict("linear", np.array([[1, 2, 3]], np.float32)).tolist() == [[1.75]]
    assert api.predict("linear", np.zeros((1, 3), np.float32)).tolist() == [[0.25]]
    assert api.predict_from_blob("linear", np.zeros(3, np.float32).tobytes()).tolist() == [[0.25]]
    api.unload_model("linear")


def test_reference_golden_multi_output(api, models):
    # test/sql/test_multi_output.test:23-26
    api.load_model("multi_output", models["multi_output"])
    out = api.predict("multi_output", np.array([[1, 2, 3, 4]], np.float32))
    assert out.tolist() == [[1.0, 2.0, 3.0, 4.0]]
    api.unload_model("multi_output")


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 2048, 4096 + 17])
def test_mlp_fused_vs_oracle(api, models, rows):
    from infera_amd import synth
    from oracle import oracle

    api.load_model("mlp", models["mlp"])
    assert api.get_plan("mlp")["exec"][0] == "mlp3_fused"
    x = synth.table(42, 1000, rows, 128)
    assert_close(api.predict("mlp", x), oracle.Model(models["mlp"]).predict(x))


@pytest.mark.parametrize("rows", [1, 5, 2048, 5000])
def test_logreg_softmax_vs_oracle(api, models, rows):
    from infera_amd import synth
    from oracle import oracle

    api.load_model("logreg", models["logreg"])
    x = synth.table(7, 0, rows, 128)
    got = api.predict("logreg", x)
    assert_close(got, oracle.Model(models["logreg"]).predict(x))
    np.testing.assert_allclose(got.sum(axis=1), 1.0, atol=1e-5)


def test_linear_dyn_rows(api, models):
    from oracle import oracle

    api.load_model("lin", models["linear_dyn"])
    x = np.array([[1, 2, 3], [0.5, 1, 1.5], [-1, 0, 2]], np.float32)
    got = api.predict("lin", x)
    assert got.ravel().tolist() == [1.75, 1.0, -0.75]
    assert_close(got, oracle.Model(models["linear_dyn"]).predict(x))


def test_device_path_and_synth_fill(api, models):
    from infera_amd import synth

    api.load_model("mlp", models["mlp"])
    rows = 10_000
    dev = api.device_ordinal(0)
    d_in = api.DeviceBuffer(dev, rows * 128 * 4)
    d_out = api.DeviceBuffer(dev, rows * 4)
    api.synth_fill(d_in, 42, 123, rows, 128)
    x = synth.table(42, 123, rows, 128)
    assert np.array_equal(d_in.download((rows, 128)), x)  # generator is bit-exact across numpy / HIP
    assert api.predict_device("mlp", d_in, rows, 128, d_out) == (rows, 1)
    assert np.array_equal(d_out.download((rows, 1)), api.predict("mlp", x))


def test_committed_golden_vectors_on_gpu(api, tmp_path):
    """HIP path against tests/golden/vectors.npz (oracle outputs frozen by tests/golden/make_golden.py)."""
    import os

    from infera_amd import onnx_writer as W, synth

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    src = os.path.join(gold, "make_golden.py")
    ns = {"__file__": src, "__name__": "golden_cases"}
    exec(compile(open(src).read(), src, "exec"), ns)
    vec = np.load(os.path.join(gold, "vectors.npz"))
    for name, (blob, seed, row0, rows, cols) in ns["CASES"].items():
        api.load_model("g_" + name, W.write(str(tmp_path / (name + ".onnx")), blob))
        assert_close(api.predict("g_" + name, synth.table(seed, row0, rows, cols)), vec[name + "_y"])
        api.unload_model("g_" + name)
    blob = W.resnet18(classes=10, in_hw=32, width=8)
    api.load_model("g_rn", W.write(str(tmp_path / "rn.onnx"), blob))
    x = synth.table(11, 0, 2, 3 * 32 * 32)
    assert_close(api.predict_from_blob("g_rn", x.tobytes()), vec["resnet_small_y"])
    api.unload_model("g_rn")


@pytest.mark.parametrize("dims,acts", [((3, 1), [""]), ((5, 7, 2), ["Relu", ""]), ((130, 33, 70, 9), ["Sigmoid", "Tanh", "LeakyRelu"]),
                                       ((64, 300), ["Relu"]), ((128, 256, 64, 3), ["Relu", "Relu", ""]), ((128, 256, 64, 10), ["Relu", "Relu", ""]), ((17, 64, 64, 64, 5), ["Relu"] * 3 + [""])])
@pytest.mark.parametrize("rows", [1, 129, 1000])
def test_generic_dense_shapes_vs_oracle(api, tmp_path, dims, acts, rows):
    """Ragged K / M (not multiples of 4, 8 or 32), every activation, chains that do not match the
    fused instantiation: the layer-by-layer MFMA kernel."""
    from infera_amd import onnx_writer as W, synth
    from oracle import oracle

    path = W.write(str(tmp_path / "m.onnx"), W.mlp(dims, acts=acts))
    api.load_model("gen", path)
    x = synth.table(9, 77, rows, dims[0])
    assert_close(api.predict("gen", x), oracle.Model(path).predict(x))
    api.unload_model("gen")


def test_matmul_add_and_transb_forms(api, tmp_path):
    from infera_amd import onnx_writer as W, synth
    from oracle import oracle

    x = synth.table(1, 0, 333, 24)
    for kw in ({}, {"use_matmul_add": True}, {"trans_b": True}):
        path = W.write(str(tmp_path / "f.onnx"), W.mlp((24, 8, 5), acts=["Relu", ""], **kw))
        api.load_model("form", path)
        assert_close(api.predict("form", x), oracle.Model(path).predict(x))
    api.unload_model("form")


def test_identity_and_predict_into(api, models):
    api.load_model("idn", models["identity_dyn"])
    x = np.arange(40, dtype=np.float32).reshape(10, 4)
    assert np.array_equal(api.predict("idn", x), x)
    out = np.zeros((10, 4), np.float32)
    assert api.predict_into("idn", x, out) == (10, 4) and np.array_equal(out, x)
    with pytest.raises(api.InferaError, match="output buffer too small"):
        api.predict_into("idn", x, np.zeros(5, np.float32))
    api.unload_model("idn")


def test_blob_paths(api, models, tmp_path):
    from infera_amd import onnx_writer as W, synth
    from oracle import oracle

    # dynamic-batch model: a blob holding k samples infers batch = k (engine.rs:233-238)
    api.load_model("mlp", models["mlp"])
    x = synth.table(2, 5, 7, 128)
    assert_close(api.predict_from_blob("mlp", x.tobytes()), oracle.Model(models["mlp"]).predict_blob(x.tobytes()))
    with pytest.raises(api.InferaError, match=r"Expected 128 elements, but BLOB contained 100\."):
        api.predict_from_blob("mlp", b"\0" * 400)
    # batched-blob entry: one call for a chunk of single-sample blobs == per-blob calls
    path = W.write(str(tmp_path / "rn.onnx"), W.resnet18(classes=10, in_hw=32, width=8))
    api.load_model("rn", path)
    imgs = synth.table(13, 0, 5, 3 * 32 * 32)
    per = np.concatenate([api.predict_from_blob("rn", imgs[i].tobytes()) for i in range(5)])
    bat = api.predict_from_blob_batch("rn", [imgs[i].tobytes() for i in range(5)])
    assert bat.shape == (5, 10)
    assert_close(bat, per)
    assert_close(bat, oracle.Model(path).predict_blob(imgs.tobytes()))
    with pytest.raises(api.InferaError, match="Invalid BLOB size"):
        api.predict_from_blob_batch("rn", [b"\0" * 5])
    api.unload_model("rn")


def test_columnar_gather_path(api, models):
    from oracle import oracle

    api.load_model("lin", models["linear_dyn"])
    rows = 300
    rng = np.random.default_rng(0)
    cols = [rng.standard_normal(rows), rng.integers(-5, 5, rows).astype(np.int32), rng.integers(-5, 5, rows).astype(np.int64)]
    want = oracle.Model(models["linear_dyn"]).predict(oracle.extract_features(cols))
    assert_close(api.predict_columns("lin", cols), want)
    # CONSTANT_VECTOR column + FLOAT column
    cols2 = [np.array([1.5], np.float32), cols[0].astype(np.float32), cols[0].astype(np.float32)]
    feat = np.stack([np.full(rows, 1.5, np.float32), cols2[1], cols2[2]], axis=1)
    assert_close(api.predict_columns("lin", cols2, rows=rows), oracle.Model(models["linear_dyn"]).predict(feat))
    # a NULL cell is an error (infera_extension.cpp:207-209)
    valid = np.full((rows + 63) // 64, np.uint64(0xFFFFFFFFFFFFFFFF))
    valid[1] &= ~np.uint64(1 << 3)
    with pytest.raises(api.InferaError, match="^Feature values cannot be NULL$"):
        api.predict_columns("lin", cols, validity=[None, valid, None])
    api.unload_model("lin")


def test_concurrency_like_reference(api, models):
    """test/concurrency/test_concurrency.py:25-50, 71-78 against the C ABI: 8 threads x 10 x
    (load lin_{t}_{i}, predict (1,2,3) == 1.75 +- 1e-5, unload), extra unload of a missing name
    fails harmlessly, and nothing stays loaded."""
    import threading

    errors = []

    def worker(t):
        try:
            for i in range(10):
                n = f"lin_{t}_{i}"
                api.load_model(n, models["linear"])
                r = api.predict(n, np.array([[1.0, 2.0, 3.0]], np.float32))
                assert abs(float(r[0, 0]) - 1.75) <= 1e-5
                api.unload_model(n)
            assert api.load_library().infera_unload_model(b"non_existent_again") == -1
        except Exception as e:  # pragma: no cover
            errors.append(f"thread {t}: {e!r}")

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert [m for m in api.get_loaded_models() if m.startswith("lin_")] == []


def test_concurrent_scans_share_one_model(api, models):
    """Many DuckDB-style worker threads, one model, disjoint 2048-row chunks."""
    import threading

    from infera_amd import synth
    from oracle import oracle

    api.load_model("mlp", models["mlp"])
    nchunks, errors = 32, []
    want = oracle.Model(models["mlp"]).predict(synth.table(42, 0, nchunks * 2048, 128))
    got = np.zeros_like(want)

    def worker(t):
        try:
            for c in range(t, nchunks, 8):
                got[c * 2048:(c + 1) * 2048] = api.predict("mlp", synth.table(42, c * 2048, 2048, 128))
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert_close(got, want)


def test_unload_and_replace_during_inference(api, models, tmp_path):
    """The registry hands out shared ownership (host/engine.cpp): a model unloaded or REPLACED under the same
    name while other threads are inside infera_predict must neither crash nor corrupt results -- every answer is
    the old model's or the new model's, bit for bit, and an unloaded name fails with the reference's text.
    (The reference blocks unload until readers drain, model.rs:41-42; here the HBM is released by the last user.)"""
    import threading
    import time

    from infera_amd import onnx_writer as W, synth
    from oracle import oracle

    pa = W.write(str(tmp_path / "va.onnx"), W.mlp((128, 256, 64, 1), seed=1))
    pb = W.write(str(tmp_path / "vb.onnx"), W.mlp((128, 256, 64, 1), seed=2))
    x = synth.table(3, 0, 2048, 128)
    ya, yb = oracle.Model(pa).predict(x), oracle.Model(pb).predict(x)
    assert np.abs(ya - yb).max() > 1e-2
    api.load_model("swap", pa)
    ga = api.predict("swap", x)
    api.load_model("swap", pb)
    gb = api.predict("swap", x)
    assert_close(ga, ya)
    assert_close(gb, yb)
    stop, errors, counts = threading.Event(), [], {"a": 0, "b": 0, "missing": 0}

    def reader():
        try:
            while not stop.is_set():
                try:
                    y = api.predict("swap", x)
                except api.InferaError as e:
                    assert str(e) == "Model not found: swap", str(e)
                    counts["missing"] += 1
                    continue
                if np.array_equal(y, ga):
                    counts["a"] += 1
                elif np.array_equal(y, gb):
                    counts["b"] += 1
                else:
                    raise AssertionError("result is neither model's output")
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    th = [threading.Thread(target=reader) for _ in range(6)]
    [t.start() for t in th]
    t_end = time.time() + 3.0
    i = 0
    while time.time() < t_end:
        i += 1
        if i % 3 == 0:
            api.load_library().infera_unload_model(b"swap")
        api.load_model("swap", pa if i % 2 else pb)
    stop.set()
    [t.join() for t in th]
    assert not errors, errors[:3]
    assert counts["a"] > 0 and counts["b"] > 0, counts
    api.load_library().infera_unload_model(b"swap")


def test_pipelined_host_staging_matches_single_pass(api, models):
    """Host inputs above 24 MB go through the two-slot staging pipeline (16 MB passes, CPU copy of pass i+1
    overlapping the GPU work of pass i): same values, bit for bit, as the same rows sent in small calls, for a
    row count that leaves a ragged last pass; likewise a BLOB batch that spans several passes."""
    from infera_amd import onnx_writer as W, synth

    api.load_model("mlp", models["mlp"])
    rows = 3 * 32768 + 4321  # 52.5 MB of features -> 4 passes, the last one short
    x = synth.table(77, 0, rows, 128)
    big = api.predict("mlp", x)
    assert big.shape == (rows, 1)
    for s in (0, 32768 - 5, 2 * 32768, rows - 2048):
        assert np.array_equal(api.predict("mlp", x[s:s + 2048]), big[s:s + 2048])
    api.load_model("lr", models["logreg"])
    bigl = api.predict("lr", x)
    assert np.array_equal(api.predict("lr", x[40000:42048]), bigl[40000:42048])
    api.unload_model("lr")


def test_full_size_c2_properties(api, models):
    """BASELINE config C2 at full size (10M rows x 128, one device-resident scan): the oracle cannot
    cover 10M rows in seconds, so check size-independent properties --
      * determinism: two scans are bit-identical;
      * chunk invariance: any 2048-row DataChunk pushed through infera_predict (host ABI) gives
        bit-identical values to its slice of the big scan;
      * oracle parity on 48 sampled chunks spread over the table (incl. first, last, ragged tail);
      * checksum of checksums: per-range f64 sums add up to the whole-table sum."""
    from infera_amd import synth
    from oracle import oracle

    rows, cols = 10_000_000 + 1234, 128  # ragged tail on purpose
    api.load_model("mlp", models["mlp"])
    dev = api.device_ordinal(0)
    d_in = api.DeviceBuffer(dev, rows * cols * 4)
    d_out = api.DeviceBuffer(dev, rows * 4)
    api.synth_fill(d_in, 42, 0, rows, cols)
    assert api.predict_device("mlp", d_in, rows, cols, d_out) == (rows, 1)
    y1 = d_out.download((rows,))
    api.predict_device("mlp", d_in, rows, cols, d_out)
    y2 = d_out.download((rows,))
    assert np.array_equal(y1, y2) and np.isfinite(y1).all()
    om = oracle.Model(models["mlp"])
    starts = sorted(set([0, rows - 2048, rows - 1234 - 2048] + list(np.random.default_rng(1).integers(0, rows - 2048, 45))))
    for s in starts:
        x = synth.table(42, int(s), 2048, cols)
        host = api.predict("mlp", x).ravel()
        assert np.array_equal(host, y1[s:s + 2048]), f"chunk at row {s} differs between host ABI and device scan"
        assert_close(host.reshape(-1, 1), om.predict(x))
    total = y1.astype(np.float64).sum()
    parts = sum(y1[a:a + 1_000_000].astype(np.float64).sum() for a in range(0, rows, 1_000_000))
    assert abs(total - parts) <= 1e-6 * abs(total) + 1e-6


def test_full_size_c2_host_scan_equals_device_scan(api, models):
    """BASELINE config C2 at full size through the WHOLE host path: a 10M-row x 128-col columnar table materialised in host
    memory (DuckDB's storage shape), scanned by 16 worker threads in 2048-row chunks through the SQL surface
    (gather -> pinned staging -> H2D -> kernel -> result vector), must produce the same outputs as one device-resident
    scan of the same table: every chunk is bit-identical to its slice (asserted on samples elsewhere), so the f64 sum of all
    10M outputs -- a checksum over every chunk, thread and staging context -- agrees to summation-order rounding; the
    per-slot row counter accounts for every row exactly once."""
    from infera_amd import sqlharness

    rows, cols = 10_000_000, 128
    api.load_model("mlp", models["mlp"])
    dev = api.device_ordinal(0)
    d_in = api.DeviceBuffer(dev, rows * cols * 4)
    d_out = api.DeviceBuffer(dev, rows * 4)
    api.synth_fill(d_in, 42, 0, rows, cols)
    api.predict_device("mlp", d_in, rows, cols, d_out)
    want = d_out.download((rows,)).astype(np.float64).sum()
    del d_in, d_out
    table = sqlharness.synth_table(rows, cols, 42, 16)
    before = sum(d["host_rows"] for d in api.get_devices()["devices"])
    for threads in (16, 3):
        secs, checksum = sqlharness.bench_scan_table("infera_predict", "mlp", table, rows, cols, threads, 1)
        assert abs(checksum - want) <= 1e-9 * abs(want) + 1e-6, (threads, checksum, want)
    assert sum(d["host_rows"] for d in api.get_devices()["devices"]) - before == 2 * rows


def test_full_size_c3_row_range_sharding_on_one_gpu(api, models):
    """BASELINE config C3 -- the same MLP over 100M rows x 128, row-range sharded over 8 GPUs -- as far as ONE GPU allows:
    the 51.2 GB table fits in HBM (288 GB), so the eight 12.5M-row shards a rank each would scan (shard.row_range) are
    scanned one after the other from their own freshly generated row ranges, results placed by row offset, and compared
    with ONE scan of the whole 100M-row table: bit-identical (row-range sharding does not change any row's arithmetic).
    Oracle parity on chunks sampled from every shard, and the checksum of the shards' checksums equals the table's."""
    from infera_amd import shard, synth
    from oracle import oracle

    world, per_rank, cols = 8, 12_500_000, 128
    rows = world * per_rank
    api.load_model("mlp", models["mlp"])
    dev = api.device_ordinal(0)
    d_in = api.DeviceBuffer(dev, rows * cols * 4)
    d_out = api.DeviceBuffer(dev, rows * 4)
    api.synth_fill(d_in, 42, 0, rows, cols)
    assert api.predict_device("mlp", d_in, rows, cols, d_out) == (rows, 1)
    whole = d_out.download((rows,))
    assert np.isfinite(whole).all()
    d_shard_in = api.DeviceBuffer(dev, per_rank * cols * 4)
    d_shard_out = api.DeviceBuffer(dev, per_rank * 4)
    om = oracle.Model(models["mlp"])
    rng = np.random.default_rng(3)
    shard_sums = []
    for rank in range(world):
        r0, r1 = shard.row_range(rank, world, per_rank)
        api.synth_fill(d_shard_in, 42, r0, per_rank, cols)       # what rank `rank` generates for itself (bench.py)
        assert api.predict_device("mlp", d_shard_in, per_rank, cols, d_shard_out) == (per_rank, 1)
        part = d_shard_out.download((per_rank,))
        assert np.array_equal(part, whole[r0:r1]), f"shard {rank} differs from its row range of the whole-table scan"
        shard_sums.append(part.astype(np.float64).sum())
        for s in [0, per_rank - 2048] + list(rng.integers(0, per_rank - 2048, 3)):
            x = synth.table(42, r0 + int(s), 2048, cols)
            assert_close(part[s:s + 2048].reshape(-1, 1), om.predict(x))
    total = whole.astype(np.float64).sum()
    assert abs(total - sum(shard_sums)) <= 1e-6 * abs(total) + 1e-6


def test_full_size_c4_properties(api, models):
    """BASELINE config C4 at full size (50M rows x 128 -> softmax over 10 classes, device-resident):
      * every output row is a probability vector (sums to 1 within 4 ulp * 10, entries in [0, 1]);
      * determinism across two scans; chunk invariance against the host ABI; oracle parity on sampled chunks;
      * invariance to a row permutation of the input (rows are independent): a reversed copy of a slice
        gives the reversed outputs bit for bit."""
    from infera_amd import synth
    from oracle import oracle

    rows, cols, classes = 50_000_000 + 777, 128, 10
    api.load_model("lr", models["logreg"])
    dev = api.device_ordinal(0)
    d_in = api.DeviceBuffer(dev, rows * cols * 4)
    d_out = api.DeviceBuffer(dev, rows * classes * 4)
    api.synth_fill(d_in, 42, 0, rows, cols)
    assert api.predict_device("lr", d_in, rows, cols, d_out) == (rows, classes)
    y1 = d_out.download((rows, classes))
    api.predict_device("lr", d_in, rows, cols, d_out)
    assert np.array_equal(y1, d_out.download((rows, classes)))
    sums = y1.astype(np.float64).sum(axis=1)
    assert np.abs(sums - 1.0).max() < 5e-6 and y1.min() >= 0.0 and y1.max() <= 1.0
    om = oracle.Model(models["logreg"])
    for s in [0, rows - 2048, rows - 777 - 2048] + list(np.random.default_rng(2).integers(0, rows - 2048, 13)):
        x = synth.table(42, int(s), 2048, cols)
        host = api.predict("lr", x)
        assert np.array_equal(host, y1[s:s + 2048])
        assert_close(host, om.predict(x))
        assert np.array_equal(api.predict("lr", x[::-1].copy()), host[::-1])
    api.unload_model("lr")


def test_c5_batch_position_invariance(api, tmp_path):
    """C5 (ResNet-18 at the BASELINE resolution 224x224, full width): an image's logits do not depend on the
    batch it travels in or on its position -- 40 images in one call == the same images one by one and in
    reversed order (this crosses pixel-tile boundaries of every conv kernel: 40*49 pixels is not a multiple of
    128) -- and 2 of them match the oracle."""
    from infera_amd import onnx_writer as W, synth
    from oracle import oracle

    path = W.write(str(tmp_path / "rn224.onnx"), W.resnet18())
    api.load_model("rn224", path)
    n, per = 40, 3 * 224 * 224
    imgs = synth.table(5, 0, n, per)
    y = api.predict_from_blob("rn224", imgs.tobytes())
    assert y.shape == (n, 1000) and np.isfinite(y).all()
    yr = api.predict_from_blob("rn224", imgs[::-1].copy().tobytes())
    assert np.array_equal(yr[::-1], y)
    for i in (0, 17, 39):
        assert np.array_equal(api.predict_from_blob("rn224", imgs[i:i + 1].tobytes())[0], y[i])
    want = oracle.Model(path).predict_blob(imgs[:2].tobytes())
    assert_close(y[:2], want)
    api.unload_model("rn224")


def test_hipgraph_mode_matches_direct_mode(models):
    """INFERA_HIPGRAPH=1 (a captured {H2D, kernels, D2H} graph per (model, rows), north_star) in a fresh
    process: same values as the default direct-enqueue mode, and safe against concurrent model loads
    in other threads (loads must not touch the legacy stream while a capture is open)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, threading, json, numpy as np
sys.path.insert(0, %r)
from infera_amd import capi, synth
capi.load_model("mlp", %r)
x = synth.table(42, 0, 2048, 128)
errors, outs = [], {}
def scan(t):
    try:
        for i in range(20):
            outs[(t, i)] = capi.predict("mlp", x)
    except Exception as e:
        errors.append(repr(e))
def loader(t):
    try:
        for i in range(10):
            capi.load_model(f"l{t}_{i}", %r); capi.unload_model(f"l{t}_{i}")
    except Exception as e:
        errors.append(repr(e))
th = [threading.Thread(target=scan, args=(t,)) for t in range(4)] + [threading.Thread(target=loader, args=(t,)) for t in range(2)]
[t.start() for t in th]; [t.join() for t in th]
ref = next(iter(outs.values()), np.zeros(1))
print(json.dumps({"errors": errors, "same": len(outs) == 80 and all(np.array_equal(v, ref) for v in outs.values()), "sum": float(ref.astype(np.float64).sum())}))
''' % (root, models["mlp"], models["linear"])
    res = {}
    for mode in ("0", "1"):
        env = dict(os.environ, INFERA_HIPGRAPH=mode)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        res[mode] = json.loads(out.stdout.strip().splitlines()[-1])
        assert res[mode]["errors"] == [] and res[mode]["same"], res[mode]
    assert res["0"]["sum"] == res["1"]["sum"]


def test_resnet18_full_width_vs_oracle(api, tmp_path):
    """Full-width ResNet-18 topology (64..512 channels: every tiled channel-quad conv variant, stride-2 and 1x1
    downsample convs, residual adds, BN folding, max/global-average pooling) at 64x64 input."""
    from infera_amd import onnx_writer as W, synth
    from oracle import oracle

    path = W.write(str(tmp_path / "rn64.onnx"), W.resnet18(classes=10, in_hw=64, width=64))
    api.load_model("rn64", path)
    plan = api.get_plan("rn64")
    # (default plan: every tiled convolution on the bf16 matrix cores, operands cut exactly into three parts -- DESIGN.md 3.3c)
    assert plan["activation_layout"] == "NC/4HW4" and plan["exec"].count("conv_split_bf16x6") == 16 and len(plan["folded_shortcuts"]) == 3 and "bf16x6" in plan["conv_precision"]
    imgs = synth.table(21, 0, 3, 3 * 64 * 64)
    got = api.predict_from_blob("rn64", imgs.tobytes())
    want = oracle.Model(path).predict_blob(imgs.tobytes())
    assert got.shape == (3, 10)
    assert_close(got, want)
    api.unload_model("rn64")


def test_tiled_conv_feature_tile_variants(api, tmp_path):
    """Every feature-tile count of the tiled conv (M/32 = 1, 2, 3, 4, 5, 6 -> MT 1, 2, 3, 4, 1, 3) with both channel
    depths (C % 64 == 0 or not), 1x1 and 3x3, stride 1 and 2: MobileNet-style widths."""
    from infera_amd import onnx_writer as W, synth
    from oracle import oracle

    rng = np.random.default_rng(11)
    chain = [(32, 3, 1), (96, 1, 1), (160, 3, 2), (64, 1, 1), (192, 3, 1), (128, 1, 2), (32, 3, 1)]
    nodes, inits, x, cin = [], [], "X", 4
    for i, (cout, k, stride) in enumerate(chain):
        w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        inits += [W.tensor(f"w{i}", w), W.tensor(f"b{i}", b)]
        nodes.append(W.node("Conv", [x, f"w{i}", f"b{i}"], [f"c{i}"], [W.attr_ints("kernel_shape", [k, k]), W.attr_ints("strides", [stride] * 2),
                                                                   W.attr_ints("pads", [k // 2] * 4)]))
        nodes.append(W.node("Relu", [f"c{i}"], [f"r{i}"]))
        x, cin = f"r{i}", cout
    nodes += [W.node("GlobalAveragePool", [x], ["g"]), W.node("Flatten", ["g"], ["Y"], [W.attr_i("axis", 1)])]
    path = W.write(str(tmp_path / "widths.onnx"), W.model("widths", nodes, inits, [W.value_info("X", ["N", 4, 20, 20])], [W.value_info("Y", ["N", 32])]))
    import os

    os.environ["INFERA_PRECISION"] = "fp32"  # (read when the model is scheduled: this test is about the exact-fp32 tiled kernel's instantiations)
    try:
        api.load_model("widths", path)
    finally:
        os.environ.pop("INFERA_PRECISION", None)
    plan = api.get_plan("widths")
    assert plan["activation_layout"] == "NC/4HW4" and plan["exec"].count("conv_tiled_cq") == 6, plan["exec"]
    imgs = synth.table(13, 0, 7, 4 * 20 * 20)
    assert_close(api.predict_from_blob("widths", imgs.tobytes()), oracle.Model(path).predict_blob(imgs.tobytes()))
    api.unload_model("widths")


def test_conv_plan_falls_back_to_nchw_when_flatten_needs_it(api, tmp_path):
    """Conv -> Flatten(C*H*W) -> MatMul reads the feature map through the layer's permuted weight rows (the conv layout
    stays); a Softmax over the flattened features looks at them in NCHW order: that plan must stay NCHW."""
    from infera_amd import onnx_writer as W
    from oracle import oracle

    rng = np.random.default_rng(3)
    w = rng.standard_normal((8, 3, 3, 3)).astype(np.float32) * 0.2
    fc = rng.standard_normal((8 * 6 * 6, 5)).astype(np.float32) * 0.1
    x = rng.standard_normal((4, 3, 8, 8)).astype(np.float32)
    for softmax, layout in [(False, "NC/4HW4"), (True, "NCHW")]:
        nodes = [W.node("Conv", ["X", "w"], ["c"], [W.attr_ints("kernel_shape", [3, 3])]), W.node("Relu", ["c"], ["r"]),
                 W.node("Flatten", ["r"], ["f"], [W.attr_i("axis", 1)])]
        if softmax:
            nodes.append(W.node("Softmax", ["f"], ["s"], [W.attr_i("axis", 1)]))
        nodes.append(W.node("MatMul", ["s" if softmax else "f", "fc"], ["Y"]))
        blob = W.model("convfc", nodes, [W.tensor("w", w), W.tensor("fc", fc)], [W.value_info("X", ["N", 3, 8, 8])], [W.value_info("Y", ["N", 5])])
        path = W.write(str(tmp_path / f"convfc{int(softmax)}.onnx"), blob)
        api.load_model("convfc", path)
        assert api.get_plan("convfc")["activation_layout"] == layout
        assert_close(api.predict_from_blob("convfc", x.tobytes()), oracle.Model(path).predict_blob(x.tobytes()))
        api.unload_model("convfc")


def test_empty_and_multi_pass_host_inputs(api, models):
    """Edge sizes of the host ABI: zero rows (empty DataChunk) and a call larger than the 64 MiB pinned
    staging window (several H2D/D2H passes inside one infera_predict)."""
    from infera_amd import synth
    from oracle import oracle

    api.load_model("mlp", models["mlp"])
    out = api.predict("mlp", np.zeros((0, 128), np.float32))
    assert out.size == 0
    rows = 300_001  # 153.6 MB of features -> 3 staging passes, ragged tail
    x = synth.table(5, 0, rows, 128)
    got = api.predict("mlp", x)
    om = oracle.Model(models["mlp"])
    for s in (0, 131_000, rows - 4097):
        assert_close(got[s:s + 4097], om.predict(x[s:s + 4097]))
    api.load_model("idn", models["identity_dyn"])
    assert api.predict("idn", np.zeros((0, 4), np.float32)).size == 0
    api.unload_model("idn")
    # unaligned BLOB pointer (DuckDB string_t payloads carry no alignment guarantee)
    raw = bytearray(1 + 128 * 4 * 3)
    raw[1:] = x[:3].tobytes()
    import ctypes as C
    buf = (C.c_uint8 * len(raw)).from_buffer(raw)
    L = api.load_library()
    res = L.infera_predict_from_blob(b"mlp", C.addressof(buf) + 1, 128 * 4 * 3)
    assert res.status == 0 and res.rows == 3
    got3 = np.ctypeslib.as_array(res.data, shape=(3,)).copy()
    L.infera_free_result(res)
    assert np.array_equal(got3, got[:3].ravel())


@pytest.mark.parametrize("dims,acts,kernel", [
    ((64, 128, 32, 1), ["Relu", "Relu", ""], "mlp3_split_kernel"),      # regression head, 2 waves/SIMD
    ((32, 64, 64, 2), ["Sigmoid", "Tanh", ""], "mlp3_split_kernel"),    # other activations baked in
    ((128, 96, 32, 3), ["Relu", "Relu", "Sigmoid"], "mlp3_split_kernel"),  # odd tile count (MT1 = 3)
    ((16, 32, 32, 7), ["Relu", "", ""], "mlp3_kernel"),                 # wide head -> MFMA head, 1 wave/SIMD
    ((256, 128, 128, 16), ["Relu", "Relu", ""], "mlp3_kernel"),         # wide input (x = 128 regs)
])
def test_load_time_specialised_fused_chains(api, tmp_path, dims, acts, kernel):
    """Chains with no ahead-of-time instantiation are specialised by hipRTC at infera_load_model time
    from the same device source; results must match the oracle and the layer-by-layer plan."""
    import os
    import subprocess
    import sys

    from infera_amd import onnx_writer as W, synth
    from oracle import oracle

    path = W.write(str(tmp_path / "m.onnx"), W.mlp(dims, acts=acts))
    api.load_model("jit", path)
    plan = api.get_plan("jit")
    assert plan["exec"] == ["mlp3_fused", "skipped", "skipped"], plan
    assert kernel in plan["fused_kernel"] and "hipRTC" in plan["fused_kernel"], plan["fused_kernel"]
    rows = 5000 + 13
    x = synth.table(17, 3, rows, dims[0])
    got = api.predict("jit", x)
    assert_close(got, oracle.Model(path).predict(x))
    api.unload_model("jit")
    # same model with fusion disabled (fresh process: INFERA_FUSED_MLP is read once): layer-by-layer kernels
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from infera_amd import capi, synth\n"
            "capi.load_model('m', %r); assert capi.get_plan('m')['exec'][0] in ('normal', 'dense_tiled')\n"
            "np.save(%r, capi.predict('m', synth.table(17, 3, %d, %d)))\n") % (
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, str(tmp_path / "unfused.npy"), rows, dims[0])
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, INFERA_FUSED_MLP="0"), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-1500:]
    assert_close(got, np.load(tmp_path / "unfused.npy"))


def test_config_c1_linear_1k_rows_with_batch_split(models):
    """BASELINE config C1: the reference's linear.onnx (fixed batch of 1) on a 1,000-row x 3 FLOAT table.
    The reference rejects any chunk with more than one row (fixed batch, test/models/README.md:5); with
    INFERA_BATCH_SPLIT=1 (read once, hence a fresh process) the fixed leading dim is treated as the row
    axis: expected 2*x1 - x2 + 0.5*x3 + 0.25 per row, and the dynamic-batch twin gives the same values."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from infera_amd import capi, synth
from oracle import oracle
capi.load_model("linear", %r); capi.load_model("twin", %r)
x = synth.table(3, 0, 1000, 3)
y = capi.predict("linear", x)
want = (2 * x[:, 0].astype(np.float64) - x[:, 1] + 0.5 * x[:, 2] + 0.25).reshape(-1, 1)
assert y.shape == (1000, 1) and np.allclose(y, want, rtol=1e-5, atol=1e-6)
assert np.array_equal(y, capi.predict("twin", x))
assert np.allclose(y, oracle.Model(%r).predict(x), rtol=1e-6, atol=1e-7)
print("ok")
''' % (root, models["linear"], models["linear_dyn"], models["linear_dyn"])
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, INFERA_BATCH_SPLIT="1"), capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]


def test_many_more_caller_threads_than_admitted_submissions(api, models):
    """64 caller threads (a DuckDB scan on a many-core host) against the default admission limit of 12 in-flight calls per
    GPU: every chunk's result is right, nobody starves or deadlocks"""
    import threading

    from infera_amd import synth
    from oracle import oracle

    api.load_model("mlp", models["mlp"])
    x = synth.table(77, 0, 2048, 128)
    want = oracle.Model(models["mlp"]).predict(x)
    errors = []

    def worker(t):
        try:
            for it in range(12):
                lo = (37 * t + 101 * it) % 1500
                got = api.predict("mlp", x[lo:lo + 500])
                assert_close(got, want[lo:lo + 500])
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(64)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in threads), "a caller is stuck behind the admission gate"
    assert not errors, errors[:3]
    api.unload_model("mlp")
