"""CPU tests pinning the ORACLE (oracle/infera_oracle.c) -- the checker the GPU parity tests trust.

1. against every golden value / error string the reference's own tests hold for this path
    -- this is what "parity pinned" rests on for MatMul+Add and Identity;
2. against an independent float64 numpy evaluation for the operators the reference never tests
   (Gemm, Relu, Sigmoid, Tanh, Softmax) -- "parity unpinned" there, the ONNX spec is the authority;
3. against the committed golden vectors (tests/golden/vectors.npz).
"""
import hashlib
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def O(built):
    from oracle import oracle

    return oracle


def test_fixture_files_are_the_reference_fixtures():
    # sha256 recorded in SURVEY.md section 8c
    want = {"linear.onnx": "d9b9cbd7ee5d1aa43f6b49a3689c5f282c2462fccc71cde50228456007d90527",
            "multi_output.onnx": "2e86afcbca8920dcfcc1c18c8a2bafc8160b6a9f079c1b4880af427f4e4fecc5"}
    for f, h in want.items():
        assert hashlib.sha256(open(os.path.join(GOLD, f), "rb").read()).hexdigest() == h


def test_linear_reference_goldens(O):
    m = O.Model(os.path.join(GOLD, "linear.onnx"))
    assert m.input_shape == [1, 3] and m.output_shape == [1, 1]  # test_core_functionality.test:41-44
    # test/sql/test_core_functionality.test:48-56, test_predict_multi_list.test:26-29, test_concurrency.py:40-42
    assert m.predict(np.array([[1.0, 2.0, 3.0]], np.float32)).tolist() == [[1.75]]
    # zero blob of 12 bytes returns the bias (test_volatile_and_null_safety.test:56-59)
    assert m.predict_blob(b"\0" * 12).tolist() == [[0.25]]
    # docs/examples/e3_integration_and_errors.sql:14-24 intended per-row values
    assert m.predict(np.array([[0.5, 1.0, 1.5]], np.float32)).tolist() == [[1.0]]
    assert m.predict(np.array([[-1.0, 0.0, 2.0]], np.float32)).tolist() == [[-0.75]]


def test_multi_output_reference_goldens(O):
    m = O.Model(os.path.join(GOLD, "multi_output.onnx"))
    assert m.input_shape == [1, 4] and m.output_shape == [1, 4]  # test_multi_output.test:17-20
    y = m.predict(np.array([[1, 2, 3, 4]], np.float32))
    assert y.tolist() == [[1.0, 2.0, 3.0, 4.0]]  # test_predict_multi_list.test:20-23
    assert O.predict_multi_json(y) == ["[1,2,3,4]"]  # test_multi_output.test:23-26
    with pytest.raises(O.OracleError, match=r"Model output shape mismatch. Expected \(1, 1\), but got \(1, 4\)\."):
        O.check_predict_shape(1, 4, 1)  # test_multi_output.test:29-32


def test_reference_error_strings(O):
    m = O.Model(os.path.join(GOLD, "linear.onnx"))
    with pytest.raises(O.OracleError, match="^Invalid BLOB size: length must be a multiple of 4$"):
        m.predict_blob(b"\0" * 5)  # test_edge_cases.test:27-30, lib.rs:579-601
    with pytest.raises(O.OracleError, match=r"Expected 3 elements, but BLOB contained 4\.$"):
        m.predict_blob(b"\0" * 16)  # test_edge_cases.test:33-36
    with pytest.raises(O.OracleError, match=r"^Invalid input shape: expected batch x \[3\], got 1 x 2$"):
        m.predict(np.zeros((1, 2), np.float32))  # lib.rs:603-630
    with pytest.raises(O.OracleError, match="^ONNX error: "):
        m.predict(np.zeros((2, 3), np.float32))  # fixed batch of 1 (test/models/README.md:5)
    with pytest.raises(O.OracleError, match="^ONNX error: "):
        m.predict_blob(b"\0" * 24)  # passes the modulo check, from_shape fails


def test_shape_rows_cols_table(O):
    # engine.rs:321-328
    assert O.shape_rows_cols([]) == (1, 1)
    assert O.shape_rows_cols([5]) == (5, 1)
    assert O.shape_rows_cols([2, 3]) == (2, 3)
    assert O.shape_rows_cols([2, 3, 4]) == (2, 12)
    assert O.shape_rows_cols([1, 1, 1, 1]) == (1, 1)


def test_binding_formatting(O):
    assert O.format_g(0.1) == "0.1" and O.format_g(1e-7) == "1e-07" and O.format_g(123456789.0) == "1.23457e+08"
    cols = [np.array([1.5, 2.5], np.float64), np.array([1, 2], np.int32), np.array([3, 4], np.int64), np.array([0.5, 0.25], np.float32)]
    f = O.extract_features(cols)
    assert f.dtype == np.float32 and f.tolist() == [[1.5, 1.0, 3.0, 0.5], [2.5, 2.0, 4.0, 0.25]]
    with pytest.raises(O.OracleError, match="Feature values cannot be NULL"):  # test_integration_and_errors.test:53-57
        O.extract_features([np.ma.masked_array([1.0, 2.0], mask=[False, True])])


def test_synth_generator_numpy_equals_c(O):
    from infera_amd import synth

    for seed, row0, rows, cols in [(42, 0, 100, 128), (7, 123456789, 33, 3), (1, 2 ** 40, 5, 17)]:
        assert np.array_equal(synth.table(seed, row0, rows, cols), O.synth_table(seed, row0, rows, cols))
    x = synth.table(42, 0, 4096, 128)
    assert x.min() >= -1.0 and x.max() < 1.0 and abs(float(x.mean())) < 0.01


def _mlp_weights(dims, seed=1234):
    """Re-derives the writer's weights independently of the ONNX file (same documented recipe)."""
    from infera_amd.onnx_writer import _WeightStream

    ws = _WeightStream(seed)
    return [(ws.take((k, m), k), ws.take((m,), k)) for k, m in zip(dims[:-1], dims[1:])]


def test_oracle_vs_numpy_float64_mlp(O, models):
    from infera_amd import synth

    x = synth.table(42, 0, 512, 128)
    h = x.astype(np.float64)
    layers = _mlp_weights((128, 256, 64, 1))
    for i, (w, b) in enumerate(layers):
        h = h @ w.astype(np.float64) + b.astype(np.float64)
        if i < len(layers) - 1:
            h = np.maximum(h, 0.0)
    y = O.Model(models["mlp"]).predict(x)
    np.testing.assert_allclose(y, h, rtol=2e-5, atol=2e-6)


def test_oracle_vs_numpy_float64_logreg_softmax(O, models):
    from infera_amd import synth

    x = synth.table(3, 9, 300, 128)
    (w, b), = _mlp_weights((128, 10))
    z = x.astype(np.float64) @ w.astype(np.float64) + b
    e = np.exp(z - z.max(axis=1, keepdims=True))
    y = O.Model(models["logreg"]).predict(x)
    np.testing.assert_allclose(y, e / e.sum(axis=1, keepdims=True), rtol=2e-5, atol=1e-7)


def test_oracle_vs_numpy_float64_activations(O, tmp_path):
    from infera_amd import onnx_writer as W, synth

    p = W.write(str(tmp_path / "m.onnx"), W.mlp((16, 32, 32, 3), acts=["Sigmoid", "Tanh", ""]))
    x = synth.table(5, 10, 64, 16)
    (w1, b1), (w2, b2), (w3, b3) = _mlp_weights((16, 32, 32, 3))
    h = 1.0 / (1.0 + np.exp(-(x.astype(np.float64) @ w1 + b1)))
    h = np.tanh(h @ w2 + b2)
    np.testing.assert_allclose(O.Model(p).predict(x), h @ w3 + b3, rtol=2e-5, atol=2e-6)


def test_oracle_gemm_attributes_and_matmul_add_equivalence(O, tmp_path):
    from infera_amd import onnx_writer as W, synth

    x = synth.table(1, 0, 40, 24)
    a = O.Model(W.write(str(tmp_path / "a.onnx"), W.mlp((24, 8, 5), acts=["Relu", ""]))).predict(x)
    b = O.Model(W.write(str(tmp_path / "b.onnx"), W.mlp((24, 8, 5), acts=["Relu", ""], use_matmul_add=True))).predict(x)
    c = O.Model(W.write(str(tmp_path / "c.onnx"), W.mlp((24, 8, 5), acts=["Relu", ""], trans_b=True))).predict(x)
    assert np.array_equal(a, b) and np.array_equal(a, c)


def test_oracle_conv_vs_numpy(O, tmp_path):
    """Conv + BatchNormalization + Relu + MaxPool + GlobalAveragePool against a direct float64 loop."""
    from infera_amd import onnx_writer as W

    rng = np.random.default_rng(0)
    w = rng.standard_normal((4, 3, 3, 3)).astype(np.float32)
    bias = rng.standard_normal(4).astype(np.float32)
    sc, be, mu, var = (rng.standard_normal(4).astype(np.float32) for _ in range(4))
    var = np.abs(var) + 0.5
    nodes = [W.node("Conv", ["X", "w", "b"], ["c"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("strides", [2, 2]), W.attr_ints("pads", [1, 1, 1, 1])]),
             W.node("BatchNormalization", ["c", "sc", "be", "mu", "var"], ["n"], [W.attr_f("epsilon", 1e-5)]),
             W.node("Relu", ["n"], ["r"]),
             W.node("MaxPool", ["r"], ["p"], [W.attr_ints("kernel_shape", [2, 2]), W.attr_ints("strides", [2, 2])]),
             W.node("GlobalAveragePool", ["p"], ["g"]), W.node("Flatten", ["g"], ["Y"], [W.attr_i("axis", 1)])]
    inits = [W.tensor(n, a) for n, a in [("w", w), ("b", bias), ("sc", sc), ("be", be), ("mu", mu), ("var", var)]]
    blob = W.model("conv", nodes, inits, [W.value_info("X", ["N", 3, 9, 9])], [W.value_info("Y", ["N", 4])])
    m = O.Model(W.write(str(tmp_path / "conv.onnx"), blob))
    assert m.input_shape == [-1, 3, 9, 9] and m.output_shape == [-1, 4]
    x = rng.standard_normal((2, 3, 9, 9)).astype(np.float32)
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)))
    conv = np.zeros((2, 4, 5, 5))
    for oy in range(5):
        for ox in range(5):
            patch = xp[:, :, oy * 2:oy * 2 + 3, ox * 2:ox * 2 + 3]
            conv[:, :, oy, ox] = np.einsum("nchw,mchw->nm", patch, w.astype(np.float64)) + bias
    bn = (conv - mu[None, :, None, None]) / np.sqrt(var[None, :, None, None] + 1e-5) * sc[None, :, None, None] + be[None, :, None, None]
    r = np.maximum(bn, 0)
    pool = r[:, :, :4, :4].reshape(2, 4, 2, 2, 2, 2).max(axis=(3, 5))
    want = pool.mean(axis=(2, 3))
    np.testing.assert_allclose(m.predict_blob(x.tobytes()), want, rtol=1e-5, atol=1e-6)


def test_committed_golden_vectors(O, tmp_path):
    from infera_amd import onnx_writer as W, synth

    sys_path = os.path.join(GOLD, "make_golden.py")
    ns = {"__file__": sys_path, "__name__": "golden_cases"}
    exec(compile(open(sys_path).read(), sys_path, "exec"), ns)
    vec = np.load(os.path.join(GOLD, "vectors.npz"))
    for name, (blob, seed, row0, rows, cols) in ns["CASES"].items():
        assert hashlib.sha256(blob).digest() == vec[name + "_sha256"].tobytes(), f"ONNX writer output for {name} drifted"
        assert vec[name + "_meta"].tolist() == [seed, row0, rows, cols]
        y = O.Model(W.write(str(tmp_path / (name + ".onnx")), blob)).predict(synth.table(seed, row0, rows, cols))
        np.testing.assert_allclose(y, vec[name + "_y"], rtol=1e-6, atol=1e-7)
