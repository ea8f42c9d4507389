"""ai.onnx.ml operators: the graphs sklearn exporters write (Scaler -> LinearClassifier / LinearRegressor [-> Normalizer]).

tract-onnx serves these for the reference (engine.rs:49-56).  Its sources are not in /root/reference, so the three
layers below are pinned on the ONNX-ML operator specification: a float64 numpy restatement written here, the C oracle
(oracle/infera_oracle.c op_ml_*) and the product (lowering.cpp ml_node -> Dense / ArgMax / Softmax kernels).
"""
from __future__ import annotations

import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth

RTOL, ATOL = 1e-4, 1e-6


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


def np_pipeline(x, features, classes, kind, post, labels, normalizer, scaler, output, seed=99):
    """float64 restatement of W.sklearn_pipeline from the operator specification (same weight stream)."""
    ws = W._WeightStream(seed)
    x = x.astype(np.float64)
    if scaler:
        off = ws.take((features,), 1).astype(np.float64)
        sc = (1.0 + 0.5 * ws.take((features,), 1)).astype(np.float32).astype(np.float64)
        x = (x - off) * sc
    coef = ws.take((classes, features), features).astype(np.float64)
    icpt = ws.take((classes,), features).astype(np.float64)
    raw = x @ coef.T + icpt

    def transform(s):
        if post == "LOGISTIC":
            return 1.0 / (1.0 + np.exp(-s))
        if post == "SOFTMAX":
            e = np.exp(s - s.max(axis=1, keepdims=True))
            return e / e.sum(axis=1, keepdims=True)
        return s

    if kind == "regressor":
        return transform(raw), raw
    if output == "label":
        lab = np.asarray(labels if labels is not None else range(classes), np.float64)
        return lab[np.argmax(raw, axis=1)], raw
    s = transform(raw)
    if normalizer == "MAX":
        s = s / np.maximum(np.abs(s).max(axis=1, keepdims=True), 1e-30)
    elif normalizer == "L1":
        s = s / np.maximum(np.abs(s).sum(axis=1, keepdims=True), 1e-30)
    elif normalizer == "L2":
        s = s / np.maximum(np.sqrt((s * s).sum(axis=1, keepdims=True)), 1e-30)
    return s, raw


CASES = [
    # features, classes, kind, post, labels, normalizer, scaler, output
    (30, 3, "classifier", "SOFTMAX", None, None, True, "label"),
    (30, 3, "classifier", "SOFTMAX", None, "L1", True, "scores"),
    (13, 2, "classifier", "LOGISTIC", [-1, 1], "L1", True, "label"),
    (13, 2, "classifier", "LOGISTIC", None, "L1", True, "scores"),
    (64, 10, "classifier", "NONE", list(range(1, 11)), "L2", False, "scores"),
    (64, 10, "classifier", "NONE", list(range(10, 110, 10)), None, False, "label"),
    (7, 4, "classifier", "NONE", None, "MAX", True, "scores"),
    (30, 1, "regressor", "NONE", None, None, True, "scores"),
    (128, 5, "regressor", "SOFTMAX", None, None, False, "scores"),
    (9, 2, "regressor", "LOGISTIC", None, None, True, "scores"),
]


def _ids(c):
    f, e, kind, post, labels, norm, scaler, output = c
    return f"{kind}-{f}x{e}-{post}-{'sc' if scaler else 'raw'}-{norm or 'nonorm'}-{output}{'-lab' if labels else ''}"


def _decisive(raw):
    """rows whose two best raw scores are far enough apart that f32 rounding cannot flip the label"""
    s = np.sort(raw, axis=1)
    return (s[:, -1] - s[:, -2]) > 1e-4 * np.maximum(1.0, np.abs(s[:, -1]))


def _model(tmp_path, c):
    f, e, kind, post, labels, norm, scaler, output = c
    blob = W.sklearn_pipeline(f, e, kind, post, labels, norm, scaler, output)
    return W.write(str(tmp_path / "skl.onnx"), blob)


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_oracle_ml_operators_vs_specification(O, tmp_path, case):
    f, e, kind, post, labels, norm, scaler, output = case
    path = _model(tmp_path, case)
    x = synth.table(17, 0, 257, f)
    want, raw = np_pipeline(x, *case)
    got = O.Model(path).predict(x)
    if kind == "classifier" and output == "label":
        assert got.shape == (257, 1) or got.shape == (257,), got.shape
        ok = _decisive(raw)
        assert ok.sum() > 200
        assert np.array_equal(got.reshape(-1)[ok], want[ok].astype(np.float32))
    else:
        assert_close(got, want.astype(np.float32), rtol=3e-5, atol=2e-6)


def test_ml_nodes_lower_onto_the_standard_kernels(built, tmp_path):
    from infera_amd import capi

    def plan_of(case, name):
        capi.load_model(name, _model(tmp_path, case))
        steps = capi.get_plan(name)["plan"]["steps"]
        info = capi.get_model_info(name)
        capi.unload_model(name)
        return [s["kind"] for s in steps], steps, info

    # the Scaler disappears into the classifier's weights; the dead probability branch costs nothing
    kinds, steps, info = plan_of(CASES[0], "ml0")
    assert kinds == ["Dense", "ArgMax"], steps
    assert "Scaler" in steps[0]["origin"] and "LinearClassifier" in steps[0]["origin"], steps[0]["origin"]
    assert info["output_shape"] == [-1]
    # probabilities as output 0: Scaler+scores+Softmax in one Dense(+Softmax) pair, then the L1 Normalizer
    kinds, steps, _ = plan_of(CASES[1], "ml1")
    assert kinds == ["Dense", "Softmax", "Softmax"], steps
    # {-1, +1} labels: the class table is an arithmetic progression -> multiply-add on the index
    kinds, steps, _ = plan_of(CASES[2], "ml2")
    assert kinds == ["Dense", "ArgMax", "BinaryConst", "BinaryConst"], steps
    # regressor with a scaler: one Dense
    kinds, steps, info = plan_of(CASES[7], "ml7")
    assert kinds == ["Dense"] and info["output_shape"] == [-1, 1], steps


def test_ml_unsupported_forms_fail_loudly(built, tmp_path):
    from infera_amd import capi

    coef = [0.5] * 8
    bad = [
        ("strings", W.node("LinearClassifier", ["X"], ["label", "scores"],
                           [W.attr_floats("coefficients", coef), W.attr_strings("classlabels_strings", ["a", "b"])],
                           domain=W.ML_DOMAIN), r"string class labels"),
        ("uneven", W.node("LinearClassifier", ["X"], ["label", "scores"],
                          [W.attr_floats("coefficients", coef + [0.1] * 4), W.attr_ints("classlabels_ints", [0, 1, 5])],
                          domain=W.ML_DOMAIN), r"evenly spaced"),
        ("probit", W.node("LinearClassifier", ["X"], ["scores_unused", "label"],
                          [W.attr_floats("coefficients", coef), W.attr_ints("classlabels_ints", [0, 1]),
                           W.attr_s("post_transform", "PROBIT")], domain=W.ML_DOMAIN), r"post_transform PROBIT"),
        ("coefcount", W.node("LinearRegressor", ["X"], ["label"], [W.attr_floats("coefficients", coef), W.attr_i("targets", 3)],
                             domain=W.ML_DOMAIN), r"coefficients holds 8 values, expected 12"),
        ("tree", W.node("TreeEnsembleClassifier", ["X"], ["label", "scores"], [], domain=W.ML_DOMAIN),
         r"TreeEnsembleClassifier\): unsupported operator"),
        ("otherdomain", W.node("Foo", ["X"], ["label"], [], domain="com.example"), r"operator domain 'com.example'"),
    ]
    for name, nd, pat in bad:
        p = W.write(str(tmp_path / f"{name}.onnx"),
                    W.model(name, [nd], [], [W.value_info("X", ["N", 4])], [W.value_info("label", ["N"], W.INT64 if name in ("strings", "uneven") else W.FLOAT)],
                            ml_opset=1))
        with pytest.raises(capi.InferaError, match=pat):
            capi.load_model("bad_" + name, p)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_gpu_ml_operators_vs_oracle(api, O, tmp_path, case):
    f, e, kind, post, labels, norm, scaler, output = case
    path = _model(tmp_path, case)
    x = synth.table(23, 0, 6007, f)
    want, raw = np_pipeline(x, *case)
    api.load_model("ml", path)
    try:
        got = api.predict("ml", x)
    finally:
        api.unload_model("ml")
    ref = O.Model(path).predict(x)
    assert got.shape == ref.shape
    if kind == "classifier" and output == "label":
        ok = _decisive(raw)
        assert np.array_equal(got.reshape(-1)[ok], ref.reshape(-1)[ok])
        assert np.array_equal(got.reshape(-1)[ok], want[ok].astype(np.float32))
    else:
        assert_close(got, ref)
        assert_close(got, want.astype(np.float32), rtol=3e-5, atol=2e-6)


def _classifier(tmp_path, features, coef, icpt):
    e = coef.shape[0]
    nd = W.node("LinearClassifier", ["X"], ["label", "scores"],
                [W.attr_floats("coefficients", coef.ravel()), W.attr_floats("intercepts", icpt),
                 W.attr_ints("classlabels_ints", range(e))], domain=W.ML_DOMAIN)
    blob = W.model("clf", [nd], [], [W.value_info("X", ["N", features])], [W.value_info("label", ["N"], W.INT64)], ml_opset=1)
    return W.write(str(tmp_path / f"clf_{features}x{e}.onnx"), blob)


# (features, classes, rows, expected exec of the Dense step): the label is picked in the epilogue of the skinny kernel
# (few / odd features), of the two 16x16x4 streaming kernels (staged for big scans, direct below 4096 rows), and by the
# wide-table kernel (rows longer than 128 floats of any alignment), of the one-layer chain kernel (17..128 classes over up
# to 128 columns), and by the stand-alone ArgMax kernel behind every other Dense kernel
ARGMAX_CASES = [(30, 3, 5001, "dense_argmax"), (5, 2, 257, "dense_argmax"), (100, 16, 3000, "dense_argmax"),
                (64, 10, 20011, "dense_argmax"), (128, 7, 9000, "dense_argmax"), (256, 16, 4099, "dense_argmax"),
                (64, 10, 1000, "dense_argmax"), (48, 5, 7001, "dense_argmax"), (1024, 3, 513, "dense_argmax"),
                (200, 6, 2500, "dense_argmax"), (561, 6, 3001, "dense_argmax"), (301, 16, 777, "dense_argmax"),
                (2000, 4, 600, "dense_argmax"), (64, 20, 2500, "chain_fused"), (32, 40, 2500, "chain_fused"), (100, 128, 1500, "chain_fused"),
                (300, 40, 2500, "dense_argmax"), (301, 40, 2500, "dense_argmax"), (561, 64, 900, "dense_argmax"), (300, 100, 2500, "dense_tiled"), (300, 26, 2500, "dense_argmax"), (2050, 32, 700, "dense_argmax")]


@pytest.mark.gpu
@pytest.mark.parametrize("features,classes,rows,exec_kind", ARGMAX_CASES)
def test_gpu_label_in_the_dense_epilogue(api, O, tmp_path, features, classes, rows, exec_kind):
    ws = W._WeightStream(7 + features)
    coef, icpt = ws.take((classes, features), features), ws.take((classes,), features)
    path = _classifier(tmp_path, features, coef, icpt)
    x = synth.table(5, 0, rows, features)
    raw = x.astype(np.float64) @ coef.astype(np.float64).T + icpt
    api.load_model("clf", path)
    try:
        assert api.get_plan("clf")["exec"][0] == exec_kind, api.get_plan("clf")["exec"]
        got = api.predict("clf", x).reshape(-1)
    finally:
        api.unload_model("clf")
    ref = O.Model(path).predict(x).reshape(-1)
    ok = _decisive(raw)
    assert ok.sum() > 0.9 * rows
    assert np.array_equal(got[ok], ref[ok])
    assert np.array_equal(got[ok], np.argmax(raw, axis=1)[ok].astype(np.float32))
    assert set(np.unique(got)) <= set(range(classes))


@pytest.mark.gpu
@pytest.mark.parametrize("features,rows", [(30, 3001), (64, 8192), (48, 700)])
def test_gpu_label_ties_go_to_the_first_class(api, O, tmp_path, features, rows):
    ws = W._WeightStream(3)
    a, b = ws.take((features,), features), ws.take((features,), features)
    coef = np.stack([a, b, b, a, a, b])  # classes 0,3,4 always tie, and so do 1,2,5: only 0 or 1 may ever win
    path = _classifier(tmp_path, features, coef, np.zeros(6, np.float32))
    x = synth.table(9, 0, rows, features)
    api.load_model("tie", path)
    try:
        got = api.predict("tie", x).reshape(-1)
    finally:
        api.unload_model("tie")
    assert set(np.unique(got)) == {0.0, 1.0}
    raw = x.astype(np.float64) @ coef[:2].astype(np.float64).T
    ok = _decisive(raw)
    assert np.array_equal(got[ok], np.argmax(raw, axis=1)[ok].astype(np.float32))
    assert np.array_equal(got[ok], O.Model(path).predict(x).reshape(-1)[ok])


def _mlp_classifier_export(tmp_path, classes_table, k=12, h=20):
    """The graph skl2onnx writes for MLPClassifier: MatMul/Add/Relu, MatMul/Add, Softmax (probabilities, second output),
    ArgMax -> ArrayFeatureExtractor(classes_, index) -> Cast(int64) (label, first output); a feature sub-range is picked
    with ArrayFeatureExtractor first, as ColumnTransformer pipelines do."""
    e = len(classes_table)
    ws = W._WeightStream(21)
    w1, b1, w2, b2 = ws.take((k - 2, h), k), ws.take((h,), k), ws.take((h, e), h), ws.take((e,), h)
    nodes = [
        W.node("ArrayFeatureExtractor", ["X", "cols"], ["Xsel"], domain=W.ML_DOMAIN),
        W.node("MatMul", ["Xsel", "w1"], ["z1"]), W.node("Add", ["z1", "b1"], ["a1"]), W.node("Relu", ["a1"], ["h1"]),
        W.node("MatMul", ["h1", "w2"], ["z2"]), W.node("Add", ["z2", "b2"], ["scores"]),
        W.node("Softmax", ["scores"], ["probabilities"], [W.attr_i("axis", 1)]),
        W.node("ArgMax", ["probabilities"], ["index"], [W.attr_i("axis", 1), W.attr_i("keepdims", 0)]),
        W.node("ArrayFeatureExtractor", ["classes", "index"], ["picked"], domain=W.ML_DOMAIN),
        W.node("Cast", ["picked"], ["label"], [W.attr_i("to", W.INT64)]),
    ]
    inits = [W.tensor("cols", np.arange(1, k - 1, dtype=np.int64)), W.tensor("w1", w1), W.tensor("b1", b1), W.tensor("w2", w2),
             W.tensor("b2", b2), W.tensor("classes", np.asarray(classes_table, np.int64))]
    blob = W.model("mlpc", nodes, inits, [W.value_info("X", ["N", k])],
                   [W.value_info("label", ["N"], W.INT64), W.value_info("probabilities", ["N", e])], ml_opset=1)

    def ref(x):
        hid = np.maximum(x[:, 1:k - 1].astype(np.float64) @ w1 + b1, 0)
        return hid @ w2.astype(np.float64) + b2

    return W.write(str(tmp_path / "mlpc.onnx"), blob), ref


@pytest.mark.parametrize("classes_table", [[0, 1, 2, 3], [3, 5, 7]])
def test_exported_mlp_classifier_label_path(O, built, tmp_path, classes_table):
    from infera_amd import capi

    path, ref = _mlp_classifier_export(tmp_path, classes_table)
    x = synth.table(2, 0, 300, 12)
    raw = ref(x)
    ok = _decisive(raw)
    want = np.asarray(classes_table, np.float32)[np.argmax(raw, axis=1)]
    got = O.Model(path).predict(x).reshape(-1)
    assert np.array_equal(got[ok], want[ok])
    capi.load_model("mlpc", path)
    kinds = [s["kind"] for s in capi.get_plan("mlpc")["plan"]["steps"]]
    capi.unload_model("mlpc")
    # column pick, two layers, softmax (ArgMax reads the probabilities), label; Cast of whole numbers is an alias
    extra = [] if classes_table[0] == 0 and classes_table[1] == 1 else ["BinaryConst", "BinaryConst"]
    assert kinds == ["SliceCols", "Dense", "Dense", "Softmax", "ArgMax"] + extra, kinds


@pytest.mark.gpu
@pytest.mark.parametrize("classes_table", [[0, 1, 2, 3], [3, 5, 7]])
def test_gpu_exported_mlp_classifier_label_path(api, O, tmp_path, classes_table):
    path, ref = _mlp_classifier_export(tmp_path, classes_table)
    x = synth.table(2, 0, 9001, 12)
    raw = ref(x)
    ok = _decisive(raw)
    api.load_model("mlpc", path)
    try:
        got = api.predict("mlpc", x).reshape(-1)
    finally:
        api.unload_model("mlpc")
    assert np.array_equal(got[ok], np.asarray(classes_table, np.float32)[np.argmax(raw, axis=1)][ok])
    assert np.array_equal(got[ok], O.Model(path).predict(x).reshape(-1)[ok])
