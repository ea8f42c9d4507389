"""ONNX front-end breadth: elementwise operators beyond the reference's fixtures, Concat,
ReduceMean, ArgMax with an int64 output, depthwise convolution and the exporter's Shape->Gather->Concat->Reshape
idiom.  The reference pins none of these (its fixtures are MatMul+Add and Identity): "parity unpinned" -- the
ONNX operator specification is the authority.

CPU part: the ORACLE against an independent float64 numpy evaluation, and the product's lowering (models load
and lower without a GPU).  GPU part (`-m gpu`): the HIP path against the oracle through the C ABI.
"""
import math
import os

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
def paths(tmp_path_factory, built):
    d = tmp_path_factory.mktemp("breadth")
    return {
        "zoo": W.write(str(d / "zoo.onnx"), W.unary_zoo(16)),
        "exporter": W.write(str(d / "exporter.onnx"), W.exporter_reshape()),
        "concat": W.write(str(d / "concat.onnx"), W.concat_heads(24)),
        "mnv2": W.write(str(d / "mnv2.onnx"), W.mobilenet_v2(classes=20, in_hw=32, width_mult=0.5)),
        "se": W.write(str(d / "se.onnx"), W.se_net()),
    }


# ---- independent numpy restatements (float64) -------------------------------------------------------------
def _weights(seed):
    return W._WeightStream(seed)


def np_unary_zoo(x, features=16):
    ws = _weights(77)
    w, b = ws.take((features, features), features).astype(np.float64), ws.take((features,), features).astype(np.float64)
    slope = (0.05 + 0.2 * np.abs(ws.take((features,), 1))).astype(np.float32).astype(np.float64)
    h = x.astype(np.float64) @ w + b
    h = h.astype(np.float32).astype(np.float64)  # the graph's intermediate is f32
    hpos = np.abs(h) + 1.5
    erf = np.vectorize(math.erf)
    lo = -0.25
    outs = [np.exp(h), -h, np.abs(h), np.log(np.exp(h) + 1), h * np.clip(h / 6 + 0.5, 0, 1), erf(h), np.floor(h), np.ceil(h),
            h / (1 + np.abs(h)), np.rint(h), 1 / (1 + np.exp(-h)), np.tanh(h), np.maximum(h, 0),
            np.where(h >= 0, h, 0.7 * (np.exp(h) - 1)),
            np.where(h > 0, 1.05070102214813232421875 * h, 1.05070102214813232421875 * (1.67326319217681884765625 * np.exp(h) - 1.67326319217681884765625)),
            np.clip(np.float64(np.float32(0.3)) * h + np.float64(np.float32(0.4)), 0, 1), np.where(h >= 0, h, np.float64(np.float32(0.2)) * h),
            np.log(hpos), np.sqrt(hpos), 1 / hpos, h ** 2, np.minimum(h, lo), np.maximum(h, lo), np.where(h >= 0, h, slope * h),
            np.maximum(lo, h)]
    return np.concatenate(outs, axis=1)


def np_conv2d(x, w, b, stride=1, pad=0, groups=1):
    n, c, hh, ww = x.shape
    m, cg, kh, kw = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(2, 3))[:, :, ::stride, ::stride]  # n c oh ow kh kw
    mg = m // groups
    out = np.empty((n, m, win.shape[2], win.shape[3]))
    for g in range(groups):
        out[:, g * mg:(g + 1) * mg] = np.einsum("ncxykl,mckl->nmxy", win[:, g * cg:(g + 1) * cg], w[g * mg:(g + 1) * mg])
    return out + (0 if b is None else b.reshape(1, -1, 1, 1))


def np_exporter(x):
    ws = _weights(91)
    wc, bc = ws.take((8, 3, 3, 3), 27).astype(np.float64), ws.take((8,), 27).astype(np.float64)
    wf, bf = ws.take((8 * 36, 5), 8 * 36).astype(np.float64), ws.take((5,), 8 * 36).astype(np.float64)
    r = np.maximum(np_conv2d(x.astype(np.float64), wc, bc, 1, 1), 0)
    logits = r.reshape(x.shape[0], -1) @ wf + bf
    return logits


def np_concat_heads(x, features=24):
    ws = _weights(55)
    x = x.astype(np.float64)
    towers = []
    for t, m in enumerate((12, 20)):
        w, b = ws.take((features, m), features).astype(np.float64), ws.take((m,), features).astype(np.float64)
        h = x @ w + b
        towers.append(np.tanh(h) if t else np.maximum(h, 0))
    cat = np.concatenate([towers[0], x, towers[1]], axis=1)
    k = 12 + features + 20
    w, b = ws.take((k, 7), k).astype(np.float64), ws.take((7,), k).astype(np.float64)
    z = cat @ w + b
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


# ---- CPU: the oracle against numpy --------------------------------------------------------------------------
def test_oracle_elementwise_zoo_vs_numpy(O, paths):
    x = synth.table(5, 0, 33, 16)
    got = O.Model(paths["zoo"]).predict(x)
    want = np_unary_zoo(x)
    assert got.shape == want.shape == (33, 16 * 25)
    # floor/ceil/round branches are discontinuous: compare those exactly where the f64 value is not within 1e-5 of a step
    assert_close(got, want.astype(np.float32), rtol=2e-5, atol=2e-6)


def test_oracle_exporter_idiom_and_argmax_vs_numpy(O, paths):
    m = O.Model(paths["exporter"])
    assert m.input_shape == [-1, 3, 6, 6] and m.output_shape == [-1, 1]
    x = synth.table(8, 0, 7, 3 * 6 * 6).reshape(7, 3, 6, 6)
    logits = np_exporter(x)
    top2 = np.sort(logits, axis=1)[:, -2:]
    assert (top2[:, 1] - top2[:, 0]).min() > 1e-4  # labels are well separated on this input
    got = m.predict_blob(x.tobytes())
    assert got.dtype == np.float32 and got.shape == (7, 1)
    assert got[:, 0].tolist() == logits.argmax(axis=1).astype(np.float32).tolist()


def test_oracle_concat_vs_numpy(O, paths):
    x = synth.table(9, 0, 17, 24)
    assert_close(O.Model(paths["concat"]).predict(x), np_concat_heads(x).astype(np.float32), rtol=2e-5)


@pytest.mark.parametrize("groups,stride,pad", [(1, 1, 1), (4, 2, 1), (8, 1, 0)])
def test_oracle_grouped_conv_vs_numpy(O, tmp_path, groups, stride, pad):
    rng = np.random.default_rng(groups * 10 + stride)
    w = (rng.standard_normal((8, 8 // groups, 3, 3)) * 0.3).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    x = rng.standard_normal((2, 8, 9, 9)).astype(np.float32)
    want = np_conv2d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), stride, pad, groups)
    nodes = [W.node("Conv", ["X", "w", "b"], ["Y"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("strides", [stride] * 2),
                                                     W.attr_ints("pads", [pad] * 4), W.attr_i("group", groups)])]
    blob = W.model("gconv", nodes, [W.tensor("w", w), W.tensor("b", b)], [W.value_info("X", ["N", 8, 9, 9])],
                   [W.value_info("Y", ["N", 8, want.shape[2], want.shape[3]])])
    got = O.Model(W.write(str(tmp_path / "gconv.onnx"), blob)).predict_blob(x.tobytes())
    assert_close(got.reshape(want.shape), want.astype(np.float32), rtol=2e-5, atol=2e-6)


def test_oracle_mobilenet_runs_and_is_batch_consistent(O, paths):
    m = O.Model(paths["mnv2"])
    assert m.input_shape == [-1, 3, 32, 32] and m.output_shape == [-1, 20]
    x = synth.table(4, 0, 3, 3 * 32 * 32)
    y = m.predict_blob(x.tobytes())
    assert y.shape == (3, 20) and np.isfinite(y).all() and np.abs(y).max() > 1e-3
    y1 = m.predict_blob(x[1:2].tobytes())
    assert np.array_equal(y1[0], y[1])  # rows are independent: batching must not change a row's value


def test_oracle_se_gate_broadcast_vs_numpy(O, paths):
    ws = _weights(31)
    c, hw = 16, 10
    x = synth.table(6, 0, 3, 4 * hw * hw).reshape(3, 4, hw, hw)

    def take_conv(cin, cout, k):
        return ws.take((cout, cin, k, k), cin * k * k).astype(np.float64), ws.take((cout,), cin * k * k).astype(np.float64)

    w, b = take_conv(4, c, 3)
    s = np.maximum(np_conv2d(x.astype(np.float64), w, b, 1, 1), 0)
    sq = s.mean(axis=(2, 3), keepdims=True)
    w, b = take_conv(c, c // 4, 1)
    r = np.maximum(np_conv2d(sq, w, b), 0)
    w, b = take_conv(c // 4, c, 1)
    e = np.clip(np.float64(np.float32(0.2)) * np_conv2d(r, w, b) + 0.5, 0, 1)
    tot = (s * e + e * s).mean(axis=(2, 3))
    wf, bf = ws.take((c, 6), c).astype(np.float64), ws.take((6,), c).astype(np.float64)
    want = tot @ wf + bf
    assert_close(O.Model(paths["se"]).predict_blob(x.tobytes()), want.astype(np.float32), rtol=2e-5, atol=2e-6)


def _pool_model(tmp_path, name, op, hw, k, stride, pad, ceil, extra=()):
    import math

    num = hw + 2 * pad - k
    o = (math.ceil(num / stride) if ceil else num // stride) + 1
    if ceil and (o - 1) * stride >= hw + pad:
        o -= 1
    nodes = [W.node(op, ["X"], ["P"], [W.attr_ints("kernel_shape", [k, k]), W.attr_ints("strides", [stride] * 2), W.attr_ints("pads", [pad] * 4),
                                       W.attr_i("ceil_mode", int(ceil))] + list(extra)),
             W.node("Constant", [], ["two"], [W.attr_f("value_float", 2.0)]), W.node("Mul", ["P", "two"], ["Y"])]
    blob = W.model(name, nodes, [], [W.value_info("X", ["N", 4, hw, hw])], [W.value_info("Y", ["N", 4, o, o])])
    return W.write(str(tmp_path / f"{name}.onnx"), blob), o


def np_pool(x, k, stride, pad, o, is_max):
    n, c, h, w = x.shape
    out = np.empty((n, c, o, o))
    for oy in range(o):
        for ox in range(o):
            ys, xs = oy * stride - pad, ox * stride - pad
            win = x[:, :, max(ys, 0):min(ys + k, h), max(xs, 0):min(xs + k, w)]
            out[:, :, oy, ox] = win.max(axis=(2, 3)) if is_max else win.mean(axis=(2, 3))
    return out


POOL_CASES = [("MaxPool", 7, 3, 2, 0, True), ("MaxPool", 8, 3, 2, 1, True), ("AveragePool", 7, 2, 2, 0, True), ("MaxPool", 9, 2, 2, 0, True),
              ("AveragePool", 8, 3, 2, 1, False), ("MaxPool", 6, 3, 1, 1, False), ("MaxPool", 5, 2, 1, 0, False), ("MaxPool", 10, 3, 3, 1, True)]


@pytest.mark.parametrize("op,hw,k,stride,pad,ceil", POOL_CASES)
def test_oracle_pool_ceil_mode_vs_numpy(O, tmp_path, op, hw, k, stride, pad, ceil):
    path, o = _pool_model(tmp_path, "pl", op, hw, k, stride, pad, ceil)
    x = synth.table(12, 0, 2, 4 * hw * hw).reshape(2, 4, hw, hw)
    m = O.Model(path)
    assert m.output_shape == [-1, 4, o, o]
    got = m.predict_blob(x.tobytes()).reshape(2, 4, o, o)
    assert_close(got, (2 * np_pool(x.astype(np.float64), k, stride, pad, o, op == "MaxPool")).astype(np.float32), rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("op,hw,k,stride,pad,ceil", POOL_CASES)
def test_gpu_pool_ceil_mode(api, O, tmp_path, op, hw, k, stride, pad, ceil):
    path, o = _pool_model(tmp_path, "plg", op, hw, k, stride, pad, ceil)
    x = synth.table(12, 0, 5, 4 * hw * hw)
    api.load_model("plg", path)
    assert api.get_model_info("plg")["output_shape"] == [-1, 4, o, o]
    assert_close(api.predict_from_blob("plg", x.tobytes()), O.Model(path).predict_blob(x.tobytes()))
    api.unload_model("plg")


def _two_output_model(tmp_path):
    """A classifier that exposes (label, probabilities): only output 0 (the label) is served; the second output's
    branch holds an operator nobody supports (ai.onnx.ml ZipMap) and must be ignored, not rejected."""
    rng = np.random.default_rng(4)
    w, b = (rng.standard_normal((12, 5)) * 0.5).astype(np.float32), rng.standard_normal(5).astype(np.float32)
    nodes = [W.node("Gemm", ["X", "w", "b"], ["logits"]), W.node("ArgMax", ["logits"], ["label"], [W.attr_i("axis", 1), W.attr_i("keepdims", 0)]),
             W.node("Softmax", ["logits"], ["probs"], [W.attr_i("axis", 1)]), W.node("ZipMap", ["probs"], ["prob_map"])]
    blob = W.model("two_out", nodes, [W.tensor("w", w), W.tensor("b", b)], [W.value_info("X", ["N", 12])],
                   [W.value_info("label", ["N"], W.INT64), W.value_info("prob_map", ["N", 5])])
    return W.write(str(tmp_path / "two_out.onnx"), blob), w, b


def test_dead_branches_and_extra_outputs_are_ignored(O, built, tmp_path):
    from infera_amd import capi

    path, w, b = _two_output_model(tmp_path)
    x = synth.table(2, 0, 50, 12)
    want = (x.astype(np.float64) @ w + b).argmax(axis=1).astype(np.float32)
    m = O.Model(path)
    assert m.output_shape == [-1]
    got = m.predict(x)
    assert got.ravel().tolist() == want.tolist()
    capi.load_model("two_out", path)
    assert [s["kind"] for s in capi.get_plan("two_out")["plan"]["steps"]] == ["Dense", "ArgMax"]
    assert capi.get_model_info("two_out")["output_shape"] == [-1]
    capi.unload_model("two_out")


@pytest.mark.gpu
def test_gpu_label_output_of_a_two_output_classifier(api, O, tmp_path):
    path, w, b = _two_output_model(tmp_path)
    x = synth.table(2, 0, 3000, 12)
    api.load_model("two_out", path)
    got, want = api.predict("two_out", x), O.Model(path).predict(x)
    assert got.shape == want.shape
    assert (got != want).mean() < 0.005
    api.unload_model("two_out")


def _wide_and_deep(tmp_path):
    """Two runtime inputs (dense features [N,10], wide features [N,6]) -- split across the call's 16 feature columns
    in declaration order -- plus Split and Slice on the feature axis."""
    rng = np.random.default_rng(8)
    w1, b1 = (rng.standard_normal((10, 12)) * 0.4).astype(np.float32), rng.standard_normal(12).astype(np.float32)
    w2 = (rng.standard_normal((6 + 6 + 4, 3)) * 0.4).astype(np.float32)
    nodes = [W.node("Gemm", ["dense", "w1", "b1"], ["h"]), W.node("Relu", ["h"], ["hr"]),
             W.node("Split", ["hr"], ["ha", "hb"], [W.attr_i("axis", 1)]),                       # 12 -> 6 + 6
             W.node("Slice", ["wide", "s0", "s4", "ax1"], ["wcut"]),                              # wide[:, 0:4]
             W.node("Mul", ["ha", "hb"], ["hm"]),
             W.node("Concat", ["hm", "wide", "wcut"], ["cat"], [W.attr_i("axis", 1)]),
             W.node("MatMul", ["cat", "w2"], ["Y"])]
    inits = [W.tensor("w1", w1), W.tensor("b1", b1), W.tensor("w2", w2), W.tensor("s0", np.array([0], np.int64)),
             W.tensor("s4", np.array([4], np.int64)), W.tensor("ax1", np.array([1], np.int64))]
    blob = W.model("wide_deep", nodes, inits, [W.value_info("dense", ["N", 10]), W.value_info("wide", ["N", 6])], [W.value_info("Y", ["N", 3])])

    def ref(x):
        x = x.astype(np.float64)
        d, wd = x[:, :10], x[:, 10:]
        h = np.maximum(d @ w1 + b1, 0)
        return np.concatenate([h[:, :6] * h[:, 6:], wd, wd[:, :4]], axis=1) @ w2

    return W.write(str(tmp_path / "wide_deep.onnx"), blob), ref


def test_multi_input_split_slice_oracle_and_lowering(O, built, tmp_path):
    from infera_amd import capi

    path, ref = _wide_and_deep(tmp_path)
    x = synth.table(31, 0, 40, 16)
    m = O.Model(path)
    assert m.input_shape == [-1, 16] and m.output_shape == [-1, 3]
    assert_close(m.predict(x), ref(x).astype(np.float32), rtol=2e-5, atol=2e-6)
    with pytest.raises(O.OracleError, match=r"^Invalid input shape: expected batch x \[16\], got 40 x 10$"):
        m.predict(x[:, :10])
    capi.load_model("wd", path)
    assert capi.get_model_info("wd")["input_shape"] == [-1, 16]
    kinds = [s["kind"] for s in capi.get_plan("wd")["plan"]["steps"]]
    assert kinds.count("SliceCols") == 5 and kinds.count("CopyCols") == 3, kinds  # 2 inputs + 2 split pieces + 1 slice
    capi.unload_model("wd")


@pytest.mark.gpu
def test_gpu_multi_input_split_slice(api, O, tmp_path):
    path, _ = _wide_and_deep(tmp_path)
    x = synth.table(31, 0, 3001, 16)
    api.load_model("wd", path)
    assert_close(api.predict("wd", x), O.Model(path).predict(x))
    with pytest.raises(api.InferaError, match=r"^Invalid input shape: expected batch x \[16\], got 3001 x 10$"):
        api.predict("wd", x[:, :10].copy())
    api.unload_model("wd")


def _scaler_pipeline(tmp_path, k=30, m=1):
    """sklearn-style Pipeline(StandardScaler, MinMax-ish rescale, LogisticRegression): Sub(mean) -> Div(std) -> Mul -> Add -> Gemm -> Sigmoid."""
    rng = np.random.default_rng(21)
    mean, std = rng.standard_normal(k).astype(np.float32), (0.5 + rng.random(k)).astype(np.float32)
    sc, sh = (0.5 + rng.random(k)).astype(np.float32), rng.standard_normal(k).astype(np.float32) * 0.1
    w, b = (rng.standard_normal((k, m)) * 0.3).astype(np.float32), rng.standard_normal(m).astype(np.float32)
    nodes = [W.node("Sub", ["X", "mean"], ["c"]), W.node("Div", ["c", "std"], ["z"]), W.node("Mul", ["sc", "z"], ["u"]), W.node("Add", ["u", "sh"], ["v"]),
             W.node("Gemm", ["v", "w", "b"], ["logit"]), W.node("Sigmoid", ["logit"], ["Y"])]
    inits = [W.tensor("mean", mean), W.tensor("std", std), W.tensor("sc", sc), W.tensor("sh", sh), W.tensor("w", w), W.tensor("b", b)]
    blob = W.model("scaler_logreg", nodes, inits, [W.value_info("X", ["N", k])], [W.value_info("Y", ["N", m])])

    def ref(x):
        v = ((x.astype(np.float64) - mean) / std) * sc + sh
        return 1 / (1 + np.exp(-(v @ w + b)))

    return W.write(str(tmp_path / "scaler_logreg.onnx"), blob), ref


def test_scaler_pipeline_folds_into_the_linear_layer(O, built, tmp_path):
    from infera_amd import capi

    path, ref = _scaler_pipeline(tmp_path)
    x = synth.table(41, 0, 64, 30)
    assert_close(O.Model(path).predict(x), ref(x).astype(np.float32), rtol=2e-5, atol=2e-6)
    capi.load_model("sp", path)
    steps = capi.get_plan("sp")["plan"]["steps"]
    assert [s["kind"] for s in steps] == ["Dense"], steps  # four elementwise passes over the table folded into W and b
    assert steps[0]["origin"] == "Sub+Div+Mul+Add+Gemm+Sigmoid", steps[0]["origin"]
    capi.unload_model("sp")


@pytest.mark.gpu
def test_gpu_scaler_pipeline(api, O, tmp_path):
    path, ref = _scaler_pipeline(tmp_path)
    x = synth.table(41, 0, 5003, 30)
    api.load_model("sp", path)
    got = api.predict("sp", x)
    assert_close(got, O.Model(path).predict(x))
    assert_close(got, ref(x).astype(np.float32), rtol=2e-5, atol=2e-6)
    api.unload_model("sp")


def _keras_bn_mlp(tmp_path, k=20, h=32, m=3):
    """Keras-style tabular MLP: BatchNormalization(features) -> Dense -> BatchNormalization -> Relu -> Dense -> Softmax"""
    rng = np.random.default_rng(5)

    def bn(c, tag):
        g, b = (0.5 + rng.random(c)).astype(np.float32), rng.standard_normal(c).astype(np.float32) * 0.2
        mu, var = rng.standard_normal(c).astype(np.float32) * 0.3, (0.3 + rng.random(c)).astype(np.float32)
        inits = [W.tensor(f"g{tag}", g), W.tensor(f"b{tag}", b), W.tensor(f"mu{tag}", mu), W.tensor(f"var{tag}", var)]
        return inits, lambda x: (x - mu) / np.sqrt(var.astype(np.float64) + 1e-3) * g + b

    i0, f0 = bn(k, "0")
    i1, f1 = bn(h, "1")
    w1, c1 = (rng.standard_normal((k, h)) * 0.3).astype(np.float32), rng.standard_normal(h).astype(np.float32) * 0.1
    w2, c2 = (rng.standard_normal((h, m)) * 0.3).astype(np.float32), rng.standard_normal(m).astype(np.float32) * 0.1
    eps = [W.attr_f("epsilon", 1e-3)]
    nodes = [W.node("BatchNormalization", ["X", "g0", "b0", "mu0", "var0"], ["x0"], eps),
             W.node("Gemm", ["x0", "w1", "c1"], ["z1"]),
             W.node("BatchNormalization", ["z1", "g1", "b1", "mu1", "var1"], ["n1"], eps), W.node("Relu", ["n1"], ["a1"]),
             W.node("Gemm", ["a1", "w2", "c2"], ["z2"]), W.node("Softmax", ["z2"], ["Y"], [W.attr_i("axis", 1)])]
    inits = i0 + i1 + [W.tensor("w1", w1), W.tensor("c1", c1), W.tensor("w2", w2), W.tensor("c2", c2)]
    blob = W.model("keras_bn", nodes, inits, [W.value_info("X", ["N", k])], [W.value_info("Y", ["N", m])])

    def ref(x):
        a = np.maximum(f1(f0(x.astype(np.float64)) @ w1 + c1), 0) @ w2 + c2
        e = np.exp(a - a.max(axis=1, keepdims=True))
        return e / e.sum(axis=1, keepdims=True)

    return W.write(str(tmp_path / "keras_bn.onnx"), blob), ref


def test_feature_batchnorm_folds_into_the_dense_layers(O, built, tmp_path):
    from infera_amd import capi

    path, ref = _keras_bn_mlp(tmp_path)
    x = synth.table(43, 0, 64, 20)
    assert_close(O.Model(path).predict(x), ref(x).astype(np.float32), rtol=3e-5, atol=2e-6)
    capi.load_model("kbn", path)
    steps = capi.get_plan("kbn")["plan"]["steps"]
    capi.unload_model("kbn")
    assert [s["kind"] for s in steps] == ["Dense", "Dense", "Softmax"], steps  # both BatchNormalizations are in W1 / b1
    assert steps[0]["origin"] == "BatchNormalization+Gemm+BatchNormalization+Relu", steps[0]["origin"]


@pytest.mark.gpu
def test_gpu_feature_batchnorm_mlp(api, O, tmp_path):
    path, ref = _keras_bn_mlp(tmp_path)
    x = synth.table(43, 0, 5003, 20)
    api.load_model("kbn", path)
    got = api.predict("kbn", x)
    api.unload_model("kbn")
    assert_close(got, O.Model(path).predict(x))
    assert_close(got, ref(x).astype(np.float32), rtol=3e-5, atol=2e-6)


def np_zoo_ops(x, wts):
    """float64 restatement of W.zoo_ops_net from the operator specifications"""
    relu = lambda v: np.maximum(v, 0)
    c = lambda name, v, **kw: np_conv2d(v, wts[name][0].astype(np.float64), wts[name][1].astype(np.float64), **kw)
    a = relu(c("c1", x.astype(np.float64), pad=1))
    sq = np.pad(a * a, ((0, 0), (2, 2), (0, 0), (0, 0)))
    win = sum(sq[:, k:k + a.shape[1]] for k in range(5))  # channels c-2 .. c+2
    a = a / (1.5 + 0.05 / 5 * win) ** 0.75
    a = np_pool(a, 2, 2, 0, a.shape[2] // 2, True)
    a = np.pad(a, ((0, 0), (0, 0), (0, 2), (1, 1)))
    a = relu(c("c2", a, stride=2))
    keep, work = a[:, :16], a[:, 16:]
    b = relu(c("b1", work))
    b = c("b2", b, pad=1, groups=16)
    b = relu(c("b3", b))
    cat = np.concatenate([keep, b], axis=1)
    n, _, h, w = cat.shape
    shuf = cat.reshape(n, 2, 16, h, w).transpose(0, 2, 1, 3, 4).reshape(n, 32, h, w)
    tot = c("s1", shuf[:, 8:24]) + c("s2", shuf) + shuf
    g = tot.max(axis=(2, 3))
    return g @ wts["fc"][0].astype(np.float64) + wts["fc"][1]


def test_model_zoo_operator_batch(O, built, tmp_path):
    """LRN, Pad folded into a VALID Conv, channel Split / Slice, channel shuffle, Sum, GlobalMaxPool"""
    from infera_amd import capi

    blob, wts = W.zoo_ops_net()
    path = W.write(str(tmp_path / "zoo_ops.onnx"), blob)
    x = synth.table(77, 0, 5, 3 * 16 * 16)
    assert_close(O.Model(path).predict_blob(x.tobytes()), np_zoo_ops(x.reshape(5, 3, 16, 16), wts).astype(np.float32), rtol=1e-4, atol=2e-6)
    capi.load_model("zoo_ops", path)
    steps = capi.get_plan("zoo_ops")["plan"]["steps"]
    capi.unload_model("zoo_ops")
    kinds = [s["kind"] for s in steps]
    assert kinds.count("LRN") == 1 and kinds.count("ChannelShuffle") == 1 and kinds.count("SliceCols") == 3, kinds
    assert "Pad" not in " ".join(s["origin"] for s in steps) and kinds.count("BinaryAct") == 2, kinds  # Pad is in c2's padding; Sum = two adds
    bad = [W.node("Pad", ["X", "pads"], ["p"]), W.node("Relu", ["p"], ["Y"])]
    p2 = W.write(str(tmp_path / "badpad.onnx"),
                 W.model("badpad", bad, [W.tensor("pads", np.array([0, 0, 1, 1, 0, 0, 1, 1], np.int64))],
                         [W.value_info("X", ["N", 3, 4, 4])], [W.value_info("Y", ["N", 3, 6, 6])]))
    with pytest.raises(capi.InferaError, match="Pad node can only feed a Conv"):
        capi.load_model("badpad", p2)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 9, 130])
def test_gpu_model_zoo_operator_batch(api, O, tmp_path, rows):
    blob, wts = W.zoo_ops_net()
    path = W.write(str(tmp_path / "zoo_ops.onnx"), blob)
    x = synth.table(78, 0, rows, 3 * 16 * 16)
    api.load_model("zoo_ops", path)
    try:
        assert api.get_plan("zoo_ops")["activation_layout"] == "NC/4HW4"
        got = api.predict_from_blob("zoo_ops", x.tobytes())
    finally:
        api.unload_model("zoo_ops")
    assert_close(got, O.Model(path).predict_blob(x.tobytes()))
    assert_close(got, np_zoo_ops(x.reshape(rows, 3, 16, 16), wts).astype(np.float32), rtol=1e-4, atol=2e-6)


def _nchw_lrn_shuffle(tmp_path):
    """6 channels (not whole quads): the plan stays NCHW, so LRN / channel shuffle / GlobalMaxPool run their NCHW paths"""
    ws = W._WeightStream(9)
    w, b = ws.take((6, 3, 3, 3), 27), ws.take((6,), 27)
    fw, fb = ws.take((6, 4), 6), ws.take((4,), 6)
    nodes = [W.node("Conv", ["X", "w", "b"], ["c"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("pads", [1, 1, 1, 1])]),
             W.node("LRN", ["c"], ["l"], [W.attr_i("size", 3), W.attr_f("alpha", 0.1)]),
             W.node("Reshape", ["l", "s5"], ["r5"]), W.node("Transpose", ["r5"], ["t5"], [W.attr_ints("perm", [0, 2, 1, 3, 4])]),
             W.node("Reshape", ["t5", "s4"], ["sh"]), W.node("GlobalMaxPool", ["sh"], ["g"]), W.node("Flatten", ["g"], ["f"]),
             W.node("Gemm", ["f", "fw", "fb"], ["Y"])]
    inits = [W.tensor("w", w), W.tensor("b", b), W.tensor("fw", fw), W.tensor("fb", fb),
             W.tensor("s5", np.array([0, 3, 2, 5, 5], np.int64)), W.tensor("s4", np.array([0, 6, 5, 5], np.int64))]
    blob = W.model("nchw_ops", nodes, inits, [W.value_info("X", ["N", 3, 5, 5])], [W.value_info("Y", ["N", 4])])

    def ref(x):
        a = np_conv2d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), pad=1)
        sq = np.pad(a * a, ((0, 0), (1, 1), (0, 0), (0, 0)))
        a = a / (1.0 + 0.1 / 3 * sum(sq[:, k:k + 6] for k in range(3))) ** 0.75
        a = a.reshape(-1, 3, 2, 5, 5).transpose(0, 2, 1, 3, 4).reshape(-1, 6, 5, 5)
        return a.max(axis=(2, 3)) @ fw.astype(np.float64) + fb

    return W.write(str(tmp_path / "nchw_ops.onnx"), blob), ref


def test_oracle_nchw_lrn_shuffle_vs_numpy(O, tmp_path):
    path, ref = _nchw_lrn_shuffle(tmp_path)
    x = synth.table(5, 0, 4, 75)
    assert_close(O.Model(path).predict_blob(x.tobytes()), ref(x.reshape(4, 3, 5, 5)).astype(np.float32), rtol=3e-5, atol=2e-6)


@pytest.mark.gpu
def test_gpu_nchw_lrn_shuffle(api, O, tmp_path):
    path, ref = _nchw_lrn_shuffle(tmp_path)
    x = synth.table(5, 0, 300, 75)
    api.load_model("nchw_ops", path)
    try:
        assert api.get_plan("nchw_ops")["activation_layout"] == "NCHW"
        got = api.predict_from_blob("nchw_ops", x.tobytes())
    finally:
        api.unload_model("nchw_ops")
    assert_close(got, O.Model(path).predict_blob(x.tobytes()))
    assert_close(got, ref(x.reshape(300, 3, 5, 5)).astype(np.float32), rtol=3e-5, atol=2e-6)


def _vgg_like(tmp_path, hidden=40):
    """VGG / AlexNet head: conv blocks -> Flatten(C,H,W) -> Gemm -> Relu -> Gemm (no global pooling before the classifier)"""
    ws = W._WeightStream(61)
    w1, b1 = ws.take((8, 3, 3, 3), 27), ws.take((8,), 27)
    w2, b2 = ws.take((16, 8, 3, 3), 72), ws.take((16,), 72)
    f1, g1 = ws.take((16 * 4 * 4, hidden), 256), ws.take((hidden,), 256)
    f2, g2 = ws.take((hidden, 5), hidden), ws.take((5,), hidden)
    cv = lambda x, w, b, o: W.node("Conv", [x, w, b], [o], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("pads", [1, 1, 1, 1])])
    mp = lambda x, o: W.node("MaxPool", [x], [o], [W.attr_ints("kernel_shape", [2, 2]), W.attr_ints("strides", [2, 2])])
    nodes = [cv("X", "w1", "b1", "c1"), W.node("Relu", ["c1"], ["r1"]), mp("r1", "p1"), cv("p1", "w2", "b2", "c2"), W.node("Relu", ["c2"], ["r2"]),
             mp("r2", "p2"), W.node("Flatten", ["p2"], ["flat"], [W.attr_i("axis", 1)]), W.node("Gemm", ["flat", "f1", "g1"], ["h"]),
             W.node("Relu", ["h"], ["hr"]), W.node("Gemm", ["hr", "f2", "g2"], ["Y"])]
    inits = [W.tensor(k, v) for k, v in dict(w1=w1, b1=b1, w2=w2, b2=b2, f1=f1, g1=g1, f2=f2, g2=g2).items()]
    blob = W.model("vgg_like", nodes, inits, [W.value_info("X", ["N", 3, 16, 16])], [W.value_info("Y", ["N", 5])])

    def ref(x):
        a = np.maximum(np_conv2d(x.astype(np.float64), w1.astype(np.float64), b1.astype(np.float64), pad=1), 0)
        a = np_pool(a, 2, 2, 0, 8, True)
        a = np.maximum(np_conv2d(a, w2.astype(np.float64), b2.astype(np.float64), pad=1), 0)
        a = np_pool(a, 2, 2, 0, 4, True).reshape(len(x), -1)
        return np.maximum(a @ f1.astype(np.float64) + g1, 0) @ f2.astype(np.float64) + g2

    return W.write(str(tmp_path / f"vgg_like{hidden}.onnx"), blob), ref


def test_flatten_into_gemm_keeps_the_conv_layout(O, built, tmp_path):
    from infera_amd import capi

    path, ref = _vgg_like(tmp_path)
    x = synth.table(3, 0, 4, 768)
    assert_close(O.Model(path).predict_blob(x.tobytes()), ref(x.reshape(4, 3, 16, 16)).astype(np.float32), rtol=3e-5, atol=2e-6)
    capi.load_model("vgg", path)
    plan = capi.get_plan("vgg")
    capi.unload_model("vgg")
    assert plan["activation_layout"] == "NC/4HW4", plan["activation_layout"]  # the classifier's weight rows were permuted instead
    assert any("channel-quad order" in s["origin"] for s in plan["plan"]["steps"])


@pytest.mark.gpu
@pytest.mark.parametrize("hidden", [8, 40, 100])  # narrow / chain-or-tiled first classifier layers
def test_gpu_flatten_into_gemm(api, O, tmp_path, hidden):
    path, ref = _vgg_like(tmp_path, hidden)
    x = synth.table(3, 0, 70, 768)
    api.load_model("vgg", path)
    try:
        assert api.get_plan("vgg")["activation_layout"] == "NC/4HW4"
        got = api.predict_from_blob("vgg", x.tobytes())
    finally:
        api.unload_model("vgg")
    assert_close(got, O.Model(path).predict_blob(x.tobytes()))
    assert_close(got, ref(x.reshape(70, 3, 16, 16)).astype(np.float32), rtol=3e-5, atol=2e-6)


def _swish_net(tmp_path):
    """EfficientNet-style block: Conv -> x*Sigmoid(x) -> depthwise Conv -> HardSwish -> 1x1 Conv -> Sigmoid*x (other operand
    order) -> GAP -> Gemm -> x*Sigmoid(x) -> Gemm"""
    ws = W._WeightStream(71)
    inits, nodes = [], []

    def conv(x, cin, cout, k, out, groups=1):
        w, b = ws.take((cout, cin // groups, k, k), (cin // groups) * k * k), ws.take((cout,), cin * k * k)
        inits.extend([W.tensor(out + "_w", w), W.tensor(out + "_b", b)])
        nodes.append(W.node("Conv", [x, out + "_w", out + "_b"], [out], [W.attr_ints("kernel_shape", [k, k]), W.attr_ints("pads", [k // 2] * 4),
                                                                        W.attr_i("group", groups)]))
        return out

    conv("X", 3, 32, 3, "c1")
    nodes += [W.node("Sigmoid", ["c1"], ["s1"]), W.node("Mul", ["c1", "s1"], ["a1"])]
    conv("a1", 32, 32, 3, "dw", groups=32)
    nodes.append(W.node("HardSwish", ["dw"], ["a2"]))
    conv("a2", 32, 64, 1, "pw")
    nodes += [W.node("Sigmoid", ["pw"], ["s3"]), W.node("Mul", ["s3", "pw"], ["a3"])]
    nodes += [W.node("GlobalAveragePool", ["a3"], ["g"]), W.node("Flatten", ["g"], ["f"])]
    f1, g1, f2, g2 = ws.take((64, 48), 64), ws.take((48,), 64), ws.take((48, 5), 48), ws.take((5,), 48)
    inits += [W.tensor("f1", f1), W.tensor("g1", g1), W.tensor("f2", f2), W.tensor("g2", g2)]
    nodes += [W.node("Gemm", ["f", "f1", "g1"], ["h"]), W.node("Sigmoid", ["h"], ["hs"]), W.node("Mul", ["h", "hs"], ["ha"]),
              W.node("Gemm", ["ha", "f2", "g2"], ["Y"])]
    blob = W.model("swish_net", nodes, inits, [W.value_info("X", ["N", 3, 12, 12])], [W.value_info("Y", ["N", 5])], opset=14)
    return W.write(str(tmp_path / "swish_net.onnx"), blob)


def test_swish_and_hardswish_fold_into_the_producing_layers(O, built, tmp_path):
    from infera_amd import capi

    path = _swish_net(tmp_path)
    capi.load_model("sw", path)
    steps = capi.get_plan("sw")["plan"]["steps"]
    capi.unload_model("sw")
    assert [s["kind"] for s in steps] == ["Conv2d", "Conv2d", "Conv2d", "GlobalAvgPool", "Dense", "Dense"], steps
    assert [s.get("act", "") for s in steps] == ["Swish", "HardSwish", "Swish", "", "Swish", ""], steps
    x = synth.table(9, 0, 3, 3 * 12 * 12)
    assert np.isfinite(O.Model(path).predict_blob(x.tobytes())).all()


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 50])
def test_gpu_swish_net(api, O, tmp_path, rows):
    path = _swish_net(tmp_path)
    x = synth.table(9, 0, rows, 3 * 12 * 12)
    api.load_model("sw", path)
    try:
        got = api.predict_from_blob("sw", x.tobytes())
    finally:
        api.unload_model("sw")
    assert_close(got, O.Model(path).predict_blob(x.tobytes()))


def _normalised_input_net(tmp_path, variant):
    """in-graph input normalisation in front of the first convolution: (x - mean) / std per channel (torchvision-style
    transforms exported with the model), x * (1/255) (Keras Rescaling), or a BatchNormalization of the input"""
    ws = W._WeightStream(81)
    w1, b1 = ws.take((32, 3, 3, 3), 27), ws.take((32,), 27)
    w2, b2 = ws.take((32, 32, 3, 3), 288), ws.take((32,), 288)
    fw, fb = ws.take((32, 6), 32), ws.take((6,), 32)
    inits = [W.tensor(k, v) for k, v in dict(w1=w1, b1=b1, w2=w2, b2=b2, fw=fw, fb=fb).items()]
    if variant == "meanstd":
        inits += [W.tensor("mean", np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)),
                  W.tensor("std", np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1))]
        pre = [W.node("Sub", ["X", "mean"], ["xc"]), W.node("Div", ["xc", "std"], ["xn"])]
    elif variant == "rescale":
        inits += [W.tensor("k", np.array(1.0 / 255.0, np.float32))]
        pre = [W.node("Mul", ["X", "k"], ["xn"])]
    else:
        inits += [W.tensor("g", np.array([1.1, 0.9, 1.3], np.float32)), W.tensor("b", np.array([0.1, -0.2, 0.0], np.float32)),
                  W.tensor("mu", np.array([0.2, 0.1, -0.1], np.float32)), W.tensor("var", np.array([0.5, 1.5, 1.0], np.float32))]
        pre = [W.node("BatchNormalization", ["X", "g", "b", "mu", "var"], ["xn"])]
    cv = lambda x, w, b, o, s=1: W.node("Conv", [x, w, b], [o], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("pads", [1, 1, 1, 1]), W.attr_ints("strides", [s, s])])
    nodes = pre + [cv("xn", "w1", "b1", "c1", 2), W.node("Relu", ["c1"], ["r1"]), cv("r1", "w2", "b2", "c2"), W.node("Relu", ["c2"], ["r2"]),
                   W.node("GlobalAveragePool", ["r2"], ["g2"]), W.node("Flatten", ["g2"], ["f"]), W.node("Gemm", ["f", "fw", "fb"], ["Y"])]
    blob = W.model("norm_in", nodes, inits, [W.value_info("X", ["N", 3, 20, 20])], [W.value_info("Y", ["N", 6])])
    return W.write(str(tmp_path / f"norm_in_{variant}.onnx"), blob)


@pytest.mark.parametrize("variant", ["meanstd", "rescale", "bn"])
def test_input_normalisation_does_not_cost_the_conv_layout(built, tmp_path, variant):
    from infera_amd import capi

    capi.load_model("ni", _normalised_input_net(tmp_path, variant))
    plan = capi.get_plan("ni")
    capi.unload_model("ni")
    assert plan["activation_layout"] == "NC/4HW4", (variant, plan["activation_layout"])
    convs = [e for e, s in zip(plan["exec"], plan["plan"]["steps"]) if s["kind"] == "Conv2d"]
    assert convs[0] == "conv_patch" and convs[1] in ("conv_tiled_cq", "conv_split_bf16x6") and len(convs) == 2, convs  # (bf16x6: the default form of eligible layers)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["meanstd", "rescale", "bn"])
def test_gpu_input_normalisation_net(api, O, tmp_path, variant):
    path = _normalised_input_net(tmp_path, variant)
    x = synth.table(19, 0, 37, 3 * 20 * 20) * (100.0 if variant == "rescale" else 1.0)
    api.load_model("ni", path)
    try:
        got = api.predict_from_blob("ni", x.tobytes())
    finally:
        api.unload_model("ni")
    assert_close(got, O.Model(path).predict_blob(x.tobytes()))


# ---- CPU: the product's lowering (no GPU needed to load and lower) ---------------------------------------------
def test_lowering_of_breadth_models(built, paths):
    from infera_amd import capi

    for name, p in paths.items():
        capi.load_model("b_" + name, p)
    plan = capi.get_plan("b_exporter")
    kinds = [s["kind"] for s in plan["plan"]["steps"]]
    assert kinds == ["Conv2d", "Dense", "ArgMax"], kinds  # Shape/Gather/Unsqueeze/Concat/Reshape folded away
    assert capi.get_model_info("b_exporter")["output_shape"] == [-1, 1]
    zoo = [s["kind"] for s in capi.get_plan("b_zoo")["plan"]["steps"]]
    assert zoo.count("CopyCols") == 25 and zoo[0] == "Dense"
    cat = capi.get_plan("b_concat")["plan"]["steps"]
    assert [s["kind"] for s in cat].count("CopyCols") == 3 and cat[-1]["kind"] in ("Softmax", "Dense")
    mn = capi.get_plan("b_mnv2")["plan"]["steps"]
    assert all(s["kind"] in ("Conv2d", "BinaryAct", "GlobalAvgPool", "Dense") for s in mn), {s["kind"] for s in mn}
    assert sum("BatchNormalization" in s["origin"] for s in mn) == sum(s["kind"] == "Conv2d" for s in mn)  # all folded
    se = capi.get_plan("b_se")["plan"]["steps"]
    assert [s["kind"] for s in se].count("BinaryAct") == 3  # two gates (either operand order) + the Add
    for name in paths:
        capi.unload_model("b_" + name)


def test_unsupported_forms_fail_loudly(built, tmp_path):
    from infera_amd import capi

    w = np.ones((4, 4), np.float32)
    bad = [
        ("concat_axis0", [W.node("Concat", ["X", "X"], ["Y"], [W.attr_i("axis", 0)])], [], r"Concat\): only axis 1"),
        ("argmax_axis0", [W.node("ArgMax", ["X"], ["Y"], [W.attr_i("axis", 0)])], [], r"ArgMax\): only axis 1"),
        ("gelu_tanh", [W.node("Gelu", ["X"], ["Y"], [W.attr_s("approximate", "tanh")])], [], r"Gelu\): only the exact"),
        ("unknown", [W.node("Einsum", ["X", "w"], ["Y"])], [W.tensor("w", w)], r"Einsum\): unsupported operator"),
    ]
    for name, nodes, inits, pat in bad:
        p = W.write(str(tmp_path / f"{name}.onnx"), W.model(name, nodes, inits, [W.value_info("X", ["N", 4])], [W.value_info("Y", ["N", 4])]))
        with pytest.raises(capi.InferaError, match=pat):
            capi.load_model(name, p)


# ---- GPU: the HIP path against the oracle ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def api(built):
    from infera_amd import capi

    assert capi.device_count() >= 1, capi.get_devices()
    return capi


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 77, 4099])
def test_gpu_elementwise_zoo(api, O, paths, rows):
    api.load_model("zoo", paths["zoo"])
    x = synth.table(11, 0, rows, 16)
    got, want = api.predict("zoo", x), O.Model(paths["zoo"]).predict(x)
    # Floor/Ceil/Round of values within an ulp of an integer may legitimately differ: mask those columns' steps
    step_cols = np.zeros(got.shape[1], bool)
    for k in (6, 7, 9):
        step_cols[16 * k:16 * (k + 1)] = True
    assert_close(got[:, ~step_cols], want[:, ~step_cols])
    assert (np.abs(got[:, step_cols] - want[:, step_cols]) > 0).mean() < 1e-3
    api.unload_model("zoo")


@pytest.mark.gpu
def test_gpu_exporter_idiom_and_int64_labels(api, O, paths):
    api.load_model("exp", paths["exporter"])
    x = synth.table(8, 0, 301, 3 * 6 * 6)
    got, want = api.predict_from_blob("exp", x.tobytes()), O.Model(paths["exporter"]).predict_blob(x.tobytes())
    assert got.shape == want.shape == (301, 1)
    assert (got != want).mean() <= 0.01  # a label may flip only where two logits tie within fp32 summation order
    api.unload_model("exp")


@pytest.mark.gpu
def test_gpu_concat_heads(api, O, paths):
    api.load_model("cat", paths["concat"])
    x = synth.table(9, 0, 2500, 24)
    assert_close(api.predict("cat", x), O.Model(paths["concat"]).predict(x))
    api.unload_model("cat")


@pytest.mark.gpu
def test_gpu_se_gate_broadcast(api, O, paths):
    api.load_model("se", paths["se"])
    assert api.get_plan("se")["activation_layout"] == "NC/4HW4"
    x = synth.table(6, 0, 9, 4 * 10 * 10)
    assert_close(api.predict_from_blob("se", x.tobytes()), O.Model(paths["se"]).predict_blob(x.tobytes()))
    api.unload_model("se")


@pytest.mark.gpu
def test_gpu_mobilenet_v2_vs_oracle(api, O, paths):
    api.load_model("mnv2", paths["mnv2"])
    plan = api.get_plan("mnv2")
    assert plan["activation_layout"] == "NC/4HW4" and plan["exec"].count("conv_depthwise") == 9, plan["exec"]
    x = synth.table(4, 0, 5, 3 * 32 * 32)
    got, want = api.predict_from_blob("mnv2", x.tobytes()), O.Model(paths["mnv2"]).predict_blob(x.tobytes())
    assert got.shape == (5, 20)
    assert_close(got, want)
    api.unload_model("mnv2")


@pytest.mark.gpu
@pytest.mark.parametrize("length", [3, 5, 16, 17, 64, 65, 100, 128, 129, 256, 257, 1000, 1024, 1025, 3000])
@pytest.mark.parametrize("op", ["Softmax", "LogSoftmax", "NormL1", "NormL2", "NormMAX"])
def test_gpu_row_reductions_at_every_kernel_boundary(api, O, tmp_path, op, length):
    """Softmax / LogSoftmax / Normalizer over contiguous rows: one lane per row (short), 16 / 32 / 64 lanes per row with the
    row held in registers, one wave per row in three passes (long) -- lengths on both sides of every switch"""
    if op.startswith("Norm"):
        nd = W.node("Normalizer", ["X"], ["Y"], [W.attr_s("norm", op[4:])], domain=W.ML_DOMAIN)
        blob = W.model("rr", [nd], [], [W.value_info("X", ["N", length])], [W.value_info("Y", ["N", length])], ml_opset=1)
    else:
        blob = W.model("rr", [W.node(op, ["X"], ["Y"], [W.attr_i("axis", 1)])], [], [W.value_info("X", ["N", length])], [W.value_info("Y", ["N", length])])
    path = W.write(str(tmp_path / "rr.onnx"), blob)
    x = synth.table(length, 0, 333, length) * 4.0
    api.load_model("rr", path)
    try:
        got = api.predict("rr", x)
    finally:
        api.unload_model("rr")
    assert_close(got, O.Model(path).predict(x), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("length", [2, 7, 8, 64, 65, 100, 256, 257, 1000, 5000])
def test_gpu_argmax_rows_at_every_kernel_boundary(api, O, tmp_path, length):
    """stand-alone ArgMax: ties go to the first index, NaN never wins unless it is element 0 (the sequential scan's rule)"""
    blob = W.model("am", [W.node("ArgMax", ["X"], ["Y"], [W.attr_i("axis", 1), W.attr_i("keepdims", 0)])], [],
                   [W.value_info("X", ["N", length])], [W.value_info("Y", ["N"], W.INT64)])
    path = W.write(str(tmp_path / "am.onnx"), blob)
    rng = np.random.default_rng(length)
    x = np.round(synth.table(length, 0, 500, length) * 3.0)  # few distinct values: plenty of ties
    x[7, :] = -np.inf
    x[8, min(3, length - 1)] = np.nan
    x[9, 0] = np.nan
    x[10, length - 1] = np.inf
    api.load_model("am", path)
    try:
        got = api.predict("am", x.astype(np.float32)).reshape(-1)
    finally:
        api.unload_model("am")
    want = O.Model(path).predict(x.astype(np.float32)).reshape(-1)
    assert np.array_equal(got, want)
    ok = np.ones(500, bool)
    ok[[8, 9]] = False  # numpy's argmax treats NaN as the maximum
    assert np.array_equal(got[ok], np.argmax(x[ok], axis=1).astype(np.float32))
    assert got[9] == 0.0 and got[7] == 0.0


def _conv_head_net(tmp_path, classes=10):
    """global pool -> 1x1 Conv to `classes` (not a multiple of 4) -> Flatten -> Softmax: a fully-convolutional head"""
    ws = W._WeightStream(91)
    w1, b1 = ws.take((32, 3, 3, 3), 27), ws.take((32,), 27)
    w2, b2 = ws.take((classes, 32, 1, 1), 32), ws.take((classes,), 32)
    nodes = [W.node("Conv", ["X", "w1", "b1"], ["c1"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("pads", [1, 1, 1, 1])]),
             W.node("Relu", ["c1"], ["r1"]), W.node("GlobalAveragePool", ["r1"], ["g"]),
             W.node("Conv", ["g", "w2", "b2"], ["logits"], [W.attr_ints("kernel_shape", [1, 1])]),
             W.node("Flatten", ["logits"], ["f"]), W.node("Softmax", ["f"], ["Y"], [W.attr_i("axis", 1)])]
    blob = W.model("conv_head", nodes, [W.tensor("w1", w1), W.tensor("b1", b1), W.tensor("w2", w2), W.tensor("b2", b2)],
                   [W.value_info("X", ["N", 3, 12, 12])], [W.value_info("Y", ["N", classes])])
    return W.write(str(tmp_path / "conv_head.onnx"), blob)


def test_conv_head_with_odd_class_count_keeps_the_conv_layout(built, tmp_path):
    from infera_amd import capi

    capi.load_model("ch", _conv_head_net(tmp_path))
    plan = capi.get_plan("ch")
    capi.unload_model("ch")
    assert plan["activation_layout"] == "NC/4HW4", plan["activation_layout"]


@pytest.mark.gpu
@pytest.mark.parametrize("classes", [10, 7, 12])
def test_gpu_conv_head_with_odd_class_count(api, O, tmp_path, classes):
    path = _conv_head_net(tmp_path, classes)
    x = synth.table(29, 0, 41, 3 * 12 * 12)
    api.load_model("ch", path)
    try:
        got = api.predict_from_blob("ch", x.tobytes())
    finally:
        api.unload_model("ch")
    assert_close(got, O.Model(path).predict_blob(x.tobytes()))


def _tf_style_bias_net(tmp_path):
    """what TF / Keras converters leave behind: Conv without bias -> Add(bias [1,C,1,1]) -> Mul(scale [C,1,1]) -> Relu, and a
    per-channel Mul / Add pair behind a pooling layer (no convolution to fold into)"""
    ws = W._WeightStream(97)
    w1, w2 = ws.take((32, 3, 3, 3), 27), ws.take((64, 32, 3, 3), 288)
    fw, fb = ws.take((64, 5), 64), ws.take((5,), 64)
    rng = np.random.default_rng(4)
    b1, s1 = rng.standard_normal((1, 32, 1, 1)).astype(np.float32) * 0.1, (0.5 + rng.random((32, 1, 1))).astype(np.float32)
    b2 = rng.standard_normal((1, 64, 1, 1)).astype(np.float32) * 0.1
    ps, pb = (0.5 + rng.random((1, 64, 1, 1))).astype(np.float32), rng.standard_normal((64, 1, 1)).astype(np.float32) * 0.1
    cv = lambda x, w, o: W.node("Conv", [x, w], [o], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("pads", [1, 1, 1, 1])])
    nodes = [cv("X", "w1", "c1"), W.node("Add", ["c1", "b1"], ["a1"]), W.node("Mul", ["a1", "s1"], ["m1"]), W.node("Relu", ["m1"], ["r1"]),
             cv("r1", "w2", "c2"), W.node("Add", ["b2", "c2"], ["a2"]), W.node("Relu", ["a2"], ["r2"]),
             W.node("MaxPool", ["r2"], ["p"], [W.attr_ints("kernel_shape", [2, 2]), W.attr_ints("strides", [2, 2])]),
             W.node("Mul", ["p", "ps"], ["q"]), W.node("Add", ["q", "pb"], ["t"]),
             W.node("GlobalAveragePool", ["t"], ["g"]), W.node("Flatten", ["g"], ["f"]), W.node("Gemm", ["f", "fw", "fb"], ["Y"])]
    inits = [W.tensor(k, v) for k, v in dict(w1=w1, w2=w2, fw=fw, fb=fb, b1=b1, s1=s1, b2=b2, ps=ps, pb=pb).items()]
    blob = W.model("tf_bias", nodes, inits, [W.value_info("X", ["N", 3, 12, 12])], [W.value_info("Y", ["N", 5])])
    return W.write(str(tmp_path / "tf_bias.onnx"), blob)


def test_per_channel_constants_fold_into_convolutions(built, tmp_path):
    from infera_amd import capi

    capi.load_model("tfb", _tf_style_bias_net(tmp_path))
    plan = capi.get_plan("tfb")
    capi.unload_model("tfb")
    kinds = [s["kind"] for s in plan["plan"]["steps"]]
    assert kinds == ["Conv2d", "Conv2d", "Pool2d", "AffineChannel", "GlobalAvgPool", "Dense"], kinds
    assert plan["plan"]["steps"][0]["origin"].startswith("Conv+Add+Mul+Relu"), plan["plan"]["steps"][0]["origin"]
    assert plan["activation_layout"] == "NC/4HW4"


@pytest.mark.gpu
def test_gpu_per_channel_constants_net(api, O, tmp_path):
    path = _tf_style_bias_net(tmp_path)
    x = synth.table(33, 0, 29, 3 * 12 * 12)
    api.load_model("tfb", path)
    try:
        got = api.predict_from_blob("tfb", x.tobytes())
    finally:
        api.unload_model("tfb")
    assert_close(got, O.Model(path).predict_blob(x.tobytes()))


def _prelu_net(tmp_path):
    """per-channel PRelu slopes and a Max / Min clamp by constants between convolutions (face / older detection nets)"""
    ws = W._WeightStream(101)
    w1, b1 = ws.take((32, 3, 3, 3), 27), ws.take((32,), 27)
    w2, b2 = ws.take((32, 32, 3, 3), 288), ws.take((32,), 288)
    fw, fb = ws.take((32, 4), 32), ws.take((4,), 32)
    slope = (0.05 + 0.3 * np.random.default_rng(6).random((32, 1, 1))).astype(np.float32)
    cv = lambda x, w, b, o: W.node("Conv", [x, w, b], [o], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("pads", [1, 1, 1, 1])])
    nodes = [cv("X", "w1", "b1", "c1"), W.node("PRelu", ["c1", "slope"], ["p1"]), cv("p1", "w2", "b2", "c2"),
             W.node("Max", ["c2", "lo"], ["m1"]), W.node("Min", ["m1", "hi"], ["m2"]),
             W.node("GlobalAveragePool", ["m2"], ["g"]), W.node("Flatten", ["g"], ["f"]), W.node("Gemm", ["f", "fw", "fb"], ["Y"])]
    inits = [W.tensor(k, v) for k, v in dict(w1=w1, b1=b1, w2=w2, b2=b2, fw=fw, fb=fb, slope=slope,
                                             lo=np.array(-0.2, np.float32), hi=np.array(0.3, np.float32)).items()]
    blob = W.model("prelu_net", nodes, inits, [W.value_info("X", ["N", 3, 10, 10])], [W.value_info("Y", ["N", 4])])
    return W.write(str(tmp_path / "prelu_net.onnx"), blob)


def test_prelu_and_clamps_keep_the_conv_layout(built, tmp_path):
    from infera_amd import capi

    capi.load_model("pr", _prelu_net(tmp_path))
    plan = capi.get_plan("pr")
    capi.unload_model("pr")
    assert plan["activation_layout"] == "NC/4HW4", plan["activation_layout"]


@pytest.mark.gpu
def test_gpu_prelu_net(api, O, tmp_path):
    path = _prelu_net(tmp_path)
    x = synth.table(35, 0, 31, 3 * 10 * 10)
    api.load_model("pr", path)
    try:
        got = api.predict_from_blob("pr", x.tobytes())
    finally:
        api.unload_model("pr")
    assert_close(got, O.Model(path).predict_blob(x.tobytes()))


def _narrow_stem_net(tmp_path, stem):
    """stems with 16 / 24 output channels (MobileNetV3, ShuffleNet): 7x7 or 3x3 stride-2 convolution over the NCHW image"""
    ws = W._WeightStream(111 + stem)
    k = 7 if stem == 24 else 3
    w1, b1 = ws.take((stem, 3, k, k), 3 * k * k), ws.take((stem,), 3 * k * k)
    w2, b2 = ws.take((32, stem, 3, 3), stem * 9), ws.take((32,), stem * 9)
    fw, fb = ws.take((32, 3), 32), ws.take((3,), 32)
    nodes = [W.node("Conv", ["X", "w1", "b1"], ["c1"], [W.attr_ints("kernel_shape", [k, k]), W.attr_ints("pads", [k // 2] * 4), W.attr_ints("strides", [2, 2])]),
             W.node("HardSwish", ["c1"], ["a1"]),
             W.node("Conv", ["a1", "w2", "b2"], ["c2"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("pads", [1, 1, 1, 1])]), W.node("Relu", ["c2"], ["a2"]),
             W.node("GlobalAveragePool", ["a2"], ["g"]), W.node("Flatten", ["g"], ["f"]), W.node("Gemm", ["f", "fw", "fb"], ["Y"])]
    inits = [W.tensor(n, v) for n, v in dict(w1=w1, b1=b1, w2=w2, b2=b2, fw=fw, fb=fb).items()]
    blob = W.model("narrow_stem", nodes, inits, [W.value_info("X", ["N", 3, 18, 18])], [W.value_info("Y", ["N", 3])], opset=14)
    return W.write(str(tmp_path / f"narrow_stem{stem}.onnx"), blob)


@pytest.mark.parametrize("stem", [16, 24, 32])
def test_narrow_stems_run_on_the_patch_kernel(built, tmp_path, stem):
    from infera_amd import capi

    capi.load_model("ns", _narrow_stem_net(tmp_path, stem))
    plan = capi.get_plan("ns")
    capi.unload_model("ns")
    assert plan["activation_layout"] == "NC/4HW4" and plan["exec"][0] == "conv_patch" and plan["exec"][1] in ("conv_tiled_cq", "conv_split_bf16x6"), plan["exec"]
    assert plan["plan"]["steps"][0].get("act") == "HardSwish", plan["plan"]["steps"][0]


@pytest.mark.gpu
@pytest.mark.parametrize("stem", [16, 24, 32])
def test_gpu_narrow_stems(api, O, tmp_path, stem):
    path = _narrow_stem_net(tmp_path, stem)
    x = synth.table(37, 0, 33, 3 * 18 * 18)
    api.load_model("ns", path)
    try:
        got = api.predict_from_blob("ns", x.tobytes())
    finally:
        api.unload_model("ns")
    assert_close(got, O.Model(path).predict_blob(x.tobytes()))


def _conv1d_net(tmp_path, in_ch=4):
    """1-D CNN over [N, C, L] sequences (sensor / ECG windows): Conv1d k5 -> Relu -> MaxPool1d -> Conv1d k3 s2 d1 ->
    BatchNormalization -> Relu -> AveragePool1d -> GlobalAveragePool -> Gemm"""
    ws = W._WeightStream(131)
    w1, b1 = ws.take((32, in_ch, 5), in_ch * 5), ws.take((32,), in_ch * 5)
    w2, b2 = ws.take((64, 32, 3), 96), ws.take((64,), 96)
    fw, fb = ws.take((64, 3), 64), ws.take((3,), 64)
    rng = np.random.default_rng(8)
    bn = dict(g=(0.5 + rng.random(64)).astype(np.float32), b=(rng.standard_normal(64) * 0.1).astype(np.float32),
              mu=(rng.standard_normal(64) * 0.1).astype(np.float32), var=(0.5 + rng.random(64)).astype(np.float32))
    nodes = [W.node("Conv", ["X", "w1", "b1"], ["c1"], [W.attr_ints("kernel_shape", [5]), W.attr_ints("pads", [2, 2])]), W.node("Relu", ["c1"], ["r1"]),
             W.node("MaxPool", ["r1"], ["p1"], [W.attr_ints("kernel_shape", [2]), W.attr_ints("strides", [2])]),
             W.node("Conv", ["p1", "w2", "b2"], ["c2"], [W.attr_ints("kernel_shape", [3]), W.attr_ints("pads", [1, 1]), W.attr_ints("strides", [2])]),
             W.node("BatchNormalization", ["c2", "g", "b", "mu", "var"], ["n2"]), W.node("Relu", ["n2"], ["r2"]),
             W.node("AveragePool", ["r2"], ["p2"], [W.attr_ints("kernel_shape", [3]), W.attr_ints("strides", [1]), W.attr_ints("pads", [1, 1])]),
             W.node("GlobalAveragePool", ["p2"], ["gp"]), W.node("Flatten", ["gp"], ["f"]), W.node("Gemm", ["f", "fw", "fb"], ["Y"])]
    inits = [W.tensor(k, v) for k, v in dict(w1=w1, b1=b1, w2=w2, b2=b2, fw=fw, fb=fb, **bn).items()]
    blob = W.model("conv1d", nodes, inits, [W.value_info("X", ["N", in_ch, 64])], [W.value_info("Y", ["N", 3])])

    def ref(x):
        xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (2, 2)))
        win = np.lib.stride_tricks.sliding_window_view(xp, 5, axis=2)  # n c l k
        a = np.maximum(np.einsum("nclk,mck->nml", win, w1.astype(np.float64)) + b1[None, :, None], 0)
        a = a.reshape(a.shape[0], 32, 32, 2).max(axis=3)
        ap = np.pad(a, ((0, 0), (0, 0), (1, 1)))
        win = np.lib.stride_tricks.sliding_window_view(ap, 3, axis=2)[:, :, ::2]
        c = np.einsum("nclk,mck->nml", win, w2.astype(np.float64)) + b2[None, :, None]
        c = (c - bn["mu"][None, :, None]) / np.sqrt(bn["var"].astype(np.float64) + 1e-5)[None, :, None] * bn["g"][None, :, None] + bn["b"][None, :, None]
        c = np.maximum(c, 0)
        L = c.shape[2]
        cp = np.pad(c, ((0, 0), (0, 0), (1, 1)))
        cnt = np.array([min(i + 2, L) - max(i - 1, 0) for i in range(L)], np.float64)
        p = np.lib.stride_tricks.sliding_window_view(cp, 3, axis=2).sum(axis=3) / cnt
        return p.mean(axis=2) @ fw.astype(np.float64) + fb

    return W.write(str(tmp_path / f"conv1d_{in_ch}.onnx"), blob), ref


def test_oracle_conv1d_net_vs_numpy(O, built, tmp_path):
    from infera_amd import capi

    path, ref = _conv1d_net(tmp_path)
    x = synth.table(45, 0, 6, 4 * 64)
    assert_close(O.Model(path).predict_blob(x.tobytes()), ref(x.reshape(6, 4, 64)).astype(np.float32), rtol=3e-5, atol=2e-6)
    capi.load_model("c1d", path)
    plan, info = capi.get_plan("c1d"), capi.get_model_info("c1d")
    capi.unload_model("c1d")
    assert info["input_shape"] == [-1, 4, 64] and plan["activation_layout"] == "NC/4HW4", (info, plan["activation_layout"])
    assert plan["exec"][0] == "conv_patch" and ("conv_tiled_cq" in plan["exec"] or "conv_split_bf16x6" in plan["exec"]), plan["exec"]


@pytest.mark.gpu
@pytest.mark.parametrize("in_ch", [4, 12])
def test_gpu_conv1d_net(api, O, tmp_path, in_ch):
    path, _ = _conv1d_net(tmp_path, in_ch)
    x = synth.table(45, 0, 300, in_ch * 64)
    api.load_model("c1d", path)
    try:
        got = api.predict_from_blob("c1d", x.tobytes())
    finally:
        api.unload_model("c1d")
    assert_close(got, O.Model(path).predict_blob(x.tobytes()))


@pytest.mark.parametrize("index,shape", [(np.array(1, np.int64), ["N"]), (np.array([2, 3], np.int64), ["N", 2]), (np.array([-1], np.int64), ["N", 1])])
def test_gather_of_class_columns(O, built, tmp_path, index, shape):
    """Gather(axis=1) with constant indices behind a classifier: the probability of one class / a column range"""
    from infera_amd import capi

    ws = W._WeightStream(141)
    w, b = ws.take((9, 4), 9), ws.take((4,), 9)
    nodes = [W.node("Gemm", ["X", "w", "b"], ["s"]), W.node("Softmax", ["s"], ["p"], [W.attr_i("axis", 1)]),
             W.node("Gather", ["p", "idx"], ["Y"], [W.attr_i("axis", 1)])]
    blob = W.model("gat", nodes, [W.tensor("w", w), W.tensor("b", b), W.tensor("idx", index)], [W.value_info("X", ["N", 9])], [W.value_info("Y", shape)])
    path = W.write(str(tmp_path / "gat.onnx"), blob)
    x = synth.table(3, 0, 50, 9)
    z = x.astype(np.float64) @ w + b
    p = np.exp(z - z.max(axis=1, keepdims=True))
    p /= p.sum(axis=1, keepdims=True)
    want = p[:, index] if index.ndim else p[:, int(index)]
    got = O.Model(path).predict(x)
    assert_close(got.reshape(want.shape), want.astype(np.float32), rtol=3e-5, atol=2e-6)
    capi.load_model("gat", path)
    kinds = [s["kind"] for s in capi.get_plan("gat")["plan"]["steps"]]
    capi.unload_model("gat")
    assert kinds[-1] == "SliceCols", kinds


@pytest.mark.gpu
def test_gpu_gather_of_class_columns(api, O, tmp_path):
    ws = W._WeightStream(141)
    w, b = ws.take((9, 4), 9), ws.take((4,), 9)
    nodes = [W.node("Gemm", ["X", "w", "b"], ["s"]), W.node("Softmax", ["s"], ["p"], [W.attr_i("axis", 1)]),
             W.node("Gather", ["p", "idx"], ["Y"], [W.attr_i("axis", 1)])]
    blob = W.model("gat", nodes, [W.tensor("w", w), W.tensor("b", b), W.tensor("idx", np.array(1, np.int64))], [W.value_info("X", ["N", 9])], [W.value_info("Y", ["N"])])
    path = W.write(str(tmp_path / "gat.onnx"), blob)
    x = synth.table(3, 0, 3001, 9)
    api.load_model("gat", path)
    try:
        got = api.predict("gat", x)
    finally:
        api.unload_model("gat")
    assert_close(got, O.Model(path).predict(x))
