"""Minimal ONNX protobuf writer (no `onnx` package in this image).

Emits ModelProto files for the benchmark / parity configurations of BASELINE.md section 4:
a dynamic-batch twin of the reference's ``linear.onnx`` fixture, the C2/C3 MLP
(Gemm+Relu chain), the C4 logistic-regression + Softmax model and the C5 ResNet-18 topology.
All models carry a symbolic batch dimension ``N`` so a whole DuckDB vector (<=2048 rows) or
a super-batch goes through in one call (the reference's fixture has a fixed batch of 1,
/root/reference test/models/README.md:5).

Wire format follows onnx.proto3 field numbers (ModelProto.graph=7, GraphProto.node=1 ...).
Weights are written as ``raw_data`` (little-endian f32) unless ``float_data=True`` which
reproduces the encoding of the reference fixture (packed field 4).
"""
from __future__ import annotations

import math
import struct
from typing import Iterable, Sequence

import numpy as np

from .synth import uniform_pm1

FLOAT = 1
INT64 = 7


def _varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _tag(field: int, wt: int) -> bytes:
    return _varint((field << 3) | wt)


def _ld(field: int, payload: bytes) -> bytes:
    return _tag(field, 2) + _varint(len(payload)) + payload


def _vi(field: int, v: int) -> bytes:
    return _tag(field, 0) + _varint(v)


def _s(field: int, s: str) -> bytes:
    return _ld(field, s.encode())


def tensor(name: str, arr: np.ndarray, float_data: bool = False) -> bytes:
    """TensorProto: dims=1, data_type=2, float_data=4, int64_data=7, name=8, raw_data=9."""
    out = b"".join(_vi(1, int(d)) for d in arr.shape)
    if arr.dtype == np.float32:
        out += _vi(2, FLOAT)
        if float_data:
            out += _ld(4, arr.astype("<f4").tobytes())
        else:
            out += _ld(9, arr.astype("<f4").tobytes())
    elif arr.dtype == np.int64:
        out += _vi(2, INT64)
        out += _ld(9, arr.astype("<i8").tobytes())
    else:
        raise TypeError(arr.dtype)
    return out + _s(8, name)


def attr_i(name: str, v: int) -> bytes:
    return _s(1, name) + _vi(3, v) + _vi(20, 2)


def attr_f(name: str, v: float) -> bytes:
    return _s(1, name) + _tag(2, 5) + struct.pack("<f", v) + _vi(20, 1)


def attr_ints(name: str, vs: Iterable[int]) -> bytes:
    return _s(1, name) + b"".join(_vi(8, int(v)) for v in vs) + _vi(20, 7)


def attr_s(name: str, v: str) -> bytes:
    return _s(1, name) + _ld(4, v.encode()) + _vi(20, 3)


def attr_floats(name: str, vs: Iterable[float]) -> bytes:
    return _s(1, name) + b"".join(_tag(7, 5) + struct.pack("<f", float(v)) for v in vs) + _vi(20, 6)


def attr_strings(name: str, vs: Iterable[str]) -> bytes:
    return _s(1, name) + b"".join(_ld(9, v.encode()) for v in vs) + _vi(20, 8)


ML_DOMAIN = "ai.onnx.ml"


def node(op: str, inputs: Sequence[str], outputs: Sequence[str], attrs: Sequence[bytes] = (), name: str = "",
         domain: str = "") -> bytes:
    out = b"".join(_s(1, i) for i in inputs) + b"".join(_s(2, o) for o in outputs)
    if name:
        out += _s(3, name)
    out += _s(4, op)
    out += b"".join(_ld(5, a) for a in attrs)
    if domain:
        out += _s(7, domain)
    return out


def value_info(name: str, dims: Sequence[int | str], elem_type: int = FLOAT) -> bytes:
    shape = b""
    for d in dims:
        dim = _s(2, d) if isinstance(d, str) else _vi(1, int(d))
        shape += _ld(1, dim)
    ttype = _vi(1, elem_type) + _ld(2, shape)
    return _s(1, name) + _ld(2, _ld(1, ttype))


def model(graph_name: str, nodes: Sequence[bytes], inits: Sequence[bytes], inputs: Sequence[bytes],
          outputs: Sequence[bytes], opset: int = 13, ir_version: int = 8, producer: str = "infera_amd",
          ml_opset: int | None = None) -> bytes:
    g = b"".join(_ld(1, n) for n in nodes) + _s(2, graph_name)
    g += b"".join(_ld(5, t) for t in inits)
    g += b"".join(_ld(11, i) for i in inputs) + b"".join(_ld(12, o) for o in outputs)
    opset_import = _ld(8, _s(1, "") + _vi(2, opset))
    if ml_opset is not None:  # the classical-ML operator domain sklearn exporters use
        opset_import += _ld(8, _s(1, ML_DOMAIN) + _vi(2, ml_opset))
    return _vi(1, ir_version) + _s(2, producer) + _ld(7, g) + opset_import


# ------------------------------------------------------------------------------------------
# deterministic weights: U(-1/sqrt(fan_in), 1/sqrt(fan_in)), seed 1234
# ------------------------------------------------------------------------------------------

class _WeightStream:
    def __init__(self, seed: int = 1234):
        self.seed = seed
        self.counter = 0

    def take(self, shape: Sequence[int], fan_in: int) -> np.ndarray:
        n = int(np.prod(shape))
        idx = np.arange(self.counter, self.counter + n, dtype=np.uint64)
        self.counter += n
        u = uniform_pm1(self.seed, idx).astype(np.float64)
        return (u / math.sqrt(fan_in)).astype(np.float32).reshape(shape)


def linear_dyn() -> bytes:
    """Dynamic-batch twin of the reference fixture test/models/linear.onnx: Y = X.W + B with
    W=(2,-1,0.5), B=0.25 -> (1,2,3) -> 1.75 (test/sql/test_core_functionality.test:48-56)."""
    w = np.array([[2.0], [-1.0], [0.5]], np.float32)
    b = np.array([0.25], np.float32)
    nodes = [node("MatMul", ["X", "W"], ["Z"]), node("Add", ["Z", "B"], ["Y"])]
    return model("LinearModelDyn", nodes, [tensor("W", w, float_data=True), tensor("B", b, float_data=True)],
                 [value_info("X", ["N", 3])], [value_info("Y", ["N", 1])])


def mlp(dims: Sequence[int] = (128, 256, 64, 1), acts: Sequence[str] | None = None, final_softmax: bool = False,
        seed: int = 1234, use_matmul_add: bool = False, trans_b: bool = False, opset: int = 13,
        batch: int | str = "N") -> bytes:
    """Gemm chain `dims[0] -> dims[1] -> ...` with an activation after every layer but the last
    (default Relu; names from {"Relu","Sigmoid","Tanh","LeakyRelu",""}).  C2/C3 = defaults."""
    nl = len(dims) - 1
    if acts is None:
        acts = ["Relu"] * (nl - 1) + [""]
    assert len(acts) == nl
    ws = _WeightStream(seed)
    nodes, inits = [], []
    cur = "X"
    for l in range(nl):
        k, m = dims[l], dims[l + 1]
        w = ws.take((k, m), k)
        b = ws.take((m,), k)
        out = f"H{l}" if (l < nl - 1 or acts[l] or final_softmax) else "Y"
        if use_matmul_add:
            inits += [tensor(f"W{l}", w), tensor(f"B{l}", b)]
            nodes += [node("MatMul", [cur, f"W{l}"], [f"Z{l}"]), node("Add", [f"Z{l}", f"B{l}"], [out])]
        elif trans_b:
            inits += [tensor(f"W{l}", np.ascontiguousarray(w.T)), tensor(f"B{l}", b)]
            nodes += [node("Gemm", [cur, f"W{l}", f"B{l}"], [out], [attr_i("transB", 1)])]
        else:
            inits += [tensor(f"W{l}", w), tensor(f"B{l}", b)]
            nodes += [node("Gemm", [cur, f"W{l}", f"B{l}"], [out])]
        cur = out
        if acts[l]:
            last = l == nl - 1 and not final_softmax
            out = "Y" if last else f"A{l}"
            attrs = [attr_f("alpha", 0.1)] if acts[l] == "LeakyRelu" else []
            nodes.append(node(acts[l], [cur], [out], attrs))
            cur = out
    if final_softmax:
        nodes.append(node("Softmax", [cur], ["Y"], [attr_i("axis", 1)]))
    return model("mlp_" + "x".join(map(str, dims)), nodes, inits, [value_info("X", [batch, dims[0]])],
                 [value_info("Y", [batch, dims[-1]])], opset=opset)


def logreg_softmax(features: int = 128, classes: int = 10, seed: int = 1234) -> bytes:
    """C4: Gemm(features -> classes) + Softmax(axis=1)."""
    return mlp((features, classes), acts=[""], final_softmax=True, seed=seed)


def identity(cols: int = 4, batch: int | str = "N") -> bytes:
    return model("identity_dyn", [node("Identity", ["X"], ["Y"])], [], [value_info("X", [batch, cols])],
                 [value_info("Y", [batch, cols])], opset=13)


def resnet18(classes: int = 1000, seed: int = 1234, in_hw: int = 224, width: int = 64) -> bytes:
    """C5: ResNet-18 topology (conv7x7/2 + BN + Relu + maxpool3x3/2, 4 stages x 2 BasicBlocks,
    global-avgpool, Flatten, Gemm -> classes), random weights, BatchNormalization kept as
    separate nodes so the loader's BN-fold is exercised.  Input [N,3,in_hw,in_hw]."""
    ws = _WeightStream(seed)
    nodes, inits = [], []
    uid = [0]

    def fresh(p: str) -> str:
        uid[0] += 1
        return f"{p}{uid[0]}"

    def conv_bn(x: str, cin: int, cout: int, k: int, stride: int, pad: int, relu: bool) -> str:
        w = ws.take((cout, cin, k, k), cin * k * k)
        wn, y = fresh("w"), fresh("c")
        inits.append(tensor(wn, w))
        nodes.append(node("Conv", [x, wn], [y], [attr_ints("kernel_shape", [k, k]), attr_ints("strides", [stride, stride]),
                                                 attr_ints("pads", [pad] * 4)]))
        scale = (1.0 + 0.1 * ws.take((cout,), 1)).astype(np.float32)
        beta = (0.1 * ws.take((cout,), 1)).astype(np.float32)
        mean = (0.1 * ws.take((cout,), 1)).astype(np.float32)
        var = (1.0 + 0.5 * np.abs(ws.take((cout,), 1))).astype(np.float32)
        names = [fresh("bn_s"), fresh("bn_b"), fresh("bn_m"), fresh("bn_v")]
        for nme, arr in zip(names, (scale, beta, mean, var)):
            inits.append(tensor(nme, arr))
        z = fresh("b")
        nodes.append(node("BatchNormalization", [y] + names, [z], [attr_f("epsilon", 1e-5)]))
        if relu:
            r = fresh("r")
            nodes.append(node("Relu", [z], [r]))
            return r
        return z

    x = conv_bn("X", 3, width, 7, 2, 3, True)
    p = fresh("p")
    nodes.append(node("MaxPool", [x], [p], [attr_ints("kernel_shape", [3, 3]), attr_ints("strides", [2, 2]), attr_ints("pads", [1, 1, 1, 1])]))
    x, cin = p, width
    for stage, cout in enumerate([width, width * 2, width * 4, width * 8]):
        for blk in range(2):
            stride = 2 if (stage > 0 and blk == 0) else 1
            y = conv_bn(x, cin, cout, 3, stride, 1, True)
            y = conv_bn(y, cout, cout, 3, 1, 1, False)
            sc = x
            if stride != 1 or cin != cout:
                sc = conv_bn(x, cin, cout, 1, stride, 0, False)
            s, r = fresh("s"), fresh("r")
            nodes.append(node("Add", [y, sc], [s]))
            nodes.append(node("Relu", [s], [r]))
            x, cin = r, cout
    g, f = fresh("g"), fresh("f")
    nodes.append(node("GlobalAveragePool", [x], [g]))
    nodes.append(node("Flatten", [g], [f], [attr_i("axis", 1)]))
    w = ws.take((cin, classes), cin)
    b = ws.take((classes,), cin)
    inits += [tensor("fc_w", w), tensor("fc_b", b)]
    nodes.append(node("Gemm", [f, "fc_w", "fc_b"], ["Y"]))
    return model("resnet18", nodes, inits, [value_info("X", ["N", 3, in_hw, in_hw])], [value_info("Y", ["N", classes])], opset=13)


def unary_zoo(features: int = 16) -> bytes:
    """Every elementwise operator of the breadth set in one graph: a Gemm feeds parallel branches (one per
    operator, inputs pre-conditioned where the domain needs it), the branches are concatenated (axis 1)."""
    ws = _WeightStream(77)
    w = ws.take((features, features), features)
    b = ws.take((features,), features)
    inits = [tensor("W", w), tensor("B", b), tensor("one", np.array([1.5], np.float32)), tensor("two", np.array([2.0], np.float32)),
             tensor("slope", (0.05 + 0.2 * np.abs(ws.take((features,), 1))).astype(np.float32)),
             tensor("lo", (-0.25 * np.ones((1, features))).astype(np.float32))]
    nodes = [node("Gemm", ["X", "W", "B"], ["H"]),
             node("Abs", ["H"], ["Hp0"]), node("Add", ["Hp0", "one"], ["Hpos"])]  # strictly positive branch input
    outs = []

    def br(op, src="H", attrs=()):
        o = f"o_{len(outs)}"
        nodes.append(node(op, [src], [o], list(attrs)))
        outs.append(o)

    for op in ("Exp", "Neg", "Abs", "Softplus", "HardSwish", "Erf", "Floor", "Ceil", "Softsign", "Round", "Sigmoid", "Tanh", "Relu"):
        br(op)
    br("Elu", attrs=[attr_f("alpha", 0.7)])
    br("Selu")
    br("HardSigmoid", attrs=[attr_f("alpha", 0.3), attr_f("beta", 0.4)])
    br("LeakyRelu", attrs=[attr_f("alpha", 0.2)])
    for op in ("Log", "Sqrt", "Reciprocal"):
        br(op, "Hpos")
    for op, rhs in (("Pow", "two"), ("Min", "lo"), ("Max", "lo"), ("PRelu", "slope")):
        o = f"o_{len(outs)}"
        nodes.append(node(op, ["H", rhs], [o]))
        outs.append(o)
    o = f"o_{len(outs)}"
    nodes.append(node("Max", ["lo", "H"], [o]))  # constant on the left
    outs.append(o)
    nodes.append(node("Concat", outs, ["Y"], [attr_i("axis", 1)]))
    return model("unary_zoo", nodes, inits, [value_info("X", ["N", features])], [value_info("Y", ["N", features * len(outs)])], opset=13)


def exporter_reshape(c: int = 8, hw: int = 6, classes: int = 5) -> bytes:
    """The PyTorch-exporter idiom `x.view(x.size(0), -1)`: Shape -> Gather(0) -> Unsqueeze -> Concat([n, -1]) -> Reshape,
    after a small conv + ReduceMean-free head, then Gemm + ArgMax with an INT64 output (label as f32 value)."""
    ws = _WeightStream(91)
    wc = ws.take((c, 3, 3, 3), 27)
    bc = ws.take((c,), 27)
    feat = c * hw * hw
    wf = ws.take((feat, classes), feat)
    bf = ws.take((classes,), feat)
    inits = [tensor("wc", wc), tensor("bc", bc), tensor("wf", wf), tensor("bf", bf), tensor("i0", np.array(0, np.int64).reshape(())),
             tensor("ax0", np.array([0], np.int64)), tensor("m1", np.array([-1], np.int64))]
    nodes = [node("Conv", ["X", "wc", "bc"], ["c1"], [attr_ints("kernel_shape", [3, 3]), attr_ints("pads", [1, 1, 1, 1])]),
             node("Relu", ["c1"], ["r1"]),
             node("Shape", ["r1"], ["shp"]),
             node("Gather", ["shp", "i0"], ["n"], [attr_i("axis", 0)]),
             node("Unsqueeze", ["n", "ax0"], ["n1"]),
             node("Concat", ["n1", "m1"], ["tgt"], [attr_i("axis", 0)]),
             node("Reshape", ["r1", "tgt"], ["flat"]),
             node("Gemm", ["flat", "wf", "bf"], ["logits"]),
             node("ArgMax", ["logits"], ["Y"], [attr_i("axis", 1), attr_i("keepdims", 1)])]
    return model("exporter_reshape", nodes, inits, [value_info("X", ["N", 3, hw, hw])], [value_info("Y", ["N", 1], INT64)], opset=13)


def concat_heads(features: int = 24) -> bytes:
    """Two dense towers over the same input, concatenated on the feature axis, then a head + Softmax."""
    ws = _WeightStream(55)
    inits, nodes = [], []
    for t, m in enumerate((12, 20)):
        w, b = ws.take((features, m), features), ws.take((m,), features)
        inits += [tensor(f"W{t}", w), tensor(f"B{t}", b)]
        nodes += [node("Gemm", ["X", f"W{t}", f"B{t}"], [f"h{t}"]), node("Tanh" if t else "Relu", [f"h{t}"], [f"a{t}"])]
    nodes.append(node("Concat", ["a0", "X", "a1"], ["cat"], [attr_i("axis", 1)]))
    k = 12 + features + 20
    w, b = ws.take((k, 7), k), ws.take((7,), k)
    inits += [tensor("Wh", w), tensor("Bh", b)]
    nodes += [node("Gemm", ["cat", "Wh", "Bh"], ["z"]), node("Softmax", ["z"], ["Y"], [attr_i("axis", 1)])]
    return model("concat_heads", nodes, inits, [value_info("X", ["N", features])], [value_info("Y", ["N", 7])], opset=13)


def mobilenet_v2(classes: int = 100, in_hw: int = 64, width_mult: float = 0.5, seed: int = 4321) -> bytes:
    """MobileNetV2 topology (the model family of the reference's blob test, test_advanced_features.test:46-63):
    3x3/2 stem, inverted residual blocks (1x1 expand + ReLU6, 3x3 DEPTHWISE + ReLU6, 1x1 linear project, residual Add
    when shapes match), 1x1 head, ReduceMean over the spatial axes, Gemm.  ReLU6 = Clip(0, 6) with tensor bounds;
    BatchNormalization after every conv (folded by the loader)."""
    ws = _WeightStream(seed)
    nodes, inits = [], [tensor("zero", np.array(0.0, np.float32).reshape(())), tensor("six", np.array(6.0, np.float32).reshape(()))]
    uid = [0]

    def fresh(p):
        uid[0] += 1
        return f"{p}{uid[0]}"

    def ch(v):
        return max(8, int(v * width_mult + 4) // 8 * 8)

    def conv_bn(x, cin, cout, k, stride, groups, relu6):
        w = ws.take((cout, cin // groups, k, k), (cin // groups) * k * k)
        wn, y = fresh("w"), fresh("c")
        inits.append(tensor(wn, w))
        nodes.append(node("Conv", [x, wn], [y], [attr_ints("kernel_shape", [k, k]), attr_ints("strides", [stride, stride]),
                                                 attr_ints("pads", [k // 2] * 4), attr_i("group", groups)]))
        names = [fresh("bn_s"), fresh("bn_b"), fresh("bn_m"), fresh("bn_v")]
        arrs = ((1.0 + 0.1 * ws.take((cout,), 1)), 0.1 * ws.take((cout,), 1), 0.1 * ws.take((cout,), 1), 1.0 + 0.5 * np.abs(ws.take((cout,), 1)))
        for nme, arr in zip(names, arrs):
            inits.append(tensor(nme, arr.astype(np.float32)))
        z = fresh("b")
        nodes.append(node("BatchNormalization", [y] + names, [z], [attr_f("epsilon", 1e-5)]))
        if relu6:
            r = fresh("r")
            nodes.append(node("Clip", [z, "zero", "six"], [r]))
            return r
        return z

    cin = ch(32)
    x = conv_bn("X", 3, cin, 3, 2, 1, True)
    for t, c, n, s in ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 2, 2), (6, 96, 1, 1)):
        cout = ch(c)
        for i in range(n):
            stride = s if i == 0 else 1
            hid = cin * t
            y = x
            if t != 1:
                y = conv_bn(y, cin, hid, 1, 1, 1, True)
            y = conv_bn(y, hid, hid, 3, stride, hid, True)  # depthwise
            y = conv_bn(y, hid, cout, 1, 1, 1, False)
            if stride == 1 and cin == cout:
                a = fresh("s")
                nodes.append(node("Add", [x, y], [a]))
                y = a
            x, cin = y, cout
    head = ch(640)
    x = conv_bn(x, cin, head, 1, 1, 1, True)
    g = fresh("g")
    nodes.append(node("ReduceMean", [x], [g], [attr_ints("axes", [2, 3]), attr_i("keepdims", 0)]))
    w, b = ws.take((head, classes), head), ws.take((classes,), head)
    inits += [tensor("fc_w", w), tensor("fc_b", b)]
    nodes.append(node("Gemm", [g, "fc_w", "fc_b"], ["Y"]))
    return model("mobilenet_v2", nodes, inits, [value_info("X", ["N", 3, in_hw, in_hw])], [value_info("Y", ["N", classes])], opset=13)


def se_net(c: int = 16, hw: int = 10, classes: int = 6) -> bytes:
    """Conv -> squeeze-and-excitation block (GlobalAveragePool -> 1x1 Conv -> Relu -> 1x1 Conv -> HardSigmoid ->
    Mul gate [N,C,1,1] over [N,C,H,W]) -> residual Add with a gated copy -> GAP -> Gemm.  MobileNetV3 / EfficientNet idiom."""
    ws = _WeightStream(31)
    inits, nodes = [], []

    def conv(x, cin, cout, k, out, act=None):
        w, b = ws.take((cout, cin, k, k), cin * k * k), ws.take((cout,), cin * k * k)
        inits.extend([tensor(out + "_w", w), tensor(out + "_b", b)])
        nodes.append(node("Conv", [x, out + "_w", out + "_b"], [out if act is None else out + "_pre"],
                          [attr_ints("kernel_shape", [k, k]), attr_ints("pads", [k // 2] * 4)]))
        if act:
            nodes.append(node(act, [out + "_pre"], [out]))
        return out

    x = conv("X", 4, c, 3, "stem", "Relu")
    sq = "sq"
    nodes.append(node("GlobalAveragePool", [x], [sq]))
    r = conv(sq, c, c // 4, 1, "se_r", "Relu")
    e = conv(r, c // 4, c, 1, "se_e", "HardSigmoid")
    nodes.append(node("Mul", [x, e], ["gated"]))
    nodes.append(node("Mul", [e, x], ["gated2"]))  # gate on the left: commutative form
    nodes.append(node("Add", ["gated", "gated2"], ["sum"]))
    nodes.append(node("GlobalAveragePool", ["sum"], ["g"]))
    nodes.append(node("Flatten", ["g"], ["f"], [attr_i("axis", 1)]))
    w, b = ws.take((c, classes), c), ws.take((classes,), c)
    inits += [tensor("fc_w", w), tensor("fc_b", b)]
    nodes.append(node("Gemm", ["f", "fc_w", "fc_b"], ["Y"]))
    return model("se_net", nodes, inits, [value_info("X", ["N", 4, hw, hw])], [value_info("Y", ["N", classes])], opset=14)


def zoo_ops_net(hw: int = 16, classes: int = 7) -> tuple[bytes, dict]:
    """One small network made of the operators the ONNX Model Zoo's vision models add to plain conv nets:
    LRN (AlexNet / GoogLeNet), explicit asymmetric Pad in front of a VALID Conv (TF / Keras exporters), channel Split,
    Concat and the Reshape -> Transpose -> Reshape channel shuffle (ShuffleNet), channel Slice, n-ary Sum and
    GlobalMaxPool.  Returns (model bytes, weights by name) so a test can restate it in numpy."""
    ws = _WeightStream(47)
    inits, nodes, wts = [], [], {}

    def conv(x, cin, cout, k, out, pads, stride=1, groups=1, act=None):
        w, b = ws.take((cout, cin // groups, k, k), (cin // groups) * k * k), ws.take((cout,), cin * k * k)
        wts[out] = (w, b)
        inits.extend([tensor(out + "_w", w), tensor(out + "_b", b)])
        nodes.append(node("Conv", [x, out + "_w", out + "_b"], [out if act is None else out + "_pre"],
                          [attr_ints("kernel_shape", [k, k]), attr_ints("pads", pads), attr_ints("strides", [stride, stride]),
                           attr_i("group", groups)]))
        if act:
            nodes.append(node(act, [out + "_pre"], [out]))
        return out

    conv("X", 3, 16, 3, "c1", [1, 1, 1, 1], act="Relu")
    nodes.append(node("LRN", ["c1"], ["n1"], [attr_i("size", 5), attr_f("alpha", 0.05), attr_f("beta", 0.75), attr_f("bias", 1.5)]))
    nodes.append(node("MaxPool", ["n1"], ["p1"], [attr_ints("kernel_shape", [2, 2]), attr_ints("strides", [2, 2])]))
    h1 = hw // 2
    inits.append(tensor("pads", np.array([0, 0, 0, 1, 0, 0, 2, 1], np.int64)))  # top 0, left 1, bottom 2, right 1
    nodes.append(node("Pad", ["p1", "pads"], ["pp"]))
    conv("pp", 16, 32, 3, "c2", [0, 0, 0, 0], stride=2, act="Relu")
    h2 = (h1 + 2 - 3) // 2 + 1
    nodes.append(node("Split", ["c2"], ["keep", "work"], [attr_i("axis", 1), attr_ints("split", [16, 16])]))
    conv("work", 16, 16, 1, "b1", [0, 0, 0, 0], act="Relu")
    conv("b1", 16, 16, 3, "b2", [1, 1, 1, 1], groups=16)
    conv("b2", 16, 16, 1, "b3", [0, 0, 0, 0], act="Relu")
    nodes.append(node("Concat", ["keep", "b3"], ["cat"], [attr_i("axis", 1)]))
    inits += [tensor("shape5", np.array([0, 2, 16, h2, h2], np.int64)), tensor("shape4", np.array([0, 32, h2, h2], np.int64))]
    nodes += [node("Reshape", ["cat", "shape5"], ["r5"]), node("Transpose", ["r5"], ["t5"], [attr_ints("perm", [0, 2, 1, 3, 4])]),
              node("Reshape", ["t5", "shape4"], ["shuf"])]
    inits += [tensor("s_st", np.array([8], np.int64)), tensor("s_en", np.array([24], np.int64)), tensor("s_ax", np.array([1], np.int64))]
    nodes.append(node("Slice", ["shuf", "s_st", "s_en", "s_ax"], ["mid"]))
    conv("mid", 16, 32, 1, "s1", [0, 0, 0, 0])
    conv("shuf", 32, 32, 1, "s2", [0, 0, 0, 0])
    nodes.append(node("Sum", ["s1", "s2", "shuf"], ["tot"]))
    nodes.append(node("GlobalMaxPool", ["tot"], ["g"]))
    nodes.append(node("Flatten", ["g"], ["f"], [attr_i("axis", 1)]))
    w, b = ws.take((32, classes), 32), ws.take((classes,), 32)
    wts["fc"] = (w, b)
    inits += [tensor("fc_w", w), tensor("fc_b", b)]
    nodes.append(node("Gemm", ["f", "fc_w", "fc_b"], ["Y"]))
    blob = model("zoo_ops", nodes, inits, [value_info("X", ["N", 3, hw, hw])], [value_info("Y", ["N", classes])], opset=13)
    return blob, wts


def write(path: str, blob: bytes) -> str:
    with open(path, "wb") as fh:
        fh.write(blob)
    return path


def sklearn_pipeline(features: int = 30, classes: int = 3, kind: str = "classifier", post: str = "SOFTMAX",
                     labels: Sequence[int] | None = None, normalizer: str | None = None, scaler: bool = True,
                     output: str = "label", seed: int = 99) -> bytes:
    """The graph skl2onnx writes for Pipeline(StandardScaler, LogisticRegression / LinearRegression): ai.onnx.ml
    Scaler -> LinearClassifier (label, scores) [-> Normalizer] or -> LinearRegressor.  `output` picks which value
    is graph output 0 -- the one the reference serves (engine.rs:146-149): "label" (int64 [N]), "scores" ([N,E])."""
    ws = _WeightStream(seed)
    nodes, x = [], "X"
    if scaler:
        off = ws.take((features,), 1)
        sc = (1.0 + 0.5 * ws.take((features,), 1)).astype(np.float32)
        nodes.append(node("Scaler", [x], ["Xs"], [attr_floats("offset", off), attr_floats("scale", sc)], domain=ML_DOMAIN))
        x = "Xs"
    coef = ws.take((classes, features), features)
    icpt = ws.take((classes,), features)
    if kind == "regressor":
        nodes.append(node("LinearRegressor", [x], ["Y"],
                          [attr_floats("coefficients", coef.ravel()), attr_floats("intercepts", icpt),
                           attr_i("targets", classes), attr_s("post_transform", post)], domain=ML_DOMAIN))
        outs = [value_info("Y", ["N", classes])]
    else:
        labels = list(labels) if labels is not None else list(range(classes))
        nodes.append(node("LinearClassifier", [x], ["label", "scores"],
                          [attr_floats("coefficients", coef.ravel()), attr_floats("intercepts", icpt),
                           attr_ints("classlabels_ints", labels), attr_s("post_transform", post)], domain=ML_DOMAIN))
        prob = "scores"
        if normalizer:
            nodes.append(node("Normalizer", ["scores"], ["probabilities"], [attr_s("norm", normalizer)], domain=ML_DOMAIN))
            prob = "probabilities"
        o_label, o_prob = value_info("label", ["N"], INT64), value_info(prob, ["N", classes])
        outs = [o_label, o_prob] if output == "label" else [o_prob, o_label]
    return model("sklearn_pipeline", nodes, [], [value_info("X", ["N", features])], outs, opset=13, ml_opset=1)
