"""Seeded random-graph fuzzing of the ONNX front end and the HIP executor against the oracle.

Each seed builds a small DAG out of the supported operator set -- dense layers, elementwise ops (some fusable
into the producing step, some not), constant and activation-activation binaries, residual joins, feature
concats, softmax heads; or a small conv net (conv / depthwise / BN / pooling / residual) -- with random shapes.
This exercises what hand-written cases do not enumerate: fusion legality (an activation may be folded into
its producer only when it is the sole consumer), buffer aliasing and scratch-slot reuse by liveness, multi-
consumer values, Concat pieces, layout decisions of conv plans.

CPU part: every generated model must load in the oracle AND lower in the product with the same output shape.
GPU part: values must agree within the path's tolerance.
"""
import os

import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth

RTOL, ATOL = 1e-4, 2e-6
UNARY_FUSABLE = ["Relu", "Sigmoid", "Tanh", "LeakyRelu"]
UNARY_PLAIN = ["Softplus", "HardSwish", "Erf", "Softsign", "Abs", "Neg", "Elu", "HardSigmoid", "Gelu"]


class G:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.nodes, self.inits, self.uid = [], [], 0

    def name(self, p):
        self.uid += 1
        return f"{p}{self.uid}"

    def const(self, arr):
        n = self.name("k")
        self.inits.append(W.tensor(n, np.asarray(arr, np.float32)))
        return n

    def weight(self, shape, fan_in):
        return self.const(self.rng.uniform(-1, 1, shape) / np.sqrt(fan_in))

    def op(self, op, ins, attrs=()):
        o = self.name("v")
        self.nodes.append(W.node(op, ins, [o], list(attrs)))
        return o

    def unary(self, x):
        if self.rng.random() < 0.6:
            op = self.rng.choice(UNARY_FUSABLE)
        else:
            op = self.rng.choice(UNARY_PLAIN)
        attrs = []
        if op == "LeakyRelu":
            attrs = [W.attr_f("alpha", float(self.rng.uniform(0.01, 0.3)))]
        if op == "Elu":
            attrs = [W.attr_f("alpha", float(self.rng.uniform(0.5, 1.5)))]
        if op == "Gelu":  # opset 20 operator; emitted in an opset-20 model only
            op = "Tanh"
        return self.op(str(op), [x], attrs)


def dense_graph(seed):
    g = G(seed)
    rng = g.rng
    f_in = int(rng.choice([3, 7, 8, 16, 24, 40, 100, 128, 513]))
    vals = [("X", f_in)]  # (name, width)
    for _ in range(int(rng.integers(3, 9))):
        kind = rng.choice(["dense", "dense", "unary", "binc", "bina", "concat", "matmul_add"])
        src, w = vals[int(rng.integers(0, len(vals)))]
        if kind in ("dense", "matmul_add"):
            m = int(rng.choice([1, 3, 4, 8, 10, 16, 32, 48, 64, 100, 257]))
            wn = g.weight((w, m), w)
            bn = g.weight((m,), w)
            if kind == "dense":
                if rng.random() < 0.3:
                    wt = g.const(np.ascontiguousarray(np.frombuffer(b"", np.float32)))  # placeholder, replaced below
                    g.inits.pop()
                    wn_t = g.weight((m, w), w)
                    o = g.op("Gemm", [src, wn_t, bn], [W.attr_i("transB", 1), W.attr_f("alpha", 0.5), W.attr_f("beta", 2.0)])
                else:
                    o = g.op("Gemm", [src, wn, bn])
            else:
                z = g.op("MatMul", [src, wn])
                o = g.op("Add", [z, bn])
            vals.append((o, m))
        elif kind == "unary":
            vals.append((g.unary(src), w))
        elif kind == "binc":
            c = g.const(rng.uniform(0.5, 1.5, (w,) if rng.random() < 0.7 else (1,)))
            op = str(rng.choice(["Add", "Sub", "Mul", "Div", "Min", "Max"]))
            ins = [src, c] if rng.random() < 0.7 or op in ("Div",) else [c, src]
            vals.append((g.op(op, ins), w))
        elif kind == "bina":
            same = [v for v in vals if v[1] == w and v[0] != src]
            if same:
                other = same[int(rng.integers(0, len(same)))][0]
                vals.append((g.op(str(rng.choice(["Add", "Mul", "Sub", "Max"])), [src, other]), w))
        else:
            others = [vals[int(rng.integers(0, len(vals)))] for _ in range(int(rng.integers(1, 3)))]
            parts = [(src, w)] + others
            vals.append((g.op("Concat", [p[0] for p in parts], [W.attr_i("axis", 1)]), sum(p[1] for p in parts)))
    out, w = vals[-1]
    if out == "X":
        out, w = g.unary("X"), f_in
    if rng.random() < 0.3 and w > 1:
        out = g.op("Softmax", [out], [W.attr_i("axis", 1)])
    g.nodes.append(W.node("Identity", [out], ["Y"]))
    blob = W.model(f"fuzz{seed}", g.nodes, g.inits, [W.value_info("X", ["N", f_in])], [W.value_info("Y", ["N", w])], opset=13)
    return blob, (f_in,), w


def conv_graph(seed):
    g = G(seed)
    rng = g.rng
    c, hw = int(rng.choice([3, 4, 8])), int(rng.choice([8, 12, 16]))
    x, shape = "X", (c, hw, hw)

    def conv(x, cin, cout, k, stride, groups=1, bias=True):
        wn = g.weight((cout, cin // groups, k, k), (cin // groups) * k * k)
        ins = [x, wn] + ([g.weight((cout,), cin)] if bias else [])
        return g.op("Conv", ins, [W.attr_ints("kernel_shape", [k, k]), W.attr_ints("strides", [stride] * 2),
                                  W.attr_ints("pads", [k // 2] * 4), W.attr_i("group", groups)])

    def bn(x, ch):
        ps = [g.const(1 + 0.1 * rng.standard_normal(ch)), g.const(0.1 * rng.standard_normal(ch)), g.const(0.1 * rng.standard_normal(ch)),
              g.const(1 + 0.5 * np.abs(rng.standard_normal(ch)))]
        return g.op("BatchNormalization", [x] + ps, [W.attr_f("epsilon", 1e-5)])

    for _ in range(int(rng.integers(2, 6))):
        cin, h, _ = shape
        kind = rng.choice(["conv", "conv", "dw", "pool", "res", "bn_act", "se", "cat", "unary"])
        if kind == "conv":
            cout = int(rng.choice([4, 8, 16, 32, 64]))
            k, stride = int(rng.choice([1, 3])), int(rng.choice([1, 1, 2]))
            x = conv(x, cin, cout, k, stride, bias=bool(rng.random() < 0.5))
            if rng.random() < 0.6:
                x = bn(x, cout)
            if rng.random() < 0.7:
                x = g.op(str(rng.choice(["Relu", "LeakyRelu", "HardSwish"])), [x])
            shape = (cout, (h + 2 * (k // 2) - k) // stride + 1, (h + 2 * (k // 2) - k) // stride + 1)
        elif kind == "dw" and cin % 4 == 0:
            x = conv(x, cin, cin, 3, 1, groups=cin)
            x = g.op("Clip", [x, g.const(np.array(0.0).reshape(())), g.const(np.array(6.0).reshape(()))])
        elif kind == "pool" and h >= 4:
            mx = rng.random() < 0.5
            if rng.random() < 0.5:
                x = g.op("MaxPool" if mx else "AveragePool", [x], [W.attr_ints("kernel_shape", [2, 2]), W.attr_ints("strides", [2, 2])])
                shape = (cin, h // 2, h // 2)
            else:  # 3x3 / stride 2 / pad 1 with ceil_mode
                x = g.op("MaxPool" if mx else "AveragePool", [x], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("strides", [2, 2]),
                                                                 W.attr_ints("pads", [1, 1, 1, 1]), W.attr_i("ceil_mode", 1)])
                o = -(-(h + 2 - 3) // 2) + 1
                if (o - 1) * 2 >= h + 1:
                    o -= 1
                shape = (cin, o, o)
        elif kind == "res":
            y = conv(x, cin, cin, 3, 1)
            y = g.op("Relu", [y])
            y = conv(y, cin, cin, 3, 1, bias=False)
            x = g.op("Relu", [g.op("Add", [y, x])])
        elif kind == "se" and cin % 4 == 0 and h > 1:
            sq = g.op("GlobalAveragePool", [x])
            r = g.op("Relu", [conv(sq, cin, max(4, cin // 4), 1, 1)])
            e = g.op(str(rng.choice(["Sigmoid", "HardSigmoid"])), [conv(r, max(4, cin // 4), cin, 1, 1)])
            x = g.op("Mul", [x, e] if rng.random() < 0.5 else [e, x])
        elif kind == "cat":
            cout = int(rng.choice([4, 8, 16]))
            y = g.op("Relu", [conv(x, cin, cout, 3, 1)])
            x = g.op("Concat", [x, y] if rng.random() < 0.5 else [y, x], [W.attr_i("axis", 1)])
            shape = (cin + cout, h, h)
        elif kind == "unary":
            x = g.op(str(rng.choice(["Sigmoid", "Tanh", "Softplus", "HardSwish", "Abs"])), [x])
        else:
            x = g.op("Relu", [bn(x, cin)])
    cin = shape[0]
    if rng.random() < 0.75:
        f = g.op("Flatten", [g.op("GlobalAveragePool", [x])], [W.attr_i("axis", 1)])
        feat = cin
    else:
        f = g.op("Flatten", [x], [W.attr_i("axis", 1)])  # NCHW-ordered features: forces the NCHW plan
        feat = cin * shape[1] * shape[2]
    m = int(rng.choice([1, 5, 10]))
    g.nodes.append(W.node("Gemm", [f, g.weight((feat, m), feat), g.weight((m,), feat)], ["Y"]))
    blob = W.model(f"fuzzc{seed}", g.nodes, g.inits, [W.value_info("X", ["N", c, hw, hw])], [W.value_info("Y", ["N", m])], opset=14)
    return blob, (c, hw, hw), m


def build(kind, seed, tmp_path):
    blob, in_shape, out_w = (dense_graph if kind == "dense" else conv_graph)(seed)
    return W.write(str(tmp_path / f"{kind}{seed}.onnx"), blob), in_shape, out_w


# INFERA_FUZZ_SEEDS=<n>: a longer one-off sweep (n seeds per kind, past the committed ones) -- e.g. 2000 on a GPU box after a kernel change
_EXTRA = int(os.environ.get("INFERA_FUZZ_SEEDS", "0"))
DENSE_SEEDS = list(range(120)) + list(range(100000, 100000 + _EXTRA))
CONV_SEEDS = list(range(1000, 1120)) + list(range(200000, 200000 + _EXTRA))


@pytest.mark.parametrize("kind,seeds", [("dense", DENSE_SEEDS), ("conv", CONV_SEEDS)])
def test_fuzzed_graphs_load_in_oracle_and_lower(built, tmp_path, kind, seeds):
    from infera_amd import capi
    from oracle import oracle

    for seed in seeds:
        path, in_shape, out_w = build(kind, seed, tmp_path)
        om = oracle.Model(path)
        name = f"fz_{kind}{seed}"
        capi.load_model(name, path)
        info = capi.get_model_info(name)
        assert info["input_shape"] == [-1] + list(in_shape) == om.input_shape, (seed, info)
        assert info["output_shape"] == [-1, out_w] == om.output_shape, (seed, info, om.output_shape)
        capi.unload_model(name)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seeds", [("dense", DENSE_SEEDS), ("conv", CONV_SEEDS)])
def test_fuzzed_graphs_match_oracle_on_gpu(built, tmp_path, kind, seeds):
    from infera_amd import capi
    from oracle import oracle

    worst = 0.0
    for seed in seeds:
        path, in_shape, out_w = build(kind, seed, tmp_path)
        rows = int(np.random.default_rng(seed).choice([1, 3, 33, 130, 1000 if kind == "dense" else 20]))
        x = synth.table(1000 + seed, 0, rows, int(np.prod(in_shape)))
        name = f"fg_{kind}{seed}"
        capi.load_model(name, path)
        got = capi.predict(name, x) if kind == "dense" else capi.predict_from_blob(name, x.tobytes())
        want = oracle.Model(path).predict(x) if kind == "dense" else oracle.Model(path).predict_blob(x.tobytes())
        assert got.shape == want.shape == (rows, out_w), (seed, got.shape, want.shape)
        err = np.abs(got.astype(np.float64) - want.astype(np.float64))
        tol = RTOL * np.abs(want.astype(np.float64)) + ATOL
        assert (err <= tol).all(), f"{kind} seed {seed}: {int((err > tol).sum())}/{err.size} out of tolerance, worst {err.max():.3e}; plan {capi.get_plan(name)['exec']}"
        worst = max(worst, float((err / (np.abs(want) + 1e-3)).max()))
        capi.unload_model(name)
    assert worst < 1e-3
