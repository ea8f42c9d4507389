"""Well-formed protobuf, hostile contents: declared shapes that overflow size_t, zero strides, axes far outside the
rank, negative dilations ...  A model path comes from SQL (and may be an http:// URL), so `infera_load_model` must
answer -1 with an error text for every one of them -- never SIGSEGV / SIGFPE / hang the host DuckDB process (the
reference gets this from tract's checked Rust arithmetic, engine.rs:49-56).  Each case runs in a child process so a
crash is observable as a return code."""
from __future__ import annotations

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, random, struct
sys.path.insert(0, %(root)r)
import numpy as np
from infera_amd import capi, onnx_writer as W

def raw_tensor(name, dims, dtype, payload=b"", field=9):
    out = b"".join(W._vi(1, d) for d in dims) + W._vi(2, dtype)
    if payload or field == 9:
        out += W._ld(field, payload)
    return out + W._s(8, name)

def matmul_model(wt):
    return W.model("g", [W.node("MatMul", ["X", "W"], ["Y"])], [wt], [W.value_info("X", ["N", 4])], [W.value_info("Y", ["N", 4])])

def conv_model(attrs, op="Conv"):
    w = np.ones((4, 4, 3, 3), np.float32)
    ins = ["X", "W"] if op == "Conv" else ["X"]
    inits = [W.tensor("W", w)] if op == "Conv" else []
    return W.model("g", [W.node(op, ins, ["Y"], attrs)], inits, [W.value_info("X", ["N", 4, 8, 8])], [W.value_info("Y", ["N", 4, 8, 8])])

def cases():
    F, I64, D = 1, 7, 11
    # --- ADVICE r1 high: dims whose product wraps size_t
    yield "dims_wrap_to_0", matmul_model(raw_tensor("W", [4, 1 << 62], F))
    yield "dims_2^32x2^32", matmul_model(raw_tensor("W", [1 << 32, 1 << 32], F))
    yield "dims_wrap_double", matmul_model(raw_tensor("W", [4, 1 << 62], D))
    yield "dims_wrap_int64", matmul_model(raw_tensor("W", [4, 1 << 62], I64))
    yield "dims_wrap_to_16", matmul_model(raw_tensor("W", [4, (1 << 62) + 1], F, b"\0" * 16))
    yield "empty_but_huge", matmul_model(raw_tensor("W", [0, 1 << 62], F))
    yield "huge_no_payload_double", matmul_model(raw_tensor("W", [1 << 40], D, b"", field=10))
    yield "huge_no_payload_i64", matmul_model(raw_tensor("W", [1 << 40], I64, b"", field=7))
    yield "short_payload", matmul_model(raw_tensor("W", [4, 4], F, b"\0" * 60))
    # --- ADVICE r1 medium: attributes used unchecked
    for op in ("Conv", "MaxPool", "AveragePool"):
        ks = [] if op == "Conv" else [W.attr_ints("kernel_shape", [3, 3])]
        yield op + "_stride0", conv_model(ks + [W.attr_ints("strides", [0, 0])], op)
        yield op + "_stride_neg", conv_model(ks + [W.attr_ints("strides", [-1, 1])], op)
        yield op + "_dil0", conv_model(ks + [W.attr_ints("dilations", [0, 1])], op)
        yield op + "_dil_neg", conv_model(ks + [W.attr_ints("dilations", [1, -3])], op)
        yield op + "_pads_neg", conv_model(ks + [W.attr_ints("pads", [-5, 0, 0, 0])], op)
        yield op + "_stride_huge", conv_model(ks + [W.attr_ints("strides", [1 << 62, 1 << 62])], op)
        yield op + "_pads_huge", conv_model(ks + [W.attr_ints("pads", [1 << 62, 1 << 62, 1 << 62, 1 << 62])], op)
    yield "pool_kernel0", conv_model([W.attr_ints("kernel_shape", [0, 0])], "MaxPool")
    yield "pool_kernel_neg", conv_model([W.attr_ints("kernel_shape", [-2, 3])], "AveragePool")
    for axes in ([2, 3, 1000000], [-1000000], [1 << 62], [2, 3, -(1 << 62)]):
        yield "reduce_mean_axes_%%s" %% axes[-1], W.model("g", [W.node("ReduceMean", ["X"], ["Y"], [W.attr_ints("axes", axes)])], [],
                                                        [W.value_info("X", ["N", 4, 8, 8])], [W.value_info("Y", ["N", 4, 1, 1])])
    for ax in (99, -99, 1 << 40):
        yield "softmax_axis_%%d" %% ax, W.model("g", [W.node("Softmax", ["X"], ["Y"], [W.attr_i("axis", ax)])], [],
                                               [W.value_info("X", ["N", 6])], [W.value_info("Y", ["N", 6])])
        yield "unsqueeze_axes_%%d" %% ax, W.model("g", [W.node("Unsqueeze", ["X"], ["Y"], [W.attr_ints("axes", [ax])])], [],
                                                 [W.value_info("X", ["N", 6])], [W.value_info("Y", ["N", 6, 1])], opset=11)
        yield "squeeze_axes_%%d" %% ax, W.model("g", [W.node("Squeeze", ["X"], ["Y"], [W.attr_ints("axes", [ax])])], [],
                                               [W.value_info("X", ["N", 6, 1])], [W.value_info("Y", ["N", 6])], opset=11)
        yield "concat_axis_%%d" %% ax, W.model("g", [W.node("Concat", ["X", "X"], ["Y"], [W.attr_i("axis", ax)])], [],
                                              [W.value_info("X", ["N", 6])], [W.value_info("Y", ["N", 12])])
        yield "flatten_axis_%%d" %% ax, W.model("g", [W.node("Flatten", ["X"], ["Y"], [W.attr_i("axis", ax)])], [],
                                               [W.value_info("X", ["N", 6, 2])], [W.value_info("Y", ["N", 12])])
        yield "argmax_axis_%%d" %% ax, W.model("g", [W.node("ArgMax", ["X"], ["Y"], [W.attr_i("axis", ax)])], [],
                                              [W.value_info("X", ["N", 6])], [W.value_info("Y", ["N", 1])])
    # Clip with a vector bound: rejected, not silently ignored (ADVICE r1 low)
    yield "clip_vector_min", W.model("g", [W.node("Clip", ["X", "lo"], ["Y"])], [W.tensor("lo", np.zeros(6, np.float32))],
                                     [W.value_info("X", ["N", 6])], [W.value_info("Y", ["N", 6])])
    # slices / gathers / splits with absurd bounds
    big = 1 << 62
    for st, en in ((big, -big), (-big, big), (3, 2), (7, 9)):
        yield "slice_%%d_%%d" %% (st %% 97, en %% 97), W.model(
            "g", [W.node("Slice", ["X", "s", "e", "a"], ["Y"])],
            [W.tensor("s", np.array([st], np.int64)), W.tensor("e", np.array([en], np.int64)), W.tensor("a", np.array([1], np.int64))],
            [W.value_info("X", ["N", 6])], [W.value_info("Y", ["N", 6])])
    for idx in ([100], [-100], [big], [0, 5, 3]):
        yield "gather_%%d" %% (idx[0] %% 97), W.model("g", [W.node("Gather", ["X", "i"], ["Y"], [W.attr_i("axis", 1)])],
                                                    [W.tensor("i", np.array(idx, np.int64))], [W.value_info("X", ["N", 6])], [W.value_info("Y", ["N", 1])])
    yield "split_sizes", W.model("g", [W.node("Split", ["X"], ["Y", "Z"], [W.attr_i("axis", 1), W.attr_ints("split", [big, -big + 6])])], [],
                                 [W.value_info("X", ["N", 6])], [W.value_info("Y", ["N", 3])], opset=11)
    yield "reshape_huge", W.model("g", [W.node("Reshape", ["X", "s"], ["Y"])], [W.tensor("s", np.array([0, big, big], np.int64))],
                                  [W.value_info("X", ["N", 6])], [W.value_info("Y", ["N", 6])])
    yield "reshape_neg", W.model("g", [W.node("Reshape", ["X", "s"], ["Y"])], [W.tensor("s", np.array([0, -1, -1], np.int64))],
                                 [W.value_info("X", ["N", 6])], [W.value_info("Y", ["N", 6])])
    yield "lrn_size_huge", W.model("g", [W.node("LRN", ["X"], ["Y"], [W.attr_i("size", big)])], [],
                                   [W.value_info("X", ["N", 4, 8, 8])], [W.value_info("Y", ["N", 4, 8, 8])])
    yield "conv_group0", conv_model([W.attr_i("group", 0)])
    yield "conv_group_neg", conv_model([W.attr_i("group", -4)])
    yield "input_dim_huge", W.model("g", [W.node("Relu", ["X"], ["Y"])], [], [W.value_info("X", ["N", big])], [W.value_info("Y", ["N", big])])
    yield "input_dims_wrap", W.model("g", [W.node("Relu", ["X"], ["Y"])], [], [W.value_info("X", ["N", 1 << 32, 1 << 32])], [W.value_info("Y", ["N", 1 << 32, 1 << 32])])

only = %(only)r
d = %(tmp)r
for name, blob in cases():
    if only and name != only:
        continue
    p = os.path.join(d, "h.onnx")
    with open(p, "wb") as f:
        f.write(blob)
    try:
        capi.load_model("hostile", p)
        info = capi.get_model_info("hostile")
        capi.unload_model("hostile")
        print("LOADED", name, info, flush=True)
    except capi.InferaError as e:
        print("REJECTED", name, str(e)[:100], flush=True)
print("DONE")
"""


def _run(tmp_path, only=""):
    code = CHILD % {"root": ROOT, "tmp": str(tmp_path), "only": only}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)


def test_hostile_models_are_rejected_not_fatal(built, tmp_path):
    out = _run(tmp_path)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and lines and lines[-1] == "DONE", (
        f"child died (rc={out.returncode}) after: {lines[-1] if lines else '<nothing>'}\n{out.stderr[-2000:]}")
    verdict = {l.split()[1]: l.split()[0] for l in lines[:-1]}
    # the ones an earlier build crashed or hung on must be clean rejections
    for name in ("dims_wrap_to_0", "dims_2^32x2^32", "dims_wrap_double", "dims_wrap_int64", "dims_wrap_to_16", "empty_but_huge",
                 "Conv_stride0", "MaxPool_stride0", "AveragePool_stride0", "Conv_dil_neg", "Conv_pads_neg", "pool_kernel0",
                 "reduce_mean_axes_1000000", "clip_vector_min", "conv_group0"):
        assert verdict[name] == "REJECTED", (name, verdict[name])
    assert len(verdict) > 60
