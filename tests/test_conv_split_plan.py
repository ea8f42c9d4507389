"""The convolution arithmetic's HOST logic, without a GPU (the plan is made at infera_load_model): which steps the default plan moves to the
bf16 matrix cores (three exact parts per operand), that INFERA_PRECISION is read when a model is scheduled, that an unknown mode name does not
silently select an arithmetic, and which layers stay on the exact-fp32 kernels because the split kernels do not take their shape."""
import os

import numpy as np

from infera_amd import capi
from infera_amd import onnx_writer as W


def _plan(tmp_path, name, blob, precision=None):
    path = W.write(str(tmp_path / f"{name}.onnx"), blob)
    if precision:
        os.environ["INFERA_PRECISION"] = precision
    try:
        capi.load_model("splitplan_" + name, path)
    finally:
        os.environ.pop("INFERA_PRECISION", None)
    try:
        return capi.get_plan("splitplan_" + name)
    finally:
        capi.unload_model("splitplan_" + name)


def test_resnet18_plan_default_and_fp32(built, tmp_path):
    blob = W.resnet18()
    plain = _plan(tmp_path, "rn_plain", blob, "fp32")
    default = _plan(tmp_path, "rn_default", blob)
    assert plain["exec"][0] == "conv_patch_pool" and plain["exec"].count("conv_tiled_cq") == 19 and "conv_precision" not in plain
    # the default: the stem and the same 19 layers on the bf16 matrix cores with three exact parts per operand; no maxima, so no extra scratch
    # ... and each block's 1x1 projection shortcut (three of them) folded into the block's second convolution as extra K stages: sixteen launches
    assert default["exec"][0] == "conv_patch_pool_bf16x6" and default["exec"].count("conv_split_bf16x6") == 16 and "bf16x6" in default["conv_precision"]
    assert len(default["folded_shortcuts"]) == 3 and all(default["exec"][i] == "skipped" for i in default["folded_shortcuts"])
    assert default["scratch_floats_per_row"] <= plain["scratch_floats_per_row"]
    os.environ["INFERA_CONV_FOLD_SHORTCUT"] = "0"
    try:
        unfolded = _plan(tmp_path, "rn_unfolded", blob)
    finally:
        os.environ.pop("INFERA_CONV_FOLD_SHORTCUT", None)
    assert unfolded["exec"].count("conv_split_bf16x6") == 19 and "folded_shortcuts" not in unfolded
    # same steps, same fusions (residual adds in the epilogues, the head on the exact-fp32 tiled kernel): only the names of the moved steps differ
    assert [{"conv_split_bf16x6": "conv_tiled_cq", "conv_patch_pool_bf16x6": "conv_patch_pool"}.get(e, e) for e in unfolded["exec"]] == plain["exec"]
    assert "folded_shortcuts" not in plain  # (the exact-fp32 kernels keep the shortcut as its own launch)
    # a mode name this build does not know (a typo, a mode of an earlier round) must not silently pick an arithmetic: default + a warning
    assert _plan(tmp_path, "rn_typo", blob, "f16x3")["exec"] == default["exec"]
    assert _plan(tmp_path, "rn_named", blob, "bf16x6")["exec"] == default["exec"]


def test_layers_the_split_kernels_do_not_take_stay_exact(built, tmp_path):
    rng = np.random.default_rng(1)

    def net(c_in, convs, hw=12):
        nodes, inits, x, c = [], [], "X", c_in
        for i, (cout, k, groups) in enumerate(convs):
            w = (rng.standard_normal((cout, c // groups, k, k)) * 0.1).astype(np.float32)
            inits.append(W.tensor(f"w{i}", w))
            nodes.append(W.node("Conv", [x, f"w{i}"], [f"c{i}"], [W.attr_ints("kernel_shape", [k, k]), W.attr_ints("pads", [k // 2] * 4), W.attr_i("group", groups)]))
            nodes.append(W.node("Relu", [f"c{i}"], [f"r{i}"]))
            x, c = f"r{i}", cout
        nodes += [W.node("GlobalAveragePool", [x], ["g"]), W.node("Flatten", ["g"], ["Y"], [W.attr_i("axis", 1)])]
        return W.model("n", nodes, inits, [W.value_info("X", ["N", c_in, hw, hw])], [W.value_info("Y", ["N", c])])

    # 4 -> 24 (padded-channel kernel), 24 -> 48 (channels not multiples of 32: padded-channel kernel), depthwise 48, 48 -> 64 1x1 (C % 32 != 0)
    p = _plan(tmp_path, "mobile", net(4, [(24, 3, 1), (48, 3, 1), (48, 3, 48), (64, 1, 1)]))
    assert "conv_split_bf16x6" not in p["exec"] and "conv_precision" not in p
    # 4 -> 64, then 64 -> 64 in two groups (generic kernel), then 64 -> 128 3x3 (split), 128 -> 96 (features no multiple of 64: exact-fp32 tiled kernel)
    p = _plan(tmp_path, "grouped", net(4, [(64, 3, 1), (64, 3, 2), (128, 3, 1), (96, 3, 1)]))
    assert p["exec"][:4] == ["conv_patch", "normal", "conv_split_bf16x6", "conv_tiled_cq"]  # (activations ride in the conv steps)
    # a model whose INPUT already has 32 channels: the caller's tensor is NCHW, its first convolution is not a channel-quad one
    p = _plan(tmp_path, "wide_in", net(32, [(64, 3, 1), (64, 3, 1)]))
    assert p["exec"][0] != "conv_split_bf16x6" and p["exec"].count("conv_split_bf16x6") == 1


def shortcut_block(p_between: bool, hw: int = 12, seed: int = 5) -> bytes:
    """out = Relu(Conv3x3(A) + Conv1x1(P)) with P = Relu(Conv3x3(A) + A).  `p_between`: P's nodes stand BETWEEN the block's 3x3 convolution and
    its 1x1 shortcut in the graph -- a valid ONNX order in which the shortcut's input does not exist yet when the 3x3 layer runs."""
    rng = np.random.default_rng(seed)
    t = lambda name, *shape: W.tensor(name, (rng.standard_normal(shape) / np.sqrt(max(1, int(np.prod(shape[1:]))))).astype(np.float32))
    conv = lambda x, w, b, y, k: W.node("Conv", [x, w, b], [y], [W.attr_ints("kernel_shape", [k, k]), W.attr_ints("pads", [k // 2] * 4)])
    head = [conv("X", "w0", "b0", "c0", 3), W.node("Relu", ["c0"], ["A"])]
    y_path = [conv("A", "wy", "by", "y", 3)]
    p_path = [conv("A", "wt", "bt", "t", 3), W.node("Add", ["t", "A"], ["ta"]), W.node("Relu", ["ta"], ["P"])]
    tail = [conv("P", "ws", "bs", "s", 1), W.node("Add", ["y", "s"], ["ys"]), W.node("Relu", ["ys"], ["o"]),
            W.node("GlobalAveragePool", ["o"], ["g"]), W.node("Flatten", ["g"], ["Y"], [W.attr_i("axis", 1)])]
    nodes = head + (y_path + p_path if p_between else p_path + y_path) + tail
    inits = [t("w0", 128, 4, 3, 3), t("b0", 128), t("wy", 128, 128, 3, 3), t("by", 128), t("wt", 128, 128, 3, 3), t("bt", 128), t("ws", 128, 128, 1, 1), t("bs", 128)]
    return W.model("blk", nodes, inits, [W.value_info("X", ["N", 4, hw, hw])], [W.value_info("Y", ["N", 128])])


def test_a_shortcut_whose_input_is_produced_after_the_second_convolution_is_not_folded(built, tmp_path):
    """the fold moves the read of the shortcut's input up into the block's 3x3 convolution.  That input may be the output of
    a FUSED epilogue (conv + Add + Relu) that stands between the two layers; the old check asked the per-step producer table, which knows no
    such buffer, and folded -- reading P before it was written.  Same arithmetic with P's nodes first: folds as before."""
    late = _plan(tmp_path, "p_between", shortcut_block(True))
    early = _plan(tmp_path, "p_first", shortcut_block(False))
    assert "folded_shortcuts" not in late, late["exec"]
    assert len(early["folded_shortcuts"]) == 1 and early["exec"][early["folded_shortcuts"][0]] == "skipped"
    # both plans run the split kernels; the unfolded one has one more launch (the 1x1 layer with the Add in its epilogue)
    assert late["exec"].count("conv_split_bf16x6") == early["exec"].count("conv_split_bf16x6") + 1 == 3
