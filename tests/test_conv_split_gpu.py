"""The default convolution arithmetic (csrc/hip/conv_split.hip, conv.hip's conv2d_stem_split6_kernel): the tiled convolutions and the 7x7/2
stem on the bf16 matrix cores, every fp32 operand cut EXACTLY into three bf16 parts, six partial products per product, fp32 accumulation --
no scales, no maxima, no precondition on the data.  INFERA_PRECISION=fp32 (read when a model is scheduled, so one process loads the same
network both ways) selects the exact-fp32 kernels.

Checked here: the plan says which steps run split; the results are within the parity tolerance (1e-4 |y| + 1e-6) of the oracle AND within a
few fp32 roundings of it relative to each row's scale; a row's result does not depend on its batch (bit for bit); rows of wildly different
magnitude (1e-30 ... 1e+25, all zeros) in one batch each keep their own relative accuracy."""
import os

import numpy as np
import pytest

from infera_amd import synth
from infera_amd import onnx_writer as W
from tests.test_conv_ws_gpu import CASES, _net

# (feature tiles per workgroup) x (channel blocks) x (3x3 | 1x1, i.e. many stages | a single stage), image sizes that leave ragged last tiles
VARIANTS = {
    "mt2_s1_3x3": [(32, 3, 1), (64, 3, 1)], "mt2_s1_3blocks": [(96, 3, 1), (64, 3, 1)], "mt2_s2_3x3": [(64, 3, 1), (64, 3, 1)],
    "mt4_s1_3x3": [(32, 3, 1), (128, 3, 1)], "mt4_s2_3x3": [(64, 3, 1), (128, 3, 2)],
    "mt2_s1_1x1": [(32, 3, 1), (64, 1, 1)], "mt4_s2_1x1_s2": [(64, 3, 1), (128, 1, 2)], "mt2_s2_1x1": [(64, 3, 1), (64, 1, 1)],
    "mt4_s2_2stages": [(128, 3, 1), (128, 1, 1)],
    # the software-pipelined 128-feature kernel at its shortest: ONE stage (it issues two stages ahead: everything past the end must be harmless), three stages
    "mt4_one_stage": [(32, 3, 1), (128, 1, 1)], "mt4_three_stages": [(96, 3, 1), (128, 1, 1)],
}


def _random_conv_case(seed):
    """stem 4 -> C (3x3, padded-channel fp32 kernel) + ReLU, then ONE random split-eligible convolution C -> M, global average."""
    rng = np.random.default_rng(seed)
    C = int(rng.choice([32, 64, 96, 128, 160])); M = int(rng.choice([32, 64, 96, 128, 192, 256]))
    kh, kw = int(rng.choice([1, 2, 3, 5])), int(rng.choice([1, 2, 3, 5]))
    sh, sw = int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2]))
    dh, dw = int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2]))
    hw = int(rng.integers(max(5, (kh - 1) * dh + 1, (kw - 1) * dw + 1), 25))
    pads = [int(rng.integers(0, (kh - 1) * dh // 2 + 2)), int(rng.integers(0, (kw - 1) * dw // 2 + 2)),
            int(rng.integers(0, (kh - 1) * dh // 2 + 2)), int(rng.integers(0, (kw - 1) * dw // 2 + 2))]
    rows = int(rng.integers(1, 10))
    w0 = (rng.standard_normal((C, 4, 3, 3)) / 6.0).astype(np.float32)
    b0 = (rng.standard_normal(C) * 0.1).astype(np.float32)
    w1 = (rng.standard_normal((M, C, kh, kw)) / np.sqrt(C * kh * kw)).astype(np.float32)
    b1 = (rng.standard_normal(M) * 0.1).astype(np.float32)
    nodes = [W.node("Conv", ["X", "w0", "b0"], ["c0"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("pads", [1] * 4)]), W.node("Relu", ["c0"], ["r0"]),
             W.node("Conv", ["r0", "w1", "b1"], ["c1"], [W.attr_ints("kernel_shape", [kh, kw]), W.attr_ints("strides", [sh, sw]), W.attr_ints("dilations", [dh, dw]),
                                                         W.attr_ints("pads", pads)]),
             W.node("Relu", ["c1"], ["r1"]), W.node("GlobalAveragePool", ["r1"], ["g"]), W.node("Flatten", ["g"], ["Y"], [W.attr_i("axis", 1)])]
    blob = W.model(f"rc{seed}", nodes, [W.tensor("w0", w0), W.tensor("b0", b0), W.tensor("w1", w1), W.tensor("b1", b1)],
                   [W.value_info("X", ["N", 4, hw, hw])], [W.value_info("Y", ["N", M])])
    return blob, hw, rows, dict(C=C, M=M, k=(kh, kw), s=(sh, sw), d=(dh, dw), pads=pads, hw=hw, rows=rows)


# ---- the DEFAULT convolution form (bf16x6): operands cut exactly into three bf16 parts, six partial products; no scales, no maxima, no precondition ----
def _load_mode(gpu_api, path, name, precision):
    os.environ["INFERA_PRECISION"] = precision
    try:
        gpu_api.load_model(name, path)
    finally:
        os.environ.pop("INFERA_PRECISION", None)



@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_gpu_bf16x6_conv_chains_match_oracle(gpu_api, tmp_path, case):
    """ResNet-like chains (residual adds fused into epilogues, stride-2 entries, 1x1 layers, a layer the split kernel does not take in the
    middle) on the default plan: vs the oracle, repeatable, a row alone == the row in its batch."""
    from oracle import oracle

    c = CASES[case]
    path = W.write(str(tmp_path / "net.onnx"), _net(c["chain"], c["cin"], c["hw"], c["residual_at"]))
    x = synth.table(31, 0, c["rows"], c["cin"] * c["hw"] * c["hw"])
    _load_mode(gpu_api, path, "conv_fp32", "fp32")
    gpu_api.load_model("conv_bf6", path)
    try:
        assert "conv_split_bf16x6" in gpu_api.get_plan("conv_bf6")["exec"] and "conv_split_bf16x6" not in gpu_api.get_plan("conv_fp32")["exec"]
        got = gpu_api.predict_from_blob("conv_bf6", x.tobytes())
        assert np.array_equal(got, gpu_api.predict_from_blob("conv_bf6", x.tobytes()))
        for r in (0, c["rows"] - 1):
            assert np.array_equal(gpu_api.predict_from_blob("conv_bf6", x[r].tobytes()).reshape(-1), got.reshape(c["rows"], -1)[r])
        ref32 = gpu_api.predict_from_blob("conv_fp32", x.tobytes())
    finally:
        gpu_api.unload_model("conv_bf6")
        gpu_api.unload_model("conv_fp32")
    want = oracle.Model(path).predict_blob(x.tobytes())
    err = np.abs(got - want)
    assert np.all(err <= 1e-4 * np.abs(want) + 1e-6), err.max()
    scale = np.abs(want).max()
    assert err.max() <= max(1.5e-6 * scale, 1.5 * np.abs(ref32 - want).max()), (err.max() / scale, np.abs(ref32 - want).max() / scale)


@pytest.mark.gpu
def test_gpu_bf16x6_folded_projection_shortcuts_match_the_two_launch_plan_and_the_oracle(gpu_api, tmp_path):
    """Round 4: a ResNet block's 1x1 projection shortcut computed as extra K stages of the block's second convolution (one accumulator, one
    launch) against INFERA_CONV_FOLD_SHORTCUT=0 (the shortcut as its own launch, its result added in an epilogue): the same sums in another
    order -- equal to a few fp32 roundings of the output scale -- and both within the parity bar of the oracle; a row alone == the row in its
    batch; odd image sizes (stride-2 shortcuts over 33 x 33 and 17 x 17 maps)."""
    from oracle import oracle

    for hw in (64, 66):
        path = W.write(str(tmp_path / f"rn{hw}.onnx"), W.resnet18(classes=10, in_hw=hw, width=64))
        x = synth.table(13, 0, 5, 3 * hw * hw)
        gpu_api.load_model("folded", path)
        os.environ["INFERA_CONV_FOLD_SHORTCUT"] = "0"
        try:
            gpu_api.load_model("unfolded", path)
        finally:
            os.environ.pop("INFERA_CONV_FOLD_SHORTCUT", None)
        try:
            assert len(gpu_api.get_plan("folded")["folded_shortcuts"]) == 3 and "folded_shortcuts" not in gpu_api.get_plan("unfolded")
            a = gpu_api.predict_from_blob("folded", x.tobytes())
            b = gpu_api.predict_from_blob("unfolded", x.tobytes())
            assert np.array_equal(a, gpu_api.predict_from_blob("folded", x.tobytes()))
            assert np.array_equal(a[3], gpu_api.predict_from_blob("folded", x[3].tobytes()).reshape(-1))
        finally:
            gpu_api.unload_model("folded")
            gpu_api.unload_model("unfolded")
        want = oracle.Model(path).predict_blob(x.tobytes())
        scale = np.abs(want).max()
        assert np.abs(a - b).max() <= 2e-6 * scale, np.abs(a - b).max() / scale
        for got in (a, b):
            assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-6), np.abs(got - want).max()


@pytest.mark.gpu
def test_gpu_shortcut_input_produced_between_the_two_convolutions(gpu_api, tmp_path):
    """out = Relu(Conv3x3(A) + Conv1x1(P)) with P's producer (a convolution carrying a fused Add) standing between the two layers in
    the graph -- not folded (tests/test_conv_split_plan.py); with P first it is.  Both orders against the oracle, and against each other."""
    from oracle import oracle
    from tests.test_conv_split_plan import shortcut_block

    x = synth.table(17, 0, 6, 4 * 12 * 12)
    got = {}
    for between in (True, False):
        path = W.write(str(tmp_path / f"blk{int(between)}.onnx"), shortcut_block(between))
        gpu_api.load_model("blk", path)
        try:
            assert ("folded_shortcuts" in gpu_api.get_plan("blk")) == (not between)
            got[between] = gpu_api.predict_from_blob("blk", x.tobytes())
        finally:
            gpu_api.unload_model("blk")
        want = oracle.Model(path).predict_blob(x.tobytes())
        assert np.all(np.abs(got[between] - want) <= 1e-4 * np.abs(want) + 1e-6), (between, np.abs(got[between] - want).max())
    assert np.abs(got[True] - got[False]).max() <= 2e-6 * np.abs(got[False]).max()


@pytest.mark.gpu
def test_gpu_bf16x6_non_finite_rows_stay_in_their_rows(gpu_api, tmp_path):
    """ResNet-18 (64 x 64 images, full width) over a batch that mixes magnitudes: a NaN or an infinity in one image leaves every OTHER image of
    the batch bit for bit what it is without the poisoned neighbour (no scales: nothing of one row reaches another).  What the poisoned rows
    themselves return is not pinned: an infinity reaches its outputs as NaN in three-part arithmetic (fp32 arithmetic: +-inf or NaN depending
    on the weights), and Relu maps NaN to 0 here as in the exact-fp32 kernels and the oracle."""
    path = W.write(str(tmp_path / "rn64.onnx"), W.resnet18(classes=10, in_hw=64, width=64))
    mags = np.array([1.0, 1e-20, 1e20, 0.0, 255.0, 1.0], np.float32)
    clean = (synth.table(21, 0, 6, 3 * 64 * 64) * mags[:, None]).astype(np.float32)
    bad = clean.copy()
    bad[0, 5000] = np.nan
    bad[4, 77] = np.inf
    gpu_api.load_model("conv_bf6", path)
    try:
        plan = gpu_api.get_plan("conv_bf6")  # (sixteen launches: the three 1x1 projection shortcuts ride in their blocks' second convolutions)
        assert plan["exec"].count("conv_split_bf16x6") == 16 and len(plan["folded_shortcuts"]) == 3
        y = gpu_api.predict_from_blob("conv_bf6", clean.tobytes())
        y_bad = gpu_api.predict_from_blob("conv_bf6", bad.tobytes())
    finally:
        gpu_api.unload_model("conv_bf6")
    assert np.all(np.isfinite(y))
    for r in (1, 2, 3, 5):
        assert np.array_equal(y_bad[r], y[r]), r
    assert not np.array_equal(y_bad[0], y[0]) and not np.array_equal(y_bad[4], y[4])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["mt2_s1_3x3", "mt2_s1_3blocks", "mt2_s2_3x3", "mt4_s1_3x3", "mt4_s2_3x3", "mt2_s1_1x1", "mt4_s2_1x1_s2", "mt2_s2_1x1",
                                     "mt4_s2_2stages", "mt4_one_stage", "mt4_three_stages"])
def test_gpu_bf16x6_conv_every_instantiation_is_accurate(gpu_api, tmp_path, variant):
    from oracle import oracle

    hw, rows = 17, 7
    path = W.write(str(tmp_path / "net.onnx"), _net(VARIANTS[variant], 4, hw, residual_at=()))
    # rows of wildly different magnitude: this mode has no scales, so nothing about the data's range can matter
    mags = np.array([1.0, 1e-30, 1e25, 0.0, 3e-12, 1.0, 65504.0], np.float32)
    x = (synth.table(23, 0, rows, 4 * hw * hw) * mags[:, None]).astype(np.float32)
    _load_mode(gpu_api, path, "conv_fp32", "fp32")
    gpu_api.load_model("conv_bf6", path)  # the default
    try:
        plan = gpu_api.get_plan("conv_bf6")
        assert plan["exec"].count("conv_split_bf16x6") == 1 and "bf16x6" in plan["conv_precision"]
        got = gpu_api.predict_from_blob("conv_bf6", x.tobytes())
        assert np.array_equal(got, gpu_api.predict_from_blob("conv_bf6", x.tobytes()))
        assert np.array_equal(got[5], gpu_api.predict_from_blob("conv_bf6", x[5].tobytes()).reshape(-1))
        ref32 = gpu_api.predict_from_blob("conv_fp32", x.tobytes())
    finally:
        gpu_api.unload_model("conv_bf6")
        gpu_api.unload_model("conv_fp32")
    want = oracle.Model(path).predict_blob(x.tobytes())
    assert np.all(np.isfinite(got))
    for r in range(rows):  # (or 1.5x the exact-fp32 plan's own distance: the mean over 289 pixels that follows is itself ~2e-6 off on flat rows)
        scale = np.abs(want[r]).max()
        assert np.abs(got[r] - want[r]).max() <= max(1.5e-6 * scale, 1.5 * np.abs(ref32[r] - want[r]).max()) + 1e-37, (r, np.abs(got[r] - want[r]).max() / scale)


@pytest.mark.gpu
def test_gpu_bf16x6_c5_full_width_error_against_float64(gpu_api, tmp_path):
    import torch

    from tests.test_oracle_vs_torch import torch_resnet18

    rows = 4
    path = W.write(str(tmp_path / "rn224.onnx"), W.resnet18())
    x = synth.table(5, 0, rows, 3 * 224 * 224)
    _load_mode(gpu_api, path, "conv_fp32", "fp32")
    gpu_api.load_model("conv_bf6", path)  # the default
    try:
        plan = gpu_api.get_plan("conv_bf6")
        assert plan["exec"].count("conv_split_bf16x6") == 16 and len(plan["folded_shortcuts"]) == 3 and plan["exec"][0] == "conv_patch_pool_bf16x6"
        y6 = gpu_api.predict_from_blob("conv_bf6", x.tobytes())
        y32 = gpu_api.predict_from_blob("conv_fp32", x.tobytes())
        assert np.array_equal(y6[2], gpu_api.predict_from_blob("conv_bf6", x[2].tobytes()).reshape(-1))
    finally:
        gpu_api.unload_model("conv_bf6")
        gpu_api.unload_model("conv_fp32")
    with torch.no_grad():
        ref = torch_resnet18(torch.from_numpy(x.reshape(rows, 3, 224, 224)).double(), 1000, 64, torch.float64).numpy()
    scale = np.abs(ref).max()
    e6, e32 = np.abs(y6 - ref), np.abs(y32 - ref)
    print(f"C5 vs float64: bf16x6 max {e6.max() / scale:.3e} of scale (worst |err|/(1e-4|y|+1e-6) = {(e6 / (1e-4 * np.abs(ref) + 1e-6)).max():.3f}); "
          f"fp32 plan max {e32.max() / scale:.3e} ({(e32 / (1e-4 * np.abs(ref) + 1e-6)).max():.3f})")
    assert np.all(e6 <= 1e-4 * np.abs(ref) + 1e-6)
    assert e6.max() <= 2.0 * e32.max() + 1e-7 * scale, (e6.max() / scale, e32.max() / scale)


@pytest.mark.gpu
def test_gpu_bf16x6_random_geometries(gpu_api, tmp_path):
    """The same sixty random geometries on the DEFAULT plan: layers with a multiple of 64 features run conv2d_split6_kernel, the others the
    exact-fp32 tiled kernel; every one against the oracle, a row alone == the row in its batch."""
    from oracle import oracle

    split = 0
    for seed in range(60):
        blob, hw, rows, desc = _random_conv_case(seed)
        path = W.write(str(tmp_path / f"rc{seed}.onnx"), blob)
        x = synth.table(100 + seed, 0, rows, 4 * hw * hw)
        want = oracle.Model(path).predict_blob(x.tobytes())
        _load_mode(gpu_api, path, "conv_fp32", "fp32")
        gpu_api.load_model("conv_bf6", path)
        try:
            n6 = gpu_api.get_plan("conv_bf6")["exec"].count("conv_split_bf16x6")
            assert n6 == (1 if desc["M"] % 64 == 0 else 0), desc
            split += n6
            got = gpu_api.predict_from_blob("conv_bf6", x.tobytes())
            assert np.array_equal(got[rows - 1], gpu_api.predict_from_blob("conv_bf6", x[rows - 1].tobytes()).reshape(-1)), desc
            ref32 = gpu_api.predict_from_blob("conv_fp32", x.tobytes())
        finally:
            gpu_api.unload_model("conv_bf6")
            gpu_api.unload_model("conv_fp32")
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= max(1.5e-6 * scale, 1.5 * np.abs(ref32 - want).max()) + 1e-30, (desc, np.abs(got - want).max() / scale)
    assert split >= 20


@pytest.mark.gpu
def test_gpu_bf16x6_stem_alone_and_ranges(gpu_api, tmp_path):
    """The default plan's stem + max-pool kernel by itself (7x7/2 stem, MaxPool 3x3/2, global average): images of wildly different magnitude and an
    all-zero image in one batch (no scales in this arithmetic: nothing about the range can matter), image sizes that leave ragged tiles on both
    axes, a row alone == the row in its batch, and INFERA_STEM_SPLIT=0 (the exact-fp32 stem under the same plan) as the A/B."""
    from oracle import oracle

    rng = np.random.default_rng(3)
    for hw in (64, 50, 37, 224):
        w = (rng.standard_normal((64, 3, 7, 7)) / np.sqrt(147)).astype(np.float32)
        b = (rng.standard_normal(64) * 0.1).astype(np.float32)
        nodes = [W.node("Conv", ["X", "w", "b"], ["c"], [W.attr_ints("kernel_shape", [7, 7]), W.attr_ints("strides", [2, 2]), W.attr_ints("pads", [3] * 4)]),
                 W.node("Relu", ["c"], ["r"]),
                 W.node("MaxPool", ["r"], ["p"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("strides", [2, 2]), W.attr_ints("pads", [1] * 4)]),
                 W.node("GlobalAveragePool", ["p"], ["g"]), W.node("Flatten", ["g"], ["Y"], [W.attr_i("axis", 1)])]
        path = W.write(str(tmp_path / f"stem{hw}.onnx"), W.model("stem", nodes, [W.tensor("w", w), W.tensor("b", b)], [W.value_info("X", ["N", 3, hw, hw])],
                                                           [W.value_info("Y", ["N", 64])]))
        mags = np.array([1.0, 1e-30, 1e25, 0.0, 255.0, 1.0], np.float32)
        x = (synth.table(9, 0, len(mags), 3 * hw * hw) * mags[:, None]).astype(np.float32)
        _load_mode(gpu_api, path, "conv_fp32", "fp32")
        gpu_api.load_model("conv_bf6", path)
        try:
            assert gpu_api.get_plan("conv_bf6")["exec"][0] == "conv_patch_pool_bf16x6" and gpu_api.get_plan("conv_fp32")["exec"][0] == "conv_patch_pool"
            got = gpu_api.predict_from_blob("conv_bf6", x.tobytes())
            assert np.array_equal(got, gpu_api.predict_from_blob("conv_bf6", x.tobytes()))
            assert np.array_equal(got[5], gpu_api.predict_from_blob("conv_bf6", x[5].tobytes()).reshape(-1))
            ref32 = gpu_api.predict_from_blob("conv_fp32", x.tobytes())
            os.environ["INFERA_STEM_SPLIT"] = "0"
            try:
                assert np.array_equal(ref32, gpu_api.predict_from_blob("conv_bf6", x.tobytes()))  # (nothing else in this net: the fp32 stem == the fp32 plan)
            finally:
                os.environ.pop("INFERA_STEM_SPLIT", None)
        finally:
            gpu_api.unload_model("conv_bf6")
            gpu_api.unload_model("conv_fp32")
        want = oracle.Model(path).predict_blob(x.tobytes())
        assert np.all(np.isfinite(got))
        for r in range(len(mags)):
            scale = np.abs(want[r]).max()
            assert np.abs(got[r] - want[r]).max() <= max(1.5e-6 * scale, 1.5 * np.abs(ref32[r] - want[r]).max()) + 1e-37, (hw, r, np.abs(got[r] - want[r]).max() / scale)


@pytest.mark.gpu
def test_gpu_long_conv_passes_run_as_two_lanes_bit_identical_to_short_calls(gpu_api, tmp_path):
    """hip/exec.cpp exec_plan: a pass of >= 512 rows of a convolutional plan runs as two lanes (two halves, two streams, two halves of the
    scratch).  Same kernels, same per-row arithmetic: the long call must equal the same rows served 100 at a time (single lane), bit for
    bit, for an odd row count, and twice in a row (the lanes share nothing but the weights)."""
    hw, rows = 9, 1037
    path = W.write(str(tmp_path / "net.onnx"), _net([(64, 3, 1), (64, 3, 1), (128, 3, 2), (128, 3, 1)], 4, hw, residual_at=(1, 3)))
    x = synth.table(31, 0, rows, 4 * hw * hw).astype(np.float32)
    gpu_api.load_model("lanes", path)
    try:
        plan = gpu_api.get_plan("lanes")
        assert plan["exec"].count("conv_split_bf16x6") >= 3, plan["exec"]
        long_call = gpu_api.predict_from_blob("lanes", x.tobytes())
        again = gpu_api.predict_from_blob("lanes", x.tobytes())
        short = np.concatenate([gpu_api.predict_from_blob("lanes", x[r0:r0 + 100].tobytes()).reshape(-1, long_call.shape[1]) for r0 in range(0, rows, 100)])
    finally:
        gpu_api.unload_model("lanes")
    assert long_call.shape == (rows, 128) and np.all(np.isfinite(long_call))
    assert np.array_equal(long_call, again)
    assert np.array_equal(long_call, short)
