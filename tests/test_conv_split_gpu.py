"""INFERA_PRECISION=f16x3 (csrc/hip/conv_split.hip): the tiled convolutions on the fp16 matrix cores, every fp32 operand split
in two fp16 halves after an exact power-of-two scaling (per output feature for the weights, per IMAGE for the activations --
the producing kernel's epilogue tracks each image's largest |x|), three MFMAs per product, fp32 accumulation.  The mode is
read when a model is scheduled, so one process loads the same network both ways.

Checked here: the plan says which steps run split; the results are within the parity tolerance (1e-4 |y| + 1e-6) of the oracle
AND within a few fp32 roundings of it relative to each row's scale; a row's result does not depend on its batch (bit for bit);
rows of wildly different magnitude (1e-20 ... 1e+20, all zeros) in one batch each keep their own relative accuracy."""
import os

import numpy as np
import pytest

from infera_amd import synth
from infera_amd import onnx_writer as W
from tests.test_conv_ws_gpu import CASES, _net


def _load_both(gpu_api, path):
    for name, precision in (("conv_fp32", "fp32"), ("conv_split", "f16x3")):  # (read when a model is scheduled; the default is bf16x6)
        os.environ["INFERA_PRECISION"] = precision
        try:
            gpu_api.load_model(name, path)
        finally:
            os.environ.pop("INFERA_PRECISION", None)


def _unload(gpu_api):
    for name in ("conv_fp32", "conv_split"):
        try:
            gpu_api.unload_model(name)
        except Exception:
            pass


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_gpu_split_fp16_conv_matches_oracle(gpu_api, tmp_path, case):
    from oracle import oracle

    c = CASES[case]
    path = W.write(str(tmp_path / "net.onnx"), _net(c["chain"], c["cin"], c["hw"], c["residual_at"]))
    x = synth.table(31, 0, c["rows"], c["cin"] * c["hw"] * c["hw"])
    _load_both(gpu_api, path)
    try:
        plan = gpu_api.get_plan("conv_split")
        assert "conv_split_f16x3" in plan["exec"] and "f16x3" in plan["conv_precision"]
        assert "conv_split_f16x3" not in gpu_api.get_plan("conv_fp32")["exec"] and "conv_precision" not in gpu_api.get_plan("conv_fp32")
        got = gpu_api.predict_from_blob("conv_split", x.tobytes())
        assert np.array_equal(got, gpu_api.predict_from_blob("conv_split", x.tobytes()))
        # the weight-stationary persistent form (forced on for these small inputs) and the tiled form: same fragments, same order of
        # additions per output element -> bit-identical
        for mode in ("0", "2"):
            os.environ["INFERA_CONV_WS"] = mode
            try:
                assert np.array_equal(got, gpu_api.predict_from_blob("conv_split", x.tobytes())), mode
            finally:
                os.environ.pop("INFERA_CONV_WS", None)
        ref32 = gpu_api.predict_from_blob("conv_fp32", x.tobytes())
        # one row alone == the same row inside the batch (the activation scale is per image)
        for r in (0, c["rows"] - 1):
            alone = gpu_api.predict_from_blob("conv_split", x[r].tobytes())
            assert np.array_equal(alone.reshape(-1), got.reshape(c["rows"], -1)[r])
    finally:
        _unload(gpu_api)
    want = oracle.Model(path).predict_blob(x.tobytes())
    assert got.shape == want.shape
    err = np.abs(got - want)
    assert np.all(err <= 1e-4 * np.abs(want) + 1e-6), err.max()
    # ... and far inside it: a few fp32 roundings of the output scale, like the exact-fp32 kernels' own distance from the oracle
    scale = np.abs(want).max()
    # (or the exact-fp32 kernels' own distance where that is larger: the mean over 1600 pixels of `many_tiles` alone is 1.6e-6 off)
    assert err.max() <= max(1.5e-6 * scale, 1.5 * np.abs(ref32 - want).max()), (err.max() / scale, np.abs(ref32 - want).max() / scale)


# every instantiation of both kernels: (feature tiles per workgroup MT) x (32-channel chunks per stage S) x (3x3 | 1x1, i.e. many stages | a
# SINGLE stage per tile for the weight-stationary form's cursor), image sizes that leave ragged last tiles.  The bound is 1.5e-6 of the output
# scale: one k-block's lo fragment read stale (an MFMA issued inside the wait states of an inline-asm VALU write, which the compiler's hazard
# recogniser does not see) showed up as 6e-6 ... 3e-5 here and nowhere else.
VARIANTS = {
    "mt1_s1_3x3": [(32, 3, 1), (32, 3, 1)], "mt1_s2_3x3": [(64, 3, 1), (32, 3, 1)], "mt2_s1_3x3": [(32, 3, 1), (64, 3, 1)],
    "mt2_s1_3blocks": [(96, 3, 1), (64, 3, 1)], "mt2_s2_3x3": [(64, 3, 1), (64, 3, 1)], "mt3_s1_3x3": [(32, 3, 1), (96, 3, 1)],
    "mt3_s2_3x3": [(64, 3, 1), (96, 3, 1)], "mt4_s1_3x3": [(32, 3, 1), (128, 3, 1)], "mt4_s2_3x3": [(64, 3, 1), (128, 3, 2)],
    "mt2_s1_1x1": [(32, 3, 1), (64, 1, 1)], "mt4_s2_1x1_s2": [(64, 3, 1), (128, 1, 2)], "mt2_s2_1x1": [(64, 3, 1), (64, 1, 1)],
    "mt1_s1_1x1": [(32, 3, 1), (32, 1, 1)], "mt4_s2_2stages": [(128, 3, 1), (128, 1, 1)],
}


@pytest.mark.gpu
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_gpu_split_fp16_conv_every_instantiation_is_accurate(gpu_api, tmp_path, variant):
    from oracle import oracle

    hw, rows = 17, 7
    path = W.write(str(tmp_path / "net.onnx"), _net(VARIANTS[variant], 4, hw, residual_at=()))
    x = synth.table(23, 0, rows, 4 * hw * hw)
    _load_both(gpu_api, path)
    try:
        assert gpu_api.get_plan("conv_split")["exec"].count("conv_split_f16x3") == 1
        out = {}
        for mode in ("0", "2"):
            os.environ["INFERA_CONV_WS"] = mode
            try:
                out[mode] = gpu_api.predict_from_blob("conv_split", x.tobytes())
            finally:
                os.environ.pop("INFERA_CONV_WS", None)
    finally:
        _unload(gpu_api)
    assert np.array_equal(out["0"], out["2"])
    want = oracle.Model(path).predict_blob(x.tobytes())
    scale = np.abs(want).max()
    assert np.abs(out["0"] - want).max() <= 1.5e-6 * scale, np.abs(out["0"] - want).max() / scale


@pytest.mark.gpu
def test_gpu_split_fp16_conv_scales_each_image_on_its_own(gpu_api, tmp_path):
    from oracle import oracle

    chain = [(64, 3, 1), (64, 3, 1), (128, 3, 2), (128, 1, 1)]
    cin, hw = 4, 14
    path = W.write(str(tmp_path / "net.onnx"), _net(chain, cin, hw, residual_at=(1,)))
    per_row = cin * hw * hw
    mags = np.array([1.0, 1e-20, 1e20, 0.0, 3e-7, 6.5e4, 1e-30, 1.0], np.float32)
    base = synth.table(5, 0, len(mags), per_row).reshape(len(mags), per_row)
    x = (base * mags[:, None]).astype(np.float32)
    _load_both(gpu_api, path)
    try:
        got = gpu_api.predict_from_blob("conv_split", x.tobytes()).reshape(len(mags), -1)
    finally:
        _unload(gpu_api)
    want = oracle.Model(path).predict_blob(x.tobytes()).reshape(len(mags), -1)
    assert np.all(np.isfinite(got))
    for r in range(len(mags)):
        scale = np.abs(want[r]).max()
        assert np.abs(got[r] - want[r]).max() <= 2e-6 * scale + 1e-37, (r, mags[r], np.abs(got[r] - want[r]).max(), scale)


@pytest.mark.gpu
def test_gpu_split_fp16_resnet18_full_width(gpu_api, tmp_path):
    """The full-width ResNet-18 topology (stem 7x7/2 + MaxPool in one kernel, 64 .. 512 channels, stride-2 entries, 1x1 downsamples, residual
    adds) in split-fp16 mode: the stem runs conv2d_stem_split_kernel (patch split once per tile, scaled by the tile's own maximum), every
    other convolution the split tiled kernels.  The first of those scales its input by the per-image maxima of the POOLED stem output, which
    the stem kernel tracks over its own stores.  With INFERA_STEM_SPLIT=0 the exact-fp32 stem kernels run under the same plan -- the
    two-workgroup one tracks the maxima like the split one, behind the one-workgroup one a reduction kernel computes them: a maximum is a
    maximum, those two must agree bit for bit.  Every route sits as close to the oracle as the exact-fp32 plan; a row alone == the row in a
    batch (the split stem runs for every batch size)."""
    from oracle import oracle

    path = W.write(str(tmp_path / "rn64.onnx"), W.resnet18(classes=10, in_hw=64, width=64))
    imgs = synth.table(21, 0, 5, 3 * 64 * 64)
    _load_both(gpu_api, path)
    try:
        plan = gpu_api.get_plan("conv_split")
        assert plan["exec"].count("conv_split_f16x3") == 19 and plan["exec"][0] == "conv_patch_pool_f16x3"
        got = gpu_api.predict_from_blob("conv_split", imgs.tobytes())
        one = gpu_api.predict_from_blob("conv_split", imgs[3].tobytes())
        out = {}
        os.environ["INFERA_STEM_SPLIT"] = "0"
        try:
            for mode in ("0", "2"):
                os.environ["INFERA_STEM_POOL2"] = mode
                out[mode] = gpu_api.predict_from_blob("conv_split", imgs.tobytes())
        finally:
            os.environ.pop("INFERA_STEM_SPLIT", None)
            os.environ.pop("INFERA_STEM_POOL2", None)
        ref32 = gpu_api.predict_from_blob("conv_fp32", imgs.tobytes())
    finally:
        _unload(gpu_api)
    assert np.array_equal(out["0"], out["2"])
    assert np.array_equal(one.reshape(-1), got[3])
    want = oracle.Model(path).predict_blob(imgs.tobytes())
    scale = np.abs(want).max()
    e32 = np.abs(ref32 - want).max() / scale
    for y in (got, out["2"]):
        assert np.all(np.abs(y - want) <= 1e-4 * np.abs(want) + 1e-6)
        assert np.abs(y - want).max() / scale <= max(3e-6, 3 * e32), (np.abs(y - want).max() / scale, e32)


@pytest.mark.gpu
def test_gpu_split_fp16_stem_alone_and_ranges(gpu_api, tmp_path):
    """The split stem + max-pool kernel by itself (7x7/2 stem, MaxPool 3x3/2, global average, nothing else): images of wildly different
    magnitude and an all-zero image in one batch, image sizes that leave ragged tiles on both axes."""
    from oracle import oracle

    rng = np.random.default_rng(3)
    for hw in (64, 50, 37):
        w = (rng.standard_normal((64, 3, 7, 7)) / np.sqrt(147)).astype(np.float32)
        b = (rng.standard_normal(64) * 0.1).astype(np.float32)
        nodes = [W.node("Conv", ["X", "w", "b"], ["c"], [W.attr_ints("kernel_shape", [7, 7]), W.attr_ints("strides", [2, 2]), W.attr_ints("pads", [3] * 4)]),
                 W.node("Relu", ["c"], ["r"]),
                 W.node("MaxPool", ["r"], ["p"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("strides", [2, 2]), W.attr_ints("pads", [1] * 4)]),
                 W.node("GlobalAveragePool", ["p"], ["g"]), W.node("Flatten", ["g"], ["Y"], [W.attr_i("axis", 1)])]
        path = W.write(str(tmp_path / f"stem{hw}.onnx"), W.model("stem", nodes, [W.tensor("w", w), W.tensor("b", b)], [W.value_info("X", ["N", 3, hw, hw])],
                                                           [W.value_info("Y", ["N", 64])]))
        mags = np.array([1.0, 1e-15, 1e15, 0.0, 255.0, 1.0], np.float32)
        x = (synth.table(9, 0, len(mags), 3 * hw * hw) * mags[:, None]).astype(np.float32)
        _load_both(gpu_api, path)
        try:
            assert gpu_api.get_plan("conv_split")["exec"][0] == "conv_patch_pool_f16x3"
            got = gpu_api.predict_from_blob("conv_split", x.tobytes())
            assert np.array_equal(got[5], gpu_api.predict_from_blob("conv_split", x[5].tobytes()).reshape(-1))
            ref32 = gpu_api.predict_from_blob("conv_fp32", x.tobytes())
            os.environ["INFERA_STEM_SPLIT"] = "0"  # the exact-fp32 stem under the same plan: a different kernel, so different bits somewhere
            try:
                assert not np.array_equal(got, gpu_api.predict_from_blob("conv_split", x.tobytes()))
            finally:
                os.environ.pop("INFERA_STEM_SPLIT", None)
        finally:
            _unload(gpu_api)
        want = oracle.Model(path).predict_blob(x.tobytes())
        assert np.all(np.isfinite(got))
        for r in range(len(mags)):  # (or 1.5x the exact-fp32 plan's own distance: the mean over 256 pooled pixels alone is 2e-6 off)
            scale = np.abs(want[r]).max()
            assert np.abs(got[r] - want[r]).max() <= max(2e-6 * scale, 1.5 * np.abs(ref32[r] - want[r]).max()) + 1e-37, (hw, r, np.abs(got[r] - want[r]).max() / scale)



@pytest.mark.gpu
def test_gpu_split_fp16_c5_full_width_error_against_float64(gpu_api, tmp_path):
    """BASELINE config C5 itself (ResNet-18, full width, 224x224, 1000 classes) in split-fp16 mode against a float64 evaluation of the same
    graph (PyTorch's CPU operators in double precision, rebuilt from the writer's weight stream -- tests/test_oracle_vs_torch.py): every logit
    within the parity tolerance 1e-4 |y| + 1e-6 of the float64 value, and the largest error no more than twice the exact-fp32 plan's own."""
    import torch

    from tests.test_oracle_vs_torch import torch_resnet18

    rows = 4
    path = W.write(str(tmp_path / "rn224.onnx"), W.resnet18())
    x = synth.table(5, 0, rows, 3 * 224 * 224)
    _load_both(gpu_api, path)
    try:
        assert gpu_api.get_plan("conv_split")["exec"].count("conv_split_f16x3") == 19
        y16 = gpu_api.predict_from_blob("conv_split", x.tobytes())
        y32 = gpu_api.predict_from_blob("conv_fp32", x.tobytes())
    finally:
        _unload(gpu_api)
    with torch.no_grad():
        ref = torch_resnet18(torch.from_numpy(x.reshape(rows, 3, 224, 224)).double(), 1000, 64, torch.float64).numpy()
    scale = np.abs(ref).max()
    e16, e32 = np.abs(y16 - ref), np.abs(y32 - ref)
    print(f"C5 vs float64: f16x3 max {e16.max() / scale:.3e} of scale (worst |err|/(1e-4|y|+1e-6) = {(e16 / (1e-4 * np.abs(ref) + 1e-6)).max():.3f}); "
          f"fp32 plan max {e32.max() / scale:.3e} ({(e32 / (1e-4 * np.abs(ref) + 1e-6)).max():.3f})")
    assert np.all(e16 <= 1e-4 * np.abs(ref) + 1e-6) and np.all(e32 <= 1e-4 * np.abs(ref) + 1e-6)
    assert e16.max() <= 2.0 * e32.max() + 1e-7 * scale, (e16.max() / scale, e32.max() / scale)


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


@pytest.mark.gpu
def test_gpu_split_fp16_random_geometries(gpu_api, tmp_path):
    """Sixty random geometries of one split-eligible convolution (channels 32 .. 160, features 32 .. 256, 1 .. 5 taps per axis, strides 1 / 2,
    dilations 1 / 2, asymmetric pads, 5 .. 24 pixel images, 1 .. 9 rows) in both forms against the oracle."""
    from oracle import oracle

    for seed in range(60):
        blob, hw, rows, desc = _random_conv_case(seed)
        path = W.write(str(tmp_path / f"rc{seed}.onnx"), blob)
        x = synth.table(100 + seed, 0, rows, 4 * hw * hw)
        try:
            want = oracle.Model(path).predict_blob(x.tobytes())
        except Exception:  # (a geometry with an empty output: nothing to compare)
            continue
        _load_both(gpu_api, path)
        try:
            assert gpu_api.get_plan("conv_split")["exec"].count("conv_split_f16x3") == 1, desc
            out = {}
            for mode in ("0", "2"):
                os.environ["INFERA_CONV_WS"] = mode
                try:
                    out[mode] = gpu_api.predict_from_blob("conv_split", x.tobytes())
                finally:
                    os.environ.pop("INFERA_CONV_WS", None)
            ref32 = gpu_api.predict_from_blob("conv_fp32", x.tobytes())
        finally:
            _unload(gpu_api)
        assert np.array_equal(out["0"], out["2"]), desc
        scale = np.abs(want).max()
        assert np.abs(out["0"] - want).max() <= max(1.5e-6 * scale, 1.5 * np.abs(ref32 - want).max()) + 1e-30, (desc, np.abs(out["0"] - want).max() / scale)


@pytest.mark.gpu
def test_gpu_split_fp16_non_finite_images_stay_in_their_rows(gpu_api, tmp_path):
    """A NaN or an infinity in one image: every OTHER image of the batch is bit for bit what it is without the poisoned neighbour -- the
    scales are per image (per tile in the stem), nothing of one row reaches another.  (What the poisoned image itself returns is not
    specified in this mode: an infinity takes the scale of its tile / image to the floor, and Relu maps NaN to 0 here as in the exact-fp32
    kernels and the oracle -- DESIGN.md 3.3b.)"""
    path = W.write(str(tmp_path / "rn64.onnx"), W.resnet18(classes=10, in_hw=64, width=64))
    clean = synth.table(21, 0, 6, 3 * 64 * 64)
    bad = clean.copy()
    bad[1, 5000] = np.nan
    bad[4, 77] = np.inf
    _load_both(gpu_api, path)
    try:
        y_clean = gpu_api.predict_from_blob("conv_split", clean.tobytes())
        y_bad = gpu_api.predict_from_blob("conv_split", bad.tobytes())
    finally:
        _unload(gpu_api)
    assert np.all(np.isfinite(y_clean))
    for r in (0, 2, 3, 5):
        assert np.array_equal(y_bad[r], y_clean[r]), r
    assert not np.array_equal(y_bad[4], y_clean[4])


# ---- the DEFAULT convolution form (bf16x6): operands cut exactly into three bf16 parts, six partial products; no scales, no maxima, no precondition ----
def _load_mode(gpu_api, path, name, precision):
    os.environ["INFERA_PRECISION"] = precision
    try:
        gpu_api.load_model(name, path)
    finally:
        os.environ.pop("INFERA_PRECISION", None)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["mt2_s1_3x3", "mt2_s1_3blocks", "mt2_s2_3x3", "mt4_s1_3x3", "mt4_s2_3x3", "mt2_s1_1x1", "mt4_s2_1x1_s2", "mt2_s2_1x1",
                                     "mt4_s2_2stages"])
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
        assert plan["exec"].count("conv_split_bf16x6") == 19 and plan["exec"][0] == "conv_patch_pool_bf16x6"
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
