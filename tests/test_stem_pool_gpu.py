"""Stem convolution + MaxPool 3x3/2 in one kernel (conv2d_patch_kernel<.., POOL>, csrc/hip/conv.hip): the pooled tensor must
be bit-for-bit what the stem kernel followed by the pooling kernel produce (INFERA_STEM_POOL=0 at load time) -- the same
MFMA chains per convolution pixel, and a maximum is exact in any order -- and match the oracle.  Geometries: the ResNet stem
(7x7/2 pad 3 -> pool pad 1) on image sizes whose pooled extent is and is not a multiple of the 8 x 7 pooled tile, a 3x3/2
stem with pool pad 0 and ceil_mode (SqueezeNet), 32 and 64 stem features, 1- / 3- / 4-channel images, a 5x5 stem with stride 1,
more tiles than workgroups, and a stem whose pooled output is the only thing between it and the head."""
import os

import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth


@pytest.fixture(autouse=True)
def _exact_fp32_kernels(monkeypatch):
    """This module is about the exact-fp32 stem kernels (fused or not, one workgroup or two per CU: bit-identical).  The default plan runs the stem
    in bf16 x three parts since round 3 (tests/test_conv_split_gpu.py); INFERA_PRECISION is read when a model is scheduled."""
    monkeypatch.setenv("INFERA_PRECISION", "fp32")


def _net(cin, hw, m, k, stride, pad, pool_pad, ceil, relu=True, hw2=None):
    rng = np.random.default_rng(23)
    w = (rng.standard_normal((m, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = (rng.standard_normal(m) * 0.2).astype(np.float32)
    w2 = (rng.standard_normal((m, m, 1, 1)) / np.sqrt(m)).astype(np.float32)
    nodes = [W.node("Conv", ["X", "w", "b"], ["c"], [W.attr_ints("kernel_shape", [k, k]), W.attr_ints("strides", [stride] * 2), W.attr_ints("pads", [pad] * 4)])]
    src = "c"
    if relu:
        nodes.append(W.node("Relu", ["c"], ["r"]))
        src = "r"
    nodes += [W.node("MaxPool", [src], ["p"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("strides", [2, 2]), W.attr_ints("pads", [pool_pad] * 4),
                                              W.attr_i("ceil_mode", 1 if ceil else 0)]),
              W.node("Conv", ["p", "w2"], ["c2"], [W.attr_ints("kernel_shape", [1, 1])]),
              W.node("GlobalAveragePool", ["c2"], ["g"]), W.node("Flatten", ["g"], ["Y"], [W.attr_i("axis", 1)])]
    return W.model("stem", nodes, [W.tensor("w", w), W.tensor("b", b), W.tensor("w2", w2)],
                   [W.value_info("X", ["N", cin, hw, hw2 or hw])], [W.value_info("Y", ["N", m])])


CASES = {
    "resnet_stem_64": dict(cin=3, hw=64, m=64, k=7, stride=2, pad=3, pool_pad=1, ceil=False, rows=3),       # conv 32x32 -> pooled 16x16: 2 x 3 tiles, ragged columns
    "resnet_stem_224": dict(cin=3, hw=224, m=64, k=7, stride=2, pad=3, pool_pad=1, ceil=False, rows=2),     # the real thing: 56 x 56 = 7 x 8 whole tiles
    "odd_extent": dict(cin=3, hw=50, m=64, k=7, stride=2, pad=3, pool_pad=1, ceil=False, rows=4),           # conv 25x25 -> pooled 13x13
    "squeezenet_like": dict(cin=3, hw=59, m=64, k=3, stride=2, pad=0, pool_pad=0, ceil=True, rows=3),       # conv 29x29 -> pooled 14x14 (ceil), windows past the edge
    "m32_gray": dict(cin=1, hw=40, m=32, k=5, stride=1, pad=2, pool_pad=1, ceil=False, rows=5),             # stride-1 stem, one channel, 32 features
    "four_channels_no_relu": dict(cin=4, hw=36, m=64, k=3, stride=1, pad=1, pool_pad=1, ceil=False, rows=2, relu=False),  # negative values reach the pool
    "many_tiles": dict(cin=3, hw=96, m=64, k=7, stride=2, pad=3, pool_pad=1, ceil=False, rows=40),          # 40 x 3 x 4 = 480 tiles > 256 workgroups
    "k5_rgb": dict(cin=3, hw=48, m=64, k=5, stride=2, pad=2, pool_pad=1, ceil=False, rows=6),                # 75 taps -> 10 k groups: in-loop pooling over 40 units
    "m32_rgb7": dict(cin=3, hw=40, m=32, k=7, stride=2, pad=3, pool_pad=1, ceil=False, rows=5),             # one feature tile per wave, in-loop pooling, 448 items
    "cin2_k3": dict(cin=2, hw=30, m=64, k=3, stride=1, pad=1, pool_pad=0, ceil=False, rows=3),              # 18 taps -> run-time k loop, lock-step pooling
    "rectangular": dict(cin=3, hw=44, hw2=70, m=64, k=7, stride=2, pad=3, pool_pad=1, ceil=False, rows=3),  # H != W
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_gpu_stem_with_fused_maxpool_matches_two_kernels_and_oracle(gpu_api, tmp_path, case):
    from oracle import oracle

    c = dict(CASES[case])
    rows = c.pop("rows")
    path = W.write(str(tmp_path / "stem.onnx"), _net(**c))
    x = synth.table(37, 0, rows, c["cin"] * c["hw"] * (c.get("hw2") or c["hw"]))
    out = {}
    try:
        for mode in ("1", "0"):
            os.environ["INFERA_STEM_POOL"] = mode  # read when the model is scheduled
            gpu_api.load_model("stem", path)
            plan = gpu_api.get_plan("stem")
            assert plan["activation_layout"] == "NC/4HW4"
            assert ("conv_patch_pool" in plan["exec"]) == (mode == "1"), plan["exec"]
            assert ("conv_patch" in plan["exec"]) == (mode == "0"), plan["exec"]
            out[mode] = gpu_api.predict_from_blob("stem", x.tobytes())
            assert np.array_equal(out[mode], gpu_api.predict_from_blob("stem", x.tobytes()))
            gpu_api.unload_model("stem")
    finally:
        os.environ.pop("INFERA_STEM_POOL", None)
    assert np.array_equal(out["0"], out["1"]), np.abs(out["0"] - out["1"]).max()
    want = oracle.Model(path).predict_blob(x.tobytes())
    assert out["1"].shape == want.shape
    assert np.all(np.abs(out["1"] - want) <= 1e-4 * np.abs(want) + 1e-6), np.abs(out["1"] - want).max()


@pytest.mark.gpu
def test_gpu_random_stem_geometries_fused_equals_two_kernels(gpu_api, tmp_path):
    """40 seeded random stems (1-4 input channels, 3x3 / 5x5 / 7x7, stride 1-2, any padding up to k//2, 32 or 64 features, H != W, pool
    pad 0 / 1, ceil_mode on / off, with and without ReLU): fused kernel == stem kernel + pooling kernel, bit for bit, and both match
    the oracle."""
    from oracle import oracle

    rng = np.random.default_rng(2024)
    n_fused = 0
    for case in range(40):
        k = int(rng.choice([3, 5, 7]))
        c = dict(cin=int(rng.integers(1, 5)), hw=int(rng.integers(k + 6, 72)), hw2=int(rng.integers(k + 6, 72)), m=int(rng.choice([32, 64])), k=k,
                 stride=int(rng.choice([1, 2])), pad=int(rng.integers(0, k // 2 + 1)), pool_pad=int(rng.choice([0, 1])), ceil=bool(rng.random() < 0.5),
                 relu=bool(rng.random() < 0.7))
        rows = int(rng.choice([1, 2, 5]))
        path = W.write(str(tmp_path / f"s{case}.onnx"), _net(**c))
        x = synth.table(100 + case, 0, rows, c["cin"] * c["hw"] * c["hw2"])
        out = {}
        try:
            for mode in ("1", "0"):
                os.environ["INFERA_STEM_POOL"] = mode
                gpu_api.load_model("s", path)
                if mode == "1":
                    n_fused += "conv_patch_pool" in gpu_api.get_plan("s")["exec"]
                out[mode] = gpu_api.predict_from_blob("s", x.tobytes())
                gpu_api.unload_model("s")
        finally:
            os.environ.pop("INFERA_STEM_POOL", None)
        assert np.array_equal(out["0"], out["1"]), (case, c, float(np.abs(out["0"] - out["1"]).max()))
        want = oracle.Model(path).predict_blob(x.tobytes())
        assert np.all(np.abs(out["1"] - want) <= 1e-4 * np.abs(want) + 1e-6), (case, c, float(np.abs(out["1"] - want).max()))
    assert n_fused >= 30  # (a few geometries have no pooled pixel with a whole window inside the image or exceed the LDS budget)


@pytest.mark.gpu
def test_gpu_fused_stem_pool_treats_nan_and_inf_like_the_two_kernels(gpu_api, tmp_path):
    """Images with NaN, +inf and -inf pixels (no ReLU, so they reach the pooling): the fused kernel must give the same bit patterns
    as the stem kernel followed by the pooling kernel -- a window holding only NaNs pools to -inf in both, a NaN beside finite
    values is ignored by both."""
    c = dict(cin=3, hw=40, m=64, k=7, stride=2, pad=3, pool_pad=1, ceil=False, relu=False)
    path = W.write(str(tmp_path / "nan.onnx"), _net(**c))
    x = synth.table(5, 0, 3, 3 * 40 * 40).reshape(3, 3, 40, 40).copy()
    x[0, :, 10:14, 10:14] = np.nan   # a patch of NaNs: a whole neighbourhood of convolution outputs is NaN
    x[1, 1, 20, 20] = np.inf
    x[1, 2, 5, 30] = -np.inf
    x[2, 0, 0, 0] = np.nan           # a corner: windows that also hold padding
    out = {}
    try:
        for mode in ("1", "0"):
            os.environ["INFERA_STEM_POOL"] = mode
            gpu_api.load_model("nan", path)
            out[mode] = gpu_api.predict_from_blob("nan", x.astype(np.float32).tobytes())
            gpu_api.unload_model("nan")
    finally:
        os.environ.pop("INFERA_STEM_POOL", None)
    assert np.array_equal(out["0"].view(np.uint32), out["1"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["resnet_stem_64", "resnet_stem_224", "odd_extent", "many_tiles", "k5_rgb", "rectangular"])
def test_gpu_two_half_workgroup_stem_pool_kernel_is_bit_identical(gpu_api, tmp_path, case):
    """Round 3: 64-feature stems with a compile-time k loop (7x7 and 5x5 over 3 channels) run as TWO half-channel 256-thread
    workgroups per CU (conv2d_stem_pool2_kernel; INFERA_STEM_POOL2=2 forces it even for launches of a few tiles, 0 = the
    one-workgroup kernel, read per launch).  Same k order per output element, max-pooling per channel: the pooled tensor must be
    bit-for-bit that of the one-workgroup fused kernel and of the stem kernel + pooling kernel, with and without the odd half's
    start delay."""
    from oracle import oracle

    c = dict(CASES[case])
    rows = c.pop("rows")
    path = W.write(str(tmp_path / "stem.onnx"), _net(**c))
    x = synth.table(41, 0, rows, c["cin"] * c["hw"] * (c.get("hw2") or c["hw"]))
    out = {}
    try:
        for tag, env in (("two_kernels", {"INFERA_STEM_POOL": "0"}), ("one_wg", {"INFERA_STEM_POOL2": "0"}), ("two_half_wgs", {"INFERA_STEM_POOL2": "2"}),
                         ("two_half_wgs_no_delay", {"INFERA_STEM_POOL2": "2", "INFERA_STEM_POOL2_DESYNC": "0"})):
            os.environ.update(env)
            try:
                gpu_api.load_model("stem", path)
                out[tag] = gpu_api.predict_from_blob("stem", x.tobytes())
                assert np.array_equal(out[tag], gpu_api.predict_from_blob("stem", x.tobytes()))
                gpu_api.unload_model("stem")
            finally:
                for k in env:
                    os.environ.pop(k, None)
    finally:
        for k in ("INFERA_STEM_POOL", "INFERA_STEM_POOL2", "INFERA_STEM_POOL2_DESYNC"):
            os.environ.pop(k, None)
    for tag in ("one_wg", "two_half_wgs", "two_half_wgs_no_delay"):
        assert np.array_equal(out[tag], out["two_kernels"]), (tag, float(np.abs(out[tag] - out["two_kernels"]).max()))
    want = oracle.Model(path).predict_blob(x.tobytes())
    assert np.all(np.abs(out["two_half_wgs"] - want) <= 1e-4 * np.abs(want) + 1e-6)
