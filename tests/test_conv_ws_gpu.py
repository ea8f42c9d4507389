"""The weight-stationary persistent convolution kernel (conv2d_ws_kernel, csrc/hip/conv.hip): forced on for small inputs
(INFERA_CONV_WS=2) it must produce bit-for-bit what the tiled kernel produces (same packed weights, same summation order)
and match the oracle: ResNet-style 64-channel 3x3 layers with residual adds, a stride-2 entry in two 64-feature slices,
1x1 downsamples, odd and even stage counts, ragged last tiles, more tiles than the persistent grid has waves."""
import os

import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth


def _net(chain, cin, hw, residual_at=()):
    """chain: (cout, k, stride); residual_at: indices i whose conv output gets Add(input of conv i) before the Relu."""
    rng = np.random.default_rng(17)
    nodes, inits, x = [], [], "X"
    c = cin
    for i, (cout, k, stride) in enumerate(chain):
        w = (rng.standard_normal((cout, c, k, k)) / np.sqrt(c * k * k)).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        inits += [W.tensor(f"w{i}", w), W.tensor(f"b{i}", b)]
        nodes.append(W.node("Conv", [x, f"w{i}", f"b{i}"], [f"c{i}"], [W.attr_ints("kernel_shape", [k, k]), W.attr_ints("strides", [stride] * 2),
                                                                   W.attr_ints("pads", [k // 2] * 4)]))
        src = f"c{i}"
        if i in residual_at:
            nodes.append(W.node("Add", [src, x], [f"a{i}"]))
            src = f"a{i}"
        nodes.append(W.node("Relu", [src], [f"r{i}"]))
        x, c = f"r{i}", cout
    nodes += [W.node("GlobalAveragePool", [x], ["g"]), W.node("Flatten", ["g"], ["Y"], [W.attr_i("axis", 1)])]
    return W.model("wsnet", nodes, inits, [W.value_info("X", ["N", cin, hw, hw])], [W.value_info("Y", ["N", c])])


CASES = {
    # 4-channel stem (padded-channel tiled kernel), then: 3x3 64->64 (+residual) x2 [9 stages, MT 2], 3x3/2 64->128 [two
    # 64-feature slices], 1x1 128->64 [2 stages], 1x1/2 64->128 [1 stage, MT 4], 1x1 128->256 [two 128-feature slices]
    "resnet_like": dict(chain=[(64, 3, 1), (64, 3, 1), (64, 3, 1), (128, 3, 2), (64, 1, 1), (128, 1, 2), (256, 1, 1)], cin=4, hw=22, residual_at=(1, 2), rows=5),
    # 32-channel layers (S = 1): 3x3 32->64 [9 stages of one chunk], 1x1 64->32 -> not whole 64: MT=1 -> stays on the tiled kernel
    "narrow": dict(chain=[(32, 3, 1), (64, 3, 1), (64, 3, 2), (32, 1, 1)], cin=4, hw=17, residual_at=(), rows=3),
    # many more tiles than the persistent grid has waves (256 CUs x 8 waves): 40 x 40 x 24 = 38400 pixels = 1200 tiles ... x 3 rows
    "many_tiles": dict(chain=[(64, 3, 1), (64, 3, 1)], cin=4, hw=40, residual_at=(1,), rows=24),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_gpu_weight_stationary_conv_matches_tiled_and_oracle(gpu_api, tmp_path, case):
    from oracle import oracle

    c = CASES[case]
    path = W.write(str(tmp_path / "ws.onnx"), _net(c["chain"], c["cin"], c["hw"], c["residual_at"]))
    x = synth.table(31, 0, c["rows"], c["cin"] * c["hw"] * c["hw"])
    os.environ["INFERA_PRECISION"] = "fp32"  # (read when the model is scheduled: the exact-fp32 kernels are what this test is about)
    try:
        gpu_api.load_model("wsnet", path)
    finally:
        os.environ.pop("INFERA_PRECISION", None)
    try:
        assert gpu_api.get_plan("wsnet")["activation_layout"] == "NC/4HW4"
        out = {}
        for mode in ("0", "2"):
            os.environ["INFERA_CONV_WS"] = mode
            out[mode] = gpu_api.predict_from_blob("wsnet", x.tobytes())
            again = gpu_api.predict_from_blob("wsnet", x.tobytes())
            assert np.array_equal(out[mode], again)
    finally:
        os.environ.pop("INFERA_CONV_WS", None)
        gpu_api.unload_model("wsnet")
    assert np.array_equal(out["0"], out["2"]), np.abs(out["0"] - out["2"]).max()  # same order of additions: bit-identical
    want = oracle.Model(path).predict_blob(x.tobytes())
    assert out["2"].shape == want.shape
    assert np.all(np.abs(out["2"] - want) <= 1e-4 * np.abs(want) + 1e-6), np.abs(out["2"] - want).max()


@pytest.mark.gpu
def test_gpu_tiled_conv_tail_split_is_bit_identical(gpu_api, tmp_path):
    """Round 3: a wide tiled launch that ends in a short last round of workgroups (ResNet's 256 -> 512 stride-2 entry: 1568 workgroups on
    512 slots) runs its last pixel blocks as a second launch of 32-feature tiles (conv2d_tiled, `blk0`).  Same sums per output element:
    INFERA_CONV_TAIL_SPLIT=0 / 1 (read per launch) must agree bit for bit, and with the oracle.  Shape: 256 -> 128 channels, 3x3 stride 2,
    32x32 images -> 16x16 outputs, 261 images = 522 pixel blocks x 1 feature slice = one full round of 512 + 10 blocks of tail."""
    from oracle import oracle

    rows = 261
    path = W.write(str(tmp_path / "tail.onnx"), _net([(256, 1, 1), (128, 3, 2)], 4, 32))
    x = synth.table(33, 0, rows, 4 * 32 * 32)
    os.environ["INFERA_PRECISION"] = "fp32"
    try:
        gpu_api.load_model("tailnet", path)
    finally:
        os.environ.pop("INFERA_PRECISION", None)
    try:
        out = {}
        for mode in ("1", "0"):
            os.environ["INFERA_CONV_TAIL_SPLIT"] = mode
            out[mode] = gpu_api.predict_from_blob("tailnet", x.tobytes())
            assert np.array_equal(out[mode], gpu_api.predict_from_blob("tailnet", x.tobytes()))
    finally:
        os.environ.pop("INFERA_CONV_TAIL_SPLIT", None)
        gpu_api.unload_model("tailnet")
    assert np.array_equal(out["0"], out["1"]), float(np.abs(out["0"] - out["1"]).max())
    want = oracle.Model(path).predict_blob(x[:3].tobytes())
    assert np.all(np.abs(out["1"][:3] - want) <= 1e-4 * np.abs(want) + 1e-6)
