"""A channel slice that does not start on a 4-channel boundary cannot be a flat copy of a channel-quad tensor
(ADVICE round 1): Conv(8ch) -> Slice[:, 2:6] -> Conv must either keep the plan in NCHW or slice layout-aware; the
quad-aligned twin Slice[:, 4:8] keeps the fast layout.  CPU: the layout decision.  GPU: HIP vs oracle."""
import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth


def slice_net(lo, hi, hw=8):
    ws = W._WeightStream(99)
    w1, b1 = ws.take((8, 4, 3, 3), 36), ws.take((8,), 36)
    c = hi - lo
    w2, b2 = ws.take((4, c, 3, 3), 9 * c), ws.take((4,), 9 * c)
    fc, fb = ws.take((4, 3), 4), ws.take((3,), 4)
    nodes = [
        W.node("Conv", ["X", "w1", "b1"], ["a"], [W.attr_ints("pads", [1, 1, 1, 1])]),
        W.node("Relu", ["a"], ["ar"]),
        W.node("Slice", ["ar", "s", "e", "ax"], ["cut"]),
        W.node("Conv", ["cut", "w2", "b2"], ["b"], [W.attr_ints("pads", [1, 1, 1, 1])]),
        W.node("GlobalAveragePool", ["b"], ["g"]),
        W.node("Flatten", ["g"], ["f"]),
        W.node("Gemm", ["f", "fc", "fb"], ["Y"]),
    ]
    inits = [W.tensor("w1", w1), W.tensor("b1", b1), W.tensor("w2", w2), W.tensor("b2", b2), W.tensor("fc", fc), W.tensor("fb", fb),
             W.tensor("s", np.array([lo], np.int64)), W.tensor("e", np.array([hi], np.int64)), W.tensor("ax", np.array([1], np.int64))]
    return W.model("slice_net", nodes, inits, [W.value_info("X", ["N", 4, hw, hw])], [W.value_info("Y", ["N", 3])])


@pytest.mark.parametrize("lo,hi,want", [(2, 6, "NCHW"), (4, 8, "NC/4HW4"), (0, 4, "NC/4HW4")])
def test_layout_decision(built, tmp_path, lo, hi, want):
    from infera_amd import capi

    p = W.write(str(tmp_path / "s.onnx"), slice_net(lo, hi))
    capi.load_model("slice_net", p)
    try:
        assert capi.get_plan("slice_net")["activation_layout"] == want
    finally:
        capi.unload_model("slice_net")


@pytest.mark.gpu
@pytest.mark.parametrize("lo,hi", [(2, 6), (4, 8), (0, 4), (1, 5)])
def test_gpu_channel_slice_parity(gpu_api, built, tmp_path, lo, hi):
    from oracle import oracle

    p = W.write(str(tmp_path / "s.onnx"), slice_net(lo, hi))
    x = synth.table(5, 0, 37, 4 * 8 * 8)
    want = oracle.Model(p).predict_blob(x.tobytes())
    gpu_api.load_model("slice_net", p)
    try:
        got = gpu_api.predict_from_blob("slice_net", x.tobytes())
    finally:
        gpu_api.unload_model("slice_net")
    assert got.shape == want.shape
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-6), np.abs(got - want).max()
