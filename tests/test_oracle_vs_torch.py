"""The oracle against PyTorch's CPU operators -- an implementation of Conv / BatchNorm / MaxPool / GlobalAveragePool / Linear /
Softmax that is independent of both this repository and the ONNX text the oracle was written from.  The reference's own
tests pin only MatMul+Add and Identity; for everything C2 / C4 / C5 execute, the authority is the ONNX operator
specification -- this test adds a second, widely used fp32 and fp64 evaluation of the same graphs (it does not make the pin a
reference pin: Tract is still not buildable here).  The torch graph is rebuilt from the onnx_writer weight stream."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from infera_amd import onnx_writer as W
from infera_amd import synth


def torch_resnet18(x, classes, width, dtype):
    """onnx_writer.resnet18 rebuilt from the same weight stream (oracle/torch_ref.py), BatchNorm as its own operator."""
    from oracle import torch_ref

    return torch_ref.resnet18_forward(torch_ref.resnet18_params(classes, width, dtype), x)


@pytest.mark.parametrize("hw,width", [(64, 16), (40, 8)])
def test_oracle_resnet18_topology_vs_torch(built, tmp_path, hw, width):
    from oracle import oracle

    classes, rows = 10, 3
    path = W.write(str(tmp_path / "rn.onnx"), W.resnet18(classes=classes, in_hw=hw, width=width))
    x = synth.table(21, 0, rows, 3 * hw * hw)
    got = oracle.Model(path).predict_blob(x.tobytes())
    xt = torch.from_numpy(x.reshape(rows, 3, hw, hw))
    with torch.no_grad():
        ref64 = torch_resnet18(xt.double(), classes, width, torch.float64).numpy()
        ref32 = torch_resnet18(xt, classes, width, torch.float32).numpy()
    scale = np.abs(ref64).max()
    assert got.shape == ref64.shape == (rows, classes)
    assert np.abs(got - ref64).max() <= 2e-5 * scale + 1e-6, np.abs(got - ref64).max()   # oracle (fp32, k-ordered fmaf) vs fp64 truth
    assert np.abs(ref32 - ref64).max() <= 2e-5 * scale + 1e-6                            # torch's own fp32 is as far from it
    assert np.abs(got - ref32).max() <= 4e-5 * scale + 2e-6


def test_oracle_mlp_and_softmax_vs_torch(built, tmp_path):
    from oracle import oracle

    x = synth.table(42, 0, 777, 128)
    ws = W._WeightStream(1234)
    layers = [(ws.take((k, m), k), ws.take((m,), k)) for k, m in zip((128, 256, 64), (256, 64, 1))]
    h = torch.from_numpy(x).double()
    for i, (w, b) in enumerate(layers):
        h = F.linear(h, torch.from_numpy(w.T.copy()).double(), torch.from_numpy(b).double())
        if i < 2:
            h = F.relu(h)
    got = oracle.Model(W.write(str(tmp_path / "mlp.onnx"), W.mlp((128, 256, 64, 1)))).predict(x)
    np.testing.assert_allclose(got, h.numpy(), rtol=2e-5, atol=2e-6)
    ws = W._WeightStream(1234)
    w, b = ws.take((128, 10), 128), ws.take((10,), 128)
    want = F.softmax(F.linear(torch.from_numpy(x).double(), torch.from_numpy(w.T.copy()).double(), torch.from_numpy(b).double()), dim=1).numpy()
    got = oracle.Model(W.write(str(tmp_path / "lr.onnx"), W.logreg_softmax(128, 10))).predict(x)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-7)
