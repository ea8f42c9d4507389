"""Debug helper: split-fp16 conv, tiled vs weight-stationary form, per chain."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infera_amd import capi, synth
from infera_amd import onnx_writer as W
from tests.test_conv_ws_gpu import _net

CHAINS = {
    "32->64 3x3": ([(32, 3, 1), (64, 3, 1)], 17),
    "32->64 1x1": ([(32, 3, 1), (64, 1, 1)], 17),
    "32->64->64s2": ([(32, 3, 1), (64, 3, 1), (64, 3, 2)], 17),
    "64->64": ([(64, 3, 1), (64, 3, 1)], 17),
    "32->32 (MT1 S1)": ([(32, 3, 1), (32, 3, 1)], 17),
    "64->32 (MT1 S2)": ([(64, 3, 1), (32, 3, 1)], 17),
    "32->128 (MT4 S1)": ([(32, 3, 1), (128, 3, 1)], 17),
    "32->96 (MT3 S1)": ([(32, 3, 1), (96, 3, 1)], 17),
    "96->64 (MT2 S1 3 blocks)": ([(96, 3, 1), (64, 3, 1)], 17),
}
os.environ["INFERA_PRECISION"] = "f16x3"
d = tempfile.mkdtemp()
for name, (chain, hw) in CHAINS.items():
    path = W.write(os.path.join(d, "n.onnx"), _net(chain, 4, hw))
    capi.load_model("m", path)
    x = synth.table(31, 0, 3, 4 * hw * hw)
    out = {}
    for mode in ("0", "2"):
        os.environ["INFERA_CONV_WS"] = mode
        out[mode] = capi.predict_from_blob("m", x.tobytes())
    capi.unload_model("m")
    from oracle import oracle
    want = oracle.Model(path).predict_blob(x.tobytes())
    sc = np.abs(want).max()
    print("   err vs oracle / scale: tiled", np.abs(out["0"] - want).max() / sc, "ws", np.abs(out["2"] - want).max() / sc)
    diff = np.abs(out["0"] - out["2"])
    print(name, "max diff", diff.max(), "rel", diff.max() / np.abs(out["0"]).max(), "rows differing", np.nonzero(diff.max(axis=1))[0].tolist(), "nfeat", int((diff > 0).sum()))
