import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infera_amd import capi, synth
from infera_amd import onnx_writer as W
from oracle import oracle
d = tempfile.mkdtemp()
path = W.write(os.path.join(d, "rn64.onnx"), W.resnet18(classes=10, in_hw=64, width=64))
imgs = synth.table(21, 0, 5, 3 * 64 * 64)
want = oracle.Model(path).predict_blob(imgs.tobytes())
sc = np.abs(want).max()
capi.load_model("f32", path)
print("fp32", np.abs(capi.predict_from_blob("f32", imgs.tobytes()) - want).max() / sc)
os.environ["INFERA_PRECISION"] = "f16x3"
capi.load_model("s", path)
for mode in ("0", "2", "0", "2"):
    os.environ["INFERA_STEM_POOL2"] = mode
    y = capi.predict_from_blob("s", imgs.tobytes())
    print("stem_pool2 =", mode, np.abs(y - want).max() / sc, np.isfinite(y).all())
