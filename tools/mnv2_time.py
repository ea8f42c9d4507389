import sys, os, tempfile, time, json, collections
sys.path.insert(0, os.getcwd())
import numpy as np
from infera_amd import capi, onnx_writer as W
d = tempfile.mkdtemp()
p = W.write(d + "/m.onnx", W.mobilenet_v2(classes=1000, in_hw=224, width_mult=1.0))
capi.load_model("m", p)
plan = capi.get_plan("m")
print(plan["activation_layout"], collections.Counter(plan["exec"]), "GFLOP/img", plan["plan"]["flops_per_row"] / 1e9)
rows, cols = 256, 3 * 224 * 224
dev = capi.device_ordinal(0)
d_in, d_out = capi.DeviceBuffer(dev, rows * cols * 4), capi.DeviceBuffer(dev, rows * 1000 * 4)
capi.synth_fill(d_in, 42, 0, rows, cols)
capi.predict_device("m", d_in, rows, cols, d_out)
ms = capi.time_predict_device("m", d_in, rows, cols, d_out, 5) / 5
print(f"{ms:.2f} ms per {rows} images = {rows / ms * 1e3:.0f} img/s, {plan['plan']['flops_per_row'] * rows / ms / 1e9:.1f} TFLOP/s")
