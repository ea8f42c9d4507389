"""VGG-16 topology (random weights, 224x224): load time, forward time and kernel mix for a 64-image batch.
usage (GPU box): python tools/vgg16_time.py [batch]"""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from infera_amd import capi, onnx_writer as W

ws = W._WeightStream(7)
nodes, inits = [], []
cur, cin, hw = "X", 3, 224
for bi, (reps, cout) in enumerate([(2, 64), (2, 128), (3, 256), (3, 512), (3, 512)]):
    for r in range(reps):
        name = f"c{bi}_{r}"
        inits += [W.tensor(name + "w", ws.take((cout, cin, 3, 3), cin * 9)), W.tensor(name + "b", ws.take((cout,), cin * 9))]
        nodes += [W.node("Conv", [cur, name + "w", name + "b"], [name], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("pads", [1, 1, 1, 1])]),
                  W.node("Relu", [name], [name + "r"])]
        cur, cin = name + "r", cout
    nodes.append(W.node("MaxPool", [cur], [f"p{bi}"], [W.attr_ints("kernel_shape", [2, 2]), W.attr_ints("strides", [2, 2])]))
    cur, hw = f"p{bi}", hw // 2
nodes.append(W.node("Flatten", [cur], ["flat"], [W.attr_i("axis", 1)]))
cur, k = "flat", 512 * 7 * 7
for i, m in enumerate([4096, 4096, 1000]):
    inits += [W.tensor(f"f{i}w", ws.take((k, m), k)), W.tensor(f"f{i}b", ws.take((m,), k))]
    nodes.append(W.node("Gemm", [cur, f"f{i}w", f"f{i}b"], [f"f{i}" if i < 2 else "Y"]))
    if i < 2:
        nodes.append(W.node("Relu", [f"f{i}"], [f"f{i}r"]))
    cur, k = f"f{i}r", m
d = tempfile.mkdtemp()
path = W.write(f"{d}/vgg16.onnx", W.model("vgg16", nodes, inits, [W.value_info("X", ["N", 3, 224, 224])], [W.value_info("Y", ["N", 1000])]))
print("model file MB", os.path.getsize(path) / 1e6)
t = time.perf_counter(); capi.load_model("vgg", path); print("load s", time.perf_counter() - t)
plan = capi.get_plan("vgg")
print(plan["activation_layout"], plan["exec"])
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = capi.device_ordinal(0)
d_in, d_out = capi.DeviceBuffer(dev, rows * 3 * 224 * 224 * 4), capi.DeviceBuffer(dev, rows * 1000 * 4)
capi.synth_fill(d_in, 42, 0, rows, 3 * 224 * 224)
capi.predict_device("vgg", d_in, rows, 3 * 224 * 224, d_out)
ms = capi.time_predict_device("vgg", d_in, rows, 3 * 224 * 224, d_out, 3) / 3
flops = 2 * 15.47e9 * rows  # 15.47 GMAC per image
y = d_out.download((2, 1000))
print(f"batch {rows}: {ms:.2f} ms  {rows / ms * 1e3:.0f} img/s  {flops / ms / 1e9:.1f} TFLOP/s  finite={np.isfinite(y).all()}")
