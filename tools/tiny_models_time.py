"""Times small tabular models (single layers, softmax heads, two-layer MLPs) on a 20M-row table.
usage (GPU box): python tools/tiny_models_time.py"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
from infera_amd import capi, onnx_writer as W
d = tempfile.mkdtemp(); dev = capi.device_ordinal(0)
CASES = [((3, 1), False), ((13, 1), False), ((30, 1), False), ((4, 3), True), ((30, 2), True), ((30, 3), True), ((30, 8), False),
         ((100, 10), True), ((128, 10), True), ((64, 1), False), ((20, 16, 1), False), ((30, 8, 1), False), ((50, 1), False),
         ((100, 1), False), ((77, 5), True), ((48, 10), True), ((96, 4), False), ((13, 3), True), ((120, 16), False),
         ((4, 10, 3), True), ((30, 100, 2), True), ((13, 64, 32, 1), False), ((30, 100), False), ((100, 100, 100, 10), True),
         ((128, 128, 128, 16), False),
         ((30, 100, 100, 2), True), ((64, 100, 10), True), ((100, 50, 1), False), ((20, 64, 64, 64, 1), False)]
for dims, sm in CASES:
    rows = 20_000_000
    name = "t" + "x".join(map(str, dims))
    capi.load_model(name, W.write(f"{d}/{name}.onnx", W.mlp(dims, final_softmax=sm)))
    plan = capi.get_plan(name)
    d_in, d_out = capi.DeviceBuffer(dev, rows * dims[0] * 4), capi.DeviceBuffer(dev, rows * dims[-1] * 4)
    capi.synth_fill(d_in, 42, 0, rows, dims[0])
    capi.predict_device(name, d_in, rows, dims[0], d_out)
    ms = capi.time_predict_device(name, d_in, rows, dims[0], d_out, 5) / 5
    byts = rows * 4 * (dims[0] + dims[-1])
    print(f"{'x'.join(map(str, dims)):<10} sm={int(sm)} {ms:8.3f} ms  {byts / ms / 1e9:6.2f} TB/s(in+out)  {rows / ms / 1e6:7.1f} G rows/s  {(plan.get('chain_kernels') or [plan.get('fused_kernel', ','.join(plan['exec']))])[0][:50]}")
    capi.unload_model(name); del d_in, d_out
