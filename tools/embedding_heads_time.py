import os, sys, tempfile
sys.path.insert(0, os.getcwd())
from infera_amd import capi, onnx_writer as W
d = tempfile.mkdtemp(); dev = capi.device_ordinal(0)
for dims, sm in [((768, 10), True), ((1024, 2), True), ((1536, 10), True), ((2048, 10), True), ((2048, 1), False), ((4096, 16), True), ((3000, 5), True), ((2048, 64, 1), False), ((1536, 128, 10), True)]:
    rows = 1_000_000
    name = "t" + "x".join(map(str, dims))
    capi.load_model(name, W.write(f"{d}/{name}.onnx", W.mlp(dims, final_softmax=sm)))
    plan = capi.get_plan(name)
    d_in, d_out = capi.DeviceBuffer(dev, rows * dims[0] * 4), capi.DeviceBuffer(dev, rows * dims[-1] * 4)
    capi.synth_fill(d_in, 42, 0, rows, dims[0])
    capi.predict_device(name, d_in, rows, dims[0], d_out)
    ms = capi.time_predict_device(name, d_in, rows, dims[0], d_out, 5) / 5
    byts = rows * 4 * (dims[0] + dims[-1])
    print(f"{'x'.join(map(str, dims)):<14} sm={int(sm)} {ms:8.3f} ms  {byts / ms / 1e9:6.2f} TB/s(in+out)  {(plan.get('chain_kernels') or [','.join(plan['exec'])])[0][:60]}")
    capi.unload_model(name); del d_in, d_out
