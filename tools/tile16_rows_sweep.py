"""Where the 16-row tile kernel (mlp3_tile16_kernel) hands over to the 32-row one: device-resident C2 launches of N rows under each, mean of 200 back-to-back launches
(HIP events on the launching stream).  usage (GPU box): INFERA_MLP_TILE16_MAX_ROWS=0|1048576 python tools/tile16_rows_sweep.py"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from infera_amd import capi, onnx_writer as W
d = tempfile.mkdtemp()
capi.load_model("m", W.write(f"{d}/m.onnx", W.mlp()))
dev = capi.device_ordinal(0)
rows_max = 32768
d_in, d_out = capi.DeviceBuffer(dev, rows_max * 128 * 4), capi.DeviceBuffer(dev, rows_max * 4)
capi.synth_fill(d_in, 42, 0, rows_max, 128)
out = []
for rows in (16, 256, 1024, 2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768):
    capi.time_predict_device("m", d_in, rows, 128, d_out, 20)
    ms = capi.time_predict_device("m", d_in, rows, 128, d_out, 200) / 200
    out.append(f"{rows}: {1e3 * ms:.2f} us")
print(f"INFERA_MLP_TILE16_MAX_ROWS={os.environ.get('INFERA_MLP_TILE16_MAX_ROWS', 'default')}  " + "  ".join(out))
