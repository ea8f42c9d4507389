import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
from infera_amd import capi, onnx_writer as W
d = tempfile.mkdtemp()
p = W.write(d + "/m.onnx", W.mlp((128, 256, 128, 1)))
capi.load_model("warm", W.write(d + "/w.onnx", W.mlp((13, 1))))  # HIP init outside the measurement
t0 = time.perf_counter(); capi.load_model("a", p); t1 = time.perf_counter()
capi.load_model("b", p); t2 = time.perf_counter()
print(f"JIT MLP 128x256x128x1: first load {1e3*(t1-t0):.0f} ms (INFERA_JIT_CACHE_DIR={os.environ.get('INFERA_JIT_CACHE_DIR','default')}), second model of the same shape {1e3*(t2-t1):.1f} ms")
