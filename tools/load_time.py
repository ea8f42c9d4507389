"""infera_load_model latency by model kind (parse + lower + schedule [+ hipRTC] + pack + upload), first and second load in
a process, with a cold and a warm on-disk JIT cache.  usage (GPU box): python tools/load_time.py"""
import os, sys, tempfile, time, shutil
cache = tempfile.mkdtemp(prefix="jitc_")
os.environ["INFERA_JIT_CACHE_DIR"] = cache
sys.path.insert(0, os.getcwd())
from infera_amd import capi, onnx_writer as W
d = tempfile.mkdtemp()
models = {"linear 3->1": W.linear_dyn(), "C2 mlp (AOT fused)": W.mlp(), "30->100->2 (hipRTC chain)": W.mlp((30, 100, 2), final_softmax=True),
          "32->64->32->1 (hipRTC mlp3)": W.mlp((32, 64, 32, 1)), "sklearn pipeline": W.sklearn_pipeline(30, 3), "resnet18 (45 MB)": W.resnet18(),
          "mobilenet_v2": W.mobilenet_v2(classes=1000, in_hw=224, width_mult=1.0)}
for name, blob in models.items():
    p = W.write(f"{d}/m.onnx", blob)
    ts = []
    for i in range(2):
        t0 = time.perf_counter(); capi.load_model(f"m{i}", p); ts.append(time.perf_counter() - t0); capi.unload_model(f"m{i}")
    print(f"{name:<30} {os.path.getsize(p) / 1e6:7.2f} MB   first {ts[0] * 1e3:8.1f} ms   again {ts[1] * 1e3:8.1f} ms")
print("jit cache files:", len(os.listdir(cache)))
