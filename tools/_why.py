import os, sys, tempfile
sys.path.insert(0, os.getcwd())
from infera_amd import capi, onnx_writer as W
d = tempfile.mkdtemp()
capi.load_model("a", W.write(f"{d}/a.onnx", W.mlp((4, 10, 3), final_softmax=True)))
print(capi.get_plan("a")["exec"])
