"""Times the sklearn-exporter graphs (ai.onnx.ml Scaler -> LinearClassifier / LinearRegressor) on 20M-row tables.
usage (GPU box): python tools/sklearn_models_time.py"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
from infera_amd import capi, onnx_writer as W
d = tempfile.mkdtemp(); dev = capi.device_ordinal(0)
CASES = [(30, 3, "classifier", "SOFTMAX", None, None, True, "label"),
         (30, 3, "classifier", "SOFTMAX", None, "L1", True, "scores"),
         (13, 2, "classifier", "LOGISTIC", [-1, 1], "L1", True, "label"),
         (13, 2, "classifier", "LOGISTIC", None, "L1", True, "scores"),
         (64, 10, "classifier", "NONE", None, None, False, "label"),
         (30, 1, "regressor", "NONE", None, None, True, "scores"),
         (100, 1, "regressor", "NONE", None, None, True, "scores")]
for i, c in enumerate(CASES):
    f, e, kind, post, labels, norm, scaler, output = c
    rows = 20_000_000
    name = f"skl{i}"
    capi.load_model(name, W.write(f"{d}/{name}.onnx", W.sklearn_pipeline(*c)))
    plan = capi.get_plan(name)
    oc = 1 if (output == "label" and kind == "classifier") else e
    d_in, d_out = capi.DeviceBuffer(dev, rows * f * 4), capi.DeviceBuffer(dev, rows * oc * 4)
    capi.synth_fill(d_in, 42, 0, rows, f)
    capi.predict_device(name, d_in, rows, f, d_out)
    ms = capi.time_predict_device(name, d_in, rows, f, d_out, 5) / 5
    byts = rows * 4 * (f + oc)
    kinds = "+".join(s["kind"] for s in plan["plan"]["steps"])
    print(f"{kind[:5]} {f:>3}x{e:<2} {post:<8} norm={norm} out={output:<6} {ms:8.3f} ms  {byts / ms / 1e9:6.2f} TB/s(in+out)  {rows / ms / 1e6:7.1f} G rows/s  {kinds}  [{','.join(plan['exec'])}]")
    capi.unload_model(name); del d_in, d_out
