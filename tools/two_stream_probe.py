"""C5 (ResNet-18, 1024 images resident in HBM) as ONE device-resident scan versus the same images split over T caller
threads (each its own stream / scratch): does running two half-batches side by side fill the tails and overlap the
HBM-bound kernels (max-pool, residual adds) with the MFMA-bound ones?   usage (GPU box): python tools/two_stream_probe.py"""
import os, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infera_amd import capi, onnx_writer

rows, cols, out_cols, reps = 1024, 3 * 224 * 224, 1000, 6
tmp = tempfile.mkdtemp()
capi.load_model("r18", onnx_writer.write(os.path.join(tmp, "r18.onnx"), onnx_writer.resnet18(in_hw=224)))
d_in = capi.DeviceBuffer(0, rows * cols * 4)
d_out = capi.DeviceBuffer(0, rows * out_cols * 4)
capi.synth_fill(d_in, 42, 0, rows, cols)


def scan(threads: int) -> float:
    per = rows // threads
    def work(t):
        for _ in range(reps):
            capi.predict_device("r18", d_in, per, cols, d_out, sync=True, in_offset_bytes=t * per * cols * 4, out_offset_bytes=t * per * out_cols * 4)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return (time.perf_counter() - t0) / reps * 1e3


for t in (1, 2, 4):
    scan(t)  # warm: contexts, scratch
for rnd in range(2):
    for t in (1, 2, 3, 4, 8):
        if rows % t: continue
        ms = scan(t)
        print(f"threads {t}: {ms:.2f} ms per 1024 images = {rows / ms * 1e3:.0f} img/s", flush=True)
