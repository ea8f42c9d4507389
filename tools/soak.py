"""Soak: many caller threads mixing host-ABI predictions on several models (tabular + conv, chain / mlp3 / tiled kernels)
with concurrent load / unload cycles of other models, for a fixed wall time; every result is compared with the first one
computed for the same (model, slice).  usage (GPU box): python tools/soak.py [seconds] [threads]"""
import os, sys, tempfile, threading, time, random
sys.path.insert(0, os.getcwd())
import numpy as np
from infera_amd import capi, onnx_writer as W, synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = tempfile.mkdtemp()
specs = {"c2": W.mlp(), "logreg": W.logreg_softmax(), "skl": W.mlp((30, 100, 2), final_softmax=True), "tiny": W.mlp((13, 1)),
         "wide": W.mlp((561, 6), final_softmax=True), "lab": W.sklearn_pipeline(30, 3)}
cols = {"c2": 128, "logreg": 128, "skl": 30, "tiny": 13, "wide": 561, "lab": 30}
paths = {k: W.write(f"{d}/{k}.onnx", v) for k, v in specs.items()}
blob_path = W.write(f"{d}/zoo.onnx", W.zoo_ops_net()[0])
for k, p in paths.items():
    capi.load_model(k, p)
capi.load_model("zoo", blob_path)
capi.load_model("r18s", W.write(f"{d}/r18s.onnx", W.resnet18(in_hw=64)))  # fused stem + pool, weight-stationary and tiled convolutions
imgs64 = synth.table(9, 0, 6, 3 * 64 * 64)
imgs64_long = synth.table(10, 0, 531, 3 * 64 * 64)  # >= 512 rows: the pass runs as two lanes on two streams (hip/exec.cpp exec_plan)
tables = {k: synth.table(7, 0, 4096, c) for k, c in cols.items()}
imgs = synth.table(8, 0, 24, 3 * 16 * 16)
# the same tables column-major in REGISTERED memory: predict_columns over their runs is served zero-copy (the GPU reads them in place)
# and must give the bits of the row-major host entry on the same rows
tables_cm = {k: np.ascontiguousarray(t.T) for k, t in tables.items()}
for t in tables_cm.values():
    capi.register_host_memory(t)
ref, ref_mu = {}, threading.Lock()
errors, counts = [], [0] * nthreads
stop = time.time() + secs


def check(key, got):
    with ref_mu:
        want = ref.setdefault(key, got.copy())
    if not np.array_equal(want, got):
        errors.append((key, float(np.abs(want - got).max())))


def worker(t):
    rng = random.Random(t)
    try:
        while time.time() < stop and not errors:
            r = rng.random()
            if r < 0.60:
                k = rng.choice(list(cols))
                lo = rng.randrange(0, 2048, 256)
                n = rng.choice([1, 33, 500, 2048])
                check((k, lo, n), capi.predict(k, tables[k][lo:lo + n]))
            elif r < 0.80:
                k = rng.choice(list(cols))
                lo = rng.randrange(0, 2048, 256)
                n = rng.choice([1, 33, 500, 2048])
                check((k, lo, n), capi.predict_columns(k, [tables_cm[k][c, lo:lo + n] for c in range(cols[k])], rows=n))
            elif r < 0.88:
                n = rng.choice([1, 5, 24])
                check(("zoo", n), capi.predict_from_blob("zoo", imgs[:n].tobytes()))
            elif r < 0.915:
                n = rng.choice([1, 3, 6])
                check(("r18s", n), capi.predict_from_blob("r18s", imgs64[:n].tobytes()))
            elif r < 0.92:
                check(("r18s_long", 531), capi.predict_from_blob("r18s", imgs64_long.tobytes()))
            else:
                name = f"tmp{t}"
                capi.load_model(name, paths[rng.choice(["skl", "tiny", "wide"])])
                capi.unload_model(name)
            counts[t] += 1
    except Exception as e:  # noqa: BLE001
        errors.append((t, repr(e)))


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024
    return -1.0


def sampler():  # host RSS every few seconds: load / unload cycles and leased contexts must not grow it without bound
    while time.time() < stop:
        samples.append((round(time.time() - t_start, 1), round(rss_mb())))
        time.sleep(max(2.0, secs / 12))


samples, t_start = [], time.time()
threading.Thread(target=sampler, daemon=True).start()
th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
[x.start() for x in th]
[x.join(timeout=secs + 120) for x in th]
stuck = sum(x.is_alive() for x in th)
print(f"calls={sum(counts)} threads={nthreads} stuck={stuck} errors={errors[:3]} zero_copy_calls={capi.zero_copy_calls()} "
      f"hipgraph={os.environ.get('INFERA_HIPGRAPH', '0')}")
print("host RSS (s, MB):", samples)
sys.exit(1 if (errors or stuck) else 0)
