"""The multi-device code path on a ONE-GPU box.

`INFERA_DEVICES=0,0` makes two device *slots* (both on HIP ordinal 0): weights are uploaded once per slot, caller
threads are dealt round-robin over the slots, each slot has its own staging contexts and submission gate.  Results
placed by row offset must equal the single-slot scan bit for bit and the device-resident scan of the same rows.
Also: bench.py's N>1 control path (two ranks sharing device 0) and its single-process `--host-path` shape.
The library reads INFERA_DEVICES once per process, so every case runs in a child process."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCAN = r"""
import os, sys, json, threading
sys.path.insert(0, %(root)r)
import numpy as np
from infera_amd import capi, onnx_writer, synth
rows, cols, CH, T = %(rows)d, 128, 2048, %(threads)d
path = onnx_writer.write(os.path.join(%(tmp)r, "mlp.onnx"), onnx_writer.mlp((128, 256, 64, 1)))
capi.load_model("m", path)
x = synth.table(42, 0, rows, cols)
colmajor = np.ascontiguousarray(x.T)            # one contiguous run per column, like a DuckDB vector
out = np.empty((rows, 1), np.float32)
nchunks = (rows + CH - 1) // CH
lock, nxt, errs = threading.Lock(), [0], []
def worker():
    try:
        while True:
            with lock:
                c = nxt[0]; nxt[0] += 1
            if c >= nchunks: return
            r0, r1 = c * CH, min(rows, (c + 1) * CH)
            out[r0:r1] = capi.predict_columns("m", [colmajor[j, r0:r1] for j in range(cols)])
    except Exception as e:  # noqa
        errs.append(repr(e))
ths = [threading.Thread(target=worker) for _ in range(T)]
[t.start() for t in ths]; [t.join() for t in ths]
assert not errs, errs
# the same rows through the device-resident entry (slot of ordinal 0)
dev = capi.device_ordinal(0)
d_in = capi.DeviceBuffer(dev, rows * cols * 4); d_out = capi.DeviceBuffer(dev, rows * 4)
capi.synth_fill(d_in, 42, 0, rows, cols)
capi.predict_device("m", d_in, rows, cols, d_out)
resident = d_out.download((rows, 1))
np.save(os.path.join(%(tmp)r, "out_%(tag)s.npy"), out)
print("RESULT " + json.dumps({"devices": capi.get_devices()["devices"], "resident_equal": bool(np.array_equal(resident, out)),
                              "count": capi.device_count()}))
"""


def _child(code, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


@pytest.mark.gpu
def test_two_device_slots_on_one_gpu_match_single_slot(gpu_api, tmp_path):
    rows = 2048 * 37 + 311
    args = {"root": ROOT, "rows": rows, "threads": 6, "tmp": str(tmp_path)}
    two = _child(SCAN % dict(args, tag="two"), {"INFERA_DEVICES": "0,0"})
    one = _child(SCAN % dict(args, tag="one"), {"INFERA_DEVICES": "0"})
    assert two["count"] == 2 and one["count"] == 1
    slots = two["devices"]
    assert [d["slot"] for d in slots] == [0, 1] and all(d["ordinal"] == 0 for d in slots)
    # both slots really took chunks (threads are dealt round-robin), and every row was served exactly once
    assert all(d["host_rows"] > 0 and d["host_calls"] > 0 for d in slots), slots
    assert sum(d["host_rows"] for d in slots) == rows and one["devices"][0]["host_rows"] == rows
    a, b = np.load(tmp_path / "out_two.npy"), np.load(tmp_path / "out_one.npy")
    assert np.array_equal(a, b)                      # placement by row offset == single-slot scan, bit for bit
    assert two["resident_equal"] and one["resident_equal"]
    from infera_amd import synth
    from oracle import oracle

    path = os.path.join(str(tmp_path), "mlp.onnx")
    want = oracle.Model(path).predict(synth.table(42, 0, rows, 128))
    assert np.all(np.abs(a - want) <= 1e-4 * np.abs(want) + 1e-6)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_bench_two_ranks_sharing_one_gpu(gpu_api):
    """bench.py's N>1 control path (gloo barrier, max over ranks, per-rank row ranges, concurrent host scans)."""
    from tests.conftest import run_bench

    env = dict(os.environ)
    env.pop("INFERA_DEVICES", None)
    line, full = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", str(2048 * 300), "--share-device", "0", "--e2e-threads", "4", "--e2e-reps", "3"],
                           env=env, launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                              "--master-port", str(_free_port())], timeout=900)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["rows_per_gpu"] == 2048 * 300 and line["config"]["parallelism"] == "row-range x2"
    assert line["end_to_end"]["rows_per_s"] > 0 and line["end_to_end"]["threads"] == 4
    e = full["end_to_end"]
    assert e["ranks"] == 2 and e["threads_per_rank"] == 4 and len(e["scan_seconds"]) == 3 and e["rows_per_s"] > 0
    assert "cpu_baseline" not in line and "cpu_baseline" not in full  # rank 0 at N=1 only


@pytest.mark.gpu
def test_bench_eight_ranks_sharing_one_gpu_is_what_the_driver_will_launch(gpu_api, tmp_path):
    """the driver's 8-GPU run, dry on ONE GPU -- `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`
    with every rank on device 0 (--share-device) and the rows cut to fit a test.  ONE JSON line <= 4 KB from rank 0; eight ranks scanned at
    once through the host path, each its own table; every rank's rows are accounted for and ranks 0 and 7 reproduce the ORACLE's checksum of
    their table; the N > 1 line carries what answers the scaling question (callers per GPU, CPU per chunk, host read rate, the registered
    scan) and no cpu_baseline (rank 0 at N = 1 only)."""
    import time

    from infera_amd import onnx_writer as W
    from infera_amd import sqlharness
    from oracle import oracle
    from tests.conftest import run_bench

    rows, reps = 2048 * 120, 2
    env = dict(os.environ)
    env.pop("INFERA_DEVICES", None)
    t0 = time.time()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    # NO launcher: a plain `python bench.py --gpus 8` -- the shape the driver uses at N = 1 -- must run 8 ranks all the same
    # (bench.py re-executes itself under torch.distributed.run); the 2-rank test above keeps the driver's own launcher shape covered
    line, full = run_bench(["--gpus", "8", "--steps", "3", "--warmup", "1", "--rows", str(rows), "--share-device", "0", "--e2e-reps", str(reps)],
                           env=env, timeout=1500)
    wall = time.time() - t0
    assert wall < 900, wall  # (the driver's own limit per bench run is well above this; 8 ranks on one GPU take ~2 minutes)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0 and line["config"]["parallelism"] == "row-range x8"
    assert "cpu_baseline" not in line and line["value_is"] == "device_resident" and line["value_end_to_end"] > 0
    e = full["end_to_end"]
    assert e["ranks"] == 8 and len(e["per_rank"]) == 8 and sorted(r["rank"] for r in e["per_rank"]) == list(range(8))
    assert all(r["ordinal"] == 0 and r["rows_this_run"] == rows * reps for r in e["per_rank"]), e["per_rank"]
    assert e["callers_per_gpu"] == e["threads_per_rank"] >= 2 and e["host_cpu_cost"]["cpu_us_per_chunk"] > 0
    assert e["rows_per_s"] == pytest.approx(8 * rows / e["median_scan_seconds"], rel=1e-6) and e["host_read_gbs"] == pytest.approx(e["rows_per_s"] * 512 / 1e9)
    le = line["end_to_end"]
    assert le["callers_per_gpu"] >= 2 and le["cpu_us_per_chunk"] > 0 and le["host_read_gbs"] > 0 and le["rows_per_s_per_gpu"] == pytest.approx(le["rows_per_s"] / 8, rel=1e-3)
    g = line["end_to_end_registered"]
    assert "error" not in g and g["rows_per_s"] > 0 and g["zero_copy_calls"] > 0 and g["few_callers"]["rows_per_s"] > 0
    # self-normalising: rank 0 scanned ALONE first, in this run, on this box; the scaling figure is all ranks over that
    assert line["value_end_to_end_alone"] > 0 and line["scaling_vs_alone"] == pytest.approx(line["value_end_to_end"] / line["value_end_to_end_alone"], rel=1e-3)
    assert e["alone"]["rows_per_s"] == e["alone_rows_per_s"] and e["alone"]["cpu_us_per_chunk"] > 0 and e["scaling_vs_alone"] > 0
    assert g["alone_rows_per_s"] > 0 and g["scaling_vs_alone"] > 0 and g["vs_staged_alone"] > 0
    # each rank scanned ITS table (seed 42 + rank): the oracle's scan of the same table gives the same sum of outputs (fp32 results summed in
    # double: equal to the 1e-4 parity bar, scaled by the rows)
    model = oracle.Model(W.write(str(tmp_path / "mlp.onnx"), W.mlp((128, 256, 64, 1))))
    for r in (0, 7):
        table = sqlharness.synth_table(rows, 128, 42 + r, 8)
        _, want = oracle.bench_scan_table(model, table, rows, 128, threads=8, boxed=2)
        got = [x for x in e["per_rank"] if x["rank"] == r][0]["checksum"]
        assert abs(got - want) <= 1e-4 * rows * 0.05 + 1e-3, (r, got, want)


@pytest.mark.gpu
def test_bench_host_path_single_process_eight_slots(gpu_api):
    """... and DuckDB's shape at 8 GPUs: ONE process, 16 worker threads dealt over eight device slots (all on GPU 0 here)."""
    from tests.conftest import run_bench

    line, full = run_bench(["--host-path", "--gpus", "8", "--share-device", "0", "--rows", str(2048 * 480), "--e2e-threads", "16", "--e2e-reps", "3"], timeout=900)
    assert line["n_gpus"] == 8 and line["config"]["INFERA_DEVICES"] == ",".join(["0"] * 8) and line["value_is"] == "end_to_end"
    slots = full["end_to_end"]["device_slots"]
    assert len(slots) == 8 and all(s["rows_this_run"] > 0 for s in slots), slots  # 16 threads: two per slot
    assert sum(s["rows_this_run"] for s in slots) == 3 * 2048 * 480


@pytest.mark.gpu
def test_bench_host_path_single_process_two_slots(gpu_api):
    """DuckDB's shape: one process, worker threads dealt over two device slots (both on GPU 0 here)."""
    from tests.conftest import run_bench

    line, full = run_bench(["--host-path", "--gpus", "2", "--share-device", "0", "--rows", str(2048 * 400), "--e2e-threads", "8", "--e2e-reps", "3"], timeout=900)
    assert line["n_gpus"] == 2 and line["config"]["INFERA_DEVICES"] == "0,0" and line["value_is"] == "end_to_end"
    slots = full["end_to_end"]["device_slots"]
    assert len(slots) == 2 and all(s["rows_this_run"] > 0 for s in slots)
    assert sum(s["rows_this_run"] for s in slots) == 3 * 2048 * 400


@pytest.mark.gpu
@pytest.mark.perf
def test_two_slots_on_one_gpu_hold_the_single_slot_rate_at_32_threads(gpu_api):
    """admission is per PHYSICAL GPU, so two device slots on one GPU admit as many calls onto its submission
    path as one slot does -- the 2-slot scan at 32 caller threads used to fall to 66 M rows/s where the 1-slot scan held 94-110.
    Asserted: within 15 % of the 1-slot scan (same process shape, same table size, medians of 5 scans; measured 0.99-1.01 -- the margin is for
    run-to-run noise between two processes, the regression it guards against was 0.6)."""
    from tests.conftest import run_bench

    rates = {}
    for slots in (1, 2):
        env = dict(os.environ)
        env.pop("INFERA_DEVICES", None)
        line, full = run_bench(["--host-path", "--gpus", str(slots), "--share-device", "0", "--rows", "6000000", "--e2e-threads", "32", "--e2e-reps", "5"], env=env, timeout=900)
        assert line["value_is"] == "end_to_end" and len(full["end_to_end"]["device_slots"]) == slots
        rates[slots] = full["end_to_end"]["rows_per_s"]
    assert rates[2] >= 0.85 * rates[1], rates


FAULT_CHILD = r"""
import json, os, sys, threading
sys.path.insert(0, %(root)r)
import numpy as np
from infera_amd import capi, onnx_writer as W, synth
capi.load_model("m", W.write(os.path.join(%(tmp)r, "m.onnx"), W.mlp((128, 256, 64, 1))))
x = synth.table(9, 0, 2048 * 24, 128)
want = capi.predict("m", x)  # (one thread, before any fault: slot of this thread)
bad, errors = [], []
def work(t):
    for i in range(t, 24 * 6, 8):
        c = i %% 24
        try:
            got = capi.predict("m", x[c * 2048:(c + 1) * 2048])
        except Exception as exc:
            errors.append(str(exc))
            continue
        if not np.array_equal(got, want[c * 2048:(c + 1) * 2048]):
            bad.append(i)
th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
[t.start() for t in th]
[t.join() for t in th]
print("RESULT " + json.dumps({"bad": bad, "errors": errors, "devices": capi.get_devices()["devices"]}))
"""


@pytest.mark.gpu
def test_a_faulting_slot_is_taken_out_of_service_and_its_callers_are_redealt(gpu_api, tmp_path):
    """SURVEY 5 "failure detection": two device slots (INFERA_DEVICES=0,0), a launch failure injected on slot 1 from its 20th call on
    (INFERA_FAULT_INJECT=1:20, the test hook).  No caller sees an error: the failing chunk and everything after it run on slot 0, bit for bit;
    infera_hip_get_devices reports slot 1 as unhealthy with the error text."""
    env = dict(os.environ, INFERA_DEVICES="0,0", INFERA_FAULT_INJECT="1:20", INFERA_LOG_LEVEL="ERROR")
    p = subprocess.run([sys.executable, "-c", FAULT_CHILD % {"root": ROOT, "tmp": str(tmp_path)}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert r["bad"] == [] and r["errors"] == [], r
    d0, d1 = r["devices"]
    assert d0["healthy"] and "fault" not in d0
    assert not d1["healthy"] and "injected fault" in d1["fault"]
    assert d0["host_rows"] > d1["host_rows"] and d0["host_rows"] + d1["host_rows"] >= 2048 * (24 + 24 * 6)
    assert "out of service" in p.stderr  # logged once, at ERROR level
    # ... and with the ONLY slot failing the error reaches the caller as a status, with the reference's "ONNX error: ..." shape
    env = dict(os.environ, INFERA_DEVICES="0", INFERA_FAULT_INJECT="0:1", INFERA_LOG_LEVEL="ERROR")
    p = subprocess.run([sys.executable, "-c", FAULT_CHILD % {"root": ROOT, "tmp": str(tmp_path)}], env=env, capture_output=True, text=True, timeout=600)
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert r["bad"] == [] and len(r["errors"]) == 24 * 6 and all(e.startswith("ONNX error: HIP: injected fault") for e in r["errors"]), r["errors"][:2]


@pytest.mark.gpu
def test_pinned_staging_of_image_batches_is_bounded_per_gpu(gpu_api, tmp_path):
    """contexts that serve big-row (BLOB) batches keep two passes of pinned staging each, for the life of the process.  32 caller
    threads x 300-image calls: at most INFERA_HOST_CONTEXTS (24) contexts exist per GPU and together they hold no more than the slot's budget
    (6 GiB + the small-chunk staging), whatever the thread count; results are the same for every thread."""
    import threading

    import numpy as np

    from infera_amd import onnx_writer as W
    from infera_amd import synth

    rng = np.random.default_rng(2)
    w = (rng.standard_normal((8, 3, 3, 3)) * 0.2).astype(np.float32)
    nodes = [W.node("Conv", ["X", "w"], ["c"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("strides", [2, 2]), W.attr_ints("pads", [1] * 4)]), W.node("Relu", ["c"], ["r"]),
             W.node("GlobalAveragePool", ["r"], ["g"]), W.node("Flatten", ["g"], ["Y"], [W.attr_i("axis", 1)])]
    path = W.write(str(tmp_path / "img.onnx"), W.model("img", nodes, [W.tensor("w", w)], [W.value_info("X", ["N", 3, 224, 224])], [W.value_info("Y", ["N", 8])]))
    x = synth.table(4, 0, 300, 3 * 224 * 224)
    blob = x.tobytes()
    gpu_api.load_model("img", path)
    try:
        want = gpu_api.predict_from_blob("img", blob)
        outs = [None] * 32

        def work(t):
            outs[t] = gpu_api.predict_from_blob("img", blob)

        th = [threading.Thread(target=work, args=(t,)) for t in range(32)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert all(o is not None and np.array_equal(o, want) for o in outs)
        pinned = sum(d["pinned_staging_bytes"] for d in gpu_api.get_devices()["devices"])
        assert 0 < pinned <= (6 << 30) + 24 * (160 << 20), pinned
    finally:
        gpu_api.unload_model("img")
