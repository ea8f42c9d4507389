#!/usr/bin/env python3
"""PCIe-inclusive rate of the SQL path: T worker threads x 2048-row chunks through
infera_sql_call("infera_predict") (gather -> C ABI -> H2D -> kernel -> D2H -> result vector)."""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infera_amd import capi, onnx_writer, sqlharness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4_000_000)
ap.add_argument("--threads", default="1,2,4,8,16,32,64")
ap.add_argument("--workload", default="mlp")
ap.add_argument("--dims", default="", help="custom MLP instead of a named workload, e.g. 30,100,2 (first = table columns)")
ap.add_argument("--softmax", action="store_true")
ap.add_argument("--pool", action="store_true", help="cache-resident per-thread chunk pool instead of the materialised table (round-1 behaviour)")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--double", action="store_true", help="DOUBLE columns (DuckDB's default floating type) instead of FLOAT")
ap.add_argument("--register", action="store_true", help="register the host table (infera_hip_register_host_memory): the opt-in zero-copy path")
ap.add_argument("--align", type=int, default=0, help="allocate the host table on this byte boundary (0: wherever numpy puts it)")
ap.add_argument("--huge", action="store_true", help="back the host table with transparent huge pages (madvise)")
ap.add_argument("--numa", default="off", choices=["auto", "off"], help="auto: bind the process to the CPUs of the (first) GPU's NUMA node before any thread exists")
a = ap.parse_args()
if a.numa == "auto":
    node = capi.get_devices()["devices"][0].get("numa_node", -1)
    if node >= 0:
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        os.sched_setaffinity(0, cpus & os.sched_getaffinity(0))
        print(f"bound to NUMA node {node}: {len(os.sched_getaffinity(0))} CPUs")
tmp = tempfile.mkdtemp()
if a.dims:
    dims = tuple(int(x) for x in a.dims.split(","))
    blob, cols, fn = onnx_writer.mlp(dims, final_softmax=a.softmax), dims[0], "infera_predict" if dims[-1] == 1 else "infera_predict_array"
else:
    blob = onnx_writer.mlp() if a.workload == "mlp" else onnx_writer.logreg_softmax()
    cols, fn = 128, "infera_predict" if a.workload == "mlp" else "infera_predict_array"
capi.load_model("m", onnx_writer.write(os.path.join(tmp, "m.onnx"), blob))
sqlharness.bench_scan(fn, "m", 2048 * 64, cols, 4)  # warm
print(f"devices={capi.get_devices()['devices']} workload={a.dims or a.workload} rows={a.rows} exec={capi.get_plan('m')['exec']} "
      f"env={ {k: v for k, v in os.environ.items() if k.startswith('INFERA_')} } source={'per-thread chunk pool' if a.pool else 'materialised columnar host table'}")
import numpy as np  # noqa: E402
table = None if a.pool else sqlharness.synth_table(a.rows, cols, 42, 16, np.float64 if a.double else np.float32, align=a.align, huge=a.huge)
if table is not None:
    print(f"table at 0x{table.ctypes.data:x} (offset in its 4 KiB page: {table.ctypes.data % 4096}, in its 2 MiB page: {table.ctypes.data % (2 << 20)}) huge={a.huge}")
print("column type:", "DOUBLE" if a.double else "FLOAT")
if a.register and table is not None:
    import time
    t0 = time.perf_counter()
    capi.register_host_memory(table)
    print(f"host table registered in {time.perf_counter() - t0:.3f} s ({table.nbytes / 1e9:.2f} GB): chunks are read in place by the GPU")
for t in [int(x) for x in a.threads.split(",")]:
    if a.pool:
        sec, cs = sqlharness.bench_scan(fn, "m", a.rows, cols, t)
        secs = [sec]
    else:
        (secs, cs), phases = sqlharness.phase_breakdown(sqlharness.bench_scan_table, fn, "m", table, a.rows, cols, t, a.reps)
    sec = sorted(secs)[len(secs) // 2]
    print(f"threads={t:>3}  {a.rows / sec / 1e6:>9.2f} M rows/s  ({a.rows * cols * 4 / sec / 1e9:.2f} GB/s of features)  "
          f"scans={[round(x, 4) for x in secs]}  checksum={cs:.4f}" + ("" if a.pool else f"\n             us/chunk/thread: {phases}"))
if a.register and table is not None:
    print("zero-copy calls:", capi.zero_copy_calls())
    capi.unregister_host_memory(table)
