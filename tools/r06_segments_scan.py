#!/usr/bin/env python3
"""(round 6) ONE table shape of tools/r06_duckdb_blocks.py, ONE caller count -- the command rocprofv3 wraps for the per-chunk timeline
(tools/r06_segments_ranges.sh).   usage: r06_segments_scan.py staged|rect|segments [rows] [callers] [reps]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from infera_amd import capi, onnx_writer, sqlharness  # noqa: E402

mode = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_500_000
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
tmp = tempfile.mkdtemp()
capi.load_model("m", onnx_writer.write(os.path.join(tmp, "mlp.onnx"), onnx_writer.mlp((128, 256, 64, 1))))
node = capi.get_devices()["devices"][0].get("numa_node", -1)
if node >= 0:
    try:
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, cpus & os.sched_getaffinity(0))
    except OSError:
        pass
if mode == "segments":
    os.environ["INFERA_ZERO_COPY_ALLOCATOR"] = "1"
    table, scan = sqlharness.SegmentTable(rows, 128, 42, 16), sqlharness.bench_scan_segments
else:
    table, scan = sqlharness.synth_table(rows, 128, 42, 16), sqlharness.bench_scan_table
    if mode == "rect":
        capi.register_host_memory(table)
scan("infera_predict", "m", table, min(rows, 2048 * 200), 128, threads, 1)
(secs, cs), ph = sqlharness.phase_breakdown(scan, "infera_predict", "m", table, rows, 128, threads, reps)
med = sorted(secs)[len(secs) // 2]
print(f"threads={threads:3d} {mode}: {rows / med / 1e6:8.2f} M rows/s  scans={[round(s, 4) for s in secs]}  checksum={cs:.4f}")
print("phases (us per chunk): " + "  ".join(f"{k}={v}" for k, v in ph.items()))
if mode == "segments":
    table.close()
elif mode == "rect":
    capi.unregister_host_memory(table)
