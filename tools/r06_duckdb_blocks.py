#!/usr/bin/env python3
"""(round 6) the C2 scan through the extension over three shapes of the same 10M x 128 host table, by callers per GPU:
  staged       one contiguous columnar table, nothing registered (the drop-in path)
  rect         the same table registered as ONE range (round 5's measurement: qualifies for 2-D copies, <= 3 in flight + pulling kernel)
  segments     DuckDB's shape: per (row group, column) two 256 KiB blocks from the extension's registering allocator, 8-byte block header,
               65,534 values per segment, 128 unrelated block addresses per chunk -> the pulling kernel for every chunk; one chunk in 60
               straddles two segments and is staged
M rows/s and process CPU microseconds per chunk (getrusage), median of `reps` scans.   usage: r06_duckdb_blocks.py [rows] [reps] [callers,...]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from infera_amd import capi, onnx_writer, sqlharness  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
callers = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4, 6, 8, 16]
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


def measure(label, scan, table):
    out = []
    scan("infera_predict", "m", table, min(rows, 2048 * 400), 128, callers[0], 1)
    for t in callers:
        (secs, cs), ph = sqlharness.phase_breakdown(scan, "infera_predict", "m", table, rows, 128, t, reps)
        med = sorted(secs)[len(secs) // 2]
        out.append(f"{rows / med / 1e6:.1f} / {ph['cpu_us_per_chunk']:.0f}")
    print(f"| {label} | " + " | ".join(out) + " |", flush=True)


print(f"C2 scan, {rows} rows, median of {reps} scans, M rows/s / CPU us per chunk")
print("| callers per GPU | " + " | ".join(str(t) for t in callers) + " |")
print("|---|" + "---|" * len(callers))
flat = sqlharness.synth_table(rows, 128, 42, 16)
measure("staged (contiguous table, not registered)", sqlharness.bench_scan_table, flat)
capi.register_host_memory(flat)
before = capi.zero_copy_calls()
measure("registered, ONE contiguous range (2-D copies + pulling kernel)", sqlharness.bench_scan_table, flat)
capi.unregister_host_memory(flat)
del flat
os.environ["INFERA_ZERO_COPY_ALLOCATOR"] = "1"
for label, arena, shuffled, alloc_threads in (
        ("registered, DuckDB segments, ARENA allocator, a row group's blocks allocated back to back by ONE thread (2-D copies + pulling kernel)", "1", False, 1),
        ("... by FOUR threads at once, each a row group's segment at a time (a parallel load)", "1", False, 4),
        ("registered, DuckDB segments, ARENA allocator, blocks allocated in a random order (pulling kernel only)", "1", True, 1),
        ("registered, DuckDB segments, one registration per block (round 6's first shape; pulling kernel only)", "0", True, 1)):
    os.environ["INFERA_ZERO_COPY_ARENA"] = arena
    seg = sqlharness.SegmentTable(rows, 128, 42, 16, shuffled=shuffled, alloc_threads=alloc_threads)
    print(f"(segment table: {seg.blocks} blocks of 256 KiB, registering allocator = {seg.registering_allocator}, arena = {arena}, shuffled = {shuffled}, made in "
          f"{seg.create_seconds:.2f} s; registered ranges now {capi.get_devices()['registered_host_ranges']})", flush=True)
    before = capi.zero_copy_calls()
    measure(label, sqlharness.bench_scan_segments, seg)
    print(f"(zero-copy calls {capi.zero_copy_calls() - before}, assembled chunks in the last scan set {seg.assembled_chunks})")
    seg.close()
