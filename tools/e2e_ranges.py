#!/usr/bin/env python3
"""Per-chunk timeline of the host path from a rocprofv3 rocpd .db recorded with --kernel-trace --memory-copy-trace --marker-trace and
INFERA_PROFILE=1 (the roctx ranges of csrc/hip/profile.hpp): for every "infera:chunk" range of a caller thread the named stages on the CPU
side and, from the device records dispatched inside it, where the chunk's wall time goes on the device side --

    chunk begin -> [gather] -> [gate] -> [enqueue] -> first device record begins -> ... -> last device record ends -> [wait] returns -> [copy_out]

medians over the steady part of the scan (the first 20 % of chunks are skipped).  usage: e2e_ranges.py <results.db> [label]"""
import sqlite3
import statistics
import sys
from bisect import bisect_left
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
label = sys.argv[2] if len(sys.argv) > 2 else ""
# (rocprofv3 7.2 names every roctx range "roctxThreadRangeA"; the pushed string is the "message" of the row's extdata)
regs = [r for r in db.execute("select tid, coalesce(json_extract(extdata, '$.message'), name), start, end from regions order by start") if str(r[1]).startswith("infera:")]
kern = list(db.execute("select tid, start, end, name from kernels order by start"))
try:
    cops = list(db.execute("select tid, start, end, size, name from memory_copies order by start"))
except sqlite3.OperationalError:
    cops = []
by_tid = defaultdict(lambda: defaultdict(list))
for tid, name, s, e in regs:
    by_tid[tid][name].append((s, e))
dev_by_tid = defaultdict(list)
for tid, s, e, name in kern:
    dev_by_tid[tid].append((s, e, "k:" + name.split("<")[0].split("(")[0].split("::")[-1][:28]))
for tid, s, e, size, name in cops:
    dev_by_tid[tid].append((s, e, f"copy:{'H2D' if 'HOST_TO' in name.upper() or 'H2D' in name.upper() else 'D2H' if 'TO_HOST' in name.upper() or 'D2H' in name.upper() else name[:12]}:{size}"))
for v in dev_by_tid.values():
    v.sort()


def med(xs):
    return statistics.median(xs) / 1e3 if xs else float("nan")


def p90(xs):
    return sorted(xs)[int(len(xs) * 0.9)] / 1e3 if xs else float("nan")


rows = defaultdict(list)
seq_names = defaultdict(int)
nchunks = 0
for tid, names in by_tid.items():
    chunks = names.get("infera:chunk", [])
    chunks = chunks[len(chunks) // 5:]
    dev = dev_by_tid.get(tid, [])
    starts = [d[0] for d in dev]
    sub = {n: sorted(v) for n, v in names.items() if n != "infera:chunk"}
    sub_starts = {n: [x[0] for x in v] for n, v in sub.items()}
    prev_end = None
    for cs, ce in chunks:
        nchunks += 1
        rows["chunk (call begin -> return)"].append(ce - cs)
        if prev_end is not None:
            rows["between calls (binding + scan loop)"].append(cs - prev_end)
        prev_end = ce
        stage = {}
        for n in sub:
            i = bisect_left(sub_starts[n], cs)
            if i < len(sub[n]) and sub[n][i][1] <= ce:
                stage[n] = sub[n][i]
                rows["cpu " + n].append(sub[n][i][1] - sub[n][i][0])
        i = bisect_left(starts, cs)
        mine = []
        while i < len(dev) and dev[i][0] < ce:
            mine.append(dev[i])
            i += 1
        if not mine:
            continue
        seq_names[" -> ".join(d[2] for d in mine)] += 1
        enq = stage.get("infera:enqueue")
        wait = stage.get("infera:wait")
        if enq:
            rows["dev  enqueue begin -> first device record begins"].append(mine[0][0] - enq[0])
            rows["dev  enqueue end   -> first device record begins"].append(mine[0][0] - enq[1])
        for j, d in enumerate(mine):
            rows[f"dev  [{j}] {d[2]} runs"].append(d[1] - d[0])
            if j:
                rows[f"dev  gap [{j - 1}] end -> [{j}] begin"].append(d[0] - mine[j - 1][1])
        rows["dev  first record begins -> last ends"].append(mine[-1][1] - mine[0][0])
        if wait:
            rows["dev  last device record ends -> wait returns"].append(wait[1] - mine[-1][1])
print(f"== {label} {sys.argv[1]}: {nchunks} chunks on {len(by_tid)} caller thread(s) (steady part); microseconds, median / p90")
for k, v in rows.items():
    print(f"  {k:<58} {med(v):8.1f} {p90(v):8.1f}   n={len(v)}")
print("  device records per chunk:")
for s, n in sorted(seq_names.items(), key=lambda x: -x[1])[:4]:
    print(f"    {n:>6} x  {s}")
# which hardware queue served which caller thread, and how much of the kernels' time overlapped a kernel of ANOTHER caller
qs = defaultdict(set)
iv = []
for tid, s, e, q in db.execute("select tid, start, end, queue_id from kernels order by start"):
    qs[tid].add(q)
    iv.append((s, e, tid))
print("  hardware queue ids by dispatching thread:", {t: sorted(v) for t, v in qs.items()})
iv = iv[len(iv) // 5:]
tot = sum(e - s for s, e, _ in iv)
ov = 0
for i, (s, e, t) in enumerate(iv):
    j = i + 1
    while j < len(iv) and iv[j][0] < e:
        if iv[j][2] != t:
            ov += min(e, iv[j][1]) - iv[j][0]
        j += 1
if tot:
    print(f"  kernel time overlapped by another caller's kernel: {ov / tot:.3f} of {tot / 1e6:.1f} ms of kernels")
