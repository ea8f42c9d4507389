#!/usr/bin/env python3
"""From a rocprofv3 rocpd .db: over the window of the LONGEST run of kernels without a gap > 50 ms, the share of wall time with at least one
kernel running (union of [start, end]) and the average number of kernels running.  Says whether an end-to-end scan is bound by the GPU."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
endcol = "end" if "end" in cols else None
endexpr = '"end"' if endcol else "start + duration"
rows = list(db.execute("select start, " + endexpr + ", name from kernels order by start"))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
# windows split at gaps > 50 ms
wins, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-64:]) > 50e6:
        wins.append(cur)
        cur = []
    cur.append(r)
wins.append(cur)
for w in wins:
    if len(w) < 200:
        continue
    t0, t1 = w[0][0], max(x[1] for x in w)
    ev = sorted([(s, 1) for s, e, n in w] + [(e, -1) for s, e, n in w])
    busy = 0
    depth = 0
    last = t0
    area = 0
    for t, d in ev:
        if depth > 0:
            busy += t - last
        area += depth * (t - last)
        depth += d
        last = t
    names = {}
    for s, e, n in w:
        k = n.split("(")[0][-40:]
        names[k] = names.get(k, 0) + (e - s)
    top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
    print(f"window {(t1 - t0) / 1e6:9.1f} ms  kernels {len(w):6d}  busy {busy / (t1 - t0):.3f}  avg running {area / (t1 - t0):.2f}  top: " + ", ".join(f"{k} {v / 1e6:.0f} ms" for k, v in top))
