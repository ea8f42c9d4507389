#!/usr/bin/env python3
"""From a rocprofv3 rocpd .db with kernel and memory-copy traces: inside the longest busy window, the share of wall time with (a) a kernel
running, (b) a copy running, (c) neither; the longest kernel-idle gaps and whether copies covered them."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
def cols(t): return [r[1] for r in db.execute(f"pragma table_info({t})")]
k = list(db.execute('select start, "end" from kernels order by start'))
ct = [t for t in tabs if "memory_cop" in t.lower() and "rocpd" not in t.lower()] or [t for t in tabs if "memory_cop" in t.lower()]
print("copy tables:", ct[:3])
c = []
if ct:
    cc = cols(ct[0])
    print(cc)
    c = list(db.execute(f'select start, "end" from {ct[0]} order by start'))
# window: last 80 % of kernels' span (the scan), skipping warm-up at the front
t0 = k[len(k) // 5][0]
t1 = max(e for s, e in k)
def union(iv):
    out = []
    for s, e in sorted(iv):
        if e <= t0 or s >= t1: continue
        s, e = max(s, t0), min(e, t1)
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
uk, uc = union(k), union(c)
lk = sum(e - s for s, e in uk); lc = sum(e - s for s, e in uc)
both = union(k + c); lb = sum(e - s for s, e in both)
W = t1 - t0
print(f"window {W / 1e6:.1f} ms: kernel running {lk / W:.3f}, copy running {lc / W:.3f}, kernel or copy {lb / W:.3f}, neither {1 - lb / W:.3f}")
gaps = sorted(((uk[i + 1][0] - uk[i][1], uk[i][1]) for i in range(len(uk) - 1)), reverse=True)
tot_gap = sum(g for g, _ in gaps)
print(f"kernel-idle gaps: {len(gaps)}, total {tot_gap / 1e6:.1f} ms; > 100 us: {sum(1 for g, _ in gaps if g > 1e5)} ({sum(g for g, _ in gaps if g > 1e5) / 1e6:.1f} ms); 10-100 us: {sum(1 for g, _ in gaps if 1e4 < g <= 1e5)} ({sum(g for g, _ in gaps if 1e4 < g <= 1e5) / 1e6:.1f} ms); < 10 us: {sum(1 for g, _ in gaps if g <= 1e4)} ({sum(g for g, _ in gaps if g <= 1e4) / 1e6:.1f} ms)")
