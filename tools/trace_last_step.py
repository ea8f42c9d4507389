#!/usr/bin/env python3
"""Print the kernels of the last bench step from a rocprofv3 rocpd .db, in launch order (name, grid, us)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, grid_x, grid_y, duration from kernels order by start"))
anchor = sys.argv[2] if len(sys.argv) > 2 else "generic"
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
start = idx[-1] if idx else 0
tot = 0.0
for name, gx, gy, dur in rows[start:]:
    m = re.search(r"(\w+<[^>]*>|\w+)\(", name)
    print(f"{(m.group(1) if m else name)[:44]:<46} {gx:>9} {gy:>3} {dur/1e3:>9.1f} us")
    tot += dur
print(f"total {tot/1e6:.3f} ms")
