#!/usr/bin/env python3
"""Per-kernel-name table from rocprofv3 PMC .db files: avg duration and counters (summed over XCDs/SEs)."""
import re
import sqlite3
import sys
from collections import defaultdict

rows = defaultdict(dict)
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    for name, n, avg in db.execute("select name, count(*), avg(duration) from kernels group by name"):
        rows[name]["calls"] = n
        rows[name]["us"] = avg / 1e3
    q = ("select kernel_name, counter_name, avg(v) from (select kernel_name, counter_name, dispatch_id, sum(value) as v "
         "from counters_collection group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name")
    for k, c, v in db.execute(q):
        rows[k][c] = v
for name, d in sorted(rows.items(), key=lambda kv: -kv[1].get("us", 0) * kv[1].get("calls", 0)):
    m = re.search(r"(\w+<[^>]*>|\w+)\(", name)
    short = m.group(1) if m else name[:50]
    line = f"{short:<40} calls {d.get('calls', 0):>4} avg {d.get('us', 0):>9.1f} us"
    if "GRBM_GUI_ACTIVE" in d and d.get("us"):
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        line += f"  clk {cyc / d['us'] / 1e3:.2f} GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            line += f"  mfma_busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}"
    for c in sorted(d):
        if c not in ("calls", "us"):
            line += f"\n      {c:<34} {d[c]:.4g}"
    print(line)
