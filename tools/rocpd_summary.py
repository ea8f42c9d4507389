#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) output as text: per-kernel stats and per-kernel PMC sums.

usage: tools/rocpd_summary.py <results.db> [...]   (stdout is what gets committed under profiles/)
"""
import sqlite3
import sys


def cols(cur, table):
    return [r[1] for r in cur.execute(f"pragma table_info('{table}')")]


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"== {path}")
        print(f"{'kernel':<100} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  grid wg lds vgpr agpr")
        q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), max(workgroup_x), "
             "max(lds_size), max(vgpr_count), max(accum_vgpr_count) from kernels group by name order by sum(duration) desc")
        for n, c, tot, avg, mn, mx, g, wg, lds, v, a in cur.execute(q):
            print(f"{n[:100]:<100} {c:>6} {tot/1e6:>10.3f} {avg/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f}  {g} {wg} {lds} {v} {a}")
        try:
            n = list(cur.execute("select count(*) from counters_collection"))[0][0]
            if n:
                print("-- PMC per kernel (value summed over XCDs/SEs by rocprofv3; avg / min / max over dispatches of that kernel)")
                print(f"{'kernel':<60} {'counter':<30} {'disp':>5} {'avg':>18} {'min':>18} {'max':>18}")
                q2 = ("select kernel_name, counter_name, count(*), avg(v), min(v), max(v) from "
                      "(select kernel_name, counter_name, dispatch_id, sum(value) as v from counters_collection "
                      " group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name")
                for k, c, d, a, mn, mx in cur.execute(q2):
                    short = k.replace("infera_hip::kern::(anonymous namespace)::", "")
                    print(f"{short[:60]:<60} {c:<30} {d:>5} {a:>18.1f} {mn:>18.1f} {mx:>18.1f}")
        except sqlite3.Error as e:
            print("pmc query failed:", e)
        print()


if __name__ == "__main__":
    main()
