#!/usr/bin/env python3
"""HBM traffic of one whole pass of a multi-kernel plan (ResNet-18) from the FETCH_SIZE / WRITE_SIZE PMC passes.

usage: tools/traffic_pass_json.py <prof_dir> <kernel-that-runs-once-per-pass> <rows> <tag> > profiles/traffic_resnet18.json
Sums every kernel's counter over all dispatches and divides by the number of passes (= dispatches of the named
kernel).  Units and the gfx950 FETCH_SIZE correction as in traffic_json.py.
"""
import json
import re
import sqlite3
import sys


def by_kernel(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    q = ("select kernel_name, count(*), sum(v) from (select kernel_name, dispatch_id, sum(value) as v from counters_collection "
         "where counter_name=? group by kernel_name, dispatch_id) group by kernel_name")
    out = {}
    for name, n, v in cur.execute(q, (counter,)):
        m = re.search(r"(\w+<[^>]*>|\w+)\(", name)
        out[m.group(1) if m else name[:60]] = (n, v)
    return out


def main():
    d, once, rows, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, w = by_kernel(f"{d}/pmc_fetch/bench_results.db", "FETCH_SIZE"), by_kernel(f"{d}/pmc_write/bench_results.db", "WRITE_SIZE")
    pf = next(n for k, (n, _) in f.items() if once in k)
    pw = next(n for k, (n, _) in w.items() if once in k)
    skip = ("synth_fill", "copyBuffer")
    rk = {k: v * 1024 * 2 / pf for k, (n, v) in f.items() if not any(s in k for s in skip)}
    wk = {k: v * 1024 / pw for k, (n, v) in w.items() if not any(s in k for s in skip)}
    rb, wb = sum(rk.values()), sum(wk.values())
    print(json.dumps({"kernel": "whole forward (all kernels of one pass)", "passes": [pf, pw], "read_bytes": rb, "write_bytes": wb,
                      "traffic_bytes_per_launch": rb + wb, "read_bytes_by_kernel": {k: round(v) for k, v in rk.items()},
                      "write_bytes_by_kernel": {k: round(v) for k, v in wk.items()},
                      "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); summed over the kernels of a pass",
                      "rows": rows, "round": tag, "workload": "resnet18"}))


if __name__ == "__main__":
    main()
