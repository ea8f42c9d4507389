#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernel from the FETCH_SIZE / WRITE_SIZE PMC passes.

usage: tools/traffic_json.py <prof_dir> <kernel-substring> [<rows> <workload> <round tag>] > profiles/traffic_<workload>.json
(rows / workload / round are what bench.py and tests/test_traffic_profiles.py match the file against)
FETCH_SIZE / WRITE_SIZE are in KiB... (rocprofv3 derives them as bytes/1024); on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM
section: 128-B requests tallied at 64 B) -> the read side is doubled.  WRITE_SIZE is taken as is (the
kernel's own store volume calibrates it: it matches rows*out_cols*4 exactly).
"""
import json
import sqlite3
import sys


def per_dispatch(db_path, counter, kernel_sub):
    cur = sqlite3.connect(db_path).cursor()
    q = ("select avg(v), count(*) from (select dispatch_id, sum(value) as v from counters_collection "
         "where counter_name=? and kernel_name like ? group by dispatch_id)")
    return cur.execute(q, (counter, f"%{kernel_sub}%")).fetchone()


def main():
    d, sub = sys.argv[1], sys.argv[2]
    f, nf = per_dispatch(f"{d}/pmc_fetch/bench_results.db", "FETCH_SIZE", sub)
    w, nw = per_dispatch(f"{d}/pmc_write/bench_results.db", "WRITE_SIZE", sub)
    read_bytes = f * 1024 * 2  # gfx950 correction
    write_bytes = w * 1024
    extra = {}
    if len(sys.argv) >= 6:
        extra = {"rows": int(sys.argv[3]), "workload": sys.argv[4], "round": sys.argv[5]}
    print(json.dumps({**extra, "kernel": sub, "dispatches": [nf, nw], "fetch_size_kib_raw": f, "write_size_kib_raw": w,
                      "read_bytes": read_bytes, "write_bytes": write_bytes, "traffic_bytes_per_launch": read_bytes + write_bytes,
                      "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)"}))


if __name__ == "__main__":
    main()
