#!/usr/bin/env python3
"""(round 6) makes a bench line's roofline fraction reproducible from the tracked profile ALONE: for the dominant kernel of a
rocprofv3 --kernel-trace .db -- launches, median, the WARM average (launches slower than 1.1 x the median are first-touch / clock ramp: dropped
and counted), achieved = algorithmic work per launch / warm average, fraction of the peak -- beside the sustained clock of the same box from its
GRBM_GUI_ACTIVE pass and the hipEvent `kernel_ms` / `frac` of an UNPROFILED bench.py run on the same box in the same gpurun call.

usage: warm_summary.py <trace.db> <kernel substring> <work per launch> <peak> <unit: TFLOP/s|GB/s> [--grbm <pmc_grbm.db>] [--line <unprofiled line.json>]"""
import json
import sqlite3
import statistics
import sys

args = sys.argv[1:]
trace, sub, work, peak, unit = args[0], args[1], float(args[2]), float(args[3]), args[4]
grbm = args[args.index("--grbm") + 1] if "--grbm" in args else None
line = args[args.index("--line") + 1] if "--line" in args else None
scale = 1e12 if unit.startswith("TFLOP") else 1e9
db = sqlite3.connect(trace)
d = [r[0] / 1e3 for r in db.execute("select duration from kernels where name like ? order by start", (f"%{sub}%",))]
if not d:
    sys.exit(f"no kernel matching {sub!r} in {trace}")
med = statistics.median(d)
warm = [x for x in d if x <= 1.1 * med]
avg_all, avg_warm = sum(d) / len(d), sum(warm) / len(warm)
print(f"kernel *{sub}*: {len(d)} launches under rocprofv3 --kernel-trace; all: avg {avg_all:.1f} us, min {min(d):.1f}, median {med:.1f}, max {max(d):.1f}")
print(f"  warm launches (<= 1.1 x median): {len(warm)} of {len(d)} ({len(d) - len(warm)} dropped: {', '.join(f'{x:.0f}' for x in d if x > 1.1 * med) or 'none'} us), warm avg {avg_warm:.1f} us")
ach_all, ach_warm = work / (avg_all * 1e-6) / scale, work / (avg_warm * 1e-6) / scale
print(f"  work per launch {work:.6g} -> achieved {ach_warm:.2f} {unit} = {ach_warm / peak:.4f} of {peak:g} (warm avg);  over ALL launches {ach_all:.2f} = {ach_all / peak:.4f};  fastest launch {work / (min(d) * 1e-6) / scale / peak:.4f}")
if grbm:
    g = sqlite3.connect(grbm)
    q = ("select avg(k.duration), avg(c.v) from kernels k join (select dispatch_id, sum(value) as v from counters_collection where counter_name = 'GRBM_GUI_ACTIVE' "
         "group by dispatch_id) c on c.dispatch_id = k.dispatch_id where k.name like ?")
    try:
        dur, cyc = list(g.execute(q, (f"%{sub}%",)))[0]
        if dur and cyc:
            print(f"  sustained clock under this kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration, counter pass of the same box): {cyc / 8 / dur:.3f} GHz")
    except sqlite3.Error as e:
        try:  # older rocpd layout: kernel_name in counters_collection
            rows = list(g.execute("select avg(v) from (select dispatch_id, sum(value) as v from counters_collection where counter_name = 'GRBM_GUI_ACTIVE' and kernel_name like ? group by dispatch_id)", (f"%{sub}%",)))
            dur = list(g.execute("select avg(duration) from kernels where name like ?", (f"%{sub}%",)))[0][0]
            if rows and rows[0][0] and dur:
                print(f"  sustained clock under this kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration, counter pass of the same box): {rows[0][0] / 8 / dur:.3f} GHz")
        except sqlite3.Error as e2:
            print(f"  (clock: {e}; {e2})")
if line:
    l = json.loads([x for x in open(line).read().splitlines() if x.startswith("{")][-1])
    r = l["roofline"]
    print(f"  UNPROFILED bench.py on the same box, same call: kernel_ms {r['kernel_ms']:.4f} (HIP events on the launching stream), ms_per_step {l['ms_per_step']:.4f}, "
          f"achieved {r['achieved']:.2f} {r['unit']}, frac {r['frac']:.4f}")
    print(f"  => profile-alone fraction (warm avg) {ach_warm / peak:.4f} vs the line's {r['frac']:.4f}: {100 * (ach_warm / peak / r['frac'] - 1):+.2f} %")
