#!/bin/bash
# (A/B-only override INFERA_LANE_MIN_ROWS) two lanes also for the short passes of the host path?  C5 end to end + single-caller calls of 256 images
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_lanemin; mkdir -p $O; : > $O/ab.txt
for rep in 1 2; do
for mr in 512 128; do
  for th in 1 4 16; do
    INFERA_LANE_MIN_ROWS=$mr timeout 600 python bench.py --workload resnet18 --steps 4 --warmup 2 --no-cpu-baseline --e2e-threads $th --e2e-reps 3 --detail /tmp/d.json > /tmp/line.json 2>/dev/null
    python - $mr $th >> $O/ab.txt <<'P'
import json,sys
d=json.load(open('/tmp/d.json')); e=d.get('end_to_end') or {}
print('lane_min_rows', sys.argv[1], 'callers', sys.argv[2], ' e2e img/s', round(e.get('rows_per_s',0)), ' resident img/s', round(d['value']))
P
  done
done
done
cat $O/ab.txt
