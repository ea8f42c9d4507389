#!/bin/bash
# C5 end to end (BLOB chunks through the SQL surface): pass size of the big-row pipeline x contexts per GPU x callers
# (INFERA_BLOB_PASS_ROWS: an A/B-only override of the 256-row cap, patched in for the measurement -- profiles/r04_blob_pass_ab.txt)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_blobpass; mkdir -p $O; : > $O/ab.txt
for rep in 1 2; do
for cfg in "INFERA_HOST_CONTEXTS=24 INFERA_BLOB_PASS_ROWS=256" "INFERA_HOST_CONTEXTS=24 INFERA_BLOB_PASS_ROWS=128" "INFERA_HOST_CONTEXTS=12 INFERA_BLOB_PASS_ROWS=512" "INFERA_HOST_CONTEXTS=8 INFERA_BLOB_PASS_ROWS=683"; do
  for th in 4 16; do
    env $cfg timeout 600 python bench.py --workload resnet18 --steps 3 --warmup 1 --no-cpu-baseline --e2e-threads $th --e2e-reps 3 --detail /tmp/d.json > /tmp/line.json 2>/dev/null
    python - "$cfg" $th >> $O/ab.txt <<'P'
import json,sys
d=json.load(open('/tmp/d.json')); e=d.get('end_to_end') or {}
print(sys.argv[1], 'callers', sys.argv[2], ' e2e img/s', round(e.get('rows_per_s',0)), ' resident img/s', round(d['value']))
P
  done
done
done
cat $O/ab.txt
