#!/bin/bash
# Does the last partial round of workgroups cost what the round arithmetic says?  ResNet-18 at 1003 images (every split6 launch a whole number of
# rounds on 256 CUs) against 1024 (12.25 / 6.125 / 3.06 rounds): per-kernel times from rocprofv3 kernel traces.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04_tail; mkdir -p $O; : > $O/tail.txt
for rows in 1024 1003 1024 1003; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_t -o t -- python $OLDPWD/bench.py --workload resnet18 --rows $rows --steps 6 --warmup 2 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json > /tmp/line_t.json 2>/dev/null )
  echo "== rows $rows" >> $O/tail.txt
  python tools/trace_last_step.py $(find /tmp/p_t -name "*.db") stem 2>/dev/null | grep -E "split6|stem" | tail -17 | cut -c1-120 >> $O/tail.txt
  python -c "
import json; d=json.loads(open('/tmp/line_t.json').read().strip().splitlines()[-1]); print('rows $rows  pass ms', d['ms_per_step'], ' us per image', 1e3*d['ms_per_step']/$rows)" >> $O/tail.txt
  rm -rf /tmp/p_t
done
cat $O/tail.txt
