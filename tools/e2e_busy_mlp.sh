#!/bin/bash
# the C2 end-to-end scan: kernel + memory-copy trace (tools/e2e_timeline.py) -- how busy is the link?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_e2e_busy
TH=${1:-16}
( cd /tmp && rm -rf /tmp/p_e2e && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/p_e2e -o t -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --no-registered --e2e-threads $TH --e2e-reps 3 --detail /tmp/d.json > /tmp/line.json 2>/dev/null )
python tools/e2e_timeline.py $(find /tmp/p_e2e -name "*.db") | tee gpurun_out/r04_e2e_busy/mlp_$TH.txt
python -c "
import json; d=json.load(open('/tmp/d.json')); e=d['end_to_end']; print('callers $TH  e2e rows/s', e['rows_per_s'])" | tee -a gpurun_out/r04_e2e_busy/mlp_$TH.txt
