#!/bin/bash
# per-kernel A/B of environment settings on ResNet-18 (1024 resident images): tools/ab_env.sh <tag> <kernel regex> "A=1" "A=2 B=3" ...
# two interleaved rounds; per setting the pass time from bench.py and the rocprofv3 kernel-trace lines matching the regex
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; re=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O; : > $O/ab.txt
for i in 1 2; do
  for cfg in "$@"; do
    ( cd /tmp && env $cfg timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_ab -o t -- python $OLDPWD/bench.py --workload resnet18 --steps 6 --warmup 2 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json > /tmp/line_ab.json 2>/dev/null )
    python tools/rocpd_summary.py $(find /tmp/p_ab -name "*.db") 2>/dev/null | grep -E "$re" | sed 's/void infera_hip::kern::(anonymous namespace):://; s/(float const.*float c[a-z]*//; s/infera_hip::kern::(anonymous namespace):://' | cut -c1-150 | sed "s|^|$cfg  |" >> $O/ab.txt
    python -c "
import json; d=json.loads(open('/tmp/line_ab.json').read().strip().splitlines()[-1]); print('$cfg  pass ms', d['ms_per_step'], 'frac', d['roofline']['frac'])" >> $O/ab.txt
    rm -rf /tmp/p_ab
  done
done
cat $O/ab.txt
