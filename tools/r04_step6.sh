#!/bin/bash
# round 4: 64-feature split convolutions with two pixel tiles per wave -- tests + A/B (rocprofv3 kernel trace of the C5 pass, both forms)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04_step6
mkdir -p $O
( timeout 900 python -m pytest tests/test_conv_split_gpu.py tests/test_conv_ws_gpu.py tests/test_parity_gpu.py -m gpu -q 2>&1 | tail -8 ) > $O/pytest.txt
for pt in 1 2 1 2; do
  ( cd /tmp && INFERA_SPLIT6_PT=$pt timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$pt -o t -- python $OLDPWD/bench.py --workload resnet18 --steps 6 --warmup 2 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json > /tmp/line_$pt.json 2>/dev/null )
  python tools/rocpd_summary.py $(find /tmp/p_$pt -name "*.db") 2>/dev/null | grep "split6_kernel" | sed 's/void infera_hip::kern::(anonymous namespace):://; s/(float const.*float c[a-z]*//' | sed "s/^/INFERA_SPLIT6_PT=$pt  /" >> $O/split6_pt_ab.txt
  python -c "
import json; d=json.loads(open('/tmp/line_$pt.json').read().strip().splitlines()[-1]); print('INFERA_SPLIT6_PT=$pt  pass ms', d['ms_per_step'], 'frac', d['roofline']['frac'])" >> $O/split6_pt_ab.txt
  rm -rf /tmp/p_$pt
done
echo done > $O/done.txt
