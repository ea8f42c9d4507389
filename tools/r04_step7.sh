#!/bin/bash
# round 4: tap-triple stages for three-column filters (the three kx taps of a filter row in ONE stage: two of three gathers served by the L1) -- tests + A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04_step7
mkdir -p $O
( timeout 900 python -m pytest tests/test_conv_split_gpu.py tests/test_conv_ws_gpu.py tests/test_parity_gpu.py tests/test_fuzz_graphs.py -m gpu -q 2>&1 | tail -12 ) > $O/pytest.txt
for tt in 0 1 0 1; do
  ( cd /tmp && INFERA_SPLIT6_TT=$tt timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$tt -o t -- python $OLDPWD/bench.py --workload resnet18 --steps 6 --warmup 2 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json > /tmp/line_$tt.json 2>/dev/null )
  python tools/rocpd_summary.py $(find /tmp/p_$tt -name "*.db") 2>/dev/null | grep "split6_kernel" | sed 's/void infera_hip::kern::(anonymous namespace):://; s/(float const.*float c[a-z]*//; s/infera_hip::kern::(anonymous namespace):://' | cut -c1-150 | sed "s/^/INFERA_SPLIT6_TT=$tt  /" >> $O/split6_tt_ab.txt
  python -c "
import json; d=json.loads(open('/tmp/line_$tt.json').read().strip().splitlines()[-1]); print('INFERA_SPLIT6_TT=$tt  pass ms', d['ms_per_step'], 'frac', d['roofline']['frac'])" >> $O/split6_tt_ab.txt
  if [ $tt = 1 ]; then python tools/trace_last_step.py $(find /tmp/p_$tt -name "*.db") > $O/last_pass_tt1.txt 2>&1; fi
  rm -rf /tmp/p_$tt
done
echo done > $O/done.txt
