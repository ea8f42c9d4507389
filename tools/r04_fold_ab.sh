#!/bin/bash
# round 4: projection shortcuts folded into the second convolution of their block -- tests + A/B (INFERA_CONV_FOLD_SHORTCUT, read when the model is scheduled)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04_fold
mkdir -p $O
( timeout 900 python -m pytest tests/test_conv_split_gpu.py tests/test_conv_ws_gpu.py tests/test_parity_gpu.py tests/test_fuzz_graphs.py -m gpu -q 2>&1 | tail -12 ) > $O/pytest.txt
for f in 0 1 0 1; do
  ( cd /tmp && INFERA_CONV_FOLD_SHORTCUT=$f timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$f -o t -- python $OLDPWD/bench.py --workload resnet18 --steps 6 --warmup 2 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json > /tmp/line_$f.json 2>/dev/null )
  python tools/rocpd_summary.py $(find /tmp/p_$f -name "*.db") 2>/dev/null | grep "split6_kernel" | sed 's/void infera_hip::kern::(anonymous namespace):://; s/(float const.*float c[a-z]*//; s/infera_hip::kern::(anonymous namespace):://' | cut -c1-150 | sed "s/^/INFERA_CONV_FOLD_SHORTCUT=$f  /" >> $O/fold_ab.txt
  python -c "
import json; d=json.loads(open('/tmp/line_$f.json').read().strip().splitlines()[-1]); print('INFERA_CONV_FOLD_SHORTCUT=$f  pass ms', d['ms_per_step'], 'frac', d['roofline']['frac'])" >> $O/fold_ab.txt
  rm -rf /tmp/p_$f
done
