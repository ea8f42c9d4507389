#!/bin/bash
# round 4: the split stem's LDS row pitch (8-byte patch words on ds_read2_b64's 32 banks): A/B + the tests that pin the stem
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04_step5
mkdir -p $O
( timeout 900 python -m pytest tests/test_conv_split_gpu.py tests/test_stem_pool_gpu.py tests/test_parity_gpu.py -m gpu -q -k "stem or c5 or C5 or resnet or bf16x6" 2>&1 | tail -8 ) > $O/pytest.txt
for pad in 0 1 0 1; do
  ( cd /tmp && INFERA_STEM_ROWS_PAD=$pad timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$pad -o t -- python $OLDPWD/bench.py --workload resnet18 --steps 6 --warmup 2 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json > /tmp/line_$pad.json 2>/dev/null )
  python tools/rocpd_summary.py $(find /tmp/p_$pad -name "*.db") 2>/dev/null | grep "stem_split6\|^kernel" | sed "s/^/INFERA_STEM_ROWS_PAD=$pad  /" >> $O/stem_pitch_ab.txt
  python -c "
import json; d=json.loads(open('/tmp/line_$pad.json').read().strip().splitlines()[-1]); print('INFERA_STEM_ROWS_PAD=$pad  pass ms', d['ms_per_step'], 'frac', d['roofline']['frac'])" >> $O/stem_pitch_ab.txt
  rm -rf /tmp/p_$pad
done
echo done > $O/done.txt
