#!/bin/bash
# Stem + pool kernel A/B on ResNet-18 (1024 resident images): one-workgroup kernel vs two half-channel workgroups per CU.
tag=${1:-r03}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_stem_ab.txt
{
  python -m pytest tests/test_stem_pool_gpu.py -m gpu -x -q 2>&1 | tail -4
  for rep in 1 2; do
  for cfg in "INFERA_STEM_POOL2=0" "INFERA_STEM_POOL2=1 INFERA_STEM_POOL2_DESYNC=0" "INFERA_STEM_POOL2=1 INFERA_STEM_POOL2_DESYNC=1" "INFERA_STEM_POOL2=1 INFERA_STEM_POOL2_DESYNC=2"; do
    echo "== $cfg"
    env $cfg python bench.py --workload resnet18 --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('img/s', round(d['value']), 'ms', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))"
  done
  done
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  for cfg in "INFERA_STEM_POOL2=0" "INFERA_STEM_POOL2=1"; do
    O=gpurun_out/${tag}_stem_trace_${cfg##*=}
    env $cfg rocprofv3 --kernel-trace --stats -d $O -o bench -- python bench.py --workload resnet18 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $O.log 2>&1
    echo "== trace $cfg"; python tools/rocpd_summary.py $(find $O -name "*.db" | head -1) 2>/dev/null | grep -E "patch|stem" | head -5
  done
} > $out 2>&1
cat $out | tail -40
