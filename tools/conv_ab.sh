#!/bin/bash
# ResNet-18 (C5) forward on 1024 resident images: bench line + per-kernel trace, for each value of an env knob.
# usage (GPU box): bash tools/conv_ab.sh <tag> <ENV_NAME> <v1> [v2 ...]
TAG=$1; KNOB=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
for v in "$@"; do
  O=gpurun_out/$TAG/${KNOB}_$v
  env $KNOB=$v rocprofv3 --kernel-trace --stats -d $O -o bench -- python bench.py --workload resnet18 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $O.log 2>&1
  echo "== $KNOB=$v"; grep '^{' $O.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('img/s', round(d['value']), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3))"
  python tools/trace_last_step.py $(find $O -name "*.db" | head -1) 2>/dev/null | tail -28
done
