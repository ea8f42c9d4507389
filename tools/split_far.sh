#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for f in ${*:-0 1 2}; do
  export INFERA_SPLIT_FAR=$f
  echo "== far $f"
  python -m pytest tests/test_conv_split_gpu.py -x -q 2>&1 | tail -2
  INFERA_PRECISION=f16x3 rocprofv3 --kernel-trace --stats -d gpurun_out/split_far$f -o t -- python bench.py --workload resnet18 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-host-probe > gpurun_out/split_far$f.log 2>&1
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/split_far$f.log
  python tools/trace_last_step.py gpurun_out/split_far$f/t_results.db 2>&1 | grep split | tail -19 | cut -c1-100
done
