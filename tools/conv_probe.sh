#!/bin/bash
# Timing probes for conv2d_tiled_kernel (PROBES=1 build only; results are wrong by construction).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/probe
for p in ${*:-0 1 2 3 4}; do
  INFERA_CONV_PROBE=$p timeout 200 rocprofv3 --kernel-trace -d gpurun_out/probe/p$p -o b -- python bench.py --workload resnet18 --rows 1024 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/probe/p$p.log 2>&1
  echo "probe $p rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/probe/p$p.log)"
done
