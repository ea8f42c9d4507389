#!/bin/bash
# Where the split-fp16 convolution spends its time (GPU box, PROBES build = libinfera_probes.so): the ResNet-18 bench under a kernel trace
# with the kernel's parts removed one at a time (INFERA_SPLIT_PROBE: 1 no operand split, 2 no gathers, 3 no weight staging / barrier,
# 4 bare matrix stream).  Results are wrong in the probe modes; only the kernel times count.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export INFERA_LIB_PATH=$GRAFT_REPO_ROOT/infera_amd/libinfera_probes.so
for p in ${*:-0 1 2 3 4}; do
  INFERA_SPLIT_PROBE=$p INFERA_PRECISION=f16x3 rocprofv3 --kernel-trace --stats -d gpurun_out/split_probe$p -o t -- python bench.py --workload resnet18 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-host-probe > gpurun_out/split_probe$p.log 2>&1
  echo "== probe $p $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/split_probe$p.log)"
  python tools/rocpd_summary.py $(find gpurun_out/split_probe$p -name "*.db") 2>&1 | sed -n 3,4p | cut -c1-60,100-150
done
