#!/bin/bash
# Quick A/B loop for the split-fp16 convolution (runs on the GPU box via gpurun): parity tests, then the ResNet-18 bench under a kernel
# trace; prints the bench's ms per 1024 images and the last pass's kernels in order.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-split_q}
python -m pytest tests/test_conv_split_gpu.py -x -q 2>&1 | tail -4
INFERA_PRECISION=f16x3 rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG -o t -- python bench.py --workload resnet18 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads --no-host-probe > gpurun_out/$TAG.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/$TAG.log
python tools/trace_last_step.py $(find gpurun_out/$TAG -name "*.db") 2>&1 | grep -v copyBuffer | tail -26 | cut -c1-100
