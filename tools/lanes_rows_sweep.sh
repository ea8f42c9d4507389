#!/bin/bash
# two lanes vs one over batch sizes (device-resident ResNet-18): is the 512-row minimum in the right place?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_lanes_rows; mkdir -p $O; : > $O/sweep.txt
for rows in 256 384 512 640 768 1024 1536 2048; do
  for l in 1 2; do
    INFERA_CONV_LANES=$l timeout 300 python bench.py --workload resnet18 --rows $rows --steps 8 --warmup 2 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rows $rows lanes_knob $l  ms', d['ms_per_step'], ' img/s', round(d['value']))" >> $O/sweep.txt
  done
done
cat $O/sweep.txt
