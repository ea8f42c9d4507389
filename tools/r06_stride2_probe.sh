#!/bin/bash
# (round 6) what a column-phase de-interleaved input could buy the three stride-2 3x3 entry layers of ResNet-18: a PROBES build
# whose stride-2 launches gather CONSECUTIVE 16-byte pieces (wrong results, the timing of perfectly coalesced taps) against the same build
# without the probe; 1024 resident images, three interleaved rounds + one kernel trace each (per-launch durations in launch order).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp INFERA_LIB_PATH=$PWD/infera_amd/libinfera_probes.so
O=${1:-gpurun_out/r06_stride2_probe.txt}
: > $O
for i in 1 2 3; do
  for cfg in "X=0" "INFERA_CONV_PROBE_SW1=1"; do
    env $cfg INFERA_CONV_PROBE=1 python bench.py --workload resnet18 --steps 10 --warmup 3 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg  round $i  pass ms %.4f  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" >> $O
  done
done
for cfg in "X=0" "INFERA_CONV_PROBE_SW1=1"; do
  ( cd /tmp && rm -rf /tmp/p_s2 && env $cfg INFERA_CONV_PROBE=1 INFERA_CONV_LANES=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/p_s2 -o t -- python $OLDPWD/bench.py --workload resnet18 --steps 4 --warmup 2 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json > /dev/null 2>&1 )
  echo "== $cfg: kernels of the last pass in launch order (single lane)" >> $O
  python tools/trace_last_step.py $(find /tmp/p_s2 -name "*.db" | head -1) 2>/dev/null | cut -c1-160 >> $O
done
cat $O
