#!/bin/bash
# round 4, first GPU session: parity tests, the compact bench line, C5 kernel trace of the pre-split plan, hipGraph A/B on CPU time per chunk
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04_step1
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > $O/pytest.txt
( timeout 600 python bench.py --steps 20 --warmup 5 --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench_stderr.txt; echo "rc=$? bytes=$(wc -c < $O/bench_line.json)" ) > $O/bench_rc.txt 2>&1
for p in default fp32; do
  for ps in 1 0; do
    [ "$p" = fp32 ] && [ "$ps" = 0 ] && continue
    ( INFERA_CONV_PRESPLIT=$ps timeout 300 python bench.py --workload resnet18 --precision $p --steps 10 --warmup 3 --no-end-to-end --no-cpu-baseline --detail $O/rn_${p}_${ps}.json 2>/dev/null ) > $O/rn_line_${p}_${ps}.json
  done
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_rn -o rn -- python $OLDPWD/bench.py --workload resnet18 --steps 10 --warmup 2 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json > /dev/null 2>&1 )
find $O/prof_rn -name "*kernel_stats*" | head -1 | xargs -r head -30 > $O/rn_kernel_stats.txt
# hipGraph A/B: CPU per chunk, rows/s at 4/8/16 callers x 1/2/4 slots
for hg in 0 1; do
  for slots in 1 2 4; do
    dev=$(python -c "print(','.join(['0']*$slots))")
    echo "=== INFERA_HIPGRAPH=$hg slots=$slots" >> $O/hipgraph_ab.txt
    INFERA_HIPGRAPH=$hg INFERA_DEVICES=$dev timeout 300 python tools/host_scan_bench.py --rows 6000000 --threads 4,8,16 --reps 3 --numa auto 2>&1 | grep -v "^devices=" >> $O/hipgraph_ab.txt
  done
done
echo done > $O/done.txt
