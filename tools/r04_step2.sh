#!/bin/bash
# round 4, second GPU session: full parity tests, ThreadSanitizer runs of the native harnesses, zero-copy rectangle copy A/B, stage-order A/B of the split convolution
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04_step2
mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/pytest.txt
python -c "
from infera_amd import onnx_writer as W
W.write('/tmp/mlp128.onnx', W.mlp((128,256,64,1)))"
# ThreadSanitizer: host objects of the library + the harness instrumented; the HIP runtime is not (its internals are suppressed by module)
cat > /tmp/tsan.supp <<'S'
called_from_lib:libamdhip64.so
called_from_lib:libhsa-runtime64.so
called_from_lib:libhiprtc.so
S
( TSAN_OPTIONS="halt_on_error=0 suppressions=/tmp/tsan.supp history_size=4" timeout 600 tests/native/concurrency_harness_tsan tests/golden/linear.onnx 2>&1 | grep -v "^\[WARN\]" | tail -60 ) > $O/tsan_concurrency.txt
( TSAN_OPTIONS="halt_on_error=0 suppressions=/tmp/tsan.supp history_size=4" timeout 900 tests/native/scan_stress_tsan /tmp/mlp128.onnx tests/golden/linear.onnx 3 16 2>&1 | grep -v "^\[WARN\]" | tail -150 ) > $O/tsan_scan_stress.txt
( tests/native/scan_stress /tmp/mlp128.onnx tests/golden/linear.onnx 3 16 2>&1 | tail -3 ) > $O/scan_stress.txt
# zero-copy: one 2-D copy per chunk on the copy engines (INFERA_ZERO_COPY_RECT=1) vs the pulling kernel (0), and the staged path beside them
for rect in 1 0; do
  echo "=== registered table, INFERA_ZERO_COPY_RECT=$rect" >> $O/zero_copy_rect_ab.txt
  INFERA_ZERO_COPY_RECT=$rect timeout 300 python tools/host_scan_bench.py --rows 8000000 --threads 1,2,4,8,16,24 --reps 3 --numa auto --register 2>&1 | grep -v "^devices=" >> $O/zero_copy_rect_ab.txt
done
echo "=== staged path (table not registered)" >> $O/zero_copy_rect_ab.txt
timeout 300 python tools/host_scan_bench.py --rows 8000000 --threads 1,2,4,8,16,24 --reps 3 --numa auto 2>&1 | grep -v "^devices=" >> $O/zero_copy_rect_ab.txt
# split convolution: channel blocks of one chunk (taps of ONE chunk in consecutive stages) vs two
for sb in 0 1 0 1; do
  ( INFERA_SPLIT6_SB=$sb timeout 300 python bench.py --workload resnet18 --steps 10 --warmup 3 --no-end-to-end --no-cpu-baseline --detail /tmp/d.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('INFERA_SPLIT6_SB=$sb', d['ms_per_step'], d['roofline']['frac'])" ) >> $O/split6_sb_ab.txt
done
( timeout 600 python bench.py --steps 20 --warmup 5 --detail $O/bench_detail.json > $O/bench_line.json 2> /dev/null; echo "rc=$? bytes=$(wc -c < $O/bench_line.json)" ) > $O/bench_rc.txt 2>&1
echo done > $O/done.txt
