#!/bin/bash
# round 4, third GPU session: tests again, ThreadSanitizer (ASLR off: gcc 11's runtime cannot map its shadow under the box's mmap_rnd_bits), hardware-queue count
# against the zero-copy plateau
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04_step3
mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/pytest.txt
python -c "
from infera_amd import onnx_writer as W
W.write('/tmp/mlp128.onnx', W.mlp((128,256,64,1)))"
cat > /tmp/tsan.supp <<'S'
called_from_lib:libamdhip64.so
called_from_lib:libhsa-runtime64.so
called_from_lib:libhiprtc.so
S
export TSAN_OPTIONS="halt_on_error=0 suppressions=/tmp/tsan.supp history_size=4 report_signal_unsafe=0"
( setarch $(uname -m) -R timeout 600 tests/native/concurrency_harness_tsan tests/golden/linear.onnx 2>&1 | grep -v "^\[WARN\]" | tail -80 ) > $O/tsan_concurrency.txt
( setarch $(uname -m) -R timeout 900 tests/native/scan_stress_tsan /tmp/mlp128.onnx tests/golden/linear.onnx 3 16 2>&1 | grep -v "^\[WARN\]" | tail -250 ) > $O/tsan_scan_stress.txt
# zero-copy plateau vs the number of hardware queues the runtime creates (default 4): pulling kernel and 2-D copy
for q in 4 8 16; do
  for rect in 0 1; do
    echo "=== GPU_MAX_HW_QUEUES=$q INFERA_ZERO_COPY_RECT=$rect registered table" >> $O/hw_queues.txt
    GPU_MAX_HW_QUEUES=$q INFERA_ZERO_COPY_RECT=$rect timeout 300 python tools/host_scan_bench.py --rows 8000000 --threads 2,8,16 --reps 3 --numa auto --register 2>&1 | grep "^threads\|cpu_us" >> $O/hw_queues.txt
  done
  echo "=== GPU_MAX_HW_QUEUES=$q staged" >> $O/hw_queues.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/host_scan_bench.py --rows 8000000 --threads 8,16,24 --reps 3 --numa auto 2>&1 | grep "^threads\|cpu_us" >> $O/hw_queues.txt
done
echo done > $O/done.txt
