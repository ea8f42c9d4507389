#!/bin/bash
# ThreadSanitizer and AddressSanitizer runs of the host path on a GPU box (profiles/r04_tsan.txt, r04_asan.txt).  Build first, HERE or on the
# box: make -C tests/native tsan asan   (libinfera_{tsan,asan}.so = the library's host objects under the sanitizer + the normal gfx950 kernel
# objects; both harnesses instrumented).
# ASLR is switched off for the run: gcc 11's TSan runtime cannot map its shadow under large mmap_rnd_bits ("unexpected memory mapping").
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/tsan
mkdir -p $O
python -c "
from infera_amd import onnx_writer as W
W.write('/tmp/mlp128.onnx', W.mlp((128,256,64,1)))"
printf 'called_from_lib:libamdhip64.so\ncalled_from_lib:libhsa-runtime64.so\ncalled_from_lib:libhiprtc.so\n' > /tmp/tsan.supp
export TSAN_OPTIONS="halt_on_error=0 suppressions=/tmp/tsan.supp history_size=4 report_signal_unsafe=0"
export INFERA_ZERO_COPY_MAX_INFLIGHT=0  # every chunk of registered blocks fetched in place: the registry under maximum pressure
( setarch $(uname -m) -R timeout 600 tests/native/concurrency_harness_tsan tests/golden/linear.onnx 2>&1 | grep -v "^\[WARN\]" | tail -80 ) > $O/concurrency.txt
( setarch $(uname -m) -R timeout 900 tests/native/scan_stress_tsan /tmp/mlp128.onnx tests/golden/linear.onnx 3 16 2>&1 | grep -v "^\[WARN\]" | tail -250 ) > $O/scan_stress.txt
( tests/native/scan_stress /tmp/mlp128.onnx tests/golden/linear.onnx 3 16 2>&1 | tail -1 ) > $O/scan_stress_plain.txt
grep -c "WARNING: ThreadSanitizer" $O/concurrency.txt $O/scan_stress.txt; tail -1 $O/scan_stress.txt
# (use_sigaltstack=0: with the ROCm runtime in the process ASan's own thread teardown cannot unmap its alternate signal stack -- "failed to deallocate")
export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:use_sigaltstack=0"
( timeout 600 tests/native/concurrency_harness_asan tests/golden/linear.onnx 2>&1 | grep -v "^\[WARN\]" | tail -60 ) > $O/asan_concurrency.txt
( timeout 900 tests/native/scan_stress_asan /tmp/mlp128.onnx tests/golden/linear.onnx 3 16 2>&1 | grep -v "^\[WARN\]" | tail -120 ) > $O/asan_scan_stress.txt
grep -c "ERROR: AddressSanitizer" $O/asan_concurrency.txt $O/asan_scan_stress.txt; tail -1 $O/asan_scan_stress.txt
