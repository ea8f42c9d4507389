#!/bin/bash
# end-of-round soak (64 caller threads, mixed models incl. zero-copy chunks and load/unload cycles) and the extended graph fuzz on the final tree
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04_soak
( timeout 400 python tools/soak.py 180 64 2>&1 | tail -6 ) > gpurun_out/r04_soak/soak.txt
( INFERA_FUZZ_SEEDS=400 timeout 900 python -m pytest tests/test_fuzz_graphs.py -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r04_soak/fuzz.txt
