#!/bin/bash
# a longer soak and a wider fuzz sweep on the final tree (spare GPU minutes at the end of round 4)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04_long
( timeout 1000 python tools/soak.py 720 64 2>&1 | tail -6 ) > gpurun_out/r04_long/soak.txt
( INFERA_FUZZ_SEEDS=1500 timeout 1500 python -m pytest tests/test_fuzz_graphs.py -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r04_long/fuzz.txt
cat gpurun_out/r04_long/soak.txt gpurun_out/r04_long/fuzz.txt
