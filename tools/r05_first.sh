#!/bin/bash
# Round-5 first GPU pass on the restored tree: full -m gpu suite, the driver-shaped default bench line, kernel trace of the same command.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_first
( timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r05_first/pytest.txt
( timeout 900 python bench.py --detail gpurun_out/r05_first/line_mlp_detail.json > gpurun_out/r05_first/line_mlp.json 2> gpurun_out/r05_first/line_mlp.err; echo "rc=$? bytes=$(wc -c < gpurun_out/r05_first/line_mlp.json)" ) > gpurun_out/r05_first/bench_rc.txt 2>&1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/r05_first/trace -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads > gpurun_out/r05_first/bench_trace.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r05_first/trace -name "*.db") 2>/dev/null | head -60 > gpurun_out/r05_first/mlp_bench.txt
cat gpurun_out/r05_first/pytest.txt gpurun_out/r05_first/bench_rc.txt; head -c 1500 gpurun_out/r05_first/line_mlp.json
