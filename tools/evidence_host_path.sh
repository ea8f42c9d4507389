#!/bin/bash
# Raw logs behind every PCIe-inclusive number quoted in DESIGN.md section 6 (VERDICT r1 "keep the evidence").
# Runs on the GPU box via gpurun; outputs in gpurun_out/<tag>/, copied to profiles/<tag>_*.txt afterwards.
set -u
TAG=${1:-r02_host}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
{ echo "# nproc=$(nproc) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; lscpu | grep -E "Model name|Socket|NUMA node\(s\)"; } > "$OUT/host.txt"
python tools/host_scan_bench.py --rows 10000000 --threads 1,2,4,8,12,16,24,32,48,64 > "$OUT/thread_sweep_c2.txt" 2>&1
INFERA_HIPGRAPH=1 python tools/host_scan_bench.py --rows 10000000 --threads 8,16,32 > "$OUT/thread_sweep_c2_hipgraph.txt" 2>&1
python tools/host_scan_bench.py --rows 10000000 --threads 8,16,32 --workload logreg > "$OUT/thread_sweep_c4.txt" 2>&1
python tools/call_latency.py > "$OUT/call_latency.txt" 2>&1
python tools/blob_scan_bench.py --batch 256 --calls 4 --threads 1,8,16 > "$OUT/blob_scan_c5.txt" 2>&1
# one traced scan: HIP API calls, memory copies and kernels per stream (no counters in this run)
rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --stats --output-format csv -d "$OUT/trace" -o scan -- \
    python tools/host_scan_bench.py --rows 4000000 --threads 16 --reps 2 > "$OUT/traced_scan.txt" 2>&1
find "$OUT/trace" -name "*stats*.csv" | sort
for f in $(find "$OUT/trace" -name "*_stats.csv" | sort); do echo "== $f"; head -25 "$f"; done > "$OUT/trace_stats.txt"
tail -3 "$OUT/thread_sweep_c2.txt"
