#!/bin/bash
# Streamed chunks against staged chunks on one GPU (profiles/r05_stream_chunk_ab.txt): C2 scan (10M x 128 FLOAT host table, 2048-row chunks through
# the SQL surface) at 1 / 2 / 4 / 8 / 16 callers with INFERA_STREAM_MAX_INFLIGHT = 0 (every chunk staged: round 4's path), 2, 4 (default), 8 --
# rows/s and CPU us per chunk (VERDICT r4 item 3's yardsticks: >= 55 M rows/s at 2 callers, <= 75 us of CPU per chunk).
#   usage: gpurun -- bash tools/stream_chunk_ab.sh [tag]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-r05}
O=gpurun_out/${tag}_stream_chunk_ab.txt
: > $O
python -m pytest tests/test_stream_chunks_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $O
for lim in 0 4 2 8 0 4; do
  echo "=== INFERA_STREAM_MAX_INFLIGHT=$lim" >> $O
  INFERA_STREAM_MAX_INFLIGHT=$lim python tools/host_scan_bench.py --rows 8000000 --threads 1,2,4,8,16 --reps 3 --numa auto 2>&1 | grep "^threads\|us/chunk" >> $O
  python - >> $O <<'P'
P
done
echo "=== DOUBLE columns, limit 0 then 4" >> $O
for lim in 0 4; do
  INFERA_STREAM_MAX_INFLIGHT=$lim python tools/host_scan_bench.py --rows 4000000 --threads 1,2,4 --reps 3 --numa auto --double 2>&1 | grep "^threads\|us/chunk" >> $O
done
cat $O
