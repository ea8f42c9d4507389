cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_stream_queues.txt; : > $O
python -m pytest tests/test_stream_chunks_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $O
for q in 4 8 16; do for lim in 4 0; do
echo "=== GPU_MAX_HW_QUEUES=$q INFERA_STREAM_MAX_INFLIGHT=$lim" >> $O
GPU_MAX_HW_QUEUES=$q INFERA_STREAM_MAX_INFLIGHT=$lim python tools/host_scan_bench.py --rows 6000000 --threads 1,2,2,4,8,16 --reps 3 --numa auto 2>&1 | grep "^threads\|us/chunk" >> $O
done; done
cat $O
