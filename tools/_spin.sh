cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_spin_wait.txt; : > $O
for mode in "INFERA_EXP_SPIN_WAIT=0 INFERA_STREAM_MAX_INFLIGHT=0" "INFERA_EXP_SPIN_WAIT=1 INFERA_STREAM_MAX_INFLIGHT=0" "INFERA_EXP_SPIN_WAIT=1 INFERA_STREAM_MAX_INFLIGHT=4" "INFERA_EXP_SPIN_WAIT=0 INFERA_STREAM_MAX_INFLIGHT=4"; do
echo "=== $mode" >> $O
env $mode python tools/host_scan_bench.py --rows 6000000 --threads 1,2,4,8 --reps 3 --numa auto 2>&1 | grep "^threads\|us/chunk" >> $O
done
cat $O
