#!/bin/bash
# Round-5 end-of-work evidence: rocprofv3 kernel trace + PMC passes of the C5 / C2 / C4 benches, summaries as text.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_lines
bash tools/profile_bench.sh r05_final_resnet18 --workload resnet18 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r05_lines/prof_resnet.log 2>&1
bash tools/profile_bench.sh r05_final_mlp --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads > gpurun_out/r05_lines/prof_mlp.log 2>&1
bash tools/profile_bench.sh r05_final_logreg --workload logreg --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r05_lines/prof_logreg.log 2>&1
for w in resnet18 mlp logreg; do
  d=gpurun_out/r05_final_$w
  python tools/rocpd_summary.py $(find $d/trace -name "*.db") 2>/dev/null | head -60 > gpurun_out/r05_final_${w}_bench.txt
  cp $d/bench_line.json gpurun_out/r05_final_${w}_bench_line.json
  python tools/pmc_table.py $(find $d/pmc_sq $d/pmc_grbm -name "*.db") > gpurun_out/r05_final_${w}_pmc_table.txt 2>&1
done
python tools/traffic_json.py gpurun_out/r05_final_mlp mlp3_split 10000000 mlp "r05 final" > gpurun_out/traffic_mlp.json
python tools/traffic_json.py gpurun_out/r05_final_logreg dense_narrow16 50000000 logreg "r05 final" > gpurun_out/traffic_logreg.json
python tools/traffic_pass_json.py gpurun_out/r05_final_resnet18 global_avgpool 1024 "r05 final" > gpurun_out/traffic_resnet18.json
ls -la gpurun_out/ | tail -20
find gpurun_out/r05_final_mlp -name "*.db" | head
# the driver-shaped default run (compact line + detail) and the full GPU suite on the final tree
( timeout 900 python bench.py --steps 20 --warmup 5 --detail gpurun_out/r05_line_mlp_detail.json > gpurun_out/r05_line_mlp.json 2> /dev/null; echo "rc=$? bytes=$(wc -c < gpurun_out/r05_line_mlp.json)" ) > gpurun_out/r05_lines/bench_rc.txt 2>&1
( timeout 600 python bench.py --workload resnet18 --steps 10 --warmup 3 --detail gpurun_out/r05_line_resnet18_detail.json > gpurun_out/r05_line_resnet18.json 2> /dev/null )
( timeout 600 python bench.py --workload logreg --steps 20 --warmup 3 --detail gpurun_out/r05_line_logreg_detail.json > gpurun_out/r05_line_logreg.json 2> /dev/null )
python tools/trace_last_step.py $(find gpurun_out/r05_final_resnet18/trace -name "*.db") > gpurun_out/r05_final_resnet18_last_pass.txt 2>&1
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r05_lines/pytest.txt
# round 5: the per-chunk timelines of the host path on the final tree (roctx ranges; tools/e2e_ranges.py)
bash tools/r05_ranges.sh > /dev/null 2>&1
cp gpurun_out/r05_ranges/all.txt gpurun_out/r05_final_e2e_ranges.txt
