#!/bin/bash
# Round-6 end-of-work evidence on ONE box in ONE gpurun call: rocprofv3 kernel trace + PMC passes of the C2 / C4 / C5 benches, the unprofiled
# bench lines of the same box, and per workload the summary from which the line's roofline fraction follows from the profile alone
# (tools/warm_summary.py: warm-launch average, sustained clock, the unprofiled kernel_ms beside it) -- VERDICT r5 item 6.
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_final; mkdir -p $O
# unprofiled lines first (the box's own numbers), then the profiled passes
( timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/line_mlp_detail.json > $O/line_mlp.json 2> /dev/null; echo "rc=$? bytes=$(wc -c < $O/line_mlp.json)" ) > $O/bench_rc.txt 2>&1
( timeout 600 python bench.py --workload logreg --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --detail $O/line_logreg_detail.json > $O/line_logreg.json 2> /dev/null )
( timeout 600 python bench.py --workload resnet18 --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --detail $O/line_resnet18_detail.json > $O/line_resnet18.json 2> /dev/null )
bash tools/profile_bench.sh r06_final/mlp --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-other-workloads > $O/prof_mlp.log 2>&1
bash tools/profile_bench.sh r06_final/logreg --workload logreg --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $O/prof_logreg.log 2>&1
bash tools/profile_bench.sh r06_final/resnet18 --workload resnet18 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $O/prof_resnet.log 2>&1
for w in mlp logreg resnet18; do
  d=$O/$w
  python tools/rocpd_summary.py $(find $d/trace -name "*.db") 2>/dev/null | head -60 > $O/r06_final_${w}_bench.txt
  python tools/pmc_table.py $(find $d/pmc_sq $d/pmc_grbm -name "*.db") > $O/r06_final_${w}_pmc_table.txt 2>&1
done
{ echo; echo "-- the line's fraction from this profile alone (tools/warm_summary.py)"
  python tools/warm_summary.py $(find $O/mlp/trace -name "*.db") mlp3_split 0.98432e12 157.3 TFLOP/s --grbm $(find $O/mlp/pmc_grbm -name "*.db") --line $O/line_mlp.json; } >> $O/r06_final_mlp_bench.txt 2>&1
{ echo; echo "-- the line's fraction from this profile alone (tools/warm_summary.py)"
  python tools/warm_summary.py $(find $O/logreg/trace -name "*.db") dense_narrow16s 27.6e9 8000 GB/s --grbm $(find $O/logreg/pmc_grbm -name "*.db") --line $O/line_logreg.json; } >> $O/r06_final_logreg_bench.txt 2>&1
python tools/traffic_json.py $O/mlp mlp3_split 10000000 mlp "r06 final" > $O/traffic_mlp.json
python tools/traffic_json.py $O/logreg dense_narrow16 50000000 logreg "r06 final" > $O/traffic_logreg.json
python tools/traffic_pass_json.py $O/resnet18 global_avgpool 1024 "r06 final" > $O/traffic_resnet18.json
python tools/trace_last_step.py $(find $O/resnet18/trace -name "*.db") > $O/r06_final_resnet18_last_pass.txt 2>&1
tail -8 $O/r06_final_mlp_bench.txt; tail -6 $O/r06_final_logreg_bench.txt; cat $O/traffic_mlp.json $O/traffic_logreg.json | head -40
