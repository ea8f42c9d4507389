set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02_lines
bash tools/profile_bench.sh r02_final_resnet18 --workload resnet18 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r02_lines/prof_resnet.log 2>&1
bash tools/profile_bench.sh r02_final_mlp --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads > gpurun_out/r02_lines/prof_mlp.log 2>&1
python bench.py > gpurun_out/r02_lines/mlp.log 2>&1
python bench.py --workload logreg > gpurun_out/r02_lines/logreg.log 2>&1
python bench.py --workload resnet18 > gpurun_out/r02_lines/resnet18.log 2>&1
python bench.py --precision bf16x3 > gpurun_out/r02_lines/mlp_bf16x3.log 2>&1
for f in mlp logreg resnet18 mlp_bf16x3; do grep '^{' gpurun_out/r02_lines/$f.log | tail -1 | cut -c1-300; done
