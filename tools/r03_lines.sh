#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03_lines
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r03_lines/gpu_tests.txt
python bench.py > gpurun_out/r03_lines/mlp.json 2> gpurun_out/r03_lines/mlp.err
python bench.py --workload logreg > gpurun_out/r03_lines/logreg.json 2> gpurun_out/r03_lines/logreg.err
python bench.py --workload resnet18 > gpurun_out/r03_lines/resnet18.json 2> gpurun_out/r03_lines/resnet18.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_lines/smoke.txt 2>&1
cat gpurun_out/r03_lines/gpu_tests.txt gpurun_out/r03_lines/smoke.txt
for f in mlp logreg resnet18; do grep '^{' gpurun_out/r03_lines/$f.json | tail -1 | cut -c1-200; done
