#!/bin/bash
# the driver-shaped default run + the full GPU suite + sanitizer runs on the final tree
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04_final_check
mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.txt
( timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/r04_line_mlp_detail.json > $O/r04_line_mlp.json 2> /dev/null; echo "rc=$? bytes=$(wc -c < $O/r04_line_mlp.json)" ) > $O/bench_rc.txt 2>&1
bash tools/sanitizers_run.sh > $O/sanitizers.txt 2>&1
cp gpurun_out/tsan/*.txt $O/ 2>/dev/null
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.txt
