#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the SAME bench.py
# command, as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE in their own passes; no
# sys/hip/hsa trace domains together with --pmc).  Raw .db files land in gpurun_out/<tag>/; the text
# summary and traffic JSON are produced afterwards with tools/rocpd_summary.py / tools/traffic_json.py.
set -u
TAG=${1:-prof}
shift || true
BENCH_ARGS=${*:---steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python bench.py $BENCH_ARGS > "$OUT/bench_trace.log" 2>&1
grep '^{' "$OUT/bench_trace.log" > "$OUT/bench_line.json"
# (counter passes serialise kernels: the two lanes of a long convolutional pass would show up as half-size launches one after the other --
#  the PMC passes run single-lane, i.e. per-kernel figures of full-size launches; the kernel trace above is the product's default)
export INFERA_CONV_LANES=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    -d "$OUT/pmc_sq" -o bench -- python bench.py $BENCH_ARGS > "$OUT/bench_pmc_sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc_grbm" -o bench -- python bench.py $BENCH_ARGS > "$OUT/bench_pmc_grbm.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- python bench.py $BENCH_ARGS > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- python bench.py $BENCH_ARGS > "$OUT/bench_pmc_write.log" 2>&1
cat "$OUT/bench_line.json" | cut -c1-400
find "$OUT" -name "*.db" | sort
