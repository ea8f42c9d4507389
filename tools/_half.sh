cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_half_chunk_ab.txt; : > $O
python - >> $O 2>&1 <<'P'
# bit-identity of the half-chunk experiment against the whole-chunk staged path
import os, subprocess, sys
code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from infera_amd import capi, onnx_writer as W, synth
capi.load_model("m", W.write("/tmp/m_half.onnx", W.mlp((128, 256, 64, 1))))
x = synth.table(3, 0, 2048, 128)
print(capi.predict_columns("m", [np.ascontiguousarray(x[:, j]) for j in range(128)]).tobytes().hex()[:64], float(capi.predict("m", x).sum()))
'''
a = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, INFERA_STREAM_MAX_INFLIGHT="0"), capture_output=True, text=True).stdout
b = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, INFERA_STREAM_MAX_INFLIGHT="0", INFERA_EXP_HALF_CHUNK="1"), capture_output=True, text=True).stdout
print("half-chunk results bit-identical to whole-chunk:", a == b and len(a) > 10)
P
for mode in "INFERA_EXP_HALF_CHUNK=0" "INFERA_EXP_HALF_CHUNK=1" "INFERA_EXP_HALF_CHUNK=0" "INFERA_EXP_HALF_CHUNK=1"; do
echo "=== $mode (INFERA_STREAM_MAX_INFLIGHT=0)" >> $O
env $mode INFERA_STREAM_MAX_INFLIGHT=0 python tools/host_scan_bench.py --rows 6000000 --threads 1,2,4,8,16 --reps 3 --numa auto 2>&1 | grep "^threads\|us/chunk" >> $O
done
cat $O
