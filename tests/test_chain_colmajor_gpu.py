"""The fused small-MLP chain kernel reading a COLUMN-MAJOR chunk (chain_device.inc, x_aligned16 bit 1): the host path stages a
DataChunk's flat columns as they lie and the kernel reads them itself -- no transpose launch in front.  Results through the
columnar entry (`infera_predict_columns`) must be bit-for-bit those of the row-major host entry (`infera_predict`: same kernel,
same sums, only the operand fetch differs) and match the oracle: table widths that are and are not multiples of 4, row counts
that are not multiples of 4 (unaligned column starts -> element-wise fetch), one row, several 32-row tiles, DOUBLE / INTEGER
columns, a chain behind PadCols, softmax / argmax epilogues.  INFERA_CHAIN_XCM=0 at load time = transpose launch first."""
import os

import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth

SHAPES = {
    "sklearn_default": dict(dims=(30, 100, 2), softmax=True),
    "regressor": dict(dims=(13, 64, 32, 1), softmax=False),
    "iris": dict(dims=(4, 10, 3), softmax=True),
    "wide_first": dict(dims=(100, 64, 4), softmax=False),
    "odd_columns": dict(dims=(77, 20, 5), softmax=True),
}


@pytest.mark.gpu
@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("rows", [1, 33, 1234, 2048, 4099])
def test_gpu_chain_kernel_reads_column_major_chunks(gpu_api, tmp_path, shape, rows):
    from oracle import oracle

    sp = SHAPES[shape]
    k = sp["dims"][0]
    path = W.write(str(tmp_path / "m.onnx"), W.mlp(sp["dims"], final_softmax=sp["softmax"]))
    x = synth.table(11, 0, rows, k)
    cols = [np.ascontiguousarray(x[:, c]) for c in range(k)]
    cols[0] = cols[0].astype(np.float64)  # DuckDB's default floating type: converted while staged
    if k > 2:
        cols[2] = np.round(cols[2] * 100).astype(np.int32)
        x = x.copy()
        x[:, 2] = cols[2].astype(np.float32)
    out = {}
    try:
        for mode in ("1", "0"):
            os.environ["INFERA_CHAIN_XCM"] = mode  # read when the model is scheduled
            gpu_api.load_model("m", path)
            assert "chain_fused" in gpu_api.get_plan("m")["exec"]
            out[mode] = gpu_api.predict_columns("m", cols)
            assert np.array_equal(out[mode], gpu_api.predict_columns("m", cols))
            if mode == "1":
                row_major = gpu_api.predict("m", x)
            gpu_api.unload_model("m")
    finally:
        os.environ.pop("INFERA_CHAIN_XCM", None)
    assert np.array_equal(out["1"], out["0"]), np.abs(out["1"] - out["0"]).max()
    assert np.array_equal(out["1"], row_major), np.abs(out["1"] - row_major).max()
    want = oracle.Model(path).predict(x)
    assert out["1"].shape == want.shape
    assert np.all(np.abs(out["1"] - want) <= 1e-4 * np.abs(want) + 1e-6), np.abs(out["1"] - want).max()


DENSE_SHAPES = {
    "linreg_13": dict(dims=(13, 1), softmax=False),          # one lane per row, fmaf chains (skinny kernel)
    "linreg_3": dict(dims=(3, 1), softmax=False),
    "logreg_30x2": dict(dims=(30, 2), softmax=True),
    "softmax_30x3": dict(dims=(30, 3), softmax=True),        # 16x16x4 streaming kernel from here on
    "softmax_100x10": dict(dims=(100, 10), softmax=True),
    "c4_128x10": dict(dims=(128, 10), softmax=True),         # BASELINE C4: 64+ columns keep the transposed path (still checked here)
    "dense_64x5": dict(dims=(64, 5), softmax=False),
    "dense_8x16": dict(dims=(8, 16), softmax=False),
    "dense_77x1": dict(dims=(77, 1), softmax=False),
}


@pytest.mark.gpu
@pytest.mark.parametrize("shape", sorted(DENSE_SHAPES))
@pytest.mark.parametrize("rows", [1, 33, 1234, 2048, 4099])
def test_gpu_single_layer_kernels_read_column_major_chunks(gpu_api, tmp_path, shape, rows):
    from oracle import oracle

    sp = DENSE_SHAPES[shape]
    k = sp["dims"][0]
    path = W.write(str(tmp_path / "m.onnx"), W.mlp(sp["dims"], final_softmax=sp["softmax"]))
    x = synth.table(13, 0, rows, k)
    cols = [np.ascontiguousarray(x[:, c]) for c in range(k)]
    cols[0] = cols[0].astype(np.float64)
    out = {}
    try:
        for mode in ("1", "0"):
            os.environ["INFERA_DENSE_XCM"] = mode  # read when the model is scheduled
            gpu_api.load_model("m", path)
            assert gpu_api.get_plan("m")["exec"][0] in ("normal", "dense_softmax"), gpu_api.get_plan("m")["exec"]
            out[mode] = gpu_api.predict_columns("m", cols)
            assert np.array_equal(out[mode], gpu_api.predict_columns("m", cols))
            if mode == "1":
                row_major = gpu_api.predict("m", x)
            gpu_api.unload_model("m")
    finally:
        os.environ.pop("INFERA_DENSE_XCM", None)
    assert np.array_equal(out["1"], out["0"]), np.abs(out["1"] - out["0"]).max()
    assert np.array_equal(out["1"], row_major), np.abs(out["1"] - row_major).max()
    want = oracle.Model(path).predict(x)
    assert out["1"].shape == want.shape
    assert np.all(np.abs(out["1"] - want) <= 1e-4 * np.abs(want) + 1e-6), np.abs(out["1"] - want).max()


@pytest.mark.gpu
def test_gpu_label_epilogue_reads_column_major_chunks(gpu_api, tmp_path):
    """Dense -> ArgMax (a classifier serving its label) through the columnar entry: the label kernel reads the chunk column-major."""
    from oracle import oracle

    path = W.write(str(tmp_path / "lab.onnx"), W.sklearn_pipeline(30, 3))
    x = synth.table(17, 0, 1500, 30)
    cols = [np.ascontiguousarray(x[:, c]) for c in range(30)]
    gpu_api.load_model("lab", path)
    try:
        got = gpu_api.predict_columns("lab", cols)
        assert np.array_equal(got, gpu_api.predict("lab", x))
    finally:
        gpu_api.unload_model("lab")
    assert np.array_equal(got, oracle.Model(path).predict(x))


@pytest.mark.gpu
@pytest.mark.parametrize("hipgraph", ["0", "1"])
def test_gpu_column_major_chunk_longer_than_one_device_pass(tmp_path, hipgraph):
    """a plan whose FIRST kernel reads column-major chunks (narrow Dense 16 -> 8) followed by unfused wide
    intermediates (8 -> 520 -> 1: scratch of > 256 floats per row) cuts a 300k-row call into two device passes -- a [K][rows]
    chunk cannot be cut into row passes, so such a call must be staged the transposing way.  Through the columnar entry the
    results must equal the row-major entry bit for bit, and the oracle.  Child process: INFERA_HIPGRAPH is read once."""
    import subprocess
    import sys

    path = W.write(str(tmp_path / "m.onnx"), W.mlp((16, 8, 520, 1)))
    code = f"""
import numpy as np
from infera_amd import capi, synth
from oracle import oracle
rows, k = 300_000, 16
capi.load_model("m", {path!r})
plan = capi.get_plan("m")
assert plan["scratch_floats_per_row"] > 256, plan
x = synth.table(23, 0, rows, k)
cols = [np.ascontiguousarray(x[:, c]) for c in range(k)]
got = capi.predict_columns("m", cols)
assert np.array_equal(got, capi.predict("m", x)), float(np.abs(got - capi.predict("m", x)).max())
short = capi.predict_columns("m", [c[:2048] for c in cols])
assert np.array_equal(short, got[:2048])
want = oracle.Model({path!r}).predict(x[:4096])
assert np.all(np.abs(got[:4096] - want) <= 1e-4 * np.abs(want) + 1e-6), float(np.abs(got[:4096] - want).max())
tail = oracle.Model({path!r}).predict(x[-4096:])
assert np.all(np.abs(got[-4096:] - tail) <= 1e-4 * np.abs(tail) + 1e-6), float(np.abs(got[-4096:] - tail).max())
print("ok", plan["exec"])
"""
    env = dict(os.environ, INFERA_HIPGRAPH=hipgraph)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout + r.stderr
