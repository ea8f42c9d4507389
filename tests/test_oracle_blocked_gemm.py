"""The "best CPU" leg of bench.py's cpu_baseline (oracle/infera_oracle.c gemm_block_avx512 / gemm_block_avx2): a register-blocked
micro-kernel GEMM must give the SAME BITS as the oracle's plain k-ordered loop -- every output element stays one fmaf chain over
k from 0 -- so the faster CPU baseline is the same computation, only scheduled the way a packed SIMD matmul (Tract's) is."""
import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import sqlharness, synth
from oracle import oracle


@pytest.mark.parametrize("dims", [(128, 256, 64, 1), (128, 10), (30, 100, 2), (7, 33, 5), (64, 96, 40, 3)])
@pytest.mark.parametrize("rows", [1, 5, 6, 13, 2048])
def test_blocked_gemm_bit_identical(tmp_path, dims, rows):
    path = W.write(str(tmp_path / "m.onnx"), W.mlp(dims, final_softmax=dims[-1] > 1))
    m = oracle.Model(path)
    x = synth.table(5, 0, rows, dims[0])
    plain = m.predict(x)
    oracle.set_blocked_gemm(True)
    try:
        blocked = m.predict(x)
    finally:
        oracle.set_blocked_gemm(False)
    assert np.array_equal(plain, blocked)


def test_best_cpu_scan_same_checksum(tmp_path):
    """bench_scan_table's three gather / GEMM variants scan the same table to the same checksum."""
    path = W.write(str(tmp_path / "m.onnx"), W.mlp((128, 256, 64, 1)))
    m = oracle.Model(path)
    rows = sqlharness.ROW_GROUP + 4096  # one full row group + a ragged one
    table = sqlharness.synth_table(rows, 128, 42, 2)
    sums = [oracle.bench_scan_table(m, table, rows, 128, threads=2, boxed=b)[1] for b in (1, 0, 2)]
    assert sums[0] == sums[1] == sums[2], sums
