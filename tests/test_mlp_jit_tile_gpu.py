"""Load-time specialised fused MLP chains (hipRTC, mlp_jit.cpp) on the latency-shaped tile kernel: a chain without an
ahead-of-time instantiation whose shape allows it (VALU head, D2 <= 128) gets mlp3_tile_kernel<Cfg, XCM> compiled
beside its persistent kernel, so that short launches -- the host path's 2048-row chunks -- neither sit behind a 160 KB LDS image nor need
a transpose launch.  Every row must come out bit-for-bit as the persistent kernel computes it (a chunk == its slice of a long scan),
through the row-major entry and through the columnar entry, and match the oracle.  Shapes that do not qualify keep one kernel."""
import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth

SHAPES = [((128, 256, 128, 1), True), ((64, 128, 64, 1), True), ((32, 384, 96, 2), True), ((128, 128, 32, 3), True),
          ((64, 64, 64, 1), True),     # D1 = 64: two layer-1 slices
          ((64, 96, 32, 1), True),     # D1 = 96: one slice of three tiles
          ((128, 256, 64, 8), False)]  # 8 outputs: MFMA head -> persistent kernel only


@pytest.mark.gpu
@pytest.mark.parametrize("dims,tile", SHAPES)
def test_gpu_jit_chain_tile_kernel_matches_persistent_kernel(gpu_api, tmp_path, dims, tile):
    from oracle import oracle

    path = W.write(str(tmp_path / "m.onnx"), W.mlp(dims))
    k, out = dims[0], dims[-1]
    big = 40000  # > 32768 rows: the persistent kernel
    x = synth.table(21, 0, big, k)
    gpu_api.load_model("m", path)
    try:
        plan = gpu_api.get_plan("m")
        assert plan["exec"][0] == "mlp3_fused" and "hipRTC" in plan.get("fused_kernel", ""), plan
        long_scan = gpu_api.predict("m", x)
        for lo, n in ((0, 2048), (4096, 1), (8000, 33), (30000, 1234), (100, 32768)):
            chunk = gpu_api.predict("m", x[lo:lo + n])
            assert np.array_equal(chunk, long_scan[lo:lo + n]), (dims, lo, n, float(np.abs(chunk - long_scan[lo:lo + n]).max()))
            cols = [np.ascontiguousarray(x[lo:lo + n, c]) for c in range(k)]
            assert np.array_equal(gpu_api.predict_columns("m", cols), chunk), (dims, lo, n)
        cols = [np.ascontiguousarray(x[:, c]) for c in range(k)]  # longer than the tile kernel's range: transposed, persistent kernel
        assert np.array_equal(gpu_api.predict_columns("m", cols), long_scan)
    finally:
        gpu_api.unload_model("m")
    want = oracle.Model(path).predict(x[:4096])
    assert long_scan.shape == (big, out)
    assert np.all(np.abs(long_scan[:4096] - want) <= 1e-4 * np.abs(want) + 1e-6), np.abs(long_scan[:4096] - want).max()
