"""Zero-copy host path (include/infera_hip.h: infera_hip_register_host_memory; the reference's ROADMAP.md:44 "zero-copy"): when the
application has registered the memory its column runs live in, `infera_predict_columns` lets the GPU read the runs in place over PCIe --
no CPU gather, no pinned staging, no H2D copy.  The results must be bit-for-bit those of the staged path (same kernels behind the
gather, conversions with the reference's static_cast<float> roundings, infera_extension.cpp:211-222), for the BASELINE C2 / C4 models,
for plans whose first kernel cannot read a column-major chunk (GPU transpose behind the gather), for DOUBLE / INTEGER / BIGINT / constant
columns, unaligned runs and ragged row counts; a chunk with ANY column outside the registered ranges takes the staged path; registration
errors are errors.  The oracle is the referee for one case of each model."""
import ctypes as C
import threading

import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth


def _page_aligned(shape, dtype):
    """An array that owns whole 4 KiB pages (registration is per page: two registered ranges must not share one)."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.zeros((n + 4095) // 4096 * 4096 + 4096, np.uint8)
    off = (-raw.ctypes.data) % 4096
    a = raw[off:off + n].view(dtype).reshape(shape)
    assert a.ctypes.data % 4096 == 0 and a.flags.c_contiguous
    return a


def _table(k, n, seed=3):
    """A [k][n] column-major table in ONE page-aligned buffer (what gets registered) and its row-major twin."""
    x = synth.table(seed, 0, n, k)
    big = _page_aligned((k, n), np.float32)
    big[...] = x.T
    return big, x


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["c2_mlp", "c4_logreg", "chain_30_100_2", "wide_dense_128_512_8"])
@pytest.mark.parametrize("rows", [1, 33, 2048, 4099])
def test_gpu_zero_copy_equals_staged_path(gpu_api, tmp_path, model, rows):
    from oracle import oracle

    blob, k = {"c2_mlp": (W.mlp((128, 256, 64, 1)), 128), "c4_logreg": (W.logreg_softmax(128, 10), 128),
               "chain_30_100_2": (W.mlp((30, 100, 2), final_softmax=True), 30),
               "wide_dense_128_512_8": (W.mlp((128, 512, 8)), 128)}[model]  # (first kernel = tiled Dense: no column-major reader -> GPU transpose)
    path = W.write(str(tmp_path / "m.onnx"), blob)
    big, x = _table(k, rows + 5)
    cols = [big[c, 3:3 + rows] for c in range(k)]  # runs that start 12 bytes into their row: not 16-byte aligned
    gpu_api.load_model("zc", path)
    try:
        staged = gpu_api.predict_columns("zc", cols)
        before = gpu_api.zero_copy_calls()
        gpu_api.register_host_memory(big)
        try:
            got = gpu_api.predict_columns("zc", cols)
            assert gpu_api.zero_copy_calls() == before + 1
            assert np.array_equal(got, staged), float(np.abs(got - staged).max())
            aligned = [big[c, :rows] for c in range(k)]  # 16-byte aligned runs: the float4 path (when rows % 4 == 0)
            assert np.array_equal(gpu_api.predict_columns("zc", aligned), gpu_api.predict("zc", x[:rows]))
            assert gpu_api.zero_copy_calls() == before + 2
        finally:
            gpu_api.unregister_host_memory(big)
        assert np.array_equal(gpu_api.predict_columns("zc", cols), staged) and gpu_api.zero_copy_calls() == before + 2  # staged again
    finally:
        gpu_api.unload_model("zc")
    if rows in (33, 2048):
        want = oracle.Model(path).predict(x[3:3 + rows])
        assert np.all(np.abs(staged - want) <= 1e-4 * np.abs(want) + 1e-6)


@pytest.mark.gpu
def test_gpu_zero_copy_typed_and_constant_columns(gpu_api, tmp_path):
    """DOUBLE / INTEGER / BIGINT / constant columns are converted by the GPU exactly as the CPU gather converts them."""
    k, rows = 16, 3001
    path = W.write(str(tmp_path / "m.onnx"), W.mlp((k, 32, 1)))
    rng = np.random.default_rng(9)
    f64, i32, i64, f32, const = (_page_aligned((4, rows), np.float64), _page_aligned((4, rows), np.int32), _page_aligned((4, rows), np.int64),
                                 _page_aligned((3, rows), np.float32), _page_aligned((1,), np.float64))
    f64[...] = rng.standard_normal((4, rows)) * (1 + 1e-9)                                 # values that round on the way to f32
    i32[...] = rng.integers(-2**31, 2**31 - 1, (4, rows), dtype=np.int64).astype(np.int32)  # beyond 2^24: inexact in f32
    i64[...] = rng.integers(-2**62, 2**62, (4, rows), dtype=np.int64)
    f32[...] = rng.standard_normal((3, rows)).astype(np.float32)
    const[0] = 0.375
    cols = [f64[0], i32[0], i64[0], f32[0], f64[1], i32[1], i64[1], f32[1], f64[2], i32[2], i64[2], f32[2], f64[3], i32[3], i64[3], const]
    gpu_api.load_model("zt", path)
    regs = [f64, i32, i64, f32, const]
    try:
        staged = gpu_api.predict_columns("zt", cols, rows=rows)
        before = gpu_api.zero_copy_calls()
        for a in regs:
            gpu_api.register_host_memory(a)
        try:
            got = gpu_api.predict_columns("zt", cols, rows=rows)
            assert gpu_api.zero_copy_calls() == before + 1
            assert np.array_equal(got.view(np.uint32), staged.view(np.uint32))
            # one column from memory nobody registered: the whole chunk is staged, same result
            outsider = list(cols)
            outsider[3] = f32[0].copy()
            assert np.array_equal(gpu_api.predict_columns("zt", outsider, rows=rows), staged)
            assert gpu_api.zero_copy_calls() == before + 1
        finally:
            for a in regs:
                gpu_api.unregister_host_memory(a)
    finally:
        gpu_api.unload_model("zt")


@pytest.mark.gpu
def test_gpu_zero_copy_long_call_and_registration_errors(gpu_api, tmp_path):
    k = 16
    path = W.write(str(tmp_path / "m.onnx"), W.mlp((k, 8, 520, 1)))  # 300k rows: two device passes behind one gather (ADVICE r2 shape)
    rows = 300_000
    big, x = _table(k, rows)
    cols = [big[c] for c in range(k)]
    gpu_api.load_model("zl", path)
    try:
        staged = gpu_api.predict_columns("zl", cols)
        gpu_api.register_host_memory(big)
        try:
            before = gpu_api.zero_copy_calls()
            assert np.array_equal(gpu_api.predict_columns("zl", cols), staged)
            assert gpu_api.zero_copy_calls() == before + 1
            with pytest.raises(gpu_api.InferaError, match="overlaps"):
                gpu_api.register_host_memory(big[2:4])
            assert np.array_equal(gpu_api.predict_columns("zl", cols), staged)
        finally:
            gpu_api.unregister_host_memory(big)
        with pytest.raises(gpu_api.InferaError, match="not registered"):
            gpu_api.unregister_host_memory(big)
    finally:
        gpu_api.unload_model("zl")


@pytest.mark.gpu
def test_gpu_zero_copy_concurrent_callers(gpu_api, tmp_path):
    """8 threads x 40 chunks each over one registered table: every chunk equals its slice of one big staged call, whether it was fetched in
    place or -- the GPU already running its four zero-copy fetches -- staged."""
    k, chunk, nchunks = 128, 2048, 40
    path = W.write(str(tmp_path / "m.onnx"), W.mlp((128, 256, 64, 1)))
    big, x = _table(k, chunk * nchunks, seed=11)
    gpu_api.load_model("zcc", path)
    try:
        want = gpu_api.predict("zcc", x)
        gpu_api.register_host_memory(big)
        bad = []

        def work(t):
            for i in range(t, nchunks, 8):
                cols = [big[c, i * chunk:(i + 1) * chunk] for c in range(k)]
                got = gpu_api.predict_columns("zcc", cols)
                if not np.array_equal(got, want[i * chunk:(i + 1) * chunk]):
                    bad.append(i)

        try:
            before = gpu_api.zero_copy_calls()
            th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not bad, bad
            # (under an INFERA_ZERO_COPY_MAX_INFLIGHT limit -- round 4's default was 4, round 5's is none -- surplus chunks are staged: same results, fewer in-place fetches)
            assert before + nchunks // 4 <= gpu_api.zero_copy_calls() <= before + nchunks
        finally:
            gpu_api.unregister_host_memory(big)
    finally:
        gpu_api.unload_model("zcc")


@pytest.mark.gpu
def test_gpu_zero_copy_ranges_that_share_pages(gpu_api, tmp_path):
    """The runtime pins whole pages, callers register byte ranges: neighbours on the heap (numpy arrays, malloc'ed buffers) share
    pages.  Ranges whose page spans touch are served by one registration of their union; a shared block lives until its LAST range
    is unregistered; columns may come from several ranges at once."""
    k, rows = 16, 1000  # 4000-byte runs: every column shares pages with its neighbours
    path = W.write(str(tmp_path / "m.onnx"), W.mlp((k, 32, 1)))
    raw = _page_aligned((k * rows + 64,), np.float32)
    x = synth.table(21, 0, rows, k)
    cols = []
    for c in range(k):
        run = raw[c * rows + 3:c * rows + 3 + rows - 6]  # k separate, non-overlapping ranges packed into the same pages
        run[...] = x[:rows - 6, c]
        cols.append(run)
    gpu_api.load_model("zp", path)
    try:
        staged = gpu_api.predict_columns("zp", cols)
        before = gpu_api.zero_copy_calls()
        for run in cols[::-1]:  # (any order)
            gpu_api.register_host_memory(run)
        try:
            assert np.array_equal(gpu_api.predict_columns("zp", cols), staged) and gpu_api.zero_copy_calls() == before + 1
            for run in cols[:8]:  # half of the ranges go: the shared pages must stay pinned for the others
                gpu_api.unregister_host_memory(run)
            assert np.array_equal(gpu_api.predict_columns("zp", cols), staged) and gpu_api.zero_copy_calls() == before + 1  # staged: 8 columns are outside
            assert np.array_equal(gpu_api.predict_columns("zp", cols[8:] + cols[8:]), gpu_api.predict("zp", np.concatenate([x[:rows - 6, 8:]] * 2, axis=1)))
            assert gpu_api.zero_copy_calls() == before + 2
            for run in cols[:8]:  # ... and can come back
                gpu_api.register_host_memory(run)
            assert np.array_equal(gpu_api.predict_columns("zp", cols), staged) and gpu_api.zero_copy_calls() == before + 3
        finally:
            for run in cols:
                try:
                    gpu_api.unregister_host_memory(run)
                except gpu_api.InferaError:
                    pass
        assert np.array_equal(gpu_api.predict_columns("zp", cols), staged) and gpu_api.zero_copy_calls() == before + 3
    finally:
        gpu_api.unload_model("zp")


RECT_CHILD = r"""
import hashlib, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from infera_amd import capi, onnx_writer as W, synth
capi.load_model("zr", W.write(os.path.join(%(tmp)r, "m.onnx"), W.mlp((128, 256, 64, 1))))
rows, k = 2048 * 6 + 77, 128
raw = np.zeros(k * rows * 4 + 8192, np.uint8)
off = (-raw.ctypes.data) %% 4096
big = raw[off:off + k * rows * 4].view(np.float32).reshape(k, rows)
big[...] = synth.table(3, 0, rows, k).T
capi.register_host_memory(big)
h = hashlib.sha256()
z0 = capi.zero_copy_calls()
for r0, n in ((0, 2048), (2048, 2048), (4099, 1000), (rows - 77, 77), (5, 1)):
    h.update(capi.predict_columns("zr", [big[c, r0:r0 + n] for c in range(k)]).tobytes())
print("RESULT " + json.dumps({"sha": h.hexdigest(), "zero_copy_calls": capi.zero_copy_calls() - z0}))
"""


@pytest.mark.gpu
def test_gpu_zero_copy_rect_copy_equals_pulling_kernel(gpu_api, tmp_path):
    """Round 4: chunks of a pitched FLOAT table are fetched by ONE 2-D copy (INFERA_ZERO_COPY_RECT=1, default); =0 keeps the pulling kernel.
    Same bytes in HBM either way, so the results must be identical -- aligned and unaligned row offsets, ragged tails, a single row."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for rect in ("1", "0"):
        env = dict(os.environ, INFERA_ZERO_COPY_RECT=rect)
        p = subprocess.run([sys.executable, "-c", RECT_CHILD % {"root": root, "tmp": str(tmp_path)}], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        out[rect] = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
        assert out[rect]["zero_copy_calls"] == 5
    assert out["1"]["sha"] == out["0"]["sha"]


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [4, 252, 2044, 2048])
def test_gpu_zero_copy_runs_eight_bytes_past_a_16_byte_boundary(gpu_api, tmp_path, rows):
    """A FLAT vector inside a DuckDB buffer-manager block starts 8 bytes past a 16-byte boundary (the block header is 8 bytes): round 6 gave
    such runs their own path in the pulling kernel (aligned 16-byte pieces, shifted by half a piece) -- bit-identical to the staged path, and
    no access leaves the run's own pages: the first column starts 8 bytes into the FIRST registered page and the last column ends 8 bytes
    before the end of the LAST one (a 16-byte read one piece further out would touch unregistered, possibly unmapped memory)."""
    k = 128
    path = W.write(str(tmp_path / "m.onnx"), W.mlp((128, 256, 64, 1)))
    x = synth.table(11, 0, rows, k)
    # every column run in its own page-aligned slab, 8 bytes in; the last run placed so that it ENDS 8 bytes before its slab's end
    slab_floats = (rows * 4 + 16 + 4095) // 4096 * 4096 // 4
    big = _page_aligned((k, slab_floats), np.float32)
    cols = []
    for c in range(k):
        start = 2 if c < k - 1 else slab_floats - 2 - rows
        big[c, start:start + rows] = x[:, c]
        cols.append(big[c, start:start + rows])
        assert cols[-1].ctypes.data % 16 == 8
    gpu_api.load_model("zc8", path)
    try:
        staged = gpu_api.predict_columns("zc8", cols)
        before = gpu_api.zero_copy_calls()
        gpu_api.register_host_memory(big)
        try:
            got = gpu_api.predict_columns("zc8", cols)
            assert gpu_api.zero_copy_calls() == before + 1
        finally:
            gpu_api.unregister_host_memory(big)
        assert np.array_equal(got, staged) and np.array_equal(got, gpu_api.predict("zc8", x))
    finally:
        gpu_api.unload_model("zc8")


@pytest.mark.gpu
def test_gpu_scan_over_duckdb_shaped_segments_equals_the_staged_scan(gpu_api, tmp_path, monkeypatch):
    """the registered scan on DuckDB's block layout -- per (row group, column) two separately allocated 256 KiB blocks from the
    extension's REGISTERING allocator (8-byte header, 65,534 values), 128 unrelated block addresses per chunk, the chunk that straddles two
    segments assembled in ordinary memory.  Every chunk but those is fetched in place; the scan's checksum is the staged scan's over the
    contiguous twin of the table (same values), and the allocator's blocks are all unregistered again when the table is closed."""
    from infera_amd import sqlharness

    rows, k = 122_880 * 2 + 70_000, 128
    path = W.write(str(tmp_path / "m.onnx"), W.mlp((128, 256, 64, 1)))
    gpu_api.load_model("seg", path)
    try:
        flat = sqlharness.synth_table(rows, k, 42, 8)
        (secs, want) = sqlharness.bench_scan_table("infera_predict", "seg", flat, rows, k, 4, 1)
        monkeypatch.setenv("INFERA_ZERO_COPY_ALLOCATOR", "1")
        ranges0 = gpu_api.get_devices()["registered_host_ranges"]
        nchunks = (rows + 2047) // 2048
        # the allocator's ARENA (slabs of 256 blocks, one registration each) with the blocks handed out in order / in a random order, and the
        # per-block registrations of INFERA_ZERO_COPY_ARENA=0: the same results whichever mechanism fetches a chunk
        for arena, shuffled, ranges in (("1", False, 3), ("1", True, 3), ("0", True, 128 * 6)):
            monkeypatch.setenv("INFERA_ZERO_COPY_ARENA", arena)
            t = sqlharness.SegmentTable(rows, k, 42, 8, shuffled=shuffled)
            try:
                assert t.registering_allocator and t.blocks == 128 * 6   # three row groups x two segments
                assert gpu_api.get_devices()["registered_host_ranges"] == ranges0 + ranges, (arena, shuffled)
                before = gpu_api.zero_copy_calls()
                (secs, got) = sqlharness.bench_scan_segments("infera_predict", "seg", t, rows, k, 4, 1)
                assert t.assembled_chunks == 3                                             # one straddling chunk per row group
                assert gpu_api.zero_copy_calls() - before == nchunks - t.assembled_chunks    # every other chunk in place
                assert abs(got - want) <= 1e-6 * rows, (got, want, arena, shuffled)        # same fp32 results, summed in double in another order
            finally:
                t.close()
            # (the arena keeps ONE empty slab for the next allocation; everything else is unregistered again)
            assert gpu_api.get_devices()["registered_host_ranges"] in (ranges0, ranges0 + 1), (arena, shuffled)
        monkeypatch.delenv("INFERA_ZERO_COPY_ARENA")
        # ... and without the hook the same table is ordinary memory: every chunk staged, same checksum
        monkeypatch.delenv("INFERA_ZERO_COPY_ALLOCATOR")
        t = sqlharness.SegmentTable(rows, k, 42, 8)
        try:
            assert not t.registering_allocator
            before = gpu_api.zero_copy_calls()
            (secs, got2) = sqlharness.bench_scan_segments("infera_predict", "seg", t, rows, k, 4, 1)
            assert gpu_api.zero_copy_calls() == before and abs(got2 - want) <= 1e-6 * rows
        finally:
            t.close()
    finally:
        gpu_api.unload_model("seg")
