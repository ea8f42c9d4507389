"""ctypes driver of the chunk harness: tests/duckdb_stub/libinfera_duckdb_stub.so = the REAL DuckDB extension source
(infera_amd/csrc/binding/infera_extension_hip.cpp) compiled against the test-only stand-in for duckdb.hpp, the one-chunk driver
(`infera_sql_call`, tests/duckdb_stub/driver.cpp) and the table-scan drivers bench.py times (csrc/binding/scan_driver.cpp); C ABI in
csrc/binding/sql_surface.h.  Lets the tests read like the reference's sqllogictests (/root/reference test/sql/*.test):
`sql("infera_predict", "linear", 1.0, 2.0, 3.0)`.  Test / bench infrastructure only.  (Until round 4 a second, mock implementation
of the SQL functions stood beside the extension source and carried the benchmarks; it is gone: one SQL layer.)

Argument conventions (one call = one DataChunk of <= 2048 rows):
  * Python str / bytes / float / int / None  -> CONSTANT_VECTOR (None = SQL NULL)
  * numpy array (float32/float64/int32/int64) -> FLAT_VECTOR; numpy.ma masked entries = NULL
  * list of str / bytes / None                -> FLAT VARCHAR / BLOB vector
"""
from __future__ import annotations

import ctypes as C
import json
import time
import os
from typing import Any

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "tests", "duckdb_stub", "libinfera_duckdb_stub.so")


class SqlError(RuntimeError):
    """Carries the message exactly as DuckDB would print it ("Invalid Input Error: ...")."""


class _Vector(C.Structure):
    _fields_ = [("type", C.c_int32), ("is_constant", C.c_int32), ("data", C.c_void_p), ("lens", C.POINTER(C.c_uint64)),
                ("validity", C.POINTER(C.c_uint64))]


class _Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("error", C.c_char_p), ("type", C.c_int32), ("is_constant", C.c_int32),
                ("rows", C.c_uint64), ("f32", C.POINTER(C.c_float)), ("boolean", C.POINTER(C.c_uint8)),
                ("strings", C.POINTER(C.c_char_p)), ("list_offsets", C.POINTER(C.c_uint64)),
                ("list_values", C.POINTER(C.c_float)), ("validity", C.POINTER(C.c_uint64))]


VARCHAR, FLOAT, DOUBLE, INTEGER, BIGINT, BLOB, BOOLEAN, LIST_FLOAT = range(8)
_NP = {np.dtype(np.float32): FLOAT, np.dtype(np.float64): DOUBLE, np.dtype(np.int32): INTEGER, np.dtype(np.int64): BIGINT}



_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        capi.load_library()  # libinfera.so first (the instance the harness library links against)
        if not os.path.exists(LIB_PATH):
            raise capi.InferaError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.infera_sql_call.argtypes = [C.c_char_p, C.POINTER(_Vector), C.c_size_t, C.c_size_t, C.POINTER(_Result)]
        L.infera_sql_call.restype = C.c_int32
        L.infera_sql_free_result.argtypes = [C.POINTER(_Result)]
        L.infera_sql_list_functions.restype = C.c_void_p
        L.infera_sql_bench_scan.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64,
                                            C.POINTER(C.c_double), C.c_char_p, C.c_uint64]
        L.infera_sql_bench_scan.restype = C.c_double
        L.infera_sql_table_floats.argtypes = [C.c_uint64, C.c_uint32]
        L.infera_sql_table_floats.restype = C.c_uint64
        L.infera_sql_synth_table.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int32]
        L.infera_sql_synth_table.restype = None
        L.infera_sql_bench_scan_table.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int32,
                                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_char_p, C.c_uint64]
        L.infera_sql_bench_scan_table.restype = C.c_int32
        L.infera_sql_synth_table_f64.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int32]
        L.infera_sql_synth_table_f64.restype = None
        L.infera_sql_bench_scan_table_typed.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint32, C.c_int32, C.c_int32,
                                                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_char_p, C.c_uint64]
        L.infera_sql_bench_scan_table_typed.restype = C.c_int32
        L.infera_sql_bench_blob_scan.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32,
                                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_char_p, C.c_uint64]
        L.infera_sql_bench_blob_scan.restype = C.c_int32
        L.infera_sql_bench_last_times.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.infera_sql_bench_last_times.restype = None
        L.infera_sql_bench_last_cpu.argtypes = [C.POINTER(C.c_double)] * 3
        L.infera_sql_bench_last_cpu.restype = None
        L.infera_sql_bench_gather_only.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.infera_sql_bench_gather_only.restype = C.c_int32
        L.infera_stub_segment_table_create.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_int32, C.POINTER(C.c_int32), C.c_uint64, C.c_int32]
        L.infera_stub_segment_table_create.restype = C.c_void_p
        L.infera_stub_segment_table_get.argtypes = [C.c_void_p]
        L.infera_stub_segment_table_get.restype = C.c_void_p
        L.infera_stub_segment_table_destroy.argtypes = [C.c_void_p]
        L.infera_stub_segment_table_destroy.restype = None
        L.infera_sql_segment_table_blocks.argtypes = [C.c_void_p]
        L.infera_sql_segment_table_blocks.restype = C.c_uint64
        L.infera_sql_bench_scan_segments.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_double),
                                                     C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64]
        L.infera_sql_bench_scan_segments.restype = C.c_int32
        _lib = L
    return _lib


def list_functions() -> list[dict]:
    p = lib().infera_sql_list_functions()
    s = C.string_at(p).decode()
    C.CDLL(None).free(C.c_void_p(p))
    return json.loads(s)


class Decimal:
    """A DECIMAL(18, scale) argument vector for the duckdb_stub backend (test-only): int64 values scaled by 10**scale."""

    def __init__(self, values, scale: int = 3):
        self.scale = scale
        self.raw = np.ascontiguousarray(np.round(np.asarray(values, np.float64) * 10 ** scale).astype(np.int64))


def _validity_words(mask_valid: np.ndarray) -> np.ndarray:
    n = len(mask_valid)
    words = np.zeros((n + 63) // 64, np.uint64)
    for i in np.nonzero(mask_valid)[0]:
        words[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    return words


def _make_vector(arg: Any, keep: list) -> tuple[_Vector, int | None]:
    """Returns (vector, row count or None for constants)."""
    v = _Vector()
    if arg is None or isinstance(arg, (str, bytes, float, int)):
        v.is_constant = 1
        if arg is None:
            v.type = VARCHAR
            words = np.zeros(1, np.uint64)
            buf = C.create_string_buffer(b"", 1)
            ptrs = (C.c_void_p * 1)(C.addressof(buf))
            lens = np.zeros(1, np.uint64)
            keep += [words, buf, ptrs, lens]
            v.data = C.addressof(ptrs)
            v.lens = lens.ctypes.data_as(C.POINTER(C.c_uint64))
            v.validity = words.ctypes.data_as(C.POINTER(C.c_uint64))
        elif isinstance(arg, (str, bytes)):
            raw = arg.encode() if isinstance(arg, str) else arg
            v.type = VARCHAR if isinstance(arg, str) else BLOB
            buf = C.create_string_buffer(raw, max(len(raw), 1))
            ptrs = (C.c_void_p * 1)(C.addressof(buf))
            lens = np.array([len(raw)], np.uint64)
            keep += [buf, ptrs, lens]
            v.data = C.addressof(ptrs)
            v.lens = lens.ctypes.data_as(C.POINTER(C.c_uint64))
        else:
            a = np.array([arg], np.float64 if isinstance(arg, float) else np.int32)
            keep.append(a)
            v.type = _NP[a.dtype]
            v.data = a.ctypes.data
        return v, None
    if isinstance(arg, (list, tuple)):
        n = len(arg)
        is_blob = any(isinstance(x, bytes) for x in arg)
        v.type = BLOB if is_blob else VARCHAR
        bufs, valid = [], np.ones(n, bool)
        for i, x in enumerate(arg):
            if x is None:
                valid[i] = False
                x = b""
            raw = x.encode() if isinstance(x, str) else x
            bufs.append((C.create_string_buffer(raw, max(len(raw), 1)), len(raw)))
        ptrs = (C.c_void_p * max(n, 1))(*[C.addressof(b) for b, _ in bufs])
        lens = np.array([ln for _, ln in bufs] or [0], np.uint64)
        keep += [bufs, ptrs, lens]
        v.data = C.addressof(ptrs)
        v.lens = lens.ctypes.data_as(C.POINTER(C.c_uint64))
        if not valid.all():
            words = _validity_words(valid)
            keep.append(words)
            v.validity = words.ctypes.data_as(C.POINTER(C.c_uint64))
        return v, n
    if isinstance(arg, Decimal):
        keep.append(arg.raw)
        v.type = 8 | (arg.scale << 8)
        v.data = arg.raw.ctypes.data
        return v, len(arg.raw)
    a = arg
    mask = None
    if isinstance(a, np.ma.MaskedArray):
        mask = ~np.ma.getmaskarray(a)
        a = a.filled(0)
    a = np.ascontiguousarray(a)
    if a.dtype not in _NP:
        raise TypeError(f"unsupported column dtype {a.dtype}")
    keep.append(a)
    v.type = _NP[a.dtype]
    v.data = a.ctypes.data
    if mask is not None and not mask.all():
        words = _validity_words(mask)
        keep.append(words)
        v.validity = words.ctypes.data_as(C.POINTER(C.c_uint64))
    return v, len(a)


def sql(function: str, *args: Any, rows: int | None = None):
    """Executes one SQL scalar function call on one chunk and returns Python values:
    FLOAT -> np.ndarray[rows]; VARCHAR -> list[str] (or str if constant); BOOLEAN -> bool;
    LIST<FLOAT> -> list[np.ndarray | None]; a constant NULL result -> None."""
    keep: list = []
    vecs = (_Vector * max(len(args), 1))()
    counts = []
    for i, a in enumerate(args):
        vecs[i], n = _make_vector(a, keep)
        if n is not None:
            counts.append(n)
    if rows is None:
        rows = counts[0] if counts else 1
    assert all(c == rows for c in counts), "all flat vectors of a chunk must have the same row count"
    res = _Result()
    rc = lib().infera_sql_call(function.encode(), vecs, len(args), rows, C.byref(res))
    try:
        if rc != 0:
            raise SqlError(res.error.decode() if res.error else "unknown error")

        def is_valid(i):
            if not res.validity:
                return True
            return bool((res.validity[i >> 6] >> (i & 63)) & 1)

        if res.is_constant and not is_valid(0):
            return None
        if res.type == FLOAT:
            return np.ctypeslib.as_array(res.f32, shape=(rows,)).copy() if rows else np.zeros(0, np.float32)
        if res.type == BOOLEAN:
            return bool(res.boolean[0]) if res.boolean else None
        if res.type == VARCHAR:
            if res.is_constant:
                return res.strings[0].decode()
            return [res.strings[i].decode() for i in range(rows)] if res.strings else []
        out = []
        for i in range(rows if res.list_offsets else 0):
            if not is_valid(i):
                out.append(None)
                continue
            a, b = res.list_offsets[i], res.list_offsets[i + 1]
            out.append(np.array([res.list_values[j] for j in range(a, b)], np.float32) if b - a < 64 else
                       np.ctypeslib.as_array(res.list_values, shape=(res.list_offsets[rows],))[a:b].copy())
        return out
    finally:
        lib().infera_sql_free_result(C.byref(res))


def bench_scan(function: str, model: str, rows: int, ncols: int, threads: int, pool_chunks: int = 8, seed: int = 42):
    """Native multi-threaded table scan through the SQL surface; returns (seconds, checksum)."""
    cs = C.c_double()
    err = C.create_string_buffer(512)
    sec = lib().infera_sql_bench_scan(function.encode(), model.encode(), rows, ncols, threads, pool_chunks, seed, C.byref(cs),
                                      err, len(err))
    if sec < 0:
        raise SqlError(err.value.decode())
    return sec, cs.value


ROW_GROUP = 122880  # INFERA_SQL_ROW_GROUP


def aligned_empty(n: int, dtype, align: int = 4096, huge: bool = False) -> np.ndarray:
    """n elements of dtype whose first byte sits on an `align` boundary (column stores align their buffers; a 2048-row chunk of a
    page-aligned FLOAT column is exactly two 4 KiB pages, of an unaligned one three).  huge: ask for transparent huge pages (2 MiB)."""
    item = np.dtype(dtype).itemsize
    if huge:
        import mmap

        size = (n * item + (2 << 20) - 1) // (2 << 20) * (2 << 20) + (2 << 20)
        m = mmap.mmap(-1, size)
        try:
            m.madvise(mmap.MADV_HUGEPAGE)
        except (AttributeError, OSError):
            pass
        raw = np.frombuffer(m, np.uint8)
        off = (-raw.ctypes.data) % (2 << 20)
        return raw[off:off + n * item].view(dtype)
    raw = np.empty(n * item + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n * item].view(dtype)


def synth_table(rows: int, ncols: int, seed: int = 42, threads: int = 8, dtype=np.float32, align: int = 0, huge: bool = False) -> np.ndarray:
    """A materialised columnar table in host memory (row groups of 122,880 rows, one contiguous run per column inside a
    group), filled with the generator of SURVEY.md 8d.  Flat array of rows*ncols elements, float32 (FLOAT columns) or
    float64 (DOUBLE columns, the same values widened).  align / huge: see aligned_empty (default: wherever numpy puts it)."""
    n = int(lib().infera_sql_table_floats(rows, ncols))
    t = aligned_empty(n, dtype, align or 4096, huge) if (align or huge) else np.empty(n, dtype)
    if t.dtype == np.float64:
        lib().infera_sql_synth_table_f64(t.ctypes.data, seed, rows, ncols, threads)
    else:
        lib().infera_sql_synth_table(t.ctypes.data, seed, rows, ncols, threads)
    return t


def table_rows(table: np.ndarray, rows: int, ncols: int, r0: int, n: int) -> np.ndarray:
    """Rows [r0, r0+n) of such a table as a row-major [n, ncols] array (tests)."""
    out = np.empty((n, ncols), np.float32)
    for i in range(n):
        r = r0 + i
        g0 = r // ROW_GROUP * ROW_GROUP
        gr = min(ROW_GROUP, rows - g0)
        out[i] = table[g0 * ncols + (r - g0): g0 * ncols + ncols * gr: gr][:ncols]
    return out


def bench_scan_table(function: str, model: str, table: np.ndarray, rows: int, ncols: int, threads: int, reps: int = 1):
    """`reps` complete scans of a materialised table through the SQL surface; returns ([seconds per scan], checksum)."""
    assert table.dtype in (np.float32, np.float64) and table.flags.c_contiguous and table.size >= rows * ncols
    secs = (C.c_double * reps)()
    cs = C.c_double()
    err = C.create_string_buffer(512)
    rc = lib().infera_sql_bench_scan_table_typed(function.encode(), model.encode(), table.ctypes.data, DOUBLE if table.dtype == np.float64 else FLOAT,
                                                 rows, ncols, threads, reps, secs, C.byref(cs), err, len(err))
    if rc != 0:
        raise SqlError(err.value.decode())
    return list(secs), cs.value


def bench_gather_only(table: np.ndarray, rows: int, ncols: int, threads: int, reps: int = 3) -> dict:
    """The host side of the staged scan alone (no GPU): `threads` workers gather every 2048-row chunk of the table into their own buffers with the
    staged path's own routine.  GB/s of gathered features (best rep) and process CPU microseconds per chunk."""
    assert table.dtype == np.float32 and table.flags.c_contiguous and table.size >= rows * ncols
    secs = (C.c_double * reps)()
    cpu = C.c_double()
    if lib().infera_sql_bench_gather_only(table.ctypes.data, rows, ncols, threads, reps, secs, C.byref(cpu)) != 0:
        raise SqlError("gather-only scan failed")
    best = min(secs)
    return {"threads": threads, "gb_per_s": rows * ncols * 4 / best / 1e9, "rows_per_s": rows / best,
            "cpu_us_per_chunk": cpu.value / reps / ((rows + 2047) // 2048) * 1e6, "scan_seconds": list(secs)}


class SegmentTable:
    """The table of synth_table() in DuckDB's SEGMENT shape (round 6): per (row group, column) two separately allocated 256 KiB blocks
    (8-byte block header, 65,534 values each) taken from the DuckDB stand-in's DBConfig::allocator -- the extension's REGISTERING allocator
    when INFERA_ZERO_COPY_ALLOCATOR=1 is set in the environment when the table is made (every block pinned where it lies as it is handed out),
    malloc otherwise.  A chunk's 128 FLAT vectors point into 128 unrelated blocks; the one chunk per row group that straddles two segments is
    assembled in ordinary memory (as DuckDB's scan does).  Blocks go back through the allocator on close()."""

    def __init__(self, rows: int, ncols: int, seed: int = 42, threads: int = 8, shuffled: bool = False, alloc_threads: int = 1):
        """shuffled: every block of the table allocated in a random order (unrelated addresses per chunk whatever the allocator does: parallel
        loads, evictions, reloads); default: a (row group, segment)'s 128 column blocks back to back (a table loaded by one thread -- with the
        extension's ARENA allocator they then lie at one 256 KiB stride inside one registration, which the engine's 2-D copy can fetch)."""
        hooked = C.c_int32()
        t0 = time.perf_counter()
        self.shuffled = shuffled
        # alloc_threads > 1 (not shuffled): that many threads allocate whole (row group, segment) sets concurrently -- a parallel load
        self.handle = lib().infera_stub_segment_table_create(rows, ncols, seed, threads, C.byref(hooked), 0x5EED if shuffled else 0, alloc_threads)
        if not self.handle:
            raise SqlError("segment table: allocation failed")
        self.create_seconds = time.perf_counter() - t0
        self.rows, self.ncols, self.registering_allocator = rows, ncols, bool(hooked.value)
        self.blocks = int(lib().infera_sql_segment_table_blocks(lib().infera_stub_segment_table_get(self.handle)))
        self.assembled_chunks = 0

    def close(self):
        if self.handle:
            lib().infera_stub_segment_table_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


def bench_scan_segments(function: str, model: str, table: SegmentTable, rows: int, ncols: int, threads: int, reps: int = 1):
    """bench_scan_table() over a SegmentTable (same workers, same chunk order, same result consumer); returns ([seconds per scan], checksum)."""
    assert table.handle and rows <= table.rows and ncols == table.ncols
    secs = (C.c_double * reps)()
    cs = C.c_double()
    asm = C.c_uint64()
    err = C.create_string_buffer(512)
    rc = lib().infera_sql_bench_scan_segments(function.encode(), model.encode(), lib().infera_stub_segment_table_get(table.handle), rows, threads, reps,
                                              secs, C.byref(cs), C.byref(asm), err, len(err))
    if rc != 0:
        raise SqlError(err.value.decode())
    table.assembled_chunks = asm.value
    return list(secs), cs.value


def bench_last_times() -> tuple[int, int]:
    """(ns inside infera_sql_call, ns inside the worker loops) of the last bench_scan_table, summed over threads and reps."""
    a, b = C.c_uint64(), C.c_uint64()
    lib().infera_sql_bench_last_times(C.byref(a), C.byref(b))
    return a.value, b.value


def bench_last_cpu() -> tuple[float, float, float]:
    """(process CPU seconds, of which system, wall seconds) of the last bench_scan_table call, summed over its reps."""
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().infera_sql_bench_last_cpu(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def phase_breakdown(fn, *args, **kw):
    """Runs fn (a bench_scan_table call) and returns (its result, per-chunk microseconds by phase): the engine's host-path
    phases (infera_hip_get_devices) plus what the binding layer and the scan loop add around them."""
    from . import capi

    before = capi.get_devices()["host_phases"]
    out = fn(*args, **kw)
    after = capi.get_devices()["host_phases"]
    n = max(1, after["passes"] - before["passes"])
    ph = {k[:-3]: (after[k] - before[k]) / n / 1e3 for k in after if k.endswith("_ns")}
    call_ns, thread_ns = bench_last_times()
    ph["binding_layer"] = call_ns / n / 1e3 - sum(ph.values())
    ph["scan_loop"] = (thread_ns - call_ns) / n / 1e3
    ph["chunks"] = n
    cpu_s, sys_s, wall_s = bench_last_cpu()
    ph["cpu_us_per_chunk"] = cpu_s / n * 1e6          # process CPU time (user + sys, every thread) per chunk
    ph["sys_us_per_chunk"] = sys_s / n * 1e6
    ph["cpus_busy"] = cpu_s / wall_s if wall_s > 0 else 0.0
    return out, {k: round(v, 1) for k, v in ph.items()}


def bench_blob_scan(model: str, blobs: np.ndarray, blob_bytes: int, rows: int, threads: int, reps: int = 1):
    """`reps` scans of `rows` BLOB rows (cycling over the blobs held in `blobs`, a contiguous byte/float array) through
    infera_predict_from_blob in 2048-row chunks; returns ([seconds per scan], checksum)."""
    assert blobs.flags.c_contiguous and blobs.nbytes % blob_bytes == 0
    secs = (C.c_double * reps)()
    cs = C.c_double()
    err = C.create_string_buffer(512)
    rc = lib().infera_sql_bench_blob_scan(model.encode(), blobs.ctypes.data, blobs.nbytes // blob_bytes, blob_bytes, rows, threads, reps,
                                          secs, C.byref(cs), err, len(err))
    if rc != 0:
        raise SqlError(err.value.decode())
    return list(secs), cs.value
