"""ctypes wrapper around the CPU oracle (oracle/infera_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under infera_amd/ does.  PARITY PIN STATUS: see infera_oracle.h (pinned on the
reference's MatMul+Add and Identity golden values and error strings; unpinned elsewhere).

Besides the engine mirror this module restates, in Python, the value formatting the reference's
DuckDB binding applies above the C ABI (infera_extension.cpp:199-227, 275-284, 397-416, 451-459)
so binding-level parity tests have a checker too.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libinfera_oracle.so")


class OracleError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("infera_oracle.c", "infera_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libinfera_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


class _Result(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("len", C.c_size_t), ("rows", C.c_size_t), ("cols", C.c_size_t)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_load.restype = C.c_void_p
        L.orc_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.orc_free_model.argtypes = [C.c_void_p]
        L.orc_input_rank.argtypes = [C.c_void_p]
        L.orc_output_rank.argtypes = [C.c_void_p]
        L.orc_input_shape.restype = C.POINTER(C.c_int64)
        L.orc_input_shape.argtypes = [C.c_void_p]
        L.orc_output_shape.restype = C.POINTER(C.c_int64)
        L.orc_output_shape.argtypes = [C.c_void_p]
        L.orc_shape_rows_cols.argtypes = [C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(_Result), C.c_char_p, C.c_size_t]
        L.orc_predict_blob.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(_Result), C.c_char_p, C.c_size_t]
        L.orc_free_result.argtypes = [C.POINTER(_Result)]
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_synth_fill_rowmajor.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_bench_scan.restype = C.c_double
        L.orc_bench_scan.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_set_blocked_gemm.restype = None
        L.orc_set_blocked_gemm.argtypes = [C.c_int]
        L.orc_bench_scan_table.restype = C.c_double
        L.orc_bench_scan_table.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def shape_rows_cols(shape: Sequence[int]) -> tuple[int, int]:
    arr = (C.c_size_t * max(len(shape), 1))(*shape)
    r, c = C.c_size_t(), C.c_size_t()
    lib().orc_shape_rows_cols(arr, len(shape), C.byref(r), C.byref(c))
    return r.value, c.value


class Model:
    """engine.rs OnnxModel mirror: .input_shape/.output_shape with -1 for symbolic dims."""

    def __init__(self, path: str):
        err = C.create_string_buffer(512)
        self._h = lib().orc_load(os.fsencode(path), err, len(err))
        if not self._h:
            raise OracleError(err.value.decode(errors="replace"))
        self.input_shape = [lib().orc_input_shape(self._h)[i] for i in range(lib().orc_input_rank(self._h))]
        self.output_shape = [lib().orc_output_shape(self._h)[i] for i in range(lib().orc_output_rank(self._h))]

    def close(self):
        if self._h:
            lib().orc_free_model(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _take(self, res: _Result) -> np.ndarray:
        out = np.ctypeslib.as_array(res.data, shape=(res.len,)).copy() if res.len else np.zeros(0, np.float32)
        rows, cols = res.rows, res.cols
        lib().orc_free_result(C.byref(res))
        return out.reshape(rows, cols) if rows * cols == out.size else out

    def predict(self, x: np.ndarray) -> np.ndarray:
        """run_inference_impl: x is [rows, cols] f32 row-major; returns [rows_out, cols_out]."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        rows, cols = x.shape
        res, err = _Result(), C.create_string_buffer(512)
        rc = lib().orc_predict(self._h, x.ctypes.data, rows, cols, C.byref(res), err, len(err))
        if rc:
            raise OracleError(err.value.decode(errors="replace"))
        return self._take(res)

    def predict_blob(self, blob: bytes) -> np.ndarray:
        buf = (C.c_uint8 * max(len(blob), 1)).from_buffer_copy(blob if blob else b"\0")
        res, err = _Result(), C.create_string_buffer(512)
        rc = lib().orc_predict_blob(self._h, buf, len(blob), C.byref(res), err, len(err))
        if rc:
            raise OracleError(err.value.decode(errors="replace"))
        return self._take(res)

    def bench_scan(self, rows: int, ncols: int, seed: int = 42, threads: int = 1, chunk_rows: int = 2048,
                   boxed: bool | int = True) -> tuple[float, float]:
        cs = C.c_double()
        sec = lib().orc_bench_scan(self._h, rows, ncols, seed, threads, chunk_rows, int(boxed), C.byref(cs))
        if sec < 0:
            raise OracleError("bench scan failed")
        return sec, cs.value


def set_blocked_gemm(on: bool) -> None:
    """Test hook: later predict() calls on THIS thread use the register-blocked GEMM of the bench's best-CPU leg (bit-identical)."""
    lib().orc_set_blocked_gemm(int(on))


def bench_scan_table(model: "Model", table: np.ndarray, rows: int, ncols: int, threads: int = 1, chunk_rows: int = 2048,
                     boxed: bool | int = True) -> tuple[float, float]:
    """Scan of a materialised columnar table (row groups of 122,880 rows, see infera_oracle.h); (seconds, checksum).
    boxed: True / 1 = reference-shaped (per-cell boxed gather, plain GEMM loop), False / 0 = plain strided gather,
    2 = best CPU (tiled gather + register-blocked AVX-512 / AVX2 GEMM, bit-identical results)."""
    assert table.dtype == np.float32 and table.flags.c_contiguous and table.size >= rows * ncols
    cs = C.c_double()
    sec = lib().orc_bench_scan_table(model._h, table.ctypes.data, rows, ncols, threads, chunk_rows, int(boxed), C.byref(cs))
    if sec < 0:
        raise OracleError("bench scan failed")
    return sec, cs.value


def synth_table(seed: int, row0: int, rows: int, ncols: int) -> np.ndarray:
    out = np.empty((rows, ncols), np.float32)
    lib().orc_synth_fill_rowmajor(out.ctypes.data, seed, row0, rows, ncols)
    return out


# ----------------------------------------------------------------------------------------------
# Binding-level restatement (what the C++ DuckDB glue does around the C ABI)
# ----------------------------------------------------------------------------------------------

def extract_features(columns: Sequence[np.ndarray]) -> np.ndarray:
    """ExtractFeatures (infera_extension.cpp:199-227): typed columns -> row-major f32 [rows, F].
    FLOAT stays; DOUBLE/INTEGER/BIGINT are static_cast<float> (round-to-nearest-even)."""
    cols = []
    for c in columns:
        if isinstance(c, np.ma.MaskedArray) and c.mask.any():
            raise OracleError("Feature values cannot be NULL")
        a = np.asarray(c)
        if a.dtype not in (np.float32, np.float64, np.int32, np.int64):
            raise OracleError(f"Unsupported feature type: {a.dtype}")
        cols.append(a.astype(np.float32))
    return np.ascontiguousarray(np.stack(cols, axis=1))


def format_g(v: float) -> str:
    """C++ `ostream << float` with default flags == printf("%g") (infera_extension.cpp:405-416)."""
    return "%g" % float(np.float32(v))


def predict_multi_json(out: np.ndarray) -> list[str]:
    """PredictMulti row formatting: "[a,b,c]" with %g and no spaces."""
    return ["[" + ",".join(format_g(v) for v in row) + "]" for row in np.atleast_2d(out)]


def check_predict_shape(out_rows: int, out_cols: int, batch: int) -> None:
    """Predict's post-condition (infera_extension.cpp:275-279)."""
    if out_rows != batch or out_cols != 1:
        raise OracleError(f"Model output shape mismatch. Expected ({batch}, 1), but got ({out_rows}, {out_cols}).")
