"""ctypes binding of libinfera.so (include/infera.h + include/infera_hip.h).

This is the host-side mirror used by the tests, bench.py and __graft_entry__: function names,
argument meaning and error behaviour are those of the reference's C ABI
(/root/reference infera/bindings/include/rust.h), so the parity tests read like the reference's own
Rust unit tests (infera/src/lib.rs:427-657).  Nothing here computes anything: every call goes
straight into the shared library, and if the library (or a GPU) is missing the call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INFERA_LIB_PATH") or os.path.join(_HERE, "libinfera.so")  # override: A/B builds (tools/)

REFERENCE_SYMBOLS = [
    "infera_load_model", "infera_unload_model", "infera_predict", "infera_predict_from_blob",
    "infera_get_model_info", "infera_get_loaded_models", "infera_get_version", "infera_clear_cache",
    "infera_get_cache_info", "infera_set_autoload_dir", "infera_last_error", "infera_free", "infera_free_result",
]
EXTENSION_SYMBOLS = [
    "infera_hip_device_count", "infera_hip_device_ordinal", "infera_hip_get_devices", "infera_hip_get_plan",
    "infera_hip_predict_device", "infera_hip_sync", "infera_hip_time_predict_device", "infera_hip_malloc",
    "infera_hip_free", "infera_hip_memcpy_h2d", "infera_hip_memcpy_d2h", "infera_hip_synth_fill",
    "infera_predict_into", "infera_predict_columns", "infera_predict_from_blob_batch", "infera_gather_columns",
    "infera_hip_sha256_hex", "infera_hip_shape_rows_cols", "infera_hip_h2d_probe", "infera_hip_choose_slot",
    "infera_gather_columns_colmajor", "infera_hip_choose_slot_balanced", "infera_hip_register_host_memory", "infera_hip_unregister_host_memory", "infera_hip_zero_copy_calls",
]


class InferaInferenceResult(C.Structure):
    """rust.h:28-49 -- 40 bytes on LP64, passed and returned by value."""
    _fields_ = [("data", C.POINTER(C.c_float)), ("len", C.c_size_t), ("rows", C.c_size_t), ("cols", C.c_size_t),
                ("status", C.c_int32)]


class InferaColumn(C.Structure):
    _fields_ = [("data", C.c_void_p), ("validity", C.POINTER(C.c_uint64)), ("type", C.c_int32), ("is_constant", C.c_int32)]


COL_FLOAT, COL_DOUBLE, COL_INTEGER, COL_BIGINT = 0, 1, 2, 3
_NP_TO_COL = {np.dtype(np.float32): COL_FLOAT, np.dtype(np.float64): COL_DOUBLE, np.dtype(np.int32): COL_INTEGER,
              np.dtype(np.int64): COL_BIGINT}


class InferaError(RuntimeError):
    pass


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise InferaError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(there is no fallback implementation)")
    L = C.CDLL(path)
    c_char_pp = C.c_void_p  # returned char* must stay a raw pointer so infera_free gets the same address
    L.infera_load_model.argtypes = [C.c_char_p, C.c_char_p]
    L.infera_load_model.restype = C.c_int32
    L.infera_unload_model.argtypes = [C.c_char_p]
    L.infera_unload_model.restype = C.c_int32
    L.infera_predict.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t]
    L.infera_predict.restype = InferaInferenceResult
    L.infera_predict_from_blob.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    L.infera_predict_from_blob.restype = InferaInferenceResult
    for fn in ("infera_get_model_info", "infera_set_autoload_dir", "infera_hip_get_plan"):
        getattr(L, fn).argtypes = [C.c_char_p]
        getattr(L, fn).restype = c_char_pp
    for fn in ("infera_get_loaded_models", "infera_get_version", "infera_get_cache_info", "infera_hip_get_devices"):
        getattr(L, fn).argtypes = []
        getattr(L, fn).restype = c_char_pp
    L.infera_clear_cache.argtypes = []
    L.infera_clear_cache.restype = C.c_int32
    L.infera_last_error.argtypes = []
    L.infera_last_error.restype = C.c_char_p
    L.infera_free.argtypes = [C.c_void_p]
    L.infera_free.restype = None
    L.infera_free_result.argtypes = [InferaInferenceResult]
    L.infera_free_result.restype = None
    # extensions
    L.infera_hip_device_count.restype = C.c_int32
    L.infera_hip_device_ordinal.argtypes = [C.c_int32]
    L.infera_hip_device_ordinal.restype = C.c_int32
    L.infera_hip_predict_device.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.infera_hip_predict_device.restype = C.c_int32
    L.infera_hip_sync.argtypes = [C.c_int32]
    L.infera_hip_sync.restype = C.c_int32
    L.infera_hip_time_predict_device.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                                 C.c_uint64, C.c_int32, C.POINTER(C.c_float)]
    L.infera_hip_time_predict_device.restype = C.c_int32
    L.infera_hip_malloc.argtypes = [C.c_int32, C.c_uint64]
    L.infera_hip_malloc.restype = C.c_void_p
    L.infera_hip_free.argtypes = [C.c_int32, C.c_void_p]
    L.infera_hip_free.restype = C.c_int32
    L.infera_hip_memcpy_h2d.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64]
    L.infera_hip_memcpy_h2d.restype = C.c_int32
    L.infera_hip_memcpy_d2h.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64]
    L.infera_hip_memcpy_d2h.restype = C.c_int32
    L.infera_hip_synth_fill.argtypes = [C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    L.infera_hip_synth_fill.restype = C.c_int32
    L.infera_predict_into.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.infera_predict_into.restype = C.c_int32
    L.infera_predict_columns.argtypes = [C.c_char_p, C.POINTER(InferaColumn), C.c_size_t, C.c_size_t]
    L.infera_predict_columns.restype = InferaInferenceResult
    L.infera_gather_columns.argtypes = [C.POINTER(InferaColumn), C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    L.infera_gather_columns.restype = C.c_int32
    L.infera_gather_columns_colmajor.argtypes = [C.POINTER(InferaColumn), C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    L.infera_gather_columns_colmajor.restype = C.c_int32
    L.infera_predict_from_blob_batch.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t]
    L.infera_predict_from_blob_batch.restype = InferaInferenceResult
    L.infera_hip_shape_rows_cols.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.infera_hip_shape_rows_cols.restype = None
    L.infera_hip_choose_slot.argtypes = [C.POINTER(C.c_int32), C.c_size_t, C.c_int32, C.c_uint64, C.c_uint64]
    L.infera_hip_choose_slot.restype = C.c_int32
    L.infera_hip_choose_slot_balanced.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_size_t, C.c_int32]
    L.infera_hip_choose_slot_balanced.restype = C.c_int32
    L.infera_hip_register_host_memory.argtypes = [C.c_void_p, C.c_uint64]
    L.infera_hip_register_host_memory.restype = C.c_int32
    L.infera_hip_unregister_host_memory.argtypes = [C.c_void_p]
    L.infera_hip_unregister_host_memory.restype = C.c_int32
    L.infera_hip_zero_copy_calls.argtypes = []
    L.infera_hip_zero_copy_calls.restype = C.c_uint64
    L.infera_hip_h2d_probe.argtypes = [C.c_int32, C.c_uint64, C.c_int32, C.c_int32]
    L.infera_hip_h2d_probe.restype = C.c_double
    _lib = L
    return L


def last_error() -> str | None:
    p = load_library().infera_last_error()
    return p.decode() if p else None


def _take_str(ptr) -> str:
    L = load_library()
    if not ptr:
        raise InferaError("NULL string returned")
    s = C.string_at(ptr).decode()
    L.infera_free(ptr)
    return s


def _enc(s: str | bytes | None):
    if s is None:
        return None
    return s if isinstance(s, bytes) else s.encode()


def _take_result(res: InferaInferenceResult, what: str) -> np.ndarray:
    L = load_library()
    if res.status != 0:
        L.infera_free_result(res)  # callers free even on failure (infera_extension.cpp:271-273)
        raise InferaError(last_error() or f"{what} failed")
    out = np.ctypeslib.as_array(res.data, shape=(res.len,)).copy() if res.len else np.zeros(0, np.float32)
    rows, cols = res.rows, res.cols
    L.infera_free_result(res)
    return out.reshape(rows, cols) if rows * cols == out.size else out


# ---- the 13 reference functions ----------------------------------------------------------------

def load_model(name: str, path: str) -> None:
    if load_library().infera_load_model(_enc(name), _enc(path)) != 0:
        raise InferaError(last_error())


def unload_model(name: str) -> None:
    if load_library().infera_unload_model(_enc(name)) != 0:
        raise InferaError(last_error())


def predict(name: str, x: np.ndarray) -> np.ndarray:
    """infera_predict: x is [rows, cols] f32 row-major in host memory; returns [rows_out, cols_out]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim != 2:
        raise ValueError("x must be 2-D")
    res = load_library().infera_predict(_enc(name), x.ctypes.data, x.shape[0], x.shape[1])
    return _take_result(res, "infera_predict")


def predict_from_blob(name: str, blob: bytes) -> np.ndarray:
    # (no copy: a pointer into the bytes object, which outlives the call -- a 38 MB batch used to spend 9 of its 13 ms in
    # create_string_buffer)
    ptr = C.cast(C.c_char_p(blob if len(blob) else b"\0"), C.c_void_p)
    res = load_library().infera_predict_from_blob(_enc(name), ptr, len(blob))
    return _take_result(res, "infera_predict_from_blob")


def get_model_info(name: str) -> dict:
    return json.loads(_take_str(load_library().infera_get_model_info(_enc(name))))


def get_loaded_models() -> list[str]:
    return json.loads(_take_str(load_library().infera_get_loaded_models()))


def get_version() -> dict:
    return json.loads(_take_str(load_library().infera_get_version()))


def clear_cache() -> None:
    if load_library().infera_clear_cache() != 0:
        raise InferaError(last_error())


def get_cache_info() -> dict:
    return json.loads(_take_str(load_library().infera_get_cache_info()))


def set_autoload_dir(path: str) -> dict:
    return json.loads(_take_str(load_library().infera_set_autoload_dir(_enc(path))))


# ---- additive MI355X entry points ----------------------------------------------------------------

def shape_rows_cols(shape: Sequence[int]) -> tuple[int, int]:
    """The product's engine.rs:19-29 rule (what every result's rows/cols come from)."""
    arr = (C.c_uint64 * max(len(shape), 1))(*shape)
    r, c = C.c_uint64(), C.c_uint64()
    load_library().infera_hip_shape_rows_cols(arr, len(shape), C.byref(r), C.byref(c))
    return r.value, c.value


def h2d_probe(device: int, nbytes: int = 8 << 20, iters: int = 64, threads: int = 4) -> float:
    """GB/s of plain pinned hipMemcpyAsync H2D on this box (the host link's practical ceiling)."""
    return float(load_library().infera_hip_h2d_probe(device, nbytes, iters, threads))


def choose_slot(slot_numa: Sequence[int], thread_node: int, ticket_on_node: int, ticket_global: int) -> int:
    """The thread -> device-slot dealing policy (NUMA-local slots first), as the library applies it."""
    arr = (C.c_int32 * max(len(slot_numa), 1))(*slot_numa)
    return int(load_library().infera_hip_choose_slot(arr, len(slot_numa), thread_node, ticket_on_node, ticket_global))


def register_host_memory(arr: np.ndarray) -> None:
    """Registers a (contiguous) array's bytes for the zero-copy host path; keep the array alive until unregister_host_memory."""
    assert arr.flags.c_contiguous
    if load_library().infera_hip_register_host_memory(arr.ctypes.data, arr.nbytes) != 0:
        raise InferaError(last_error())


def unregister_host_memory(arr: np.ndarray) -> None:
    if load_library().infera_hip_unregister_host_memory(arr.ctypes.data) != 0:
        raise InferaError(last_error())


def zero_copy_calls() -> int:
    return int(load_library().infera_hip_zero_copy_calls())


def choose_slot_balanced(slot_numa: Sequence[int], slot_threads: Sequence[int], thread_node: int) -> int:
    """The load-aware dealing the library applies to a caller thread's first call (NUMA-local first, bounded by load)."""
    n = len(slot_numa)
    a = (C.c_int32 * max(n, 1))(*slot_numa)
    b = (C.c_int32 * max(n, 1))(*slot_threads)
    return int(load_library().infera_hip_choose_slot_balanced(a, b, n, thread_node))


def device_count() -> int:
    return int(load_library().infera_hip_device_count())


def device_ordinal(i: int) -> int:
    return int(load_library().infera_hip_device_ordinal(i))


def get_devices() -> dict:
    return json.loads(_take_str(load_library().infera_hip_get_devices()))


def get_plan(name: str) -> dict:
    return json.loads(_take_str(load_library().infera_hip_get_plan(_enc(name))))


class DeviceBuffer:
    """A chunk of HBM owned through the library's allocator helpers."""

    def __init__(self, device: int, nbytes: int):
        self.device, self.nbytes = device, nbytes
        self.ptr = load_library().infera_hip_malloc(device, nbytes)
        if not self.ptr:
            raise InferaError(last_error() or "hipMalloc failed")

    def upload(self, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        if load_library().infera_hip_memcpy_h2d(self.device, self.ptr, arr.ctypes.data, arr.nbytes) != 0:
            raise InferaError(last_error())
        return self

    def download(self, shape, dtype=np.float32, offset_bytes: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert offset_bytes + out.nbytes <= self.nbytes
        if load_library().infera_hip_memcpy_d2h(self.device, out.ctypes.data, self.ptr + offset_bytes, out.nbytes) != 0:
            raise InferaError(last_error())
        return out

    def free(self):
        if self.ptr:
            load_library().infera_hip_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synth_fill(buf: DeviceBuffer, seed: int, row0: int, rows: int, cols: int, offset_bytes: int = 0) -> None:
    if load_library().infera_hip_synth_fill(buf.device, buf.ptr + offset_bytes, seed, row0, rows, cols) != 0:
        raise InferaError(last_error())


def predict_device(name: str, d_in: DeviceBuffer, rows: int, cols: int, d_out: DeviceBuffer, sync: bool = True,
                   in_offset_bytes: int = 0, out_offset_bytes: int = 0) -> tuple[int, int]:
    L = load_library()
    r, c = C.c_uint64(), C.c_uint64()
    cap = (d_out.nbytes - out_offset_bytes) // 4
    if L.infera_hip_predict_device(_enc(name), d_in.device, d_in.ptr + in_offset_bytes, rows, cols,
                                   d_out.ptr + out_offset_bytes, cap, C.byref(r), C.byref(c)) != 0:
        raise InferaError(last_error())
    if sync and L.infera_hip_sync(d_in.device) != 0:
        raise InferaError(last_error())
    return r.value, c.value


def sync(device: int) -> None:
    if load_library().infera_hip_sync(device) != 0:
        raise InferaError(last_error())


def time_predict_device(name: str, d_in: DeviceBuffer, rows: int, cols: int, d_out: DeviceBuffer, iters: int) -> float:
    """Elapsed milliseconds (HIP events on the launching stream) for `iters` back-to-back passes."""
    ms = C.c_float()
    if load_library().infera_hip_time_predict_device(_enc(name), d_in.device, d_in.ptr, rows, cols, d_out.ptr,
                                                     d_out.nbytes // 4, iters, C.byref(ms)) != 0:
        raise InferaError(last_error())
    return float(ms.value)


def predict_into(name: str, x: np.ndarray, out: np.ndarray) -> tuple[int, int]:
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous
    r, c = C.c_uint64(), C.c_uint64()
    if load_library().infera_predict_into(_enc(name), x.ctypes.data, x.shape[0], x.shape[1], out.ctypes.data, out.size,
                                          C.byref(r), C.byref(c)) != 0:
        raise InferaError(last_error())
    return r.value, c.value


def _make_columns(columns, rows, validity):
    n = len(columns)
    cols = (InferaColumn * max(n, 1))()
    keep = []
    if rows is None:
        rows = max(len(c) for c in columns)
    for i, c in enumerate(columns):
        a = np.ascontiguousarray(c)
        keep.append(a)
        cols[i].data = a.ctypes.data
        cols[i].type = _NP_TO_COL[a.dtype]
        cols[i].is_constant = int(len(a) == 1 and rows != 1)
        v = validity[i] if validity is not None else None
        if v is not None:
            v = np.ascontiguousarray(v, dtype=np.uint64)
            keep.append(v)
            cols[i].validity = v.ctypes.data_as(C.POINTER(C.c_uint64))
    return cols, n, rows, keep


def predict_columns(name: str, columns: Sequence[np.ndarray], rows: int | None = None,
                    validity: Sequence[np.ndarray | None] | None = None) -> np.ndarray:
    """Columnar gather path: each column a flat typed vector (float32/float64/int32/int64);
    a length-1 column is a CONSTANT_VECTOR."""
    cols, n, rows, _keep = _make_columns(columns, rows, validity)
    res = load_library().infera_predict_columns(_enc(name), cols, n, rows)
    return _take_result(res, "infera_predict_columns")


def gather_columns(columns: Sequence[np.ndarray], rows: int | None = None, row0: int = 0, nrows: int | None = None,
                   validity: Sequence[np.ndarray | None] | None = None) -> np.ndarray:
    """The gather step alone (CPU): typed columns -> row-major f32 [nrows, ncols]."""
    cols, n, rows, _keep = _make_columns(columns, rows, validity)
    nrows = rows - row0 if nrows is None else nrows
    out = np.empty((nrows, n), np.float32)
    if load_library().infera_gather_columns(cols, n, row0, nrows, out.ctypes.data) != 0:
        raise InferaError(last_error())
    return out


def gather_columns_colmajor(columns: Sequence[np.ndarray], rows: int | None = None, row0: int = 0, nrows: int | None = None) -> np.ndarray:
    """The staged path's gather step alone (CPU): typed columns -> ONE column-major f32 chunk [ncols, nrows]."""
    cols, n, rows, _keep = _make_columns(columns, rows, None)
    nrows = rows - row0 if nrows is None else nrows
    out = np.empty((n, nrows), np.float32)
    if load_library().infera_gather_columns_colmajor(cols, n, row0, nrows, out.ctypes.data) != 0:
        raise InferaError(last_error())
    return out


def predict_from_blob_batch(name: str, blobs: Sequence[bytes]) -> np.ndarray:
    n = len(blobs)
    keep = [b if len(b) else b"\0" for b in blobs]  # (pointers into the bytes objects: no copies)
    ptrs = (C.c_void_p * max(n, 1))(*[C.cast(C.c_char_p(b), C.c_void_p).value for b in keep])
    lens = (C.c_size_t * max(n, 1))(*[len(b) for b in blobs])
    res = load_library().infera_predict_from_blob_batch(_enc(name), ptrs, lens, n)
    return _take_result(res, "infera_predict_from_blob_batch")
