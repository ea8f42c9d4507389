"""The staged host path's gather step (host/gather.cpp gather_column_major: every column run copied / converted as it lies into one
column-major f32 chunk) checked WITHOUT a GPU: four FLOAT (or DOUBLE) runs in lockstep, 512 bytes of each in turn, every other column
kind run by run.  Every cell must be static_cast<float> of the source (the reference's ExtractFeatures casts, infera_extension.cpp:211-222).
Runs in a child process (a clean library instance)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, ROOT)
from infera_amd import capi
rng = np.random.default_rng(7)
checked = 0
for ncols in (1, 2, 3, 4, 5, 9, 13, 128):
    for total, row0, nrows in ((1, 0, 1), (40, 3, 7), (300, 1, 127), (300, 0, 128), (2500, 5, 2048), (2500, 449, 2051), (700, 2, 517)):
        cols, want = [], []
        for c in range(ncols):
            kind = (c * 7 + ncols) % 11
            if kind < 5 or ncols == 128 and c % 16:   # mostly FLOAT (long runs of it: the interleaved loop), the rest mixed in
                a = rng.standard_normal(total).astype(np.float32)
            elif kind < 7:
                a = rng.standard_normal(total) * (1 + 1e-9)            # DOUBLE values that round on the way to f32
            elif kind == 7:
                a = rng.integers(-2**31, 2**31 - 1, total, dtype=np.int64).astype(np.int32)
            elif kind == 8:
                a = rng.integers(-2**62, 2**62, total, dtype=np.int64)
            elif kind == 9:
                a = np.array([rng.standard_normal()], np.float64)       # a CONSTANT_VECTOR (unless the chunk has one row)
            else:
                a = rng.standard_normal(total).astype(np.float32)
            cols.append(a)
            src = a if len(a) > 1 or total == 1 else np.repeat(a, total)
            want.append(src[row0:row0 + nrows].astype(np.float32))
        got = capi.gather_columns_colmajor(cols, rows=total, row0=row0, nrows=nrows)
        w = np.stack(want)
        assert got.shape == w.shape and np.array_equal(got.view(np.uint32), w.view(np.uint32)), (ncols, total, row0, nrows)
        checked += 1
print("GATHER-OK", checked)
'''



def test_column_major_gather_gives_static_cast_bytes(built):
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0 and "GATHER-OK 56" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
