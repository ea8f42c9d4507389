"""Counter-based synthetic table generator (SURVEY.md section 8d / BASELINE.md section 4).

    u = splitmix64(seed ^ (row*F + col));  x = ((u >> 40) * 2^-24) * 2 - 1     (f32, in [-1, 1))

The same function exists as a HIP fill kernel (csrc/hip/eltwise.hip: synth_fill_kernel) and in the C oracle
(oracle/infera_oracle.c: orc_synth_value); all three agree bit-for-bit because every
intermediate is exactly representable in f32.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def uniform_pm1(seed: int, counter: np.ndarray) -> np.ndarray:
    """f32 uniform in [-1, 1) for each 64-bit counter."""
    u = splitmix64(np.uint64(seed) ^ np.asarray(counter, dtype=np.uint64))
    m = (u >> np.uint64(40)).astype(np.float32)
    return m * np.float32(2.0 ** -24) * np.float32(2.0) - np.float32(1.0)


def table(seed: int, row0: int, rows: int, ncols: int) -> np.ndarray:
    """Row-major [rows, ncols] f32 slice of the synthetic table starting at absolute row `row0`."""
    r = np.arange(row0, row0 + rows, dtype=np.uint64)[:, None]
    c = np.arange(ncols, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        ctr = r * np.uint64(ncols) + c
    return uniform_pm1(seed, ctr)
