"""Seeded tile generators shared by the CPU and GPU tests."""
import zlib

import numpy as np

NULL = -(2 ** 31)

KINDS = ["ramp", "smooth", "noise8", "noise16", "noise32", "uniform", "extremes", "steps", "sparse_big"]


def make_tile(kind, n_rows, n_cols, seed=0):
    # (a stable hash of the kind: Python's own string hash changes from process to process)
    rng = np.random.default_rng((zlib.crc32(kind.encode()) & 0xFFFF) * 7919 + n_rows * 131 + n_cols + seed)
    n = n_rows * n_cols
    if kind == "ramp":
        return (np.arange(n, dtype=np.int64) - 1).astype(np.int32)
    if kind == "smooth":
        r = np.arange(n_rows)[:, None]
        c = np.arange(n_cols)[None, :]
        return (1000 * np.sin(r / 7.0) * np.cos(c / 5.0) + rng.integers(-3, 4, (n_rows, n_cols))).astype(np.int32).ravel()
    if kind == "noise8":
        return rng.integers(-100, 101, n).astype(np.int32)
    if kind == "noise16":
        return rng.integers(-32768, 32768, n).astype(np.int32)
    if kind == "noise32":
        return rng.integers(-2 ** 31 + 1, 2 ** 31, n, dtype=np.int64).astype(np.int32)
    if kind == "uniform":
        return np.full(n, 77, np.int32)
    if kind == "extremes":
        v = np.zeros(n, np.int32)
        v[::2] = 2 ** 31 - 1
        v[1::2] = -(2 ** 31) + 1
        return v
    if kind == "steps":
        return (np.arange(n) // 37 * 300).astype(np.int32)
    if kind == "sparse_big":
        v = rng.integers(-2, 3, n).astype(np.int64).cumsum()
        idx = rng.integers(0, n, max(1, n // 50))
        v[idx] += rng.integers(-3000000, 3000000, idx.size)
        return v.astype(np.int32)
    raise ValueError(kind)


def add_nulls(v, n_rows, n_cols, frac, seed=0, blocks=False):
    rng = np.random.default_rng(seed + 17)
    v = v.copy()
    if blocks:
        m = np.zeros((n_rows, n_cols), bool)
        for _ in range(max(1, int(frac * 10))):
            r0, c0 = rng.integers(0, n_rows), rng.integers(0, n_cols)
            m[r0:r0 + max(1, n_rows // 4), c0:c0 + max(1, n_cols // 3)] = True
        v[m.ravel()] = NULL
    else:
        v[rng.random(v.size) < frac] = NULL
    return v
