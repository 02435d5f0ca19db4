"""CPU checks of the host/device-shared headers of the HIP codec (gridfour_amd/csrc/*.h):
the O(1)-per-pop tree construction must give exactly the tree of the reference's linked-list
algorithm (oracle), the M32 byte helpers must equal CodecM32, and the stream-order map must
equal the order in which the predictors emit residuals."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hh():
    src = os.path.join(HERE, "csrc", "host_harness.cpp")
    so = os.path.join(HERE, "csrc", "libhost_harness.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, src])
    L = C.CDLL(so)
    u8p = C.POINTER(C.c_uint8)
    L.hh_huffman_encode.argtypes = [u8p, C.c_size_t, C.POINTER(C.c_size_t), u8p, C.c_size_t, u8p]
    L.hh_huffman_encode_rounds.argtypes = [u8p, C.c_size_t, C.POINTER(C.c_size_t), u8p, C.c_size_t, u8p]
    L.hh_stream_cell.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
    L.hh_stream_cell.restype = C.c_uint32
    L.hh_m32_len.argtypes = [C.c_uint32]
    L.hh_m32_byte.argtypes = [C.c_uint32, C.c_int, C.c_int]
    L.hh_m32_byte.restype = C.c_uint32
    return L


def _hh_encode(hh, symbols, fn="hh_huffman_encode"):
    s = np.ascontiguousarray(symbols, np.uint8)
    cap = 400 + 40 * s.size
    buf = np.zeros(cap, np.uint8)
    pos = C.c_size_t(3)        # deliberately unaligned start
    cl = np.zeros(256, np.uint8)
    rc = getattr(hh, fn)(buf.ctypes.data_as(C.POINTER(C.c_uint8)), cap * 8, C.byref(pos),
                              s.ctypes.data_as(C.POINTER(C.c_uint8)), s.size,
                              cl.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0
    return bytes(buf[:(pos.value + 7) // 8]), pos.value, cl


def _symbol_sets():
    rng = np.random.default_rng(1234)
    yield "two", np.array([5, 9, 9], np.uint8)
    yield "one", np.full(17, 200, np.uint8)
    yield "all-equal-256", np.tile(np.arange(256, dtype=np.uint8), 3)
    yield "all-equal-7", np.tile(np.arange(7, dtype=np.uint8) * 31, 5)
    yield "powers", np.repeat(np.arange(12, dtype=np.uint8), 2 ** np.arange(12))
    yield "fib", np.repeat(np.arange(16, dtype=np.uint8) + 100,
                           [1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987])
    for k in range(40):
        n_sym = int(rng.integers(2, 257))
        # small counts -> many ties between leaves and between branches
        hi = int(rng.choice([2, 3, 5, 20, 1000]))
        counts = rng.integers(1, hi + 1, n_sym)
        syms = rng.permutation(256)[:n_sym].astype(np.uint8)
        data = np.repeat(syms, counts)
        rng.shuffle(data)
        yield "rand%d" % k, data
    # geometric, DEM-like
    g = np.clip(np.round(rng.laplace(0, 6, 20000)), -126, 126).astype(np.int8).view(np.uint8)
    yield "laplace", g


@pytest.mark.parametrize("name,data", list(_symbol_sets()), ids=lambda x: x if isinstance(x, str) else "")
def test_tree_matches_reference_algorithm(hh, name, data):
    got, pos, cl = _hh_encode(hh, data)
    ref, rpos, rcl, _ = oracle.huffman_encode(data, bit_pos=3, prefix=b"\x00")
    assert pos == rpos
    assert np.array_equal(cl, rcl)
    assert got == ref
    back, _ = oracle.huffman_decode(got, len(data), 3)
    assert back == bytes(data)


@pytest.mark.parametrize("name,data", list(_symbol_sets()), ids=lambda x: x if isinstance(x, str) else "")
def test_rounds_construction_matches_reference(hh, name, data):
    """The data-parallel 'rounds' construction (what the device runs) gives the reference tree."""
    got, pos, cl = _hh_encode(hh, data, "hh_huffman_encode_rounds")
    ref, rpos, rcl, _ = oracle.huffman_encode(data, bit_pos=3, prefix=b"\x00")
    assert pos == rpos and np.array_equal(cl, rcl) and got == ref


def test_m32_helpers(hh):
    rng = np.random.default_rng(7)
    edge = [0, 1, -1, 126, 127, -126, -127, -128, 254, 255, -254, -255, 16638, 16639, -16638,
            -16639, 2113790, 2113791, 270549246, 270549247, 2 ** 31 - 1, -(2 ** 31) + 1, -(2 ** 31)]
    vals = edge + [int(x) for x in rng.integers(-2 ** 31, 2 ** 31, 3000)]
    vals += [int(x) for x in rng.integers(-70000, 70000, 3000)]
    for v in vals:
        ref = oracle.m32_encode(v)
        u = v & 0xFFFFFFFF
        n = hh.hh_m32_len(u)
        assert n == len(ref), v
        assert bytes(hh.hh_m32_byte(u, n, k) for k in range(n)) == ref, v


@pytest.mark.parametrize("shape", [(2, 2), (2, 3), (3, 2), (5, 7), (7, 5), (1, 6), (10, 10)])
@pytest.mark.parametrize("model", [1, 2, 3])
def test_stream_order(hh, model, shape):
    n_rows, n_cols = shape
    if model == 3 and n_rows < 2:
        pytest.skip("triangle declines")
    # values chosen so that every residual is distinct and identifies its cell: encode a tile
    # that is zero except one cell with a large value, and see which stream slots change.
    n = n_rows * n_cols
    base = np.zeros(n, np.int32)
    m0, _ = oracle.predictor_encode(model, n_rows, n_cols, base)
    assert len(m0) == n - 1
    for s in range(n - 1):
        cell = hh.hh_stream_cell(model, n_rows, n_cols, s)
        assert 1 <= cell < n
        v = base.copy()
        v[cell] = 1
        m, _ = oracle.predictor_encode(model, n_rows, n_cols, v)
        assert len(m) == n - 1          # all residuals stay single-byte
        changed = [i for i in range(n - 1) if m[i] != 0]
        # the earliest slot that changes is the residual of the cell itself
        assert changed and changed[0] == s, (s, cell, changed)
    cells = sorted(hh.hh_stream_cell(model, n_rows, n_cols, s) for s in range(n - 1))
    assert cells == list(range(1, n))
