"""Pins the LSOP12 restatement (oracle/gvrs_oracle_lsop.c) against the reference fixture
Sample14_LSOP.gvrs (core/src/test/resources/org/gridfour/gvrs/SampleFiles): one 101x101 tile written
from floor(1000*sin(x*pi)*sin(y*pi)+0.5) in the LEGACY container (header without the revision flag,
legacy Huffman of the two M32 streams).  Not pinned by any fixture: the current canonical-Huffman
container and the Deflate container -- those are round-trip tested only."""
import math
import os
import struct

import numpy as np
import pytest

import oracle
from gvrs_walk import tile_packings


@pytest.fixture(scope="module")
def sample14(golden_dir):
    (packing,) = tile_packings(os.path.join(golden_dir, "ref_samples", "Sample14_LSOP.gvrs"))[0]
    assert len(packing) == 1597
    return packing


def _surface(n=101):
    # the sample's generator (reference: demo/test writer of Sample14): z = floor(1000 sin(x pi) sin(y pi) + 0.5)
    v = np.zeros((n, n), np.int32)
    for r in range(n):
        for c in range(n):
            x, y = c / (n - 1.0), r / (n - 1.0)
            v[r, c] = int(math.floor(1000.0 * math.sin(x * math.pi) * math.sin(y * math.pi) + 0.5))
    return v


def test_sample14_decode_is_the_analytic_surface(sample14):
    got = oracle.lsop12_decode(101, 101, sample14).reshape(101, 101)
    want = _surface()
    assert np.array_equal(got, want)


def test_sample14_coefficients_bit_exact(sample14):
    vals = oracle.lsop12_decode(101, 101, sample14)
    stored = np.frombuffer(sample14[6:54], "<f4")
    u = oracle.lsop12_coefficients(101, 101, vals)
    assert u is not None
    assert u.tobytes() == stored.tobytes()


def test_sample14_reencode_legacy_container_byte_exact(sample14):
    vals = oracle.lsop12_decode(101, 101, sample14)
    again = oracle.lsop12_encode_legacy_huffman(0, 101, 101, vals)
    assert again == sample14


def test_sample14_residual_streams(sample14):
    vals = oracle.lsop12_decode(101, 101, sample14)
    seed, u, init, inter = oracle.lsop12_residuals(101, 101, vals)
    assert seed == 0 and init.size == 4 * 101 + 2 * 101 - 9 and inter.size == 99 * 97
    n_init, n_interior = struct.unpack_from("<ii", sample14, 54)
    assert len(oracle.m32_encode_seq(init)) == n_init
    assert len(oracle.m32_encode_seq(inter)) == n_interior


@pytest.mark.parametrize("shape", [(6, 6), (7, 9), (32, 32), (120, 150), (9, 200)])
@pytest.mark.parametrize("deflate", [False, True])
def test_lsop_roundtrip_current_containers(shape, deflate):
    nr, nc = shape
    rng = np.random.default_rng(nr * 1000 + nc)
    y, x = np.mgrid[0:nr, 0:nc]
    v = (800 * np.sin(x / 9.0) * np.cos(y / 7.0) + rng.integers(-3, 4, (nr, nc))).astype(np.int32)
    packing, typ = oracle.lsop12_encode(5, nr, nc, v, deflate)
    assert packing is not None and typ in ((1, 2) if deflate else (2,))
    assert packing[0] == 5 and packing[1] == (0x40 | typ) and packing[2] == 12
    assert np.array_equal(oracle.lsop12_decode(nr, nc, packing), v.ravel())


def test_lsop_declines():
    assert oracle.lsop12_encode(0, 5, 50, np.zeros(250, np.int32))[0] is None       # < 6 rows
    assert oracle.lsop12_encode(0, 50, 5, np.zeros(250, np.int32))[0] is None       # < 6 columns
    assert oracle.lsop12_encode(0, 20, 20, np.full(400, 7, np.int32))[0] is None    # singular normal equations


def test_lsop_large_values_roundtrip():
    rng = np.random.default_rng(7)
    v = rng.integers(-2**31, 2**31 - 1, (12, 13), dtype=np.int64).astype(np.int32)
    packing, typ = oracle.lsop12_encode(1, 12, 13, v, True)
    if packing is not None:
        assert np.array_equal(oracle.lsop12_decode(12, 13, packing), v.ravel())
