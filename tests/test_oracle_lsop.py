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


# ---- value checksum (LsEncoder12.setValueChecksumEnabled :117-119, LsHeader.computeChecksum :391-406) ----
def test_crc32c_pinned_by_the_reference_sample_records(golden_dir):
    """util/GridfourCRC32C as the oracle restates it, against checksums the REFERENCE wrote: the last four bytes of every tile
    record of the sample files are the CRC-32C of the record before them (RecordManager.writeTile)."""
    import os
    import struct
    from gvrs_walk import walk_records
    checked = 0
    for name in ("Sample05_IntComp.gvrs", "Sample04_ShortComp.gvrs", "Sample01_IntNoComp.gvrs", "Sample00_ShortNoComp.gvrs"):
        with open(os.path.join(golden_dir, "ref_samples", name), "rb") as f:
            data = f.read()
        for pos, size, rtype, _ in walk_records(data):
            if rtype != 2:
                continue
            rec = data[pos:pos + size]
            assert struct.unpack_from("<I", rec, size - 4)[0] == oracle.crc32c(rec[:size - 4]), (name, pos)
            checked += 1
    assert checked == 16
    assert oracle.crc32c(b"123456789") == 0xE3069283          # the polynomial's published check value


@pytest.mark.parametrize("deflate", [False, True])
def test_value_checksum_header(deflate):
    """With the checksum enabled the packing is the one without it plus four bytes behind the header's last field, bit 7 of
    byte 1 set (LsHeader.packHeader :245-262); the decoder skips them (LsDecoder12 prints a mismatch, nothing else)."""
    import struct
    nr, nc = 24, 40
    v = oracle.dem_tiles(oracle.DEM_SEED + 5, nr, nc, 16, 0, 1)[0]
    plain, typ = oracle.lsop12_encode(3, nr, nc, v, deflate)
    summed, typ2 = oracle.lsop12_encode(3, nr, nc, v, deflate, value_checksum=True)
    assert typ == typ2
    hdr = 55 if typ == 2 else 63
    assert len(summed) == len(plain) + 4
    assert summed[1] == plain[1] | 0x80 and summed[0] == plain[0] and summed[2:hdr] == plain[2:hdr]
    assert struct.unpack_from("<I", summed, hdr)[0] == oracle.lsop_value_checksum(nr, nc, v)
    assert oracle.lsop_value_checksum(nr, nc, v) == oracle.crc32c(v.astype("<i4").tobytes())
    assert summed[hdr + 4:] == plain[hdr:]
    assert np.array_equal(oracle.lsop12_decode(nr, nc, summed), v)
