"""Pins the CPU oracle against the reference's own fixtures."""
import math
import os
import struct

import numpy as np
import pytest

import oracle
from gvrs_walk import tile_packings


def _sample(golden_dir, name):
    return os.path.join(golden_dir, "ref_samples", name)


# ---- CodecM32 known answers: CodecM32Test.java:95-112, CodecM32.java:82-89 ----
M32_SIZES = [(0, 1), (126, 1), (127, 2), (-128, 2), (-127, 2), (128, 2), (-129, 2), (254, 2),
             (255, 3), (16638, 3), (16639, 4), (2113790, 4), (2113791, 5), (270549246, 5),
             (270549247, 6), (2 ** 31 - 1, 6), (-(2 ** 31) + 1, 6), (-(2 ** 31), 1)]
M32_BYTES = [(126, "7e"), (127, "7f00"), (128, "7f01"), (255, "7f8000"), (16638, "7fff7f"),
             (16639, "7f808000"), (16640, "7f808001"), (-(2 ** 31), "80"), (-127, "8100"),
             (-126, "82"), (-1, "ff"), (-255, "818000")]


@pytest.mark.parametrize("value,size", M32_SIZES)
def test_m32_sizes(value, size):
    b = oracle.m32_encode(value)
    assert len(b) == size
    vals, used = oracle.m32_decode_seq(b, 1)
    assert vals == [value] and used == size


@pytest.mark.parametrize("value,hexbytes", M32_BYTES)
def test_m32_bytes(value, hexbytes):
    assert oracle.m32_encode(value).hex() == hexbytes


def test_m32_sequence_roundtrip():
    # CodecM32Test.java:118-131: -32780..32780
    vals = list(range(-32780, 32781, 7)) + [2 ** 31 - 1, -(2 ** 31), 0]
    data = oracle.m32_encode_seq(vals)
    out, used = oracle.m32_decode_seq(data, len(vals))
    assert out == vals and used == len(data)


# ---- Sample05 (= 04, 07): CodecDeflate + Linear + M32 ----
@pytest.mark.parametrize("name", ["Sample04_ShortComp.gvrs", "Sample05_IntComp.gvrs",
                                  "Sample07_ICFComp.gvrs"])
def test_sample_intcomp(golden_dir, name):
    tiles = tile_packings(_sample(golden_dir, name))
    assert sorted(tiles) == [0, 1, 2, 3]
    for idx, (packing,) in tiles.items():
        assert len(packing) == 40
        assert packing[0] == 1 and packing[1] == oracle.PM_LINEAR
        assert struct.unpack_from("<i", packing, 6)[0] == 2499
        tr, tc = divmod(idx, 2)
        rows = np.arange(50)[:, None] + tr * 50
        cols = np.arange(50)[None, :] + tc * 50
        expect = (rows * 100 + cols - 1).astype(np.int32)
        got = oracle.codec_deflate_decode(50, 50, packing).reshape(50, 50)
        assert np.array_equal(got, expect)
        # the M32 stream the reference stored == restated Linear predictor + M32
        import zlib
        m32_ref = zlib.decompress(packing[10:])
        m32, seed = oracle.predictor_encode(oracle.PM_LINEAR, 50, 50, expect)
        assert m32 == m32_ref and seed == struct.unpack_from("<i", packing, 2)[0]
        # whole packing byte-for-byte (zlib level 6; the zlib in this image reproduces it)
        re, used = oracle.codec_deflate_encode(1, 50, 50, expect)
        assert used == oracle.PM_LINEAR
        assert re == packing


def test_sample05_tile0_m32_layout(golden_dir):
    # 01, (64,01)x49, 00 x2400
    import zlib
    (packing,) = tile_packings(_sample(golden_dir, "Sample05_IntComp.gvrs"))[0]
    m = zlib.decompress(packing[10:])
    assert m == b"\x01" + b"\x64\x01" * 49 + b"\x00" * 2400


# ---- Sample14: legacy Huffman streams inside the legacy LSOP container ----
def _sample14(golden_dir):
    (packing,) = tile_packings(_sample(golden_dir, "Sample14_LSOP.gvrs"))[0]
    assert len(packing) == 1597
    assert packing[0] == 0 and packing[1] == 12
    n_init, n_interior = struct.unpack_from("<ii", packing, 54)
    assert (n_init, n_interior, packing[62]) == (597, 9603, 0)
    return packing, n_init, n_interior


def test_sample14_huffman_decode_reencode(golden_dir):
    packing, n_init, n_interior = _sample14(golden_dir)
    body = packing[63:]
    init, pos1 = oracle.huffman_decode(body, n_init, 0)
    interior, pos2 = oracle.huffman_decode(body, n_interior, pos1)
    assert pos1 == 1030
    assert (pos2 + 7) // 8 == len(body)
    hist = np.bincount(np.frombuffer(init, np.uint8), minlength=256)
    assert {k: int(v) for k, v in enumerate(hist) if v} == {0: 350, 1: 122, 2: 2, 0xFE: 2, 0xFF: 121}
    # re-encode both segments back to back: must equal the stored bytes exactly
    b1, p1, cl1, tb1 = oracle.huffman_encode(init)
    assert p1 == 1030 and tb1 == 8 + 2 * 5 - 1 + 8 * 5
    assert [int(cl1[s]) for s in (0x01, 0x02, 0xFE, 0xFF, 0x00)] == [2, 4, 4, 3, 1]
    assert b1[:8].hex() == "040ca040ffff01fe"
    b2, p2, _, _ = oracle.huffman_encode(interior, bit_pos=p1, prefix=b1)
    assert p2 == pos2
    assert b2 == body


def test_sample14_m32_values(golden_dir):
    # the initialiser stream starts with row-0 differences of floor(1000*sin(x*pi)*sin(y*pi)+0.5) = 0
    packing, n_init, _ = _sample14(golden_dir)
    init, _ = oracle.huffman_decode(packing[63:], n_init, 0)
    vals, used = oracle.m32_decode_seq(init, 4 * 101 + 2 * 101 - 9)
    assert used == n_init
    assert vals[:100] == [0] * 100            # row 0 is all zeros (sin(0) = 0)


# ---- Sample06: CodecFloat ----
def test_sample06_float(golden_dir):
    import zlib
    tiles = tile_packings(_sample(golden_dir, "Sample06_FltComp.gvrs"))
    for idx, (packing,) in tiles.items():
        assert packing[0] == 2 and packing[1] == 0
        tr, tc = divmod(idx, 2)
        rows = np.arange(50)[:, None] + tr * 50
        cols = np.arange(50)[None, :] + tc * 50
        expect = (rows * 100 + cols - 1).astype(np.float32)
        raw = oracle.codec_float_decode(50, 50, packing)
        assert np.array_equal(raw.view(np.float32).reshape(50, 50), expect)
        # the five stored planes == restated planes
        planes = oracle.float_planes_encode(50, 50, expect.view(np.uint32))
        off, poff = 2, 0
        for p in range(5):
            (zn,) = struct.unpack_from("<i", packing, off)
            pl = (2500 + 7) // 8 if p == 0 else 2500
            stored = zlib.decompress(packing[off + 4:off + 4 + zn])
            assert stored == bytes(planes[poff:poff + pl])
            off += 4 + zn
            poff += pl
        assert off == len(packing)
        # stored zlib streams are level 6 (file predates Deflater(9)); reproduce exactly
        assert oracle.codec_float_encode(2, 50, 50, expect.view(np.uint32), level=6) == packing
        # planes round trip
        assert np.array_equal(oracle.float_planes_decode(50, 50, planes), expect.view(np.uint32).ravel())
