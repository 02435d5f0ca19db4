"""Tile records: RecordManager.writeTile / readTile framing around the GPU codecs, pinned on the
reference's own sample files -- every tile record of Sample05_IntComp (int, Deflate packing), Sample04_ShortComp (short
element), Sample01_IntNoComp and Sample00_ShortNoComp (standard form) is reproduced byte for byte from the cell values,
record header, padding and CRC-32C included."""
import os
import struct

import numpy as np
import pytest

from gvrs_walk import walk_records
from tilegen import make_tile

pytestmark = pytest.mark.gpu
NULL = -2**31


def _file_tile_records(golden_dir, name):
    with open(os.path.join(golden_dir, "ref_samples", name), "rb") as f:
        data = f.read()
    out = {}
    for pos, size, rtype, content in walk_records(data):
        if rtype == 2:
            out[struct.unpack_from("<i", content, 0)[0]] = data[pos:pos + size]
    return out


def _ramp(idx, grid_cols, tile):
    tr, tc = divmod(idx, grid_cols // tile)
    rows = np.arange(tile)[:, None] + tr * tile
    cols = np.arange(tile)[None, :] + tc * tile
    return (rows * grid_cols + cols - 1).ravel()


CASES = [
    # file, element, tile size, grid columns, codec list (None = the standard list)
    ("Sample05_IntComp.gvrs", "int", 50, 100, None),
    ("Sample04_ShortComp.gvrs", "short", 50, 100, None),
    ("Sample01_IntNoComp.gvrs", "int", 5, 10, []),
    ("Sample00_ShortNoComp.gvrs", "short", 5, 10, []),
]


@pytest.mark.parametrize("name,element,tile,grid_cols,codecs", CASES, ids=[c[0][:8] for c in CASES])
def test_reference_sample_records_byte_exact(golden_dir, name, element, tile, grid_cols, codecs):
    import gridfour_amd
    master = gridfour_amd.CodecMasterHip() if codecs is None else gridfour_amd.CodecMasterHip(codec_list=codecs)
    want = _file_tile_records(golden_dir, name)
    assert len(want) == 4
    idx = sorted(want)
    vals = np.stack([_ramp(i, grid_cols, tile) for i in idx]).astype(np.int16 if element == "short" else np.int32)
    recs, used = master.tile_records(tile, tile, idx, vals, element=element, checksums=True)
    for k, i in enumerate(idx):
        assert recs[k] == want[i], (name, i, len(recs[k]), len(want[i]))
    assert (used == (255 if codecs == [] else 1)).all()
    # and back: straight from the file's bytes
    got_idx, got, st = master.tiles_from_records(tile, tile, [want[i] for i in idx], element=element)
    assert (st == 0).all() and list(got_idx) == idx and np.array_equal(got, vals)


def _crc32c(b):
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        tab.append(c)
    c = 0xFFFFFFFF
    for x in b:
        c = tab[(c ^ x) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


@pytest.mark.parametrize("element", ["int", "short"])
@pytest.mark.parametrize("checksums", [True, False])
def test_record_roundtrip_mixed_batch(element, checksums):
    import gridfour_amd
    master = gridfour_amd.CodecMasterHip()
    nr, nc = 40, 60
    kinds = ["smooth", "noise32", "uniform", "ramp", "steps", "smooth"]
    tiles = [make_tile(k, nr, nc, seed=3 + i).copy() for i, k in enumerate(kinds)]
    tiles[4].reshape(nr, nc)[10:20, 5:50] = NULL               # a block of nulls
    if element == "short":
        tiles = [np.where(t == NULL, -32768, np.clip(t, -32767, 32767)).astype(np.int16) for t in tiles]
        tiles[1] = np.random.default_rng(1).integers(-32767, 32768, nr * nc).astype(np.int16)   # incompressible
    vals = np.stack(tiles)
    idx = [7, 0, 123456, 3, 2**31 - 1, 5]
    recs, used = master.tile_records(nr, nc, idx, vals, element=element, fill_value=-32768, checksums=checksums)
    std = (nr * nc * (2 if element == "short" else 4) + 3) // 4 * 4
    assert used[1] == 255
    for k, r in enumerate(recs):
        size, rtype, pad = struct.unpack_from("<iB3s", r, 0)
        assert size == len(r) and size % 8 == 0 and rtype == 2 and pad == b"\0\0\0"
        tile_index, n = struct.unpack_from("<ii", r, 8)
        assert tile_index == idx[k]
        assert size == (8 + n + 12 + 7) // 8 * 8 and (n == std) == (used[k] == 255)
        assert r[16 + n:size - 4] == b"\0" * (size - 4 - 16 - n)
        assert struct.unpack_from("<I", r, size - 4)[0] == (_crc32c(r[:size - 4]) if checksums else 0)
    got_idx, got, st = master.tiles_from_records(nr, nc, recs, element=element, verify_checksums=checksums)
    assert (st == 0).all() and list(got_idx) == idx
    assert np.array_equal(got, vals)
    if checksums:                                             # a flipped bit anywhere is caught by the checksum
        bad = bytearray(recs[0])
        bad[len(bad) // 2] ^= 0x10
        _, _, st = master.tiles_from_records(nr, nc, [bytes(bad), recs[1]], element=element)
        assert st[0] == -1 and st[1] == 0
    # not a tile record / truncated
    other = bytearray(recs[2])
    other[4] = 3
    _, _, st = master.tiles_from_records(nr, nc, [bytes(other), recs[2][:12]], element=element, verify_checksums=False)
    assert st[0] == -1 and st[1] == -2


def test_short_fill_value_maps_to_null_code():
    """TileElementShort.encode: cells equal to the element's fill value reach the codecs as INT4_NULL_CODE; on the way back
    INT4_NULL_CODE becomes Short.MIN_VALUE whatever the fill value is (TileElementShort.java:216-219, 241-243)."""
    import gridfour_amd
    import oracle
    master = gridfour_amd.CodecMasterHip()
    nr, nc = 30, 30
    v = make_tile("smooth", nr, nc, seed=4).reshape(nr, nc).astype(np.int16)
    v[5:9, 3:20] = -9999
    recs, used = master.tile_records(nr, nc, [1], v[None], element="short", fill_value=-9999)
    n = struct.unpack_from("<i", recs[0], 12)[0]
    packing = recs[0][16:16 + n]
    as_int = np.where(v == -9999, NULL, v.astype(np.int32))
    want, want_used = None, None
    for k, enc in ((0, oracle.codec_huffman_encode), (1, oracle.codec_deflate_encode), (3, oracle.codec_canon_encode)):
        pk = enc(k, nr, nc, as_int)
        pk = pk[0] if isinstance(pk, tuple) else pk
        if pk is not None and (want is None or len(pk) < len(want)):
            want, want_used = pk, k
    assert packing == want and used[0] == want_used
    _, got, st = master.tiles_from_records(nr, nc, recs, element="short")
    assert st[0] == 0 and np.array_equal(got[0], np.where(v == -9999, -32768, v).ravel())


@pytest.mark.parametrize("verify", [True, False], ids=["crc", "nocrc"])
def test_every_single_bit_flip_of_a_record(verify):
    """Every one-bit damage of one tile record among three, all variants in ONE batch call.  With the CRC-32C verified every
    flip is caught (a CRC detects all single-bit errors); without it the framing is on its own -- record size, type,
    element length and packing bytes all get damaged: nothing may crash, the neighbours decode untouched, a record the
    framing still accepts keeps its tile index unless the flip was in the index itself."""
    import gridfour_amd
    master = gridfour_amd.CodecMasterHip()
    nr, nc = 10, 14
    tiles = np.stack([make_tile("smooth", nr, nc, seed=21), make_tile("steps", nr, nc, seed=22), make_tile("ramp", nr, nc, seed=23)])
    idx = [11, 12, 13]
    recs, used = master.tile_records(nr, nc, idx, tiles, element="int", checksums=True)
    victim = recs[1]
    batch, where = [], []
    for i in range(len(victim)):
        for b in range(8):
            x = bytearray(victim)
            x[i] ^= 1 << b
            batch += [recs[0], bytes(x), recs[2]]
            where.append((i, b))
    got_idx, got, st = master.tiles_from_records(nr, nc, batch, element="int", verify_checksums=verify)
    st = st.reshape(-1, 3)
    got = got.reshape(-1, 3, nr * nc)
    got_idx = got_idx.reshape(-1, 3)
    assert (st[:, 0] == 0).all() and (st[:, 2] == 0).all()
    assert (got[:, 0] == tiles[0]).all() and (got[:, 2] == tiles[2]).all()
    assert (got_idx[:, 0] == 11).all() and (got_idx[:, 2] == 13).all()
    if verify:
        assert (st[:, 1] != 0).all()
    else:
        accepted = st[:, 1] == 0
        assert accepted.any() and (~accepted).any()
        for k in np.nonzero(accepted)[0]:
            i, b = where[k]
            if not 8 <= i < 12:                        # (bytes 8..11 hold the tile index)
                assert got_idx[k, 1] == 12, (i, b)
