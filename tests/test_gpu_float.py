"""GPU parity of the CodecFloat path: byte planes on the GPU + host zlib."""
import os
import struct

import numpy as np
import pytest

import oracle
from gvrs_walk import tile_packings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fcodec():
    import gridfour_amd
    return gridfour_amd.CodecFloatHip(level=6)


def _float_tiles(rng, n_rows, n_cols):
    n = n_rows * n_cols
    r = np.arange(n_rows)[:, None]
    c = np.arange(n_cols)[None, :]
    smooth = (np.sin(r / 9.0) * np.cos(c / 7.0) * 1000.0).astype(np.float32).ravel()
    ramp = (np.arange(n, dtype=np.float32) - 1.0)
    noise = rng.standard_normal(n).astype(np.float32) * 1e4
    special = smooth.copy()
    special[::7] = np.nan
    special[3::11] = -0.0
    special[5::13] = np.inf
    special[1] = np.float32(-np.inf)
    tiny = (rng.standard_normal(n) * 1e-40).astype(np.float32)          # denormals
    return [smooth, ramp, noise, special, tiny]


@pytest.mark.parametrize("shape", [(50, 50), (7, 9), (1, 5), (5, 1), (64, 65), (256, 256), (3, 1100), (5, 12), (7, 20), (1, 4), (120, 152), (9, 260)], ids=lambda s: "%dx%d" % s)
def test_planes_and_packings_match_oracle(fcodec, shape):
    import gridfour_amd
    from gridfour_amd import DeviceBuffer, lib
    n_rows, n_cols = shape
    rng = np.random.default_rng(n_rows * 1000 + n_cols)
    tiles = np.stack(_float_tiles(rng, n_rows, n_cols))
    nt, n = tiles.shape
    # planes: bit-exact with the restated plane split / delta rule
    L = lib()
    stride = (int(L.gf_float_planes_bytes(n_rows, n_cols)) + 15) // 16 * 16
    d_vals = DeviceBuffer(fcodec.ctx, tiles.nbytes).upload(tiles)
    d_planes = DeviceBuffer(fcodec.ctx, nt * stride).fill(0xAB)
    gridfour_amd.codec.check(L.gf_float_planes_encode_dev(fcodec.ctx.handle, None, n_rows, n_cols, nt, d_vals.ptr, d_planes.ptr, stride))
    fcodec.ctx.synchronize()
    planes = d_planes.download(np.uint8, nt * stride).reshape(nt, stride)
    for t in range(nt):
        ref = oracle.float_planes_encode(n_rows, n_cols, tiles[t].view(np.uint32))
        assert np.array_equal(planes[t, :ref.size], ref), (shape, t)
    # planes -> raw bits on the device
    d_back = DeviceBuffer(fcodec.ctx, tiles.nbytes).fill(0)
    gridfour_amd.codec.check(L.gf_float_planes_decode_dev(fcodec.ctx.handle, None, n_rows, n_cols, nt, d_planes.ptr, stride, d_back.ptr))
    fcodec.ctx.synchronize()
    back = d_back.download(np.uint32, nt * n).reshape(nt, n)
    assert np.array_equal(back, tiles.view(np.uint32))
    # whole packings (zlib level 6 on both sides) and decode
    packs = fcodec.encode_floats_batch(2, n_rows, n_cols, tiles)
    for t in range(nt):
        assert packs[t] == oracle.codec_float_encode(2, n_rows, n_cols, tiles[t].view(np.uint32), level=6), (shape, t)
    vals, st = fcodec.decode_floats_batch(n_rows, n_cols, packs)
    assert (st == 0).all() and np.array_equal(vals.view(np.uint32), tiles.view(np.uint32))
    for b in (d_vals, d_planes, d_back):
        b.free()


def test_reference_sample06(fcodec, golden_dir):
    """The reference's own float fixture: stored packings decode to row*100+col-1 and are reproduced exactly."""
    tiles = tile_packings(os.path.join(golden_dir, "ref_samples", "Sample06_FltComp.gvrs"))
    for idx, (packing,) in tiles.items():
        tr, tc = divmod(idx, 2)
        rows = np.arange(50)[:, None] + tr * 50
        cols = np.arange(50)[None, :] + tc * 50
        expect = (rows * 100 + cols - 1).astype(np.float32).ravel()
        got = fcodec.decodeFloats(50, 50, packing)
        assert np.array_equal(got.view(np.uint32), expect.view(np.uint32))
        assert fcodec.encodeFloats(2, 50, 50, expect) == packing
    assert fcodec.implementsFloatingPointEncoding() and not fcodec.implementsIntegerEncoding()
    assert fcodec.encode(0, 2, 2, np.zeros(4, np.int32)) is None
    with pytest.raises(IOError):
        fcodec.decodeFloats(50, 50, packing[:40])


def test_every_single_bit_flip_matches_the_oracle(fcodec):
    """Every one-bit damage of a CodecFloat packing (two header bytes, five length-prefixed zlib streams) in one batch: the
    device accepts exactly what the oracle (host zlib driven as java.util.zip.Inflater) accepts, with the same floats."""
    nr, nc = 9, 14
    rng = np.random.default_rng(77)
    f = (np.sin(np.arange(nr * nc) / 5.0) * 300.0 + rng.standard_normal(nr * nc)).astype(np.float32)
    good = fcodec.encodeFloats(3, nr, nc, f)
    packs = []
    for i in range(2, len(good)):                      # (bytes 0 and 1 are not looked at by decodeFloats)
        for b in range(8):
            x = bytearray(good)
            x[i] ^= 1 << b
            packs.append(bytes(x))
    vals, st = fcodec.decode_floats_batch(nr, nc, packs)
    n_ok = n_err = 0
    for k, pk in enumerate(packs):
        where = (k // 8 + 2, k % 8)
        try:
            want = oracle.codec_float_decode(nr, nc, pk)
        except IOError:
            want = None
        if want is None:
            assert st[k] != 0, where
            n_err += 1
        else:
            assert st[k] == 0 and np.array_equal(vals[k].view(np.uint32), want), (where, int(st[k]))
            n_ok += 1
    assert n_ok > 0 and n_err > 0


def test_short_plane_keeps_what_the_plane_before_left(fcodec):
    """CodecFloat.decodeFloats inflates all five planes into ONE scratch array and decodes the mantissa deltas in place
    (CodecFloat.java:397-446): a plane whose stream gives fewer bytes than the tile has cells -- no exception in the
    reference -- continues with what the previous plane left behind.  Packings rebuilt with deliberately short (but valid)
    zlib streams for one plane each; device = oracle (which restates the method statement by statement)."""
    import zlib
    nr, nc = 10, 16
    n = nr * nc
    rng = np.random.default_rng(9)
    f = (np.cos(np.arange(n) / 7.0) * 1000.0 + rng.standard_normal(n) * 3).astype(np.float32)
    good = fcodec.encodeFloats(3, nr, nc, f)
    # split into its five streams
    streams, off = [], 2
    for _ in range(5):
        zn = struct.unpack_from("<I", good, off)[0]
        streams.append(good[off + 4:off + 4 + zn])
        off += 4 + zn
    planes = [zlib.decompress(s) for s in streams]
    packs = []
    for p in range(5):
        for keep in (0, 1, len(planes[p]) // 3, len(planes[p]) - 1):
            ss = list(streams)
            ss[p] = zlib.compress(planes[p][:keep], 6)
            packs.append(good[:2] + b"".join(struct.pack("<I", len(s)) + s for s in ss))
    # two planes short at once
    ss = list(streams)
    ss[2] = zlib.compress(planes[2][:40], 6)
    ss[4] = zlib.compress(planes[4][:7], 6)
    packs.append(good[:2] + b"".join(struct.pack("<I", len(s)) + s for s in ss))
    vals, st = fcodec.decode_floats_batch(nr, nc, packs)
    differs = 0
    for k, pk in enumerate(packs):
        want = oracle.codec_float_decode(nr, nc, pk)
        assert st[k] == 0 and np.array_equal(vals[k].view(np.uint32), want), k
        differs += int(not np.array_equal(want, f.view(np.uint32)))
    assert differs >= len(packs) - 6
