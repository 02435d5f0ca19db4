"""GPU parity tests of the LSOP12 path (gf_lsop12_* of the C ABI) against the CPU oracle and against the reference
fixture Sample14_LSOP.gvrs: coefficients bit-exact, residual streams, reconstruction, containers."""
import os
import struct

import numpy as np
import pytest

import oracle
from gvrs_walk import tile_packings
from tilegen import make_tile

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    import gridfour_amd
    return gridfour_amd.LsCodecHip()


@pytest.fixture(scope="module")
def sample14(golden_dir):
    (packing,) = tile_packings(os.path.join(golden_dir, "ref_samples", "Sample14_LSOP.gvrs"))[0]
    return packing


def _terrain(nr, nc, seed=0, amp=800):
    rng = np.random.default_rng(seed * 7919 + nr * 131 + nc)
    y, x = np.mgrid[0:nr, 0:nc]
    return (amp * np.sin(x / 9.0) * np.cos(y / 7.0) + 50 * np.sin(x * y / 300.0) + rng.integers(-3, 4, (nr, nc))).astype(np.int32)


def test_sample14_coefficients_and_streams_on_device(codec, sample14):
    # the fixture's tile, its 12 stored float32 coefficients and its two M32 streams
    vals = oracle.lsop12_decode(101, 101, sample14)
    seeds, coefs, res, status = codec.predict(101, 101, vals[None, :])
    assert status[0] == 0 and seeds[0] == 0
    assert coefs[0].tobytes() == sample14[6:54]                       # bit-exact with the reference's encoder
    n_init, n_interior = struct.unpack_from("<ii", sample14, 54)
    init, _ = oracle.huffman_decode(sample14[63:], n_init, 0)
    want_init, _ = oracle.m32_decode_seq(init, 4 * 101 + 2 * 101 - 9)
    assert res[0, :len(want_init)].tolist() == want_init
    o_seed, o_u, o_init, o_inter = oracle.lsop12_residuals(101, 101, vals)
    assert np.array_equal(res[0], np.concatenate([o_init, o_inter]))


def test_sample14_reconstruct_on_device(codec, sample14):
    want = oracle.lsop12_decode(101, 101, sample14)
    seed, u, init, inter = oracle.lsop12_residuals(101, 101, want)
    stored = np.frombuffer(sample14[6:54], "<f4")
    got, status = codec.reconstruct(101, 101, [seed], stored[None, :], np.concatenate([init, inter])[None, :])
    assert status[0] == 0
    assert np.array_equal(got[0], want)


SHAPES = [(6, 6), (7, 9), (16, 16), (33, 65), (120, 150), (200, 200), (9, 300), (130, 7)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d" % s)
def test_predict_parity(codec, shape):
    nr, nc = shape
    tiles = np.stack([_terrain(nr, nc, k).ravel() for k in range(3)] +
                     [make_tile("noise16", nr, nc), make_tile("noise32", nr, nc), make_tile("sparse_big", nr, nc)])
    seeds, coefs, res, status = codec.predict(nr, nc, tiles)
    for t, v in enumerate(tiles):
        ref = oracle.lsop12_residuals(nr, nc, v)
        if ref is None:
            assert status[t] == 1, (t, status[t])
            continue
        o_seed, o_u, o_init, o_inter = ref
        assert status[t] == 0 and seeds[t] == o_seed
        assert coefs[t].tobytes() == o_u.tobytes(), (t, coefs[t], o_u)
        assert np.array_equal(res[t], np.concatenate([o_init, o_inter])), t


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d" % s)
@pytest.mark.parametrize("deflate", [False, True], ids=["canon", "deflate"])
def test_container_parity_and_roundtrip(shape, deflate):
    import gridfour_amd
    codec = gridfour_amd.LsCodecHip(deflate_enabled=deflate)
    nr, nc = shape
    tiles = np.stack([_terrain(nr, nc, k).ravel() for k in range(4)] + [make_tile("noise16", nr, nc),
                                                                        make_tile("uniform", nr, nc)])
    packs, types, status = codec.encode_batch(4, nr, nc, tiles)
    good, good_idx = [], []
    for t, v in enumerate(tiles):
        ref, typ = oracle.lsop12_encode(4, nr, nc, v, deflate)
        if ref is None:
            assert packs[t] is None and status[t] == 1, (t, status[t])
            continue
        assert status[t] == 0 and types[t] == typ, (t, status[t], types[t], typ)
        assert packs[t] == ref, (t, len(packs[t]), len(ref))
        good.append(packs[t])
        good_idx.append(t)
    vals, st = codec.decode_batch(nr, nc, good)
    for k, t in enumerate(good_idx):
        assert st[k] == 0, (t, st[k])
        assert np.array_equal(vals[k], tiles[t]), t


def test_declines_and_errors(codec):
    assert codec.encode(0, 5, 40, np.arange(200, dtype=np.int32)) is None
    assert codec.encode(0, 40, 5, np.arange(200, dtype=np.int32)) is None
    assert codec.encode(0, 20, 20, np.full(400, 3, np.int32)) is None          # singular normal equations
    good = codec.encode(1, 20, 30, _terrain(20, 30))
    assert np.array_equal(codec.decode(20, 30, good), _terrain(20, 30).ravel())
    with pytest.raises(IOError):
        codec.decode(20, 30, good[:len(good) // 2])


def test_legacy_type0_container_is_reported_unsupported(codec, sample14):
    import gridfour_amd
    with pytest.raises(gridfour_amd.GvrsHipError):
        codec.decode(101, 101, sample14)


def test_device_batch_dem_roundtrip():
    import ctypes as C
    import gridfour_amd
    from gridfour_amd import DeviceBuffer, lib
    ctx = gridfour_amd.GvrsHipContext(0)
    nr, nc, nt = 120, 150, 512
    b = gridfour_amd.DeviceTileBatch(ctx, nr, nc, nt)
    b.synth_dem(0x9E3779B97F4A7C15 + 9, 32)
    n = int(lib().gf_lsop12_residual_count(nr, nc))
    stride = (n + 3) // 4 * 4
    dres, dco, dsc = DeviceBuffer(ctx, nt * stride * 4), DeviceBuffer(ctx, nt * 64), DeviceBuffer(ctx, nt * 4)
    gridfour_amd._lib.check(lib().gf_lsop12_encode_batch_i32_dev(ctx.handle, None, 2, nr, nc, nt, b.values.ptr, b.slots.ptr,
                                                                  b.stride, b.lengths.ptr, b.enc_status.ptr, dres.ptr, stride,
                                                                  dco.ptr, dsc.ptr), "enc")
    gridfour_amd._lib.check(lib().gf_lsop12_decode_batch_i32_dev(ctx.handle, None, nr, nc, nt, b.slots.ptr, nt * b.stride, None,
                                                                  b.stride, b.lengths.ptr, b.decoded.ptr, b.dec_status.ptr,
                                                                  dres.ptr, stride, dco.ptr, dsc.ptr), "dec")
    ctx.synchronize()
    assert np.all(b.get_enc_status() == 0) and np.all(b.get_dec_status() == 0)
    vals = b.get_values()
    assert np.array_equal(b.get_decoded(), vals)
    lengths = b.get_lengths()
    for t in range(0, nt, 61):
        ref, typ = oracle.lsop12_encode(2, nr, nc, vals[t], False)
        assert b.get_packing(t, int(lengths[t])) == ref
    for x in (dres, dco, dsc):
        x.free()
    b.free()
