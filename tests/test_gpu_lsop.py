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


def _surface(n=101):
    # the generator behind Sample14: z = floor(1000 sin(x pi) sin(y pi) + 0.5)
    import math
    v = np.zeros((n, n), np.int32)
    for r in range(n):
        for c in range(n):
            x, y = c / (n - 1.0), r / (n - 1.0)
            v[r, c] = int(math.floor(1000.0 * math.sin(x * math.pi) * math.sin(y * math.pi) + 0.5))
    return v


def test_sample14_reference_packing_decodes_on_device(codec, sample14):
    """The one LSOP12 packing the reference's own resources hold (legacy header, two legacy-Huffman segments of M32
    bytes in one bit store) goes through the C ABI and comes back as the analytic surface it was written from."""
    got = codec.decode(101, 101, sample14)
    assert np.array_equal(got.reshape(101, 101), _surface())
    vals, st = codec.decode_batch(101, 101, [sample14] * 5)
    assert np.all(st == 0) and all(np.array_equal(v, got) for v in vals)


LEGACY_SHAPES = [(6, 6), (7, 9), (20, 30), (64, 64), (120, 150), (9, 200), (200, 200)]


@pytest.mark.parametrize("shape", LEGACY_SHAPES, ids=lambda s: "%dx%d" % s)
def test_legacy_huffman_container_decode(codec, shape):
    """Type-0 containers as old Gridfour versions wrote them (made by the oracle's restatement, which reproduces
    Sample14 byte for byte): smooth terrain, large residuals (multi-byte M32 values, long codes), and mixed batches
    with the current container types."""
    nr, nc = shape
    tiles = [_terrain(nr, nc, k).ravel() for k in range(3)]
    tiles.append(make_tile("noise16", nr, nc))
    tiles.append((_terrain(nr, nc, 9, amp=200000).ravel() * 37).astype(np.int32))
    packs, want = [], []
    for k, v in enumerate(tiles):
        try:
            pk = oracle.lsop12_encode_legacy_huffman(k, nr, nc, v)
        except ValueError:
            continue                                      # singular system: the encoder declines
        assert np.array_equal(oracle.lsop12_decode(nr, nc, pk), v)
        packs.append(pk)
        want.append(v)
        cur, _ = oracle.lsop12_encode(k, nr, nc, v, k % 2 == 0)           # canonical / Deflate neighbours
        if cur is not None:
            packs.append(cur)
            want.append(v)
    assert packs
    vals, st = codec.decode_batch(nr, nc, packs)
    for k, v in enumerate(want):
        assert st[k] == 0, (k, st[k])
        assert np.array_equal(vals[k], v), k


def _handmade_type0(nr, nc, seed, coefs, init, interior, revised=False, checksum=False):
    """A type-0 container around chosen residual streams (LsHeader.java:139-185 both revisions, LsDecoder12.java:116-124)."""
    m_init, m_int = oracle.m32_encode_seq(init), oracle.m32_encode_seq(interior)
    co = np.asarray(coefs, "<f4").tobytes()
    if revised:
        head = bytes([3, 0x40 | (0x80 if checksum else 0), 12]) + struct.pack("<i", seed) + co + struct.pack("<ii", len(m_init), len(m_int))
    else:
        head = bytes([3, 12]) + struct.pack("<i", seed) + co + struct.pack("<ii", len(m_init), len(m_int)) + bytes([0x80 if checksum else 0])
    if checksum:
        head += b"\x12\x34\x56\x78"
    buf, pos, _, _ = oracle.huffman_encode(np.frombuffer(m_init, np.uint8), len(head) * 8, head)
    buf, pos, _, _ = oracle.huffman_encode(np.frombuffer(m_int, np.uint8), pos, buf)
    return buf


@pytest.mark.parametrize("revised,checksum", [(False, False), (True, False), (True, True), (False, True)])
def test_legacy_container_handmade_streams(codec, revised, checksum):
    """Segments the encoder seldom produces: single-symbol trees (HuffmanEncoder.java:147-157, 17 bits and no text) in
    either position, segments that start at every bit phase, values of every M32 length, long codes."""
    nr, nc = 24, 40
    n_init, n_int = 4 * nr + 2 * nc - 9, (nr - 2) * (nc - 4)
    rng = np.random.default_rng(2024)
    coefs = (rng.standard_normal(12) * 0.2).astype(np.float32)
    wide = rng.integers(-2**31, 2**31, n_int, dtype=np.int64).astype(np.int32)
    wide[rng.random(n_int) < 0.7] = 0
    skew = np.where(rng.random(n_int) < 0.97, 0, rng.integers(-120, 120, n_int)).astype(np.int32)   # long codes
    streams = [
        (rng.integers(-5, 6, n_init), np.zeros(n_int, np.int64)),                    # interior: one symbol
        (np.full(n_init, 7), rng.integers(-300, 300, n_int)),                        # initialisers: one symbol
        (np.zeros(n_init, np.int64), np.zeros(n_int, np.int64)),                     # both
        (rng.integers(-70000, 70000, n_init), wide),
        (rng.integers(-2, 3, n_init), skew),
    ]
    for k in range(8):                                                               # bit phases of the second segment
        streams.append((rng.integers(-3, 4, n_init)[: n_init], rng.integers(-(k + 2), k + 3, n_int)))
    packs = [_handmade_type0(nr, nc, 1000 + i, coefs, a, b, revised, checksum) for i, (a, b) in enumerate(streams)]
    want = [oracle.lsop12_decode(nr, nc, pk) for pk in packs]
    vals, st = codec.decode_batch(nr, nc, packs)
    for k in range(len(packs)):
        assert st[k] == 0, (k, st[k])
        assert np.array_equal(vals[k], want[k]), k


def test_legacy_container_damage_is_reported(codec, sample14):
    rng = np.random.default_rng(11)
    packs = [sample14[:n] for n in (2, 40, 62, 63, 64, 200, 800, 1596)]
    for k in range(40):
        b = bytearray(sample14)
        for _ in range(1 + k % 4):
            b[int(rng.integers(1, len(b)))] ^= 1 << int(rng.integers(0, 8))
        packs.append(bytes(b))
    vals, st = codec.decode_batch(101, 101, packs)
    assert set(int(x) for x in st) <= {0, -1, -2, -7}
    for k, pk in enumerate(packs):
        try:
            ref = oracle.lsop12_decode(101, 101, pk)
        except Exception:
            ref = None
        if ref is None:
            continue
        if st[k] == 0:
            assert np.array_equal(vals[k], ref), k


def test_device_resident_legacy_containers(sample14):
    """gf_lsop12_decode_batch_i32_dev on packings already in HBM: every container type (legacy Huffman, canonical, Deflate)
    is decoded entirely on the device."""
    import gridfour_amd
    from gridfour_amd import DeviceBuffer, lib
    ctx = gridfour_amd.GvrsHipContext(0)
    nr = nc = 101
    v = oracle.lsop12_decode(nr, nc, sample14)
    canon, typ = oracle.lsop12_encode(0, nr, nc, v, False)
    defl, typ1 = oracle.lsop12_encode(0, nr, nc, v, True)
    assert typ == 2
    packs = [sample14, canon, defl, sample14]
    stride = 4096 * 4
    nt = len(packs)
    blob = np.zeros(nt * stride, np.uint8)
    for k, pk in enumerate(packs):
        blob[k * stride:k * stride + len(pk)] = np.frombuffer(pk, np.uint8)
    lengths = np.array([len(pk) for pk in packs], np.uint32)
    n = int(lib().gf_lsop12_residual_count(nr, nc))
    rs = (n + 3) // 4 * 4
    dblob, dlen, dval, dst = (DeviceBuffer(ctx, blob.nbytes), DeviceBuffer(ctx, nt * 4), DeviceBuffer(ctx, nt * nr * nc * 4),
                              DeviceBuffer(ctx, nt * 4))
    dres, dco, dsc = DeviceBuffer(ctx, nt * rs * 4), DeviceBuffer(ctx, nt * 64), DeviceBuffer(ctx, nt * 4)
    dblob.upload(blob)
    dlen.upload(lengths)
    gridfour_amd._lib.check(lib().gf_lsop12_decode_batch_i32_dev(ctx.handle, None, nr, nc, nt, dblob.ptr, blob.nbytes, None, stride,
                                                                  dlen.ptr, dval.ptr, dst.ptr, dres.ptr, rs, dco.ptr, dsc.ptr), "dec")
    ctx.synchronize()
    st = dst.download(np.int32, nt)
    got = dval.download(np.int32, nt * nr * nc).reshape(nt, -1)
    for k in (0, 1, 3):
        assert st[k] == 0 and np.array_equal(got[k], v), (k, st[k])
    assert st[2] == 0 and np.array_equal(got[2], v)          # Deflate containers are inflated on the device too


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


def test_config5ii_int_coded_float_256x256():
    """BASELINE config 5(ii): 256x256 float elevation tiles stored as int-coded floats (scale 10,
    GvrsElementIntCodedFloat.java:205: (int) Math.floor((f - offset) * scale + 0.5)) through LSOP12, device-resident:
    every packing equals the oracle's on a sample, every tile survives the round trip."""
    import gridfour_amd
    ctx = gridfour_amd.GvrsHipContext(0)
    nr, nc, nt = 256, 256, 96
    b = gridfour_amd.DeviceTileBatch(ctx, nr, nc, nt, codec="lsop")
    b.synth_dem(0x9E3779B97F4A7C15 + 5, 8)
    ctx.synchronize()
    f = b.get_values().astype(np.float32) * np.float32(0.1)
    coded = np.floor((f * np.float32(10.0)).astype(np.float64) + 0.5).astype(np.int32)
    # a block of larger jitter in some tiles so that the residual streams are not all one-byte codes
    rng = np.random.default_rng(5)
    coded[::7, 1000:30000] += rng.integers(-400, 400, 29000).astype(np.int32)
    b.values.upload(coded)
    b.encode(codec_index=3)
    b.decode()
    ctx.synchronize()
    assert np.all(b.get_enc_status() == 0) and np.all(b.get_dec_status() == 0)
    assert np.array_equal(b.get_decoded(), coded)
    lengths = b.get_lengths()
    for t in range(0, nt, 5):
        ref, typ = oracle.lsop12_encode(3, nr, nc, coded[t], False)
        assert typ == 2 and b.get_packing(t, int(lengths[t])) == ref, t
    b.free()


def test_deflate_container_damage_matches_the_oracle(codec):
    """Deflate (type 1) containers are inflated on the device, the second zlib stream behind what the first one consumed.
    Truncated and bit-flipped containers: whenever the oracle (host zlib, as java.util.zip.Inflater) decodes one, the device
    decodes it to the same cells; whenever the oracle rejects one, the device does not report success with other cells."""
    nr, nc = 64, 80
    y, x = np.mgrid[0:nr, 0:nc]
    v = (3 * x + 5 * y + 40 * ((x // 16 + y // 16) % 2)).astype(np.int32).ravel()      # a plane with steps: long runs of equal residuals
    defl, typ = oracle.lsop12_encode(1, nr, nc, v, True)
    assert typ == 1, "the Deflate container should win on this tile"
    rng = np.random.default_rng(23)
    packs = [defl] + [defl[:n] for n in (60, 71, 72, 80, len(defl) // 2, len(defl) - 5, len(defl) - 1)]
    for k in range(60):
        b = bytearray(defl)
        for _ in range(1 + k % 3):
            b[int(rng.integers(63, len(b)))] ^= 1 << int(rng.integers(0, 8))
        packs.append(bytes(b))
    vals, st = codec.decode_batch(nr, nc, packs)
    assert st[0] == 0 and np.array_equal(vals[0], v)
    n_ok = 0
    for k, pk in enumerate(packs):
        try:
            ref = oracle.lsop12_decode(nr, nc, pk)
        except Exception:
            ref = None
        if ref is not None:
            assert st[k] == 0 and np.array_equal(vals[k], ref), (k, st[k])
            n_ok += 1
        else:
            assert st[k] != 0, k
    assert n_ok >= 1


def test_every_single_bit_flip_of_a_deflate_container_matches_the_oracle(codec):
    """Every one-bit damage of an LSOP12 Deflate container (header, two chained zlib streams), in one batch, against the
    oracle's verdict and cells."""
    nr, nc = 20, 24
    y, x = np.mgrid[0:nr, 0:nc]
    v = (3 * x + 5 * y + 40 * ((x // 8 + y // 8) % 2)).astype(np.int32).ravel()
    defl, typ = oracle.lsop12_encode(1, nr, nc, v, True)
    assert typ == 1
    packs = []
    for i in range(1, len(defl)):
        for b in range(8):
            z = bytearray(defl)
            z[i] ^= 1 << b
            packs.append(bytes(z))
    vals, st = codec.decode_batch(nr, nc, packs)
    n_ok = n_err = 0
    for k, pk in enumerate(packs):
        where = (k // 8 + 1, k % 8)
        try:
            want = oracle.lsop12_decode(nr, nc, pk)
        except Exception:
            want = None
        if want is None:
            assert st[k] != 0, where
            n_err += 1
        else:
            assert st[k] == 0 and np.array_equal(vals[k], want), (where, int(st[k]))
            n_ok += 1
    assert n_ok > 0 and n_err > 0


# ---- LsEncoder12.setValueChecksumEnabled (lsop/LsEncoder12.java:117-119, LsHeader.computeChecksum :391-406) ----
@pytest.mark.parametrize("shape", [(6, 6), (24, 40), (101, 101), (120, 150), (37, 203)], ids=lambda s: "%dx%d" % s)
@pytest.mark.parametrize("deflate", [False, True], ids=["canon", "deflate"])
def test_value_checksum_containers(shape, deflate):
    """Both containers with the value checksum: byte-equal to the oracle (whose CRC-32C is pinned by the reference's own tile
    records, tests/test_oracle_lsop.py), decodable by the library and by the oracle."""
    import gridfour_amd
    codec = gridfour_amd.LsCodecHip(deflate_enabled=deflate)
    codec.setValueChecksumEnabled(True)
    nr, nc = shape
    rng = np.random.default_rng(nr * 1000 + nc)
    tiles = np.stack([_terrain(nr, nc, k).ravel() for k in range(3)] +
                     [rng.integers(-2**31, 2**31 - 1, nr * nc, dtype=np.int64).astype(np.int32),      # every byte value in the CRC
                      make_tile("noise16", nr, nc)])
    packs, types, status = codec.encode_batch(1, nr, nc, tiles)
    good, idx = [], []
    for t, v in enumerate(tiles):
        ref, typ = oracle.lsop12_encode(1, nr, nc, v, deflate, value_checksum=True)
        if ref is None:
            assert packs[t] is None and status[t] == 1
            continue
        assert status[t] == 0 and types[t] == typ, (t, status[t], types[t], typ)
        assert packs[t][1] & 0x80
        hdr = 55 if typ == 2 else 63
        assert struct.unpack_from("<I", packs[t], hdr)[0] == oracle.lsop_value_checksum(nr, nc, v), t
        assert packs[t] == ref, (t, len(packs[t]), len(ref))
        good.append(packs[t])
        idx.append(t)
    vals, st = codec.decode_batch(nr, nc, good)
    assert (st == 0).all() and np.array_equal(vals, tiles[idx])
    # the same tiles without the switch: the four bytes and the flag are the whole difference
    codec.setValueChecksumEnabled(False)
    plain, types2, _ = codec.encode_batch(1, nr, nc, tiles[idx])
    for k, t in enumerate(idx):
        hdr = 55 if types[t] == 2 else 63
        assert types2[k] == types[t] and len(good[k]) == len(plain[k]) + 4
        assert good[k][hdr + 4:] == plain[k][hdr:] and good[k][2:hdr] == plain[k][2:hdr]


def test_value_checksum_device_resident_batch():
    import gridfour_amd
    ctx = gridfour_amd.GvrsHipContext(0)
    nr, nc, nt = 120, 150, 300
    b = gridfour_amd.DeviceTileBatch(ctx, nr, nc, nt, codec="lsop")
    b.synth_dem(0x9E3779B97F4A7C15 + 19, 32)
    b.encode(codec_index=2, lsop_flags=gridfour_amd._lib.LSOP_VALUE_CHECKSUM)
    b.decode()
    ctx.synchronize()
    assert np.all(b.get_enc_status() == 0) and np.all(b.get_dec_status() == 0)
    vals = b.get_values()
    assert np.array_equal(b.get_decoded(), vals)
    lengths = b.get_lengths()
    for t in range(0, nt, 37):
        ref, _ = oracle.lsop12_encode(2, nr, nc, vals[t], False, value_checksum=True)
        assert b.get_packing(t, int(lengths[t])) == ref, t
    b.free()
