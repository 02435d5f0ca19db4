"""GPU parity tests of the CodecDeflate path (predictor + CodecM32 on the GPU, zlib level 6 on the host) against the reference's
own fixtures (Sample04/05/07: 40-byte packings reproduced byte for byte) and against the CPU oracle."""
import os
import struct

import numpy as np
import pytest

import oracle
from gvrs_walk import tile_packings
from tilegen import KINDS, NULL, add_nulls, make_tile

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    import gridfour_amd
    return gridfour_amd.CodecDeflateHip()


@pytest.mark.parametrize("name", ["Sample04_ShortComp.gvrs", "Sample05_IntComp.gvrs", "Sample07_ICFComp.gvrs"])
def test_reference_fixtures_byte_exact(codec, golden_dir, name):
    tiles = tile_packings(os.path.join(golden_dir, "ref_samples", name))
    vals, packs = [], []
    for idx in sorted(tiles):
        (packing,) = tiles[idx]
        tr, tc = divmod(idx, 2)
        rows = np.arange(50)[:, None] + tr * 50
        cols = np.arange(50)[None, :] + tc * 50
        vals.append((rows * 100 + cols - 1).astype(np.int32).ravel())
        packs.append(packing)
    got, preds, status = codec.encode_batch(1, 50, 50, np.stack(vals))
    assert (status == 0).all() and (preds == oracle.PM_LINEAR).all()
    assert got == packs                                      # the GPU + zlib packing IS the reference's stored packing
    dec, st = codec.decode_batch(50, 50, packs)
    assert (st == 0).all() and np.array_equal(dec, np.stack(vals))


def _check(codec, n_rows, n_cols, tiles):
    packs, preds, status = codec.encode_batch(2, n_rows, n_cols, tiles)
    good, idx = [], []
    for t, v in enumerate(tiles):
        try:
            ref, used = oracle.codec_deflate_encode(2, n_rows, n_cols, v)
        except ValueError:
            assert packs[t] is None and status[t] == -2, (t, status[t])
            continue
        if ref is None:
            assert packs[t] is None and status[t] == 1, (t, status[t])
            continue
        assert status[t] == 0 and preds[t] == used, (t, status[t], preds[t], used)
        assert packs[t] == ref, (t, len(packs[t]), len(ref))
        good.append(ref)
        idx.append(t)
    if good:
        vals, st = codec.decode_batch(n_rows, n_cols, good)
        for k, t in enumerate(idx):
            assert st[k] == 0 and np.array_equal(vals[k], tiles[t]), (t, st[k])


@pytest.mark.parametrize("shape", [(10, 10), (2, 2), (7, 9), (1, 37), (33, 65), (120, 150), (200, 200), (5, 300), (9, 1)],
                         ids=lambda s: "%dx%d" % s)
def test_parity_kinds(codec, shape):
    n_rows, n_cols = shape
    _check(codec, n_rows, n_cols, np.stack([make_tile(k, n_rows, n_cols) for k in KINDS]))


@pytest.mark.parametrize("shape", [(10, 10), (6, 17), (1, 9), (120, 150)], ids=lambda s: "%dx%d" % s)
def test_nulls_parity(codec, shape):
    n_rows, n_cols = shape
    tiles = [add_nulls(make_tile("smooth", n_rows, n_cols), n_rows, n_cols, f, blocks=b)
             for f, b in ((0.02, False), (0.3, False), (0.9, False), (0.2, True))]
    tiles.append(np.full(n_rows * n_cols, NULL, np.int32))
    _check(codec, n_rows, n_cols, np.stack(tiles))


def test_m32_stage_on_device():
    import gridfour_amd
    from gridfour_amd import DeviceBuffer, lib
    ctx = gridfour_amd.GvrsHipContext(0)
    n_rows, n_cols, nt = 64, 96, 20
    tiles = np.stack([make_tile(KINDS[i % len(KINDS)], n_rows, n_cols, seed=i) for i in range(nt)])
    sub = int(lib().gf_m32_max_stream(n_rows, n_cols))
    dv, ds = DeviceBuffer(ctx, tiles.nbytes), DeviceBuffer(ctx, nt * 3 * sub + 16)
    dl, dm, dsd, dst = DeviceBuffer(ctx, nt * 12), DeviceBuffer(ctx, nt * 3), DeviceBuffer(ctx, nt * 4), DeviceBuffer(ctx, nt * 4)
    dv.upload(tiles)
    gridfour_amd._lib.check(lib().gf_m32_encode_batch_i32_dev(ctx.handle, None, n_rows, n_cols, nt, dv.ptr, ds.ptr, sub, dl.ptr,
                                                               dm.ptr, dsd.ptr, dst.ptr), "m32 encode")
    ctx.synchronize()
    lens, models, seeds = dl.download(np.uint32, nt * 3), dm.download(np.uint8, nt * 3), dsd.download(np.int32, nt)
    streams = ds.download(np.uint8, nt * 3 * sub)
    assert (dst.download(np.int32, nt) == 0).all()
    for t in range(nt):
        for p in range(3):
            m32, seed = oracle.predictor_encode(p + 1, n_rows, n_cols, tiles[t])
            assert models[t * 3 + p] == p + 1 and lens[t * 3 + p] == len(m32) and seeds[t] == seed
            off = (t * 3 + p) * sub
            assert bytes(streams[off:off + len(m32)]) == m32, (t, p)
    for b in (dv, ds, dl, dm, dsd, dst):
        b.free()


def test_codec_master_standard_list():
    # CodecMaster.java:150-169 over the standard codec list {Huffman, Deflate, Float, CanonicalHuffman}
    import gridfour_amd
    master = gridfour_amd.CodecMasterHip()
    n_rows, n_cols = 60, 90
    tiles = np.stack([make_tile(k, n_rows, n_cols) for k in KINDS] +
                     [add_nulls(make_tile("smooth", n_rows, n_cols), n_rows, n_cols, 0.1), np.full(n_rows * n_cols, NULL, np.int32)])
    packs, used, status = master.encode_batch(n_rows, n_cols, tiles)
    encoders = {0: oracle.codec_huffman_encode, 1: lambda ci, r, c, v: oracle.codec_deflate_encode(ci, r, c, v),
                3: oracle.codec_canon_encode}
    for t, v in enumerate(tiles):
        best, best_k = None, 255
        for k in (0, 1, 3):
            ref = encoders[k](k, n_rows, n_cols, v)[0]
            if ref is not None and (best is None or len(ref) < len(best)):
                best, best_k = ref, k
        if best is None:
            assert packs[t] is None and used[t] == 255 and status[t] == 1
        else:
            assert status[t] == 0 and used[t] == best_k and packs[t] == best, (t, used[t], best_k)
    good = [p for p in packs if p is not None]
    vals, st = master.decode_batch(n_rows, n_cols, good)
    k = 0
    for t, v in enumerate(tiles):
        if packs[t] is None:
            continue
        assert st[k] == 0 and np.array_equal(vals[k], v), t
        k += 1
    _, st = master.decode_batch(n_rows, n_cols, [bytes([9]) + good[0][1:]])
    assert st[0] == -1                                   # "Invalid compression-type code"


def test_tile_payloads_equal_the_fixture_records(golden_dir):
    # the tile records of Sample05_IntComp.gvrs hold [int32 tileIndex][int32 n][packing]: everything behind the tile index
    # is reproduced from the cell values (standard codec list, CodecDeflate wins on these ramps as it did in the reference)
    import gridfour_amd
    master = gridfour_amd.CodecMasterHip()
    tiles = tile_packings(os.path.join(golden_dir, "ref_samples", "Sample05_IntComp.gvrs"))
    vals, want = [], []
    for idx in sorted(tiles):
        (packing,) = tiles[idx]
        tr, tc = divmod(idx, 2)
        rows = np.arange(50)[:, None] + tr * 50
        cols = np.arange(50)[None, :] + tc * 50
        vals.append((rows * 100 + cols - 1).astype(np.int32).ravel())
        want.append(struct.pack("<i", len(packing)) + packing)
    payloads, used = master.tile_payloads(50, 50, np.stack(vals))
    assert payloads == want and (used == 1).all()
    back, st = master.tiles_from_payloads(50, 50, payloads)
    assert (st == 0).all() and np.array_equal(back, np.stack(vals))
    # incompressible and all-null tiles fall back to the raw cells (TileElementInt.java:198-204)
    noise = make_tile("noise32", 50, 50)
    nulls = np.full(2500, NULL, np.int32)
    payloads, used = master.tile_payloads(50, 50, np.stack([noise, nulls]))
    assert (used == 255).all()
    assert payloads[0] == struct.pack("<i", 10000) + noise.astype("<i4").tobytes()
    back, st = master.tiles_from_payloads(50, 50, payloads)
    assert (st == 0).all() and np.array_equal(back[0], noise) and np.array_equal(back[1], nulls)


def test_decode_returns_null_when_the_inflater_gives_nothing(codec):
    """CodecDeflate.decode :139-154: `int test = inflater.inflate(codeM32s); if (test > 0) {...} return null;` -- a packing
    whose stream inflates to nothing (here: nM32 = 0, so there is no room) is null, not an exception; a stream that is no
    deflate data is the IOException that wraps DataFormatException."""
    import zlib
    packing = bytes([2, 1]) + struct.pack("<i", 1234) + struct.pack("<i", 0) + zlib.compress(b"", 6)
    assert codec.decode(8, 8, packing) is None
    rc = oracle.lib().gvo_codec_deflate_decode(8, 8, oracle._p(oracle._u8(packing), oracle.C.c_uint8), len(packing),
                                               oracle._p(np.zeros(64, np.int32), oracle.C.c_int32))
    assert rc == oracle.DECLINED
    not_deflate = bytes([2, 1]) + struct.pack("<i", 1234) + struct.pack("<i", 63) + bytes([0x78, 0x9C, 0xFF, 0xFF, 0xFF, 0xFF])
    with pytest.raises(IOError):
        codec.decode(8, 8, not_deflate)


def test_every_single_bit_flip_matches_the_oracle(codec):
    """Every one-bit damage of a CodecDeflate packing, all in one batch: where the oracle (the host's zlib driven as
    java.util.zip.Inflater drives it) still decodes, the device decodes to the same cells; where the oracle returns null
    (nothing inflated) so does the device; where the oracle throws, the device reports an error."""
    nr, nc = 12, 40
    v = make_tile("smooth", nr, nc, seed=5)
    good = codec.encode(2, nr, nc, v)
    assert good is not None
    _every_bit_flip(codec, nr, nc, good)


def test_every_single_bit_flip_of_a_long_run_packing(codec):
    """The same over a packing that is nearly all run-length codes in its code-length header (a 171x123 tile of constant
    residuals, 53 bytes; tools/soak.py, seed 31415926): a stream that ends inside the extra bits of a repeat code waits for
    input -- Inflater.inflate returns 0, decode returns null -- it is not 'invalid bit length repeat'."""
    good = bytes.fromhex("07017910020028520000789cedc13101000000c2a0f54f6d0d0fa00000000000000000000000000000000000000000783052280001")
    bad = bytearray(good)
    bad[20] ^= 0x80
    _, st = codec.decode_batch(171, 123, [bytes(bad)])
    assert st[0] == 1
    _every_bit_flip(codec, 171, 123, good, need_errors=False)


def _every_bit_flip(codec, nr, nc, good, need_errors=True):
    packs = []
    for i in range(1, len(good)):                     # (byte 0 is the codec index, not looked at by decode)
        for b in range(8):
            x = bytearray(good)
            x[i] ^= 1 << b
            packs.append(bytes(x))
    vals, st = codec.decode_batch(nr, nc, packs)
    n_ok = n_null = n_err = 0
    for k, pk in enumerate(packs):
        out = np.zeros(nr * nc, np.int32)
        rc = oracle.lib().gvo_codec_deflate_decode(nr, nc, oracle._p(oracle._u8(pk), oracle.C.c_uint8), len(pk),
                                                   oracle._p(out, oracle.C.c_int32))
        where = (k // 8 + 1, k % 8)
        n_m32 = struct.unpack_from("<I", pk, 6)[0]
        if n_m32 > 6 * nr * nc:
            # documented deviation: a byte count no encoder can produce (more than six bytes per cell) is
            # rejected up front instead of sizing buffers by it
            assert st[k] < 0, (where, int(st[k]))
            continue
        if rc == oracle.OK:
            assert st[k] == 0 and np.array_equal(vals[k], out), (where, int(st[k]))
            n_ok += 1
        elif rc == oracle.DECLINED:
            assert st[k] == 1, (where, int(st[k]))
            n_null += 1
        else:
            assert st[k] < 0, (where, int(st[k]), rc)
            n_err += 1
    assert n_ok > 0 and (n_err > 0 or not need_errors)


def test_stream_that_ends_early_leaves_zeros_like_a_fresh_java_array(codec):
    """Found by tools/soak.py: one flipped bit makes the zlib stream run out of input after 3,253 of 16,824 bytes.
    Inflater.inflate returns what it has, the rest of `new byte[nM32]` is zero, and the predictor reads on into it
    (CodecDeflate.java:139-148) -- the device must see zeros there too, not what an earlier tile left in its scratch."""
    import base64
    pk = base64.b64decode("BwE87f//uEEAAHic7cQxDQAACAOwG9UL8uYKDfzt0XS2E9u2bZu2bdu2bdu2bdu2bdu2bdu2bdu2bT8/jzft9Q==")
    nr, nc = 79, 71
    want = oracle.codec_deflate_decode(nr, nc, pk)
    filler = codec.encode(1, nr, nc, make_tile("noise16", nr, nc, seed=3))      # leaves non-zero bytes in the scratch first
    vals, st = codec.decode_batch(nr, nc, [filler, filler, filler])
    assert (st == 0).all()
    for _ in range(2):
        vals, st = codec.decode_batch(nr, nc, [pk, pk, filler, pk])
        assert list(st) == [0, 0, 0, 0]
        for k in (0, 1, 3):
            assert np.array_equal(vals[k], want), k


def test_match_behind_the_room_is_not_looked_at():
    """Found by tools/soak.py: the last-block bit of a stored block flipped, so zlib goes on into the Adler-32 bytes as if
    they were another block; they happen to spell a match whose distance reaches before the start of the output.  zlib only
    validates a distance when it has room to copy (inflate.c, state MATCH), and the room is exactly used up: Inflater.inflate
    returns the 228 bytes without an exception, and so must the device."""
    import base64
    import gridfour_amd
    pk = base64.b64decode("BwLen+ry5AAAAHicAOQAG/+BhOWO+UOBhIGTnF6Bgs2thUF/g+C400+BhuThzVR/i4eCD4GF09XabYGD78eKJ3/v1rVJgYC0wfghf4Ll8+BU"
                          "gYOi9OQOgYPo/NwNgYK59qRdf4bO/p08gYHG5qRMgYTww/sFf4HRgbt7f4Twt+4KgYSfsMVEgYaByY4AgYPhjIIVgYbLptpGgYK2hJp3"
                          "gYOu9+EygYbh2osNgYb2kqYgf4Hfzv0wf+XMt12BqLSlSX+FgY+NM3+T5YUrgYCbuZNBf4X0qORMf4Lfkrg0gZXksTWBgt/11wh/heaQ"
                          "9yCBhejemQYbuoUw")
    codec = gridfour_amd.CodecDeflateHip()
    want = oracle.codec_deflate_decode(10, 4, pk)
    vals, st = codec.decode_batch(10, 4, [pk])
    assert st[0] == 0 and np.array_equal(vals[0], want)


def _nulls_packings(n_rows, n_cols, seed, residuals):
    """CodecDeflate and CodecHuffman packings of predictor 4 (DifferencingWithNulls) around a hand-made residual stream."""
    import zlib
    m32 = oracle.m32_encode_seq(np.asarray(residuals, np.int64).astype(np.int32))
    head = bytes([0, 4]) + struct.pack("<iI", seed, len(m32))
    return head + zlib.compress(m32, 6), oracle.huffman_encode(np.frombuffer(m32, np.uint8), 80, head)[0]


def test_a_sum_that_comes_out_as_the_null_code_is_not_a_null():
    """PredictorModelDifferencingWithNulls.decode (:137-166): inside a row the null flag follows the RESIDUAL, at a row start
    the VALUE of the previous row's first cell.  A sum can come out as Integer.MIN_VALUE -- the null code -- without a null
    residual (damaged or hand-made input only; found by tools/soak.py, seed 424242).  The row goes on from that sum; the
    next row then starts from the seed."""
    import gridfour_amd
    ctx = gridfour_amd.GvrsHipContext(0)
    null, imin = int(NULL), -2**31
    nr, nc, seed = 6, 5, 5
    wrap = lambda x: (x + 2**31) % 2**32 - 2**31
    res = np.array([[wrap(imin - seed), 7, null, 3, 4],        # (0,0): seed + r = MIN_VALUE; (0,1) continues from it
                    [wrap(imin - seed), 1, 2, 3, 4],           # row start: prior MIN_VALUE counts as null -> seed again -> MIN_VALUE again
                    [9, wrap(imin - seed - 9), 2, null, 1],    # the null-code sum in the middle of a row
                    [null, 4, 5, 6, 7],                        # a real null first
                    [wrap(imin - seed), null, null, 8, 9],     # nulls straight after the sum
                    [1, 2, 3, 4, 5]], np.int64)
    cases = [(nr, nc, seed, res.ravel())]
    rng = np.random.default_rng(77)
    for _ in range(6):                                          # every row starts with such a sum, random rest, odd shapes
        r, c, s = int(rng.integers(1, 40)), int(rng.integers(1, 9)), int(rng.integers(-1000, 1000))
        x = rng.integers(-300, 300, (r, c)).astype(np.int64)
        x[rng.random((r, c)) < 0.15] = null
        x[:, 0] = wrap(imin - s)
        if c > 2:
            x[::3, 1] = 11                                      # a plain value after the sum
        cases.append((r, c, s, x.ravel()))
    for r, c, s, x in cases:
        pk_deflate, pk_huffman = _nulls_packings(r, c, s, x)
        for codec, dec, pk in ((gridfour_amd.CodecDeflateHip(context=ctx), oracle.codec_deflate_decode, pk_deflate),
                               (gridfour_amd.CodecHuffmanHip(context=ctx), oracle.codec_huffman_decode, pk_huffman)):
            ref = dec(r, c, pk)
            assert (ref[::c] == imin).sum() >= 1                # the case is in there
            vals, st = codec.decode_batch(r, c, [pk, pk])
            assert st[0] == 0 and st[1] == 0 and np.array_equal(vals[0], ref) and np.array_equal(vals[1], ref), (r, c, s)


def test_a_header_that_asks_for_a_preset_dictionary_inflates_nothing(codec):
    """zlib answers Z_NEED_DICT, java.util.zip.Inflater.inflate returns 0 with needsDictionary() set, CodecDeflate.decode
    returns null (:141-154) -- no exception.  The same when the stream ends behind such a header."""
    import zlib
    raw = zlib.compress(b"\x01" * 50, 6)
    for stream in (bytes([0x78, 0x20, 0, 0, 0, 1]) + raw[2:], bytes([0x78, 0x20])):
        pk = bytes([0, 1]) + struct.pack("<iI", 5, 50) + stream
        with pytest.raises(IOError, match="rc=1"):
            oracle.codec_deflate_decode(5, 10, pk)
        _, st = codec.decode_batch(5, 10, [pk, pk])
        assert st[0] == 1 and st[1] == 1


def test_introducer_candidates_are_told_exactly():
    """tools/soak.py, seed 30031003 (round 3): a damaged CodecDeflate packing whose M32 bytes hold a six-byte value that ends in
    0x81 (five continuation bytes: CodecM32.java:335-356 returns the bits read so far) with the null code 0x80 right behind it.
    The value starts are worked out from byte masks (m32_mark_starts); the usual zero-byte test marks a 0x01 above a zero byte
    too, so 0x80 behind 0x81 looked like an introducer and swallowed the two values behind it.  The saved packing, and
    hand-made streams with the byte pairs that test gets wrong (0x81 0x80, 0x7f 0x7e) at every alignment."""
    import os
    import zlib
    import gridfour_amd
    ctx = gridfour_amd.GvrsHipContext(0)
    codec = gridfour_amd.CodecDeflateHip(context=ctx)
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "soak", "soak_r03_30031003_case1694.npz"))
    nr, nc = map(int, d["shape"])
    bad = bytes(d["bad"])
    vals, st = codec.decode_batch(nr, nc, [bad])
    assert st[0] == 0 and np.array_equal(vals[0], oracle.codec_deflate_decode(nr, nc, bad))
    # hand-made: model 4 (every cell in the stream), values of one byte around the pattern
    nr, nc = 9, 16
    rng = np.random.default_rng(3)
    packs = []
    for pattern in (bytes([0x7f, 0x81, 0xaa, 0x81, 0xec, 0x81, 0x80, 0xe5, 0x21, 0xd0]),      # the soak's
                    bytes([0x81, 0xff, 0xff, 0xff, 0xff, 0x7f, 0x7e, 0x05]),                  # ... 0x7f (a payload byte) 0x7e (a start)
                    bytes([0x7f, 0x81, 0x81, 0x81, 0x81, 0x81, 0x80, 0x80, 0x01]),
                    bytes([0x81, 0xff, 0xff, 0xff, 0xff, 0x81, 0x80, 0x7f, 0x7e])):
        for shift in range(8):
            body = bytearray(rng.integers(0, 100, nr * nc + 32).astype(np.uint8))             # plain one-byte values
            at = 40 + shift
            body[at:at + len(pattern)] = pattern
            # as many bytes as the stream needs values for (the pattern shortens the count: pad generously, then cut by parsing)
            m32 = bytes(body)
            n_vals, i = 0, 0
            while n_vals < nr * nc and i < len(m32):
                b = m32[i]
                i += 1
                if b in (0x7f, 0x81):
                    for _ in range(5):
                        x = m32[i]
                        i += 1
                        if not x & 0x80:
                            break
                n_vals += 1
            m32 = m32[:i]
            packs.append(bytes([3, 4]) + (5).to_bytes(4, "little", signed=True) + len(m32).to_bytes(4, "little") + zlib.compress(m32, 6))
    vals, st = codec.decode_batch(nr, nc, packs)
    for k, pk in enumerate(packs):
        ref = oracle.codec_deflate_decode(nr, nc, pk)
        assert st[k] == 0 and np.array_equal(vals[k], ref), k
