"""GPU parity tests of the CodecCanonHuffman path (gf_canon_* of the C ABI) against the CPU oracle
(oracle/gvrs_oracle_canon.c), bit for bit.  The oracle itself is unpinned for this codec (no reference
fixture holds canonical-Huffman bytes)."""
import numpy as np
import pytest

import oracle
from tilegen import KINDS, NULL, add_nulls, make_tile

pytestmark = pytest.mark.gpu

SHAPES = [(10, 10), (2, 2), (7, 9), (1, 2), (1, 37), (33, 65), (120, 150), (200, 200), (3, 2), (64, 64), (5, 300), (9, 1)]


@pytest.fixture(scope="module")
def codec():
    import gridfour_amd
    return gridfour_amd.CodecCanonHuffmanHip()


def _check_tiles(codec, n_rows, n_cols, tiles, codec_index=3):
    packs, preds, status = codec.encode_batch(codec_index, n_rows, n_cols, tiles)
    for t, v in enumerate(tiles):
        try:
            ref, used = oracle.codec_canon_encode(codec_index, n_rows, n_cols, v)
        except ValueError as ex:
            want = -2 if "rc=-2" in str(ex) else -4      # AIOOBE (Linear, 1 column) / IllegalArgument (Triangle, 1 row)
            assert packs[t] is None and status[t] == want, (t, status[t], str(ex))
            continue
        if ref is None:
            assert packs[t] is None and status[t] == 1, (t, status[t])
            continue
        assert status[t] == 0, (t, status[t])
        assert preds[t] == used, ("predictor", t, preds[t], used)
        assert len(packs[t]) == len(ref), ("length", t, len(packs[t]), len(ref), used)
        if packs[t] != ref:
            first = next(i for i in range(len(ref)) if packs[t][i] != ref[i])
            raise AssertionError("tile %d differs at byte %d of %d (model %d): got %s want %s" % (
                t, first, len(ref), used, packs[t][first:first + 8].hex(), ref[first:first + 8].hex()))
    good = [p for p in packs if p is not None]
    if not good:
        return
    vals, st = codec.decode_batch(n_rows, n_cols, good)
    k = 0
    for t, v in enumerate(tiles):
        if packs[t] is None:
            continue
        assert st[k] == 0, (t, st[k])
        want = oracle.codec_canon_decode(n_rows, n_cols, packs[t])      # == v except for the reference's range quirk
        if not np.array_equal(vals[k], want):
            bad = np.nonzero(vals[k] != want)[0]
            raise AssertionError("decode of tile %d (model %d) differs at %d cells, first %d: got %d want %d" % (
                t, preds[t], bad.size, bad[0], vals[k][bad[0]], want[bad[0]]))
        k += 1


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d" % s)
def test_encode_decode_parity_kinds(codec, shape):
    n_rows, n_cols = shape
    tiles = np.stack([make_tile(k, n_rows, n_cols) for k in KINDS])
    _check_tiles(codec, n_rows, n_cols, tiles)


@pytest.mark.parametrize("shape", [(10, 10), (6, 17), (1, 9), (9, 1), (120, 150), (50, 50)], ids=lambda s: "%dx%d" % s)
def test_nulls_parity(codec, shape):
    n_rows, n_cols = shape
    tiles = []
    for frac, blocks in ((0.02, False), (0.3, False), (0.9, False), (0.2, True)):
        tiles.append(add_nulls(make_tile("smooth", n_rows, n_cols), n_rows, n_cols, frac, blocks=blocks))
    v = make_tile("smooth", n_rows, n_cols)
    v[0] = NULL
    tiles.append(v)
    v = make_tile("ramp", n_rows, n_cols)
    v[::n_cols] = NULL
    tiles.append(v)
    tiles.append(np.full(n_rows * n_cols, NULL, np.int32))
    _check_tiles(codec, n_rows, n_cols, np.stack(tiles))


def test_escape_classes_and_quirk(codec):
    # every escape class of CanonicalHuffman.java:223-275 incl. the -8333608 / -8388608 mismatch
    rng = np.random.default_rng(5)
    base = rng.integers(-3, 4, 40 * 50).cumsum().astype(np.int64)
    tiles = []
    for span in (500, 2000, 8000, 32000, 8_000_000, 2_000_000_000):
        v = base.copy()
        idx = rng.integers(1, v.size, 60)
        v[idx] += rng.integers(-span, span, idx.size)
        tiles.append(v.astype(np.int32))
    v = base.copy()
    v[100] += -8388608 + 17                       # a residual inside the quirk's gap
    v[101:] += -8388608 + 17
    tiles.append(v.astype(np.int32))
    _check_tiles(codec, 40, 50, np.stack(tiles))


def test_length_limited_codes():
    # Fibonacci-like residual counts: unrestricted Huffman depth > 15 -> the PackageMerge path on the device.
    # The tile is built so that the Differencing residuals are exactly that multiset; only Differencing is tried.
    import gridfour_amd
    n_rows, n_cols = 150, 190
    fib = [1, 1]
    while len(fib) < 22:
        fib.append(fib[-1] + fib[-2])
    res = np.concatenate([np.full(c, i - 11, np.int32) for i, c in enumerate(fib)])
    np.random.default_rng(3).shuffle(res)
    res = np.resize(res, n_rows * n_cols - 1)
    v = oracle.predictor_decode_int(1, 1000, n_rows, n_cols, res)
    ref, used = oracle.codec_canon_encode(4, n_rows, n_cols, v, predictor_mask=1)
    _, _, cl = oracle.canon_encode(res)
    assert cl.max() <= 15 and used == 1
    ctx = gridfour_amd.GvrsHipContext(0)
    b = gridfour_amd.DeviceTileBatch(ctx, n_rows, n_cols, 1, codec="canon")
    b.values.upload(v)
    b.encode(codec_index=4, predictor_mask=1)
    b.decode()
    ctx.synchronize()
    assert b.get_enc_status()[0] == 0 and b.get_dec_status()[0] == 0
    assert b.get_packing(0) == ref
    assert np.array_equal(b.get_decoded()[0], v)
    b.free()


def test_single_tile_interface(codec):
    v = make_tile("smooth", 120, 150)
    ref, used = oracle.codec_canon_encode(2, 120, 150, v)
    got = codec.encode(2, 120, 150, v)
    assert got == ref
    assert np.array_equal(codec.decode(120, 150, got), v)
    assert codec.encode(2, 4, 4, np.full(16, NULL, np.int32)) is None
    u = codec.encode(7, 4, 4, np.full(16, 9, np.int32))
    assert u == bytes([7, 0, 9, 0, 0, 0])
    assert np.array_equal(codec.decode(4, 4, u), np.full(16, 9, np.int32))


def test_decode_errors(codec):
    v = make_tile("smooth", 20, 30)
    good = codec.encode(1, 20, 30, v)
    with pytest.raises(IOError):
        codec.decode(20, 30, good[:1] + bytes([9]) + good[2:])      # unknown predictor
    with pytest.raises(IOError):
        codec.decode(20, 30, good[:len(good) // 2])                 # text cut off: no end-of-text symbol
    with pytest.raises(IOError):
        codec.decode(20, 30, good[:4])                              # shorter than the header


def test_device_batch_dem_roundtrip_and_sampled_parity():
    import gridfour_amd
    ctx = gridfour_amd.GvrsHipContext(0)
    n_rows, n_cols, nt = 120, 150, 600
    b = gridfour_amd.DeviceTileBatch(ctx, n_rows, n_cols, nt, codec="canon")
    b.synth_dem(0x9E3779B97F4A7C15 + 3, 30)
    b.encode(codec_index=1)
    b.decode()
    ctx.synchronize()
    assert np.all(b.get_enc_status() == 0) and np.all(b.get_dec_status() == 0)
    vals = b.get_values()
    assert np.array_equal(b.get_decoded(), vals)
    lengths = b.get_lengths()
    preds = b.get_predictors()
    for t in range(0, nt, 37):
        ref, used = oracle.codec_canon_encode(1, n_rows, n_cols, vals[t])
        assert preds[t] == used and lengths[t] == len(ref)
        assert b.get_packing(t, int(lengths[t])) == ref
    b.free()


@pytest.mark.parametrize("shape", [(120, 150), (200, 200), (100, 256), (90, 257), (64, 193), (129, 128), (2, 8000), (300, 64), (500, 40)],
                         ids=lambda s: "%dx%d" % s)
def test_triangle_packings_every_kind(codec, shape):
    """Packings the oracle made with the Triangle predictor alone, through the decoder: the 512-thread build turns the staged
    residuals of such tiles into the tile in one go (cd_fused_triangle) -- small residuals from the LDS stage, wide ones and
    whatever the stage has no room for read back from their cells; shapes either side of its limits (256 columns, the scratch
    of (waves + 1) rows) take the two-step path."""
    n_rows, n_cols = shape
    tiles = [make_tile(k, n_rows, n_cols) for k in KINDS]
    tiles.append(add_nulls(make_tile("smooth", n_rows, n_cols), n_rows, n_cols, 0.05, blocks=False))
    rng = np.random.default_rng(n_rows * 1000 + n_cols)
    v = rng.integers(-2, 3, n_rows * n_cols).cumsum().astype(np.int64)      # small residuals, a few wide ones, int32 wrap-around
    idx = rng.integers(0, v.size, 40)
    v[idx] += rng.integers(-2_000_000_000, 2_000_000_000, idx.size)
    tiles.append((v & 0xFFFFFFFF).astype(np.uint32).view(np.int32))
    packs = []
    for v in tiles:
        p, used = oracle.codec_canon_encode(3, n_rows, n_cols, v, predictor_mask=1 << 2)
        if p is not None:
            assert used in (0, 3)                                  # (0: the uniform tile's six-byte packing)
            packs.append(p)
    assert len(packs) >= 6
    vals, st = codec.decode_batch(n_rows, n_cols, packs)
    for k, p in enumerate(packs):
        assert st[k] == 0, (k, st[k])
        want = oracle.codec_canon_decode(n_rows, n_cols, p)
        bad = np.nonzero(vals[k] != want)[0]
        assert bad.size == 0, (k, bad.size, int(bad[0]), int(vals[k][bad[0]]), int(want[bad[0]]))


def test_saved_soak_case_quirk_stream(codec, golden_dir):
    """tools/soak.py, seed 30031004 case 9894 (3 x 6, residuals of ~1e8 under Differencing): the packing the encoder writes for
    one of its tiles holds a residual inside the gap between CanonicalHuffman.java:258 and :395 -- the reference cannot read it
    back; the library writes the same bytes and refuses them with the same status as the oracle."""
    import os
    d = np.load(os.path.join(golden_dir, "soak", "soak_r03_30031004_case9894.npz"))
    tiles, (n_rows, n_cols) = d["tiles"], map(int, d["shape"])
    n_rows, n_cols = int(d["shape"][0]), int(d["shape"][1])
    packs, _, status = codec.encode_batch(7, n_rows, n_cols, tiles)
    refs = [oracle.codec_canon_encode(7, n_rows, n_cols, v)[0] for v in tiles]
    assert [p for p in packs] == refs and (np.asarray(status) == 0).all()
    vals, st = codec.decode_batch(n_rows, n_cols, refs)
    n_bad = 0
    for k, p in enumerate(refs):
        try:
            want = oracle.codec_canon_decode(n_rows, n_cols, p)
        except IOError as ex:
            n_bad += 1
            assert st[k] == (-2 if "rc=-2" in str(ex) else -1), (k, st[k], str(ex))
            continue
        assert st[k] == 0 and np.array_equal(vals[k], want), k
    assert n_bad == 1
