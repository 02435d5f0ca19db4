"""GPU parity tests: the HIP codec (through the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

import oracle
from tilegen import KINDS, NULL, add_nulls, make_tile

pytestmark = pytest.mark.gpu

SHAPES = [(10, 10), (2, 2), (7, 9), (1, 2), (1, 37), (33, 65), (120, 150), (200, 200), (3, 2), (64, 64), (5, 300)]


@pytest.fixture(scope="module")
def codec():
    import gridfour_amd
    return gridfour_amd.CodecHuffmanHip()


def _check_tiles(codec, n_rows, n_cols, tiles, codec_index=3):
    packs, preds, status = codec.encode_batch(codec_index, n_rows, n_cols, tiles)
    for t, v in enumerate(tiles):
        try:
            ref, used = oracle.codec_huffman_encode(codec_index, n_rows, n_cols, v)
        except ValueError:
            # the reference throws ArrayIndexOutOfBounds (no nulls and nCols < 2)
            assert packs[t] is None and status[t] == -2, (t, status[t])
            continue
        if ref is None:
            assert packs[t] is None and status[t] == 1, (t, status[t])
            continue
        assert status[t] == 0, (t, status[t])
        assert preds[t] == used, ("predictor", t, preds[t], used)
        assert len(packs[t]) == len(ref), ("length", t, len(packs[t]), len(ref))
        if packs[t] != ref:
            first = next(i for i in range(len(ref)) if packs[t][i] != ref[i])
            raise AssertionError("tile %d differs at byte %d of %d (model %d): got %s want %s" % (
                t, first, len(ref), used, packs[t][first:first + 8].hex(), ref[first:first + 8].hex()))
    good = [p for p in packs if p is not None]
    vals, st = codec.decode_batch(n_rows, n_cols, good)
    k = 0
    for t, v in enumerate(tiles):
        if packs[t] is None:
            continue
        assert st[k] == 0, (t, st[k])
        if not np.array_equal(vals[k], v):
            bad = np.nonzero(vals[k] != v)[0]
            raise AssertionError("decode of tile %d (model %d) differs at %d cells, first %d: got %d want %d" % (
                t, preds[t], bad.size, bad[0], vals[k][bad[0]], v[bad[0]]))
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
    v[::n_cols] = NULL                      # whole first column null: exercises the row-start rule
    tiles.append(v)
    tiles.append(np.full(n_rows * n_cols, NULL, np.int32))   # all null -> declined
    _check_tiles(codec, n_rows, n_cols, np.stack(tiles))


def test_single_tile_interface(codec):
    v = make_tile("smooth", 120, 150)
    ref, used = oracle.codec_huffman_encode(2, 120, 150, v)
    got = codec.encode(2, 120, 150, v)
    assert got == ref
    assert np.array_equal(codec.decode(120, 150, got), v)
    assert codec.encode(0, 4, 4, np.full(16, NULL, np.int32)) is None
    assert codec.implementsIntegerEncoding() and not codec.implementsFloatingPointEncoding()
    assert codec.encodeFloats(0, 2, 2, np.zeros(4, np.float32)) is None
    with pytest.raises(IndexError):
        codec.encode(0, 5, 1, np.arange(5, dtype=np.int32))       # reference: AIOOBE (nCols < 2)


def test_decode_of_oracle_packings_and_errors(codec):
    v = make_tile("smooth", 33, 65)
    for mask in (1, 2, 4):
        ref, used = oracle.codec_huffman_encode(1, 33, 65, v, predictor_mask=mask)
        assert np.array_equal(codec.decode(33, 65, ref), v)
    ref, _ = oracle.codec_huffman_encode(1, 33, 65, v)
    bad = bytearray(ref)
    bad[1] = 9
    with pytest.raises(IOError):
        codec.decode(33, 65, bytes(bad))
    with pytest.raises(IOError):
        codec.decode(33, 65, ref[:len(ref) // 2])
    with pytest.raises(IOError):
        codec.decode(33, 65, ref[:7])


def test_each_predictor_alone(codec):
    import gridfour_amd
    from gridfour_amd import DeviceTileBatch
    n_rows, n_cols = 40, 50
    tiles = np.stack([make_tile(k, n_rows, n_cols, seed=3) for k in KINDS])
    # worst-case slots: the incompressible kinds exceed the default (raw-size) slot
    stride = int(gridfour_amd.lib().gf_huffman_max_packing(n_rows, n_cols))
    b = DeviceTileBatch(codec.ctx, n_rows, n_cols, len(tiles), slot_stride=stride)
    b.values.upload(tiles)
    for model in (1, 2, 3):
        b.encode(codec_index=7, predictor_mask=1 << (model - 1))
        codec.ctx.synchronize()
        lengths = b.get_lengths()
        assert (b.get_enc_status() == 0).all()
        assert (b.get_predictors() == model).all()
        for t in range(len(tiles)):
            ref, _ = oracle.codec_huffman_encode(7, n_rows, n_cols, tiles[t], predictor_mask=1 << (model - 1))
            assert b.get_packing(t, int(lengths[t])) == ref, (model, t)
        b.decode()
        codec.ctx.synchronize()
        assert (b.get_dec_status() == 0).all()
        assert np.array_equal(b.get_decoded(), tiles)
    b.free()


def test_golden_oracle_vectors(codec, golden_dir):
    import json
    import os
    vec = json.load(open(os.path.join(golden_dir, "oracle_vectors.json")))
    for case in vec["cases"]:
        v = np.array(case["values"], np.int32)
        got = codec.encode(case["codec_index"], case["n_rows"], case["n_cols"], v)
        want = bytes.fromhex(case["packing"]) if case["packing"] is not None else None
        assert got == want, case["name"]
        if want is not None:
            assert np.array_equal(codec.decode(case["n_rows"], case["n_cols"], want), v), case["name"]


def test_synth_dem_matches_oracle(codec):
    from gridfour_amd import DeviceTileBatch
    b = DeviceTileBatch(codec.ctx, 120, 150, 6)
    b.synth_dem(oracle.DEM_SEED + 2, 144, tile0=141)
    codec.ctx.synchronize()
    assert np.array_equal(b.get_values(), oracle.dem_tiles(oracle.DEM_SEED + 2, 120, 150, 144, 141, 6))
    b.free()


def test_synth_dem_masked_matches_oracle(codec):
    """The nulls workload: 5 % of the grid's 16 x 16 blocks are null; generator == oracle, and the tiles take the
    nulls predictor through the batch path with the oracle's bytes."""
    from gridfour_amd import DeviceTileBatch
    b = DeviceTileBatch(codec.ctx, 120, 150, 8)
    b.synth_dem(oracle.DEM_SEED + 2, 144, tile0=141, mask_per_mille=50)
    codec.ctx.synchronize()
    ref = oracle.dem_tiles(oracle.DEM_SEED + 2, 120, 150, 144, 141, 8, mask_per_mille=50)
    vals = b.get_values()
    assert np.array_equal(vals, ref)
    frac = float((ref == np.int32(-2 ** 31)).mean())
    assert 0.01 < frac < 0.12
    b.encode(codec_index=0)
    b.decode()
    codec.ctx.synchronize()
    assert (b.get_enc_status() == 0).all() and (b.get_dec_status() == 0).all()
    assert np.array_equal(b.get_decoded(), vals)
    preds, lengths = b.get_predictors(), b.get_lengths()
    assert (preds == 4).sum() >= 6                               # nearly every tile has a masked block
    for t in range(8):
        want, used = oracle.codec_huffman_encode(0, 120, 150, ref[t])
        assert preds[t] == used and b.get_packing(t) == want
    b.free()


def test_synth_dem_rough_matches_oracle(codec):
    """The rough surface (a tail into 2-3 byte M32 codes; bench.py --workload etopo1_rough): generator == oracle
    across province borders, every predictor wins tiles of the sample, multi-byte values are there, and the batch path gives the
    oracle's bytes for every tile (the decoder's byte path and its general stage side by side)."""
    import struct
    from gridfour_amd import DeviceTileBatch
    seed, tpr = oracle.DEM_SEED + 2, 144
    picks = [0, 5, 6, 7, 8, 13, 14, 1000, 1001, 1008, 2100, 4000, 6143, 6144, 9000, 12000, 12959]
    got, ref = [], []
    b = DeviceTileBatch(codec.ctx, 120, 150, 1)
    for t in picks:
        b.synth_dem(seed, tpr, tile0=t, style=oracle.DEM_STYLE_ROUGH)
        codec.ctx.synchronize()
        got.append(b.get_values()[0].copy())
        ref.append(oracle.dem_tiles(seed, 120, 150, tpr, t, 1, style=oracle.DEM_STYLE_ROUGH)[0])
    b.free()
    assert np.array_equal(np.stack(got), np.stack(ref))
    # a contiguous stretch of the grid through the batch path
    n = 96
    b = DeviceTileBatch(codec.ctx, 120, 150, n)
    b.synth_dem(seed, tpr, tile0=1000, style=oracle.DEM_STYLE_ROUGH)
    codec.ctx.synchronize()
    vals = b.get_values()
    assert np.array_equal(vals, oracle.dem_tiles(seed, 120, 150, tpr, 1000, n, style=oracle.DEM_STYLE_ROUGH))
    b.encode(codec_index=0)
    b.decode()
    codec.ctx.synchronize()
    assert (b.get_enc_status() == 0).all() and (b.get_dec_status() == 0).all()
    assert np.array_equal(b.get_decoded(), vals)
    preds = b.get_predictors()
    wide = 0
    for t in range(n):
        want, used = oracle.codec_huffman_encode(0, 120, 150, vals[t])
        assert preds[t] == used and b.get_packing(t) == want, t
        wide += struct.unpack("<I", want[6:10])[0] != 120 * 150 - 1
    assert 0 < wide < n                                             # tiles with and without multi-byte M32 values
    b.free()


def test_batch_dem_roundtrip_and_sampled_parity(codec):
    """BASELINE config 2 as bench.py --workload dem1024 runs it: 1,024 tiles of 200x200, all three predictors + Huffman
    (round trip on every tile, every 17th tile byte for byte against the oracle)."""
    from gridfour_amd import DeviceTileBatch
    n_rows, n_cols, nt = 200, 200, 1024
    b = DeviceTileBatch(codec.ctx, n_rows, n_cols, nt)
    b.synth_dem(oracle.DEM_SEED + 1, 32)
    b.encode(codec_index=0)
    b.decode()
    codec.ctx.synchronize()
    assert (b.get_enc_status() == 0).all() and (b.get_dec_status() == 0).all()
    vals = b.get_values()
    assert np.array_equal(b.get_decoded(), vals)             # encode -> decode round trip, every tile
    lengths = b.get_lengths()
    preds = b.get_predictors()
    for t in range(0, nt, 17):                               # sampled bit-exactness against the oracle
        ref, used = oracle.codec_huffman_encode(0, n_rows, n_cols, vals[t])
        assert preds[t] == used and b.get_packing(t, int(lengths[t])) == ref, t
    b.free()


def test_compact_blob_roundtrip(codec):
    """Packings gathered into one contiguous blob (unaligned starts) decode identically."""
    tiles = np.stack([make_tile(k, 31, 47, seed=9) for k in KINDS])
    packs, preds, status = codec.encode_batch(1, 31, 47, tiles)
    assert all(p is not None for p in packs)
    vals, st = codec.decode_batch(31, 47, packs)
    assert (st == 0).all() and np.array_equal(vals, tiles)


def test_slot_overflow_is_reported_not_truncated(codec):
    """A packing longer than its slot is flagged GF_OVERFLOW with its true length; the host batch
    API transparently redoes such tiles into a worst-case slot (bytes still equal the oracle's)."""
    from gridfour_amd import DeviceTileBatch
    n_rows, n_cols = 40, 50
    tiles = np.stack([make_tile("noise32", n_rows, n_cols), make_tile("smooth", n_rows, n_cols)])
    b = DeviceTileBatch(codec.ctx, n_rows, n_cols, 2, slot_stride=1024)
    b.values.upload(tiles)
    b.encode(codec_index=0)
    codec.ctx.synchronize()
    st, ln = b.get_enc_status(), b.get_lengths()
    ref0, _ = oracle.codec_huffman_encode(0, n_rows, n_cols, tiles[0])
    ref1, _ = oracle.codec_huffman_encode(0, n_rows, n_cols, tiles[1])
    assert st[0] == 2 and ln[0] == len(ref0)
    assert (st[1], ln[1]) == ((0, len(ref1)) if len(ref1) <= 1024 else (2, len(ref1)))
    b.free()
    packs, _, status = codec.encode_batch(0, n_rows, n_cols, tiles)
    assert list(status) == [0, 0] and packs[0] == ref0 and packs[1] == ref1


def _tie_symbol_sets():
    rng = np.random.default_rng(4321)
    yield np.array([5, 9, 9], np.uint8)
    yield np.tile(np.arange(250, dtype=np.uint8), 3)                    # all counts equal
    yield np.tile(np.arange(7, dtype=np.uint8) * 31, 5)
    yield np.repeat(np.arange(12, dtype=np.uint8), 2 ** np.arange(12))  # powers of two
    yield np.repeat(np.arange(16, dtype=np.uint8) + 100,
                    [1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987])   # Fibonacci: deepest tree
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    yield np.repeat(np.arange(24, dtype=np.uint8) + 7, fib)              # depth 23: beyond both decode LUT levels
    for _ in range(24):
        n_sym = int(rng.integers(2, 251))
        hi = int(rng.choice([1, 2, 3, 5, 20, 1000]))
        counts = rng.integers(1, hi + 1, n_sym)
        syms = rng.permutation(250)[:n_sym].astype(np.uint8)
        data = np.repeat(syms, counts)
        rng.shuffle(data)
        yield data


def test_tree_tie_breaking_on_device(codec):
    """Histograms full of equal counts (leaf/leaf, leaf/branch and branch/branch ties): the packing
    must still equal the reference's linked-list construction byte for byte.  The symbol stream is
    forced by a 1-row tile under the Differencing predictor (residual k <-> M32 byte k)."""
    import gridfour_amd
    from gridfour_amd import DeviceTileBatch
    # 250 usable single-byte residual values: -125..124 (avoids the introducer/null bytes 7f, 80, 81)
    for data in _tie_symbol_sets():
        res = data.astype(np.int64) - 125
        v = np.concatenate([[1000], 1000 + np.cumsum(res)]).astype(np.int32)
        n = v.size
        stride = int(gridfour_amd.lib().gf_huffman_max_packing(1, n))
        b = DeviceTileBatch(codec.ctx, 1, n, 1, slot_stride=stride)
        b.values.upload(v)
        b.encode(codec_index=0, predictor_mask=1)
        codec.ctx.synchronize()
        ref, _ = oracle.codec_huffman_encode(0, 1, n, v, predictor_mask=1)
        assert b.get_enc_status()[0] == 0
        got = b.get_packing(0)
        assert got == ref, ("tree mismatch", len(set(data.tolist())), len(got), len(ref))
        b.decode()
        codec.ctx.synchronize()
        assert b.get_dec_status()[0] == 0 and np.array_equal(b.get_decoded()[0], v)
        b.free()


def test_trees_of_every_shape_in_batches_of_the_wave_per_tile_prepass(codec):
    """The serialised trees of the tie sets above -- two leaves, 250 equal ones, the deepest ones (depth 15 and 23), random ones -- in
    batches of 64 .. 300 packings: the size at which a wave parses a tile's tree, with its scans (round 6: record starts by a 9-state
    transducer, leaf and branch counts by prefix sums, the code lengths by a recurrence over the leaves) in front of the walks.  Each
    packing decodes to the oracle's cells; a damaged copy of each (a flipped bit inside the serialised tree) to the oracle's verdict."""
    sets = list(_tie_symbol_sets())
    rng = np.random.default_rng(99)
    for data in sets:
        res = data.astype(np.int64) - 125
        v = np.concatenate([[1000], 1000 + np.cumsum(res)]).astype(np.int32)
        n = v.size
        ref, _ = oracle.codec_huffman_encode(0, 1, n, v, predictor_mask=1)
        n_leaves = ref[10] + 1
        tree_bits = 10 * n_leaves + 7 if n_leaves >= 2 else 17
        packs, damaged = [], []
        for k in range(int(rng.integers(64, 301))):
            if k % 3 == 2:
                b = bytearray(ref)
                bit = 80 + int(rng.integers(0, tree_bits))
                b[bit >> 3] ^= 1 << (bit & 7)
                packs.append(bytes(b))
                damaged.append(True)
            else:
                packs.append(ref)
                damaged.append(False)
        vals, st = codec.decode_batch(1, n, packs)
        for k, pk in enumerate(packs):
            try:
                want = oracle.codec_huffman_decode(1, n, pk)
            except Exception:
                want = None
            if want is None:
                assert st[k] != 0, (len(set(data.tolist())), k)
            else:
                assert st[k] == 0 and np.array_equal(vals[k], want), (len(set(data.tolist())), k, int(st[k]), damaged[k])


@pytest.mark.parametrize("shape", [(120, 150), (200, 200), (64, 64)], ids=lambda s: "%dx%d" % s)
def test_bits_crowded_into_part_of_the_tile(codec, shape):
    """The packer deals a tile's cells out to its waves by count, with room for a quarter more bits than the average share: tiles
    whose bits crowd into one stretch of the cells (noise in the first, a middle or the last rows, flat elsewhere; every other
    row; long codes in a corner) overrun a wave's window and are packed again by k_huffman_pack_rare -- same bytes as the
    oracle's either way."""
    n_rows, n_cols = shape
    rng = np.random.default_rng(n_rows + n_cols)
    tiles = []
    for lo, hi in ((0, n_rows // 5), (n_rows // 2, n_rows // 2 + n_rows // 6), (n_rows - n_rows // 7, n_rows), (0, n_rows)):
        for amp in (200, 30000):
            v = np.full((n_rows, n_cols), 1000, np.int64)
            v[lo:hi] += rng.integers(-amp, amp, (hi - lo, n_cols))
            tiles.append(v.ravel().astype(np.int32))
    v = np.full((n_rows, n_cols), 7, np.int64)
    v[::2] += rng.integers(-90, 90, (len(range(0, n_rows, 2)), n_cols))
    tiles.append(v.ravel().astype(np.int32))
    v = rng.integers(-2, 3, (n_rows, n_cols)).astype(np.int64)
    v[: n_rows // 8, : n_cols // 3] = rng.integers(-2_000_000_000, 2_000_000_000, (n_rows // 8, n_cols // 3))
    tiles.append(v.ravel().astype(np.int32))
    _check_tiles(codec, n_rows, n_cols, np.stack(tiles))


def test_one_tile_per_call_replays(codec):
    """BASELINE config 1 through the replayed-graph path (gf_huffman_{encode,decode}_i32 after their first call of a shape): new
    tiles through the same graph, every data kind, a tile with nulls, damaged packings (the reference's exception, then a good
    packing again), a second shape in between, and the canonical codec on the same context."""
    import gridfour_amd
    canon = gridfour_amd.CodecCanonHuffmanHip(context=codec.ctx)
    shapes = [(120, 150), (33, 65), (120, 150)]
    for n_rows, n_cols in shapes:
        tiles = [make_tile(k, n_rows, n_cols, seed=5) for k in KINDS]
        tiles.append(add_nulls(make_tile("smooth", n_rows, n_cols), n_rows, n_cols, 0.1))
        tiles.append(np.full(n_rows * n_cols, NULL, np.int32))
        for rep in range(2):
            for v in tiles:
                ref, _ = oracle.codec_huffman_encode(3, n_rows, n_cols, v)
                got = codec.encode(3, n_rows, n_cols, v)
                assert got == ref
                if ref is None:
                    continue
                assert np.array_equal(codec.decode(n_rows, n_cols, got), v)
                cref, _ = oracle.codec_canon_encode(4, n_rows, n_cols, v)
                cgot = canon.encode(4, n_rows, n_cols, v)
                assert cgot == cref
                assert np.array_equal(canon.decode(n_rows, n_cols, cgot), v)
        good, _ = oracle.codec_huffman_encode(3, n_rows, n_cols, tiles[1])
        for cut in (7, 9, 40, len(good) // 2):
            with pytest.raises(IOError):
                codec.decode(n_rows, n_cols, good[:cut])
        bad = bytearray(good)
        bad[1] = 77
        with pytest.raises(IOError):
            codec.decode(n_rows, n_cols, bytes(bad))
        assert np.array_equal(codec.decode(n_rows, n_cols, good), tiles[1])


@pytest.mark.parametrize("shape", [(2, 2), (3, 5), (9, 8), (16, 16), (40, 50), (64, 253), (17, 300), (120, 150), (200, 200), (256, 256),
                                   (1100, 8), (7, 129), (300, 400)], ids=lambda s: "%dx%d" % s)
def test_one_tile_per_call_shapes(codec, shape):
    """The one-tile call runs builds of its own (1024-thread workgroups: k_huffman_encode / k_huffman_pack of gvrs_encode.hip with
    sixteen waves, the 1024-thread decoder): tile shapes from 2 x 2 to 300 x 400, every data kind, through the replayed graph
    (second call of a shape on) against the oracle's bytes."""
    n_rows, n_cols = shape
    tiles = [make_tile(k, n_rows, n_cols, seed=11) for k in KINDS]
    tiles.append(add_nulls(make_tile("smooth", n_rows, n_cols, seed=3), n_rows, n_cols, 0.05))
    rng = np.random.default_rng(n_rows * 1000 + n_cols)
    tiles.append(rng.integers(-300, 301, n_rows * n_cols).astype(np.int32))           # two- and three-byte M32 values
    tiles.append((rng.integers(-2, 3, n_rows * n_cols).cumsum() % 251).astype(np.int32))
    for rep in range(2):
        for ci, v in enumerate(tiles):
            ref, _ = oracle.codec_huffman_encode(ci & 3, n_rows, n_cols, v)
            got = codec.encode(ci & 3, n_rows, n_cols, v)
            assert got == ref, (rep, ci)
            if ref is not None:
                assert np.array_equal(codec.decode(n_rows, n_cols, got), v), (rep, ci)


@pytest.mark.parametrize("shape", [(120, 150), (200, 200), (64, 256), (300, 40)], ids=lambda s: "%dx%d" % s)
def test_packer_on_tiles_with_their_bits_in_one_corner(codec, shape):
    """k_huffman_pack deals a range of cells to its waves in equal shares, each with a window of its own: a tile whose long codes
    sit in one part of it (rough ground in one quarter, the rest flat) overruns a share's window although the range would fit.
    Such a range is packed again in halves (pack_flat_ranges); what still does not fit goes to k_huffman_pack_rare.  The bytes
    are the oracle's either way."""
    n_rows, n_cols = shape
    cells = n_rows * n_cols
    rng = np.random.default_rng(n_rows * 31 + n_cols)
    tiles = []
    for part in range(4):                                   # the noisy quarter first, second, third, last
        v = np.full(cells, 1000, np.int64)
        a, b = part * cells // 4, (part + 1) * cells // 4
        v[a:b] += rng.integers(-20000, 20001, b - a)
        tiles.append(v.astype(np.int32))
    v = np.full(cells, -7, np.int64)                        # a noisy stripe of two rows in the middle, values of three and four bytes
    a = (n_rows // 2) * n_cols
    v[a:a + 2 * n_cols] += rng.integers(-3000000, 3000001, 2 * n_cols)
    tiles.append(v.astype(np.int32))
    v = rng.integers(-2, 3, cells).cumsum()                 # gentle everywhere but the last eighth
    v[-cells // 8:] += rng.integers(-500, 501, cells // 8)
    tiles.append(v.astype(np.int32))
    tiles = np.stack(tiles)
    packs, preds, status = codec.encode_batch(2, n_rows, n_cols, tiles)
    assert (np.asarray(status) == 0).all()
    for t, v in enumerate(tiles):
        ref, used = oracle.codec_huffman_encode(2, n_rows, n_cols, v)
        assert packs[t] == ref and preds[t] == used, t
    vals, st = codec.decode_batch(n_rows, n_cols, packs)
    assert (st == 0).all() and np.array_equal(vals, tiles)
