"""The decoder's byte path (gvrs_decode.hip: m32_bytes_to_tile) against the oracle.

A CodecHuffman packing whose tree holds neither an M32 introducer (0x7f / 0x81) nor the null code (0x80) is a text of one-byte
values: byte j of the Huffman output is stream element j (CodecM32.java:327-356), and the fast kernel turns the bytes into the
tile row by row without start marks.  These tests force each predictor (PredictorModelDifferencing / Linear / Triangle) onto
such data over the tile shapes that decide how the rows are dealt out -- every residue of nCols mod 4, one to 256 columns
and beyond (the wider ones take the general stage), few and many rows, the three workgroup sizes -- and put tiles that do NOT
qualify (a wide residual, a null code) into the same batch.
"""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

SHAPES = [(2, 4), (2, 5), (3, 6), (5, 7), (9, 8), (16, 16), (33, 65), (40, 50), (120, 150), (64, 253), (64, 254), (64, 255),
          (64, 256), (64, 257), (17, 300), (200, 200), (256, 256), (300, 40), (1100, 8), (7, 129)]


@pytest.fixture(scope="module")
def codec():
    import gridfour_amd
    return gridfour_amd.CodecHuffmanHip()


def gentle_tiles(n_rows, n_cols, seed):
    """Tiles whose residuals stay inside one M32 byte for all three predictors (|second differences| <= 80), plus the edge
    residuals +126 / -126, a flat tile (single-symbol tree) and a plane."""
    rng = np.random.default_rng(seed * 7919 + n_rows * 131 + n_cols)
    r = np.arange(n_rows, dtype=np.int64)[:, None]
    c = np.arange(n_cols, dtype=np.int64)[None, :]
    tiles = []
    tiles.append(3 * r + 2 * c + rng.integers(-10, 11, (n_rows, n_cols)))
    tiles.append(-7 * r + 5 * c - 4000 + rng.integers(-20, 21, (n_rows, n_cols)))
    tiles.append(np.full((n_rows, n_cols), -123456, np.int64))
    tiles.append(100 * r - 60 * c + 2000000000)                      # a plane near the top of the int32 range
    t = rng.integers(-1, 2, (n_rows, n_cols)).cumsum(axis=1) + rng.integers(-1, 2, (n_rows, 1)).cumsum(axis=0)
    tiles.append(t)
    # +126 / -126 between neighbours of a row (Differencing's extremes): columns alternate
    e = np.zeros((n_rows, n_cols), np.int64)
    e[:, 1::2] = 126
    tiles.append(e + 2 * r)
    # second differences of +126 / -126 along the rows (Linear's extremes): first differences alternate 0, 126
    d = np.zeros((n_rows, n_cols), np.int64)
    d[:, 1::2] = 126
    tiles.append(d.cumsum(axis=1) - 3 * r)
    # Triangle residuals of +126 / -126: the 2-D prefix sum of a checkerboard
    g = 126 * (1 - 2 * ((r + c) & 1))
    g[0, :] = 1
    g[:, 0] = -2
    tiles.append(g.cumsum(axis=1).cumsum(axis=0))
    return np.stack([x.astype(np.int32).ravel() for x in tiles])


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d" % s)
def test_each_predictor_on_one_byte_residuals(codec, shape):
    n_rows, n_cols = shape
    tiles = gentle_tiles(n_rows, n_cols, 1)
    for model in (1, 2, 3):
        packs, keep = [], []
        for t, v in enumerate(tiles):
            ref, used = oracle.codec_huffman_encode(5, n_rows, n_cols, v, predictor_mask=1 << (model - 1))
            if ref is None:
                continue
            assert used == model
            packs.append(ref)
            keep.append(t)
        assert packs
        vals, st = codec.decode_batch(n_rows, n_cols, packs)
        assert (st == 0).all(), (model, st)
        for k, t in enumerate(keep):
            if not np.array_equal(vals[k], tiles[t]):
                bad = np.nonzero(vals[k] != tiles[t])[0]
                raise AssertionError("model %d tile %d: %d cells differ, first at %d (row %d col %d): got %d want %d" % (
                    model, t, bad.size, bad[0], bad[0] // n_cols, bad[0] % n_cols, vals[k][bad[0]], tiles[t][bad[0]]))


@pytest.mark.parametrize("shape", [(120, 150), (33, 65), (200, 200), (64, 254)], ids=lambda s: "%dx%d" % s)
def test_tiles_that_do_not_qualify_share_the_batch(codec, shape):
    """One wide residual (an introducer among the leaves) or one null-code residual in a tile sends THAT tile through the
    general value stage; its neighbours in the batch keep the byte path."""
    n_rows, n_cols = shape
    base = gentle_tiles(n_rows, n_cols, 2)
    tiles = [base[0], base[1].copy(), base[4].copy(), base[0].copy(), base[4]]
    tiles[1][n_rows * n_cols // 2] += 5000                           # a step: 2- and 3-byte M32 values around it
    tiles[2][n_cols + 3] += 127                                      # exactly the first two-byte value (0x7f 0x00) for Differencing
    tiles[3][7] = -(2 ** 31) + tiles[3][6]                           # a residual of Integer.MIN_VALUE (byte 0x80) for Differencing
    tiles = np.stack(tiles)
    for mask in (1, 2, 4, 7):
        packs = []
        for v in tiles:
            ref, _ = oracle.codec_huffman_encode(0, n_rows, n_cols, v, predictor_mask=mask)
            packs.append(ref)
        vals, st = codec.decode_batch(n_rows, n_cols, packs)
        assert (st == 0).all(), (mask, st)
        assert np.array_equal(vals, tiles), mask


def test_encode_then_decode_of_gentle_batches(codec):
    """The whole round trip through the library's own encoder (all three predictors compete)."""
    for n_rows, n_cols in ((120, 150), (50, 50), (200, 200)):
        tiles = gentle_tiles(n_rows, n_cols, 3)
        packs, preds, status = codec.encode_batch(1, n_rows, n_cols, tiles)
        assert (np.asarray(status) == 0).all()
        for t, v in enumerate(tiles):
            ref, used = oracle.codec_huffman_encode(1, n_rows, n_cols, v)
            assert packs[t] == ref and preds[t] == used, t
        vals, st = codec.decode_batch(n_rows, n_cols, packs)
        assert (st == 0).all() and np.array_equal(vals, tiles)


@pytest.mark.parametrize("shape", [(120, 150), (200, 200), (40, 50)], ids=lambda s: "%dx%d" % s)
def test_streams_between_one_and_two_bytes_per_cell(codec, shape):
    """Tiles whose M32 stream outgrows the fast kernel's usual LDS buffer (1.125 bytes per cell) but not its second run's
    (two bytes per cell): half of the residuals need two or three M32 bytes.  They share the batch with tiles that fit and with
    tiles that need the general kernel's workspace (three bytes per value)."""
    import struct
    n_rows, n_cols = shape
    rng = np.random.default_rng(n_rows * 7 + n_cols)
    tiles = [rng.integers(-200, 201, n_rows * n_cols).astype(np.int32),          # ~1.5 bytes per value
             rng.integers(-90, 91, n_rows * n_cols).astype(np.int32),            # ~1.2
             gentle_tiles(n_rows, n_cols, 4)[0],                                 # 1.0: the byte path
             rng.integers(-20000, 20001, n_rows * n_cols).astype(np.int32),      # 3 and more: the general kernel
             (rng.integers(-200, 201, n_rows * n_cols).cumsum() % 5000).astype(np.int32)]
    packs = []
    for v in tiles:
        ref, _ = oracle.codec_huffman_encode(0, n_rows, n_cols, v)
        packs.append(ref)
    ratios = [struct.unpack("<I", p[6:10])[0] / (n_rows * n_cols) for p in packs]
    assert 1.125 < ratios[0] < 2.0 and ratios[3] > 2.0, ratios
    vals, st = codec.decode_batch(n_rows, n_cols, packs)
    assert (st == 0).all(), st
    assert np.array_equal(vals, np.stack(tiles))
    got, preds, status = codec.encode_batch(0, n_rows, n_cols, np.stack(tiles))
    assert list(got) == packs


@pytest.mark.parametrize("shape", [(120, 150), (200, 200), (256, 256), (400, 400), (90, 1000)], ids=lambda s: "%dx%d" % s)
def test_subsequences_that_fill_their_share_of_the_symbol_pool(codec, shape):
    """One-bit and two-bit codes: a 128-bit subsequence of the Huffman text holds exactly 128 symbols, which is exactly the share
    of the symbol pool a subsequence has (gvrs_decode.hip: fast_sync_pass); at 160 bits and more it overruns the share and the
    tile takes the two-pass form."""
    n_rows, n_cols = shape
    rng = np.random.default_rng(n_rows * 11 + n_cols)
    tiles = []
    for n_sym in (2, 3, 4):
        steps = rng.integers(0, n_sym, (n_rows, n_cols))
        steps[:, 0] = 0
        tiles.append(steps.cumsum(axis=1).astype(np.int32).ravel())
    packs = [oracle.codec_huffman_encode(0, n_rows, n_cols, v)[0] for v in tiles]
    vals, st = codec.decode_batch(n_rows, n_cols, packs)
    assert (st == 0).all(), st
    for t, v in enumerate(tiles):
        bad = np.nonzero(vals[t] != v)[0]
        assert bad.size == 0, (t, bad.size, int(bad[0]))


def test_saved_soak_case_full_share_in_the_second_run(codec):
    """tools/soak.py, seed 40041003 case 1023 (228 x 276, six tiles): tile 2 (DifferencingWithNulls, 1.5 M32 bytes per cell) is
    decoded by the fast kernel's second run, and one subsequence of its text holds exactly as many symbols as its share of the
    pool while its neighbours in the wave are still decoding (the first form of the pool let the idle steps of such a cursor
    clear the last word of its share)."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "soak", "soak_40041003_case1023.npz"))
    tiles, n_rows, n_cols = d["tiles"], int(d["shape"][0]), int(d["shape"][1])
    packs = [oracle.codec_huffman_encode(0, n_rows, n_cols, v)[0] for v in tiles]
    got, _, status = codec.encode_batch(0, n_rows, n_cols, tiles)
    assert (np.asarray(status) == 0).all() and list(got) == packs
    vals, st = codec.decode_batch(n_rows, n_cols, packs)
    assert (st == 0).all(), st
    assert np.array_equal(vals, tiles)
    for t, p in enumerate(packs):
        assert np.array_equal(codec.decode(n_rows, n_cols, p), tiles[t]), t
