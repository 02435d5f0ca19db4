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
