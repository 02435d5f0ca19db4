"""The encoder's byte plane of raw row differences (round 6): k_huffman_encode<true, 1, true> leaves every cell's Differencing
residual as one byte, k_huffman_pack derives the winner's residuals from it when every byte is its value and the winner's stream
is plain.  Tiles on either side of every condition -- differences of exactly 126 / 127 / 128 and their negatives at the places the
predictors treat apart (first row, first and second column, last column), each predictor as the winner, shapes whose stream heads
need more than one turn, tiles of one batch on different paths -- against the CPU oracle, byte for byte."""
import numpy as np
import pytest

import oracle
from tilegen import make_tile

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["huffman", "canon"])
def codec(request):
    """CodecHuffman and CodecCanonHuffman (k_huffman_pack / k_canon_pack: both pack a narrow stream from the plane)"""
    import gridfour_amd
    c = gridfour_amd.CodecHuffmanHip() if request.param == "huffman" else gridfour_amd.CodecCanonHuffmanHip()
    c.oracle_encode = oracle.codec_huffman_encode if request.param == "huffman" else oracle.codec_canon_encode
    return c


def _check(codec, nr, nc, tiles, want_models=None):
    tiles = np.ascontiguousarray(np.stack(tiles), np.int32)
    packs, preds, status = codec.encode_batch(1, nr, nc, tiles)
    for t, v in enumerate(tiles):
        ref, used = codec.oracle_encode(1, nr, nc, v)
        assert status[t] == 0 and preds[t] == used, (t, status[t], preds[t], used)
        if packs[t] != ref:
            first = next(i for i in range(min(len(ref), len(packs[t]))) if packs[t][i] != ref[i])
            raise AssertionError("tile %d (%dx%d, model %d) differs at byte %d of %d / %d" % (t, nr, nc, used, first, len(packs[t]), len(ref)))
    if want_models:
        assert set(want_models) <= set(int(p) for p in preds), (want_models, sorted(set(int(p) for p in preds)))
    vals, st = codec.decode_batch(nr, nc, packs)
    assert (st == 0).all() and np.array_equal(vals, tiles)


def _smooth(nr, nc, seed=0):
    rng = np.random.default_rng(seed * 31 + nr * 7 + nc)
    y, x = np.mgrid[0:nr, 0:nc]
    return (400 * np.sin(x / 11.0 + seed) * np.cos(y / 9.0) + rng.integers(-2, 3, (nr, nc))).astype(np.int32)


def _rows_of_walks(nr, nc, seed=0):
    """rows that are independent random walks: the row difference is small, nothing ties a row to the one above (Differencing wins)"""
    rng = np.random.default_rng(seed + 11)
    steps = rng.integers(-3, 4, (nr, nc))
    steps[:, 0] = rng.integers(-100, 101, nr)
    return np.cumsum(steps, axis=1).astype(np.int32) + np.cumsum(steps[:, :1], axis=0).astype(np.int32)


def _ramps(nr, nc, seed=0):
    """every row a straight line of its own slope (Linear wins)"""
    rng = np.random.default_rng(seed + 5)
    slope = rng.integers(-60, 61, nr)[:, None]
    start = rng.integers(-500, 500, nr)[:, None]
    return (start + slope * np.arange(nc)[None, :] + rng.integers(0, 2, (nr, nc))).astype(np.int32)


SHAPES = [(120, 150), (2, 8), (3, 9), (40, 15), (33, 16), (17, 17), (7, 149), (9, 151), (5, 256), (4, 300), (200, 200)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d" % s)
def test_each_predictor_wins_from_the_plane(codec, shape):
    nr, nc = shape
    tiles = []
    for s in range(3):
        tiles += [_smooth(nr, nc, s).ravel(), _rows_of_walks(nr, nc, s).ravel(), _ramps(nr, nc, s).ravel()]
    _check(codec, nr, nc, tiles)


@pytest.mark.parametrize("shape", [(520, 8), (600, 9), (300, 10), (257, 12), (1030, 8)], ids=lambda s: "%dx%d" % s)
def test_stream_heads_of_more_than_one_turn(codec, shape):
    """Triangle's head has nC + nR - 2 elements, Linear's 2 nR - 1: beyond 512 the head takes wave 0 more than one turn"""
    nr, nc = shape
    _check(codec, nr, nc, [_smooth(nr, nc, 1).ravel(), _ramps(nr, nc, 2).ravel(), _rows_of_walks(nr, nc, 3).ravel()])


@pytest.mark.parametrize("delta", [126, 127, 128, 129, 254, 255, -126, -127, -128, -129, -255])
def test_row_differences_at_the_edges_of_a_byte(codec, delta):
    """one cell lifted so that a row difference (and its neighbours' residuals) sits at the edge of the plain range, of the byte's
    range, beyond it: the plane holds the tile or not, the winner's stream is plain or not -- the packing is the oracle's either way"""
    nr, nc = 24, 40
    tiles = []
    for (r, c) in ((0, 1), (0, 39), (1, 0), (5, 0), (5, 1), (5, 2), (5, 20), (5, 39), (23, 39), (12, 7)):
        for base in (_smooth(nr, nc, 2), _ramps(nr, nc, 1), _rows_of_walks(nr, nc, 4)):
            v = base.copy()
            left = v[r, c - 1] if c > 0 else v[r - 1, 0]
            v[r, c] = left + delta
            tiles.append(v.ravel())
    _check(codec, nr, nc, tiles)


def test_one_batch_many_paths(codec):
    """tiles of one batch on every path of the packer: from the plane (each model), from the tile (wide values, nulls), declined"""
    nr, nc = 50, 64
    tiles = [_smooth(nr, nc, 0).ravel(), make_tile("noise16", nr, nc), _ramps(nr, nc, 0).ravel(), make_tile("sparse_big", nr, nc),
             _rows_of_walks(nr, nc, 0).ravel(), make_tile("uniform", nr, nc), make_tile("noise8", nr, nc), make_tile("steps", nr, nc)]
    v = _smooth(nr, nc, 5).ravel().copy()
    v[100:140] = -(2 ** 31)
    tiles.append(v)
    _check(codec, nr, nc, tiles * 3, want_models=[1, 2, 3, 4])
