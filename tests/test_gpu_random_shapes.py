"""Randomised shape sweep: every integer codec of the library against the oracle on tiles of odd shapes (thin, tiny, wide,
not multiples of anything), a few data kinds each.  Complements the fixed shapes of the per-codec parity tests."""
import numpy as np
import pytest

import oracle
from tilegen import NULL, add_nulls, make_tile

pytestmark = pytest.mark.gpu

_rng = np.random.default_rng(20260101)
SHAPES = sorted({(int(_rng.integers(1, 70)), int(_rng.integers(1, 90))) for _ in range(28)} |
                {(1, 1), (1, 2), (2, 1), (2, 3), (3, 2), (6, 6), (6, 7), (7, 6), (64, 64), (65, 63), (1, 513), (257, 2), (300, 301), (400, 500)})


def _tiles(nr, nc):
    t = [make_tile("smooth", nr, nc, seed=1), make_tile("noise16", nr, nc, seed=2), make_tile("sparse_big", nr, nc, seed=3),
         make_tile("steps", nr, nc)]
    if nr * nc > 1:
        t.append(add_nulls(make_tile("smooth", nr, nc), nr, nc, 0.2))
    return np.stack(t)


def _expect(fn, *args):
    """(packing | None, error status) of an oracle encoder: -2 where Java throws AIOOBE, -4 IllegalArgumentException"""
    try:
        return fn(*args)[0], 0
    except ValueError as ex:
        return None, (-2 if "rc=-2" in str(ex) else -4)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d" % s)
def test_all_codecs(shape):
    import gridfour_amd
    nr, nc = shape
    tiles = _tiles(nr, nc)
    ctx = gridfour_amd.GvrsHipContext(0)
    families = [("huffman", gridfour_amd.CodecHuffmanHip(context=ctx), oracle.codec_huffman_encode, oracle.codec_huffman_decode),
                ("canon", gridfour_amd.CodecCanonHuffmanHip(context=ctx), oracle.codec_canon_encode, oracle.codec_canon_decode),
                ("deflate", gridfour_amd.CodecDeflateHip(context=ctx), oracle.codec_deflate_encode, oracle.codec_deflate_decode)]
    for name, codec, enc, dec in families:
        packs, _, status = codec.encode_batch(5, nr, nc, tiles)
        good, idx = [], []
        for t, v in enumerate(tiles):
            ref, err = _expect(enc, 5, nr, nc, v)
            if err:
                assert packs[t] is None and status[t] == err, (name, t, status[t], err)
            elif ref is None:
                assert packs[t] is None and status[t] == 1, (name, t, status[t])
            else:
                assert status[t] == 0 and packs[t] == ref, (name, t, status[t])
                good.append(ref)
                idx.append(t)
        if good:
            vals, st = codec.decode_batch(nr, nc, good)
            for k, t in enumerate(idx):
                assert st[k] == 0 and np.array_equal(vals[k], dec(nr, nc, good[k])), (name, t, st[k])
    lsop = gridfour_amd.LsCodecHip(context=ctx, deflate_enabled=True)
    packs, types, status = lsop.encode_batch(5, nr, nc, tiles)
    good, idx = [], []
    for t, v in enumerate(tiles):
        ref, typ = oracle.lsop12_encode(5, nr, nc, v, True)
        if ref is None:
            assert packs[t] is None and status[t] == 1, ("lsop", t, status[t])
        else:
            assert status[t] == 0 and types[t] == typ and packs[t] == ref, ("lsop", t, status[t], types[t], typ)
            good.append(ref)
            idx.append(t)
    if good:
        vals, st = lsop.decode_batch(nr, nc, good)
        for k, t in enumerate(idx):
            assert st[k] == 0 and np.array_equal(vals[k], tiles[t]), ("lsop", t, st[k])
