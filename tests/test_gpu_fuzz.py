"""Decoder robustness on damaged packings: every decoder of the C ABI must finish (no hang, no fault), report a status
from the documented set, and -- where the oracle also accepts the damaged packing -- produce the oracle's values.
The reference's behaviour on arbitrary garbage is an exception of some kind; which one is not part of the contract."""
import numpy as np
import pytest

import oracle
from tilegen import add_nulls, make_tile

pytestmark = pytest.mark.gpu

ALLOWED = {0, -1, -2, -7}          # OK, FORMAT, BOUNDS, UNSUPPORTED


def _damage(rng, packing, n_variants):
    out = []
    p = bytearray(packing)
    for i in range(n_variants):
        q = bytearray(p)
        kind = i % 5
        if kind == 0:                                   # single bit flip anywhere
            j = int(rng.integers(0, len(q) * 8))
            q[j >> 3] ^= 1 << (j & 7)
        elif kind == 1:                                 # bit flips in the code-table region
            for _ in range(3):
                j = int(rng.integers(16, min(len(q), 200) * 8))
                q[j >> 3] ^= 1 << (j & 7)
        elif kind == 2:                                 # truncation
            q = q[:int(rng.integers(1, len(q)))]
        elif kind == 3:                                 # random bytes in the middle
            a = int(rng.integers(10, max(11, len(q) - 8)))
            q[a:a + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
        else:                                           # header damage
            j = int(rng.integers(0, min(10, len(q))))
            q[j] = int(rng.integers(0, 256))
        out.append(bytes(q))
    return out


def _tiles(n_rows, n_cols):
    return [make_tile("smooth", n_rows, n_cols), make_tile("noise16", n_rows, n_cols),
            add_nulls(make_tile("smooth", n_rows, n_cols), n_rows, n_cols, 0.1), make_tile("sparse_big", n_rows, n_cols)]


@pytest.mark.parametrize("family", ["huffman", "canon", "lsop"])
def test_damaged_packings(family):
    import gridfour_amd
    n_rows, n_cols = 40, 60
    rng = np.random.default_rng({"huffman": 1, "canon": 2, "lsop": 3}[family])
    if family == "huffman":
        codec, enc, dec = gridfour_amd.CodecHuffmanHip(), oracle.codec_huffman_encode, oracle.codec_huffman_decode
    elif family == "canon":
        codec, enc, dec = gridfour_amd.CodecCanonHuffmanHip(), oracle.codec_canon_encode, oracle.codec_canon_decode
    else:
        codec = gridfour_amd.LsCodecHip(deflate_enabled=False)
        enc = lambda ci, r, c, v: oracle.lsop12_encode(ci, r, c, v, False)
        dec = oracle.lsop12_decode
    damaged = []
    for v in _tiles(n_rows, n_cols):
        ref = enc(1, n_rows, n_cols, v)[0]
        if ref is None:
            continue
        damaged += _damage(rng, ref, 60)
    vals, status = codec.decode_batch(n_rows, n_cols, damaged)
    n_ok = 0
    for i, pk in enumerate(damaged):
        assert int(status[i]) in ALLOWED, (i, int(status[i]))
        try:
            want = dec(n_rows, n_cols, pk)
        except (IOError, ValueError):
            continue
        if status[i] == 0:
            n_ok += 1
            assert np.array_equal(vals[i], want), i
    assert n_ok > 0          # some damage is harmless (padding bits, unused table entries): those must still agree
