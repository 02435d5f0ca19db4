"""Round-trip / edge-case tests of the oracle, modelled on the reference's own unit tests
(PredictorModel*Test.java:52-78: 10x10 round trips; CodecM32Test) plus the codec-level cases
the reference never pins (uniform tiles, nulls, extremes)."""
import numpy as np
import pytest

import oracle


def _tile(rng, n_rows, n_cols, kind):
    n = n_rows * n_cols
    if kind == "ramp":
        return (np.arange(n, dtype=np.int64) - 1).astype(np.int32)
    if kind == "smooth":
        r = np.arange(n_rows)[:, None]
        c = np.arange(n_cols)[None, :]
        return (1000 * np.sin(r / 7.0) * np.cos(c / 5.0) + rng.integers(-3, 4, (n_rows, n_cols))).astype(np.int32).ravel()
    if kind == "noise16":
        return rng.integers(-32768, 32768, n).astype(np.int32)
    if kind == "noise32":
        return rng.integers(-2 ** 31 + 1, 2 ** 31, n, dtype=np.int64).astype(np.int32)
    if kind == "uniform":
        return np.full(n, 77, np.int32)
    if kind == "extremes":
        v = np.zeros(n, np.int32)
        v[::2] = 2 ** 31 - 1
        v[1::2] = -(2 ** 31) + 1
        return v
    raise ValueError(kind)


@pytest.mark.parametrize("model", [1, 2, 3])
@pytest.mark.parametrize("kind", ["ramp", "smooth", "noise16", "noise32", "extremes"])
@pytest.mark.parametrize("shape", [(10, 10), (2, 2), (7, 9), (1, 5), (5, 1)])
def test_predictor_roundtrip(model, kind, shape):
    n_rows, n_cols = shape
    if model == oracle.PM_LINEAR and n_cols < 2:
        pytest.skip("reference Linear predictor needs >= 2 columns (AIOOBE otherwise)")
    rng = np.random.default_rng(hash((model, kind, shape)) & 0xFFFF)
    v = _tile(rng, n_rows, n_cols, kind)
    m32, seed = oracle.predictor_encode(model, n_rows, n_cols, v)
    if model == oracle.PM_TRIANGLE and (n_rows < 2 or n_cols < 2):
        assert m32 is None
        return
    assert seed == v[0]
    out = oracle.predictor_decode(model, seed, n_rows, n_cols, m32)
    assert np.array_equal(out, v)


@pytest.mark.parametrize("frac", [0.02, 0.3, 0.9])
@pytest.mark.parametrize("shape", [(10, 10), (6, 17), (1, 9), (9, 1)])
def test_predictor_nulls_roundtrip(frac, shape):
    n_rows, n_cols = shape
    rng = np.random.default_rng(int(frac * 100) + n_rows)
    v = _tile(rng, n_rows, n_cols, "smooth")
    mask = rng.random(v.size) < frac
    mask[0] = True
    mask[-1] = False
    v[mask] = oracle.INT4_NULL
    m32, seed = oracle.predictor_encode(oracle.PM_DIFFERENCING_NULLS, n_rows, n_cols, v)
    assert len(m32) >= v.size            # one M32 symbol per cell
    out = oracle.predictor_decode(oracle.PM_DIFFERENCING_NULLS, seed, n_rows, n_cols, m32)
    assert np.array_equal(out, v)


@pytest.mark.parametrize("kind", ["ramp", "smooth", "noise16", "noise32", "uniform", "extremes"])
@pytest.mark.parametrize("shape", [(10, 10), (2, 2), (120, 150), (1, 2), (3, 1), (33, 65)])
def test_codec_huffman_roundtrip(kind, shape):
    n_rows, n_cols = shape
    rng = np.random.default_rng(len(kind) * 1000 + n_rows)
    v = _tile(rng, n_rows, n_cols, kind)
    if n_cols < 2:
        with pytest.raises(ValueError):
            oracle.codec_huffman_encode(0, n_rows, n_cols, v)
        return
    packing, used = oracle.codec_huffman_encode(3, n_rows, n_cols, v)
    assert packing[0] == 3 and packing[1] == used and used in (1, 2, 3)
    out = oracle.codec_huffman_decode(n_rows, n_cols, packing)
    assert np.array_equal(out, v)
    # the winner is the strictly shortest in the order D, L, T
    sizes = {}
    for m in (1, 2, 3):
        p, _ = oracle.codec_huffman_encode(3, n_rows, n_cols, v, predictor_mask=1 << (m - 1))
        if p is not None:
            sizes[m] = len(p)
    best = min(sizes.values())
    assert used == min(m for m, s in sizes.items() if s == best)
    assert len(packing) == best


def test_codec_huffman_nulls_and_declines():
    rng = np.random.default_rng(5)
    v = _tile(rng, 20, 30, "smooth")
    v[rng.random(v.size) < 0.2] = oracle.INT4_NULL
    packing, used = oracle.codec_huffman_encode(0, 20, 30, v)
    assert used == oracle.PM_DIFFERENCING_NULLS and packing[1] == 4
    assert np.array_equal(oracle.codec_huffman_decode(20, 30, packing), v)
    # all nulls -> the encoder returns null (CodecHuffman.java:80-82)
    assert oracle.codec_huffman_encode(0, 4, 4, np.full(16, oracle.INT4_NULL, np.int32)) == (None, 0)
    # 1x1 tile: Differencing yields no residuals, then PredictorModelLinear indexes values[1]
    # (PredictorModelLinear.java:113) -> ArrayIndexOutOfBoundsException in the reference
    with pytest.raises(ValueError):
        oracle.codec_huffman_encode(0, 1, 1, np.array([5], np.int32))


def test_codec_huffman_uniform_special_case():
    # one distinct M32 symbol -> 9-bit tree, no text (HuffmanEncoder.java:147-157)
    v = np.full(100, 9, np.int32)
    packing, used = oracle.codec_huffman_encode(0, 10, 10, v)
    assert used == 1
    assert len(packing) == 10 + 3     # 80 + 17 bits -> 13 bytes
    # bits: 00000000 | 1 | 00000000 (symbol 0x00 = residual 0), LSB-first
    assert packing[10:] == bytes([0x00, 0x01, 0x00])
    assert np.array_equal(oracle.codec_huffman_decode(10, 10, packing), v)


def test_decode_errors():
    v = np.arange(100, dtype=np.int32)
    packing, _ = oracle.codec_huffman_encode(0, 10, 10, v * 3 % 17)
    bad = bytearray(packing)
    bad[1] = 9                          # unknown predictor -> IOException
    with pytest.raises(IOError):
        oracle.codec_huffman_decode(10, 10, bytes(bad))
    with pytest.raises(IOError):        # truncated text -> read past end
        oracle.codec_huffman_decode(10, 10, packing[:len(packing) // 2])


def test_dem_generator_is_dem_like():
    t = oracle.dem_tiles(oracle.DEM_SEED + 2, 120, 150, 144, 0, 4)
    assert t.min() >= -11000 and t.max() <= 8848
    d = np.diff(t.reshape(4, 120, 150), axis=2)
    assert np.abs(d).max() < 127 * 3
    assert (np.abs(d) <= 126).mean() > 0.97
    # deterministic
    t2 = oracle.dem_tiles(oracle.DEM_SEED + 2, 120, 150, 144, 1, 2)
    assert np.array_equal(t[1:3], t2)
    # adjacent tiles are continuous pieces of one grid
    a = t[0].reshape(120, 150)[:, -1].astype(np.int64)
    b = t[1].reshape(120, 150)[:, 0].astype(np.int64)
    assert np.abs(a - b).max() < 127
