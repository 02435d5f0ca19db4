"""CodecCanonHuffman packings WITHOUT escapes go through the fast legacy kernel first (DEC_FAST_CANON in gvrs_decode.hip, round 5):
the text of such a packing is a prefix-coded string of the bytes value + 128 with an end-of-text symbol behind them
(CanonicalHuffman.java:441-519), so it is decoded once into the symbol pool and finished by the byte path; what that run cannot
take it leaves to k_canon_decode.  These tests sit on the seam: tiles of either kind in one batch, the alphabets the run must turn
down (all 256 byte values in use: no byte is left for the end-of-text symbol), texts that end early or late, and every single-bit
damage of a plain packing -- always against the oracle's CodecCanonHuffman.decode."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    import gridfour_amd
    return gridfour_amd.CodecCanonHuffmanHip()


def _gentle(n_rows, n_cols, seed, amp=3):
    """terrain whose residuals stay inside a byte for every predictor"""
    rng = np.random.default_rng(seed)
    r = np.arange(n_rows)[:, None]
    c = np.arange(n_cols)[None, :]
    base = 40.0 * np.sin(r / 17.0) * np.cos(c / 13.0) + 2500
    return (base + rng.integers(-amp, amp + 1, (n_rows, n_cols))).astype(np.int32).ravel()


def _is_plain(n_rows, n_cols, packing):
    """no escape in the tile's text: every residual of the predictor the packing names fits a byte"""
    v = oracle.codec_canon_decode(n_rows, n_cols, packing)
    res, _ = oracle.predictor_encode_int(packing[1], n_rows, n_cols, v)
    return bool(np.all((res >= -128) & (res <= 127)))


def _decode_and_compare(codec, n_rows, n_cols, packs):
    vals, st = codec.decode_batch(n_rows, n_cols, packs)
    for k, pk in enumerate(packs):
        try:
            want = oracle.codec_canon_decode(n_rows, n_cols, pk)
        except Exception:
            want = None
        if want is None:
            assert st[k] < 0, (k, int(st[k]), "the oracle fails")
        else:
            assert st[k] == 0, (k, int(st[k]), "the oracle decodes")
            if not np.array_equal(vals[k], want):
                bad = np.nonzero(vals[k] != want)[0]
                raise AssertionError("packing %d (model %d) differs at %d cells, first %d: got %d want %d" % (
                    k, pk[1], bad.size, bad[0], vals[k][bad[0]], want[bad[0]]))


@pytest.mark.parametrize("shape", [(120, 150), (90, 120), (200, 200), (64, 64), (33, 65), (2, 4), (7, 9), (50, 256), (16, 255)],
                         ids=lambda s: "%dx%d" % s)
@pytest.mark.parametrize("model", [1, 2, 3])
def test_plain_tiles_every_predictor(codec, shape, model):
    n_rows, n_cols = shape
    packs = []
    for seed in range(6):
        v = _gentle(n_rows, n_cols, 100 * model + seed, amp=1 + seed)
        pk, used = oracle.codec_canon_encode(0, n_rows, n_cols, v, predictor_mask=1 << (model - 1))
        assert pk is not None and used == model
        packs.append(pk)
    assert any(_is_plain(n_rows, n_cols, pk) for pk in packs)
    _decode_and_compare(codec, n_rows, n_cols, packs)


def test_plain_and_escape_tiles_in_one_batch(codec):
    n_rows, n_cols = 120, 150
    rng = np.random.default_rng(9)
    packs, kinds = [], []
    for t in range(96):
        v = _gentle(n_rows, n_cols, 500 + t, amp=2 + t % 5)
        if t % 3 == 1:                                       # a few steps that need escapes
            idx = rng.integers(0, v.size, 1 + t % 7)
            v[idx] += rng.integers(-4000, 4000, idx.size).astype(np.int32)
        if t % 11 == 5:
            v[:] = 77                                        # uniform: the six-byte packing
        pk, _ = oracle.codec_canon_encode(1, n_rows, n_cols, v)
        packs.append(pk)
        kinds.append(len(pk) > 6 and _is_plain(n_rows, n_cols, pk))
    assert 20 < sum(kinds) < 90
    _decode_and_compare(codec, n_rows, n_cols, packs)


def test_all_256_byte_values_in_use(codec):
    """257 leaves with the end-of-text symbol: no byte value is free for it -- the tile belongs to k_canon_decode"""
    n_rows, n_cols = 40, 50
    res = np.resize(np.arange(-128, 128, dtype=np.int32), n_rows * n_cols - 1)
    np.random.default_rng(2).shuffle(res)
    v = oracle.predictor_decode_int(1, 5000, n_rows, n_cols, res)
    pk, used = oracle.codec_canon_encode(0, n_rows, n_cols, v, predictor_mask=1)
    assert used == 1
    _, _, cl = oracle.canon_encode(res)
    assert np.count_nonzero(cl[:256]) == 256 and cl[256] == cl[257] == cl[258] == 0
    # ... and with one value missing: 256 leaves, the free byte is the end-of-text symbol's
    res2 = res.copy()
    res2[res2 == 77] = 76
    v2 = oracle.predictor_decode_int(1, 5000, n_rows, n_cols, res2)
    pk2, _ = oracle.codec_canon_encode(0, n_rows, n_cols, v2, predictor_mask=1)
    res3 = res.copy()
    res3[res3 == -128] = -127                                 # (the byte 0x80 ^ 0x80 = 0: the first one looked at)
    v3 = oracle.predictor_decode_int(1, 5000, n_rows, n_cols, res3)
    pk3, _ = oracle.codec_canon_encode(0, n_rows, n_cols, v3, predictor_mask=1)
    _decode_and_compare(codec, n_rows, n_cols, [pk, pk2, pk3, pk, pk3])


def test_two_symbol_and_one_symbol_alphabets(codec):
    n_rows, n_cols = 30, 40
    packs = []
    for res_of in (lambda n: np.zeros(n, np.int32), lambda n: (np.arange(n) % 2).astype(np.int32),
                   lambda n: np.where(np.arange(n) % 97 == 0, 127, -128).astype(np.int32)):
        for model in (1, 2, 3):
            n = {1: n_rows * n_cols - 1, 2: n_rows * n_cols - 1, 3: n_rows * n_cols - 1}[model]
            v = oracle.predictor_decode_int(model, -3, n_rows, n_cols, res_of(n))
            pk, used = oracle.codec_canon_encode(0, n_rows, n_cols, v, predictor_mask=1 << (model - 1))
            if pk is not None:
                packs.append(pk)
    assert packs
    _decode_and_compare(codec, n_rows, n_cols, packs)


@pytest.mark.parametrize("model", [1, 2, 3])
def test_text_that_ends_early_or_late(codec, model):
    """CanonicalHuffman.decode stops at the end-of-text symbol wherever it stands (the cells behind it stay zero residuals) and
    overruns its array on a text that is too long: packings built by hand from the oracle's stream encoder"""
    n_rows, n_cols = 24, 36
    v = _gentle(n_rows, n_cols, 70 + model)
    res, seed = oracle.predictor_encode_int(model, n_rows, n_cols, v)
    head = bytes([0, model]) + int(seed).to_bytes(4, "little", signed=True)
    packs = []
    for cut in (0, -1, -2, -37, -(res.size // 2), 5 - res.size):
        text = res if cut == 0 else res[:cut]
        stream, _, _ = oracle.canon_encode(text, bit_pos=48, prefix=head)
        packs.append(stream)
    for extra in (1, 2, 9):
        stream, _, _ = oracle.canon_encode(np.concatenate([res, np.arange(extra, dtype=np.int32)]), bit_pos=48, prefix=head)
        packs.append(stream)
    good, _ = oracle.codec_canon_encode(0, n_rows, n_cols, v, predictor_mask=1 << (model - 1))
    assert packs[0] == good                                   # (the hand-built form is the encoder's)
    _decode_and_compare(codec, n_rows, n_cols, packs)


@pytest.mark.parametrize("model,seed", [(1, 1), (2, 2), (3, 3), (3, 4)])
def test_every_single_bit_flip_of_a_plain_packing(codec, model, seed):
    n_rows, n_cols = 9, 22
    v = _gentle(n_rows, n_cols, seed, amp=4)
    good, used = oracle.codec_canon_encode(0, n_rows, n_cols, v, predictor_mask=1 << (model - 1))
    assert used == model and _is_plain(n_rows, n_cols, good)
    packs = [good]
    for i in range(1, len(good)):
        for b in range(8):
            x = bytearray(good)
            x[i] ^= 1 << b
            packs.append(bytes(x))
    for n in range(6, len(good)):                             # ... and every truncation
        packs.append(good[:n])
    _decode_and_compare(codec, n_rows, n_cols, packs)


def test_device_batch_plain_terrain(codec):
    import gridfour_amd
    ctx = gridfour_amd.GvrsHipContext(0)
    n_rows, n_cols, nt = 120, 150, 2000
    b = gridfour_amd.DeviceTileBatch(ctx, n_rows, n_cols, nt, codec="canon")
    for style in (0, 1):                                      # the smooth surface (every tile plain) and the rough one (a mix)
        b.synth_dem(0x9E3779B97F4A7C15 + 2, 144, style=style)
        b.encode(codec_index=1)
        b.decode()
        ctx.synchronize()
        assert np.all(b.get_enc_status() == 0) and np.all(b.get_dec_status() == 0)
        assert np.array_equal(b.get_decoded(), b.get_values())
    b.free()


@pytest.mark.parametrize("model,seed", [(1, 11), (3, 12)])
def test_random_damage_of_the_code_tables_and_the_text(codec, model, seed):
    """heavier damage than single bits: runs of random bytes over the serialised code tables (the lengths the fast run turns into leaf
    records, or leaves to k_canon_decode when they make no complete code) and over the text"""
    n_rows, n_cols = 24, 36
    v = _gentle(n_rows, n_cols, 300 + seed, amp=5)
    good, used = oracle.codec_canon_encode(0, n_rows, n_cols, v, predictor_mask=1 << (model - 1))
    assert used == model and _is_plain(n_rows, n_cols, good)
    rng = np.random.default_rng(seed)
    packs = [good]
    for k in range(600):
        x = bytearray(good)
        if k % 3 == 0:                                        # the head of the stream: the meta lengths and the coded lengths
            a = int(rng.integers(6, 40))
        else:
            a = int(rng.integers(6, len(good) - 1))
        n = int(rng.integers(1, 9))
        x[a:a + n] = rng.integers(0, 256, min(n, len(good) - a), dtype=np.uint8).tobytes()
        packs.append(bytes(x))
        if k % 50 == 0:                                       # ... and a tail of garbage behind a good packing
            packs.append(good + rng.integers(0, 256, 5, dtype=np.uint8).tobytes())
    _decode_and_compare(codec, n_rows, n_cols, packs)
