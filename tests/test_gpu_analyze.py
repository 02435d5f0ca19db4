"""ICompressionDecoder.analyze of CodecHuffman: the sums of CodecStats (compress/CodecStats.java:100-141)
gathered from a GPU pass over a batch of packings, against the same sums computed from the oracle's Huffman decode."""
import io
import math
import struct

import numpy as np
import pytest

import oracle
from tilegen import make_tile

pytestmark = pytest.mark.gpu
NULL = -2**31


def _expected(n_rows, n_cols, packings):
    """CodecHuffman.analyze restated over the oracle's HuffmanDecoder (CodecHuffman.java:172-199)."""
    stats = np.zeros((6, 7), np.int64)
    entropy = np.zeros(6)
    log2 = math.log(2.0)
    for pk in packings:
        n_m32 = struct.unpack_from("<i", pk, 6)[0]
        m32, end = oracle.huffman_decode(pk[10:], n_m32, 0)
        # bits in tree: single-symbol form 17, else 8 + (2n - 1) + 8n for n leaves
        n_leaf = pk[10] + 1
        bits_in_tree = 17 if ((pk[11] & 1) == 1) else 8 + 2 * n_leaf - 1 + 8 * n_leaf
        counts = np.bincount(np.frombuffer(m32[:n_m32], np.uint8), minlength=256)
        s = 0.0
        for i in range(256):
            if counts[i] > 0:
                p = counts[i] / float(n_m32)
                s += p * math.log(p) / log2
        for k in (pk[1], 5):
            stats[k] += [1, len(pk) - 10, n_rows * n_cols, bits_in_tree, 1, n_m32, int((counts > 0).sum())]
            entropy[k] -= s
    return stats, entropy


def test_analysis_sums_match_the_oracle():
    import gridfour_amd
    codec = gridfour_amd.CodecHuffmanHip()
    nr, nc = 60, 90
    tiles = [make_tile(k, nr, nc, seed=s) for s, k in enumerate(["smooth", "ramp", "noise8", "noise16", "steps", "uniform", "sparse_big",
                                                                   "smooth", "extremes"])]
    with_nulls = make_tile("smooth", nr, nc, seed=40).copy()
    with_nulls.reshape(nr, nc)[20:30, 10:70] = NULL
    tiles.append(with_nulls)
    packs, preds, st = codec.encode_batch(0, nr, nc, np.stack(tiles))
    packs = [p for p in packs if p is not None]
    assert len(packs) >= 9 and len(set(preds)) >= 3
    codec.clearAnalysisData()
    status = codec.analyze_batch(nr, nc, packs)
    assert (status == 0).all()
    want, want_e = _expected(nr, nc, packs)
    got = codec.analysis_data()
    for k in range(6):
        have = [int(got[k][f]) for f in ("n_tiles", "n_bytes", "n_symbols", "n_bits_overhead", "n_m32_counted", "sum_length_m32",
                                         "sum_observed_m32")]
        assert have == list(want[k]), (k, have, list(want[k]))
        assert got[k]["sum_entropy_m32"] == pytest.approx(want_e[k], rel=1e-12, abs=1e-12)
    # a second batch accumulates, one packing at a time behaves the same, clear resets
    codec.analyze(nr, nc, packs[0])
    assert int(codec.analysis_data()[5]["n_tiles"]) == len(packs) + 1
    out = io.StringIO()
    codec.reportAnalysisData(out, len(packs) + 1)
    text = out.getvalue()
    assert "Differencing" in text and "All Predictors" in text and "bits in tree" in text
    codec.clearAnalysisData()
    out = io.StringIO()
    codec.reportAnalysisData(out, 10)
    assert "Tiles Compressed:  0" in out.getvalue()


def _expected_h2(packings, k):
    """CodecStats.addCountsForM32 :150-156 and getH2 :157-190 restated: pair counts of neighbouring M32 bytes, then the
    conditional entropy (natural logarithm) the reference computes from them."""
    sA = np.zeros(256, np.int64)
    sB = np.zeros(65536, np.int64)
    for pk in packings:
        if k != 5 and pk[1] != k:
            continue
        n_m32 = struct.unpack_from("<i", pk, 6)[0]
        m32, _ = oracle.huffman_decode(pk[10:], n_m32, 0)
        b = np.frombuffer(m32[:n_m32], np.uint8).astype(np.int64)
        if n_m32 < 2:
            continue
        np.add.at(sA, b[1:], 1)
        np.add.at(sB, (b[:-1] << 8) | b[1:], 1)
    total = int(sA.sum())
    if total == 0:
        return 0.0, sB
    h2 = 0.0
    for i in range(256):
        if sA[i] > 0:
            p_i = sA[i] / float(total)
            row = sB[i * 256:(i + 1) * 256]
            n = float(row.sum())
            s = sum((c / n) * math.log(c / n) for c in row if c > 0)
            h2 += p_i * s
    return -h2, sB


def test_h2_pair_counts_match_the_oracle():
    """The sA / sB tables of CodecStats (the pair counts behind getH2) from the GPU pass, per predictor and for all."""
    import gridfour_amd
    codec = gridfour_amd.CodecHuffmanHip()
    nr, nc = 60, 90
    tiles = [make_tile(k, nr, nc, seed=s) for s, k in enumerate(["smooth", "ramp", "noise8", "noise16", "steps", "sparse_big", "smooth",
                                                                   "noise8", "steps"])]
    with_nulls = make_tile("smooth", nr, nc, seed=41).copy()
    with_nulls.reshape(nr, nc)[5:15, 20:60] = NULL
    tiles.append(with_nulls)
    packs, preds, st = codec.encode_batch(0, nr, nc, np.stack(tiles))
    packs = [p for p in packs if p is not None]
    codec.clearAnalysisData()
    assert (codec.analyze_batch(nr, nc, packs[:4]) == 0).all()
    assert (codec.analyze_batch(nr, nc, packs[4:]) == 0).all()          # a second batch accumulates
    got = codec.pair_counts()
    for k in range(6):
        want_h2, want_sB = _expected_h2(packs, k)
        assert np.array_equal(got[k], want_sB), k
        assert codec.getH2(k) == pytest.approx(want_h2, rel=1e-12, abs=1e-12)
    assert codec.getH2(5) > 0.0
    codec.clearAnalysisData()
    assert codec.getH2(5) == 0.0


def test_analysis_rejects_damaged_packings():
    import gridfour_amd
    codec = gridfour_amd.CodecHuffmanHip()
    nr, nc = 40, 40
    good = codec.encode(0, nr, nc, make_tile("smooth", nr, nc))
    codec.clearAnalysisData()
    st = codec.analyze_batch(nr, nc, [good, good[:30], good[:5]])
    assert st[0] == 0 and st[1] < 0 and st[2] < 0
    assert int(codec.analysis_data()[5]["n_tiles"]) == 1
    with pytest.raises(IOError):
        codec.analyze(nr, nc, good[:30])


def test_analysis_large_batch_dem():
    """the bench workload's tile shape: statistics of 300 DEM tiles in one pass"""
    import gridfour_amd
    ctx = gridfour_amd.GvrsHipContext(0)
    nr, nc, nt = 120, 150, 300
    b = gridfour_amd.DeviceTileBatch(ctx, nr, nc, nt)
    b.synth_dem(0x9E3779B97F4A7C15 + 2, 144)
    b.encode(codec_index=0)
    ctx.synchronize()
    lengths = b.get_lengths()
    packs = [b.get_packing(t, int(lengths[t])) for t in range(nt)]
    codec = gridfour_amd.CodecHuffmanHip(context=ctx)
    assert (codec.analyze_batch(nr, nc, packs) == 0).all()
    want, want_e = _expected(nr, nc, packs)
    got = codec.analysis_data()
    for k in range(6):
        assert int(got[k]["n_tiles"]) == want[k][0] and int(got[k]["sum_observed_m32"]) == want[k][6]
        assert int(got[k]["n_bits_overhead"]) == want[k][3] and int(got[k]["sum_length_m32"]) == want[k][5]
        assert got[k]["sum_entropy_m32"] == pytest.approx(want_e[k], rel=1e-12, abs=1e-12)
