"""The GPU's CodecCanonHuffman (through the C ABI) against the golden vectors of the independent Python restatement
(oracle/canon_ref.py -> tests/golden/canon_vectors.json): packing bytes, chosen predictor, decode of those bytes."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VEC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "canon_vectors.json")))


@pytest.fixture(scope="module")
def codec():
    import gridfour_amd
    return gridfour_amd.CodecCanonHuffmanHip()


@pytest.mark.parametrize("case", VEC["codec"], ids=lambda c: c["name"])
def test_codec_vectors_on_gpu(codec, case):
    v = np.array(case["values"], dtype=np.int64).astype(np.int32)
    if case.get("throws"):
        with pytest.raises(ValueError):                        # IllegalArgumentException in the reference
            codec.encode(case["codec_index"], case["rows"], case["cols"], v)
        return
    got = codec.encode(case["codec_index"], case["rows"], case["cols"], v)
    if case["hex"] is None:
        assert got is None
        return
    assert got is not None and got.hex() == case["hex"], case["name"]
    assert got[1] == case["predictor"]
    assert np.array_equal(codec.decode(case["rows"], case["cols"], bytes.fromhex(case["hex"])), v)


def test_codec_vectors_as_one_batch(codec):
    """the same tiles of equal shape through the batch entry point (one launch)"""
    same = [c for c in VEC["codec"] if (c["rows"], c["cols"]) == (4, 5) and not c.get("throws")]
    tiles = np.array([c["values"] for c in same], dtype=np.int64).astype(np.int32)
    packs, preds, st = codec.encode_batch(same[0]["codec_index"], 4, 5, tiles)
    for c, p in zip(same, packs):
        if c["codec_index"] == same[0]["codec_index"]:
            assert (p.hex() if p is not None else None) == c["hex"]
