"""Generates tests/golden/oracle_vectors.json from the C oracle (oracle-derived vectors).

Run from the repo root after the oracle passes tests/test_oracle_golden.py:
    python tests/golden/make_oracle_vectors.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from tilegen import NULL, add_nulls, make_tile  # noqa: E402


def main():
    cases = []

    def add(name, n_rows, n_cols, v, codec_index=0):
        p, used = oracle.codec_huffman_encode(codec_index, n_rows, n_cols, v)
        cases.append({"name": name, "codec_index": codec_index, "n_rows": n_rows, "n_cols": n_cols,
                      "values": [int(x) for x in v], "predictor": used,
                      "packing": None if p is None else p.hex()})

    for kind in ("ramp", "smooth", "noise8", "noise16", "noise32", "uniform", "extremes", "sparse_big"):
        add(kind + "_16x16", 16, 16, make_tile(kind, 16, 16))
        add(kind + "_7x9", 7, 9, make_tile(kind, 7, 9), codec_index=5)
    add("nulls_16x16", 16, 16, add_nulls(make_tile("smooth", 16, 16), 16, 16, 0.2))
    add("nulls_blocks_12x20", 12, 20, add_nulls(make_tile("smooth", 12, 20), 12, 20, 0.3, blocks=True))
    add("all_null_4x4", 4, 4, np.full(16, NULL, np.int32))
    add("two_cells", 1, 2, np.array([5, -900], np.int32))
    # the reference sample-file tile (Sample05, tile 0) through CodecHuffman instead of CodecDeflate
    r = np.arange(50)[:, None]
    c = np.arange(50)[None, :]
    add("sample05_tile0_50x50", 50, 50, (r * 100 + c - 1).astype(np.int32).ravel(), codec_index=0)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_oracle_vectors.py", "source": "oracle-derived", "cases": cases}, f)
    print(out, len(cases), "cases", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
