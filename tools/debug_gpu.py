"""GPU debug helper: per-predictor comparison of HIP packings with the oracle (test-side tool)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gridfour_amd  # noqa: E402
import oracle  # noqa: E402
from gridfour_amd import DeviceTileBatch  # noqa: E402
from tilegen import KINDS, make_tile, add_nulls  # noqa: E402


def main():
    t0 = time.time()
    codec = gridfour_amd.CodecHuffmanHip()
    print("context up in %.1fs" % (time.time() - t0), flush=True)
    shapes = [(10, 10), (7, 9), (33, 65), (120, 150)]
    for n_rows, n_cols in shapes:
        tiles = np.stack([make_tile(k, n_rows, n_cols) for k in KINDS])
        b = DeviceTileBatch(codec.ctx, n_rows, n_cols, len(tiles))
        b.values.upload(tiles)
        for model in (1, 2, 3):
            b.encode(codec_index=3, predictor_mask=1 << (model - 1))
            codec.ctx.synchronize()
            lengths = b.get_lengths()
            st = b.get_enc_status()
            for t, kind in enumerate(KINDS):
                ref, _ = oracle.codec_huffman_encode(3, n_rows, n_cols, tiles[t], predictor_mask=1 << (model - 1))
                got = b.get_packing(t, int(lengths[t]))
                if got == ref:
                    continue
                first = next((i for i in range(min(len(ref), len(got))) if got[i] != ref[i]), -1)
                print("MISMATCH %dx%d %s model %d status %d: len got %d want %d, first diff byte %d" % (
                    n_rows, n_cols, kind, model, st[t], len(got), len(ref), first), flush=True)
                if first >= 0:
                    print("   got ", got[max(0, first - 4):first + 12].hex())
                    print("   want", ref[max(0, first - 4):first + 12].hex())
        # decode of oracle packings, per model
        for model in (1, 2, 3):
            packs = [oracle.codec_huffman_encode(3, n_rows, n_cols, tiles[t], predictor_mask=1 << (model - 1))[0]
                     for t in range(len(tiles))]
            vals, dst = codec.decode_batch(n_rows, n_cols, packs)
            for t, kind in enumerate(KINDS):
                if dst[t] != 0 or not np.array_equal(vals[t], tiles[t]):
                    bad = np.nonzero(vals[t] != tiles[t])[0]
                    print("DECODE MISMATCH %dx%d %s model %d status %d: %d cells differ, first %s" % (
                        n_rows, n_cols, kind, model, dst[t], bad.size, bad[:5]), flush=True)
        # nulls
        nt = add_nulls(make_tile("smooth", n_rows, n_cols), n_rows, n_cols, 0.2)
        ref, _ = oracle.codec_huffman_encode(3, n_rows, n_cols, nt)
        got = codec.encode(3, n_rows, n_cols, nt)
        print("nulls %dx%d encode %s" % (n_rows, n_cols, "ok" if got == ref else "MISMATCH len %d vs %d" % (len(got or b''), len(ref))), flush=True)
        try:
            d = codec.decode(n_rows, n_cols, ref)
            print("nulls decode", "ok" if np.array_equal(d, nt) else "MISMATCH %d cells" % int((d != nt).sum()), flush=True)
        except Exception as e:
            print("nulls decode raised", e, flush=True)
        b.free()
        print("shape %dx%d done" % (n_rows, n_cols), flush=True)


if __name__ == "__main__":
    main()
