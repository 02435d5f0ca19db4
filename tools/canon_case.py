"""One saved soak case (tools/soak.py replay ...) through the canonical codec of the library and the oracle: which tiles differ,
alone and as a batch; with GVRS_HIP_VARIANT=<name> through an experiment build (bisecting).
    python tools/canon_case.py tests/golden/soak/<case>.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gridfour_amd, oracle
d = np.load(sys.argv[1])
tiles = d["tiles"]; nr, nc = map(int, d["shape"])
codec = gridfour_amd.CodecCanonHuffmanHip()
packs = [oracle.codec_canon_encode(7, nr, nc, v)[0] for v in tiles]
good = [p for p in packs if p is not None]
vals, st = codec.decode_batch(nr, nc, good)
for k, p in enumerate(good):
    try:
        want = oracle.codec_canon_decode(nr, nc, p)
    except IOError as ex:
        print("batch: tile", k, "model", p[1], "status", st[k], "oracle:", ex)
        continue
    print("batch: tile", k, "model", p[1], "status", st[k], "equal", bool(st[k] == 0 and np.array_equal(vals[k], want)))
for k, p in enumerate(good):
    v1, s1 = codec.decode_batch(nr, nc, [p])
    try:
        want = oracle.codec_canon_decode(nr, nc, p)
    except IOError as ex:
        print("alone: tile", k, "status", s1[0], "oracle:", ex)
        continue
    print("alone: tile", k, "status", s1[0], "equal", bool(s1[0] == 0 and np.array_equal(v1[0], want)))
