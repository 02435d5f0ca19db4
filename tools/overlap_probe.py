"""Do the codec's kernels gain from running beside each other?  One batch on one context against the same tiles as K shares on K
contexts of the same device (each context has its own stream and scratch buffers): wall clock over N steps, one synchronisation at
the end.  usage: python tools/overlap_probe.py [K=2] [nRows nCols nTiles] [codec]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gridfour_amd
from gridfour_amd import DeviceTileBatch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nr, nc, nt = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (120, 150, 12960)
codec = sys.argv[5] if len(sys.argv) > 5 else "huffman"
N = 50
stride = (2 * nr * nc + 1024 + 15) // 16 * 16

def make(n_ctx):
    out = []
    share = nt // n_ctx
    for k in range(n_ctx):
        ctx = gridfour_amd.GvrsHipContext(0)
        b = DeviceTileBatch(ctx, nr, nc, share, slot_stride=stride, codec=codec)
        b.synth_dem(0x9E3779B97F4A7C15 + 2, 144, tile0=k * share, style=int(os.environ.get("GF_DEM_STYLE", "0")))
        ctx.synchronize()
        out.append((ctx, b))
    return out

def run(sets, what):
    def step():
        for _, b in sets:
            getattr(b, what)()
    for _ in range(3):
        step()
    for c, _ in sets:
        c.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    for c, _ in sets:
        c.synchronize()
    return (time.perf_counter() - t0) / N * 1e3

one = make(1)
e1, d1 = run(one, "encode"), run(one, "decode")
many = make(K)
eK, dK = run(many, "encode"), run(many, "decode")
ok = all(bool(np.array_equal(b.get_decoded(), b.get_values())) for _, b in many)
print("%dx%d x %d (%s): one context encode %.3f decode %.3f ms | %d contexts side by side encode %.3f decode %.3f ms | ok %s" % (
    nr, nc, nt, codec, e1, d1, K, eK, dK, ok))
