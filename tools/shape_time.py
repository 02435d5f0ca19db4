"""Encode / decode time of a DEM batch of an arbitrary tile shape (diagnostic: occupancy experiments).
    python tools/shape_time.py nRows nCols nTiles [codec]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gridfour_amd
from gridfour_amd import DeviceTileBatch, GpuTimer
nr, nc, nt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
codec = sys.argv[4] if len(sys.argv) > 4 else "huffman"
ctx = gridfour_amd.GvrsHipContext(0)
b = DeviceTileBatch(ctx, nr, nc, nt, slot_stride=(2 * nr * nc + 1024 + 15) // 16 * 16, codec=codec)
b.synth_dem(0x9E3779B97F4A7C15 + 2, 144, style=int(os.environ.get("GF_DEM_STYLE", "0")))
t = GpuTimer(ctx)
def timeit(fn, reps=7):
    fn(); ctx.synchronize()
    ms = []
    for _ in range(reps):
        t.start(); fn(); t.stop(); ms.append(t.elapsed_ms())
    return float(np.median(ms))
e = timeit(b.encode); d = timeit(b.decode)
ok = bool(np.array_equal(b.get_decoded(), b.get_values()))
print("%dx%d x %d tiles (%s): encode %.3f ms, decode %.3f ms, %.1f / %.1f Gcell/s, roundtrip ok %s" % (
    nr, nc, nt, codec, e, d, nr * nc * nt / e / 1e6, nr * nc * nt / d / 1e6, ok))
