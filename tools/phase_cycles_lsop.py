"""Per-phase shader-clock cycles of k_lsop_unpack2 per tile (s_memtime stamps of cd_decode_stream for the two streams; diagnostic)."""
import ctypes as C, os, sys
os.environ["GVRS_HIP_DIAG"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gridfour_amd
from gridfour_amd import DeviceBuffer, DeviceTileBatch, lib
nt, n_rows, n_cols = (int(sys.argv[3]), int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 3 else (12960, 120, 150)
ctx = gridfour_amd.GvrsHipContext(0)
b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(2 * n_rows * n_cols + 1024 + 15) // 16 * 16, codec="lsop")
b.synth_dem(0x9E3779B97F4A7C15 + 2, 144)
L = lib(); L.gf_internal_set_decode_debug.argtypes = [C.c_void_p]
dbg = DeviceBuffer(ctx, 16 * 4 * nt).fill(0)
b.encode(); b.decode(); ctx.synchronize()
L.gf_internal_set_decode_debug(dbg.ptr); b.decode(); ctx.synchronize(); L.gf_internal_set_decode_debug(None)
st = dbg.download(np.uint32, 16 * nt).reshape(nt, 16).astype(np.int64)
names = ["code lengths + tables", "LUT, pairs, tokens", "sync pass + fix-ups", "chain end + prefix sums", "values (+ expand)"]
for base, what in ((0, "initialisers (%d values)" % (4 * n_rows + 2 * n_cols - 9)), (8, "interior (%d values)" % ((n_rows - 2) * (n_cols - 4)))):
    print(what)
    for i, nme in enumerate(names):
        d = (st[:, base + i + 1] - st[:, base + i]) & 0xFFFFFFFF
        print("  %-28s median %9d  p90 %9d" % (nme, np.median(d), np.percentile(d, 90)))
hd = [(st[:, i + 1] - st[:, i]) & 0xFFFFFFFF for i in range(1, 5)]
if np.median(hd[0]) < 10**8:
    print("k_lsop_head (a lane per tile): staging %d, tables %d, initialisers %d, second stream's lengths %d cycles (medians)" % tuple(np.median(x) for x in hd))
print("between the streams           median %9d" % np.median((st[:, 8] - st[:, 5]) & 0xFFFFFFFF))
print("plane dump (wave 0, inside the value passes)  median %9d" % np.median(st[:, 15]))
tot = (st[:, 14] - st[:, 0]) & 0xFFFFFFFF
print("whole tile (behind the text's staging) median %d p90 %d" % (np.median(tot), np.percentile(tot, 90)))
