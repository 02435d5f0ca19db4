"""Per-phase shader-clock cycles of k_canon_decode per tile (s_memtime stamps; diagnostic)."""
import ctypes as C, os, sys
os.environ["GVRS_HIP_DIAG"] = "1"           # the diagnostic flavour of the library carries the stamps
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gridfour_amd
from gridfour_amd import DeviceBuffer, DeviceTileBatch, lib
nt, n_rows, n_cols = 12960, 120, 150
ctx = gridfour_amd.GvrsHipContext(0)
b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(2 * n_rows * n_cols + 1024 + 15) // 16 * 16, codec="canon")
b.synth_dem(0x9E3779B97F4A7C15 + 2, 144)
L = lib(); L.gf_internal_set_decode_debug.argtypes = [C.c_void_p]
dbg = DeviceBuffer(ctx, 16 * 4 * nt).fill(0)
b.encode(); b.decode(); ctx.synchronize()
L.gf_internal_set_phase_limits.argtypes = [C.c_int, C.c_int]
L.gf_internal_set_phase_limits(0, int(sys.argv[1]) if len(sys.argv) > 1 else 0)
L.gf_internal_set_decode_debug(dbg.ptr); b.decode(); ctx.synchronize(); L.gf_internal_set_decode_debug(None)
L.gf_internal_set_phase_limits(0, 0)
st = dbg.download(np.uint32, 16 * nt).reshape(nt, 16).astype(np.int64)
d = np.diff(st[:, :7], axis=1) & 0xFFFFFFFF
for i, nme in enumerate(["0 code lengths (wave 0)", "0 tables + LUT", "1 sync pass + fix-ups", "1 chain end + prefix sums", "2 values to cells", "3 inverse"]):
    print("  %-28s median %9d  p90 %9d" % (nme, np.median(d[:, i]), np.percentile(d[:, i], 90)))
tot = (st[:, 6] - st[:, 0]) & 0xFFFFFFFF
print("  total per tile median %d p90 %d" % (np.median(tot), np.percentile(tot, 90)))
