"""Decode time vs warm-up length of the Huffman subsequence synchronisation (experiment)."""
import ctypes as C, os, sys
os.environ["GVRS_HIP_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridfour_amd
from gridfour_amd import DeviceTileBatch, GpuTimer, lib
ctx = gridfour_amd.GvrsHipContext(0)
n_rows, n_cols, nt = 120, 150, 12960
b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(2 * n_rows * n_cols + 1024 + 15) // 16 * 16)
b.synth_dem(0x9E3779B97F4A7C15 + 2, 144, style=int(os.environ.get("GF_DEM_STYLE", "0")))     # GF_DEM_STYLE=1: the rough surface
L = lib(); L.gf_internal_set_phase_limits.argtypes = [C.c_int, C.c_int]
b.encode(); ctx.synchronize()
vals = b.get_values()
for warm in [int(x) for x in sys.argv[1:]] or (128, 64, 96, 112, 144, 160, 192, 224, 255, 128):
    L.gf_internal_set_phase_limits(0, warm << 8)
    for _ in range(2): b.decode()
    tm = GpuTimer(ctx); tm.start()
    for _ in range(10): b.decode()
    tm.stop(); ms = tm.elapsed_ms() / 10
    ok = np.array_equal(b.get_decoded(), vals) and (b.get_dec_status() == 0).all()
    print("warm %3d bits: decode %.3f ms ok=%s" % (warm, ms, ok))
L.gf_internal_set_phase_limits(0, 0)
