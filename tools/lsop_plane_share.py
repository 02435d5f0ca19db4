"""How many tiles of a batch leave k_lsop_unpack2 as byte planes (word 14 of a tile's coefficient record), and why the others do not:
residuals beyond a byte in the interior / among the rows' initialisers.   python tools/lsop_plane_share.py [workload]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gridfour_amd
from gridfour_amd import DeviceTileBatch
wl = sys.argv[1] if len(sys.argv) > 1 else "float256_lsop"
nr, nc, nt = (256, 256, 4096) if wl == "float256_lsop" else (120, 150, 12960)
ctx = gridfour_amd.GvrsHipContext(0)
b = DeviceTileBatch(ctx, nr, nc, nt, slot_stride=(2 * nr * nc + 1024 + 15) // 16 * 16, codec="lsop")
if wl == "float256_lsop":
    b.synth_dem(0x9E3779B97F4A7C15 + 5, 64)
    ctx.synchronize()
    f = b.get_values().astype(np.float32) * np.float32(0.1)
    b.values.upload(np.floor((f * np.float32(10.0)).astype(np.float64) + 0.5).astype(np.int32))
else:
    b.synth_dem(0x9E3779B97F4A7C15 + 2, 144)
b.encode(); b.decode(); ctx.synchronize()
fmt = b.coefs.download(np.uint32, nt * 16).reshape(nt, 16)[:, 14]
print("tiles", nt, "plane", int((fmt == 1).sum()), "int32", int((fmt == 0).sum()), "other", int((fmt > 1).sum()))
ok = bool(np.array_equal(b.get_decoded(), b.get_values()))
print("roundtrip ok", ok)
