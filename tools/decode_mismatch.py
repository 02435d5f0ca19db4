import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gridfour_amd
from gridfour_amd import DeviceTileBatch
nr, nc, nt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = gridfour_amd.GvrsHipContext(0)
b = DeviceTileBatch(ctx, nr, nc, nt, slot_stride=(2 * nr * nc + 1024 + 15) // 16 * 16)
b.synth_dem(0x9E3779B97F4A7C15 + 2, 144)
b.encode(); b.decode(); ctx.synchronize()
d = b.get_decoded().reshape(nt, nr, nc); v = b.get_values().reshape(nt, nr, nc)
bad = np.argwhere(d != v)
print("mismatches", len(bad), "tiles", len(set(bad[:, 0])) if len(bad) else 0)
if len(bad):
    t = bad[0][0]
    bt = bad[bad[:, 0] == t]
    print("tile", t, "rows", sorted(set(bt[:, 1]))[:20], "cols min/max", bt[:, 2].min(), bt[:, 2].max(), "n", len(bt))
    print("pred", b.get_predictors()[t])
    r, c = bt[0][1], bt[0][2]
    print("first", r, c, d[t, r, c], v[t, r, c], d[t, r, c] - v[t, r, c])
