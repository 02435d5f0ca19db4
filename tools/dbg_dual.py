import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, ctypes as C
import gridfour_amd, oracle
ctx = gridfour_amd.GvrsHipContext(0)
n_rows, n_cols, nt = 120, 150, 12960
b = gridfour_amd.DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(2 * n_rows * n_cols + 1024 + 15) // 16 * 16)
b.synth_dem(0x9E3779B97F4A7C15 + 2, 144)
b.encode(); b.decode(); ctx.synchronize()
vals, dec, st = b.get_values(), b.get_decoded(), b.get_dec_status()
badt = [t for t in range(nt) if st[t] != 0 or not np.array_equal(vals[t], dec[t])]
print("bad tiles", len(badt), badt[:10])
for t in badt[:4]:
    bad = np.nonzero(dec[t] != vals[t])[0]
    pk = b.get_packing(t)
    model = pk[1]
    m32, seed = oracle.predictor_encode(model, n_rows, n_cols, vals[t])
    _, _, cl, _ = oracle.huffman_encode(m32)
    sy = np.frombuffer(m32, np.uint8)
    print(t, "status", st[t], "model", model, "nbad", bad.size, "first", bad[:6], "maxlen", cl.max(), "nM32", len(m32), "nsyms used", (cl > 0).sum())
    # where in the symbol stream is the first bad cell?  (model 3: stream index ~ cell order for interior)
    print("   lens of used symbols:", sorted(set(cl[cl > 0].tolist())))
