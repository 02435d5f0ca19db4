"""Where an LSOP12 decode differs from the cells it was made from: rows / columns of the first bad tile (after a change to
k_lsop_reconstruct).   python tools/lsop_mismatch.py [rows cols]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gridfour_amd, oracle
nr, nc = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (120, 150)
tiles = np.asarray(oracle.dem_tiles(oracle.DEM_SEED + 2, nr, nc, 144, 0, 4)).reshape(4, -1)
codec = gridfour_amd.LsCodecHip(deflate_enabled=False)
packs, types, status = codec.encode_batch(0, nr, nc, tiles)
vals, st = codec.decode_batch(nr, nc, packs)
for t in range(len(packs)):
    got = np.asarray(vals[t]).reshape(nr, nc); want = tiles[t].reshape(nr, nc)
    bad = np.argwhere(got != want)
    print("tile", t, "status", st[t], "bad cells", len(bad))
    if len(bad):
        rows = sorted(set(bad[:, 0].tolist()))
        print("  rows with errors:", rows[:20], "...", rows[-5:])
        r = rows[0]
        cols = bad[bad[:, 0] == r][:, 1]
        print("  first bad row", r, "cols", cols[:12].tolist(), "n", len(cols), "got", got[r, cols[:4]].tolist(), "want", want[r, cols[:4]].tolist())
        break
