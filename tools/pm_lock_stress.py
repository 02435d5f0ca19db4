"""Stress of the shared package-merge scratch (CanonPM, one per workgroup under a lock): with a library built with
-DGF_CN_FORCE_PM every code table of every tile goes through the package-merge, so the three waves of a canonical encoder
workgroup (two in the LSOP packer) contend for the lock on every tile.  Checks that nothing hangs and that every tile
survives the round trip (the packings are valid canonical-Huffman streams, though not the reference's byte for byte: the
package-merge breaks ties its own way when the length limit does not bind).
    python -m gridfour_amd.build --variant forcepm -DGF_CN_FORCE_PM ; GVRS_HIP_VARIANT=forcepm python tools/pm_lock_stress.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gridfour_amd  # noqa: E402
from gridfour_amd import DeviceTileBatch  # noqa: E402


def main():
    ctx = gridfour_amd.GvrsHipContext(0)
    for codec, shape in (("canon", (120, 150)), ("lsop", (120, 150)), ("canon", (17, 23))):
        n_rows, n_cols = shape
        cells = n_rows * n_cols
        nt = 12960
        b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(4 * cells + 1024 + 15) // 16 * 16, codec=codec)
        b.synth_dem(0x9E3779B97F4A7C15 + 5, 144)
        for _ in range(3):
            b.encode()
            b.decode()
        ctx.synchronize()
        es, ds = b.get_enc_status(), b.get_dec_status()
        ok = es == 0
        same = np.array_equal(b.get_decoded()[ok], b.get_values()[ok])
        print(codec, shape, "encoded", int(ok.sum()), "of", nt, "decode ok", bool((ds[ok] == 0).all()), "round trip", same)
        assert ok.sum() > nt // 2 and (ds[ok] == 0).all() and same
        b.free()
    print("pm lock stress ok")


if __name__ == "__main__":
    main()
