"""Phase ablation of the encode/decode kernels on a bench workload (diagnostic tool)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gridfour_amd  # noqa: E402
from gridfour_amd import DeviceTileBatch, GpuTimer, lib  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "etopo1"
    n_rows, n_cols, nt, tpr = {"etopo1": (120, 150, 12960, 144), "dem1024": (200, 200, 1024, 32),
                               "gebco_shard": (200, 200, 11664, 432)}[wl]
    ctx = gridfour_amd.GvrsHipContext(0)
    cells = n_rows * n_cols
    b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(2 * cells + 1024 + 15) // 16 * 16)
    b.synth_dem(0x9E3779B97F4A7C15 + 2, tpr)
    L = lib()
    L.gf_internal_set_phase_limits.argtypes = [C.c_int, C.c_int]
    t = GpuTimer(ctx)

    def timeit(fn, reps=5):
        fn()
        ctx.synchronize()
        ms = []
        for _ in range(reps):
            t.start()
            fn()
            t.stop()
            ms.append(t.elapsed_ms())
        return float(np.median(ms))

    L.gf_internal_set_phase_limits(0, 0)
    full_enc = timeit(b.encode)
    full_dec = timeit(b.decode)
    print("%s: encode full %.3f ms, decode full %.3f ms" % (wl, full_enc, full_dec))
    for lim, name in ((1, "A (null scan + 3 histograms)"), (2, "A+B (+ trees)")):
        L.gf_internal_set_phase_limits(lim, 0)
        print("  encode up to %-32s %.3f ms" % (name, timeit(b.encode)))
    L.gf_internal_set_phase_limits(0, 0)
    b.encode()
    ctx.synchronize()
    for lim, name in ((1, "0 (header, tree parse)"), (2, "0+1 (+LUT, Huffman -> M32)"), (3, "0+1+2 (+M32 -> residual scatter)")):
        L.gf_internal_set_phase_limits(0, lim)
        print("  decode up to %-32s %.3f ms" % (name, timeit(b.decode)))
    L.gf_internal_set_phase_limits(0, 0)


if __name__ == "__main__":
    main()
