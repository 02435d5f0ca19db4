"""Runs the encode and/or decode kernel a few times on a bench workload, optionally phase-limited
(diagnostic driver for rocprofv3 --pmc runs).  usage: run_kernels.py <enc|dec|both> <encLimit> <decLimit> <reps> [workload] [huffman|canon|lsop]"""
import ctypes as C
import os
import sys

# phase limits exist in the diagnostic flavour of the library only; without limits (the HBM traffic passes of pmc_hbm.sh) the
# SHIPPING library is measured -- the diagnostic build's kernels carry stamps and other register budgets
if (len(sys.argv) > 3 and (int(sys.argv[2]) or int(sys.argv[3]))) or os.environ.get("GVRS_HIP_DIAG"):
    os.environ["GVRS_HIP_DIAG"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gridfour_amd  # noqa: E402
from gridfour_amd import DeviceTileBatch, lib  # noqa: E402


def main():
    which, el, dl, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    wl = sys.argv[5] if len(sys.argv) > 5 else "etopo1"
    codec = sys.argv[6] if len(sys.argv) > 6 else "huffman"
    n_rows, n_cols, nt, tpr = {"etopo1": (120, 150, 12960, 144), "etopo1_rough": (120, 150, 12960, 144), "dem1024": (200, 200, 1024, 32)}[wl]
    ctx = gridfour_amd.GvrsHipContext(0)
    cells = n_rows * n_cols
    b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(2 * cells + 1024 + 15) // 16 * 16, codec=codec)
    b.synth_dem(0x9E3779B97F4A7C15 + 2, tpr, style=1 if wl == "etopo1_rough" else 0)
    L = lib()
    diag = bool(os.environ.get("GVRS_HIP_DIAG"))
    if diag:
        L.gf_internal_set_phase_limits.argtypes = [C.c_int, C.c_int]
    b.encode()
    ctx.synchronize()
    if diag:
        L.gf_internal_set_phase_limits(el, dl)
    for _ in range(reps):
        if which in ("enc", "both"):
            b.encode()
        if which in ("dec", "both"):
            b.decode()
    ctx.synchronize()
    if diag:
        L.gf_internal_set_phase_limits(0, 0)


if __name__ == "__main__":
    main()
