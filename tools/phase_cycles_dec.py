"""Per-phase shader-clock cycles of the decode kernel per tile (s_memtime stamps; diagnostic)."""
import ctypes as C
import os
import sys

os.environ["GVRS_HIP_DIAG"] = "1"           # the diagnostic flavour of the library carries the stamps

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gridfour_amd  # noqa: E402
from gridfour_amd import DeviceBuffer, DeviceTileBatch, lib  # noqa: E402


def main():
    nt = int(os.environ.get("GF_NT", "12960"))                   # GF_NT=1: a workgroup alone on the chip (the one-tile call)
    n_rows, n_cols = 120, 150
    ctx = gridfour_amd.GvrsHipContext(0)
    cells = n_rows * n_cols
    b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(2 * cells + 1024 + 15) // 16 * 16)
    b.synth_dem(0x9E3779B97F4A7C15 + 2, 144, style=int(os.environ.get("GF_DEM_STYLE", "0")))     # GF_DEM_STYLE=1: the rough surface
    L = lib()
    L.gf_internal_set_decode_debug.argtypes = [C.c_void_p]
    dbg = DeviceBuffer(ctx, 16 * 4 * nt).fill(0)
    b.encode()
    b.decode()
    ctx.synchronize()
    L.gf_internal_set_phase_limits.argtypes = [C.c_int, C.c_int]
    warm = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    L.gf_internal_set_phase_limits(0, (warm << 8) if warm < 256 else warm)     # >= 256: raw diagnostic flags (0x10000: no stores)
    L.gf_internal_set_decode_debug(dbg.ptr)
    b.decode()
    ctx.synchronize()
    L.gf_internal_set_decode_debug(None)
    L.gf_internal_set_phase_limits(0, 0)
    st = dbg.download(np.uint32, 16 * nt).reshape(nt, 16).astype(np.int64)
    rel = (st[:, :11] - st[:, :1]) & 0xFFFFFFFF
    names = ["0 start", "1 header", "2 tree records", "3 LUT built", "4 huffman sync pass", "5 huffman write pass", "6 value starts marked + ranked", "7 border prologue",
             "8 values (+ inverse when fused)", "9 -", "10 end"]
    for i in (1, 2, 3, 4, 5, 6, 7, 8, 10):
        print("  after %-34s median %9d  p90 %9d" % (names[i], np.median(rel[:, i]), np.percentile(rel[:, i], 90)))
    print("  huffman sync pass: rounds median %d p90 %d max %d" % (np.median(st[:, 11]), np.percentile(st[:, 11], 90), st[:, 11].max()))
    for i, nme in ((13, "text staged in LDS (+ count table)"), (12, "round 1, wave 0"), (14, "round 1, all waves (first barrier)"),
                   (15, "whole pass (rounds + prefix sums)")):
        print("    %-40s median %9d  p90 %9d" % (nme, np.median(st[:, i]), np.percentile(st[:, i], 90)))
    if warm & 0x20000:
        print("  chunk loop, wave 0, summed over the chunks (instead of the sync stamps above):")
        for i, nme in ((12, "decode + wave scans"), (13, "scan barrier"), (14, "bases + ring writes (+ barrier)"), (15, "rows finished (+ barrier)")):
            print("    %-40s median %9d  p90 %9d" % (nme, np.median(st[:, i]), np.percentile(st[:, i], 90)))
    pred = b.get_predictors()
    if True:
        print("  predictors chosen:", {int(k): int(v) for k, v in zip(*np.unique(pred, return_counts=True))})
    tot = (st[:, 10] - st[:, 0]) & 0xFFFFFFFF
    print("  total per tile median %d  p90 %d  p99 %d  max %d  mean %d" % (np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), tot.mean()))
    lengths = b.get_lengths()
    for t in np.argsort(tot)[::-1][:6]:                          # the slowest tiles, stamp by stamp
        hdr = b.get_packing(int(t), 10)
        print("  slow tile %5d: predictor %d, packing %d bytes, nM32 %d, stamps %s, sync rounds %d" % (
            t, pred[t], lengths[t], int.from_bytes(hdr[6:10], "little"), [int(x) for x in rel[t, 1:11]], st[t, 11]))
    # (tiles a second run of the kernel redid carry that run's stamps)
    for m in (1, 2, 3, 4):
        sel = pred == m
        if sel.any():
            print("  predictor %d (%d tiles): mean cycles to LUT %d, sync %d, write %d, values %d, end %d" % (
                m, sel.sum(), rel[sel, 3].mean(), rel[sel, 4].mean(), rel[sel, 5].mean(), rel[sel, 8].mean(), rel[sel, 10].mean()))


if __name__ == "__main__":
    main()
