"""Per-phase shader-clock cycles of the decode kernel per tile (s_memtime stamps; diagnostic)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gridfour_amd  # noqa: E402
from gridfour_amd import DeviceBuffer, DeviceTileBatch, lib  # noqa: E402


def main():
    nt = 12960
    n_rows, n_cols = 120, 150
    ctx = gridfour_amd.GvrsHipContext(0)
    cells = n_rows * n_cols
    b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(2 * cells + 1024 + 15) // 16 * 16)
    b.synth_dem(0x9E3779B97F4A7C15 + 2, 144)
    L = lib()
    L.gf_internal_set_decode_debug.argtypes = [C.c_void_p]
    dbg = DeviceBuffer(ctx, 16 * 4 * nt).fill(0)
    b.encode()
    b.decode()
    ctx.synchronize()
    L.gf_internal_set_phase_limits.argtypes = [C.c_int, C.c_int]
    warm = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    L.gf_internal_set_phase_limits(0, warm << 8)
    L.gf_internal_set_decode_debug(dbg.ptr)
    b.decode()
    ctx.synchronize()
    L.gf_internal_set_decode_debug(None)
    L.gf_internal_set_phase_limits(0, 0)
    st = dbg.download(np.uint32, 16 * nt).reshape(nt, 16).astype(np.int64)
    d = np.diff(st[:, :11], axis=1) & 0xFFFFFFFF
    names = ["0 header copy", "0 tree parse", "1 LUT build", "1 huffman chain sync", "1 huffman write pass", "2 m32 chain sync",
             "2 bitmap+rank", "2 value decode+store", "3 colsum+col0", "3 row scans"]
    for i, nme in enumerate(names):
        print("  %-26s median %9d  p90 %9d" % (nme, np.median(d[:, i]), np.percentile(d[:, i], 90)))
    print("  huffman chain: rounds median %d p90 %d max %d; first pass cycles median %d" % (
        np.median(st[:, 11]), np.percentile(st[:, 11], 90), st[:, 11].max(), np.median(st[:, 12])))
    print("  m32 chain:     rounds median %d p90 %d max %d; first pass cycles median %d" % (
        np.median(st[:, 13]), np.percentile(st[:, 13], 90), st[:, 13].max(), np.median(st[:, 14])))
    tot = (st[:, 10] - st[:, 0]) & 0xFFFFFFFF
    print("  total per tile median %d  p90 %d" % (np.median(tot), np.percentile(tot, 90)))


if __name__ == "__main__":
    main()
