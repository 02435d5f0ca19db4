"""Per-phase shader-clock cycles of the encode kernel per tile (s_memtime stamps; diagnostic)."""
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
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 12960
    n_rows, n_cols = 120, 150
    ctx = gridfour_amd.GvrsHipContext(0)
    cells = n_rows * n_cols
    b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(2 * cells + 1024 + 15) // 16 * 16)
    b.synth_dem(0x9E3779B97F4A7C15 + 2, 144, style=int(os.environ.get("GF_DEM_STYLE", "0")))
    L = lib()
    L.gf_internal_encode_debug_words.restype = C.c_size_t
    L.gf_internal_set_encode_debug.argtypes = [C.c_void_p]
    words = L.gf_internal_encode_debug_words()
    dbg = DeviceBuffer(ctx, words * 4 * nt).fill(0)
    b.encode()
    ctx.synchronize()
    L.gf_internal_set_encode_debug(dbg.ptr)
    b.encode()
    ctx.synchronize()
    L.gf_internal_set_encode_debug(None)
    raw = dbg.download(np.uint32, words * nt).reshape(nt, words)
    st = raw[:, words - 16:words - 8].astype(np.int64)
    d = np.diff(st, axis=1) & 0xFFFFFFFF
    names = ["A hist pass", "A->reduce (nulls/replicas)", "B1 sort", "B2 merge", "B3 codes",
             "(k_huffman_encode -> k_huffman_pack: not a phase)", "C pack (k_huffman_pack: window init, pack)"]
    print("tiles", nt, "median / p90 ticks per phase (s_memtime).  The diagnostic flavour runs the ONE-KERNEL encoder (phases A and B in\n"
          "k_huffman_encode); the shipping batches run part 1 + k_huffman_trees + k_huffman_pack (gvrs_encode.hip)")
    for i, nme in enumerate(names):
        print("  %-28s median %9d  p90 %9d  max %9d" % (nme, np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
    tot = ((st[:, 5] - st[:, 0]) & 0xFFFFFFFF) + ((st[:, 7] - st[:, 6]) & 0xFFFFFFFF)
    print("  both kernels per tile median %d  p90 %d" % (np.median(tot), np.percentile(tot, 90)))


if __name__ == "__main__":
    main()
