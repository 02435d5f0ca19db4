"""s_memtime stamps inside k_huffman_pack per tile (diagnostic flavour, three-kernel encoder: GF_DIAG_SPLIT=1): the table load, the
header + tree image, the head of the winner's stream, the flat scan with the final words.  GF_DEM_STYLE=1 = the rough surface."""
import ctypes as C
import os
import sys

os.environ["GVRS_HIP_DIAG"] = "1"
os.environ["GF_DIAG_SPLIT"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gridfour_amd  # noqa: E402
from gridfour_amd import DeviceBuffer, DeviceTileBatch, lib  # noqa: E402

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 12960
n_rows, n_cols = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (120, 150)
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
st = dbg.download(np.uint32, words * nt).reshape(nt, words)[:, words - 16:].astype(np.int64)
d = lambda a, c: (st[:, c] - st[:, a]) & 0xFFFFFFFF
rows = [("record -> LDS (table, tree image)", d(6, 8)), ("header + tree image out", d(8, 9)), ("head of the stream", d(9, 10)),
        ("flat scan + last words", d(10, 7)), ("whole tile", d(6, 7)), ("phase A (k_huffman_encode<true, 1, true>)", d(0, 1))]
print("tiles %d of %dx%d: ticks per tile (s_memtime, 100 MHz) -- median / p90 / max" % (nt, n_rows, n_cols))
for name, v in rows:
    print("  %-44s %8d %8d %8d" % (name, np.median(v), np.percentile(v, 90), v.max()))
