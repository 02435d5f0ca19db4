"""CodecFloat through the host-memory entry points on the whole float256 workload (4096 tiles of 256x256 float32): encode =
GPU planes + Deflate on the host's zlib threads; decode = container walk + inflate + plane merge on the GPU behind the
pipelined staging.  Raw C calls on numpy buffers; the decode is timed warm."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridfour_amd
from gridfour_amd import lib, DeviceTileBatch
from gridfour_amd.sharding import _ptr

n_rows, n_cols, nt = 256, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cells = n_rows * n_cols
ctx = gridfour_amd.GvrsHipContext(0)
gen = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=16)
gen.synth_dem(0x9E3779B97F4A7C15 + 5, 64)
ctx.synchronize()
vals = (gen.get_values().astype(np.float32) * np.float32(0.1)).reshape(nt, cells)
del gen
cap = nt * (5 * cells + 4096)
blob = np.empty(cap, np.uint8)
off = np.zeros(nt + 1, np.uint64)
t0 = time.perf_counter()
assert lib().gf_float_encode_batch_f32(ctx.handle, 0, n_rows, n_cols, nt, _ptr(vals), level, _ptr(blob), cap, _ptr(off)) == 0
t1 = time.perf_counter()
out = np.empty_like(vals)
st = np.zeros(nt, np.int32)
best = 1e9
for i in range(3):
    a = time.perf_counter()
    assert lib().gf_float_decode_batch_f32(ctx.handle, n_rows, n_cols, nt, _ptr(blob), _ptr(off), _ptr(out), _ptr(st)) == 0
    b = time.perf_counter()
    if i:
        best = min(best, b - a)
assert (st == 0).all() and np.array_equal(out.view(np.uint32), vals.view(np.uint32))
gb = vals.nbytes / 1e9
print(json.dumps({"workload": "float256", "tiles": nt, "zlib_level": level, "encode_GBps": round(gb / (t1 - t0), 3),
                  "decode_GBps": round(gb / best, 2), "compressed_bytes_per_cell": round(int(off[nt]) / (nt * cells), 4)}))
