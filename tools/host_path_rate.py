"""Throughput of the host-memory entry points (gf_*_batch_i32: host buffers in, packings out and back), i.e. the
PCIe-inclusive figure beside the device-resident one.  Raw C calls on numpy buffers, warm (second) call timed.
usage: host_path_rate.py [workload] [shards]     (shards > 1: gf_multi over that many contexts on device 0)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridfour_amd
from gridfour_amd import lib, PinnedArray, DeviceTileBatch
from gridfour_amd.sharding import _ptr

wl = sys.argv[1] if len(sys.argv) > 1 else "etopo1"
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_rows, n_cols, nt, tpr = {"etopo1": (120, 150, 12960, 144), "gebco_full": (200, 200, 93312, 432), "dem1024": (200, 200, 1024, 32)}[wl]
cells = n_rows * n_cols
ctx = gridfour_amd.GvrsHipContext(0)
# the tiles: generated on the device in pieces, brought to host memory
vals = np.empty((nt, cells), np.int32)
piece = 4096
for t0 in range(0, nt, piece):
    n = min(piece, nt - t0)
    b = DeviceTileBatch(ctx, n_rows, n_cols, n)
    b.synth_dem(0x9E3779B97F4A7C15 + 2, tpr, tile0=t0)
    ctx.synchronize()
    vals[t0:t0 + n] = b.get_values()
    del b
res = {"workload": wl, "tiles": nt, "bytes": int(vals.nbytes), "shards": shards}
multi = gridfour_amd.GvrsHipMulti([0] * shards) if shards > 1 else None
modes = (("pageable", False),) if vals.nbytes > (4 << 30) else (("pageable", False), ("pinned", True))   # big inputs: one copy in RAM is enough
for name, pinned in modes:
    src = vals
    if pinned:
        pin = PinnedArray(vals.shape, np.int32)
        pin.array[:] = vals
        src = pin.array
    cap = nt * cells * 2 if vals.nbytes <= (4 << 30) else nt * cells
    keep = [PinnedArray(cap, np.uint8), PinnedArray(vals.shape, np.int32)] if pinned else None     # owners of the pinned arrays
    blob = keep[0].array if pinned else np.empty(cap, np.uint8)
    out = keep[1].array if pinned else np.empty_like(vals)
    off = np.zeros(nt + 1, np.uint64)
    st = np.zeros(nt, np.int32)
    def enc():
        if multi:
            return lib().gf_huffman_encode_batch_i32_multi(multi.handle, 0, n_rows, n_cols, nt, _ptr(src), _ptr(blob), cap, _ptr(off), None, _ptr(st))
        return lib().gf_huffman_encode_batch_i32(ctx.handle, 0, n_rows, n_cols, nt, _ptr(src), _ptr(blob), cap, _ptr(off), None, _ptr(st))
    def dec():
        if multi:
            return lib().gf_huffman_decode_batch_i32_multi(multi.handle, n_rows, n_cols, nt, _ptr(blob), _ptr(off), _ptr(out), _ptr(st))
        return lib().gf_huffman_decode_batch_i32(ctx.handle, n_rows, n_cols, nt, _ptr(blob), _ptr(off), _ptr(out), _ptr(st))
    assert enc() == 0 and dec() == 0                 # warm-up: staging buffers get allocated
    best = [1e9, 1e9]
    for _ in range(3 if vals.nbytes <= (4 << 30) else 1):
        t0 = time.perf_counter(); rc = enc(); t1 = time.perf_counter()
        assert rc == 0 and (st == 0).all()
        rc = dec(); t2 = time.perf_counter()
        assert rc == 0 and (st == 0).all()
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1)]
    assert np.array_equal(out, vals)
    gb = vals.nbytes / 1e9
    res[name] = {"encode_GBps": round(gb / best[0], 2), "decode_GBps": round(gb / best[1], 2),
                 "roundtrip_GBps": round(gb / (best[0] + best[1]), 2), "compressed_bytes": int(off[nt])}
    print(name, json.dumps(res[name]), flush=True)
print(json.dumps(res))
