"""Throughput of the default-codec-list path: what CodecMaster.encode (gvrs/CodecMaster.java:142-193) and
RecordManager.writeTile (gvrs/RecordManager.java:386-490) would call with the standard codec list Huffman / Deflate / Float /
CanonHuffman... of GvrsFileSpecification.java:221-230 -- gf_codec_master_{encode,decode}_batch_i32 and
gf_tile_record_{encode,decode}_batch on an ETOPO1-shaped batch in host memory; next to it the rate of the host's zlib alone on
the M32 streams the Deflate candidates consist of (the bound of anything that must reproduce zlib's bytes).
    python tools/codec_master_rate.py [nTiles] [codec list, e.g. 1,2,0,3]"""
import ctypes as C
import json
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gridfour_amd  # noqa: E402
from gridfour_amd import DeviceTileBatch, lib  # noqa: E402
from gridfour_amd._lib import check  # noqa: E402


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 12960
    codecs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,0,3").split(",")]
    n_rows, n_cols = 120, 150
    cells = n_rows * n_cols
    ctx = gridfour_amd.GvrsHipContext(0)
    b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=16)
    b.synth_dem(0x9E3779B97F4A7C15 + 2, 144)
    ctx.synchronize()
    vals = b.get_values().reshape(nt, cells)
    del b
    L = lib()
    cd = (C.c_int * len(codecs))(*codecs)
    gb = vals.nbytes / 1e9
    out = {"workload": "etopo1: %d tiles of %dx%d int32 (%.2f GB) in pageable host memory" % (nt, n_rows, n_cols, gb), "codec_list": codecs,
           "threads": len(os.sched_getaffinity(0))}

    cap = nt * (cells + 4096)
    blob = np.empty(cap, np.uint8)
    off = np.zeros(nt + 1, np.uint64)
    used = np.zeros(nt, np.uint8)
    st = np.zeros(nt, np.int32)

    def timed(fn, reps=2):
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    t = timed(lambda: check(L.gf_codec_master_encode_batch_i32(ctx.handle, cd, len(codecs), n_rows, n_cols, nt, _p(vals), _p(blob), cap,
                                                               _p(off), _p(used), _p(st)), "codec_master_encode"))
    total = int(off[nt])
    u, c = np.unique(used, return_counts=True)
    out["codec_master_encode"] = {"seconds": round(t, 4), "GBps": round(gb / t, 3), "bytes_per_cell": round(total / (nt * cells), 4),
                                  "winners": {int(a): int(n) for a, n in zip(u, c)}}
    back = np.empty_like(vals)
    t = timed(lambda: check(L.gf_codec_master_decode_batch_i32(ctx.handle, cd, len(codecs), n_rows, n_cols, nt, _p(blob), _p(off), _p(back),
                                                               _p(st)), "codec_master_decode"))
    assert (st == 0).all() and np.array_equal(back, vals)
    out["codec_master_decode"] = {"seconds": round(t, 4), "GBps": round(gb / t, 3)}

    # tile records (RecordManager.writeTile framing)
    rcap = nt * int(L.gf_tile_record_max_bytes(0, n_rows, n_cols))
    rblob = np.empty(rcap, np.uint8)
    roff = np.zeros(nt + 1, np.uint64)
    idx = np.arange(nt, dtype=np.int32)
    t = timed(lambda: check(L.gf_tile_record_encode_batch(ctx.handle, cd, len(codecs), 0, -2 ** 31, n_rows, n_cols, nt, _p(idx), _p(vals), 1,
                                                          _p(rblob), rcap, _p(roff), _p(used)), "tile_record_encode"))
    out["tile_record_encode"] = {"seconds": round(t, 4), "GBps": round(gb / t, 3), "record_bytes_per_cell": round(int(roff[nt]) / (nt * cells), 4)}
    ridx = np.zeros(nt, np.int32)
    t = timed(lambda: check(L.gf_tile_record_decode_batch(ctx.handle, cd, len(codecs), 0, n_rows, n_cols, nt, _p(rblob), _p(roff), 1, _p(ridx),
                                                          _p(back), _p(st)), "tile_record_decode"))
    assert (st == 0).all() and np.array_equal(back, vals) and np.array_equal(ridx, idx)
    out["tile_record_decode"] = {"seconds": round(t, 4), "GBps": round(gb / t, 3)}

    # the single codecs through their host entry points (what the list is made of)
    for name in ("huffman", "canon", "deflate"):
        fn = getattr(L, "gf_%s_encode_batch_i32" % name)
        t = timed(lambda: check(fn(ctx.handle, 0, n_rows, n_cols, nt, _p(vals), _p(blob), cap, _p(off), _p(used), _p(st)), name))
        out["%s_encode_host_path" % name] = {"seconds": round(t, 4), "GBps": round(gb / t, 3), "bytes_per_cell": round(int(off[nt]) / (nt * cells), 4)}

    # the bound: zlib level 6 alone on the M32 streams of the Deflate candidates (three per tile without nulls), all threads
    import oracle
    sample = min(nt, 256)
    streams = []
    for ti in range(sample):
        for p in (1, 2, 3):
            streams.append(bytes(oracle.predictor_encode(p, n_rows, n_cols, vals[ti])[0]))
    nthreads = len(os.sched_getaffinity(0))

    def work(chunk):
        return sum(len(zlib.compress(s, 6)) for s in chunk)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthreads) as ex:
        list(ex.map(work, [streams[i::nthreads] for i in range(nthreads)]))
    tz = time.perf_counter() - t0
    t0 = time.perf_counter()
    work(streams[:48])
    t1 = time.perf_counter() - t0
    m32_bytes = sum(map(len, streams))
    out["zlib_alone"] = {"sample_tiles": sample, "m32_bytes_per_tile": round(m32_bytes / sample), "level": 6,
                         "one_thread_MBps_of_m32": round(sum(map(len, streams[:48])) / t1 / 1e6, 1),
                         "all_threads_MBps_of_m32": round(m32_bytes / tz / 1e6, 1),
                         "bound_GBps_of_cells": round(sample * cells * 4 / tz / 1e9, 3),
                         "note": "three candidate streams per tile (CodecDeflate.java:176-199) deflated by zlib level 6 on every thread "
                                 "of the box: no implementation that reproduces zlib's bytes encodes the list faster than this"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
