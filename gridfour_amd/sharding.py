"""Sharding of tile batches over the GPUs of a node.

Tiles are encoded/decoded with no reference to any other tile (reference:
gvrs/RasterTile.java:237-241 loops elements independently; codecs take only
(nRows, nCols, values)), so a batch shards over the GPUs of a node as contiguous tile-index
ranges with NO data-path collective: shard g of G gets tiles [g*T/G, (g+1)*T/G).

Two ways to use several GPUs, both on this partition:
  * one process per GPU (bench.py under torchrun): `shard_range` gives each rank its range;
  * one process for the node (what a JVM would do): `GvrsHipMulti` wraps the C ABI's gf_multi_* entry points --
    one context and one host thread per device inside libgvrs_hip.so, packings concatenated by an exclusive scan.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


def shard_range(n_tiles, rank, world_size):
    """Returns (first_tile, n_local) of the contiguous range owned by `rank`."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    lo = (n_tiles * rank) // world_size
    hi = (n_tiles * (rank + 1)) // world_size
    return lo, hi - lo


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class PinnedArray:
    """Page-locked host memory from gf_host_alloc as a numpy array (moved over PCIe in place by the host batch calls)."""

    def __init__(self, shape, dtype):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(np.atleast_1d(shape))
        n = int(np.prod(self.shape)) * self.dtype.itemsize
        self._p = C.c_void_p()
        check(lib().gf_host_alloc(max(n, 1), C.byref(self._p)), "gf_host_alloc")
        buf = (C.c_uint8 * max(n, 1)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def close(self):
        if self._p:
            self.array = None
            lib().gf_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GvrsHipMulti:
    """gf_multi: one context per listed device (a device may be listed more than once), batches sharded by tile range."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = C.c_void_p()
        check(lib().gf_multi_create(devs, len(devices), C.byref(self._h)), "gf_multi_create")
        self.devices = [int(d) for d in devices]

    @property
    def handle(self):
        return self._h

    def __len__(self):
        return int(lib().gf_multi_count(self._h))

    def partition(self, n_tiles, i):
        t0, t1 = C.c_size_t(0), C.c_size_t(0)
        lib().gf_multi_partition(n_tiles, len(self), i, C.byref(t0), C.byref(t1))
        return t0.value, t1.value

    def synchronize(self):
        check(lib().gf_multi_synchronize(self._h), "gf_multi_synchronize")

    def encode_batch(self, codecIndex, nRows, nCols, tiles, codec="huffman", deflate_enabled=True, level=9):
        """Host memory in, (blob uint8, offsets uint64[n+1], predictors, status) out -- the arrays of the C call.
        codec: huffman, canon, deflate, lsop12 (predictors = container types; deflate_enabled as LsEncoder12), float (float32
        cells, zlib level; no per-tile byte, no status: those come back as zeros)."""
        cells = nRows * nCols
        is_float = codec == "float"
        v = np.ascontiguousarray(tiles, dtype=np.float32 if is_float else np.int32).reshape(-1, cells)
        nt = v.shape[0]
        name = "gf_float_encode_batch_f32_multi" if is_float else "gf_%s_encode_batch_i32_multi" % codec
        fn = getattr(lib(), name)
        if is_float:
            cap = nt * (5 * cells + 4096)
        elif codec == "lsop12":
            cap = nt * (int(lib().gf_lsop12_max_packing(nRows, nCols)) // 2 + 256) + 4096
        else:
            cap = nt * int(lib().gf_huffman_default_stride(nRows, nCols)) // 2 + 4096
        offsets = np.zeros(nt + 1, np.uint64)
        preds = np.zeros(nt, np.uint8)
        status = np.zeros(nt, np.int32)
        for attempt in range(2):                              # one regrow at most: a second GF_ERR_CAPACITY is an error
            blob = np.empty(cap, np.uint8)
            if is_float:
                st = fn(self._h, codecIndex, nRows, nCols, nt, _ptr(v), int(level), _ptr(blob), cap, _ptr(offsets))
            elif codec == "lsop12":
                st = fn(self._h, codecIndex, nRows, nCols, nt, _ptr(v), 1 if deflate_enabled else 0, _ptr(blob), cap, _ptr(offsets),
                        _ptr(preds), _ptr(status))
            else:
                st = fn(self._h, codecIndex, nRows, nCols, nt, _ptr(v), _ptr(blob), cap, _ptr(offsets), _ptr(preds), _ptr(status))
            if st == _lib.ERR_CAPACITY and attempt == 0 and int(offsets[nt]) > 0:
                cap = int(offsets[nt]) + 64
                continue
            check(st, name)
            return blob[:int(offsets[nt])], offsets, preds, status

    def decode_batch(self, nRows, nCols, blob, offsets, codec="huffman"):
        nt = len(offsets) - 1
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        if b.size < int(offsets[nt]) + 16:
            b = np.concatenate([b, np.zeros(16, np.uint8)])
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        is_float = codec == "float"
        out = np.empty((nt, nRows * nCols), np.float32 if is_float else np.int32)
        status = np.zeros(nt, np.int32)
        name = "gf_float_decode_batch_f32_multi" if is_float else "gf_%s_decode_batch_i32_multi" % codec
        check(getattr(lib(), name)(self._h, nRows, nCols, nt, _ptr(b), _ptr(off), _ptr(out), _ptr(status)), name)
        return out, status

    def close(self):
        if self._h:
            lib().gf_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TileReadAhead:
    """gf_readahead: the tile cache's reading assistant (gvrs/TileDecompressionAssistant.java) as an N-tile prefetch queue.
    submit() copies a tile's element bytes and returns; a background thread inside the library decodes whatever is queued as
    one GPU batch; take() is getTilesWithWaitForIndex."""

    def __init__(self, n_rows, n_cols, codecs=(1, 2, 0, 3), device=0, max_batch=1024):
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.cells = self.n_rows * self.n_cols
        cd = (C.c_int * len(codecs))(*[int(c) for c in codecs])
        self._h = C.c_void_p()
        check(lib().gf_readahead_create(int(device), cd, len(codecs), self.n_rows, self.n_cols, int(max_batch), C.byref(self._h)),
              "gf_readahead_create")

    def submit(self, tile_index, packing):
        b = np.frombuffer(bytes(packing), dtype=np.uint8)
        check(lib().gf_readahead_submit(self._h, int(tile_index), _ptr(b) if b.size else None, b.size), "gf_readahead_submit")

    def pending(self):
        return int(lib().gf_readahead_pending(self._h))

    def take(self, wait_index, max_tiles=64):
        """Returns {tile_index: (values int32[cells], status)} of up to max_tiles finished tiles, wait_index first."""
        idx = np.zeros(max_tiles, np.int32)
        st = np.zeros(max_tiles, np.int32)
        vals = np.empty((max_tiles, self.cells), np.int32)
        n = C.c_size_t(0)
        check(lib().gf_readahead_take(self._h, int(wait_index), max_tiles, _ptr(idx), _ptr(vals), _ptr(st), C.byref(n)),
              "gf_readahead_take")
        return {int(idx[i]): (vals[i].copy(), int(st[i])) for i in range(n.value)}

    def counters(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        lib().gf_readahead_counters(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def close(self):
        if self._h:
            lib().gf_readahead_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
