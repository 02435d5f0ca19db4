"""ctypes front end of the CPU oracle (oracle/gvrs_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from gridfour_amd/.  See gvrs_oracle.h for
how the restatement is pinned against the reference's fixtures.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgvrs_oracle.so")

OK, DECLINED = 0, 1
ERR_FORMAT, ERR_BOUNDS, ERR_CAPACITY, ERR_ARG = -1, -2, -3, -4
INT4_NULL = -(2 ** 31)
PM_DIFFERENCING, PM_LINEAR, PM_TRIANGLE, PM_DIFFERENCING_NULLS = 1, 2, 3, 4


def build(force=False):
    """Compile libgvrs_oracle.so with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(f) for f in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libgvrs_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            build()
        except Exception:
            if not os.path.exists(_SO):
                raise
        L = C.CDLL(_SO)
        u8p, i32p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
        szp, ip = C.POINTER(C.c_size_t), C.POINTER(C.c_int)
        L.gvo_m32_encode.argtypes = [C.c_int32, u8p]
        L.gvo_m32_decode.argtypes = [u8p, szp]
        L.gvo_m32_decode.restype = C.c_int32
        L.gvo_predictor_encode.argtypes = [C.c_int, C.c_int, C.c_int, i32p, u8p, i32p]
        L.gvo_predictor_decode.argtypes = [C.c_int, C.c_int32, C.c_int, C.c_int, u8p, C.c_size_t, i32p]
        L.gvo_huffman_encode.argtypes = [u8p, C.c_size_t, szp, u8p, C.c_size_t, u8p, szp]
        L.gvo_huffman_decode.argtypes = [u8p, C.c_size_t, szp, u8p, C.c_size_t]
        L.gvo_codec_huffman_encode.argtypes = [C.c_int, C.c_int, C.c_int, i32p, u8p, C.c_size_t,
                                               szp, C.c_int, ip]
        L.gvo_codec_huffman_decode.argtypes = [C.c_int, C.c_int, u8p, C.c_size_t, i32p]
        L.gvo_codec_huffman_bound.argtypes = [C.c_size_t]
        L.gvo_codec_huffman_bound.restype = C.c_size_t
        L.gvo_codec_deflate_encode.argtypes = [C.c_int, C.c_int, C.c_int, i32p, u8p, C.c_size_t, szp, ip]
        L.gvo_codec_deflate_decode.argtypes = [C.c_int, C.c_int, u8p, C.c_size_t, i32p]
        L.gvo_float_planes_encode.argtypes = [C.c_int, C.c_int, u32p, u8p]
        L.gvo_float_planes_decode.argtypes = [C.c_int, C.c_int, u8p, u32p]
        L.gvo_codec_float_encode.argtypes = [C.c_int, C.c_int, C.c_int, u32p, C.c_int, u8p,
                                             C.c_size_t, szp]
        L.gvo_codec_float_decode.argtypes = [C.c_int, C.c_int, u8p, C.c_size_t, u32p]
        L.gvo_batch_huffman_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, i32p, u8p,
                                               C.c_size_t, u32p, u8p]
        L.gvo_batch_huffman_decode.argtypes = [C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t,
                                               u32p, i32p]
        L.gvo_huffman_roundtrip_threads.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, i32p, u8p, C.c_size_t, u32p,
                                                    i32p, C.POINTER(C.c_double)]
        L.gvo_splitmix64.argtypes = [C.c_uint64]
        L.gvo_splitmix64.restype = C.c_uint64
        L.gvo_dem_value.argtypes = [C.c_uint64, C.c_int64, C.c_int64]
        L.gvo_dem_value.restype = C.c_int32
        L.gvo_dem_fill_tiles.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                         C.c_int64, i32p]
        L.gvo_dem_fill_tiles.restype = None
        L.gvo_dem_fill_tiles_masked.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, i32p]
        L.gvo_dem_fill_tiles_masked.restype = None
        L.gvo_dem_fill_tiles_style.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, i32p]
        L.gvo_dem_fill_tiles_style.restype = None
        f32p = C.POINTER(C.c_float)
        L.gvo_canon_encode.argtypes = [u8p, C.c_size_t, szp, i32p, C.c_size_t, u8p]
        L.gvo_canon_decode.argtypes = [u8p, C.c_size_t, szp, i32p, C.c_size_t, szp]
        L.gvo_predictor_encode_int.argtypes = [C.c_int, C.c_int, C.c_int, i32p, i32p, i32p]
        L.gvo_predictor_decode_int.argtypes = [C.c_int, C.c_int32, C.c_int, C.c_int, i32p, i32p]
        L.gvo_codec_canon_encode.argtypes = [C.c_int, C.c_int, C.c_int, i32p, u8p, C.c_size_t, szp, C.c_int, ip]
        L.gvo_codec_canon_decode.argtypes = [C.c_int, C.c_int, u8p, C.c_size_t, i32p]
        L.gvo_codec_canon_bound.argtypes = [C.c_size_t]
        L.gvo_codec_canon_bound.restype = C.c_size_t
        L.gvo_batch_canon_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, i32p, u8p, C.c_size_t, u32p, u8p]
        L.gvo_batch_canon_decode.argtypes = [C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t, u32p, i32p]
        L.gvo_lsop12_coefficients.argtypes = [C.c_int, C.c_int, i32p, f32p]
        L.gvo_lsop12_residuals.argtypes = [C.c_int, C.c_int, i32p, i32p, f32p, i32p, i32p]
        L.gvo_lsop12_encode.argtypes = [C.c_int, C.c_int, C.c_int, i32p, C.c_int, u8p, C.c_size_t, szp, ip]
        L.gvo_lsop12_decode.argtypes = [C.c_int, C.c_int, u8p, C.c_size_t, i32p]
        L.gvo_lsop12_encode_legacy_huffman.argtypes = [C.c_int, C.c_int, C.c_int, i32p, u8p, C.c_size_t, szp]
        L.gvo_lsop12_encode_ex.argtypes = [C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, u8p, C.c_size_t, szp, ip]
        L.gvo_crc32c.argtypes = [u8p, C.c_size_t]
        L.gvo_crc32c.restype = C.c_uint32
        L.gvo_lsop_value_checksum.argtypes = [C.c_int, C.c_int, i32p]
        L.gvo_lsop_value_checksum.restype = C.c_uint32
        L.gvo_lsop12_bound.argtypes = [C.c_size_t]
        L.gvo_lsop12_bound.restype = C.c_size_t
        L.gvo_batch_lsop12_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, i32p, C.c_int, u8p, C.c_size_t,
                                              u32p, u8p]
        L.gvo_batch_lsop12_decode.argtypes = [C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t, u32p, i32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    if isinstance(a, (bytes, bytearray)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


# ---- M32 ----
def m32_encode(value):
    out = np.zeros(8, np.uint8)
    n = lib().gvo_m32_encode(int(value), _p(out, C.c_uint8))
    return bytes(out[:n])


def m32_encode_seq(values):
    return b"".join(m32_encode(v) for v in values)


def m32_decode_seq(data, count):
    buf = _u8(data)
    buf = np.concatenate([buf, np.zeros(8, np.uint8)])
    pos = C.c_size_t(0)
    out = []
    for _ in range(count):
        out.append(lib().gvo_m32_decode(_p(buf, C.c_uint8), C.byref(pos)))
    return out, pos.value


# ---- predictors ----
def predictor_encode(model, n_rows, n_cols, values):
    v = _i32(values).ravel()
    out = np.zeros(6 * v.size + 8, np.uint8)
    seed = C.c_int32(0)
    n = lib().gvo_predictor_encode(model, n_rows, n_cols, _p(v, C.c_int32), _p(out, C.c_uint8),
                                   C.byref(seed))
    if n < 0:
        return None, 0
    return bytes(out[:n]), seed.value


def predictor_decode(model, seed, n_rows, n_cols, m32):
    m = np.concatenate([_u8(m32), np.zeros(6 * n_rows * n_cols + 8, np.uint8)])
    out = np.zeros(n_rows * n_cols, np.int32)
    rc = lib().gvo_predictor_decode(model, seed, n_rows, n_cols, _p(m, C.c_uint8), len(m32),
                                    _p(out, C.c_int32))
    if rc != OK:
        raise ValueError("predictor_decode rc=%d" % rc)
    return out


# ---- Huffman over a bit buffer ----
def huffman_encode(symbols, bit_pos=0, prefix=b""):
    """Returns (bytes, end_bit_pos, code_lengths[256], tree_bits)."""
    s = _u8(symbols)
    cap = len(prefix) + 400 + 32 * s.size + 64
    buf = np.zeros(cap, np.uint8)
    buf[:len(prefix)] = np.frombuffer(prefix, np.uint8) if prefix else []
    pos = C.c_size_t(bit_pos)
    cl = np.zeros(256, np.uint8)
    tb = C.c_size_t(0)
    rc = lib().gvo_huffman_encode(_p(buf, C.c_uint8), cap * 8, C.byref(pos), _p(s, C.c_uint8),
                                  s.size, _p(cl, C.c_uint8), C.byref(tb))
    if rc != OK:
        raise ValueError("huffman_encode rc=%d" % rc)
    return bytes(buf[:(pos.value + 7) // 8]), pos.value, cl, tb.value


def huffman_decode(data, n_symbols, bit_pos=0):
    b = _u8(data)
    out = np.zeros(max(n_symbols, 1), np.uint8)
    pos = C.c_size_t(bit_pos)
    rc = lib().gvo_huffman_decode(_p(b, C.c_uint8), b.size * 8, C.byref(pos), _p(out, C.c_uint8),
                                  n_symbols)
    if rc != OK:
        raise ValueError("huffman_decode rc=%d" % rc)
    return bytes(out[:n_symbols]), pos.value


# ---- CodecHuffman ----
def codec_huffman_encode(codec_index, n_rows, n_cols, values, predictor_mask=0xF):
    """Returns (packing bytes | None, predictor_used)."""
    v = _i32(values).ravel()
    assert v.size == n_rows * n_cols
    cap = 4 * v.size + 4096
    while True:
        out = np.zeros(cap, np.uint8)
        n = C.c_size_t(0)
        used = C.c_int(0)
        rc = lib().gvo_codec_huffman_encode(codec_index, n_rows, n_cols, _p(v, C.c_int32),
                                            _p(out, C.c_uint8), cap, C.byref(n), predictor_mask,
                                            C.byref(used))
        if rc == ERR_CAPACITY:
            cap = n.value + 16
            continue
        break
    if rc == DECLINED:
        return None, 0
    if rc != OK:
        raise ValueError("codec_huffman_encode rc=%d" % rc)
    return bytes(out[:n.value]), used.value


def codec_huffman_decode(n_rows, n_cols, packing):
    p = _u8(packing)
    out = np.zeros(n_rows * n_cols, np.int32)
    rc = lib().gvo_codec_huffman_decode(n_rows, n_cols, _p(p, C.c_uint8), p.size, _p(out, C.c_int32))
    if rc != OK:
        raise IOError("codec_huffman_decode rc=%d" % rc)
    return out


# ---- CodecDeflate ----
def codec_deflate_encode(codec_index, n_rows, n_cols, values):
    v = _i32(values).ravel()
    cap = 6 * v.size + 256
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    used = C.c_int(0)
    rc = lib().gvo_codec_deflate_encode(codec_index, n_rows, n_cols, _p(v, C.c_int32),
                                        _p(out, C.c_uint8), cap, C.byref(n), C.byref(used))
    if rc == DECLINED:
        return None, 0
    if rc != OK:
        raise ValueError("codec_deflate_encode rc=%d" % rc)
    return bytes(out[:n.value]), used.value


def codec_deflate_decode(n_rows, n_cols, packing):
    p = _u8(packing)
    out = np.zeros(n_rows * n_cols, np.int32)
    rc = lib().gvo_codec_deflate_decode(n_rows, n_cols, _p(p, C.c_uint8), p.size, _p(out, C.c_int32))
    if rc != OK:
        raise IOError("codec_deflate_decode rc=%d" % rc)
    return out


# ---- CodecFloat ----
def float_planes_encode(n_rows, n_cols, raw_bits):
    c = np.ascontiguousarray(raw_bits, dtype=np.uint32).ravel()
    n = c.size
    planes = np.zeros((n + 7) // 8 + 4 * n, np.uint8)
    lib().gvo_float_planes_encode(n_rows, n_cols, _p(c, C.c_uint32), _p(planes, C.c_uint8))
    return planes


def float_planes_decode(n_rows, n_cols, planes):
    p = _u8(planes)
    out = np.zeros(n_rows * n_cols, np.uint32)
    lib().gvo_float_planes_decode(n_rows, n_cols, _p(p, C.c_uint8), _p(out, C.c_uint32))
    return out


def codec_float_encode(codec_index, n_rows, n_cols, raw_bits, level=9):
    c = np.ascontiguousarray(raw_bits, dtype=np.uint32).ravel()
    cap = 5 * c.size + 1024
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    rc = lib().gvo_codec_float_encode(codec_index, n_rows, n_cols, _p(c, C.c_uint32), level,
                                      _p(out, C.c_uint8), cap, C.byref(n))
    if rc != OK:
        raise ValueError("codec_float_encode rc=%d" % rc)
    return bytes(out[:n.value])


def codec_float_decode(n_rows, n_cols, packing):
    p = _u8(packing)
    out = np.zeros(n_rows * n_cols, np.uint32)
    rc = lib().gvo_codec_float_decode(n_rows, n_cols, _p(p, C.c_uint8), p.size, _p(out, C.c_uint32))
    if rc != OK:
        raise IOError("codec_float_decode rc=%d" % rc)
    return out


# ---- batch (CPU baseline) ----
def batch_huffman_encode(codec_index, n_rows, n_cols, tiles, stride=None):
    """tiles: int32 [n_tiles, n_rows*n_cols].  Returns (out[n_tiles, stride], lengths, predictors)."""
    v = _i32(tiles).reshape(-1, n_rows * n_cols)
    nt = v.shape[0]
    if stride is None:
        stride = 4 * n_rows * n_cols + 4096
    out = np.zeros((nt, stride), np.uint8)
    lengths = np.zeros(nt, np.uint32)
    preds = np.zeros(nt, np.uint8)
    rc = lib().gvo_batch_huffman_encode(codec_index, n_rows, n_cols, nt, _p(v, C.c_int32),
                                        _p(out, C.c_uint8), stride, _p(lengths, C.c_uint32),
                                        _p(preds, C.c_uint8))
    if rc != OK:
        raise ValueError("batch_huffman_encode rc=%d" % rc)
    return out, lengths, preds


def batch_huffman_decode(n_rows, n_cols, packings, lengths):
    p = _u8(packings)
    nt, stride = p.shape
    ln = np.ascontiguousarray(lengths, dtype=np.uint32)
    out = np.zeros((nt, n_rows * n_cols), np.int32)
    rc = lib().gvo_batch_huffman_decode(n_rows, n_cols, nt, _p(p, C.c_uint8), stride,
                                        _p(ln, C.c_uint32), _p(out, C.c_int32))
    if rc != OK:
        raise IOError("batch_huffman_decode rc=%d" % rc)
    return out


def huffman_roundtrip_threads(n_threads, codec_index, n_rows, n_cols, tiles):
    """CodecHuffman encode + decode of every tile on n_threads NATIVE threads (pthreads inside the oracle, one contiguous
    share of the tiles each).  Returns (encode seconds, decode seconds); raises if a round trip fails."""
    v = _i32(tiles).reshape(-1, n_rows * n_cols)
    nt = v.shape[0]
    stride = 4 * n_rows * n_cols + 4096
    out = np.empty((nt, stride), np.uint8)
    lengths = np.zeros(nt, np.uint32)
    back = np.empty_like(v)
    out[:, ::4096] = 0                             # touch every page first: the timed threads must not queue on page faults
    back[:] = 0
    sec = (C.c_double * 2)()
    rc = lib().gvo_huffman_roundtrip_threads(int(n_threads), codec_index, n_rows, n_cols, nt, _p(v, C.c_int32), _p(out, C.c_uint8),
                                             stride, _p(lengths, C.c_uint32), _p(back, C.c_int32), sec)
    if rc != OK:
        raise ValueError("huffman_roundtrip_threads rc=%d" % rc)
    return float(sec[0]), float(sec[1])


# ---- synthetic DEM ----
DEM_SEED = 0x9E3779B97F4A7C15


DEM_STYLE_ROUGH = 1


def dem_tiles(seed, n_rows, n_cols, tiles_per_row, tile0, n_tiles, mask_per_mille=0, style=0):
    """mask_per_mille > 0: the nulls workload -- that share of the grid's 16 x 16 blocks holds the null code (ocean mask).
    style 1: the rough surface (provinces of mountains / plains / stripes, cliff blocks: gvrs_oracle.c)."""
    out = np.zeros((n_tiles, n_rows * n_cols), np.int32)
    if style:
        lib().gvo_dem_fill_tiles_style(seed & (2 ** 64 - 1), n_rows, n_cols, tiles_per_row, tile0, n_tiles, mask_per_mille, style,
                                       _p(out, C.c_int32))
    elif mask_per_mille:
        lib().gvo_dem_fill_tiles_masked(seed & (2 ** 64 - 1), n_rows, n_cols, tiles_per_row, tile0, n_tiles, mask_per_mille,
                                        _p(out, C.c_int32))
    else:
        lib().gvo_dem_fill_tiles(seed & (2 ** 64 - 1), n_rows, n_cols, tiles_per_row, tile0, n_tiles,
                                 _p(out, C.c_int32))
    return out


# ---- canonical Huffman / CodecCanonHuffman (parity unpinned: see gvrs_oracle_canon.c) ----
def canon_encode(text, bit_pos=0, prefix=b""):
    """CanonicalHuffman.encode appended at bit_pos; returns (bytes, end_bit_pos, code_lengths[260])."""
    t = _i32(text).ravel()
    cap = len(prefix) + 2048 + 11 * t.size + 64
    buf = np.zeros(cap, np.uint8)
    if prefix:
        buf[:len(prefix)] = np.frombuffer(prefix, np.uint8)
    pos = C.c_size_t(bit_pos)
    cl = np.zeros(260, np.uint8)
    rc = lib().gvo_canon_encode(_p(buf, C.c_uint8), cap * 8, C.byref(pos), _p(t, C.c_int32), t.size,
                                _p(cl, C.c_uint8))
    if rc != OK:
        raise ValueError("canon_encode rc=%d" % rc)
    return bytes(buf[:(pos.value + 7) // 8]), pos.value, cl


def canon_decode(data, max_symbols, bit_pos=0):
    b = _u8(data)
    out = np.zeros(max(max_symbols, 1), np.int32)
    pos = C.c_size_t(bit_pos)
    nd = C.c_size_t(0)
    rc = lib().gvo_canon_decode(_p(b, C.c_uint8), b.size * 8, C.byref(pos), _p(out, C.c_int32), max_symbols,
                                C.byref(nd))
    if rc != OK:
        raise ValueError("canon_decode rc=%d" % rc)
    return out[:nd.value].copy(), pos.value


def predictor_encode_int(model, n_rows, n_cols, values):
    v = _i32(values).ravel()
    out = np.zeros(v.size + 8, np.int32)
    seed = C.c_int32(0)
    n = lib().gvo_predictor_encode_int(model, n_rows, n_cols, _p(v, C.c_int32), _p(out, C.c_int32), C.byref(seed))
    if n < 0:
        return None, 0
    return out[:n].copy(), seed.value


def predictor_decode_int(model, seed, n_rows, n_cols, residuals):
    e = np.concatenate([_i32(residuals).ravel(), np.zeros(n_rows * n_cols + 8, np.int32)])
    out = np.zeros(n_rows * n_cols, np.int32)
    rc = lib().gvo_predictor_decode_int(model, seed, n_rows, n_cols, _p(e, C.c_int32), _p(out, C.c_int32))
    if rc != OK:
        raise ValueError("predictor_decode_int rc=%d" % rc)
    return out


def codec_canon_encode(codec_index, n_rows, n_cols, values, predictor_mask=0xF):
    """CodecCanonHuffman.encode: (packing | None, predictor_used); ValueError where Java throws."""
    v = _i32(values).ravel()
    assert v.size == n_rows * n_cols
    cap = int(lib().gvo_codec_canon_bound(v.size))
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    used = C.c_int(0)
    rc = lib().gvo_codec_canon_encode(codec_index, n_rows, n_cols, _p(v, C.c_int32), _p(out, C.c_uint8), cap,
                                      C.byref(n), predictor_mask, C.byref(used))
    if rc == DECLINED:
        return None, 0
    if rc != OK:
        raise ValueError("codec_canon_encode rc=%d" % rc)
    return bytes(out[:n.value]), used.value


def codec_canon_decode(n_rows, n_cols, packing):
    p = _u8(packing)
    out = np.zeros(n_rows * n_cols, np.int32)
    rc = lib().gvo_codec_canon_decode(n_rows, n_cols, _p(p, C.c_uint8), p.size, _p(out, C.c_int32))
    if rc != OK:
        raise IOError("codec_canon_decode rc=%d" % rc)
    return out


def batch_canon_encode(codec_index, n_rows, n_cols, tiles, stride=None):
    v = _i32(tiles).reshape(-1, n_rows * n_cols)
    nt = v.shape[0]
    stride = stride or (4 * n_rows * n_cols + 4096)
    out = np.zeros(nt * stride, np.uint8)
    ln = np.zeros(nt, np.uint32)
    pr = np.zeros(nt, np.uint8)
    rc = lib().gvo_batch_canon_encode(codec_index, n_rows, n_cols, nt, _p(v, C.c_int32), _p(out, C.c_uint8), stride,
                                      _p(ln, C.c_uint32), _p(pr, C.c_uint8))
    if rc != OK:
        raise ValueError("batch_canon_encode rc=%d" % rc)
    return out.reshape(nt, stride), ln, pr


def batch_canon_decode(n_rows, n_cols, slots, lengths):
    s = np.ascontiguousarray(slots, np.uint8)
    nt, stride = s.shape
    ln = np.ascontiguousarray(lengths, np.uint32)
    out = np.zeros((nt, n_rows * n_cols), np.int32)
    rc = lib().gvo_batch_canon_decode(n_rows, n_cols, nt, _p(s, C.c_uint8), stride, _p(ln, C.c_uint32),
                                      _p(out, C.c_int32))
    if rc != OK:
        raise IOError("batch_canon_decode rc=%d" % rc)
    return out


# ---- LSOP12 ----
def lsop12_coefficients(n_rows, n_cols, values):
    v = _i32(values).ravel()
    u = np.zeros(12, np.float32)
    rc = lib().gvo_lsop12_coefficients(n_rows, n_cols, _p(v, C.c_int32), _p(u, C.c_float))
    return None if rc != OK else u


def lsop12_residuals(n_rows, n_cols, values):
    v = _i32(values).ravel()
    u = np.zeros(12, np.float32)
    init = np.zeros(4 * n_rows + 2 * n_cols - 9, np.int32)
    inter = np.zeros((n_rows - 2) * (n_cols - 4), np.int32)
    seed = C.c_int32(0)
    rc = lib().gvo_lsop12_residuals(n_rows, n_cols, _p(v, C.c_int32), C.byref(seed), _p(u, C.c_float),
                                    _p(init, C.c_int32), _p(inter, C.c_int32))
    if rc != OK:
        return None
    return seed.value, u, init, inter


def crc32c(data):
    """util/GridfourCRC32C: update(data), getValue()."""
    b = _u8(data)
    return int(lib().gvo_crc32c(_p(b, C.c_uint8), b.size))


def lsop_value_checksum(n_rows, n_cols, values):
    """LsHeader.computeChecksum."""
    v = _i32(values).ravel()
    return int(lib().gvo_lsop_value_checksum(n_rows, n_cols, _p(v, C.c_int32)))


def lsop12_encode(codec_index, n_rows, n_cols, values, deflate_enabled=True, value_checksum=False):
    """LsEncoder12.encode: (packing | None, container type).  value_checksum: setValueChecksumEnabled."""
    v = _i32(values).ravel()
    cap = int(lib().gvo_lsop12_bound(v.size))
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    typ = C.c_int(0)
    rc = lib().gvo_lsop12_encode_ex(codec_index, n_rows, n_cols, _p(v, C.c_int32), int(deflate_enabled), int(value_checksum),
                                    _p(out, C.c_uint8), cap, C.byref(n), C.byref(typ))
    if rc == DECLINED:
        return None, 0
    if rc != OK:
        raise ValueError("lsop12_encode rc=%d" % rc)
    return bytes(out[:n.value]), typ.value


def lsop12_encode_legacy_huffman(codec_index, n_rows, n_cols, values):
    v = _i32(values).ravel()
    cap = 8 * v.size + 4096
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    rc = lib().gvo_lsop12_encode_legacy_huffman(codec_index, n_rows, n_cols, _p(v, C.c_int32), _p(out, C.c_uint8), cap,
                                                C.byref(n))
    if rc != OK:
        raise ValueError("lsop12_encode_legacy_huffman rc=%d" % rc)
    return bytes(out[:n.value])


def lsop12_decode(n_rows, n_cols, packing):
    p = _u8(packing)
    out = np.zeros(n_rows * n_cols, np.int32)
    rc = lib().gvo_lsop12_decode(n_rows, n_cols, _p(p, C.c_uint8), p.size, _p(out, C.c_int32))
    if rc != OK:
        raise IOError("lsop12_decode rc=%d" % rc)
    return out


def batch_lsop12_encode(codec_index, n_rows, n_cols, tiles, deflate_enabled=False, stride=None):
    v = _i32(tiles).reshape(-1, n_rows * n_cols)
    nt = v.shape[0]
    stride = stride or (4 * n_rows * n_cols + 4096)
    out = np.zeros(nt * stride, np.uint8)
    ln = np.zeros(nt, np.uint32)
    ty = np.zeros(nt, np.uint8)
    rc = lib().gvo_batch_lsop12_encode(codec_index, n_rows, n_cols, nt, _p(v, C.c_int32), int(deflate_enabled),
                                       _p(out, C.c_uint8), stride, _p(ln, C.c_uint32), _p(ty, C.c_uint8))
    if rc != OK:
        raise ValueError("batch_lsop12_encode rc=%d" % rc)
    return out.reshape(nt, stride), ln


def batch_lsop12_decode(n_rows, n_cols, slots, lengths):
    s = np.ascontiguousarray(slots, np.uint8)
    nt, stride = s.shape
    ln = np.ascontiguousarray(lengths, np.uint32)
    out = np.zeros((nt, n_rows * n_cols), np.int32)
    rc = lib().gvo_batch_lsop12_decode(n_rows, n_cols, nt, _p(s, C.c_uint8), stride, _p(ln, C.c_uint32),
                                       _p(out, C.c_int32))
    if rc != OK:
        raise IOError("batch_lsop12_decode rc=%d" % rc)
    return out
