"""ctypes binding of libgvrs_hip.so (the C ABI in include/gvrs_hip_codec.h).

There is deliberately no fallback: if the library cannot be loaded the import fails, and
if no HIP device is present every compute call fails with GF_ERR_NO_DEVICE.
"""
import ctypes as C
import os

from . import build as _build

OK, DECLINED, OVERFLOW = 0, 1, 2
ERR_FORMAT, ERR_BOUNDS, ERR_CAPACITY, ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7
PM_ALL = 0xF
LSOP_DEFLATE, LSOP_VALUE_CHECKSUM = 1, 2          # LsEncoder12.setDeflateEnabled / setValueChecksumEnabled (include/gvrs_hip_codec.h)

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p

# name -> (restype, argtypes); this table is also what tests/test_abi_symbols.py checks
# against include/gvrs_hip_codec.h
SIGNATURES = {
    "gf_version": (C.c_char_p, []),
    "gf_status_string": (C.c_char_p, [C.c_int]),
    "gf_last_error": (C.c_char_p, []),
    "gf_device_count": (C.c_int, []),
    "gf_context_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "gf_context_destroy": (None, [_vp]),
    "gf_context_reserve": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t]),
    "gf_context_stream": (_vp, [_vp]),
    "gf_context_synchronize": (C.c_int, [_vp]),
    "gf_huffman_default_stride": (C.c_size_t, [C.c_int, C.c_int]),
    "gf_huffman_max_packing": (C.c_size_t, [C.c_int, C.c_int]),
    "gf_huffman_encode_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gf_huffman_decode_i32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "gf_huffman_encode_batch_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t,
                                              _vp, _vp, _vp]),
    "gf_huffman_decode_batch_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_huffman_encode_batch_i32_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp,
                                                  C.c_size_t, _vp, _vp, _vp, C.c_int]),
    "gf_huffman_decode_batch_i32_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_size_t, _vp,
                                                  C.c_size_t, _vp, _vp, _vp]),
    "gf_canon_max_packing": (C.c_size_t, [C.c_int, C.c_int]),
    "gf_canon_encode_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gf_canon_decode_i32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "gf_canon_encode_batch_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t,
                                            _vp, _vp, _vp]),
    "gf_canon_decode_batch_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_canon_encode_batch_i32_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp,
                                                C.c_size_t, _vp, _vp, _vp, C.c_int]),
    "gf_canon_decode_batch_i32_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_size_t, _vp,
                                                C.c_size_t, _vp, _vp, _vp]),
    "gf_lsop12_residual_count": (C.c_size_t, [C.c_int, C.c_int]),
    "gf_lsop12_max_packing": (C.c_size_t, [C.c_int, C.c_int]),
    "gf_lsop12_predict_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp]),
    "gf_lsop12_reconstruct_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_lsop12_encode_batch_i32_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t,
                                                 _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "gf_lsop12_encode_batch_i32_dev_ex": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int, _vp, C.c_size_t,
                                                    _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "gf_lsop12_decode_batch_i32_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_size_t, _vp,
                                                 C.c_size_t, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "gf_lsop12_encode_batch_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int, _vp, C.c_size_t,
                                             _vp, _vp, _vp]),
    "gf_lsop12_decode_batch_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_lsop12_encode_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_size_t,
                                       C.POINTER(C.c_size_t)]),
    "gf_lsop12_decode_i32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "gf_m32_default_stride": (C.c_size_t, [C.c_int, C.c_int]),
    "gf_m32_max_stream": (C.c_size_t, [C.c_int, C.c_int]),
    "gf_m32_encode_batch_i32_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_m32_decode_batch_i32_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp,
                                              _vp, _vp]),
    "gf_deflate_encode_batch_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "gf_deflate_decode_batch_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_deflate_encode_i32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gf_deflate_decode_i32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "gf_codec_master_encode_batch_i32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t, _vp,
                                                   _vp, _vp]),
    "gf_codec_master_decode_batch_i32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_tile_payload_encode_batch_i32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp]),
    "gf_tile_payload_decode_batch_i32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_huffman_analyze_batch": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_huffman_analyze_batch_h2": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp, _vp]),
    "gf_codec_stats_h2": (C.c_double, [_vp]),
    "gf_tile_record_max_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "gf_crc32c": (C.c_uint32, [_vp, C.c_size_t]),
    "gf_tile_record_encode_batch": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_int,
                                              _vp, C.c_size_t, _vp, _vp]),
    "gf_tile_record_decode_batch": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_int, _vp, _vp,
                                              _vp]),
    "gf_compact_dev": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, C.c_size_t]),
    "gf_float_planes_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "gf_float_planes_encode_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t]),
    "gf_float_planes_decode_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_size_t, _vp]),
    "gf_float_encode_batch_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int, _vp, C.c_size_t, _vp]),
    "gf_float_decode_batch_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_float_encode_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gf_float_decode_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "gf_synth_dem_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_size_t, _vp]),
    "gf_synth_dem_masked_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_size_t, C.c_int, _vp]),
    "gf_synth_dem_style_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_size_t, C.c_int, C.c_int, _vp]),
    "gf_dev_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "gf_dev_free": (C.c_int, [_vp, _vp]),
    "gf_dev_memset": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t]),
    "gf_dev_upload": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "gf_dev_download": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "gf_inflate_batch_dev": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gf_deflate_decode_batch_i32_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp]),
    "gf_float_decode_batch_f32_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    "gf_host_free": (C.c_int, [_vp]),
    "gf_readahead_create": (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(_vp)]),
    "gf_readahead_destroy": (None, [_vp]),
    "gf_readahead_submit": (C.c_int, [_vp, C.c_int32, _vp, C.c_size_t]),
    "gf_readahead_pending": (C.c_int, [_vp]),
    "gf_readahead_take": (C.c_int, [_vp, C.c_int32, C.c_size_t, _vp, _vp, _vp, C.POINTER(C.c_size_t)]),
    "gf_readahead_cells": (C.c_size_t, [_vp]),
    "gf_readahead_counters": (None, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gf_multi_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]),
    "gf_multi_destroy": (None, [_vp]),
    "gf_multi_count": (C.c_int, [_vp]),
    "gf_multi_context": (_vp, [_vp, C.c_int]),
    "gf_multi_device": (C.c_int, [_vp, C.c_int]),
    "gf_multi_partition": (None, [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "gf_multi_synchronize": (C.c_int, [_vp]),
    "gf_huffman_encode_batch_i32_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "gf_huffman_decode_batch_i32_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_canon_encode_batch_i32_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "gf_canon_decode_batch_i32_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_deflate_encode_batch_i32_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "gf_deflate_decode_batch_i32_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_lsop12_encode_batch_i32_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int, _vp, C.c_size_t, _vp, _vp, _vp]),
    "gf_lsop12_decode_batch_i32_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_float_encode_batch_f32_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int, _vp, C.c_size_t, _vp]),
    "gf_float_decode_batch_f32_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, _vp]),
    "gf_canon_encode_batch_i32_multi_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp,
                                                      C.c_int]),
    "gf_canon_decode_batch_i32_multi_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "gf_huffman_encode_batch_i32_multi_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp,
                                                        C.c_int]),
    "gf_huffman_decode_batch_i32_multi_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "gf_timer_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "gf_timer_destroy": (None, [_vp]),
    "gf_timer_start": (C.c_int, [_vp, _vp]),
    "gf_timer_stop": (C.c_int, [_vp, _vp]),
    "gf_timer_elapsed_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
}

_lib = None


class GvrsHipError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = lib().gf_status_string(status).decode()
        detail = lib().gf_last_error().decode()
        super().__init__("%s: %s%s" % (where, msg, (" [" + detail + "]") if detail else ""))


def _diag():
    return os.environ.get("GVRS_HIP_DIAG", "") not in ("", "0")


def lib_path():
    variant = os.environ.get("GVRS_HIP_VARIANT", "")          # tools/ only: an experiment build made by build.py --variant
    if variant:
        return _build.variant_path(variant)
    return _build.LIB_DIAG if _diag() else _build.LIB


def lib():
    """Loads libgvrs_hip.so (building it first if the sources are newer; builds are serialised by a file lock, see
    build.py).  Raises if it cannot be built or loaded -- the HIP library is the only implementation.
    GVRS_HIP_DIAG=1 (tools/ only) selects the diagnostic flavour libgvrs_hip_diag.so."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.environ.get("GVRS_HIP_VARIANT") and _build.needs_build(path):
            _build.build(diag=_diag())
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)           # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, where=""):
    if status < 0:
        raise GvrsHipError(status, where)
    return status
