"""gridfour_amd -- MI355X-native GVRS tile codec (hand-written HIP for gfx950).

Scope: the per-tile compression hot path of gwlucastrig/gridfour (predictor -> CodecM32 ->
CodecHuffman), bit-exact with the reference Java codec, behind the reference's own
ICompressionEncoder / ICompressionDecoder plug-in interface.  The product is the C-ABI library
`lib/libgvrs_hip.so` (include/gvrs_hip_codec.h); this package is the thin host-side mirror of
the reference interface used by tests, bench.py and Python callers.
"""
from ._lib import (GvrsHipError, OK, DECLINED, OVERFLOW, ERR_FORMAT, ERR_BOUNDS, ERR_CAPACITY,  # noqa: F401
                   ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, PM_ALL, lib, lib_path)
from .codec import (CodecMasterHip, STANDARD_CODEC_LIST, CodecHuffmanHip, CodecDeflateHip, CodecCanonHuffmanHip, CodecFloatHip, LsCodecHip, GvrsHipContext, DeviceBuffer, DeviceTileBatch, GpuTimer,  # noqa: F401
                    INT4_NULL_CODE)
from .sharding import shard_range, GvrsHipMulti, PinnedArray, TileReadAhead  # noqa: F401

__all__ = ["CodecHuffmanHip", "CodecCanonHuffmanHip", "CodecDeflateHip", "CodecFloatHip", "LsCodecHip", "GvrsHipContext", "GvrsHipError", "INT4_NULL_CODE", "lib", "lib_path",
           "shard_range", "GvrsHipMulti", "PinnedArray", "TileReadAhead"]
