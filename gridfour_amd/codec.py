"""Host-side mirror of the reference plug-in interface for the HIP codec.

`CodecHuffmanHip` carries the method names, argument meaning and error behaviour of the
reference's `ICompressionEncoder` / `ICompressionDecoder` as implemented by `CodecHuffman`
(core/src/main/java/org/gridfour/compress/ICompressionEncoder.java:61-91,
ICompressionDecoder.java:62-105, CodecHuffman.java:70-153):

  encode(codecIndex, nRows, nCols, values) -> bytes | None      (None = Java null)
  decode(nRows, nColumns, packing) -> int32 array, raises IOError (= IOException)
  encodeFloats / decodeFloats -> None, implementsIntegerEncoding() -> True, ...

plus the batched forms the GPU needs (one launch per batch of tiles, not per tile).
All compute happens in libgvrs_hip.so; nothing here has a CPU implementation.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib

INT4_NULL_CODE = -(2 ** 31)   # util/GridfourConstants.java:61


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class GvrsHipContext:
    """One device context (stream + workspace); one per process and GPU."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().gf_context_create(int(device), C.byref(self._h)), "gf_context_create")
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        return lib().gf_context_stream(self._h)

    def synchronize(self):
        check(lib().gf_context_synchronize(self._h), "gf_context_synchronize")

    def reserve(self, n_rows, n_cols, n_tiles):
        check(lib().gf_context_reserve(self._h, n_rows, n_cols, n_tiles), "gf_context_reserve")

    def close(self):
        if self._h:
            lib().gf_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# struct gf_codec_stats (include/gvrs_hip_codec.h) = the sums of compress/CodecStats.java
CODEC_STATS_DTYPE = np.dtype([("n_tiles", "<i8"), ("n_bytes", "<i8"), ("n_symbols", "<i8"), ("n_bits_overhead", "<i8"),
                              ("n_m32_counted", "<i8"), ("sum_length_m32", "<i8"), ("sum_observed_m32", "<i8"),
                              ("sum_entropy_m32", "<f8")])


class CodecHuffmanHip:
    """Drop-in for org.gridfour.compress.CodecHuffman, computed on the MI355X."""

    _PREFIX = "gf_huffman"                       # entry-point family of the C ABI

    def __init__(self, context=None, device=0):
        self.ctx = context if context is not None else GvrsHipContext(device)

    def _fn(self, name):
        return getattr(lib(), "%s_%s" % (self._PREFIX, name))

    # ---- ICompressionEncoder ----
    def encode(self, codecIndex, nRows, nCols, values):
        v = np.ascontiguousarray(values, dtype=np.int32).ravel()
        if v.size != nRows * nCols:
            raise ValueError("values.length != nRows*nCols")
        cap = int(self._fn("max_packing")(nRows, nCols))
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        st = self._fn("encode_i32")(self.ctx.handle, codecIndex, nRows, nCols, _ptr(v), _ptr(out), cap, C.byref(n))
        if st == _lib.DECLINED:
            return None
        if st == _lib.ERR_BOUNDS:
            raise IndexError("ArrayIndexOutOfBoundsException in the reference for these dimensions")
        if st == _lib.ERR_ARG and self._PREFIX == "gf_canon":
            raise ValueError("IllegalArgumentException in the reference for these dimensions")
        check(st, self._PREFIX + "_encode_i32")
        return bytes(out[:n.value])

    def encodeFloats(self, codecIndex, nRows, nCols, values):
        return None                                  # CodecHuffman.java:237-239

    def implementsFloatingPointEncoding(self):
        return False

    def implementsIntegerEncoding(self):
        return True

    # ---- ICompressionDecoder ----
    def decode(self, nRows, nColumns, packing):
        p = np.frombuffer(bytes(packing), dtype=np.uint8)
        out = np.empty(nRows * nColumns, np.int32)
        st = self._fn("decode_i32")(self.ctx.handle, nRows, nColumns, _ptr(p), p.size, _ptr(out))
        if st in (_lib.ERR_FORMAT, _lib.ERR_BOUNDS):
            raise IOError(lib().gf_status_string(st).decode())
        if st == _lib.DECLINED:                      # CodecDeflate.decode: the inflater gave nothing -> null (:143-154)
            return None
        check(st, self._PREFIX + "_decode_i32")
        return out

    def decodeFloats(self, nRows, nColumns, packing):
        return None                                  # CodecHuffman.java:242-244

    # ---- analysis (CodecHuffman.java:172-234 over CodecStats.java) ----
    _STAT_LABELS = ("None", "Differencing", "Linear", "Triangle", "DifferencingWithNulls", "All Predictors")

    def analyze(self, nRows, nColumns, packing):
        st = self.analyze_batch(nRows, nColumns, [packing])
        if st[0] != 0:
            raise IOError(lib().gf_status_string(int(st[0])).decode())

    def analyze_batch(self, nRows, nCols, packings):
        """analyze() of every packing in one GPU pass; returns the per-packing status (non-zero: analyze would throw)."""
        if self._PREFIX != "gf_huffman":
            raise NotImplementedError("statistics are gathered for CodecHuffman only")
        if getattr(self, "_stats", None) is None:
            self._stats = np.zeros(6, dtype=CODEC_STATS_DTYPE)
        nt = len(packings)
        offsets = np.zeros(nt + 1, np.uint64)
        offsets[1:] = np.cumsum([len(p) for p in packings])
        blob = np.frombuffer(b"".join(packings) + b"\0" * 16, dtype=np.uint8)
        status = np.zeros(nt, np.int32)
        if getattr(self, "_pairs", None) is None:
            self._pairs = np.zeros((6, 65536), np.int64)          # sB of the six CodecStats (CodecStats.java:64)
        check(lib().gf_huffman_analyze_batch_h2(self.ctx.handle, nRows, nCols, nt, _ptr(blob), _ptr(offsets), _ptr(self._stats),
                                                _ptr(self._pairs), _ptr(status)), "gf_huffman_analyze_batch_h2")
        return status

    def pair_counts(self):
        """sB[(prior << 8) | value] of the six CodecStats (five predictors, all)."""
        p = getattr(self, "_pairs", None)
        return None if p is None else p.copy()

    def getH2(self, k=5):
        """CodecStats.getH2 of record k (0..4 by predictor code, 5 = all predictors)."""
        p = getattr(self, "_pairs", None)
        return 0.0 if p is None else float(lib().gf_codec_stats_h2(_ptr(np.ascontiguousarray(p[k]))))

    def analysis_data(self):
        """The accumulated sums, one record per predictor code 0..4 and one for all (CodecStats fields)."""
        s = getattr(self, "_stats", None)
        return None if s is None else s.copy()

    def reportAnalysisData(self, ps, nTilesInRaster):
        ps.write("Gridfour_Huffman                               Compressed Output    |       Predictor Residuals\n")
        s = getattr(self, "_stats", None)
        if s is None or nTilesInRaster == 0:
            ps.write("   Tiles Compressed:  0\n")
            return
        ps.write("  Predictor                Times Used        bits/sym    bits/tile  |  m32 avg-len   avg-unique  entropy | bits in tree\n")
        for label, r in zip(self._STAT_LABELS, s):
            if label == "None":
                continue
            n, nm = int(r["n_tiles"]), int(r["n_m32_counted"])
            bits_per_symbol = 8.0 * r["n_bytes"] / r["n_symbols"] if r["n_symbols"] else 0.0
            ps.write("   %-20.20s %8d (%4.1f %%)     %5.2f  %12.1f   | %10.1f      %6.1f    %6.2f   | %6.1f\n" % (
                label, n, 100.0 * n / nTilesInRaster, bits_per_symbol, (r["n_bytes"] / n * 8 if n else 0.0),
                (r["sum_length_m32"] / nm if nm else 0.0), (r["sum_observed_m32"] / n if n else 0.0),
                (r["sum_entropy_m32"] / nm if nm else 0.0), (r["n_bits_overhead"] / n if n else 0.0)))

    def clearAnalysisData(self):
        self._stats = None
        self._pairs = None

    # ---- batched forms (host memory) ----
    def encode_batch(self, codecIndex, nRows, nCols, tiles):
        """tiles: int32 [nTiles, nRows*nCols].  Returns (packings: list[bytes|None], predictors, status)."""
        v = np.ascontiguousarray(tiles, dtype=np.int32).reshape(-1, nRows * nCols)
        nt = v.shape[0]
        cap = nt * int(lib().gf_huffman_default_stride(nRows, nCols))
        offsets = np.zeros(nt + 1, np.uint64)
        preds = np.zeros(nt, np.uint8)
        status = np.zeros(nt, np.int32)
        while True:
            blob = np.empty(max(cap, 16), np.uint8)
            st = self._fn("encode_batch_i32")(self.ctx.handle, codecIndex, nRows, nCols, nt, _ptr(v), _ptr(blob), cap,
                                              _ptr(offsets), _ptr(preds), _ptr(status))
            if st == _lib.ERR_CAPACITY:
                cap = int(offsets[nt]) + 16
                continue
            check(st, self._PREFIX + "_encode_batch_i32")
            break
        packs = []
        for t in range(nt):
            if status[t] == _lib.OK:
                packs.append(bytes(blob[int(offsets[t]):int(offsets[t + 1])]))
            else:
                packs.append(None)
        return packs, preds, status

    def decode_batch(self, nRows, nCols, packings):
        """packings: list of bytes.  Returns (values int32 [nTiles, cells], status)."""
        nt = len(packings)
        offsets = np.zeros(nt + 1, np.uint64)
        offsets[1:] = np.cumsum([len(p) for p in packings])
        blob = np.frombuffer(b"".join(packings) + b"\0" * 16, dtype=np.uint8)
        out = np.empty((nt, nRows * nCols), np.int32)
        status = np.zeros(nt, np.int32)
        check(self._fn("decode_batch_i32")(self.ctx.handle, nRows, nCols, nt, _ptr(blob), _ptr(offsets), _ptr(out),
                                           _ptr(status)), self._PREFIX + "_decode_batch_i32")
        return out, status


class CodecDeflateHip(CodecHuffmanHip):
    """Drop-in for org.gridfour.compress.CodecDeflate: predictor + CodecM32 on the MI355X, Deflate (level 6) on the host's
    zlib as the reference uses the JDK's."""

    _PREFIX = "gf_deflate"

    def _fn(self, name):
        if name == "max_packing":                    # nM32 + 128 bytes at most (CodecDeflate.java:204)
            return lambda r, c: int(lib().gf_m32_max_stream(r, c)) + 256
        return getattr(lib(), "%s_%s" % (self._PREFIX, name))


class CodecCanonHuffmanHip(CodecHuffmanHip):
    """Drop-in for org.gridfour.compress.canonicalHuffman.CodecCanonHuffman (the default integer codec of
    current Gridfour, GvrsFileSpecification.java:229), computed on the MI355X."""

    _PREFIX = "gf_canon"


class DeviceBuffer:
    """A raw device allocation owned through the C ABI (no torch needed)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().gf_dev_malloc(ctx.handle, max(self.nbytes, 16), C.byref(p)), "gf_dev_malloc")
        self.ptr = p

    def upload(self, array, byte_offset=0):
        a = np.ascontiguousarray(array)
        assert byte_offset + a.nbytes <= max(self.nbytes, 16)
        check(lib().gf_dev_upload(self.ctx.handle, C.c_void_p(self.ptr.value + byte_offset), _ptr(a), a.nbytes),
              "gf_dev_upload")
        return self

    def download(self, dtype, count, byte_offset=0):
        out = np.empty(count, dtype)
        assert byte_offset + out.nbytes <= max(self.nbytes, 16)
        check(lib().gf_dev_download(self.ctx.handle, _ptr(out), C.c_void_p(self.ptr.value + byte_offset),
                                    out.nbytes), "gf_dev_download")
        return out

    def fill(self, value=0):
        check(lib().gf_dev_memset(self.ctx.handle, self.ptr, value, self.nbytes), "gf_dev_memset")
        return self

    def free(self):
        if self.ptr:
            lib().gf_dev_free(self.ctx.handle, self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceTileBatch:
    """Device-resident batch of tiles and their packings: the measured hot path.

    encode(): values -> slots/lengths/predictors/status   (gf_huffman_encode_batch_i32_dev)
    decode(): slots/lengths -> decoded/status              (gf_huffman_decode_batch_i32_dev)
    Nothing synchronises unless asked; everything is enqueued on `stream` (default: the
    context's stream).
    """

    def __init__(self, ctx, n_rows, n_cols, n_tiles, slot_stride=None, codec="huffman"):
        assert codec in ("huffman", "canon", "lsop")
        self.codec = codec
        self.ctx, self.n_rows, self.n_cols, self.n_tiles = ctx, int(n_rows), int(n_cols), int(n_tiles)
        self.cells = self.n_rows * self.n_cols
        self.stride = int(slot_stride or lib().gf_huffman_default_stride(n_rows, n_cols))
        assert self.stride % 16 == 0
        nt = self.n_tiles
        self.values = DeviceBuffer(ctx, nt * self.cells * 4)
        self.decoded = DeviceBuffer(ctx, nt * self.cells * 4)
        self.slots = DeviceBuffer(ctx, nt * self.stride + 16)
        self.lengths = DeviceBuffer(ctx, nt * 4)
        self.predictors = DeviceBuffer(ctx, nt)
        self.enc_status = DeviceBuffer(ctx, nt * 4)
        self.dec_status = DeviceBuffer(ctx, nt * 4)
        ctx.reserve(n_rows, n_cols, nt)
        if codec == "lsop":                          # work buffers of the LSOP stages
            n = int(lib().gf_lsop12_residual_count(n_rows, n_cols))
            self.res_stride = (n + 3) // 4 * 4
            self.residuals = DeviceBuffer(ctx, nt * self.res_stride * 4 + 16)
            self.coefs = DeviceBuffer(ctx, nt * 64)
            self.scratch_status = DeviceBuffer(ctx, nt * 4)

    def synth_dem(self, seed, tiles_per_row, tile0=0, stream=None, mask_per_mille=0, style=0):
        if style:
            check(lib().gf_synth_dem_style_dev(self.ctx.handle, stream, seed & (2 ** 64 - 1), self.n_rows, self.n_cols,
                                               tiles_per_row, tile0, self.n_tiles, mask_per_mille, style, self.values.ptr),
                  "gf_synth_dem_style_dev")
            return
        if mask_per_mille:
            check(lib().gf_synth_dem_masked_dev(self.ctx.handle, stream, seed & (2 ** 64 - 1), self.n_rows, self.n_cols,
                                                tiles_per_row, tile0, self.n_tiles, mask_per_mille, self.values.ptr),
                  "gf_synth_dem_masked_dev")
            return
        check(lib().gf_synth_dem_dev(self.ctx.handle, stream, seed & (2 ** 64 - 1), self.n_rows, self.n_cols,
                                     tiles_per_row, tile0, self.n_tiles, self.values.ptr), "gf_synth_dem_dev")

    def encode(self, codec_index=0, predictor_mask=_lib.PM_ALL, stream=None, lsop_flags=0):
        if self.codec == "lsop":
            check(lib().gf_lsop12_encode_batch_i32_dev_ex(self.ctx.handle, stream, codec_index, self.n_rows, self.n_cols,
                                                          self.n_tiles, self.values.ptr, lsop_flags, self.slots.ptr, self.stride,
                                                          self.lengths.ptr, self.enc_status.ptr, self.residuals.ptr,
                                                          self.res_stride, self.coefs.ptr, self.scratch_status.ptr),
                  "gf_lsop12_encode_batch_i32_dev_ex")
            return
        fn = getattr(lib(), "gf_%s_encode_batch_i32_dev" % self.codec)
        check(fn(self.ctx.handle, stream, codec_index, self.n_rows, self.n_cols, self.n_tiles, self.values.ptr,
                 self.slots.ptr, self.stride, self.lengths.ptr, self.predictors.ptr, self.enc_status.ptr,
                 predictor_mask), "gf_%s_encode_batch_i32_dev" % self.codec)

    def decode(self, stream=None):
        if self.codec == "lsop":
            check(lib().gf_lsop12_decode_batch_i32_dev(self.ctx.handle, stream, self.n_rows, self.n_cols, self.n_tiles,
                                                       self.slots.ptr, self.n_tiles * self.stride, None, self.stride,
                                                       self.lengths.ptr, self.decoded.ptr, self.dec_status.ptr,
                                                       self.residuals.ptr, self.res_stride, self.coefs.ptr,
                                                       self.scratch_status.ptr), "gf_lsop12_decode_batch_i32_dev")
            return
        fn = getattr(lib(), "gf_%s_decode_batch_i32_dev" % self.codec)
        check(fn(self.ctx.handle, stream, self.n_rows, self.n_cols, self.n_tiles, self.slots.ptr,
                 self.n_tiles * self.stride, None, self.stride, self.lengths.ptr, self.decoded.ptr,
                 self.dec_status.ptr), "gf_%s_decode_batch_i32_dev" % self.codec)

    # host views (synchronising copies)
    def get_lengths(self):
        return self.lengths.download(np.uint32, self.n_tiles)

    def get_predictors(self):
        return self.predictors.download(np.uint8, self.n_tiles)

    def get_enc_status(self):
        return self.enc_status.download(np.int32, self.n_tiles)

    def get_dec_status(self):
        return self.dec_status.download(np.int32, self.n_tiles)

    def get_packing(self, t, length=None):
        if length is None:
            length = int(self.lengths.download(np.uint32, 1, 4 * t)[0])
        return bytes(self.slots.download(np.uint8, length, t * self.stride))

    def get_values(self, t0=0, n=None):
        n = self.n_tiles - t0 if n is None else n
        return self.values.download(np.int32, n * self.cells, t0 * self.cells * 4).reshape(n, self.cells)

    def get_decoded(self, t0=0, n=None):
        n = self.n_tiles - t0 if n is None else n
        return self.decoded.download(np.int32, n * self.cells, t0 * self.cells * 4).reshape(n, self.cells)

    def free(self):
        for b in (self.values, self.decoded, self.slots, self.lengths, self.predictors, self.enc_status,
                  self.dec_status):
            b.free()


class GpuTimer:
    """HIP-event timer on the stream the kernels are launched on (gf_timer_*)."""

    def __init__(self, ctx):
        self._h = C.c_void_p()
        check(lib().gf_timer_create(ctx.handle, C.byref(self._h)), "gf_timer_create")

    def start(self, stream=None):
        check(lib().gf_timer_start(self._h, stream), "gf_timer_start")

    def stop(self, stream=None):
        check(lib().gf_timer_stop(self._h, stream), "gf_timer_stop")

    def elapsed_ms(self):
        ms = C.c_float(0)
        check(lib().gf_timer_elapsed_ms(self._h, C.byref(ms)), "gf_timer_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if self._h:
                lib().gf_timer_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class CodecFloatHip:
    """Drop-in for org.gridfour.compress.CodecFloat (CodecFloat.java:328-458): float32 tiles as five
    byte planes (split/merged on the GPU) each compressed with zlib on the host.

    `level` is the zlib level: 9 is what the current reference source passes to java.util.zip.Deflater,
    6 is what the reference's sample files were written with."""

    def __init__(self, context=None, device=0, level=9):
        self.ctx = context if context is not None else GvrsHipContext(device)
        self.level = int(level)

    # ---- ICompressionEncoder ----
    def encode(self, codecIndex, nRows, nCols, values):
        return None                                  # CodecFloat.java: integer encoding not implemented

    def encodeFloats(self, codecIndex, nRows, nCols, values):
        v = np.ascontiguousarray(values, dtype=np.float32).ravel()
        if v.size != nRows * nCols:
            raise ValueError("values.length != nRows*nCols")
        cap = 5 * v.size + 4096
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        check(lib().gf_float_encode_f32(self.ctx.handle, codecIndex, nRows, nCols, _ptr(v), self.level, _ptr(out), cap,
                                        C.byref(n)), "gf_float_encode_f32")
        return bytes(out[:n.value])

    def implementsFloatingPointEncoding(self):
        return True

    def implementsIntegerEncoding(self):
        return False

    # ---- ICompressionDecoder ----
    def decode(self, nRows, nColumns, packing):
        return None

    def decodeFloats(self, nRows, nColumns, packing):
        p = np.frombuffer(bytes(packing), dtype=np.uint8)
        out = np.empty(nRows * nColumns, np.float32)
        st = lib().gf_float_decode_f32(self.ctx.handle, nRows, nColumns, _ptr(p), p.size, _ptr(out))
        if st in (_lib.ERR_FORMAT, _lib.ERR_BOUNDS):
            raise IOError(lib().gf_status_string(st).decode())
        check(st, "gf_float_decode_f32")
        return out

    # ---- batched, host memory ----
    def encode_floats_batch(self, codecIndex, nRows, nCols, tiles):
        v = np.ascontiguousarray(tiles, dtype=np.float32).reshape(-1, nRows * nCols)
        nt = v.shape[0]
        cap = nt * (5 * nRows * nCols + 4096)
        blob = np.empty(cap, np.uint8)
        offsets = np.zeros(nt + 1, np.uint64)
        check(lib().gf_float_encode_batch_f32(self.ctx.handle, codecIndex, nRows, nCols, nt, _ptr(v), self.level, _ptr(blob),
                                              cap, _ptr(offsets)), "gf_float_encode_batch_f32")
        return [bytes(blob[int(offsets[t]):int(offsets[t + 1])]) for t in range(nt)]

    def decode_floats_batch(self, nRows, nCols, packings):
        nt = len(packings)
        offsets = np.zeros(nt + 1, np.uint64)
        offsets[1:] = np.cumsum([len(p) for p in packings])
        blob = np.frombuffer(b"".join(packings) + b"\0" * 16, dtype=np.uint8)
        out = np.empty((nt, nRows * nCols), np.float32)
        status = np.zeros(nt, np.int32)
        check(lib().gf_float_decode_batch_f32(self.ctx.handle, nRows, nCols, nt, _ptr(blob), _ptr(offsets), _ptr(out),
                                              _ptr(status)), "gf_float_decode_batch_f32")
        return out, status


class LsCodecHip:
    """Drop-in for org.gridfour.lsop.LsEncoder12 + LsDecoder12 (codec id "LSOP12", LsCodecUtility.java:53),
    computed on the MI355X; the Deflate alternative container uses the host's zlib as the reference uses the JDK's."""

    def __init__(self, context=None, device=0, deflate_enabled=True, value_checksum_enabled=False):
        self.ctx = context if context is not None else GvrsHipContext(device)
        self.deflate_enabled = bool(deflate_enabled)           # LsEncoder12.setDeflateEnabled, default true
        self.value_checksum_enabled = bool(value_checksum_enabled)   # LsEncoder12.setValueChecksumEnabled, default false

    def setDeflateEnabled(self, enabled):
        self.deflate_enabled = bool(enabled)

    def setValueChecksumEnabled(self, enabled):
        self.value_checksum_enabled = bool(enabled)            # LsEncoder12.java:117-119

    # ---- ICompressionEncoder ----
    def encode(self, codecIndex, nRows, nCols, values):
        packs, _, status = self.encode_batch(codecIndex, nRows, nCols, np.asarray(values).reshape(1, -1))
        if status[0] == _lib.DECLINED:
            return None
        check(int(status[0]), "gf_lsop12_encode_batch_i32")
        return packs[0]

    def encodeFloats(self, codecIndex, nRows, nCols, values):
        return None                                             # LsEncoder12.java:222-224

    def implementsFloatingPointEncoding(self):
        return False

    def implementsIntegerEncoding(self):
        return True

    # ---- ICompressionDecoder ----
    def decode(self, nRows, nColumns, packing):
        vals, status = self.decode_batch(nRows, nColumns, [packing])
        if status[0] in (_lib.ERR_FORMAT, _lib.ERR_BOUNDS):
            raise IOError(lib().gf_status_string(int(status[0])).decode())
        check(int(status[0]), "gf_lsop12_decode_batch_i32")
        return vals[0]

    def decodeFloats(self, nRows, nColumns, packing):
        return None

    # ---- batched forms (host memory) ----
    def encode_batch(self, codecIndex, nRows, nCols, tiles):
        """Returns (packings: list[bytes|None], container types uint8, status int32)."""
        v = np.ascontiguousarray(tiles, dtype=np.int32).reshape(-1, nRows * nCols)
        nt = v.shape[0]
        cap = nt * int(lib().gf_lsop12_max_packing(nRows, nCols)) + 64
        blob = np.empty(cap, np.uint8)
        offsets = np.zeros(nt + 1, np.uint64)
        types = np.zeros(nt, np.uint8)
        status = np.zeros(nt, np.int32)
        check(lib().gf_lsop12_encode_batch_i32(self.ctx.handle, codecIndex, nRows, nCols, nt, _ptr(v),
                                               (_lib.LSOP_DEFLATE if self.deflate_enabled else 0) |
                                               (_lib.LSOP_VALUE_CHECKSUM if self.value_checksum_enabled else 0),
                                               _ptr(blob), cap, _ptr(offsets), _ptr(types),
                                               _ptr(status)), "gf_lsop12_encode_batch_i32")
        packs = [bytes(blob[int(offsets[t]):int(offsets[t + 1])]) if status[t] == _lib.OK else None for t in range(nt)]
        return packs, types, status

    def decode_batch(self, nRows, nCols, packings):
        nt = len(packings)
        offsets = np.zeros(nt + 1, np.uint64)
        offsets[1:] = np.cumsum([len(p) for p in packings])
        blob = np.frombuffer(b"".join(packings) + b"\0" * 16, dtype=np.uint8)
        out = np.zeros((nt, nRows * nCols), np.int32)
        status = np.zeros(nt, np.int32)
        check(lib().gf_lsop12_decode_batch_i32(self.ctx.handle, nRows, nCols, nt, _ptr(blob), _ptr(offsets), _ptr(out),
                                               _ptr(status)), "gf_lsop12_decode_batch_i32")
        return out, status

    # ---- the predictor stage alone (device buffers handled here; used by tests and tools) ----
    def predict(self, nRows, nCols, tiles):
        """LsOptimalPredictor12.encode: returns (seed, coefficients float32[nt,12], residuals int32[nt,n], status)."""
        v = np.ascontiguousarray(tiles, dtype=np.int32).reshape(-1, nRows * nCols)
        nt = v.shape[0]
        n = int(lib().gf_lsop12_residual_count(nRows, nCols))
        stride = (n + 3) // 4 * 4
        dv, dr, dc, ds = (DeviceBuffer(self.ctx, v.nbytes), DeviceBuffer(self.ctx, nt * stride * 4 + 16),
                          DeviceBuffer(self.ctx, nt * 64), DeviceBuffer(self.ctx, nt * 4))
        dv.upload(v)
        check(lib().gf_lsop12_predict_dev(self.ctx.handle, None, nRows, nCols, nt, dv.ptr, dr.ptr, stride, dc.ptr, ds.ptr),
              "gf_lsop12_predict_dev")
        self.ctx.synchronize()
        res = dr.download(np.int32, nt * stride).reshape(nt, stride)[:, :n]
        coefs = dc.download(np.uint32, nt * 16).reshape(nt, 16)
        status = ds.download(np.int32, nt)
        for b in (dv, dr, dc, ds):
            b.free()
        return coefs[:, 0].astype(np.int32), coefs[:, 1:13].copy().view(np.float32), res, status

    def reconstruct(self, nRows, nCols, seeds, coefficients, residuals):
        """LsDecoder12.unpackInitializers/unpackInterior: returns (values int32[nt, cells], status)."""
        res = np.ascontiguousarray(residuals, dtype=np.int32)
        nt, n = res.shape
        stride = (n + 3) // 4 * 4
        padded = np.zeros((nt, stride), np.int32)
        padded[:, :n] = res
        coefs = np.zeros((nt, 16), np.uint32)
        coefs[:, 0] = np.asarray(seeds, np.int32).view(np.uint32)
        coefs[:, 1:13] = np.ascontiguousarray(coefficients, np.float32).view(np.uint32).reshape(nt, 12)
        dv, dr, dc, ds = (DeviceBuffer(self.ctx, nt * nRows * nCols * 4), DeviceBuffer(self.ctx, padded.nbytes + 16),
                          DeviceBuffer(self.ctx, nt * 64), DeviceBuffer(self.ctx, nt * 4))
        dr.upload(padded)
        dc.upload(coefs)
        check(lib().gf_lsop12_reconstruct_dev(self.ctx.handle, None, nRows, nCols, nt, dr.ptr, stride, dc.ptr, None, dv.ptr,
                                              ds.ptr), "gf_lsop12_reconstruct_dev")
        self.ctx.synchronize()
        vals = dv.download(np.int32, nt * nRows * nCols).reshape(nt, nRows * nCols)
        status = ds.download(np.int32, nt)
        for b in (dv, dr, dc, ds):
            b.free()
        return vals, status


CODEC_NONE, CODEC_HUFFMAN, CODEC_DEFLATE, CODEC_CANON_HUFFMAN, CODEC_LSOP12 = 0, 1, 2, 3, 4
STANDARD_CODEC_LIST = (CODEC_HUFFMAN, CODEC_DEFLATE, CODEC_NONE, CODEC_CANON_HUFFMAN)    # GvrsFileSpecification.java:221-230


class CodecMasterHip:
    """org.gridfour.gvrs.CodecMaster over a codec list, batched: the strictly shortest packing per tile (list order
    breaks ties), decode dispatch on packing[0]."""

    def __init__(self, codec_list=STANDARD_CODEC_LIST, context=None, device=0):
        self.ctx = context if context is not None else GvrsHipContext(device)
        self.codecs = np.asarray(codec_list, dtype=np.int32)

    def encode_batch(self, nRows, nCols, tiles):
        """Returns (packings: list[bytes|None], codec index used uint8 (255 = none), status)."""
        v = np.ascontiguousarray(tiles, dtype=np.int32).reshape(-1, nRows * nCols)
        nt = v.shape[0]
        cap = nt * (4 * nRows * nCols + 1024) + 64
        offsets = np.zeros(nt + 1, np.uint64)
        used = np.zeros(nt, np.uint8)
        status = np.zeros(nt, np.int32)
        while True:
            blob = np.empty(cap, np.uint8)
            st = lib().gf_codec_master_encode_batch_i32(self.ctx.handle, _ptr(self.codecs), self.codecs.size, nRows, nCols, nt,
                                                        _ptr(v), _ptr(blob), cap, _ptr(offsets), _ptr(used), _ptr(status))
            if st == _lib.ERR_CAPACITY:
                cap = int(offsets[nt]) + 64
                continue
            check(st, "gf_codec_master_encode_batch_i32")
            break
        packs = [bytes(blob[int(offsets[t]):int(offsets[t + 1])]) if status[t] == _lib.OK else None for t in range(nt)]
        return packs, used, status

    def decode_batch(self, nRows, nCols, packings):
        nt = len(packings)
        offsets = np.zeros(nt + 1, np.uint64)
        offsets[1:] = np.cumsum([len(p) for p in packings])
        blob = np.frombuffer(b"".join(packings) + b"\0" * 16, dtype=np.uint8)
        out = np.zeros((nt, nRows * nCols), np.int32)
        status = np.zeros(nt, np.int32)
        check(lib().gf_codec_master_decode_batch_i32(self.ctx.handle, _ptr(self.codecs), self.codecs.size, nRows, nCols, nt,
                                                     _ptr(blob), _ptr(offsets), _ptr(out), _ptr(status)),
              "gf_codec_master_decode_batch_i32")
        return out, status

    # ---- tile payloads: RasterTile.getCompressedPacking over TileElementInt.encode (one integer element per tile) ----
    def tile_payloads(self, nRows, nCols, tiles):
        """Returns (payloads: list[bytes], codec index used uint8 (255 = raw cells))."""
        v = np.ascontiguousarray(tiles, dtype=np.int32).reshape(-1, nRows * nCols)
        nt = v.shape[0]
        cap = nt * (4 * nRows * nCols + 8)
        blob = np.empty(cap, np.uint8)
        offsets = np.zeros(nt + 1, np.uint64)
        used = np.zeros(nt, np.uint8)
        check(lib().gf_tile_payload_encode_batch_i32(self.ctx.handle, _ptr(self.codecs), self.codecs.size, nRows, nCols, nt,
                                                     _ptr(v), _ptr(blob), cap, _ptr(offsets), _ptr(used)),
              "gf_tile_payload_encode_batch_i32")
        return [bytes(blob[int(offsets[t]):int(offsets[t + 1])]) for t in range(nt)], used

    # ---- tile records: RecordManager.writeTile / readTile for a batch of dirty tiles (one integer-coded element) ----
    def tile_records(self, nRows, nCols, tile_indices, tiles, element="int", fill_value=-32768, checksums=True):
        """Returns (records: list[bytes] exactly as RecordManager appends them to the file, codec index used (255 = raw))."""
        short = element == "short"
        v = np.ascontiguousarray(tiles, dtype=np.int16 if short else np.int32).reshape(-1, nRows * nCols)
        nt = v.shape[0]
        idx = np.ascontiguousarray(tile_indices, dtype=np.int32)
        assert idx.size == nt
        cap = nt * int(lib().gf_tile_record_max_bytes(int(short), nRows, nCols))
        blob = np.empty(max(cap, 16), np.uint8)
        offsets = np.zeros(nt + 1, np.uint64)
        used = np.zeros(nt, np.uint8)
        codecs = self.codecs if self.codecs.size else np.zeros(1, np.int32)
        check(lib().gf_tile_record_encode_batch(self.ctx.handle, _ptr(codecs), self.codecs.size, int(short), int(fill_value),
                                                nRows, nCols, nt, _ptr(idx), _ptr(v), int(bool(checksums)), _ptr(blob), cap,
                                                _ptr(offsets), _ptr(used)), "gf_tile_record_encode_batch")
        return [bytes(blob[int(offsets[t]):int(offsets[t + 1])]) for t in range(nt)], used

    def tiles_from_records(self, nRows, nCols, records, element="int", verify_checksums=True):
        """Returns (tile indices, values [nt, cells] int32 / int16, status per record)."""
        short = element == "short"
        nt = len(records)
        offsets = np.zeros(nt + 1, np.uint64)
        offsets[1:] = np.cumsum([len(p) for p in records])
        blob = np.frombuffer(b"".join(records) + b"\0" * 16, dtype=np.uint8)
        out = np.zeros((nt, nRows * nCols), np.int16 if short else np.int32)
        idx = np.full(nt, -1, np.int32)
        status = np.zeros(nt, np.int32)
        codecs = self.codecs if self.codecs.size else np.zeros(1, np.int32)
        check(lib().gf_tile_record_decode_batch(self.ctx.handle, _ptr(codecs), self.codecs.size, int(short), nRows, nCols, nt,
                                                _ptr(blob), _ptr(offsets), int(bool(verify_checksums)), _ptr(idx), _ptr(out),
                                                _ptr(status)), "gf_tile_record_decode_batch")
        return idx, out, status

    def tiles_from_payloads(self, nRows, nCols, payloads):
        nt = len(payloads)
        offsets = np.zeros(nt + 1, np.uint64)
        offsets[1:] = np.cumsum([len(p) for p in payloads])
        blob = np.frombuffer(b"".join(payloads) + b"\0" * 16, dtype=np.uint8)
        out = np.zeros((nt, nRows * nCols), np.int32)
        status = np.zeros(nt, np.int32)
        check(lib().gf_tile_payload_decode_batch_i32(self.ctx.handle, _ptr(self.codecs), self.codecs.size, nRows, nCols, nt,
                                                     _ptr(blob), _ptr(offsets), _ptr(out), _ptr(status)),
              "gf_tile_payload_decode_batch_i32")
        return out, status
