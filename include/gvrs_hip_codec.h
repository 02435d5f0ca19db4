/*
 * gvrs_hip_codec.h -- C ABI of the MI355X-native GVRS tile codec (libgvrs_hip.so).
 *
 * This is the drop-in boundary for the Gridfour compression plug-in interface.
 * Reference paths are relative to core/src/main/java/org/gridfour/ of
 * gwlucastrig/gridfour.  The entry points are what a JNI binding of
 *     compress/ICompressionEncoder.java:61-91   (encode / encodeFloats)
 *     compress/ICompressionDecoder.java:62-105  (decode / decodeFloats)
 * would bind for the codec registered as "GvrsHuffman"
 * (compress/CodecHuffman.java:70-153, registered the way
 * gvrs/GvrsFileSpecification.java:1576-1631 addCompressionCodec does).
 * INTEGRATION.md shows the Java adapter + JNI stub a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types.
 *   - "host" entry points take host pointers and do their own H2D/D2H copies.
 *   - "dev" entry points take device pointers plus a HIP stream (hipStream_t
 *     passed as void*); they only enqueue work, never synchronise, never
 *     allocate (gf_context_reserve first) -> safe for hipGraph capture.
 *   - tiles are row-major int32[nRows*nCols], batches are contiguous tiles.
 *   - every function returns a gf_status; per-tile outcomes of batch calls are
 *     reported in a status array (int32 per tile).
 *   - bit-exactness contract: for every tile, the bytes produced equal the
 *     bytes CodecHuffman.encode returns for the same (codecIndex, nRows, nCols,
 *     values), and decode inverts CodecHuffman.decode exactly.
 *   - there is NO CPU fallback: without a usable HIP device every compute
 *     entry point returns GF_ERR_NO_DEVICE.
 */
#ifndef GVRS_HIP_CODEC_H
#define GVRS_HIP_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gf_status {
    GF_OK = 0,
    GF_DECLINED = 1,          /* encoder result is Java `null` (all cells null): CodecHuffman.java:80-82 */
    GF_OVERFLOW = 2,          /* packing longer than the slot/capacity handed in; length is still reported */
    GF_ERR_FORMAT = -1,       /* decoder: Java would throw IOException (CodecHuffman.java:155-169)        */
    GF_ERR_BOUNDS = -2,       /* Java would throw ArrayIndexOutOfBounds (short packing, nCols < 2, ...)   */
    GF_ERR_CAPACITY = -3,     /* host output buffer too small                                             */
    GF_ERR_ARG = -4,
    GF_ERR_NO_DEVICE = -5,    /* no HIP device / HIP runtime error at context creation                    */
    GF_ERR_HIP = -6,          /* a HIP call failed; gf_last_error() has the text                          */
    GF_ERR_UNSUPPORTED = -7
} gf_status;

/* INT4_NULL_CODE, util/GridfourConstants.java:61 */
#define GF_INT4_NULL ((int32_t)0x80000000)

/* predictor codes, compress/PredictorModelType.java:46-63 */
#define GF_PM_DIFFERENCING 1
#define GF_PM_LINEAR 2
#define GF_PM_TRIANGLE 3
#define GF_PM_DIFFERENCING_NULLS 4
/* predictor_mask bit for model m is 1 << (m-1); GF_PM_ALL = reference behaviour */
#define GF_PM_ALL 0xF

typedef struct gf_context gf_context;

/* ---- library / device ---- */
const char *gf_version(void);
const char *gf_status_string(int status);
/* text of the last HIP error seen by the calling thread ("" if none) */
const char *gf_last_error(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int gf_device_count(void);

/* One context per (process, device): owns a stream and the scratch workspace.
 * A context may be called from several threads at once, as the reference calls
 * one codec instance from its tile cache and its decompression assistant
 * (gvrs/RasterTileCache.java:418-421, TileDecompressionAssistant.java:68-73):
 * every entry point that takes a context holds the context's lock for its
 * duration, so such calls run one after the other.  The *_dev entry points
 * return while their kernels are still queued on the context's stream; a
 * caller that hands them another stream orders that stream itself.  Different
 * contexts are independent (this is how tile batches shard over the GPUs of a
 * node).                                                                      */
gf_status gf_context_create(int device, gf_context **ctx);
void gf_context_destroy(gf_context *ctx);
/* pre-allocates the workspace (decode: M32 spill per resident workgroup, per-tile records of the tree / code-length
 * pre-pass kernels; encode: per-tile selection records between the two encoder kernels) for batches up to n_tiles tiles
 * of n_rows x n_cols; without it the first larger _dev call grows it */
gf_status gf_context_reserve(gf_context *ctx, int n_rows, int n_cols, size_t n_tiles);
/* the context's own stream (hipStream_t) */
void *gf_context_stream(gf_context *ctx);
gf_status gf_context_synchronize(gf_context *ctx);

/* slot stride (bytes, multiple of 16) the batch encoders use by default:
 * room for any packing the caller would keep (RasterTile keeps a packing only
 * when shorter than 4*cells, gvrs/TileElementInt.java:198-204).              */
size_t gf_huffman_default_stride(int n_rows, int n_cols);
/* absolute worst-case packing size of CodecHuffman for a tile */
size_t gf_huffman_max_packing(int n_rows, int n_cols);

/* ---- single tile, host memory: replaces ICompressionEncoder.encode /
 *      ICompressionDecoder.decode as implemented by CodecHuffman ----------- */
/* returns GF_OK, GF_DECLINED (Java null), GF_ERR_CAPACITY (out_len = needed) */
gf_status gf_huffman_encode_i32(gf_context *ctx, int codec_index, int n_rows, int n_cols,
                                const int32_t *values, uint8_t *out, size_t out_cap,
                                size_t *out_len);
/* returns GF_OK or GF_ERR_FORMAT / GF_ERR_BOUNDS (Java IOException / AIOOBE) */
gf_status gf_huffman_decode_i32(gf_context *ctx, int n_rows, int n_cols,
                                const uint8_t *packing, size_t packing_len, int32_t *values);

/* ---- batches, host memory --------------------------------------------- */
/* Encodes n_tiles tiles.  Packings are concatenated in tile order into blob;
 * offsets[n_tiles+1] receives their byte offsets (offsets[t+1]-offsets[t] = length,
 * 0 for a declined tile).  predictors[n_tiles] (optional) receives the predictor
 * code chosen per tile, status[n_tiles] (optional) the per-tile gf_status.
 * Returns GF_ERR_CAPACITY (offsets still filled) when blob_cap is too small.  */
gf_status gf_huffman_encode_batch_i32(gf_context *ctx, int codec_index, int n_rows, int n_cols,
                                      size_t n_tiles, const int32_t *values, uint8_t *blob,
                                      size_t blob_cap, uint64_t *offsets, uint8_t *predictors,
                                      int32_t *status);
gf_status gf_huffman_decode_batch_i32(gf_context *ctx, int n_rows, int n_cols, size_t n_tiles,
                                      const uint8_t *blob, const uint64_t *offsets,
                                      int32_t *values, int32_t *status);

/* ---- batches, device-resident (the measured hot path) ------------------ */
/* d_values  : n_tiles * n_rows*n_cols int32
 * d_out     : n_tiles slots of slot_stride bytes (16-byte aligned base, stride % 16 == 0);
 *             tile t's packing starts at t*slot_stride
 * d_lengths : packing length per tile (bytes; 0 = declined)
 * d_predictors (optional, may be NULL), d_status: per tile
 * predictor_mask: GF_PM_ALL for reference behaviour; a subset restricts the
 *             models tried (test hook, mirrors the oracle)                   */
gf_status gf_huffman_encode_batch_i32_dev(gf_context *ctx, void *stream, int codec_index,
                                          int n_rows, int n_cols, size_t n_tiles,
                                          const int32_t *d_values, uint8_t *d_out,
                                          size_t slot_stride, uint32_t *d_lengths,
                                          uint8_t *d_predictors, int32_t *d_status,
                                          int predictor_mask);
/* tile t's packing = d_blob[d_offsets[t] .. d_offsets[t]+d_lengths[t]).  When
 * d_offsets is NULL the packings sit in slots: offset = t*slot_stride.
 * d_blob must be 4-byte aligned; blob_bytes = readable size of d_blob.       */
gf_status gf_huffman_decode_batch_i32_dev(gf_context *ctx, void *stream, int n_rows, int n_cols,
                                          size_t n_tiles, const uint8_t *d_blob,
                                          size_t blob_bytes, const uint64_t *d_offsets,
                                          size_t slot_stride, const uint32_t *d_lengths,
                                          int32_t *d_values, int32_t *d_status);
/* Gathers slot-strided packings into one contiguous blob (exclusive scan of the
 * lengths + copy): d_offsets[n_tiles+1], d_blob capacity blob_cap bytes.      */
gf_status gf_compact_dev(gf_context *ctx, void *stream, size_t n_tiles, const uint8_t *d_slots,
                         size_t slot_stride, const uint32_t *d_lengths, uint64_t *d_offsets,
                         uint8_t *d_blob, size_t blob_cap);

/* gf_compact_dev cannot report a blob that is too small without synchronising: a packing that would end behind blob_cap
 * is skipped; d_offsets[n_tiles] > blob_cap tells the caller (after its own synchronisation) that this happened.        */

/* ---- page-locked host memory ---------------------------------------------------------------------------------------
 * The host-memory batch entry points cut a batch into chunks of about 64 MB of cells and pipeline them through three
 * slots (copy-in of chunk k+1, device work of chunk k and copy-out of chunk k-1 overlap; device and staging memory are
 * bounded by the chunk, not by the batch).  Pageable memory is staged through page-locked buffers by helper threads;
 * memory obtained here (a JNI binding hands it to Java as a direct ByteBuffer) moves over PCIe in place.               */
gf_status gf_host_alloc(size_t bytes, void **p);
gf_status gf_host_free(void *p);

/* ---- zlib (RFC 1950 / 1951) streams inflated on the GPU, one wave per stream -------------------------------------------
 * Replaces the java.util.zip.Inflater calls of the Deflate-carrying decoders (compress/CodecDeflate.java:141-147,
 * compress/CodecFloat.java:285-298, lsop/LsDecoder12.java:127-141).  Stream i is d_in[in_offsets[i] .. + in_lengths[i]);
 * at most out_caps[i] bytes go to d_out + out_offsets[i]; d_produced[i] = bytes written, d_status[i] = GF_OK or GF_ERR_FORMAT
 * -- the outcome of ONE Inflater.inflate(byte[]) call on the whole input: it stops without error when room or input run
 * out, reports invalid data where zlib does (DataFormatException) and verifies the Adler-32 when the stream ends inside
 * the room.  The four descriptor arrays are host arrays; d_in, d_out, d_produced, d_status are device memory.
 * Unlike the other _dev entry points this one ALLOCATES and BLOCKS: it uploads the descriptor arrays (a context-owned device
 * buffer that grows on demand: hipMalloc on first use and when n_streams grows; a pageable-memory copy that the host waits
 * for), so it is not capture-safe and is not covered by gf_context_reserve.  d_in must be 4-byte aligned and readable up to
 * the end of the aligned dword that holds the last byte of every stream: the kernel reads whole aligned dwords, i.e. up to
 * three bytes before a stream's first and after its last byte are touched (never used) -- leave 4 bytes of padding behind
 * the last stream of an allocation.                                                                                     */
gf_status gf_inflate_batch_dev(gf_context *ctx, void *stream, size_t n_streams, const uint8_t *d_in,
                               const uint64_t *in_offsets, const uint32_t *in_lengths, uint8_t *d_out,
                               const uint64_t *out_offsets, const uint32_t *out_caps, uint32_t *d_produced,
                               int32_t *d_status);

/* Deflate-carrying packings decoded entirely on the device (container walk, inflate, decode; scratch bounded by chunks):
 * CodecDeflate (layout as gf_huffman_decode_batch_i32_dev) and CodecFloat (d_offsets required).  The host-memory forms
 * gf_deflate_decode_batch_i32 / gf_float_decode_batch_f32 run the same kernels behind the pipelined staging.           */
gf_status gf_deflate_decode_batch_i32_dev(gf_context *ctx, void *stream, int n_rows, int n_cols, size_t n_tiles,
                                          const uint8_t *d_blob, size_t blob_bytes, const uint64_t *d_offsets,
                                          size_t slot_stride, const uint32_t *d_lengths, int32_t *d_values, int32_t *d_status);
gf_status gf_float_decode_batch_f32_dev(gf_context *ctx, void *stream, int n_rows, int n_cols, size_t n_tiles,
                                        const uint8_t *d_blob, size_t blob_bytes, const uint64_t *d_offsets,
                                        const uint32_t *d_lengths, float *d_values, int32_t *d_status);

/* ---- several GPUs from one process -----------------------------------------------------------------
 * What gvrs/CodecMaster.java:142-203 / gvrs/RecordManager.java:386-490 would call to use a whole node from one JVM.
 * A gf_multi owns one gf_context per listed device (a device may be listed more than once).  A batch of T tiles shards
 * as contiguous ranges, shard g = tiles [g T / G, (g+1) T / G) (gf_multi_partition); tiles are independent
 * (gvrs/RasterTile.java:237-241), so there is no exchange between devices and no collective.
 *   gf_*_batch_i32_multi      host memory: one host thread per context runs the pipelined host path on its range; the
 *                             packings are concatenated by an exclusive scan of the range totals.  Arguments, results and
 *                             bytes are those of the single-context call on the whole batch.
 *   gf_*_batch_i32_multi_dev  device memory: every array argument has gf_multi_count() entries, entry g lives on
 *                             gf_multi_device(g); the call enqueues each shard on its device's stream and returns;
 *                             gf_multi_synchronize waits for all devices.                                             */
typedef struct gf_multi gf_multi;
gf_status gf_multi_create(const int *devices, int n_devices, gf_multi **multi);
void gf_multi_destroy(gf_multi *multi);
int gf_multi_count(const gf_multi *multi);
gf_context *gf_multi_context(gf_multi *multi, int i);
int gf_multi_device(const gf_multi *multi, int i);
void gf_multi_partition(size_t n_tiles, int n_shards, int i, size_t *t0, size_t *t1);
gf_status gf_multi_synchronize(gf_multi *multi);
gf_status gf_huffman_encode_batch_i32_multi(gf_multi *multi, int codec_index, int n_rows, int n_cols, size_t n_tiles,
                                            const int32_t *values, uint8_t *blob, size_t blob_cap, uint64_t *offsets,
                                            uint8_t *predictors, int32_t *status);
gf_status gf_huffman_decode_batch_i32_multi(gf_multi *multi, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                            const uint64_t *offsets, int32_t *values, int32_t *status);
gf_status gf_canon_encode_batch_i32_multi(gf_multi *multi, int codec_index, int n_rows, int n_cols, size_t n_tiles,
                                          const int32_t *values, uint8_t *blob, size_t blob_cap, uint64_t *offsets,
                                          uint8_t *predictors, int32_t *status);
gf_status gf_canon_decode_batch_i32_multi(gf_multi *multi, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                          const uint64_t *offsets, int32_t *values, int32_t *status);
/* the same partition for the other codecs (CodecDeflate; LSOP12 -- BASELINE config 5(ii) sharded; CodecFloat -- config 5(i)):
 * arguments as the single-context entry points of the same name without _multi                                          */
gf_status gf_deflate_encode_batch_i32_multi(gf_multi *multi, int codec_index, int n_rows, int n_cols, size_t n_tiles,
                                            const int32_t *values, uint8_t *blob, size_t blob_cap, uint64_t *offsets,
                                            uint8_t *predictors, int32_t *status);
gf_status gf_deflate_decode_batch_i32_multi(gf_multi *multi, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                            const uint64_t *offsets, int32_t *values, int32_t *status);
gf_status gf_lsop12_encode_batch_i32_multi(gf_multi *multi, int codec_index, int n_rows, int n_cols, size_t n_tiles,
                                           const int32_t *values, int deflate_enabled, uint8_t *blob, size_t blob_cap,
                                           uint64_t *offsets, uint8_t *types, int32_t *status);
gf_status gf_lsop12_decode_batch_i32_multi(gf_multi *multi, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                           const uint64_t *offsets, int32_t *values, int32_t *status);
gf_status gf_float_encode_batch_f32_multi(gf_multi *multi, int codec_index, int n_rows, int n_cols, size_t n_tiles,
                                          const float *values, int zlib_level, uint8_t *blob, size_t blob_cap,
                                          uint64_t *offsets);
gf_status gf_float_decode_batch_f32_multi(gf_multi *multi, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                          const uint64_t *offsets, float *values, int32_t *status);
gf_status gf_canon_encode_batch_i32_multi_dev(gf_multi *multi, int codec_index, int n_rows, int n_cols,
                                              const size_t *n_tiles, const int32_t *const *d_values, uint8_t *const *d_out,
                                              size_t slot_stride, uint32_t *const *d_lengths, uint8_t *const *d_predictors,
                                              int32_t *const *d_status, int predictor_mask);
gf_status gf_canon_decode_batch_i32_multi_dev(gf_multi *multi, int n_rows, int n_cols, const size_t *n_tiles,
                                              const uint8_t *const *d_blob, const size_t *blob_bytes,
                                              const uint64_t *const *d_offsets, size_t slot_stride,
                                              const uint32_t *const *d_lengths, int32_t *const *d_values,
                                              int32_t *const *d_status);
gf_status gf_huffman_encode_batch_i32_multi_dev(gf_multi *multi, int codec_index, int n_rows, int n_cols,
                                                const size_t *n_tiles, const int32_t *const *d_values, uint8_t *const *d_out,
                                                size_t slot_stride, uint32_t *const *d_lengths, uint8_t *const *d_predictors,
                                                int32_t *const *d_status, int predictor_mask);
gf_status gf_huffman_decode_batch_i32_multi_dev(gf_multi *multi, int n_rows, int n_cols, const size_t *n_tiles,
                                                const uint8_t *const *d_blob, const size_t *blob_bytes,
                                                const uint64_t *const *d_offsets, size_t slot_stride,
                                                const uint32_t *const *d_lengths, int32_t *const *d_values,
                                                int32_t *const *d_status);

/* ---- CodecCanonHuffman (compress/canonicalHuffman/CodecCanonHuffman.java:70-195), the default integer
 * codec of current Gridfour (gvrs/GvrsFileSpecification.java:229): same predictors, integer residuals coded
 * with the 260-symbol canonical Huffman stage (CanonicalHuffman.java:177-283, 441-519), 6-byte header
 * codec_index, predictor, seed LE; uniform tiles pack to 6 bytes with predictor 0 (:95-110).
 * Same calling conventions, layouts and statuses as the gf_huffman_* entry points; additionally
 * GF_ERR_ARG where the Java encoder throws IllegalArgumentException (Triangle on a one-row tile, :183)
 * and GF_ERR_UNSUPPORTED for tiles of 2^20 cells or more.                                             */
size_t gf_canon_max_packing(int n_rows, int n_cols);
gf_status gf_canon_encode_i32(gf_context *ctx, int codec_index, int n_rows, int n_cols, const int32_t *values,
                              uint8_t *out, size_t out_cap, size_t *out_len);
gf_status gf_canon_decode_i32(gf_context *ctx, int n_rows, int n_cols, const uint8_t *packing, size_t packing_len,
                              int32_t *values);
gf_status gf_canon_encode_batch_i32(gf_context *ctx, int codec_index, int n_rows, int n_cols, size_t n_tiles,
                                    const int32_t *values, uint8_t *blob, size_t blob_cap, uint64_t *offsets,
                                    uint8_t *predictors, int32_t *status);
gf_status gf_canon_decode_batch_i32(gf_context *ctx, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                    const uint64_t *offsets, int32_t *values, int32_t *status);
gf_status gf_canon_encode_batch_i32_dev(gf_context *ctx, void *stream, int codec_index, int n_rows, int n_cols,
                                        size_t n_tiles, const int32_t *d_values, uint8_t *d_out, size_t slot_stride,
                                        uint32_t *d_lengths, uint8_t *d_predictors, int32_t *d_status,
                                        int predictor_mask);
gf_status gf_canon_decode_batch_i32_dev(gf_context *ctx, void *stream, int n_rows, int n_cols, size_t n_tiles,
                                        const uint8_t *d_blob, size_t blob_bytes, const uint64_t *d_offsets,
                                        size_t slot_stride, const uint32_t *d_lengths, int32_t *d_values,
                                        int32_t *d_status);

/* ---- CodecDeflate (compress/CodecDeflate.java:108-228): predictor -> CodecM32 on the GPU, Deflate (level 6) on the host's
 * zlib, 10-byte header codec_index, predictor, seed LE, nM32 LE.  The encoder deflates the M32 stream of every applicable
 * predictor and keeps the strictly shortest packing (:176-199).  The predictor + M32 stage is available on its own:
 *   gf_m32_encode_batch_i32_dev: per tile three candidate streams (sub-slots of sub_stride bytes, order Differencing, Linear,
 *     Triangle; a tile with nulls has the DifferencingWithNulls stream in sub-slot 0), d_lengths[3 n], d_models[3 n]
 *     (predictor code, 0 = no candidate), d_seeds[n]; per-tile status GF_OVERFLOW when a stream is longer than sub_stride
 *     (its length is still exact).
 *   gf_m32_decode_batch_i32_dev: "raw" containers = 10-byte CodecDeflate/CodecHuffman header followed by the M32 bytes
 *     themselves -> tiles (the stage after Inflater.inflate, CodecDeflate.java:141-147); layout as gf_huffman_decode_batch_i32_dev. */
size_t gf_m32_default_stride(int n_rows, int n_cols);
size_t gf_m32_max_stream(int n_rows, int n_cols);
gf_status gf_m32_encode_batch_i32_dev(gf_context *ctx, void *stream, int n_rows, int n_cols, size_t n_tiles,
                                      const int32_t *d_values, uint8_t *d_streams, size_t sub_stride, uint32_t *d_lengths,
                                      uint8_t *d_models, uint32_t *d_seeds, int32_t *d_status);
gf_status gf_m32_decode_batch_i32_dev(gf_context *ctx, void *stream, int n_rows, int n_cols, size_t n_tiles,
                                      const uint8_t *d_blob, size_t blob_bytes, const uint64_t *d_offsets, size_t slot_stride,
                                      const uint32_t *d_lengths, int32_t *d_values, int32_t *d_status);
gf_status gf_deflate_encode_batch_i32(gf_context *ctx, int codec_index, int n_rows, int n_cols, size_t n_tiles,
                                      const int32_t *values, uint8_t *blob, size_t blob_cap, uint64_t *offsets,
                                      uint8_t *predictors, int32_t *status);
gf_status gf_deflate_decode_batch_i32(gf_context *ctx, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                      const uint64_t *offsets, int32_t *values, int32_t *status);
gf_status gf_deflate_encode_i32(gf_context *ctx, int codec_index, int n_rows, int n_cols, const int32_t *values,
                                uint8_t *out, size_t out_cap, size_t *out_len);
gf_status gf_deflate_decode_i32(gf_context *ctx, int n_rows, int n_cols, const uint8_t *packing, size_t packing_len,
                                int32_t *values);

/* ---- LSOP12 (lsop/LsEncoder12.java:122-219, lsop/LsDecoder12.java:94-160, lsop/LsOptimalPredictor12.java:109-383,
 * lsop/LsHeader.java:131-265, util/jama/LUDecomposition.java:70-134, 253-284): the optimal 12-coefficient linear
 * predictor.  Tiles need at least 6 rows and 6 columns (else GF_DECLINED, Java null); a singular system is GF_DECLINED too.
 *   residuals: per tile res_stride ints (>= gf_lsop12_residual_count): [4R+2C-9 initialisers | (R-2)(C-4) interior]
 *   coefs:     per tile 16 words: seed, the 12 float32 coefficients as bit patterns, 3 spare
 * The encoder writes the current container: header (LsHeader.packHeader) + CanonicalHuffman of the two integer streams
 * in one bit store (compression type 2).  The Deflate alternative (type 1, LsEncoder12.java:180-216) is produced by the
 * host entry points with the host's zlib when deflate_enabled != 0 (the reference's default).  Decoding accepts every
 * container the reference's decoder does (LsDecoder12.java:107-150), with either header revision: type 2 and type 0
 * (legacy Huffman of the two M32 streams, what Sample14_LSOP.gvrs holds) and type 1 (two zlib streams, inflated on the
 * device: the second one starts where the first Inflater stopped reading) -- all of them entirely on the GPU, in the host
 * and the _dev forms alike.  The _dev decode allocates inflate scratch inside the context on first use (not capture-safe
 * until it has run once for the shape).                                                                             */
size_t gf_lsop12_residual_count(int n_rows, int n_cols);
size_t gf_lsop12_max_packing(int n_rows, int n_cols);
/* LsEncoder12's two switches, as bits of the `deflate_enabled` / `flags` argument of the encode entry points:
 *   GF_LSOP_DEFLATE         setDeflateEnabled (lsop/LsEncoder12.java:92-94, default on; 1, as the argument always meant)
 *   GF_LSOP_VALUE_CHECKSUM  setValueChecksumEnabled (:117-119, default off): LsHeader.computeChecksum (LsHeader.java:391-406), the
 *                           CRC-32C of the tile's values as little-endian bytes, computed on the device, goes behind the
 *                           header's last field and bit 7 of byte 1 is set (LsHeader.packHeader :245-262).  The decoders skip
 *                           it, as the reference's decoder only prints a mismatch (LsDecoder12.java:153-158).                */
#define GF_LSOP_DEFLATE 1
#define GF_LSOP_VALUE_CHECKSUM 2
/* ABI note: `deflate_enabled` was a boolean before the value checksum existed (any non-zero value meant Deflate); it is a bit mask of
 * the two switches above now -- 2 is "checksum, no Deflate", not "Deflate" -- and any other bit is refused with GF_ERR_ARG by every
 * encode entry point (the _dev_ex form accepts GF_LSOP_DEFLATE and ignores it: Deflate needs the host's zlib).               */
/* LsOptimalPredictor12.encode: tile -> coefficients + residual streams (d_status: GF_OK / GF_DECLINED per tile) */
gf_status gf_lsop12_predict_dev(gf_context *ctx, void *stream, int n_rows, int n_cols, size_t n_tiles,
                                const int32_t *d_values, int32_t *d_residuals, size_t res_stride, uint32_t *d_coefs,
                                int32_t *d_status);
/* LsDecoder12.unpackInitializers + unpackInterior: residual streams -> tile.  d_in_status (may be NULL): tiles whose
 * entry is not GF_OK are skipped and that value is passed through to d_status.                                     */
gf_status gf_lsop12_reconstruct_dev(gf_context *ctx, void *stream, int n_rows, int n_cols, size_t n_tiles,
                                    const int32_t *d_residuals, size_t res_stride, const uint32_t *d_coefs,
                                    const int32_t *d_in_status, int32_t *d_values, int32_t *d_status);
/* device-resident batches; d_residuals / d_coefs / d_scratch_status (n_tiles ints) are caller-provided work buffers */
gf_status gf_lsop12_encode_batch_i32_dev(gf_context *ctx, void *stream, int codec_index, int n_rows, int n_cols,
                                         size_t n_tiles, const int32_t *d_values, uint8_t *d_out, size_t slot_stride,
                                         uint32_t *d_lengths, int32_t *d_status, int32_t *d_residuals, size_t res_stride,
                                         uint32_t *d_coefs, int32_t *d_scratch_status);
/* ... with flags (GF_LSOP_VALUE_CHECKSUM; the Deflate alternative is a host-side operation) */
gf_status gf_lsop12_encode_batch_i32_dev_ex(gf_context *ctx, void *stream, int codec_index, int n_rows, int n_cols,
                                            size_t n_tiles, const int32_t *d_values, int flags, uint8_t *d_out,
                                            size_t slot_stride, uint32_t *d_lengths, int32_t *d_status, int32_t *d_residuals,
                                            size_t res_stride, uint32_t *d_coefs, int32_t *d_scratch_status);
/* (the work buffers' contents behind a decode are the library's business: a tile's slot of d_residuals holds its initialisers and either
   its interior residuals as ints or -- tiles whose residuals are all bytes -- a byte plane in the order the reconstruction consumes it; words
   13 .. 15 of a tile's 16 words of d_coefs are the library's too) */
gf_status gf_lsop12_decode_batch_i32_dev(gf_context *ctx, void *stream, int n_rows, int n_cols, size_t n_tiles,
                                         const uint8_t *d_blob, size_t blob_bytes, const uint64_t *d_offsets,
                                         size_t slot_stride, const uint32_t *d_lengths, int32_t *d_values,
                                         int32_t *d_status, int32_t *d_residuals, size_t res_stride, uint32_t *d_coefs,
                                         int32_t *d_scratch_status);
/* host memory: replace LsEncoder12.encode / LsDecoder12.decode.  types[n_tiles] (optional): container type written */
gf_status gf_lsop12_encode_batch_i32(gf_context *ctx, int codec_index, int n_rows, int n_cols, size_t n_tiles,
                                     const int32_t *values, int deflate_enabled, uint8_t *blob, size_t blob_cap,
                                     uint64_t *offsets, uint8_t *types, int32_t *status);
gf_status gf_lsop12_decode_batch_i32(gf_context *ctx, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                     const uint64_t *offsets, int32_t *values, int32_t *status);
gf_status gf_lsop12_encode_i32(gf_context *ctx, int codec_index, int n_rows, int n_cols, const int32_t *values,
                               int deflate_enabled, uint8_t *out, size_t out_cap, size_t *out_len);
gf_status gf_lsop12_decode_i32(gf_context *ctx, int n_rows, int n_cols, const uint8_t *packing, size_t packing_len,
                               int32_t *values);

/* ---- CodecMaster (gvrs/CodecMaster.java:150-169, 195-203): a file's codec list in one batched call ------------------
 * codecs[k] names the k-th entry of the codec list (the standard list of gvrs/GvrsFileSpecification.java:221-230 is
 * {GF_CODEC_HUFFMAN, GF_CODEC_DEFLATE, GF_CODEC_NONE (CodecFloat), GF_CODEC_CANON_HUFFMAN}); k is the codec index written
 * to packing[0].  Encode: every integer codec of the list encodes the batch, per tile the strictly shortest non-null packing
 * wins, list order breaks ties; codec_used[t] = winning index (255: none).  Decode dispatches on packing[0]; an index outside
 * the list is GF_ERR_FORMAT (IOException "Invalid compression-type code").                                                */
#define GF_CODEC_NONE 0
#define GF_CODEC_HUFFMAN 1
#define GF_CODEC_DEFLATE 2
#define GF_CODEC_CANON_HUFFMAN 3
#define GF_CODEC_LSOP12 4
gf_status gf_codec_master_encode_batch_i32(gf_context *ctx, const int *codecs, int n_codecs, int n_rows, int n_cols,
                                           size_t n_tiles, const int32_t *values, uint8_t *blob, size_t blob_cap,
                                           uint64_t *offsets, uint8_t *codec_used, int32_t *status);
gf_status gf_codec_master_decode_batch_i32(gf_context *ctx, const int *codecs, int n_codecs, int n_rows, int n_cols,
                                           size_t n_tiles, const uint8_t *blob, const uint64_t *offsets, int32_t *values,
                                           int32_t *status);

/* ---- tile payloads (gvrs/RasterTile.java:234-256, gvrs/TileElementInt.java:196-219), tiles of one integer element:
 * per tile [int32 LE n][n bytes] = the CodecMaster packing, or the raw little-endian cells when no codec produced a packing or
 * it is not shorter than 4*cells bytes (codec_used[t] = 255).  This is what RecordManager.writeTile stores behind the tile
 * index (gvrs/RecordManager.java:417-431).  Decode treats an element of exactly 4*cells bytes as raw cells.               */
gf_status gf_tile_payload_encode_batch_i32(gf_context *ctx, const int *codecs, int n_codecs, int n_rows, int n_cols,
                                           size_t n_tiles, const int32_t *values, uint8_t *blob, size_t blob_cap,
                                           uint64_t *offsets, uint8_t *codec_used);
gf_status gf_tile_payload_decode_batch_i32(gf_context *ctx, const int *codecs, int n_codecs, int n_rows, int n_cols,
                                           size_t n_tiles, const uint8_t *blob, const uint64_t *offsets, int32_t *values,
                                           int32_t *status);

/* ---- read-ahead: gvrs/TileDecompressionAssistant.java:60-230 and its caller gvrs/RasterTileCache.java:339-426
 * as an N-tile prefetch queue.  The reference's assistant decodes ONE predicted tile on a background thread; this one takes
 * everything that is queued when it wakes up (at most max_batch tiles) and decodes it as one GPU batch through
 * gf_tile_payload_decode_batch_i32, on a context of its own on `device` (the application thread may use its own context
 * at the same time, as the reference's main thread uses its own CodecMaster).
 *   gf_readahead_submit   submitDecompression: `packing` = the element bytes RecordManager.readTilePacking returns (a
 *                         CodecMaster packing, or 4*cells raw little-endian bytes); copied, returns at once
 *   gf_readahead_pending  getPendingTaskCount: queued + in progress
 *   gf_readahead_take     getTilesWithWaitForIndex: waits while wait_index is queued or in progress, then hands over up to
 *                         max_tiles finished tiles (indices[i], values + i*cells, status[i]); the tile waited for comes first.
 *                         Tiles that do not fit stay for the next call.  A wait_index that was never submitted does not wait. */
typedef struct gf_readahead gf_readahead;
gf_status gf_readahead_create(int device, const int *codecs, int n_codecs, int n_rows, int n_cols, size_t max_batch,
                              gf_readahead **ra);
void gf_readahead_destroy(gf_readahead *ra);
gf_status gf_readahead_submit(gf_readahead *ra, int32_t tile_index, const uint8_t *packing, size_t len);
int gf_readahead_pending(gf_readahead *ra);
gf_status gf_readahead_take(gf_readahead *ra, int32_t wait_index, size_t max_tiles, int32_t *indices, int32_t *values,
                            int32_t *status, size_t *n_out);
size_t gf_readahead_cells(gf_readahead *ra);   /* n_rows * n_cols: gf_readahead_take writes that many ints per tile handed over */
void gf_readahead_counters(gf_readahead *ra, uint64_t *n_batches, uint64_t *n_tiles);   /* GPU batches run, tiles decoded */

/* ---- ICompressionDecoder.analyze for CodecHuffman (compress/CodecHuffman.java:172-234, compress/CodecStats.java:49-290):
 * the per-predictor statistics the reference gathers by decoding every packing on the CPU, from a GPU pass that
 * Huffman-decodes the batch and histograms the M32 bytes.  stats[0..4] by predictor code (None, Differencing, Linear,
 * Triangle, DifferencingWithNulls), stats[5] = "All Predictors"; the call ADDS to stats (zero it = clearAnalysisData).
 * The getters of CodecStats are ratios of these sums: bits/symbol = 8 n_bytes / n_symbols, entropy = sum_entropy_m32 /
 * n_m32_counted, ... (CodecStats.java:196-290).  status[t] != GF_OK: the reference's analyze would throw; not counted. */
typedef struct gf_codec_stats {
    int64_t n_tiles;          /* nTilesCounted      */
    int64_t n_bytes;          /* nBytesTotal        : packing bytes - 10 */
    int64_t n_symbols;        /* nSymbolsTotal      : cells */
    int64_t n_bits_overhead;  /* nBitsOverheadTotal : bits of the serialised Huffman tree */
    int64_t n_m32_counted;    /* nM32Counted        */
    int64_t sum_length_m32;   /* sumLengthM32       */
    int64_t sum_observed_m32; /* sumObservedM32     : distinct M32 byte values per tile, summed */
    double sum_entropy_m32;   /* sumEntropyM32      : zero-order entropy of the M32 bytes per tile, summed */
} gf_codec_stats;
gf_status gf_huffman_analyze_batch(gf_context *ctx, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                   const uint64_t *offsets, gf_codec_stats *stats, int32_t *status);
/* The same pass with the pair counts behind CodecStats.getH2 (CodecStats.java:63-64, 150-190): pair_counts = six tables of
 * 65536 int64, table k = sB of stats[k], entry (prior << 8) | value counts neighbouring M32 bytes; the call ADDS to them
 * (sA is the column sum of sB).  gf_codec_stats_h2(table) = getH2() of that CodecStats (natural logarithm, as there).   */
gf_status gf_huffman_analyze_batch_h2(gf_context *ctx, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                      const uint64_t *offsets, gf_codec_stats *stats, int64_t *pair_counts, int32_t *status);
double gf_codec_stats_h2(const int64_t *pair_table);

/* ---- tile records (gvrs/RecordManager.java:153-204, 217-262, 386-520; gvrs/TileElementInt.java:196-219,
 * gvrs/TileElementShort.java:211-250; util/GridfourCRC32C.java): what RecordManager.writeTile appends to the file for a
 * tile of one integer-coded element, for a whole batch of dirty tiles in one call (flush()):
 *   [int32 LE size, multiple of 8][type 2][0 0 0][int32 tileIndex][int32 n][n element bytes][zeros][CRC-32C | 0]
 * element bytes = CodecMaster packing, or the standard (raw little-endian) form when no codec is listed / produced a
 * packing / it is not shorter (codec_used[t] = 255).  GF_ELEM_SHORT: values are int16, widened with fill_value mapped to
 * INT4_NULL_CODE for the codecs; on decode INT4_NULL_CODE comes back as -32768 (TileElementShort.java:241-243).
 * Reproduces the tile records of the reference's sample files byte for byte (tests/test_gpu_records.py).             */
#define GF_ELEM_INT 0
#define GF_ELEM_SHORT 1
size_t gf_tile_record_max_bytes(int elem_type, int n_rows, int n_cols);
uint32_t gf_crc32c(const uint8_t *data, size_t n);
gf_status gf_tile_record_encode_batch(gf_context *ctx, const int *codecs, int n_codecs, int elem_type, int fill_value,
                                      int n_rows, int n_cols, size_t n_tiles, const int32_t *tile_indices, const void *values,
                                      int checksum_enabled, uint8_t *blob, size_t blob_cap, uint64_t *offsets,
                                      uint8_t *codec_used);
gf_status gf_tile_record_decode_batch(gf_context *ctx, const int *codecs, int n_codecs, int elem_type, int n_rows, int n_cols,
                                      size_t n_tiles, const uint8_t *blob, const uint64_t *offsets, int verify_checksum,
                                      int32_t *tile_indices, void *values, int32_t *status);

/* ---- CodecFloat (compress/CodecFloat.java:328-458): float32 tiles ---------------------------
 * The five byte planes (sign bits, exponent, three byte-delta coded mantissa bytes) are split and
 * merged on the GPU; the Deflate stage of each plane runs on the host's zlib (its bytes are defined
 * by zlib itself: the reference calls java.util.zip.Deflater(9), CodecFloat.java:268-283).
 * plane buffer of a tile: [sign ceil(n/8)] [exponent n] [m1 n] [m2 n] [m3 n], n = n_rows*n_cols.   */
size_t gf_float_planes_bytes(int n_rows, int n_cols);
gf_status gf_float_planes_encode_dev(gf_context *ctx, void *stream, int n_rows, int n_cols, size_t n_tiles,
                                     const float *d_values, uint8_t *d_planes, size_t plane_stride);
gf_status gf_float_planes_decode_dev(gf_context *ctx, void *stream, int n_rows, int n_cols, size_t n_tiles,
                                     const uint8_t *d_planes, size_t plane_stride, float *d_values);
/* replaces ICompressionEncoder.encodeFloats / ICompressionDecoder.decodeFloats as implemented by
 * CodecFloat, for one tile or a batch in host memory.  zlib_level: 9 = current reference source,
 * 6 = what the reference's sample files were written with.  offsets[n_tiles+1] as for the int path. */
gf_status gf_float_encode_batch_f32(gf_context *ctx, int codec_index, int n_rows, int n_cols, size_t n_tiles,
                                    const float *values, int zlib_level, uint8_t *blob, size_t blob_cap,
                                    uint64_t *offsets);
gf_status gf_float_decode_batch_f32(gf_context *ctx, int n_rows, int n_cols, size_t n_tiles, const uint8_t *blob,
                                    const uint64_t *offsets, float *values, int32_t *status);
gf_status gf_float_encode_f32(gf_context *ctx, int codec_index, int n_rows, int n_cols, const float *values,
                              int zlib_level, uint8_t *out, size_t out_cap, size_t *out_len);
gf_status gf_float_decode_f32(gf_context *ctx, int n_rows, int n_cols, const uint8_t *packing, size_t packing_len,
                              float *values);

/* ---- synthetic elevation tiles (bench / tests) ---- */
/* fills n_tiles tiles of a seeded integer value-noise DEM cut into
 * n_rows x n_cols tiles, tiles_per_row tiles across, starting at tile0.      */
gf_status gf_synth_dem_dev(gf_context *ctx, void *stream, uint64_t seed, int n_rows, int n_cols,
                           int64_t tiles_per_row, int64_t tile0, size_t n_tiles,
                           int32_t *d_values);
/* the nulls workload of SURVEY.md section 8d: the same grid with an "ocean mask" -- mask_per_mille / 1000 of its
 * 16 x 16 blocks hold GF_INT4_NULL_CODE, so that nearly every tile takes PredictorModelDifferencingWithNulls
 * (compress/CodecHuffman.java:73-98, PredictorModelDifferencingWithNulls.java:66-166)                            */
gf_status gf_synth_dem_masked_dev(gf_context *ctx, void *stream, uint64_t seed, int n_rows, int n_cols,
                                  int64_t tiles_per_row, int64_t tile0, size_t n_tiles, int mask_per_mille,
                                  int32_t *d_values);
/* the same grid as another kind of surface.  GF_DEM_STYLE_ROUGH: provinces of mountains / plains / stripes so that each of
 * PredictorModelDifferencing / Linear / Triangle wins a share of the tiles (compress/CodecHuffman.java:100-110), and cliff and
 * scree blocks whose residuals need two and three M32 bytes (compress/CodecM32.java:270-311) -- the data SURVEY.md section 8d
 * describes ("mostly within +-126 with a tail into 2-3 byte codes"); GF_DEM_STYLE_CLASSIC is gf_synth_dem_masked_dev.        */
#define GF_DEM_STYLE_CLASSIC 0
#define GF_DEM_STYLE_ROUGH 1
gf_status gf_synth_dem_style_dev(gf_context *ctx, void *stream, uint64_t seed, int n_rows, int n_cols,
                                 int64_t tiles_per_row, int64_t tile0, size_t n_tiles, int mask_per_mille, int style,
                                 int32_t *d_values);

/* ---- thin device-memory helpers so that non-HIP hosts (JNI, ctypes) can
 *      stage data without linking the HIP runtime themselves --------------- */
gf_status gf_dev_malloc(gf_context *ctx, size_t bytes, void **d_ptr);
gf_status gf_dev_free(gf_context *ctx, void *d_ptr);
gf_status gf_dev_memset(gf_context *ctx, void *d_ptr, int value, size_t bytes);
gf_status gf_dev_upload(gf_context *ctx, void *d_dst, const void *h_src, size_t bytes);
gf_status gf_dev_download(gf_context *ctx, void *h_dst, const void *d_src, size_t bytes);

/* ---- timing of the device entry points with HIP events on `stream` ------
 * gf_timer_* bracket any sequence of *_dev calls; elapsed is in milliseconds. */
typedef struct gf_timer gf_timer;
gf_status gf_timer_create(gf_context *ctx, gf_timer **t);
void gf_timer_destroy(gf_timer *t);
gf_status gf_timer_start(gf_timer *t, void *stream);
gf_status gf_timer_stop(gf_timer *t, void *stream);
gf_status gf_timer_elapsed_ms(gf_timer *t, float *ms);   /* synchronises on the stop event */

#ifdef __cplusplus
}
#endif
#endif
