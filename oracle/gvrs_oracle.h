/*
 * gvrs_oracle.h -- CPU restatement ("oracle") of the Gridfour GVRS tile-codec
 * hot path.
 *
 * THIS IS TEST INFRASTRUCTURE.  It is the parity checker for the HIP codec and
 * the `cpu_baseline` leg of bench.py.  Nothing in the product path
 * (gridfour_amd/, include/) may call, link or import it.
 *
 * Parity pinning: the restatement is checked in tests/test_oracle_golden.py
 * against the reference's own binary fixtures
 *   core/src/test/resources/org/gridfour/gvrs/SampleFiles/Sample05_IntComp.gvrs
 *   .../Sample06_FltComp.gvrs, .../Sample14_LSOP.gvrs
 * (copied as data under tests/golden/ref_samples/), the M32 known-answer table
 * of core/src/test/java/org/gridfour/compress/CodecM32Test.java:95-112 and the
 * byte examples of CodecM32.java:82-89.  What no reference fixture pins is
 * listed in DESIGN.md ("unpinned").
 *
 * Each function cites the reference file:line it follows.  Paths are relative
 * to core/src/main/java/org/gridfour/ in the reference repository.
 */
#ifndef GVRS_ORACLE_H
#define GVRS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVO_OK 0
#define GVO_DECLINED 1        /* Java encoder would return null            */
#define GVO_ERR_FORMAT (-1)   /* Java decoder would throw IOException      */
#define GVO_ERR_BOUNDS (-2)   /* Java would throw ArrayIndexOutOfBounds    */
#define GVO_ERR_CAPACITY (-3) /* caller's output buffer too small          */
#define GVO_ERR_ARG (-4)

#define GVO_INT4_NULL ((int32_t)0x80000000) /* util/GridfourConstants.java:61 */

/* predictor codes, compress/PredictorModelType.java:46-63 */
enum { GVO_PM_NONE = 0, GVO_PM_DIFFERENCING = 1, GVO_PM_LINEAR = 2,
       GVO_PM_TRIANGLE = 3, GVO_PM_DIFFERENCING_NULLS = 4 };

/* ---- CodecM32 (compress/CodecM32.java:257-311, 327-356) ---- */
/* appends the M32 form of value at out, returns the number of bytes (1..6) */
int gvo_m32_encode(int32_t value, uint8_t *out);
/* decodes one value at buf[*pos], advances *pos; no bounds checks (as the reference) */
int32_t gvo_m32_decode(const uint8_t *buf, size_t *pos);

/* ---- predictors (compress/PredictorModel*.java) ----
 * encode: returns number of M32 bytes written to out (capacity 6*nRows*nCols),
 *         -1 when the model declines (Triangle with <2 rows/cols), seed in *seed.
 * decode: fills values[nRows*nCols] from the M32 bytes.                     */
int gvo_predictor_encode(int model, int nRows, int nCols, const int32_t *values,
                         uint8_t *out, int32_t *seed);
int gvo_predictor_decode(int model, int32_t seed, int nRows, int nCols,
                         const uint8_t *m32, size_t nM32, int32_t *values);

/* ---- legacy Huffman (compress/HuffmanEncoder.java:124-305,
 *      compress/HuffmanDecoder.java:65-187) over a bit store
 *      (io/BitOutputStore.java, io/BitInputStore.java: LSB-first) ---- */
/* Appends tree+text to a zeroed bit buffer `bits` starting at *bitPos.
 * capBits = capacity in bits. Returns GVO_OK or GVO_ERR_CAPACITY.
 * codeLen256/treeBits are optional diagnostics (may be NULL).               */
int gvo_huffman_encode(uint8_t *bits, size_t capBits, size_t *bitPos,
                       const uint8_t *symbols, size_t nSymbols,
                       uint8_t *codeLen256, size_t *treeBits);
/* Decodes nSymbols starting at *bitPos from a buffer of nBitsTotal bits.    */
int gvo_huffman_decode(const uint8_t *bits, size_t nBitsTotal, size_t *bitPos,
                       uint8_t *symbols, size_t nSymbols);

/* ---- CodecHuffman (compress/CodecHuffman.java:70-153) ---- */
/* predictorMask: bit (model-1) set = model may be tried; 0xF = reference
 * behaviour (all registered models).  *predictorUsed receives the model of
 * the returned packing.  Returns GVO_OK, GVO_DECLINED (null), or error.     */
int gvo_codec_huffman_encode(int codecIndex, int nRows, int nCols,
                             const int32_t *values, uint8_t *out, size_t outCap,
                             size_t *outLen, int predictorMask, int *predictorUsed);
int gvo_codec_huffman_decode(int nRows, int nCols, const uint8_t *packing,
                             size_t len, int32_t *values);
/* worst-case packing size for a tile with nCells cells */
size_t gvo_codec_huffman_bound(size_t nCells);

/* ---- CodecDeflate (compress/CodecDeflate.java:108-228), zlib level 6 ---- */
int gvo_codec_deflate_encode(int codecIndex, int nRows, int nCols,
                             const int32_t *values, uint8_t *out, size_t outCap,
                             size_t *outLen, int *predictorUsed);
int gvo_codec_deflate_decode(int nRows, int nCols, const uint8_t *packing,
                             size_t len, int32_t *values);

/* ---- CodecFloat (compress/CodecFloat.java:300-458) ---- */
/* the five byte planes before Deflate: sign bits (ceil(n/8)), exponent (n),
 * delta-coded mantissa hi/mid/lo (n each).  planes must hold ceil(n/8)+4n.  */
int gvo_float_planes_encode(int nRows, int nCols, const uint32_t *rawBits,
                            uint8_t *planes);
int gvo_float_planes_decode(int nRows, int nCols, const uint8_t *planes,
                            uint32_t *rawBits);
/* full codec; level = zlib level (reference source: 9; sample files: 6)     */
int gvo_codec_float_encode(int codecIndex, int nRows, int nCols,
                           const uint32_t *rawBits, int level, uint8_t *out,
                           size_t outCap, size_t *outLen);
int gvo_codec_float_decode(int nRows, int nCols, const uint8_t *packing,
                           size_t len, uint32_t *rawBits);

/* ---- canonical Huffman (compress/canonicalHuffman/ *.java) -- PARITY UNPINNED, see gvrs_oracle_canon.c ---- */
/* CanonicalHuffman.encode :177-283: appends code tables + text + end-of-text to a zeroed bit buffer at
 * *bitPos.  codeLen260 (optional) receives the 260 code lengths.                                     */
int gvo_canon_encode(uint8_t *bits, size_t capBits, size_t *bitPos, const int32_t *text, size_t nSymbols,
                     uint8_t *codeLen260);
/* CanonicalHuffman.decode :441-519: reads until the end-of-text symbol; text has room for nSymbolsInText */
int gvo_canon_decode(const uint8_t *bits, size_t nBitsTotal, size_t *bitPos, int32_t *text,
                     size_t nSymbolsInText, size_t *nDecoded);
/* IPredictorModel.encodeInt / decodeInt: integer residual streams (no M32).  encode returns the
 * number of residuals, -1 = model declines (Triangle), -2 = Java would index out of bounds.          */
int gvo_predictor_encode_int(int model, int nRows, int nCols, const int32_t *values, int32_t *out, int32_t *seed);
int gvo_predictor_decode_int(int model, int32_t seed, int nRows, int nCols, const int32_t *residuals,
                             int32_t *values);
/* CodecCanonHuffman.encode :70-160 / decode :163-195 (the default integer codec of current Gridfour)  */
int gvo_codec_canon_encode(int codecIndex, int nRows, int nCols, const int32_t *values, uint8_t *out,
                           size_t outCap, size_t *outLen, int predictorMask, int *predictorUsed);
int gvo_codec_canon_decode(int nRows, int nCols, const uint8_t *packing, size_t len, int32_t *values);
size_t gvo_codec_canon_bound(size_t nCells);
int gvo_batch_canon_encode(int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                           uint8_t *out, size_t stride, uint32_t *lengths, uint8_t *predictors);
int gvo_batch_canon_decode(int nRows, int nCols, size_t nTiles, const uint8_t *packings, size_t stride,
                           const uint32_t *lengths, int32_t *values);

/* ---- LSOP12 (lsop/LsOptimalPredictor12.java, LsEncoder12.java, LsDecoder12.java, LsHeader.java,
 *      util/jama/LUDecomposition.java); see gvrs_oracle_lsop.c for what Sample14_LSOP.gvrs pins ---- */
int gvo_lsop12_coefficients(int nRows, int nCols, const int32_t *values, float *u12);
int gvo_lsop12_residuals(int nRows, int nCols, const int32_t *values, int32_t *seed, float *u12,
                         int32_t *initInt, int32_t *interiorInt);
int gvo_lsop12_encode(int codecIndex, int nRows, int nCols, const int32_t *values, int deflateEnabled,
                      uint8_t *out, size_t outCap, size_t *outLen, int *containerType);
/* ... with LsEncoder12.setValueChecksumEnabled :117-119 */
int gvo_lsop12_encode_ex(int codecIndex, int nRows, int nCols, const int32_t *values, int deflateEnabled, int checksumEnabled,
                         uint8_t *out, size_t outCap, size_t *outLen, int *containerType);
/* util/GridfourCRC32C.java:160-185; LsHeader.computeChecksum :391-406 */
uint32_t gvo_crc32c(const uint8_t *b, size_t n);
uint32_t gvo_lsop_value_checksum(int nRows, int nCols, const int32_t *values);
int gvo_lsop12_decode(int nRows, int nCols, const uint8_t *packing, size_t len, int32_t *values);
int gvo_lsop12_encode_legacy_huffman(int codecIndex, int nRows, int nCols, const int32_t *values,
                                     uint8_t *out, size_t outCap, size_t *outLen);
size_t gvo_lsop12_bound(size_t nCells);
int gvo_batch_lsop12_encode(int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                            int deflateEnabled, uint8_t *out, size_t stride, uint32_t *lengths, uint8_t *types);
int gvo_batch_lsop12_decode(int nRows, int nCols, size_t nTiles, const uint8_t *packings, size_t stride,
                            const uint32_t *lengths, int32_t *values);

/* ---- batch helpers used by the CPU baseline (plain loops over tiles) ---- */
/* tiles are contiguous, nRows*nCols each.  out slots have `stride` bytes.   */
int gvo_batch_huffman_encode(int codecIndex, int nRows, int nCols, size_t nTiles,
                             const int32_t *values, uint8_t *out, size_t stride,
                             uint32_t *lengths, uint8_t *predictors);
int gvo_batch_huffman_decode(int nRows, int nCols, size_t nTiles,
                             const uint8_t *packings, size_t stride,
                             const uint32_t *lengths, int32_t *values);
/* encode + decode of every tile on nThreads native threads (contiguous shares); seconds[2] = wall time of the two phases */
int gvo_huffman_roundtrip_threads(int nThreads, int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                                  uint8_t *out, size_t stride, uint32_t *lengths, int32_t *back, double *seconds);


/* ---- deterministic synthetic DEM, integer only ---- */
uint64_t gvo_splitmix64(uint64_t x);
/* value of the synthetic elevation field at grid cell (gx, gy)              */
int32_t gvo_dem_value(uint64_t seed, int64_t gx, int64_t gy);
/* fills nTiles tiles (tile t = tile row t / tilesPerRow, col t % tilesPerRow
 * of a grid cut into nRows x nCols tiles), starting at tile index tile0.     */
int32_t gvo_dem_value_style(uint64_t seed, int64_t gx, int64_t gy, int style);   /* style 1: the rough surface (provinces, cliffs) */
void gvo_dem_fill_tiles_style(uint64_t seed, int nRows, int nCols, int64_t tilesPerRow,
                              int64_t tile0, int64_t nTiles, int maskPerMille, int style, int32_t *values);
void gvo_dem_fill_tiles(uint64_t seed, int nRows, int nCols, int64_t tilesPerRow,
                        int64_t tile0, int64_t nTiles, int32_t *values);
/* the same with an ocean mask: 16 x 16 blocks of GF null codes, maskPerMille / 1000 of the blocks */
void gvo_dem_fill_tiles_masked(uint64_t seed, int nRows, int nCols, int64_t tilesPerRow,
                               int64_t tile0, int64_t nTiles, int maskPerMille, int32_t *values);

#ifdef __cplusplus
}
#endif
#endif
