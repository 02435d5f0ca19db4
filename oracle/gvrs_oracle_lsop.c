/*
 * gvrs_oracle_lsop.c -- CPU restatement of the LSOP12 codec (optimal 12-coefficient predictor).
 * TEST INFRASTRUCTURE: see gvrs_oracle.h.
 *
 * Pinned by the reference fixture Sample14_LSOP.gvrs (tests/test_oracle_lsop.py): decode of the legacy
 * container reproduces the analytic surface the fixture was written from; recomputed coefficients equal
 * the 12 stored floats bit for bit; re-encoded initialiser/interior M32 streams Huffman-coded by the
 * legacy encoder reproduce the stored bytes.  The CURRENT container (canonical Huffman, type 2) and the
 * Deflate container (type 1) are PARITY UNPINNED: no fixture holds them.
 *
 * Paths are relative to core/src/main/java/org/gridfour/ in the reference repository.
 */
#include "gvrs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* StrictMath.round(float) (Java 7+): closest int, ties toward +infinity, NaN -> 0, saturating */
static int32_t java_round_float(float p)
{
    if (p != p) return 0;
    double f = floor((double)p + 0.5);                  /* exact: float + 0.5 is representable in double */
    if (f >= 2147483647.0) return 2147483647;
    if (f <= -2147483648.0) return (int32_t)0x80000000;
    return (int32_t)f;
}

static inline int32_t lo32(int64_t x) { return (int32_t)(uint32_t)(uint64_t)x; }

/* lsop/LsOptimalPredictor12.java:311-383 + util/jama/LUDecomposition.java:70-134, 253-284.
 * Returns GVO_OK and u[12] (float32 casts of the FP64 solution), or GVO_DECLINED (Java returns null). */
int gvo_lsop12_coefficients(int nRows, int nCols, const int32_t *v, float *u)
{
    if (nRows < 6 || nCols < 6) return GVO_DECLINED;
    double z[13], s[13], c[13][13];
    memset(s, 0, sizeof s);
    memset(c, 0, sizeof c);
    for (int r = 2; r < nRows; r++) {
        for (int col = 2; col < nCols - 2; col++) {
            int idx = r * nCols + col;
            z[0] = v[idx];
            z[1] = v[idx - 1];
            z[2] = v[idx - nCols - 1];
            z[3] = v[idx - nCols];
            z[4] = v[idx - nCols + 1];
            z[5] = v[idx - nCols + 2];
            z[6] = v[idx - 2];
            z[7] = v[idx - nCols - 2];
            z[8] = v[idx - 2 * nCols - 2];
            z[9] = v[idx - 2 * nCols - 1];
            z[10] = v[idx - 2 * nCols];
            z[11] = v[idx - 2 * nCols + 1];
            z[12] = v[idx - 2 * nCols + 2];
            for (int i = 0; i < 13; i++) s[i] += z[i];
            for (int i = 0; i < 13; i++)
                for (int j = i; j < 13; j++) c[i][j] += z[i] * z[j];
        }
    }
    for (int i = 1; i < 13; i++)
        for (int j = 0; j < i; j++) c[i][j] = c[j][i];
    double LU[13][13], X[13];
    memset(LU, 0, sizeof LU);
    for (int i = 1; i < 13; i++) {
        for (int j = 1; j < 13; j++) LU[i - 1][j - 1] = c[i][j];
        LU[i - 1][12] = s[i];
    }
    for (int j = 1; j < 13; j++) LU[12][j - 1] = s[j];
    double b[13];
    for (int i = 1; i < 13; i++) b[i - 1] = c[0][i];
    b[12] = s[0];
    /* LUDecomposition constructor */
    const int m = 13, n = 13;
    int piv[13];
    for (int i = 0; i < m; i++) piv[i] = i;
    double LUcolj[13];
    for (int j = 0; j < n; j++) {
        for (int i = 0; i < m; i++) LUcolj[i] = LU[i][j];
        for (int i = 0; i < m; i++) {
            int kmax = i < j ? i : j;
            double sum = 0.0;
            for (int k = 0; k < kmax; k++) sum += LU[i][k] * LUcolj[k];
            LUcolj[i] -= sum;
            LU[i][j] = LUcolj[i];
        }
        int p = j;
        for (int i = j + 1; i < m; i++)
            if (fabs(LUcolj[i]) > fabs(LUcolj[p])) p = i;
        if (p != j) {
            for (int k = 0; k < n; k++) { double t = LU[p][k]; LU[p][k] = LU[j][k]; LU[j][k] = t; }
            int k = piv[p]; piv[p] = piv[j]; piv[j] = k;
        }
        if (LU[j][j] != 0.0)
            for (int i = j + 1; i < m; i++) LU[i][j] /= LU[j][j];
    }
    for (int j = 0; j < n; j++)
        if (LU[j][j] == 0) return GVO_DECLINED;           /* "Matrix is singular." -> null */
    for (int i = 0; i < 13; i++) X[i] = b[piv[i]];
    for (int k = 0; k < n; k++)
        for (int i = k + 1; i < n; i++) X[i] -= X[k] * LU[i][k];
    for (int k = n - 1; k >= 0; k--) {
        X[k] /= LU[k][k];
        for (int i = 0; i < k; i++) X[i] -= X[k] * LU[i][k];
    }
    for (int i = 0; i < 12; i++) u[i] = (float)X[i];
    return GVO_OK;
}

/* float32 prediction of one interior cell, LsOptimalPredictor12.java:254-267 */
static inline float lsop_predict(const float *u, const int32_t *v, int idx, int nC)
{
    float p = u[0] * (float)v[idx - 1]
        + u[1] * (float)v[idx - nC - 1]
        + u[2] * (float)v[idx - nC]
        + u[3] * (float)v[idx - nC + 1]
        + u[4] * (float)v[idx - nC + 2]
        + u[5] * (float)v[idx - 2]
        + u[6] * (float)v[idx - nC - 2]
        + u[7] * (float)v[idx - 2 * nC - 2]
        + u[8] * (float)v[idx - 2 * nC - 1]
        + u[9] * (float)v[idx - 2 * nC]
        + u[10] * (float)v[idx - 2 * nC + 1]
        + u[11] * (float)v[idx - 2 * nC + 2];
    return p;
}

static inline int32_t tri_resid(const int32_t *v, int idx, int nC)
{
    int64_t a = v[idx - 1], b = v[idx - nC - 1], c = v[idx - nC], t = v[idx];
    return lo32(t - ((a + c) - b));
}

/* LsOptimalPredictor12.encode :109-292.  initInt has 4*nRows+2*nCols-9 entries, interiorInt
 * (nRows-2)*(nCols-4).  Returns GVO_OK / GVO_DECLINED. */
int gvo_lsop12_residuals(int nRows, int nCols, const int32_t *v, int32_t *seed, float *u, int32_t *initInt,
                         int32_t *interiorInt)
{
    if (nRows < 6 || nCols < 6) return GVO_DECLINED;
    int k = 0;
    *seed = v[0];
    int64_t prior = v[0];
    for (int i = 1; i < nCols; i++) { initInt[k++] = lo32((int64_t)v[i] - prior); prior = v[i]; }
    prior = v[0];
    for (int i = 1; i < nRows; i++) { initInt[k++] = lo32((int64_t)v[i * nCols] - prior); prior = v[i * nCols]; }
    for (int i = 1; i < nCols; i++) initInt[k++] = tri_resid(v, nCols + i, nCols);
    for (int i = 2; i < nRows; i++) initInt[k++] = tri_resid(v, i * nCols + 1, nCols);
    for (int i = 2; i < nRows; i++) {
        int idx = i * nCols + nCols - 2;
        initInt[k++] = tri_resid(v, idx, nCols);
        initInt[k++] = tri_resid(v, idx + 1, nCols);
    }
    int rc = gvo_lsop12_coefficients(nRows, nCols, v, u);
    if (rc != GVO_OK) return rc;
    k = 0;
    for (int r = 2; r < nRows; r++)
        for (int c = 2; c < nCols - 2; c++) {
            int idx = r * nCols + c;
            int32_t estimate = java_round_float(lsop_predict(u, v, idx, nCols));
            interiorInt[k++] = (int32_t)((uint32_t)v[idx] - (uint32_t)estimate);
        }
    return GVO_OK;
}

static size_t m32_pack(const int32_t *x, size_t n, uint8_t *out)
{
    size_t k = 0;
    for (size_t i = 0; i < n; i++) k += (size_t)gvo_m32_encode(x[i], out + k);
    return k;
}

static void put_i32(uint8_t *p, uint32_t x) { p[0] = (uint8_t)x; p[1] = (uint8_t)(x >> 8); p[2] = (uint8_t)(x >> 16); p[3] = (uint8_t)(x >> 24); }
static uint32_t get_i32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* util/GridfourCRC32C.java:160-167 (update: table-driven, reflected Castagnoli polynomial; the table :81-146 is the one this
 * loop generates) over a whole buffer, :183-185 getValue. */
uint32_t gvo_crc32c(const uint8_t *b, size_t n)
{
    static uint32_t table[256];
    static int ready = 0;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? 0x82F63B78u : 0u);
            table[i] = c;
        }
        ready = 1;
    }
    uint32_t crc = 0;
    crc ^= 0xffffffffu;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ b[i]) & 0xffu] ^ (crc >> 8);
    crc ^= 0xffffffffu;
    return crc;
}

/* LsHeader.computeChecksum :391-406: CRC-32C of the values as little-endian bytes */
uint32_t gvo_lsop_value_checksum(int nRows, int nCols, const int32_t *values)
{
    const size_t n = (size_t)nRows * (size_t)nCols;
    uint8_t *b = malloc(n * 4 + 1);
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t v = (uint32_t)values[i];
        b[k++] = (uint8_t)(v & 0xff);
        b[k++] = (uint8_t)((v >> 8) & 0xff);
        b[k++] = (uint8_t)((v >> 16) & 0xff);
        b[k++] = (uint8_t)((v >> 24) & 0xff);
    }
    const uint32_t c = gvo_crc32c(b, k);
    free(b);
    return c;
}

/* LsHeader.packHeader :210-265; checksumIncluded: LsEncoder12.setValueChecksumEnabled :117-119 (default off, :80) */
static size_t pack_header(uint8_t *out, int codecIndex, int32_t seed, const float *u, uint32_t nInit, uint32_t nInterior,
                          int type, int checksumIncluded, uint32_t checksum)
{
    out[0] = (uint8_t)codecIndex;
    out[1] = (uint8_t)(type | 0x40 | (checksumIncluded ? 0x80 : 0));
    out[2] = 12;
    put_i32(out + 3, (uint32_t)seed);
    size_t o = 7;
    for (int i = 0; i < 12; i++) { uint32_t b; memcpy(&b, &u[i], 4); put_i32(out + o, b); o += 4; }
    if (type != 2) { put_i32(out + o, nInit); o += 4; put_i32(out + o, nInterior); o += 4; }
    if (checksumIncluded) { put_i32(out + o, checksum); o += 4; }
    return o;
}

static int zdeflate6(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *outLen)
{
    z_stream s;
    memset(&s, 0, sizeof s);
    if (deflateInit(&s, 6) != Z_OK) return GVO_ERR_ARG;
    s.next_in = (Bytef *)in; s.avail_in = (uInt)n; s.next_out = out; s.avail_out = (uInt)cap;
    deflate(&s, Z_FINISH);                                /* Deflater.finish(); deflate(.., FULL_FLUSH) */
    *outLen = s.total_out;
    deflateEnd(&s);
    return GVO_OK;
}

size_t gvo_lsop12_bound(size_t nCells) { return 64 + gvo_codec_canon_bound(nCells) + 1024; }

/* LsEncoder12.encode :122-219.  deflateEnabled: setDeflateEnabled (default true).
 * *containerType receives 2 (canonical Huffman) or 1 (Deflate). */
int gvo_lsop12_encode(int codecIndex, int nRows, int nCols, const int32_t *values, int deflateEnabled, uint8_t *out,
                      size_t outCap, size_t *outLen, int *containerType)
{
    return gvo_lsop12_encode_ex(codecIndex, nRows, nCols, values, deflateEnabled, 0, out, outCap, outLen, containerType);
}

/* ... with LsEncoder12.setValueChecksumEnabled (:117-119): the checksum is computed from the raw values (:127-131) and goes
 * into either header (:136-146, :202-211) */
int gvo_lsop12_encode_ex(int codecIndex, int nRows, int nCols, const int32_t *values, int deflateEnabled, int checksumEnabled,
                         uint8_t *out, size_t outCap, size_t *outLen, int *containerType)
{
    if (nRows < 6 || nCols < 6) return GVO_DECLINED;
    const size_t nInit = (size_t)nRows * 4 + (size_t)nCols * 2 - 9, nInt = (size_t)(nRows - 2) * (size_t)(nCols - 4);
    const size_t nCells = (size_t)nRows * (size_t)nCols;
    int32_t *initInt = malloc(sizeof(int32_t) * nInit), *interior = malloc(sizeof(int32_t) * nInt);
    float u[12];
    int32_t seed;
    int rc = gvo_lsop12_residuals(nRows, nCols, values, &seed, u, initInt, interior);
    const uint32_t checksum = checksumEnabled && rc == GVO_OK ? gvo_lsop_value_checksum(nRows, nCols, values) : 0u;
    uint8_t *canon = NULL, *mInit = NULL, *mInt = NULL, *zInit = NULL, *zInt = NULL;
    if (rc == GVO_OK) {
        const size_t cap = gvo_lsop12_bound(nCells);
        canon = calloc(cap, 1);
        size_t hdr = pack_header(canon, codecIndex, seed, u, 0, 0, 2, checksumEnabled, checksum);
        size_t bitPos = hdr * 8;
        rc = gvo_canon_encode(canon, cap * 8, &bitPos, initInt, nInit, NULL);
        if (rc == GVO_OK) rc = gvo_canon_encode(canon, cap * 8, &bitPos, interior, nInt, NULL);  /* same bit store :152-153 */
        size_t total = (bitPos + 7) / 8, canonLength = total - hdr;
        int type = 2;
        if (rc == GVO_OK && deflateEnabled) {
            mInit = malloc(6 * nInit + 8); mInt = malloc(6 * nInt + 8);
            size_t nMI = m32_pack(initInt, nInit, mInit), nMX = m32_pack(interior, nInt, mInt);
            zInt = malloc(nMX + 128); zInit = malloc(nMI + 128);
            size_t insideN = 0, initN = 0;
            zdeflate6(mInt, nMX, zInt, nMX + 128, &insideN);
            if (!(insideN <= 0 || insideN >= canonLength)) {
                zdeflate6(mInit, nMI, zInit, nMI + 128, &initN);
                if (!(initN <= 0 || initN + insideN >= canonLength)) {
                    uint8_t h[80];
                    size_t hl = pack_header(h, codecIndex, seed, u, (uint32_t)nMI, (uint32_t)nMX, 1, checksumEnabled, checksum);
                    total = hl + initN + insideN;
                    if (total <= outCap) {
                        memcpy(out, h, hl);
                        memcpy(out + hl, zInit, initN);
                        memcpy(out + hl + initN, zInt, insideN);
                    }
                    type = 1;
                }
            }
        }
        if (rc == GVO_OK) {
            if (total > outCap) rc = GVO_ERR_CAPACITY;
            else if (type == 2) memcpy(out, canon, total);
            *outLen = total;
            if (containerType) *containerType = type;
        }
    }
    free(initInt); free(interior); free(canon); free(mInit); free(mInt); free(zInit); free(zInt);
    return rc;
}

/* LsDecoder12.unpackInitializers (int form) :186-221 + unpackInterior :311-383 */
static void lsop_unpack(const int32_t *initInt, const int32_t *interior, int32_t seed, const float *u, int nRows, int nCols,
                        int32_t *v)
{
    int k = 0;
    v[0] = seed;
    int32_t acc = seed;
    for (int i = 1; i < nCols; i++) { acc = (int32_t)((uint32_t)acc + (uint32_t)initInt[k++]); v[i] = acc; }
    acc = seed;
    for (int i = 1; i < nRows; i++) { acc = (int32_t)((uint32_t)acc + (uint32_t)initInt[k++]); v[i * nCols] = acc; }
    for (int i = 1; i < nCols; i++) {
        int idx = nCols + i;
        int64_t a = v[idx - 1], b = v[idx - nCols - 1], c = v[idx - nCols];
        v[idx] = lo32((int64_t)initInt[k++] + ((a + c) - b));
    }
    for (int i = 2; i < nRows; i++) {
        int idx = i * nCols + 1;
        int64_t a = v[idx - 1], b = v[idx - nCols - 1], c = v[idx - nCols];
        v[idx] = lo32((int64_t)initInt[k++] + ((a + c) - b));
    }
    int ki = 0;
    for (int r = 2; r < nRows; r++) {
        for (int col = 2; col < nCols - 2; col++) {
            int idx = r * nCols + col;
            int32_t estimate = java_round_float(lsop_predict(u, v, idx, nCols));
            v[idx] = (int32_t)((uint32_t)estimate + (uint32_t)interior[ki++]);
        }
        int idx = r * nCols + nCols - 2;
        for (int q = 0; q < 2; q++, idx++) {
            int64_t a = v[idx - 1], b = v[idx - nCols - 1], c = v[idx - nCols];
            v[idx] = lo32((int64_t)initInt[k++] + ((a + c) - b));
        }
    }
}

static int m32_unpack(const uint8_t *m, size_t nBytes, int32_t *out, size_t nValues)
{
    /* CodecM32.decode has no bounds checks of its own; the byte[] does (AIOOBE) */
    uint8_t *pad = calloc(nBytes + 8, 1);
    memcpy(pad, m, nBytes);
    size_t pos = 0;
    int rc = GVO_OK;
    for (size_t i = 0; i < nValues; i++) {
        if (pos >= nBytes) { rc = GVO_ERR_BOUNDS; break; }
        out[i] = gvo_m32_decode(pad, &pos);
    }
    if (rc == GVO_OK && pos > nBytes) rc = GVO_ERR_BOUNDS;
    free(pad);
    return rc;
}

/* LsDecoder12.decode :94-160 with LsHeader(byte[],int) :131-185: legacy and current headers,
 * container types 0 (legacy Huffman of M32), 1 (Deflate of M32), 2 (canonical Huffman of ints). */
int gvo_lsop12_decode(int nRows, int nCols, const uint8_t *packing, size_t len, int32_t *values)
{
    if (nRows < 6 || nCols < 6 || len < 3) return GVO_ERR_BOUNDS;
    size_t o = 1;
    int type = 0, nCoef, hasChecksum = 0;
    uint32_t nInitCodes = 0, nIntCodes = 0;
    int32_t seed;
    float u[12];
    const int revised = packing[1] & 0x40;
    if (revised) { type = packing[o] & 0x0f; hasChecksum = (packing[o] & 0x80) != 0; o++; }
    nCoef = (int8_t)packing[o++];
    if (nCoef != 12) return GVO_ERR_FORMAT;               /* u[11] would index out of bounds otherwise */
    if (len < o + 4 + 48 + ((!revised || type != 2) ? 8u : 0u) + (revised ? 0u : 1u)) return GVO_ERR_BOUNDS;
    seed = (int32_t)get_i32(packing + o); o += 4;
    for (int i = 0; i < 12; i++) { uint32_t b = get_i32(packing + o); memcpy(&u[i], &b, 4); o += 4; }
    if (!revised) {
        nInitCodes = get_i32(packing + o); o += 4;
        nIntCodes = get_i32(packing + o); o += 4;
        type = packing[o] & 0x0f; hasChecksum = (packing[o] & 0x80) != 0; o++;
    } else if (type != 2) {
        nInitCodes = get_i32(packing + o); o += 4;
        nIntCodes = get_i32(packing + o); o += 4;
    }
    if (hasChecksum) o += 4;
    if (o > len) return GVO_ERR_BOUNDS;
    const size_t nInit = (size_t)nRows * 4 + (size_t)nCols * 2 - 9, nInt = (size_t)(nRows - 2) * (size_t)(nCols - 4);
    int32_t *initInt = calloc(nInit, sizeof(int32_t)), *interior = calloc(nInt, sizeof(int32_t));
    int rc = GVO_OK;
    if (type == 2) {
        size_t bitPos = o * 8, nd = 0;
        rc = gvo_canon_decode(packing, len * 8, &bitPos, initInt, nInit, &nd);
        if (rc == GVO_OK) rc = gvo_canon_decode(packing, len * 8, &bitPos, interior, nInt, &nd);
    } else {
        if (nInitCodes > 6 * nInit + 64 || nIntCodes > 6 * nInt + 64) rc = GVO_ERR_FORMAT;
        uint8_t *mI = NULL, *mX = NULL;
        if (rc == GVO_OK) { mI = calloc(nInitCodes + 8, 1); mX = calloc(nIntCodes + 8, 1); }
        if (rc == GVO_OK && type == 0) {
            size_t bitPos = o * 8;
            rc = gvo_huffman_decode(packing, len * 8, &bitPos, mI, nInitCodes);
            if (rc == GVO_OK) rc = gvo_huffman_decode(packing, len * 8, &bitPos, mX, nIntCodes);
        } else if (rc == GVO_OK) {
            z_stream s;
            memset(&s, 0, sizeof s);
            inflateInit(&s);
            s.next_in = (Bytef *)(packing + o); s.avail_in = (uInt)(len - o);
            s.next_out = mI; s.avail_out = (uInt)nInitCodes;
            int zr = inflate(&s, Z_PARTIAL_FLUSH);
            size_t used = s.total_in, got = s.total_out;
            inflateEnd(&s);
            if ((zr != Z_OK && zr != Z_STREAM_END && zr != Z_BUF_ERROR) || got < nInitCodes) rc = GVO_ERR_FORMAT;
            if (rc == GVO_OK) {
                memset(&s, 0, sizeof s);
                inflateInit(&s);
                s.next_in = (Bytef *)(packing + o + used); s.avail_in = (uInt)(len - o - used);
                s.next_out = mX; s.avail_out = (uInt)nIntCodes;
                zr = inflate(&s, Z_PARTIAL_FLUSH);
                got = s.total_out;
                inflateEnd(&s);
                if ((zr != Z_OK && zr != Z_STREAM_END && zr != Z_BUF_ERROR) || got < nIntCodes) rc = GVO_ERR_FORMAT;
            }
        }
        if (rc == GVO_OK) rc = m32_unpack(mI, nInitCodes, initInt, nInit);
        if (rc == GVO_OK) rc = m32_unpack(mX, nIntCodes, interior, nInt);
        free(mI); free(mX);
    }
    if (rc == GVO_OK) lsop_unpack(initInt, interior, seed, u, nRows, nCols, values);
    free(initInt); free(interior);
    return rc;
}

/* test helper: the legacy (pre-revision) container LsEncoder12 wrote when Sample14_LSOP.gvrs was made:
 * header codec, 12, seed, 12 floats, nInit, nInterior, type 0, then legacy Huffman of the two M32 streams
 * in one bit store (decoder side: LsHeader.java:139-160, LsDecoder12.java:116-121). */
int gvo_lsop12_encode_legacy_huffman(int codecIndex, int nRows, int nCols, const int32_t *values, uint8_t *out,
                                     size_t outCap, size_t *outLen)
{
    if (nRows < 6 || nCols < 6) return GVO_DECLINED;
    const size_t nInit = (size_t)nRows * 4 + (size_t)nCols * 2 - 9, nInt = (size_t)(nRows - 2) * (size_t)(nCols - 4);
    int32_t *initInt = malloc(sizeof(int32_t) * nInit), *interior = malloc(sizeof(int32_t) * nInt);
    float u[12];
    int32_t seed;
    int rc = gvo_lsop12_residuals(nRows, nCols, values, &seed, u, initInt, interior);
    if (rc == GVO_OK) {
        uint8_t *mI = malloc(6 * nInit + 8), *mX = malloc(6 * nInt + 8);
        size_t nMI = m32_pack(initInt, nInit, mI), nMX = m32_pack(interior, nInt, mX);
        memset(out, 0, outCap);
        if (outCap < 64) rc = GVO_ERR_CAPACITY;
        else {
            out[0] = (uint8_t)codecIndex;
            out[1] = 12;
            put_i32(out + 2, (uint32_t)seed);
            size_t o = 6;
            for (int i = 0; i < 12; i++) { uint32_t b; memcpy(&b, &u[i], 4); put_i32(out + o, b); o += 4; }
            put_i32(out + o, (uint32_t)nMI); o += 4;
            put_i32(out + o, (uint32_t)nMX); o += 4;
            out[o++] = 0;
            size_t bitPos = o * 8;
            rc = gvo_huffman_encode(out, outCap * 8, &bitPos, mI, nMI, NULL, NULL);
            if (rc == GVO_OK) rc = gvo_huffman_encode(out, outCap * 8, &bitPos, mX, nMX, NULL, NULL);
            *outLen = (bitPos + 7) / 8;
        }
        free(mI); free(mX);
    }
    free(initInt); free(interior);
    return rc;
}

int gvo_batch_lsop12_encode(int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values, int deflateEnabled,
                            uint8_t *out, size_t stride, uint32_t *lengths, uint8_t *types)
{
    const size_t nCells = (size_t)nRows * (size_t)nCols;
    for (size_t t = 0; t < nTiles; t++) {
        size_t len = 0;
        int type = 0;
        int rc = gvo_lsop12_encode(codecIndex, nRows, nCols, values + t * nCells, deflateEnabled, out + t * stride, stride,
                                   &len, &type);
        if (rc < 0) return rc;
        lengths[t] = rc == GVO_OK ? (uint32_t)len : 0;
        if (types) types[t] = (uint8_t)(rc == GVO_OK ? type : 0);
    }
    return GVO_OK;
}

int gvo_batch_lsop12_decode(int nRows, int nCols, size_t nTiles, const uint8_t *packings, size_t stride,
                            const uint32_t *lengths, int32_t *values)
{
    const size_t nCells = (size_t)nRows * (size_t)nCols;
    for (size_t t = 0; t < nTiles; t++) {
        int rc = gvo_lsop12_decode(nRows, nCols, packings + t * stride, lengths[t], values + t * nCells);
        if (rc != GVO_OK) return rc;
    }
    return GVO_OK;
}
