/*
 * gvrs_oracle.c -- CPU restatement ("oracle") of the Gridfour GVRS tile codec.
 *
 * TEST INFRASTRUCTURE ONLY (parity checker + bench.py cpu_baseline leg); see
 * gvrs_oracle.h.  Plain C11, single threaded, integer arithmetic only on the
 * int path.  Java semantics honoured: int overflow wraps (done in uint32_t),
 * (byte) casts truncate, >> on int is arithmetic.
 *
 * Reference paths are relative to core/src/main/java/org/gridfour/.
 */
#define _POSIX_C_SOURCE 200809L   /* pthread barriers, clock_gettime (the all-cores CPU baseline) */
#include "gvrs_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* wrap-around int32 helpers (Java int arithmetic) */
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

/* ------------------------------------------------------------------ */
/* CodecM32                                                            */
/* ------------------------------------------------------------------ */

/* compress/CodecM32.java:257-311 */
int gvo_m32_encode(int32_t value, uint8_t *out)
{
    uint32_t absValue;
    int n = 0;
    if (value < 0) {
        if (value == INT32_MIN) {             /* :267-269 */
            out[0] = 0x80;
            return 1;
        } else if (value > -127) {            /* :270-272 */
            out[0] = (uint8_t)value;
            return 1;
        }
        out[n++] = (uint8_t)(-127);           /* 0x81 introducer, :274 */
        absValue = (uint32_t)(-value);
    } else {
        if (value < 127) {                    /* :277-279 */
            out[0] = (uint8_t)value;
            return 1;
        }
        out[n++] = 127;                       /* 0x7f introducer, :281 */
        absValue = (uint32_t)value;
    }
    if (absValue <= 254) {
        out[n++] = (uint8_t)(absValue - 127);
    } else if (absValue <= 16638) {
        uint32_t d = absValue - 255;
        out[n++] = (uint8_t)(((d >> 7) & 0x7f) | 0x80);
        out[n++] = (uint8_t)(d & 0x7f);
    } else if (absValue <= 2113790) {
        uint32_t d = absValue - 16639;
        out[n++] = (uint8_t)(((d >> 14) & 0x7f) | 0x80);
        out[n++] = (uint8_t)(((d >> 7) & 0x7f) | 0x80);
        out[n++] = (uint8_t)(d & 0x7f);
    } else if (absValue <= 270549246) {
        uint32_t d = absValue - 2113791;
        out[n++] = (uint8_t)(((d >> 21) & 0x7f) | 0x80);
        out[n++] = (uint8_t)(((d >> 14) & 0x7f) | 0x80);
        out[n++] = (uint8_t)(((d >> 7) & 0x7f) | 0x80);
        out[n++] = (uint8_t)(d & 0x7f);
    } else {
        uint32_t d = absValue - 270549247;
        out[n++] = (uint8_t)(((d >> 28) & 0x7f) | 0x80);
        out[n++] = (uint8_t)(((d >> 21) & 0x7f) | 0x80);
        out[n++] = (uint8_t)(((d >> 14) & 0x7f) | 0x80);
        out[n++] = (uint8_t)(((d >> 7) & 0x7f) | 0x80);
        out[n++] = (uint8_t)(d & 0x7f);
    }
    return n;
}

/* compress/CodecM32.java:313-315 */
static const int32_t m32_segment_base[5] = {127, 255, 16639, 2113791, 270549247};

/* compress/CodecM32.java:327-356 */
int32_t gvo_m32_decode(const uint8_t *buf, size_t *pos)
{
    int32_t symbol = (int8_t)buf[(*pos)++];
    if (symbol == -128) {
        return INT32_MIN;
    } else if (-127 < symbol && symbol < 127) {
        return symbol;
    }
    int32_t delta = 0;
    for (int i = 0; i < 5; i++) {
        int32_t sample = (int8_t)buf[(*pos)++];
        delta = (int32_t)(((uint32_t)delta << 7) | ((uint32_t)sample & 0x7f));
        if ((sample & 0x80) == 0) {
            if (symbol == -127) {
                delta = wsub(wsub(0, delta), m32_segment_base[i]);
            } else {
                delta = wadd(delta, m32_segment_base[i]);
            }
            break;
        }
    }
    return delta;
}

/* ------------------------------------------------------------------ */
/* Predictors                                                          */
/* ------------------------------------------------------------------ */

/* compress/PredictorModelDifferencing.java:112-142 */
static int pm_differencing_encode(int nRows, int nCols, const int32_t *v, uint8_t *out, int32_t *seed)
{
    size_t n = 0;
    *seed = v[0];
    int32_t prior = v[0];
    for (int i = 1; i < nCols; i++) {
        int32_t test = v[i];
        n += gvo_m32_encode(wsub(test, prior), out + n);
        prior = test;
    }
    for (int iRow = 1; iRow < nRows; iRow++) {
        size_t index = (size_t)iRow * nCols;
        prior = v[index - nCols];
        for (int i = 0; i < nCols; i++) {
            int32_t test = v[index++];
            n += gvo_m32_encode(wsub(test, prior), out + n);
            prior = test;
        }
    }
    return (int)n;
}

/* compress/PredictorModelDifferencing.java:145-167 */
static void pm_differencing_decode(int32_t seed, int nRows, int nCols, const uint8_t *m, int32_t *o)
{
    size_t p = 0;
    o[0] = seed;
    int32_t prior = seed;
    for (int i = 1; i < nCols; i++) {
        prior = wadd(prior, gvo_m32_decode(m, &p));
        o[i] = prior;
    }
    for (int iRow = 1; iRow < nRows; iRow++) {
        size_t index = (size_t)iRow * nCols;
        prior = o[index - nCols];
        for (int iCol = 0; iCol < nCols; iCol++) {
            prior = wadd(prior, gvo_m32_decode(m, &p));
            o[index++] = prior;
        }
    }
}

/* compress/PredictorModelLinear.java:104-143.  The reference computes in
 * long and truncates to int; every result is the low 32 bits of an exact
 * integer expression, i.e. int32 wrap-around arithmetic.                  */
static int pm_linear_encode(int nRows, int nCols, const int32_t *v, uint8_t *out, int32_t *seed)
{
    size_t n = 0;
    *seed = v[0];
    int32_t prior = v[0];
    n += gvo_m32_encode(wsub(v[1], prior), out + n);                 /* :113-114 */
    for (int iRow = 1; iRow < nRows; iRow++) {                       /* :115-126 */
        size_t index = (size_t)iRow * nCols;
        int32_t test = v[index];
        n += gvo_m32_encode(wsub(test, prior), out + n);
        prior = test;
        test = v[index + 1];
        n += gvo_m32_encode(wsub(test, prior), out + n);
    }
    for (int iRow = 0; iRow < nRows; iRow++) {                       /* :128-141 */
        size_t index = (size_t)iRow * nCols;
        int32_t a = v[index];
        int32_t b = v[index + 1];
        for (int iCol = 2; iCol < nCols; iCol++) {
            int32_t c = v[index + iCol];
            int32_t prediction = wsub(wadd(b, b), a);               /* (int)(2L*b - a) */
            n += gvo_m32_encode(wsub(c, prediction), out + n);
            a = b;
            b = c;
        }
    }
    return (int)n;
}

/* compress/PredictorModelLinear.java:66-101 */
static void pm_linear_decode(int32_t seed, int nRows, int nCols, const uint8_t *m, int32_t *o)
{
    size_t p = 0;
    int32_t prior = seed;
    o[0] = seed;
    o[1] = wadd(gvo_m32_decode(m, &p), prior);
    for (int iRow = 1; iRow < nRows; iRow++) {
        size_t index = (size_t)iRow * nCols;
        int32_t test = wadd(gvo_m32_decode(m, &p), prior);
        o[index] = test;
        prior = test;
        o[index + 1] = wadd(gvo_m32_decode(m, &p), test);
    }
    for (int iRow = 0; iRow < nRows; iRow++) {
        size_t index = (size_t)iRow * nCols;
        int32_t a = o[index];
        int32_t b = o[index + 1];
        for (int iCol = 2; iCol < nCols; iCol++) {
            int32_t residual = gvo_m32_decode(m, &p);
            int32_t prediction = wsub(wadd(b, b), a);
            int32_t c = wadd(prediction, residual);
            a = b;
            b = c;
            o[index + iCol] = c;
        }
    }
}

/* compress/PredictorModelTriangle.java:101-145 */
static int pm_triangle_encode(int nRows, int nCols, const int32_t *v, uint8_t *out, int32_t *seed)
{
    if (nRows < 2 || nCols < 2) {
        return -1;                                                    /* :107-109 */
    }
    size_t n = 0;
    *seed = v[0];
    int32_t prior = v[0];
    for (int i = 1; i < nCols; i++) {
        int32_t test = v[i];
        n += gvo_m32_encode(wsub(test, prior), out + n);
        prior = test;
    }
    prior = v[0];
    for (int i = 1; i < nRows; i++) {
        int32_t test = v[(size_t)i * nCols];
        n += gvo_m32_encode(wsub(test, prior), out + n);
        prior = test;
    }
    for (int iRow = 1; iRow < nRows; iRow++) {
        size_t k1 = (size_t)iRow * nCols;
        size_t k0 = k1 - nCols;
        for (int i = 1; i < nCols; i++) {
            int32_t za = v[k0++];
            int32_t zb = v[k1++];
            int32_t zc = v[k0];
            int32_t prediction = wsub(wadd(zc, zb), za);
            n += gvo_m32_encode(wsub(v[k1], prediction), out + n);
        }
    }
    return (int)n;
}

/* compress/PredictorModelTriangle.java:62-98 */
static void pm_triangle_decode(int32_t seed, int nRows, int nCols, const uint8_t *m, int32_t *o)
{
    size_t p = 0;
    o[0] = seed;
    int32_t prior = seed;
    for (int i = 1; i < nCols; i++) {
        prior = wadd(prior, gvo_m32_decode(m, &p));
        o[i] = prior;
    }
    prior = seed;
    for (int i = 1; i < nRows; i++) {
        prior = wadd(prior, gvo_m32_decode(m, &p));
        o[(size_t)i * nCols] = prior;
    }
    for (int iRow = 1; iRow < nRows; iRow++) {
        size_t k1 = (size_t)iRow * nCols;
        size_t k0 = k1 - nCols;
        for (int i = 1; i < nCols; i++) {
            int32_t za = o[k0++];
            int32_t zb = o[k1++];
            int32_t zc = o[k0];
            int32_t prediction = wsub(wadd(zb, zc), za);
            o[k1] = wadd(prediction, gvo_m32_decode(m, &p));
        }
    }
}

/* Java Math.floor(x + 0.5) for doubles, then (int) cast (saturating).
 * Implemented without libm: the argument is within +-2^31 here.          */
static int32_t java_floor_to_int(double x)
{
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    int64_t t = (int64_t)x;             /* truncation toward zero */
    if ((double)t > x) t -= 1;          /* floor for negatives    */
    return (int32_t)t;
}

/* compress/PredictorModelDifferencingWithNulls.java:66-134 */
static int pm_diffnulls_encode(int nRows, int nCols, const int32_t *v, uint8_t *out, int32_t *seed)
{
    int64_t sumStart = 0;
    int32_t nStart = 0;
    int nullFlag = 1;
    for (int iRow = 0; iRow < nRows; iRow++) {                        /* :82-96 */
        size_t rowOffset = (size_t)iRow * nCols;
        for (int iCol = 0; iCol < nCols; iCol++) {
            int32_t test = v[rowOffset + iCol];
            if (test == GVO_INT4_NULL) {
                nullFlag = 1;
            } else {
                if (nullFlag) {
                    sumStart += test;
                    nStart++;
                }
                nullFlag = 0;
            }
        }
        nullFlag = v[rowOffset] == GVO_INT4_NULL;
    }
    if (nStart == 0) {
        return 0;                                                     /* :101-103 */
    }
    double avgStart = (double)sumStart / nStart;                      /* :104 */
    int32_t encodedSeed = java_floor_to_int(avgStart + 0.5);          /* :105 */
    *seed = encodedSeed;

    size_t n = 0;
    int32_t prior = encodedSeed;
    nullFlag = 0;
    for (int iRow = 0; iRow < nRows; iRow++) {                        /* :109-131 */
        size_t index = (size_t)iRow * nCols;
        for (int iCol = 0; iCol < nCols; iCol++) {
            int32_t test = v[index++];
            if (test == GVO_INT4_NULL) {
                nullFlag = 1;
                n += gvo_m32_encode(GVO_INT4_NULL, out + n);
            } else {
                if (nullFlag) {
                    prior = encodedSeed;
                    nullFlag = 0;
                }
                n += gvo_m32_encode(wsub(test, prior), out + n);
                prior = test;
            }
        }
        prior = v[(size_t)iRow * nCols];
        nullFlag = prior == GVO_INT4_NULL;
    }
    return (int)n;
}

/* compress/PredictorModelDifferencingWithNulls.java:137-166 */
static void pm_diffnulls_decode(int32_t seed, int nRows, int nCols, const uint8_t *m, int32_t *o)
{
    size_t p = 0;
    int32_t prior = seed;
    int nullFlag = 1;
    for (int iRow = 0; iRow < nRows; iRow++) {
        size_t index = (size_t)iRow * nCols;
        for (int iCol = 0; iCol < nCols; iCol++) {
            int32_t test = gvo_m32_decode(m, &p);
            if (test == GVO_INT4_NULL) {
                nullFlag = 1;
                o[index++] = GVO_INT4_NULL;
            } else {
                if (nullFlag) {
                    nullFlag = 0;
                    prior = seed;
                }
                prior = wadd(prior, test);
                o[index++] = prior;
            }
        }
        prior = o[(size_t)iRow * nCols];
        nullFlag = prior == GVO_INT4_NULL;
    }
}

int gvo_predictor_encode(int model, int nRows, int nCols, const int32_t *values,
                         uint8_t *out, int32_t *seed)
{
    switch (model) {
    case GVO_PM_DIFFERENCING: return pm_differencing_encode(nRows, nCols, values, out, seed);
    case GVO_PM_LINEAR: return pm_linear_encode(nRows, nCols, values, out, seed);
    case GVO_PM_TRIANGLE: return pm_triangle_encode(nRows, nCols, values, out, seed);
    case GVO_PM_DIFFERENCING_NULLS: return pm_diffnulls_encode(nRows, nCols, values, out, seed);
    default: return GVO_ERR_ARG;
    }
}

int gvo_predictor_decode(int model, int32_t seed, int nRows, int nCols,
                         const uint8_t *m32, size_t nM32, int32_t *values)
{
    /* The reference's predictors read codeM32s through CodecM32.decode without a check of their own (:327-356), but the
     * array is exactly nM32 bytes long (new byte[nM32], CodecHuffman.java:143 / CodecDeflate.java:139): a stream that
     * holds fewer values than the predictor reads ends in the JVM's ArrayIndexOutOfBoundsException.  The C buffers here
     * are padded, so that case is found by walking the values the predictor will read. */
    if (model >= 1 && model <= 4 && nRows > 0 && nCols > 0) {
        const size_t need = (size_t)nRows * (size_t)nCols - (model == GVO_PM_DIFFERENCING_NULLS ? 0 : 1);
        size_t p = 0;
        for (size_t k = 0; k < need; k++) {
            (void)gvo_m32_decode(m32, &p);
            if (p > nM32) return GVO_ERR_BOUNDS;
        }
    }
    switch (model) {
    case GVO_PM_DIFFERENCING: pm_differencing_decode(seed, nRows, nCols, m32, values); return GVO_OK;
    case GVO_PM_LINEAR: pm_linear_decode(seed, nRows, nCols, m32, values); return GVO_OK;
    case GVO_PM_TRIANGLE: pm_triangle_decode(seed, nRows, nCols, m32, values); return GVO_OK;
    case GVO_PM_DIFFERENCING_NULLS: pm_diffnulls_decode(seed, nRows, nCols, m32, values); return GVO_OK;
    default: return GVO_ERR_FORMAT;   /* CodecHuffman.java:155-169 */
    }
}

/* ------------------------------------------------------------------ */
/* Bit store: bit i of the stream = bit (i&7) of byte (i>>3)           */
/* io/BitOutputStore.java:46-59, 205-288; io/BitInputStore.java:112-210 */
/* ------------------------------------------------------------------ */

#include "oracle_bits.h"

/* ------------------------------------------------------------------ */
/* HuffmanEncoder                                                      */
/* ------------------------------------------------------------------ */

typedef struct hnode {
    int symbol;              /* -1 for branches */
    int isLeaf;
    uint32_t count;
    int bit;
    struct hnode *next, *left, *right;
    /* code = root->leaf path, path order */
    int nBitsInCode;
    uint8_t code[32];        /* 256 bits: depth <= 255 */
} hnode_t;

/* compress/HuffmanEncoder.java:86-92 (compareTo: count asc, symbol asc) */
static int hnode_cmp(const void *pa, const void *pb)
{
    const hnode_t *a = *(hnode_t *const *)pa, *b = *(hnode_t *const *)pb;
    if (a->count != b->count) return a->count < b->count ? -1 : 1;
    return (a->symbol > b->symbol) - (a->symbol < b->symbol);
}

int gvo_huffman_encode(uint8_t *bits, size_t capBits, size_t *bitPos,
                       const uint8_t *symbols, size_t nSymbols,
                       uint8_t *codeLen256, size_t *treeBits)
{
    bitw_t w = {bits, capBits, *bitPos, 0};
    size_t pos0 = w.pos;
    hnode_t *nodes = (hnode_t *)calloc(512, sizeof(hnode_t));
    hnode_t *sortNodes[256];
    if (!nodes) return GVO_ERR_ARG;
    int nAlloc = 256;
    for (int i = 0; i < 256; i++) {                                   /* :131-134 */
        nodes[i].symbol = i;
        nodes[i].isLeaf = 1;
        sortNodes[i] = &nodes[i];
    }
    for (size_t i = 0; i < nSymbols; i++) nodes[symbols[i]].count++;  /* :135-137 */
    qsort(sortNodes, 256, sizeof(sortNodes[0]), hnode_cmp);           /* :138 (total order) */
    if (codeLen256) memset(codeLen256, 0, 256);

    int firstIndex = -1;
    for (int i = 0; i < 256; i++) {
        if (sortNodes[i]->count > 0) { firstIndex = i; break; }
    }
    if (firstIndex < 0) {          /* nSymbols == 0: the reference would NPE; callers never do this */
        free(nodes);
        return GVO_ERR_ARG;
    }
    if (firstIndex == 255) {                                          /* :147-157 */
        bw_bits(&w, 8, 0);
        bw_bit(&w, 1);
        bw_bits(&w, 8, (uint32_t)sortNodes[255]->symbol);
        if (treeBits) *treeBits = 9;
        *bitPos = w.pos;
        free(nodes);
        return w.overflow ? GVO_ERR_CAPACITY : GVO_OK;
    }
    hnode_t *firstNode = sortNodes[firstIndex];
    for (int i = firstIndex; i < 255; i++) sortNodes[i]->next = sortNodes[i + 1];
    int nLeafNodes = 256 - firstIndex;
    hnode_t *root = NULL;
    for (;;) {                                                        /* :165-194 */
        hnode_t *left = firstNode;
        hnode_t *right = firstNode->next;
        firstNode = right->next;
        left->next = NULL;
        right->next = NULL;
        hnode_t *branch = &nodes[nAlloc++];
        branch->isLeaf = 0;
        branch->symbol = -1;
        branch->left = left;
        branch->right = right;
        branch->count = right->count + left->count;
        left->bit = 0;
        right->bit = 1;
        if (firstNode == NULL) {
            root = branch;
            break;
        } else if (firstNode->count >= branch->count) {
            branch->next = firstNode;
            firstNode = branch;
        } else {
            hnode_t *node = firstNode->next;
            hnode_t *prior = firstNode;
            while (node != NULL && node->count < branch->count) {
                prior = node;
                node = node->next;
            }
            prior->next = branch;
            branch->next = node;
        }
    }

    /* encodeTree :221-294 -- pre-order, explicit stack */
    bw_bits(&w, 8, (uint32_t)(nLeafNodes - 1));
    {
        hnode_t *path[512];
        int pathBranch[512];
        path[0] = root;
        pathBranch[0] = 0;
        int depth = 1;
        while (depth > 0) {
            int index = depth - 1;
            hnode_t *pNode = path[index];
            switch (pathBranch[index]) {
            case 0:
                if (pNode->isLeaf) {
                    bw_bit(&w, 1);
                    bw_bits(&w, 8, (uint32_t)pNode->symbol);
                    /* encodePath :298-305: bits of path[1..depth-1] */
                    pNode->nBitsInCode = depth - 1;
                    memset(pNode->code, 0, sizeof(pNode->code));
                    for (int i = 1; i < depth; i++) {
                        if (path[i]->bit) pNode->code[(i - 1) >> 3] |= (uint8_t)(1u << ((i - 1) & 7));
                    }
                    depth--;
                } else {
                    bw_bit(&w, 0);
                    pathBranch[index] = 1;
                    pathBranch[depth] = 0;
                    path[depth] = pNode->left;
                    depth++;
                }
                break;
            case 1:
                pathBranch[index] = 2;
                pathBranch[depth] = 0;
                path[depth] = pNode->right;
                depth++;
                break;
            default:
                pathBranch[index] = 0;
                depth--;
                break;
            }
        }
    }
    if (treeBits) *treeBits = w.pos - pos0;
    if (codeLen256) {
        for (int i = 0; i < 256; i++) codeLen256[i] = (uint8_t)nodes[i].nBitsInCode;
    }
    for (size_t i = 0; i < nSymbols; i++) {                           /* :198-213 */
        const hnode_t *node = &nodes[symbols[i]];
        int nFull = node->nBitsInCode / 8;
        for (int j = 0; j < nFull; j++) bw_bits(&w, 8, node->code[j]);
        int rem = node->nBitsInCode - nFull * 8;
        if (rem > 0) bw_bits(&w, rem, node->code[nFull]);
    }
    *bitPos = w.pos;
    free(nodes);
    return w.overflow ? GVO_ERR_CAPACITY : GVO_OK;
}

/* ------------------------------------------------------------------ */
/* HuffmanDecoder                                                      */
/* ------------------------------------------------------------------ */

/* compress/HuffmanDecoder.java:65-187 */
int gvo_huffman_decode(const uint8_t *bits, size_t nBitsTotal, size_t *bitPos,
                       uint8_t *symbols, size_t nSymbols)
{
    bitr_t r = {bits, nBitsTotal, *bitPos, 0};
    int nLeafsToDecode = (int)br_bits(&r, 8) + 1;
    int rootBit = br_bit(&r);
    if (r.overrun) return GVO_ERR_BOUNDS;
    if (rootBit == 1) {                                               /* :70-78, 170-177 */
        uint8_t symbol = (uint8_t)br_bits(&r, 8);
        if (r.overrun) return GVO_ERR_BOUNDS;
        memset(symbols, symbol, nSymbols);
        *bitPos = r.pos;
        return GVO_OK;
    }
    int nodeIndex[256 * 6 + 6];
    int stack[258];
    int iStack = 0;
    int nodeIndexCount = 3;
    memset(nodeIndex, 0, sizeof(nodeIndex));
    nodeIndex[0] = -1;
    stack[0] = 0;
    int nLeafsDecoded = 0;
    while (nLeafsDecoded < nLeafsToDecode) {                          /* :117-157 */
        int offset = stack[iStack];
        if (nodeIndex[offset + 1] == 0) {
            nodeIndex[offset + 1] = nodeIndexCount;
        } else {
            nodeIndex[offset + 2] = nodeIndexCount;
        }
        int bit = br_bit(&r);
        if (r.overrun) return GVO_ERR_BOUNDS;
        if (bit == 1) {
            nLeafsDecoded++;
            if (nodeIndexCount + 3 > nLeafsToDecode * 6) return GVO_ERR_BOUNDS; /* nodeIndex AIOOBE (a leaf's three slots) */
            nodeIndex[nodeIndexCount++] = (int)br_bits(&r, 8);
            nodeIndex[nodeIndexCount++] = 0;
            nodeIndex[nodeIndexCount++] = 0;
            if (r.overrun) return GVO_ERR_BOUNDS;
            if (nLeafsDecoded == nLeafsToDecode) break;
            while (nodeIndex[offset + 2] != 0) {
                iStack--;
                if (iStack < 0) return GVO_ERR_BOUNDS;   /* Java: AIOOBE on a malformed tree */
                offset = stack[iStack];
            }
        } else {
            iStack++;
            if (iStack > nLeafsToDecode) return GVO_ERR_BOUNDS; /* Java: stack AIOOBE */
            stack[iStack] = nodeIndexCount;
            if (nodeIndexCount + 3 > nLeafsToDecode * 6) return GVO_ERR_BOUNDS; /* nodeIndex AIOOBE */
            nodeIndex[nodeIndexCount++] = -1;
            nodeIndex[nodeIndexCount++] = 0;
            nodeIndex[nodeIndexCount++] = 0;
        }
    }
    for (size_t i = 0; i < nSymbols; i++) {                           /* :179-185 */
        int offset = nodeIndex[1 + br_bit(&r)];
        while (nodeIndex[offset] == -1) {
            offset = nodeIndex[offset + 1 + br_bit(&r)];
            if (r.overrun) return GVO_ERR_BOUNDS;
        }
        if (r.overrun) return GVO_ERR_BOUNDS;
        symbols[i] = (uint8_t)nodeIndex[offset];
    }
    *bitPos = r.pos;
    return GVO_OK;
}

/* ------------------------------------------------------------------ */
/* CodecHuffman                                                        */
/* ------------------------------------------------------------------ */

size_t gvo_codec_huffman_bound(size_t nCells)
{
    /* header 80 bits + tree (8 + 10*256 - 1) + 6 M32 bytes/cell at <= 255 bits/code */
    size_t bits = 80 + 8 + 2559 + nCells * 6 * 255;
    return (bits + 7) / 8 + 16;
}

/* compress/CodecHuffman.java:70-130 */
int gvo_codec_huffman_encode(int codecIndex, int nRows, int nCols,
                             const int32_t *values, uint8_t *out, size_t outCap,
                             size_t *outLen, int predictorMask, int *predictorUsed)
{
    size_t nCells = (size_t)nRows * (size_t)nCols;
    int containsNull = 0, containsValid = 0;
    if (nRows < 1 || nCols < 1) return GVO_ERR_ARG;
    for (size_t i = 0; i < nCells; i++) {                             /* :71-79 */
        if (values[i] == GVO_INT4_NULL) containsNull = 1; else containsValid = 1;
    }
    if (!containsValid) return GVO_DECLINED;                          /* :80-82 */

    uint8_t *mCode = (uint8_t *)malloc(6 * nCells + 8);
    /* text is at most ~1.1 bytes per M32 byte in practice; size for code lengths <= 32,
     * retry bigger on overflow */
    size_t capBytes = 10 + 330 + 4 * 6 * nCells / 4 + 64;
    uint8_t *test = NULL, *best = NULL;
    size_t bestLen = SIZE_MAX;
    int bestModel = 0;
    int rc = GVO_OK;
    static const int order[4] = {GVO_PM_DIFFERENCING, GVO_PM_LINEAR, GVO_PM_TRIANGLE,
                                 GVO_PM_DIFFERENCING_NULLS};      /* :60-65 */
    if (!mCode) return GVO_ERR_ARG;
    for (int k = 0; k < 4; k++) {
        int model = order[k];
        int nullModel = model == GVO_PM_DIFFERENCING_NULLS;
        if (containsNull != nullModel) continue;                      /* :89-98 */
        if (!(predictorMask & (1 << (model - 1)))) continue;
        if (model == GVO_PM_LINEAR && nCols < 2) { rc = GVO_ERR_BOUNDS; break; } /* Java AIOOBE */
        int32_t seed = 0;
        int mLen = gvo_predictor_encode(model, nRows, nCols, values, mCode, &seed);
        if (mLen <= 0) continue;                                      /* :100 */
        for (;;) {                                                    /* compress() :121-130 */
            free(test);
            test = (uint8_t *)calloc(capBytes, 1);
            if (!test) { rc = GVO_ERR_ARG; break; }
            bitw_t w = {test, capBytes * 8, 0, 0};
            bw_bits(&w, 8, (uint32_t)codecIndex);
            bw_bits(&w, 8, (uint32_t)model);
            bw_bits(&w, 32, (uint32_t)seed);
            bw_bits(&w, 32, (uint32_t)mLen);
            size_t pos = w.pos;
            int hrc = gvo_huffman_encode(test, capBytes * 8, &pos, mCode, (size_t)mLen, NULL, NULL);
            if (hrc == GVO_ERR_CAPACITY) { capBytes *= 4; continue; }
            if (hrc != GVO_OK) { rc = hrc; break; }
            size_t testLen = (pos + 7) / 8;                           /* getEncodedTextLengthInBytes */
            if (testLen < bestLen) {                                  /* :107 strict */
                free(best);
                best = test;
                test = NULL;
                bestLen = testLen;
                bestModel = model;
            }
            break;
        }
        if (rc != GVO_OK) break;
    }
    free(test);
    free(mCode);
    if (rc != GVO_OK) { free(best); return rc; }
    if (!best) return GVO_DECLINED;                                   /* :114-116 */
    if (outLen) *outLen = bestLen;
    if (predictorUsed) *predictorUsed = bestModel;
    if (bestLen > outCap) { free(best); return GVO_ERR_CAPACITY; }
    memcpy(out, best, bestLen);
    free(best);
    return GVO_OK;
}

/* compress/CodecHuffman.java:133-169 */
int gvo_codec_huffman_decode(int nRows, int nCols, const uint8_t *packing,
                             size_t len, int32_t *values)
{
    if (len < 10) return GVO_ERR_BOUNDS;
    int model = packing[1];
    if (model < 1 || model > 4) return GVO_ERR_FORMAT;                /* :155-169 */
    int32_t seed = (int32_t)((uint32_t)packing[2] | ((uint32_t)packing[3] << 8) |
                             ((uint32_t)packing[4] << 16) | ((uint32_t)packing[5] << 24));
    uint32_t nM32 = (uint32_t)packing[6] | ((uint32_t)packing[7] << 8) |
                    ((uint32_t)packing[8] << 16) | ((uint32_t)packing[9] << 24);
    if ((int32_t)nM32 < 0) return GVO_ERR_BOUNDS;                     /* NegativeArraySize */
    /* the predictors read M32 without bounds checks; pad so a malformed
     * stream cannot run off the buffer within one tile's worth of reads */
    size_t nCells = (size_t)nRows * (size_t)nCols;
    uint8_t *m = (uint8_t *)calloc((size_t)nM32 + 6 * nCells + 8, 1);
    if (!m) return GVO_ERR_ARG;
    size_t pos = 80;
    int rc = gvo_huffman_decode(packing, len * 8, &pos, m, nM32);
    if (rc == GVO_OK) rc = gvo_predictor_decode(model, seed, nRows, nCols, m, nM32, values);
    free(m);
    return rc;
}

/* ------------------------------------------------------------------ */
/* CodecDeflate                                                        */
/* ------------------------------------------------------------------ */

static int zdeflate(const uint8_t *in, size_t n, int level, uint8_t *out, size_t cap, size_t *outLen)
{
    /* java.util.zip.Deflater(level): setInput, finish, deflate(.., FULL_FLUSH)
     * == a complete zlib stream (CodecDeflate.java:204-210, CodecFloat.java:268-283) */
    z_stream s;
    memset(&s, 0, sizeof(s));
    if (deflateInit(&s, level) != Z_OK) return GVO_ERR_ARG;
    s.next_in = (Bytef *)in;
    s.avail_in = (uInt)n;
    s.next_out = out;
    s.avail_out = (uInt)cap;
    int zr = deflate(&s, Z_FINISH);
    *outLen = s.total_out;
    deflateEnd(&s);
    return zr == Z_STREAM_END ? GVO_OK : GVO_ERR_CAPACITY;
}

static int zinflate(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *outLen)
{
    z_stream s;
    memset(&s, 0, sizeof(s));
    if (inflateInit(&s) != Z_OK) return GVO_ERR_ARG;
    s.next_in = (Bytef *)in;
    s.avail_in = (uInt)n;
    s.next_out = out;
    s.avail_out = (uInt)cap;
    int zr = inflate(&s, Z_FINISH);
    *outLen = s.total_out;
    inflateEnd(&s);
    /* Z_NEED_DICT (a header with the preset-dictionary flag): java.util.zip.Inflater.inflate returns 0 with needsDictionary()
       set -- nothing inflated, no DataFormatException (Inflater.c: case Z_NEED_DICT) */
    if (zr == Z_STREAM_END || zr == Z_BUF_ERROR || zr == Z_OK || zr == Z_NEED_DICT) return GVO_OK;
    return GVO_ERR_FORMAT;
}

/* compress/CodecDeflate.java:157-228 */
int gvo_codec_deflate_encode(int codecIndex, int nRows, int nCols,
                             const int32_t *values, uint8_t *out, size_t outCap,
                             size_t *outLen, int *predictorUsed)
{
    size_t nCells = (size_t)nRows * (size_t)nCols;
    int containsNull = 0, containsValid = 0;
    for (size_t i = 0; i < nCells; i++) {
        if (values[i] == GVO_INT4_NULL) containsNull = 1; else containsValid = 1;
    }
    if (!containsValid) return GVO_DECLINED;
    /* PredictorModelLinear.encode indexes values[index + 1] (:113-126): with one column that runs past the array
     * (ArrayIndexOutOfBoundsException out of CodecDeflate.encode) */
    if (!containsNull && nCols < 2) return GVO_ERR_BOUNDS;
    uint8_t *mCode = (uint8_t *)malloc(6 * nCells + 8);
    uint8_t *test = (uint8_t *)malloc(6 * nCells + 138);
    uint8_t *best = (uint8_t *)malloc(6 * nCells + 138);
    size_t bestLen = SIZE_MAX;
    int bestModel = 0;
    static const int order[4] = {GVO_PM_DIFFERENCING, GVO_PM_LINEAR, GVO_PM_TRIANGLE,
                                 GVO_PM_DIFFERENCING_NULLS};
    for (int k = 0; k < 4; k++) {
        int model = order[k];
        if (containsNull != (model == GVO_PM_DIFFERENCING_NULLS)) continue;
        int32_t seed = 0;
        int mLen = gvo_predictor_encode(model, nRows, nCols, values, mCode, &seed);
        if (mLen <= 0) continue;
        size_t dN = 0;
        /* result buffer is nM32+128 with 10 header bytes (:208-209) */
        if (zdeflate(mCode, (size_t)mLen, 6, test + 10, (size_t)mLen + 118, &dN) != GVO_OK || dN == 0)
            continue;
        test[0] = (uint8_t)codecIndex;
        test[1] = (uint8_t)model;
        for (int b = 0; b < 4; b++) test[2 + b] = (uint8_t)((uint32_t)seed >> (8 * b));
        for (int b = 0; b < 4; b++) test[6 + b] = (uint8_t)((uint32_t)mLen >> (8 * b));
        if (dN + 10 < bestLen) {
            bestLen = dN + 10;
            bestModel = model;
            uint8_t *t = best; best = test; test = t;
        }
    }
    int rc = GVO_OK;
    if (bestLen == SIZE_MAX) rc = GVO_DECLINED;
    else if (bestLen > outCap) rc = GVO_ERR_CAPACITY;
    else memcpy(out, best, bestLen);
    if (rc != GVO_DECLINED && outLen) *outLen = bestLen;
    if (predictorUsed) *predictorUsed = bestModel;
    free(mCode); free(test); free(best);
    return rc;
}

/* compress/CodecDeflate.java:108-155 */
int gvo_codec_deflate_decode(int nRows, int nCols, const uint8_t *packing,
                             size_t len, int32_t *values)
{
    if (len < 10) return GVO_ERR_BOUNDS;
    int model = packing[1];
    if (model < 1 || model > 4) return GVO_ERR_FORMAT;
    int32_t seed = (int32_t)((uint32_t)packing[2] | ((uint32_t)packing[3] << 8) |
                             ((uint32_t)packing[4] << 16) | ((uint32_t)packing[5] << 24));
    uint32_t nM32 = (uint32_t)packing[6] | ((uint32_t)packing[7] << 8) |
                    ((uint32_t)packing[8] << 16) | ((uint32_t)packing[9] << 24);
    size_t nCells = (size_t)nRows * (size_t)nCols;
    uint8_t *m = (uint8_t *)calloc((size_t)nM32 + 6 * nCells + 8, 1);
    size_t got = 0;
    int rc = zinflate(packing + 10, len - 10, m, nM32, &got);
    if (rc == GVO_OK && got == 0) rc = GVO_DECLINED;   /* :143-148 returns null */
    if (rc == GVO_OK) rc = gvo_predictor_decode(model, seed, nRows, nCols, m, nM32, values);
    free(m);
    return rc;
}

/* ------------------------------------------------------------------ */
/* CodecFloat                                                          */
/* ------------------------------------------------------------------ */

/* compress/CodecFloat.java:300-313 */
static void float_encode_deltas(uint8_t *scratch, int nRows, int nCols)
{
    int prior0 = 0;
    size_t k = 0;
    for (int iRow = 0; iRow < nRows; iRow++) {
        int prior = prior0;
        prior0 = (int8_t)scratch[k];
        for (int iCol = 0; iCol < nCols; iCol++) {
            int test = (int8_t)scratch[k];
            scratch[k++] = (uint8_t)(test - prior);
            prior = test;
        }
    }
}

/* compress/CodecFloat.java:315-325 */
static void float_decode_deltas(uint8_t *scratch, int nRows, int nCols)
{
    int prior = 0;
    size_t k = 0;
    for (int iRow = 0; iRow < nRows; iRow++) {
        for (int iCol = 0; iCol < nCols; iCol++) {
            prior += (int8_t)scratch[k];
            scratch[k++] = (uint8_t)prior;
        }
        prior = (int8_t)scratch[(size_t)iRow * nCols];
    }
}

/* compress/CodecFloat.java:332-369 (before the doDeflate calls) */
int gvo_float_planes_encode(int nRows, int nCols, const uint32_t *c, uint8_t *planes)
{
    size_t n = (size_t)nRows * (size_t)nCols;
    size_t nSign = (n + 7) / 8;
    uint8_t *pSign = planes, *pExp = planes + nSign, *pM1 = pExp + n, *pM2 = pM1 + n, *pM3 = pM2 + n;
    memset(pSign, 0, nSign);
    for (size_t i = 0; i < n; i++) {
        if ((c[i] >> 31) & 1) pSign[i >> 3] |= (uint8_t)(1u << (i & 7));
        pExp[i] = (uint8_t)((c[i] >> 23) & 0xff);
        pM1[i] = (uint8_t)((c[i] >> 16) & 0x7f);
        pM2[i] = (uint8_t)((c[i] >> 8) & 0xff);
        pM3[i] = (uint8_t)(c[i] & 0xff);
    }
    float_encode_deltas(pM1, nRows, nCols);
    float_encode_deltas(pM2, nRows, nCols);
    float_encode_deltas(pM3, nRows, nCols);
    return GVO_OK;
}

/* compress/CodecFloat.java:395-458 (after the doInflate calls) */
int gvo_float_planes_decode(int nRows, int nCols, const uint8_t *planes, uint32_t *raw)
{
    size_t n = (size_t)nRows * (size_t)nCols;
    size_t nSign = (n + 7) / 8;
    uint8_t *tmp = (uint8_t *)malloc(3 * n + 1);
    if (!tmp) return GVO_ERR_ARG;
    const uint8_t *pSign = planes, *pExp = planes + nSign;
    memcpy(tmp, pExp + n, 3 * n);
    uint8_t *pM1 = tmp, *pM2 = tmp + n, *pM3 = tmp + 2 * n;
    float_decode_deltas(pM1, nRows, nCols);
    float_decode_deltas(pM2, nRows, nCols);
    float_decode_deltas(pM3, nRows, nCols);
    for (size_t i = 0; i < n; i++) {
        uint32_t r = (uint32_t)((pSign[i >> 3] >> (i & 7)) & 1) << 31;
        r |= (uint32_t)pExp[i] << 23;
        r |= (uint32_t)(pM1[i] & 0x7f) << 16;
        r |= (uint32_t)pM2[i] << 8;
        r |= (uint32_t)pM3[i];
        raw[i] = r;
    }
    free(tmp);
    return GVO_OK;
}

/* compress/CodecFloat.java:328-392 */
int gvo_codec_float_encode(int codecIndex, int nRows, int nCols,
                           const uint32_t *rawBits, int level, uint8_t *out,
                           size_t outCap, size_t *outLen)
{
    size_t n = (size_t)nRows * (size_t)nCols;
    size_t nSign = (n + 7) / 8;
    uint8_t *planes = (uint8_t *)malloc(nSign + 4 * n);
    uint8_t *z = (uint8_t *)malloc(n + 128);
    if (!planes || !z) { free(planes); free(z); return GVO_ERR_ARG; }
    gvo_float_planes_encode(nRows, nCols, rawBits, planes);
    size_t off = 2, planeOff = 0;
    int rc = GVO_OK;
    if (outCap < 2) rc = GVO_ERR_CAPACITY;
    else { out[0] = (uint8_t)codecIndex; out[1] = 0; }
    for (int p = 0; p < 5 && rc == GVO_OK; p++) {
        size_t pl = p == 0 ? nSign : n;
        size_t zn = 0;
        rc = zdeflate(planes + planeOff, pl, level, z, pl + 128, &zn);
        planeOff += pl;
        if (rc != GVO_OK) break;
        if (off + 4 + zn > outCap) { rc = GVO_ERR_CAPACITY; break; }
        for (int b = 0; b < 4; b++) out[off + b] = (uint8_t)((uint32_t)zn >> (8 * b));
        memcpy(out + off + 4, z, zn);
        off += 4 + zn;
    }
    if (outLen) *outLen = off;
    free(planes); free(z);
    return rc;
}

int gvo_codec_float_decode(int nRows, int nCols, const uint8_t *packing,
                           size_t len, uint32_t *rawBits)
{
    /* CodecFloat.decodeFloats :395-458, statement by statement: ONE scratch array serves all five planes
     * (`byte[] scratch = new byte[nCellsInTile]`), every doInflate writes only as many bytes as its stream gives, and
     * decodeDeltas works in place -- so behind a stream that ends early (damaged input) a plane keeps what the plane
     * before it left there (its delta-DECODED bytes for the mantissa planes), not zeros. */
    size_t n = (size_t)nRows * (size_t)nCols;
    size_t nSign = (n + 7) / 8;
    uint8_t *scratch = (uint8_t *)calloc(n + 8, 1);
    if (!scratch) return GVO_ERR_ARG;
    size_t off = 2;
    int rc = GVO_OK;
    for (size_t i = 0; i < n; i++) rawBits[i] = 0;
    for (int p = 0; p < 5; p++) {
        size_t pl = p == 0 ? nSign : n;
        if (off + 4 > len) { rc = GVO_ERR_BOUNDS; break; }
        uint32_t zn = (uint32_t)packing[off] | ((uint32_t)packing[off + 1] << 8) |
                      ((uint32_t)packing[off + 2] << 16) | ((uint32_t)packing[off + 3] << 24);
        off += 4;
        if (off + zn > len) { rc = GVO_ERR_BOUNDS; break; }
        size_t got = 0;
        rc = zinflate(packing + off, zn, scratch, pl, &got);
        if (rc != GVO_OK) break;
        off += zn;
        if (p >= 2) float_decode_deltas(scratch, nRows, nCols);
        for (size_t i = 0; i < n; i++) {
            switch (p) {
            case 0: rawBits[i] = (uint32_t)((scratch[i >> 3] >> (i & 7)) & 1) << 31; break;
            case 1: rawBits[i] |= (uint32_t)scratch[i] << 23; break;
            case 2: rawBits[i] |= (uint32_t)(scratch[i] & 0x7f) << 16; break;
            case 3: rawBits[i] |= (uint32_t)scratch[i] << 8; break;
            default: rawBits[i] |= (uint32_t)scratch[i]; break;
            }
        }
    }
    free(scratch);
    return rc;
}

/* ------------------------------------------------------------------ */
/* batch loops (CPU baseline)                                          */
/* ------------------------------------------------------------------ */

int gvo_batch_huffman_encode(int codecIndex, int nRows, int nCols, size_t nTiles,
                             const int32_t *values, uint8_t *out, size_t stride,
                             uint32_t *lengths, uint8_t *predictors)
{
    size_t nCells = (size_t)nRows * (size_t)nCols;
    for (size_t t = 0; t < nTiles; t++) {
        size_t len = 0;
        int used = 0;
        int rc = gvo_codec_huffman_encode(codecIndex, nRows, nCols, values + t * nCells,
                                          out + t * stride, stride, &len, 0xF, &used);
        if (rc == GVO_DECLINED) { len = 0; used = 0; }
        else if (rc != GVO_OK) return rc;
        lengths[t] = (uint32_t)len;
        if (predictors) predictors[t] = (uint8_t)used;
    }
    return GVO_OK;
}

int gvo_batch_huffman_decode(int nRows, int nCols, size_t nTiles,
                             const uint8_t *packings, size_t stride,
                             const uint32_t *lengths, int32_t *values)
{
    size_t nCells = (size_t)nRows * (size_t)nCols;
    for (size_t t = 0; t < nTiles; t++) {
        int rc = gvo_codec_huffman_decode(nRows, nCols, packings + t * stride, lengths[t],
                                          values + t * nCells);
        if (rc != GVO_OK) return rc;
    }
    return GVO_OK;
}

/* The CPU baseline on every host core: native threads, one contiguous share of the tiles each (tiles are independent,
 * gvrs/RasterTile.java:237-241).  Every thread encodes its tiles and decodes them again into scratch of its own and checks
 * the round trip; seconds[0] / seconds[1] receive the wall time of the encode and of the decode phase (all threads run a
 * phase together, a barrier in between).  Returns GVO_OK, or the first error / -100 for a round-trip mismatch.           */
#include <malloc.h>
#include <pthread.h>
#include <time.h>

typedef struct {
    int codecIndex, nRows, nCols;
    size_t t0, t1, stride;
    const int32_t *values;
    uint8_t *out;
    uint32_t *lengths;
    int32_t *back;
    pthread_barrier_t *bar;
    int rc;
} gvo_mt_job;

static void *gvo_mt_worker(void *arg)
{
    gvo_mt_job *j = (gvo_mt_job *)arg;
    size_t nCells = (size_t)j->nRows * (size_t)j->nCols;
    size_t n = j->t1 - j->t0;
    pthread_barrier_wait(j->bar);                           /* start of the encode phase */
    int rc = n ? gvo_batch_huffman_encode(j->codecIndex, j->nRows, j->nCols, n, j->values + j->t0 * nCells,
                                          j->out + j->t0 * j->stride, j->stride, j->lengths + j->t0, NULL) : GVO_OK;
    pthread_barrier_wait(j->bar);                           /* end of encode = start of decode */
    if (rc == GVO_OK && n)
        rc = gvo_batch_huffman_decode(j->nRows, j->nCols, n, j->out + j->t0 * j->stride, j->stride, j->lengths + j->t0,
                                      j->back + j->t0 * nCells);
    pthread_barrier_wait(j->bar);                           /* end of the decode phase */
    if (rc == GVO_OK && n && memcmp(j->back + j->t0 * nCells, j->values + j->t0 * nCells, n * nCells * 4) != 0) rc = -100;
    j->rc = rc;
    return NULL;
}

static double gvo_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int gvo_huffman_roundtrip_threads(int nThreads, int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                                  uint8_t *out, size_t stride, uint32_t *lengths, int32_t *back, double *seconds)
{
    if (nThreads < 1 || nThreads > 4096) return GVO_ERR_ARG;
    /* the per-tile work buffers of the restatement are malloc'ed per call; above glibc's mmap threshold every one of them is
     * an mmap + page faults + munmap, and a few hundred threads then queue on the kernel's address-space lock instead of
     * computing.  Keep them in the (per-thread) malloc arenas. */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nThreads);
    gvo_mt_job *job = (gvo_mt_job *)malloc(sizeof(gvo_mt_job) * (size_t)nThreads);
    pthread_barrier_t bar;
    if (!th || !job || pthread_barrier_init(&bar, NULL, (unsigned)nThreads + 1u) != 0) {
        free(th);
        free(job);
        return GVO_ERR_ARG;
    }
    int started = 0;
    for (int i = 0; i < nThreads; i++) {
        gvo_mt_job *j = &job[i];
        j->codecIndex = codecIndex; j->nRows = nRows; j->nCols = nCols;
        j->t0 = nTiles * (size_t)i / (size_t)nThreads;
        j->t1 = nTiles * (size_t)(i + 1) / (size_t)nThreads;
        j->stride = stride; j->values = values; j->out = out; j->lengths = lengths; j->back = back; j->bar = &bar; j->rc = GVO_OK;
        if (pthread_create(&th[i], NULL, gvo_mt_worker, j) != 0) break;
        started++;
    }
    int rc = GVO_OK;
    if (started == nThreads) {
        pthread_barrier_wait(&bar);
        double a = gvo_now();
        pthread_barrier_wait(&bar);
        double b = gvo_now();
        pthread_barrier_wait(&bar);
        double c = gvo_now();
        if (seconds) { seconds[0] = b - a; seconds[1] = c - b; }
        for (int i = 0; i < nThreads; i++) pthread_join(th[i], NULL);
        for (int i = 0; i < nThreads; i++) if (job[i].rc != GVO_OK && rc == GVO_OK) rc = job[i].rc;
    } else {
        rc = GVO_ERR_ARG;                                   /* could not start every thread: the started ones wait at the barrier */
        for (int i = 0; i < started; i++) pthread_cancel(th[i]);
        for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    }
    pthread_barrier_destroy(&bar);
    free(th);
    free(job);
    return rc;
}

/* ------------------------------------------------------------------ */
/* synthetic DEM (integer value noise)           */
/* ------------------------------------------------------------------ */

uint64_t gvo_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline int32_t dem_lattice(uint64_t seed, int o, int64_t i, int64_t j)
{
    uint64_t h = gvo_splitmix64(seed ^ ((uint64_t)o << 56) ^
                                (((uint64_t)j & 0xFFFFFFFULL) << 28) ^ ((uint64_t)i & 0xFFFFFFFULL));
    int32_t amp = 4096 >> o;
    return (int32_t)((((h >> 32) & 0xFFFF) * (uint64_t)(2 * amp)) >> 16) - amp;
}

int32_t gvo_dem_value(uint64_t seed, int64_t gx, int64_t gy)
{
    int64_t sum = 0;
    for (int o = 0; o < 6; o++) {
        int sh = 8 - o;
        int64_t s = (int64_t)1 << sh;
        int64_t i = gx >> sh, j = gy >> sh;
        int64_t fx = gx & (s - 1), fy = gy & (s - 1);
        int64_t l00 = dem_lattice(seed, o, i, j), l10 = dem_lattice(seed, o, i + 1, j);
        int64_t l01 = dem_lattice(seed, o, i, j + 1), l11 = dem_lattice(seed, o, i + 1, j + 1);
        int64_t top = l00 * (s - fx) + l10 * fx;
        int64_t bot = l01 * (s - fx) + l11 * fx;
        sum += (top * (s - fy) + bot * fy) >> (2 * sh);   /* arithmetic shift == floor */
    }
    /* per-cell jitter in [-2, 2]: keeps the residual entropy DEM-like (about 3 bits) */
    uint64_t h = gvo_splitmix64(seed ^ 0x7700000000000000ULL ^
                                (((uint64_t)gy & 0xFFFFFFFULL) << 28) ^ ((uint64_t)gx & 0xFFFFFFFULL));
    sum += (int64_t)((h >> 40) % 5) - 2;
    sum -= 2000;
    if (sum < -11000) sum = -11000;      /* demo/globalDEM/PackageData.java:113-114 limits */
    if (sum > 8848) sum = 8848;
    return (int32_t)sum;
}

/* ocean mask of the nulls workload ("a variant with 5 % INT4_NULL_CODE ocean-mask blocks"): the grid is
 * cut into 16 x 16 blocks; a block is masked out entirely when its hash falls below maskPerMille / 1000 */
int gvo_dem_masked(uint64_t seed, int64_t gx, int64_t gy, int maskPerMille)
{
    if (maskPerMille <= 0) return 0;
    uint64_t h = gvo_splitmix64(seed ^ 0x3300000000000000ULL ^
                                ((((uint64_t)gy >> 4) & 0xFFFFFFFULL) << 28) ^ (((uint64_t)gx >> 4) & 0xFFFFFFFULL));
    return (int)((h >> 33) % 1000u) < maskPerMille;
}

/* The ROUGH surface (style 1; round 4, SURVEY.md section 8d: "neighbouring-cell differences mostly within +-126 with a tail into
 * 2-3 byte codes").  The classic surface above is one kind of terrain everywhere: the Triangle predictor wins
 * every tile and no residual needs a second M32 byte.  Here the grid is cut into PROVINCES of 1024 x 1024 cells, each of one kind:
 *   mountains (about half)   the classic surface
 *   plains                   a sixteenth of the largest octave's relief under jitter of -6..6: noise dominates, and first
 *                            differences (PredictorModelDifferencing) carry the least of it
 *   stripes                  every grid row samples the classic surface 37 rows further on, without jitter: smooth along a row,
 *                            unrelated from row to row (a push-broom sensor with an offset per line) -- the second difference along
 *                            the row (PredictorModelLinear) is what predicts well
 * and on hashed 16 x 16 blocks of the grid there is steeper ground still: CLIFF blocks (12 of 64 mountain blocks, 4 of 64 in the
 * stripes, none in the plains) add an octave of lattice spacing 4 and amplitude 520 -- slopes beyond +-126 per cell (two M32 bytes,
 * CodecM32.java:270-311) and, at its steepest and along the block's edges, beyond +-254 (three); SCREE blocks (another 8 of 64 in
 * the mountains) add white noise of -150..150 per cell.  Measured on the ETOPO1-shaped grid (tools/rough_stats.py, 200 tiles):
 * 4.5 % of the row differences need two M32 bytes, 0.65 % three; Differencing / Linear / Triangle win 19 / 42 / 39 % of the tiles;
 * 82 % of the tiles hold at least one multi-byte value. */
#define DEM_PROVINCE_SHIFT 10
#define DEM_CLIFF_AMP 520
#define DEM_SCREE_AMP 150
static inline int dem_province(uint64_t seed, int64_t gx, int64_t gy)
{
    uint64_t h = gvo_splitmix64(seed ^ 0x5500000000000000ULL ^
                                ((((uint64_t)gy >> DEM_PROVINCE_SHIFT) & 0xFFFFFFFULL) << 28) ^ (((uint64_t)gx >> DEM_PROVINCE_SHIFT) & 0xFFFFFFFULL));
    uint32_t k = (uint32_t)((h >> 33) % 100u);
    return k < 50u ? 0 : k < 72u ? 1 : 2;                /* mountains, plains, stripes */
}

static int64_t dem_octaves(uint64_t seed, int64_t gx, int64_t gy, int first, int last)
{
    int64_t sum = 0;
    for (int o = first; o < last; o++) {
        int sh = 8 - o;
        int64_t s = (int64_t)1 << sh;
        int64_t i = gx >> sh, j = gy >> sh;
        int64_t fx = gx & (s - 1), fy = gy & (s - 1);
        int64_t l00 = dem_lattice(seed, o, i, j), l10 = dem_lattice(seed, o, i + 1, j);
        int64_t l01 = dem_lattice(seed, o, i, j + 1), l11 = dem_lattice(seed, o, i + 1, j + 1);
        int64_t top = l00 * (s - fx) + l10 * fx;
        int64_t bot = l01 * (s - fx) + l11 * fx;
        sum += (top * (s - fy) + bot * fy) >> (2 * sh);
    }
    return sum;
}

int32_t gvo_dem_value_style(uint64_t seed, int64_t gx, int64_t gy, int style)
{
    if (style != 1) return gvo_dem_value(seed, gx, gy);
    const int kind = dem_province(seed, gx, gy);
    uint64_t hj = gvo_splitmix64(seed ^ 0x7700000000000000ULL ^
                                 (((uint64_t)gy & 0xFFFFFFFULL) << 28) ^ ((uint64_t)gx & 0xFFFFFFFULL));
    int64_t sum;
    uint32_t cliffShare, screeShare;                     /* of 64 */
    if (kind == 0) {
        sum = dem_octaves(seed, gx, gy, 0, 6) + (int64_t)((hj >> 40) % 5) - 2;
        cliffShare = 12; screeShare = 8;
    } else if (kind == 1) {
        sum = (dem_octaves(seed, gx, gy, 0, 1) >> 4) + (int64_t)((hj >> 40) % 13) - 6;
        cliffShare = 0; screeShare = 0;
    } else {
        sum = dem_octaves(seed, gx, gy * 37, 0, 6);
        cliffShare = 4; screeShare = 0;
    }
    uint64_t hb = gvo_splitmix64(seed ^ 0x6600000000000000ULL ^
                                 ((((uint64_t)gy >> 4) & 0xFFFFFFFULL) << 28) ^ (((uint64_t)gx >> 4) & 0xFFFFFFFULL));
    if ((uint32_t)((hb >> 33) & 63u) < cliffShare) {
        /* one more octave, bilinear like the others: lattice spacing 4, amplitude DEM_CLIFF_AMP */
        int64_t i = gx >> 2, j = gy >> 2, fx = gx & 3, fy = gy & 3;
        int64_t l[4];
        for (int q = 0; q < 4; q++) {
            uint64_t h = gvo_splitmix64(seed ^ 0x4400000000000000ULL ^
                                        ((((uint64_t)(j + (q >> 1))) & 0xFFFFFFFULL) << 28) ^ (((uint64_t)(i + (q & 1))) & 0xFFFFFFFULL));
            l[q] = (int64_t)((((h >> 32) & 0xFFFF) * (uint64_t)(2 * DEM_CLIFF_AMP)) >> 16) - DEM_CLIFF_AMP;
        }
        int64_t top = l[0] * (4 - fx) + l[1] * fx, bot = l[2] * (4 - fx) + l[3] * fx;
        sum += (top * (4 - fy) + bot * fy) >> 4;
    } else if ((uint32_t)((hb >> 33) & 63u) < cliffShare + screeShare) {
        /* scree: white noise of -DEM_SCREE_AMP .. DEM_SCREE_AMP per cell */
        sum += (int64_t)((hj >> 20) % (2u * DEM_SCREE_AMP + 1u)) - DEM_SCREE_AMP;
    }
    sum -= 2000;
    if (sum < -11000) sum = -11000;
    if (sum > 8848) sum = 8848;
    return (int32_t)sum;
}

void gvo_dem_fill_tiles_style(uint64_t seed, int nRows, int nCols, int64_t tilesPerRow,
                              int64_t tile0, int64_t nTiles, int maskPerMille, int style, int32_t *values)
{
    size_t nCells = (size_t)nRows * (size_t)nCols;
    for (int64_t t = 0; t < nTiles; t++) {
        int64_t tile = tile0 + t;
        int64_t tr = tile / tilesPerRow, tc = tile % tilesPerRow;
        int32_t *v = values + (size_t)t * nCells;
        for (int r = 0; r < nRows; r++) {
            for (int c = 0; c < nCols; c++) {
                int64_t gx = tc * nCols + c, gy = tr * nRows + r;
                v[(size_t)r * nCols + c] = gvo_dem_masked(seed, gx, gy, maskPerMille) ? (int32_t)0x80000000u : gvo_dem_value_style(seed, gx, gy, style);
            }
        }
    }
}

void gvo_dem_fill_tiles_masked(uint64_t seed, int nRows, int nCols, int64_t tilesPerRow,
                               int64_t tile0, int64_t nTiles, int maskPerMille, int32_t *values)
{
    size_t nCells = (size_t)nRows * (size_t)nCols;
    for (int64_t t = 0; t < nTiles; t++) {
        int64_t tile = tile0 + t;
        int64_t tr = tile / tilesPerRow, tc = tile % tilesPerRow;
        int32_t *v = values + (size_t)t * nCells;
        for (int r = 0; r < nRows; r++) {
            for (int c = 0; c < nCols; c++) {
                int64_t gx = tc * nCols + c, gy = tr * nRows + r;
                v[(size_t)r * nCols + c] = gvo_dem_masked(seed, gx, gy, maskPerMille) ? (int32_t)0x80000000u : gvo_dem_value(seed, gx, gy);
            }
        }
    }
}

void gvo_dem_fill_tiles(uint64_t seed, int nRows, int nCols, int64_t tilesPerRow,
                        int64_t tile0, int64_t nTiles, int32_t *values)
{
    size_t nCells = (size_t)nRows * (size_t)nCols;
    for (int64_t t = 0; t < nTiles; t++) {
        int64_t tile = tile0 + t;
        int64_t tr = tile / tilesPerRow, tc = tile % tilesPerRow;
        int32_t *v = values + (size_t)t * nCells;
        for (int r = 0; r < nRows; r++) {
            for (int c = 0; c < nCols; c++) {
                v[(size_t)r * nCols + c] = gvo_dem_value(seed, tc * nCols + c, tr * nRows + r);
            }
        }
    }
}
