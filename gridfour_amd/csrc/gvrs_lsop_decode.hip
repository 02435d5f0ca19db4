// gvrs_lsop_decode.hip -- entropy stage of LsDecoder12.decode for the canonical-Huffman container
// (lsop/LsDecoder12.java:107-119, lsop/LsHeader.java:131-185): header, then two CanonicalHuffman streams in one
// bit store (initialisers, interior).  Output: seed + coefficients and the residual ints that
// k_lsop_reconstruct (gvrs_lsop.hip) turns into the tile.  Containers of type 0 (legacy Huffman of M32) and
// type 1 (Deflate of M32) are left marked GF_K_ERR_UNSUPPORTED here for k_lsop_unpack_m32 (gvrs_decode.hip), which
// shares the legacy Huffman / M32 device code of CodecHuffman.

#include <hip/hip_runtime.h>

#include "gvrs_kernels.h"
#include "huff_build.h"

namespace {

// 512 threads, one subsequence of a stream per thread, 64 VGPRs (round 3): four workgroups = all 32 wave slots of a CU where the
// LDS footprint allows four (18.5 KB of tables + the text + 4 KB of token table: ETOPO1-shaped tiles 37 KB); with 256 threads
// the kernel ran five workgroups = 20 waves per CU
constexpr int DEC_THREADS = 512;
constexpr int DEC_WAVES = DEC_THREADS / 64;

#include "gvrs_decode_common.h"
#include "gvrs_canon_decode_common.h"

struct GfLsopUnpackArgs {
    const uint8_t *blob;
    size_t blobBytes;
    const uint64_t *offsets;
    size_t slotStride;
    const uint32_t *lengths;
    int32_t *residuals;
    size_t resStride;
    uint32_t *coefs;
    int32_t *status;
    size_t nTiles;
    int nRows, nCols;
    uint32_t ldsTextBytes;
    const uint32_t *pre;       // code-length records of the first stream (k_canon_parse_lengths), or null
};

__global__ __launch_bounds__(DEC_THREADS, 8) void k_lsop_unpack2(GfLsopUnpackArgs a)
{
    __shared__ CanonDec S;

    const int tid = threadIdx.x;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols;
    const uint32_t nInit = 4u * nR + 2u * nC - 9u, nInt = (nR - 2u) * (nC - 4u);
    const uint32_t *__restrict__ w32 = reinterpret_cast<const uint32_t *>(a.blob);
    const uint64_t nWords = (a.blobBytes + 3) >> 2;
    const uint32_t capWords = a.ldsTextBytes >> 2;

    GF_FOR_WG_TILE(t, a.nTiles) {                                         // no tile loop: see gvrs_kernels.h
        const uint64_t off = a.offsets ? a.offsets[t] : (uint64_t)t * a.slotStride;
        const uint32_t len = a.lengths[t];
        const uint8_t *__restrict__ pk = a.blob + off;
        int32_t *res = a.residuals + t * a.resStride;

        int32_t early = GF_K_OK;
        if (len < 3 || off + len > a.blobBytes) early = GF_K_ERR_BOUNDS;
        else if (!(pk[1] & 0x40)) early = GF_K_ERR_UNSUPPORTED;            // legacy header (LsHeader.java:139-160)
        else if ((pk[1] & 0x0f) != 2) early = GF_K_ERR_UNSUPPORTED;        // Huffman-of-M32 / Deflate containers
        else if (pk[2] != 12) early = GF_K_ERR_FORMAT;                     // u[11] would index out of bounds
        else {
            const uint32_t hdr = 55u + ((pk[1] & 0x80) ? 4u : 0u);         // value checksum, if present, is skipped
            if (len < hdr) early = GF_K_ERR_BOUNDS;
        }
        if (early != GF_K_OK) {
            if (tid == 0) a.status[t] = early;
            __syncthreads();
            continue;
        }
        const uint32_t hdr = 55u + ((pk[1] & 0x80) ? 4u : 0u);
        if (tid < 13) {
            const uint8_t *p = pk + 3 + 4 * tid;
            a.coefs[t * 16 + tid] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        }

        const uint64_t word0 = off >> 2;
        const uint32_t bias = (uint32_t)(off & 3u) * 8u;
        const uint32_t endBit = bias + len * 8u;
        const uint32_t needWords = (endBit + 31u) / 32u + 4u;       // the readers look up to three words ahead
        const bool textInLds = needWords <= capWords;
        if (textInLds) cd_stage_text(w32, word0, nWords, endBit, needWords);
        const CdTextLds TL{needWords};
        const CdTextGlobal TG{w32 + word0, (uint32_t)min((uint64_t)needWords, nWords - word0)};   // huge packing: read in place
        __syncthreads();

        uint32_t pos = bias + hdr * 8u, nv;
        const CdArraySink sink0{res, nInit};
        const uint32_t *pre = a.pre ? a.pre + t * GF_CANON_REC_WORDS : nullptr;
        uint16_t *const tok = reinterpret_cast<uint16_t *>(cdLdsText + capWords);      // token table of the synchronisation passes
        // The first stream ends where its end-of-text symbol says, which nobody knows beforehand: synchronising "to the end of
        // the text" meant the whole packing, i.e. a full pass over the SECOND stream's bits with the first stream's code
        // (a seventh of the kernel's instructions, round 3).  So the first attempt looks at 16 bits per initialiser + 1 Kbit
        // only; a stream that is longer (no end-of-text symbol inside: any failure of the short attempt) is decoded again
        // with the whole packing in view -- same values, same status as before either way.
        const uint32_t pos0 = pos;
        const uint32_t endShort = min(endBit, pos0 + 16u * nInit + 1024u);
        int32_t st = GF_K_OK;
        for (int attempt = 0; attempt < 2; attempt++) {
            const uint32_t e = attempt == 0 ? endShort : endBit;
            pos = pos0;
            st = textInLds ? cd_decode_stream(S, TL, pos0, e, nInit, nInit, sink0, &pos, &nv, nullptr, pre, bias, tok)
                           : cd_decode_stream(S, TG, pos0, e, nInit, nInit, sink0, &pos, &nv, nullptr, pre, bias);
            st = (int32_t)GF_UNI((uint32_t)st);                   // (the same in every thread: a scalar, so that this is a scalar loop)
            pos = GF_UNI(pos);
            if (st == GF_K_OK || e == endBit) break;
        }
        if (st == GF_K_OK) {
            // the interior residuals go through the byte stage of the canonical decoder (round 3): a thread walks its own stretch
            // of the stream, so its 4-byte stores hit 64 different lines per instruction -- 3.7 GB of HBM writes for 0.93 GB of
            // residuals on the ETOPO1-shaped batch (profiles/hbm_traffic.json).  The stage lies over the four sync arrays (one
            // subsequence per thread: dead by then), the token table and the unused end of the text buffer; what does not fit goes straight out
            // ... and the part of the text buffer this packing does not fill (the buffer is sized for 3/4 byte per cell)
            const uint32_t usedWords = textInLds ? min(needWords, capWords) : 0u;
            const CdCellSink sink1{reinterpret_cast<uint32_t *>(res + nInit), GfCellMap::make(4, 1u, 2u), nInt, true,
                                   reinterpret_cast<uint8_t *>(S.qs), reinterpret_cast<uint8_t *>(cdLdsText + usedWords),
                                   (uint32_t)(4 * sizeof(S.qe)), (uint32_t)(4 * sizeof(S.qe)) + (capWords - usedWords) * 4u + 4096u, 0u, false};
            static_assert(CD_NCUR == 1, "the stage over all four sync arrays needs one subsequence per thread");
            st = textInLds ? cd_decode_stream(S, TL, pos, endBit, nInt, nInt, sink1, &pos, &nv, nullptr, nullptr, 0, tok)
                           : cd_decode_stream(S, TG, pos, endBit, nInt, nInt, sink1, &pos, &nv);
        }
        if (tid == 0) a.status[t] = st;
        __syncthreads();
    }
}

}  // namespace

uint32_t gf_lsop_unpack_lds_text(int nRows, int nCols)
{
    // LDS copy of the packing, as in the canonical decoder (3/4 byte per cell + 1 KB; larger packings are read in place) -- but
    // this kernel has no value stage behind it, and at 46 VGPRs the LDS alone decides how many workgroups a CU holds: where the
    // usual size lands just above a fifth of the CU's LDS (ETOPO1-shaped tiles: 18.5 KB of tables + 14.5 KB of text = 33 KB)
    // the copy is trimmed to what still lets five workgroups in, as long as that leaves half a byte per cell: LSOP12 decode of
    // the ETOPO1-shaped batch 3.73 -> 3.38 ms
    const size_t cells = (size_t)nRows * (size_t)nCols;
    size_t want = cells - cells / 4 + 1024;
    if (want > 96 * 1024) want = 96 * 1024;
    want = (want + 31) & ~(size_t)31;
    // round 3 (512 threads): a quarter of the CU's LDS per workgroup is the tier to stay in -- tables + text + the 4 KB token table
    // <= 40,448 bytes where half a byte per cell + 1 KB of text still fit (ETOPO1-shaped tiles: 37 KB, nothing to trim)
    const size_t quarter = 40 * 1024 - 512, fixed = sizeof(CanonDec) + (sizeof(uint16_t) << CD_LUT_BITS);
    if (quarter > fixed) {
        const size_t room = (quarter - fixed) & ~(size_t)31;
        if (room < want && room >= cells / 2 + 1024) want = room;
    }
    return (uint32_t)want;
}

hipError_t gf_launch_lsop_unpack2(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, size_t slotStride,
                                  const uint32_t *lengths, int32_t *residuals, size_t resStride, uint32_t *coefs,
                                  int32_t *status, size_t nTiles, int nRows, int nCols, uint32_t ldsTextBytes, unsigned grid,
                                  hipStream_t stream, const uint32_t *pre)
{
    if (nTiles == 0) return hipSuccess;
    GfLsopUnpackArgs a{blob, blobBytes, offsets, slotStride, lengths, residuals, resStride, coefs, status, nTiles, nRows, nCols,
                       ldsTextBytes, pre};
    static GfDynLdsOptIn opt;
    {
        const hipError_t e = gf_opt_in_dyn_lds(k_lsop_unpack2, ldsTextBytes + (sizeof(uint16_t) << CD_LUT_BITS), opt);
        if (e != hipSuccess) return e;
    }
    (void)grid;
    hipLaunchKernelGGL(k_lsop_unpack2, gf_tile_grid(nTiles), dim3(DEC_THREADS), ldsTextBytes + (sizeof(uint16_t) << CD_LUT_BITS), stream, a);
    return hipGetLastError();
}
