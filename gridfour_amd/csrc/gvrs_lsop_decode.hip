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
    const uint32_t *pre2;      // k_lsop_head's records of the second stream, or null (word 3 != 0: the tile's first stream is done)
    uint32_t *debug;           // diagnostic flavour: 16 cycle stamps per tile (tools/phase_cycles_lsop.py), normally null
    GfLsopPlaneGeom plane;     // plane.ok: the interior residuals of a tile whose values are all bytes leave as a byte plane in pipeline
                               // order (gvrs_kernels.h), word GF_LSOP_FMT_WORD of the tile's coefficient record says so
};


// ------------------------------------------------------------------------------------------------
// k_lsop_head (round 6): what is SERIAL in an LSOP12 container of the canonical type, ONE LANE PER TILE, in front of k_lsop_unpack2.
// The container's two streams lie back to back in one bit store (LsEncoder12.java:148-157): the second one's code lengths start
// where the first one's end-of-text symbol says.  Inside k_lsop_unpack2 that meant, per tile and with seven of the workgroup's eight
// waves waiting: a full canonical decode of the 4 nR + 2 nC - 9 initialisers (tables, lookup table, synchronisation: 73 K cycles
// for 771 values) and then the walk over the second stream's length tokens by one wave (50 K cycles) -- 123 K of a tile's 219 K
// cycles (tools/phase_cycles_lsop.py).  Here a lane walks its tile alone, as k_canon_parse_lengths does for the first stream's
// lengths: canonical tables from those lengths (a counting sort; the length of a code by fifteen compares), the initialisers symbol by symbol
// (CanonicalHuffman.decodeText :469-519) straight to the tile's residual array, then the second stream's lengths
// (cd_lane_parse_lengths) into a record of the same layout -- word 3 of it says that this tile's first stream is done.
// Anything but the plain case -- a status other than OK anywhere, a count of values other than the reader's, a code the table of a
// damaged packing does not hold -- leaves word 3 zero: k_lsop_unpack2 then decodes the tile from its first bit as before and owns
// every status.  Sixty-four packings' first 1,280 bytes are staged in LDS (coalesced); a walk that leaves them reads the packing.
// ------------------------------------------------------------------------------------------------
struct GfLsopHeadArgs {
    const uint8_t *blob;
    size_t blobBytes;
    const uint64_t *offsets;
    size_t slotStride;
    const uint32_t *lengths;
    const uint32_t *rec1;      // k_canon_parse_lengths' records of the first stream (GF_CANON_REC_WORDS per tile)
    uint32_t *rec2;            // out: the same for the second stream; word 3: 1 = the initialisers are in `residuals`
    int32_t *residuals;
    size_t resStride;
    size_t nTiles;
    uint32_t nInit;
    uint32_t *debug;           // diagnostic flavour: cycle stamps 1..5 of a tile's sixteen (tools/phase_cycles_lsop.py)
};
constexpr uint32_t LH_STAGE_WORDS = 320;                      // 1,280 bytes of every packing
constexpr uint32_t LH_ORDER = 264;                            // symbols in (length, symbol) order: 261, padded
constexpr uint32_t LH_LEN_WORDS = 68;                         // 272 bytes of code lengths
constexpr size_t LH_LDS_BYTES = (size_t)64 * (LH_STAGE_WORDS * 4 + LH_ORDER * 2 + LH_LEN_WORDS * 4 + 32 * 4);

__global__ __launch_bounds__(64) void k_lsop_head(GfLsopHeadArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lhLds[];
    uint32_t *stage = reinterpret_cast<uint32_t *>(lhLds);                                          // [LH_STAGE_WORDS][64]
    uint16_t *order = reinterpret_cast<uint16_t *>(lhLds + (size_t)64 * LH_STAGE_WORDS * 4);        // [LH_ORDER][64]
    uint32_t *lensW = reinterpret_cast<uint32_t *>(lhLds + (size_t)64 * (LH_STAGE_WORDS * 4 + LH_ORDER * 2));   // [LH_LEN_WORDS][64]
    uint32_t *cntW = lensW + (size_t)64 * LH_LEN_WORDS;                                             // [32][64]: symbols per length, running places
    const uint32_t lane = threadIdx.x;
    const size_t t0 = (size_t)blockIdx.x * 64, t = t0 + lane;
    const bool inBatch = t < a.nTiles;
    const uint64_t off = !inBatch ? 0ull : a.offsets ? a.offsets[t] : (uint64_t)t * a.slotStride;
    const uint32_t len = inBatch ? a.lengths[t] : 0u;
    const uint32_t *r1 = a.rec1 + (inBatch ? t : 0) * GF_CANON_REC_WORDS;
    uint32_t *r2 = a.rec2 + (inBatch ? t : 0) * GF_CANON_REC_WORDS;
    // (the first stream's record says whether the packing is readable at all: its header checked, its lengths parsed)
    int32_t st = inBatch && len >= 59u && off + len <= a.blobBytes ? (int32_t)r1[0] : (int32_t)GF_K_ERR_BOUNDS;
    {
        const size_t n = min((size_t)64, a.nTiles - t0) * GF_CANON_REC_WORDS;      // records start out zero
        uint32_t *z = a.rec2 + t0 * GF_CANON_REC_WORDS;
        for (size_t i = lane; i < n; i += 64) z[i] = 0;
        __threadfence_block();
    }
#ifdef GF_DIAG
#define LH_STAMP(i) do { if (a.debug && inBatch) a.debug[t * 16 + (i)] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define LH_STAMP(i) do { } while (0)
#endif
    LH_STAMP(1);
    // ---- the packings' heads into LDS: the wave fetches tile after tile, a lane a word (see k_canon_parse_lengths) ----
    {
        const uint32_t visMine = st == GF_K_OK ? min(len, LH_STAGE_WORDS * 4u) : 0u;
        auto staged_at = [](uint32_t vis, uint32_t k) -> uint32_t { return 4u * k < vis ? min(4u * k, vis - 4u) : 0u; };   // vis >= 7
        auto staged_fix = [](uint32_t w, uint32_t vis, uint32_t k) -> uint32_t {
            return 4u * k < vis ? w >> (8u * (4u * k - min(4u * k, vis - 4u))) : 0u;
        };
        constexpr uint32_t TURN = 8, PER = LH_STAGE_WORDS / 64;
        for (uint32_t j0 = 0; j0 < 64u; j0 += TURN) {
            uint32_t w[TURN][PER], visJ[TURN];
#pragma unroll
            for (uint32_t u = 0; u < TURN; u++) {
                const uint32_t j = j0 + u;
                const uint32_t vis = (uint32_t)__builtin_amdgcn_readlane((int)visMine, (int)j);
                const uint64_t offJ = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), (int)j) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, (int)j);
                const uint8_t *__restrict__ pkJ = a.blob + (vis ? offJ : 0ull);
                visJ[u] = vis;
#pragma unroll
                for (uint32_t m = 0; m < PER; m++) w[u][m] = reinterpret_cast<const CdPackedWord *>(pkJ + staged_at(vis, 64u * m + lane))->v;
            }
#pragma unroll
            for (uint32_t u = 0; u < TURN; u++)
#pragma unroll
                for (uint32_t m = 0; m < PER; m++) asm volatile("" : "+v"(w[u][m]));
#pragma unroll
            for (uint32_t u = 0; u < TURN; u++)
#pragma unroll
                for (uint32_t m = 0; m < PER; m++) stage[(64u * m + lane) * 64u + j0 + u] = staged_fix(w[u][m], visJ[u], 64u * m + lane);
        }
        // ... and every lane its own tile's code lengths (272 bytes of the first stream's record)
        for (uint32_t k = 0; k < LH_LEN_WORDS; k++) lensW[k * 64u + lane] = st == GF_K_OK ? r1[8 + k] : 0u;
    }
    __syncthreads();
    LH_STAMP(2);
    // (a lane without a readable packing stays in the wave's loops: to it the packing is empty)
    const uint32_t lenSafe = st == GF_K_OK ? len : 0u;
    const uint8_t *__restrict__ pk = a.blob + (st == GF_K_OK ? off : 0ull);
    const uint32_t endBit = lenSafe * 8u;
    auto peek = [&](uint32_t pos) -> uint32_t {               // 32 bits of the packing from bit pos, zero beyond its end
        const uint32_t wi = pos >> 5;
        if (wi + 1u < LH_STAGE_WORDS) return __builtin_amdgcn_alignbit(stage[(wi + 1u) * 64u + lane], stage[wi * 64u + lane], pos & 31u);
        const uint32_t b = pos >> 3;
        uint64_t w = 0;
        if (b + 8u <= lenSafe) {
            w = ((uint64_t)reinterpret_cast<const CdPackedWord *>(pk + b + 4)->v << 32) | reinterpret_cast<const CdPackedWord *>(pk + b)->v;
        } else {
            for (uint32_t k = 0; k < 8; k++)
                if (b + k < lenSafe) w |= (uint64_t)pk[b + k] << (8u * k);
        }
        return (uint32_t)(w >> (pos & 7u));
    };
    // ---- canonical tables of the first stream (CanonHuffTreeDecoder.java:68-131).  Everything a symbol's decode needs without a
    //      branch that a single lane of the sixty-four could drag the others into (a first-level table with a search behind it was
    //      exactly that: some lane's code was always a long one, and every symbol cost the wave both paths -- 1,800 cycles):
    //      the codes of length l are the left-aligned 15-bit values below limit[l] (limits never decrease); a code's length is one
    //      more than the number of limits at or below it -- fifteen compares on registers --, and its symbol stands at
    //      order[adj[l] + (code >> (15 - l))], adj[l] = (first place of length l) - (first code of length l). ----
    bool ok = st == GF_K_OK;
    uint32_t limit[16], nOfLen[16];
    for (uint32_t l = 0; l < 32u; l++) cntW[l * 64u + lane] = 0u;
    // symbols per length: an LDS counter per length and lane (an addition without a result: nothing waits for it)
    for (uint32_t i = 0; i <= (uint32_t)CN_SYMS; i += 4) {
        const uint32_t w4 = lensW[(i >> 2) * 64u + lane];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t l = (w4 >> (8u * q)) & 0xffu;
            const bool inRange = i + q <= (uint32_t)CN_SYMS;
            ok = ok && (!inRange || l <= 15u);
            if (inRange && l >= 1u && l <= 15u) atomicAdd(&cntW[l * 64u + lane], 1u);
        }
    }
    nOfLen[0] = 0;
#pragma unroll
    for (int l = 1; l < 16; l++) nOfLen[l] = cntW[l * 64 + lane];
    int32_t adjOf[16];
    {
        uint32_t run = 0, code = 0, kraft = 0;
#pragma unroll
        for (int l = 1; l < 16; l++) {
            cntW[(16 + l) * 64 + lane] = run;                  // the running place of length l (rows 16..31)
            adjOf[l] = (int32_t)run - (int32_t)code;
            limit[l] = (code + nOfLen[l]) << (15 - l);
            run += nOfLen[l];
            kraft += nOfLen[l] << (15 - l);
            code = (code + nOfLen[l]) << 1;
        }
        ok = ok && run >= 2u && kraft == (1u << 15);          // (a complete prefix code: what an encoder writes)
    }
    // the symbols in (length, symbol) order: a counting sort (the place comes back from the counter's addition)
    for (uint32_t i = 0; i <= (uint32_t)CN_SYMS; i += 4) {
        const uint32_t w4 = lensW[(i >> 2) * 64u + lane];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t l = (w4 >> (8u * q)) & 0xffu;
            if (ok && i + q <= (uint32_t)CN_SYMS && l != 0u) {
                const uint32_t slot = atomicAdd(&cntW[(16u + l) * 64u + lane], 1u);
                order[min(slot, LH_ORDER - 1u) * 64u + lane] = (uint16_t)(i + q);
            }
        }
    }
    LH_STAMP(3);
    // ---- the initialisers, symbol by symbol (CanonicalHuffman.decodeText :469-519).  The text runs through a 64-bit register window
    //      that is refilled a word at a time from the staged head ----
    int32_t *res = a.residuals + (inBatch ? t : 0) * a.resStride;
    uint32_t pos = ok ? r1[1] : 0u, k = 0, prior = 0;
    // (from the staged head only: a load from the packing inside the loop makes every turn wait for the turn before's store -- one
    // counter for loads and stores, and the compiler cannot tell at the loop's end whether a load is pending.  A first stream that
    // reaches beyond the staged 1,280 bytes -- no terrain tile's does -- is left to k_lsop_unpack2.)
    auto wordAt = [&](uint32_t wi) -> uint32_t { return stage[min(wi, LH_STAGE_WORDS - 1u) * 64u + lane]; };
    uint32_t wi = pos >> 5;
    unsigned long long buf = (((unsigned long long)wordAt(wi + 1u) << 32) | wordAt(wi)) >> (pos & 31u);
    uint32_t avail = 64u - (pos & 31u);
    wi += 2u;
    bool live = ok, done = false;
    while (__any(live)) {
        const uint32_t w = (uint32_t)buf;
        const uint32_t c15 = __brev(w) >> 17;                 // the next fifteen bits, first bit of the stream on top
        // (the smallest length whose limit lies above the code, and that length's adj with it: fifteen independent compares, two
        // chains of selects -- no table read between the text and the symbol's place)
        uint32_t cl = 15;
        int32_t ad = adjOf[15];
#pragma unroll
        for (int l = 14; l >= 1; l--) {
            const bool below = c15 < limit[l];
            cl = below ? (uint32_t)l : cl;
            ad = below ? adjOf[l] : ad;
        }
        const bool coded = c15 < limit[15];                   // (a complete code: always)
        const uint32_t at = (uint32_t)(ad + (int32_t)(c15 >> (15u - cl)));
        const uint32_t sym = order[min(at, LH_ORDER - 1u) * 64u + lane];
        const bool found = live && coded && pos + cl <= endBit;
        const bool isVal = sym <= (uint32_t)CN_NULL, isE1 = sym == (uint32_t)CN_ESC1, isE2 = sym == (uint32_t)CN_ESC2;
        const bool isEot = sym == (uint32_t)CN_EOT;
        const uint32_t extra = isE1 ? 8u : isE2 ? 2u : 0u;
        const uint32_t raw = (w >> cl) & ((1u << extra) - 1u);       // (cl <= 15, extra <= 8: inside the 32 bits)
        // what the reference would trip over: a value beyond the reader's array, an escape before any value
        const bool bad = live && (!found || (isVal && k >= a.nInit) || ((isE1 || isE2) && k == 0u) || pos + cl + extra > endBit);
        const bool take = live && !bad;
        if (take && isVal) {
            prior = sym == (uint32_t)CN_NULL ? GF_NULL_CODE : sym - 128u;
            res[k] = (int32_t)prior;
            k++;
        }
        if (take && (isE1 || isE2)) {
            prior = (prior << extra) | raw;
            res[k - 1u] = (int32_t)prior;
        }
        const uint32_t used = take ? cl + extra : 0u;         // (symbol 260, the spare slot: skipped, :512)
        pos += used;
        buf >>= used;
        avail -= used;
        if (avail < 32u) {
            buf |= (unsigned long long)wordAt(wi) << avail;
            avail += 32u;
            wi++;
        }
        ok = ok && wi < LH_STAGE_WORDS;                       // (the window never looks beyond the staged words)
        done = done || (take && isEot);
        ok = ok && !bad;
        live = live && ok && !(take && isEot);
    }
    ok = ok && done && k == a.nInit;
    LH_STAMP(4);
    // ---- the second stream's code lengths, from the bit behind the end-of-text symbol ----
    uint8_t *sMetaLen = reinterpret_cast<uint8_t *>(order);                   // (the first stream's tables are dead)
    uint8_t *sOrder = sMetaLen + CN_META * 64, *sLut = sOrder + CN_META * 64;
    static_assert((size_t)CN_META * 64 * 2 + 128 * 64 <= (size_t)LH_ORDER * 64 * 2, "the meta tables lie over the symbol order");
    uint32_t pos2 = 0, nonZero = 0;
    const int32_t st2 = cd_lane_parse_lengths(peek, ok ? (int32_t)GF_K_OK : (int32_t)GF_K_ERR_UNSUPPORTED, pos, endBit, lane, sMetaLen, sOrder,
                                              sLut, reinterpret_cast<uint8_t *>(r2 + 8), inBatch && ok, &pos2, &nonZero);
    LH_STAMP(5);
#undef LH_STAMP
    if (!inBatch) return;
    const bool headDone = ok && st2 == GF_K_OK;
    r2[0] = (uint32_t)st2;
    r2[1] = pos2;
    r2[2] = nonZero;
    r2[3] = headDone ? 1u : 0u;
}

#ifndef GF_LSOP_UNPACK_WAVES
#define GF_LSOP_UNPACK_WAVES 8                                 // (waves per SIMD the register budget is cut for: experiment builds)
#endif
__global__ __launch_bounds__(DEC_THREADS, GF_LSOP_UNPACK_WAVES) void k_lsop_unpack2(GfLsopUnpackArgs a)
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
        if (tid == 0) a.coefs[t * 16 + GF_LSOP_FMT_WORD] = 0u;             // (until the tile's plane is written)
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

#ifdef GF_DIAG
        uint32_t *const stamps = a.debug ? a.debug + t * 16 : nullptr;
        if (stamps && tid == 0) stamps[0] = (uint32_t)__builtin_amdgcn_s_memtime();
#else
        constexpr uint32_t *stamps = nullptr;
#endif
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
        // (round 6) k_lsop_head has walked the first stream and the second one's code lengths already, where it could
        const uint32_t *pre2 = a.pre2 ? a.pre2 + t * GF_CANON_REC_WORDS : nullptr;
        const bool headDone = pre2 && GF_UNI(pre2[3]) != 0u;
        if (!headDone) pre2 = nullptr;
        for (int attempt = 0; attempt < 2 && !headDone; attempt++) {
            const uint32_t e = attempt == 0 ? endShort : endBit;
            pos = pos0;
            st = textInLds ? cd_decode_stream(S, TL, pos0, e, nInit, nInit, sink0, &pos, &nv, stamps, pre, bias, tok)
                           : cd_decode_stream(S, TG, pos0, e, nInit, nInit, sink0, &pos, &nv, stamps, pre, bias);
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
            // (round 6) THE BYTE PLANE.  A tile whose second stream holds bytes only (a code without escapes: cd_decode_stream decides)
            // and whose rows' initialisers -- columns 0 and 1, the two tail cells of every row from the third on -- are bytes as well
            // leaves as a plane in the reconstruction's pipeline order (gvrs_kernels.h): k_lsop_reconstruct_plane adds a byte of it to
            // something it has in registers for EVERY cell of those rows.  The initialisers wait in the last bytes of the token
            // table's words (the stage ends in front of them); the stream is staged a WINDOW of whole rows at a time -- one window
            // where it fits (terrain tiles of 120 x 150: 17,228 values), three for 256 x 256 --, and behind each window its rows go
            // to the plane.
            const uint32_t nSpec = 4u * (nR - 2u), specRoom = a.plane.ok ? (nSpec + 15u) & ~15u : 0u;      // (<= 4,096: gf_lsop_plane_geom)
            const uint32_t stageCap = (uint32_t)(4 * sizeof(S.qe)) + (capWords - usedWords) * 4u + 4096u - specRoom;
            uint8_t *const spec = reinterpret_cast<uint8_t *>(cdLdsText + capWords) + 4096u - specRoom;   // [4][nR - 2]: column 0, column 1, the 2 (nR - 2) tail cells
            const uint32_t wI = nC - 4u;
            uint32_t window = 0, specMine = 0;
            auto specAt = [&](uint32_t i) -> uint32_t {
                const uint32_t base0 = nC - 1u, base1 = 2u * (nC - 1u) + nR - 1u, tailBase = base1 + nR - 2u;
                const uint32_t q = i / (nR - 2u), k = i - q * (nR - 2u);
                return q == 0u ? base0 + 1u + k : q == 1u ? base1 + k : tailBase + (q - 2u) * (nR - 2u) + k;
            };
            if (a.plane.ok) {
                // (the first stream is done, by k_lsop_head or behind this workgroup's barriers.  The table itself is written behind
                // the second stream's synchronisation pass, whose token table lies there: a thread keeps its first initialiser)
                bool wide = false;
                for (uint32_t i = (uint32_t)tid; i < nSpec; i += DEC_THREADS) {
                    const uint32_t x = (uint32_t)res[specAt(i)];
                    wide = wide || x + 128u > 255u;
                    if (i == (uint32_t)tid) specMine = x;
                }
#ifndef GF_LSOP_WINDOW_DIV
#define GF_LSOP_WINDOW_DIV 1                                   // (experiment builds: several windows where one would do)
#endif
                if (!__syncthreads_or(wide ? 1 : 0)) window = min(nInt, stageCap / GF_LSOP_WINDOW_DIV / wI * wI);
            }
            CdCellSink sink1{reinterpret_cast<uint32_t *>(res + nInit), GfCellMap::make(4, 1u, 2u), nInt, true,
                             reinterpret_cast<uint8_t *>(S.qs), reinterpret_cast<uint8_t *>(cdLdsText + usedWords),
                             (uint32_t)(4 * sizeof(S.qe)), stageCap, 0u, false};
            sink1.window = window;
            sink1.windowed = &S.changed;
            static_assert(CD_NCUR == 1, "the stage over all four sync arrays needs one subsequence per thread");
            // the rows of a window to the plane (the whole workgroup, between two barriers): values lo .. lo + count - 1 of the stream =
            // rows 2 + lo / wI .. of the tile lie in the stage.  thread = (block of a turn, lane, quarter of its sixteen steps):
            // consecutive threads write consecutive words.  Lane l's column at step s is (s - 3 l) mod P, its row 2 + 32 ((s - 3 l)
            // div P) + l; a word may reach across the end of a period into the lane's next row, but it holds INTERIOR cells of one row
            // only (five steps lie between a row's last interior cell and the next row's first): that row's window writes it, a word
            // without any the window of its first row; a word of holes is written where one window is all (whole lines leave the L2).
            // (A thread a lane's sixteen bytes -- one 16-byte store, a quarter of the turns -- was slower: 24 K cycles per tile against 17 K.)
            auto dumpRows = [&](uint32_t lo, uint32_t count) {
#ifdef GF_DIAG
                const uint32_t tDump = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
                if (lo == 0u) {                                                           // (before the first word of the plane is stored)
                    for (uint32_t i = (uint32_t)tid; i < nSpec; i += DEC_THREADS) spec[i] = (uint8_t)(i == (uint32_t)tid ? specMine : (uint32_t)res[specAt(i)]);
                    __syncthreads();
                }
                constexpr uint32_t L = GF_LSOP_PLANE_LANES, BLOCK_WORDS = L * 4u, BLOCKS_PER_TURN = DEC_THREADS / BLOCK_WORDS;
                static_assert(DEC_THREADS % BLOCK_WORDS == 0 && 16u * BLOCKS_PER_TURN < GF_LSOP_PIPE_MIN_P, "a turn moves a lane by less than a period");
                const uint32_t rLo = 2u + lo / wI, rHi = rLo + count / wI;                 // the window's rows
                const bool all = count == nInt;
                uint32_t *plane = reinterpret_cast<uint32_t *>(res + a.plane.offWords);
                const uint32_t l = ((uint32_t)tid >> 2) & (L - 1u), kq = (uint32_t)tid & 3u;
                const int32_t PP = (int32_t)a.plane.P;
                // (one window of several: a lane's rows inside the window, i.e. the blocks from the period of its first such row to the
                // period behind its last one -- the words of holes between the rows are then not written at all)
                uint32_t ph = 0, bFrom = 0, bTo = a.plane.nBlocks;
                if (!all) {
                    const uint32_t first = rLo > 2u + l ? (rLo - 2u - l + L - 1u) / L : 0u;      // the lane's first period with a row >= rLo
                    const uint32_t beyond = rHi > 2u + l ? (rHi - 2u - l + L - 1u) / L : 0u;     // ... with a row >= rHi
                    ph = first > 0u ? first - 1u : 0u;              // (a word of the period before may reach into the row)
                    bFrom = (ph * a.plane.P + 3u * l) / 16u;
                    bTo = min(bTo, (beyond * a.plane.P + 3u * l + 15u) / 16u + 1u);
                    if (first >= beyond) bTo = 0;
                }
                bFrom = bFrom / BLOCKS_PER_TURN * BLOCKS_PER_TURN + (uint32_t)tid / BLOCK_WORDS;
                int32_t c = (int32_t)(16u * bFrom + 4u * kq) - (int32_t)(3u * l + ph * a.plane.P);
                if (c < 0 && ph > 0u) { c += PP; ph--; }
                // the byte of cell (r, cc): an interior residual from the stage, an initialiser from the table above
                auto cellByte = [&](uint32_t r, int32_t cc) -> uint32_t {
                    if (cc >= 2 && cc <= (int32_t)nC - 3) return *sink1.slot((r - 2u) * wI + (uint32_t)(cc - 2) - lo);
                    const uint32_t at = cc == 0 ? r - 2u : cc == 1 ? nR - 2u + r - 2u : 2u * (nR - 2u) + 2u * (r - 2u) + (uint32_t)(cc - ((int32_t)nC - 2));
                    return spec[at];
                };
                for (uint32_t b = bFrom; b < bTo; b += BLOCKS_PER_TURN) {
                    const uint32_t r = 2u + L * ph + l;
                    uint32_t w = 0;
                    bool mine = all;
                    if (r < nR && c >= 2 && c + 3 <= (int32_t)nC - 3) {
                        // four interior cells of one row: four neighbouring bytes of the stage (two words and a funnel shift where
                        // they lie in one of its two parts and not at its very end)
                        mine = mine || (r >= rLo && r < rHi);
                        if (mine) {
                            const uint32_t rel = (r - 2u) * wI + (uint32_t)(c - 2) - lo;
                            const bool inA = rel + 8u <= sink1.capA, inB = rel >= sink1.capA && rel + 8u <= sink1.cap;
                            if (inA || inB) {
                                const uint8_t *at = inA ? sink1.stA + rel : sink1.stB + (rel - sink1.capA);
                                const uint32_t sh = (uint32_t)(uintptr_t)at & 3u;
                                const uint32_t *p4 = reinterpret_cast<const uint32_t *>(at - sh);
                                w = __builtin_amdgcn_alignbyte(p4[1], p4[0], sh);
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; j++) w |= (uint32_t)*sink1.slot(rel + (uint32_t)j) << (8 * j);
                            }
                        }
                    } else {
                        const bool cur = r < nR && c + 3 >= 0 && c < (int32_t)nC;
                        const bool next = r + L < nR && c + 3 >= PP;
                        if (!all) {
                            const bool curInt = cur && c + 3 >= 2 && c <= (int32_t)nC - 3, nextInt = next && c + 3 >= PP + 2;
                            const uint32_t owner = curInt ? r : nextInt ? r + L : cur ? r : next ? r + L : 0u;
                            mine = owner >= rLo && owner < rHi;
                        }
                        if (mine && (cur || next)) {
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int32_t cj = c + j;
                                if (cur && cj >= 0 && cj < (int32_t)nC) w |= cellByte(r, cj) << (8 * j);
                                if (next && cj >= PP) w |= cellByte(r + L, cj - PP) << (8 * j);
                            }
                        }
                    }
                    if (mine) plane[(b * L + l) * 4u + kq] = w;
                    c += 16 * (int32_t)BLOCKS_PER_TURN;
                    if (c >= PP) { c -= PP; ph++; }
                }
#ifdef GF_DIAG
                if (stamps && tid == 0) stamps[15] = (lo == 0u ? 0u : stamps[15]) + ((uint32_t)__builtin_amdgcn_s_memtime() - tDump);   // wave 0's share
#endif
            };
#ifdef GF_DIAG
            if (stamps && tid == 0) stamps[8] = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
            st = textInLds ? cd_decode_stream(S, TL, pos, endBit, nInt, nInt, sink1, &pos, &nv, stamps ? stamps + 8 : nullptr, pre2, bias, tok, 0, dumpRows)
                           : cd_decode_stream(S, TG, pos, endBit, nInt, nInt, sink1, &pos, &nv, stamps ? stamps + 8 : nullptr, pre2, bias, nullptr, 0, dumpRows);
            // (cd_decode_stream ends behind a barrier)
            if (st == GF_K_OK && window != 0u && S.changed == 1u && tid == 0) a.coefs[t * 16 + GF_LSOP_FMT_WORD] = 1u;
        }
#ifdef GF_DIAG
        if (stamps && tid == 0) stamps[14] = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
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
                                  hipStream_t stream, const uint32_t *pre, uint32_t *debug, uint32_t *pre2, bool planes)
{
    if (nTiles == 0) return hipSuccess;
    if (pre && pre2) {
        // the serial parts first, a lane per tile (large batches: gf_prepass_tiles_per_wave)
        GfLsopHeadArgs h{blob, blobBytes, offsets, slotStride, lengths, pre, pre2, residuals, resStride, nTiles,
                         (uint32_t)(4 * nRows + 2 * nCols - 9), debug};
        static GfDynLdsOptIn optH;
        const hipError_t e = gf_opt_in_dyn_lds(k_lsop_head, LH_LDS_BYTES, optH);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_lsop_head, dim3((unsigned)((nTiles + 63) / 64)), dim3(64), LH_LDS_BYTES, stream, h);
    }
    GfLsopUnpackArgs a{blob, blobBytes, offsets, slotStride, lengths, residuals, resStride, coefs, status, nTiles, nRows, nCols,
                       ldsTextBytes, pre, pre2, debug, planes ? gf_lsop_plane_geom((uint32_t)nRows, (uint32_t)nCols) : GfLsopPlaneGeom{}};
    static GfDynLdsOptIn opt;
    {
        const hipError_t e = gf_opt_in_dyn_lds(k_lsop_unpack2, ldsTextBytes + (sizeof(uint16_t) << CD_LUT_BITS), opt);
        if (e != hipSuccess) return e;
    }
    (void)grid;
    hipLaunchKernelGGL(k_lsop_unpack2, gf_tile_grid(nTiles), dim3(DEC_THREADS), ldsTextBytes + (sizeof(uint16_t) << CD_LUT_BITS), stream, a);
    return hipGetLastError();
}
