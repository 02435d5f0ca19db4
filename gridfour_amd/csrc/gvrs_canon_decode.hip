// gvrs_canon_decode.hip -- CodecCanonHuffman.decode for a batch of tiles on gfx950: persistent
// 256-thread workgroups, one tile at a time per workgroup.
//
// Reference (core/src/main/java/org/gridfour/compress/canonicalHuffman/):
//   CodecCanonHuffman.decode :163-195      6-byte header, uniform shortcut, predictor dispatch
//   CanonicalHuffman.decode :441-466, decodeText :469-519
//   LengthEncoder.readEncodedLengths :197-236, CanonHuffTreeDecoder :68-177
//   the predictors' decodeInt (compress/PredictorModel*.java)
//
// Phases
//   0  wave 0 reads the code tables (20 raw meta lengths, then the 260 coded text lengths); all threads
//      build an 11-bit lookup table by canonical search (no scatter), longer codes fall back to the search
//   1  the text is cut into <= 512 subsequences, two per thread; each one starts 128 bits early so that
//      it is synchronised with the true symbol sequence at its boundary almost always (self-synchronising
//      prefix code); barrier-synchronised fix-up rounds repair the rest; value counts are prefix-summed
//   2  every subsequence is decoded again, now writing each value (its target symbol and the escapes that
//      follow it, CanonicalHuffman.java:489-511) to the cell its stream position belongs to
//   3  predictor inverse in place (gvrs_decode_common.h)

#include <hip/hip_runtime.h>

#include "gvrs_kernels.h"
#include "huff_build.h"

namespace {

// Compiled twice: as is (256 threads per workgroup, two subsequences per thread) and with -DGF_CD_THREADS=512 -DGF_CD_VARIANT
// (one per thread, 64 VGPRs, four workgroups = all 32 wave slots of a CU where the LDS footprint allows four); the variant
// object exports gf_launch_canon_decode_t512 and the LDS sizes of that build only.  decodeBatchDev weighs the two.
#ifndef GF_CD_THREADS
#define GF_CD_THREADS 256
#endif
#ifdef GF_CD_VARIANT
#define gf_launch_canon_decode gf_launch_canon_decode_t512
#define gf_canon_decode_lds_text gf_canon_decode_lds_text_t512
#define gf_canon_decode_lds_stage gf_canon_decode_lds_stage_t512
#define gf_canon_decode_lds_per_wg gf_canon_decode_lds_per_wg_t512
#endif
constexpr int DEC_THREADS = GF_CD_THREADS;
constexpr int DEC_WAVES = DEC_THREADS / 64;

#include "gvrs_decode_common.h"

#include "gvrs_canon_decode_common.h"

__global__ __launch_bounds__(DEC_THREADS, DEC_THREADS == 256 ? 4 : 8) void k_canon_decode(GfDecodeArgs a)
{
    __shared__ CanonDec S;

    const int tid = threadIdx.x;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const uint32_t *__restrict__ w32 = reinterpret_cast<const uint32_t *>(a.blob);
    const uint64_t nWords = (a.blobBytes + 3) >> 2;
    const uint32_t capWords = a.ldsTextBytes >> 2;
    // the byte stage of phase 2 (CdCellSink): the sync arrays qe / qc are dead by then, the rest sits behind the text copy
    // (one subsequence per thread: every thread has its start in a register before the first value is staged, so all four
    // sync arrays serve as stage; two per thread: the starts of the second half are still needed, qe / qc only)
    uint8_t *const stageA = reinterpret_cast<uint8_t *>(CD_NCUR == 1 ? S.qs : S.qe);
    const uint32_t stageCapA = (uint32_t)((CD_NCUR == 1 ? 4 : 2) * sizeof(S.qe));
    static_assert(offsetof(CanonDec, qe) == offsetof(CanonDec, qs) + sizeof(S.qs) && offsetof(CanonDec, qx) == offsetof(CanonDec, qc) + sizeof(S.qc),
                  "qs .. qx form one stretch of LDS");
    static_assert(offsetof(CanonDec, qc) == offsetof(CanonDec, qe) + sizeof(S.qe), "qe and qc form one stretch of LDS");

    // behind the canonical run of the fast legacy kernel (round 5; DEC_FAST_CANON in gvrs_decode.hip, a.retryFlag non-null): only the tiles
    // that run marked, and nothing at all when it marked none
    if (a.retryFlag && a.retryFlag[0] == 0u) return;

    GF_FOR_WG_TILE(t, a.nTiles) {                                         // no tile loop: see gvrs_kernels.h
        if (a.retryFlag && a.status[t] != (int32_t)GF_K_LEAN_RETRY) continue;
        const uint64_t off = a.offsets ? a.offsets[t] : (uint64_t)t * a.slotStride;
        const uint32_t len = a.lengths[t];
        uint32_t *o = reinterpret_cast<uint32_t *>(a.values) + t * (size_t)nCells;
        const uint8_t *__restrict__ pk = a.blob + off;

        if (len < 6 || off + len > a.blobBytes) {                 // packing[1..5] -> ArrayIndexOutOfBounds
            if (tid == 0) a.status[t] = GF_K_ERR_BOUNDS;
            __syncthreads();
            continue;
        }
        const int predictor = (int8_t)pk[1];
        const uint32_t seed = (uint32_t)pk[2] | ((uint32_t)pk[3] << 8) | ((uint32_t)pk[4] << 16) | ((uint32_t)pk[5] << 24);
        if (predictor == 0 && len == 6) {                         // uniform tile, CodecCanonHuffman.java:171-176
            for (uint32_t i = tid; i < nCells; i += DEC_THREADS) o[i] = seed;
            if (tid == 0) a.status[t] = GF_K_OK;
            __syncthreads();
            continue;
        }
        if (predictor < 1 || predictor > 4 || (predictor == 2 && nC < 2)) {   // :208-209 IOException; Linear output[1]
            if (tid == 0) a.status[t] = predictor < 1 || predictor > 4 ? GF_K_ERR_FORMAT : GF_K_ERR_BOUNDS;
            __syncthreads();
            continue;
        }
        const int model = predictor;

        // the packing as words: aligned words of the blob, bit positions carry the misalignment
        const uint64_t word0 = off >> 2;
        const uint32_t bias = (uint32_t)(off & 3u) * 8u;
        const uint32_t endBit = bias + len * 8u;
        const uint32_t needWords = (endBit + 31u) / 32u + 4u;       // the readers look up to three words ahead
        const bool textInLds = needWords <= capWords;
        if (textInLds) cd_stage_text(w32, word0, nWords, endBit, needWords);
        const CdTextLds TL{needWords};
        const CdTextGlobal TG{w32 + word0, (uint32_t)min((uint64_t)needWords, nWords - word0)};   // huge packing: read in place
        __syncthreads();

        // ---------------- phases 0-2: the canonical-Huffman stream, values to their cells ----------------
        const uint32_t nStream = gf_stream_len(model, nR, nC);
        // Triangle tiles of the one-subsequence-per-thread build: the staged residuals become the tile in one go (cd_fused_triangle)
        // (the stage behind the sync arrays: what this packing leaves of the text buffer, then the bytes behind it)
        uint8_t *const stageB = reinterpret_cast<uint8_t *>(cdLdsText + (textInLds ? needWords : 0u));
        // (less the last word of part B where there is one: cd_fused_triangle reads the stage two words at a time, and the
        // word behind the last byte it may ask for has to lie inside the workgroup's LDS as well)
        const uint32_t stageBytesB = (capWords - (textInLds ? needWords : 0u)) * 4u + a.ldsStageBytes;
        const uint32_t stageCap = stageCapA + (stageBytesB >= 4u ? stageBytesB - 4u : 0u);
#ifdef GF_DIAG
        const bool fuse = cd_fuse_eligible(model, nR, nC, stageCap) && !(a.phaseLimit & 0x300);
        const CdCellSink sink{o, GfCellMap::make(model, nR, nC), nStream, !(a.phaseLimit & 0x100),
                              stageA, stageB, stageCapA, stageCap, 0u, fuse};
        uint32_t *stamps = a.debug ? a.debug + t * 16 : nullptr;
        if (stamps && tid == 0) stamps[0] = (uint32_t)__builtin_amdgcn_s_memtime();
#else
        const bool fuse = cd_fuse_eligible(model, nR, nC, stageCap);
        const CdCellSink sink{o, GfCellMap::make(model, nR, nC), nStream, true,
                              stageA, stageB, stageCapA, stageCap, 0u, fuse};
        constexpr uint32_t *stamps = nullptr;
#endif
        uint32_t endPos, nValues;
        const uint32_t *pre = a.trees ? a.trees + t * GF_CANON_REC_WORDS : nullptr;
        // the token table of the synchronisation pass: in the value stage behind the text, which is idle until phase 2
        uint16_t *const tok = a.ldsStageBytes >= (sizeof(uint16_t) << CD_LUT_BITS) ? reinterpret_cast<uint16_t *>(cdLdsText + capWords) : nullptr;
#ifdef GF_DIAG
        const int diagLimit = a.phaseLimit & 0xff;
#else
        constexpr int diagLimit = 0;
#endif
        const int32_t st = textInLds ? cd_decode_stream(S, TL, bias + 48u, endBit, nCells, nStream, sink, &endPos, &nValues, stamps, pre, bias, tok, diagLimit)
                                     : cd_decode_stream(S, TG, bias + 48u, endBit, nCells, nStream, sink, &endPos, &nValues, stamps, pre, bias);
        if (st != GF_K_OK) {
            if (tid == 0) a.status[t] = st;
            __syncthreads();
            continue;
        }

        // ---------------- phase 3: predictor inverse ----------------
        if (fuse) cd_fused_triangle(S, sink, seed, nR, nC, o);
        else gf_predictor_inverse(model, seed, o, nR, nC, nullptr);
#ifdef GF_DIAG
        if (stamps && tid == 0) stamps[6] = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
        if (tid == 0) a.status[t] = GF_K_OK;
        __syncthreads();
    }
}

#ifndef GF_CD_VARIANT
// ---------------------------------------------------------------------------------------------------------------
// Code-length pre-pass of the canonical decoder, ONE LANE PER TILE: LengthEncoder.readEncodedLengths (:197-236) and
// CanonHuffTreeDecoder.decodeTree (:133-177) are a serial walk over some three hundred tokens; inside the decode kernel
// it kept one wave busy and three idle for a fifth of the tile time.  Same walk, same checks and statuses as phase 0 of
// cd_decode_stream; bit positions are relative to the packing (the stream starts at bit 48).

template <unsigned perWave>                                 // as k_huffman_parse_trees
__global__ __launch_bounds__(64) void k_canon_parse_lengths(const uint8_t *__restrict__ blob, size_t blobBytes,
                                                            const uint64_t *__restrict__ offsets, size_t slotStride,
                                                            const uint32_t *__restrict__ lengths, uint32_t *__restrict__ recs,
                                                            size_t nTiles, int lsopContainer, uint32_t *__restrict__ clearFlags)
{
    __shared__ uint8_t sMetaLen[CN_META * 64], sOrder[CN_META * 64], sLut[128 * 64];   // per-lane columns
    const uint32_t lane = threadIdx.x;
    // (the retry words of the decode kernels that follow, cleared here instead of by a launch of their own)
    if (clearFlags && blockIdx.x == 0 && lane < 2u) clearFlags[lane] = 0u;
    const size_t t0 = (size_t)blockIdx.x * perWave;          // perWave lanes walk a tile each: gf_prepass_tiles_per_wave
    // records start out zero: only non-zero lengths are stored below
    {
        const size_t n = min((size_t)perWave, nTiles - t0) * GF_CANON_REC_WORDS;
        uint32_t *r0 = recs + t0 * GF_CANON_REC_WORDS;
        for (size_t i = lane; i < n; i += 64) r0[i] = 0;
        __threadfence_block();
    }
    // (a wave per tile, round 5: every lane walks the one tile -- the walk is wave-uniform, i.e. scalar code, and reads the packing's
    // head out of the other lanes' registers, which is only defined while they are active --, lane 0 stores)
    const size_t t = perWave == 1u ? t0 : t0 + lane;
    const bool inBatch = t < nTiles;
    const bool writer = perWave == 64u || lane == 0u;
    const uint64_t off = !inBatch ? 0ull : offsets ? offsets[t] : (uint64_t)t * slotStride;
    const uint32_t len = inBatch ? lengths[t] : 0u;
    const bool readable = inBatch && len >= 7 && off + len <= blobBytes;
    uint32_t head0 = 0, head1 = 0;                          // perWave == 1: words lane and 64 + lane of the packing (zero beyond its end)
    if (perWave == 1u && readable && blobBytes >= 7) {
        const uint32_t vis = min(len, 512u);
        const uint8_t *__restrict__ pk0 = blob + off;
        const uint32_t a0 = 4u * lane < vis ? min(4u * lane, vis - 4u) : 0u, a1 = 4u * (64u + lane) < vis ? min(4u * (64u + lane), vis - 4u) : 0u;
        head0 = reinterpret_cast<const CdPackedWord *>(pk0 + a0)->v;
        head1 = reinterpret_cast<const CdPackedWord *>(pk0 + a1)->v;
        asm volatile("" : "+v"(head0), "+v"(head1));
        head0 = 4u * lane < vis ? head0 >> (8u * (4u * lane - a0)) : 0u;
        head1 = 4u * (64u + lane) < vis ? head1 >> (8u * (4u * (64u + lane) - a1)) : 0u;
    }
    // Sixty-four tiles per wave (round 4, as k_huffman_parse_trees): the head of every packing -- the code lengths are its first few
    // hundred bytes -- goes to LDS first, the wave fetching tile after tile, a lane a word (coalesced, 32 tiles' loads in flight at
    // once), and a lane's walk reads down its own column: a peek was a dependent load from global memory per token, inside a
    // branch the compiler waits in (0.143 ms per 12,960 tiles).  A walk that leaves the staged head reads the packing itself.
    constexpr uint32_t STAGE_WORDS = 128;                    // 512 bytes
    __shared__ uint32_t stage[perWave == 64u ? STAGE_WORDS * 64 : 1];
    if (perWave == 64u) {
        const uint32_t visMine = readable ? min(len, STAGE_WORDS * 4u) : 0u;
        // word k = bytes 4k .. 4k+3 of the packing, zero from its end on; never a byte beyond the packing is touched (the load is moved
        // back to end with the last byte and shifted into place; a word that is not visible is read at byte 0 and dropped; the
        // empty asm keeps the compiler from moving each load into a branch of its own)
        auto staged_at = [](uint32_t vis, uint32_t k) -> uint32_t { return 4u * k < vis ? min(4u * k, vis - 4u) : 0u; };   // vis >= 7
        auto staged_fix = [](uint32_t w, uint32_t vis, uint32_t k) -> uint32_t {
            return 4u * k < vis ? w >> (8u * (4u * k - min(4u * k, vis - 4u))) : 0u;
        };
        constexpr uint32_t TURN = 32;
        if (blobBytes >= 7) {
            for (uint32_t j0 = 0; j0 < 64u; j0 += TURN) {
                uint32_t w0[TURN], w1[TURN], visJ[TURN];
#pragma unroll
                for (uint32_t u = 0; u < TURN; u++) {
                    const uint32_t j = j0 + u;
                    const uint32_t vis = (uint32_t)__builtin_amdgcn_readlane((int)visMine, (int)j);
                    const uint64_t offJ = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), (int)j) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, (int)j);
                    const uint8_t *__restrict__ pkJ = blob + (vis ? offJ : 0ull);
                    visJ[u] = vis;
                    w0[u] = reinterpret_cast<const CdPackedWord *>(pkJ + staged_at(vis, lane))->v;
                    w1[u] = reinterpret_cast<const CdPackedWord *>(pkJ + staged_at(vis, 64u + lane))->v;
                }
#pragma unroll
                for (uint32_t u = 0; u < TURN; u += 8u)
                    asm volatile("" : "+v"(w0[u]), "+v"(w0[u + 1]), "+v"(w0[u + 2]), "+v"(w0[u + 3]), "+v"(w0[u + 4]), "+v"(w0[u + 5]),
                                      "+v"(w0[u + 6]), "+v"(w0[u + 7]), "+v"(w1[u]), "+v"(w1[u + 1]), "+v"(w1[u + 2]), "+v"(w1[u + 3]),
                                      "+v"(w1[u + 4]), "+v"(w1[u + 5]), "+v"(w1[u + 6]), "+v"(w1[u + 7]));
#pragma unroll
                for (uint32_t u = 0; u < TURN; u++) {
                    stage[lane * 64u + j0 + u] = staged_fix(w0[u], visJ[u], lane);
                    stage[(64u + lane) * 64u + j0 + u] = staged_fix(w1[u], visJ[u], 64u + lane);
                }
            }
        }
        __syncthreads();
    }
    if (!inBatch) return;
    uint32_t *rec = recs + t * GF_CANON_REC_WORDS;
    uint8_t *outLen = reinterpret_cast<uint8_t *>(rec + 8);
    if (!readable) {                                          // no stream: the decode kernel does not look here
        if (writer) rec[0] = (uint32_t)GF_K_ERR_BOUNDS;
        return;
    }
    const uint8_t *__restrict__ pk = blob + off;
    const uint32_t endBit = len * 8u;
    // where the stream starts: behind the 6-byte header of CodecCanonHuffman, or (lsopContainer) behind the header of an
    // LSOP12 container of the canonical type (LsHeader.java:131-185; the other types are not read through this record)
    uint32_t startBit = 48u;
    if (lsopContainer) {
        const uint32_t hdr = 55u + ((pk[1] & 0x80) ? 4u : 0u);
        if (!(pk[1] & 0x40) || (pk[1] & 0x0f) != 2 || pk[2] != 12 || len < hdr) {
            if (writer) rec[0] = (uint32_t)GF_K_ERR_UNSUPPORTED;
            return;
        }
        startBit = hdr * 8u;
    }
    auto peek = [&](uint32_t pos) -> uint32_t {               // 32 bits of the packing from bit pos, zero beyond its end
        if (perWave == 64u && (pos >> 5) + 1u < STAGE_WORDS) {
            const uint32_t wi = pos >> 5;
            return __builtin_amdgcn_alignbit(stage[(wi + 1u) * 64u + lane], stage[wi * 64u + lane], pos & 31u);
        }
        if (perWave == 1u && (pos >> 5) + 1u < 128u) {        // (pos is wave-uniform: the words out of the lanes that fetched them)
            const uint32_t wi = pos >> 5, wj = wi + 1u;
            const uint32_t lo0 = (uint32_t)__builtin_amdgcn_readlane((int)head0, (int)(wi & 63u)), lo1 = (uint32_t)__builtin_amdgcn_readlane((int)head1, (int)(wi & 63u));
            const uint32_t hi0 = (uint32_t)__builtin_amdgcn_readlane((int)head0, (int)(wj & 63u)), hi1 = (uint32_t)__builtin_amdgcn_readlane((int)head1, (int)(wj & 63u));
            return __builtin_amdgcn_alignbit(wj < 64u ? hi0 : hi1, wi < 64u ? lo0 : lo1, pos & 31u);
        }
        const uint32_t b = pos >> 3;
        uint64_t w;
        if (b + 8u <= len) {
            w = ((uint64_t)reinterpret_cast<const CdPackedWord *>(pk + b + 4)->v << 32) | reinterpret_cast<const CdPackedWord *>(pk + b)->v;
        } else {
            w = 0;
            for (uint32_t k = 0; k < 8; k++)
                if (b + k < len) w |= (uint64_t)pk[b + k] << (8u * k);
        }
        return (uint32_t)(w >> (pos & 7u));
    };
    uint32_t pos = 0, nonZero = 0;
    const int32_t st = cd_lane_parse_lengths(peek, GF_K_OK, startBit, endBit, lane, sMetaLen, sOrder, sLut, outLen, writer, &pos, &nonZero);
    if (!writer) return;
    rec[0] = (uint32_t)st;
    rec[1] = pos;
    rec[2] = nonZero;
}

#endif  // GF_CD_VARIANT

}  // namespace

// LDS per workgroup that still lets FIVE workgroups run on a CU.  Measured (k_lsop_unpack2, same tables): 32,240 bytes run
// four to a CU, 31,744 five -- LDS is handed out in steps coarser than 512 bytes (1,280 fits both findings).
static constexpr size_t CD_FIFTH = 31 * 1024, CD_QUARTER = 40 * 1024;

uint32_t gf_canon_decode_lds_text(int nRows, int nCols)
{
    // LDS copy of a packing: typical terrain packs to 0.5-1 byte per cell; larger packings are read in place.  Since the tile loop
    // went the kernel needs 75 VGPRs and LDS alone decides how many workgroups a CU holds: where the usual size lands just above
    // a fifth of the CU's LDS (ETOPO1-shaped tiles: 18.5 KB of tables + 14.5 KB of text) the copy is trimmed to what lets five
    // in, as long as that leaves half a byte per cell -- decode of the bench batch 2.29 -> 2.06 ms, although the value stage then
    // shrinks to the 4 KB of the dead sync arrays (a larger stage instead of text measured the same, 2.06 / 2.10 ms)
    size_t cells = (size_t)nRows * (size_t)nCols;
    size_t want = cells - cells / 4 + 1024;
    if (want > 96 * 1024) want = 96 * 1024;
    want = (want + 31) & ~(size_t)31;
    const size_t room = (CD_FIFTH - sizeof(CanonDec)) & ~(size_t)31;
    if (DEC_THREADS == 256 && want > room && room >= cells / 2 + 1024) want = room;   // (the 512-thread build: four workgroups of a quarter each)
    return (uint32_t)want;
}

uint32_t gf_canon_decode_lds_stage(int nRows, int nCols)
{
    // what is left of a fifth resp. a quarter of the CU's LDS (the tier the tables and the text copy fall into) and never more
    // than a half of the stream could use
    const size_t cells = (size_t)nRows * (size_t)nCols;
    const size_t base = sizeof(CanonDec) + gf_canon_decode_lds_text(nRows, nCols);
    const size_t budget = DEC_THREADS == 256 && base <= CD_FIFTH ? CD_FIFTH : CD_QUARTER - 512;
    size_t room = budget > base ? budget - base : 0;
    const size_t inArrays = (DEC_THREADS == 256 ? 2 : 4) * sizeof(((CanonDec *)nullptr)->qe);      // the part of the stage over the sync arrays
    const size_t want = cells > inArrays ? cells - inArrays : 0;
    if (room > want) room = want;
    return (uint32_t)(room & ~(size_t)31);
}

hipError_t gf_launch_canon_decode(const GfDecodeArgs &a, hipStream_t stream, unsigned grid)
{
    if (a.nTiles == 0) return hipSuccess;
    const size_t dyn = (size_t)a.ldsTextBytes + a.ldsStageBytes;
    static GfDynLdsOptIn opt;
    {
        const hipError_t e = gf_opt_in_dyn_lds(k_canon_decode, dyn, opt);
        if (e != hipSuccess) return e;
    }
    (void)grid;
    hipLaunchKernelGGL(k_canon_decode, gf_tile_grid(a.nTiles), dim3(DEC_THREADS), dyn, stream, a);
    return hipGetLastError();
}

// static + dynamic LDS of one workgroup of this build for the launch a describes
size_t gf_canon_decode_lds_per_wg(const GfDecodeArgs &a) { return sizeof(CanonDec) + (size_t)a.ldsTextBytes + a.ldsStageBytes; }

#ifndef GF_CD_VARIANT
hipError_t gf_launch_canon_parse_lengths(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, size_t slotStride,
                                         const uint32_t *lengths, uint32_t *recs, size_t nTiles, int lsopContainer,
                                         hipStream_t stream, uint32_t *clearFlags)
{
    if (nTiles == 0) return hipSuccess;
#ifndef GF_CANON_PREPASS_ONE_LANE_MAX
#define GF_CANON_PREPASS_ONE_LANE_MAX 3000       // where the code lengths' walk changes from a wave to a lane per tile (round 6, decode ms of
                                                // CodecCanonHuffman batches of 120 x 150 with a wave / a lane per tile: 1,280 tiles 0.152 / 0.167, 3,000
                                                // 0.241 / 0.237, 4,096 0.317 / 0.291; 4,096 tiles of 256 x 256: 1.101 / 1.066 -- the tree walk's own
                                                // threshold, GF_PREPASS_ONE_LANE_MAX, is 4,096)
#endif
    if (nTiles <= GF_CANON_PREPASS_ONE_LANE_MAX)
        hipLaunchKernelGGL(k_canon_parse_lengths<1>, dim3((unsigned)nTiles), dim3(64), 0, stream, blob, blobBytes, offsets, slotStride,
                           lengths, recs, nTiles, lsopContainer, clearFlags);
    else
        hipLaunchKernelGGL(k_canon_parse_lengths<64>, dim3((unsigned)((nTiles + 63) / 64)), dim3(64), 0, stream, blob, blobBytes, offsets,
                           slotStride, lengths, recs, nTiles, lsopContainer, clearFlags);
    return hipGetLastError();
}
#endif  // GF_CD_VARIANT
