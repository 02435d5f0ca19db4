// gvrs_canon_encode.hip -- CodecCanonHuffman.encode (the default integer codec of current Gridfour) for a
// batch of tiles on gfx950: one 256-thread workgroup per tile.
//
// Reference (core/src/main/java/org/gridfour/compress/canonicalHuffman/):
//   CodecCanonHuffman.encode :70-143, compress :145-160  -- null / uniform shortcuts, predictors in the
//       order Differencing, Linear, Triangle (DifferencingWithNulls alone when a null is present), the
//       strictly shortest packing wins; 6-byte header codecIndex, predictor, seed LE
//   CanonicalHuffman.encode :177-283, countSymbols :352-418, buildCodeLengthTree :285-343
//   TreeBuilder, PackageMerge, LengthEncoder, HuffmanCodeBits (see gvrs_canon_common.h)
//   the predictors' encodeInt (compress/PredictorModel*.java): same residual streams as their M32 forms
//
// Phases (same shape as gvrs_encode.hip):
//   A  one flat pass over the tile, 8 cells per thread: residuals of the three predictors, their symbol
//      classes, three 260-bin LDS histograms (4 replicas)
//   B  one wave per predictor: cn_build() -> code table, serialised code tables, exact size
//   C  the shortest candidate is packed: header + tables image, then the text -- per value its target
//      symbol and, for values outside a byte, escape symbols with raw bits -- bit positions by a
//      workgroup prefix sum, ORed into an LDS window that is flushed with coalesced stores

#include <hip/hip_runtime.h>

#include "gvrs_kernels.h"
#include "gvrs_encode_layout.h"

namespace {

#include "gvrs_encode_common.h"
#include "gvrs_canon_common.h"

// per-tile record between k_canon_encode (phases A, B, selection) and k_canon_pack (phase C): model, seed, image bits, longest
// code, largest escape kind, 3 spare, then the serialised code tables and the winner's code table (GfEncodeArgs::packRecs)
constexpr int CN_PACK_REC_WORDS = 8 + CN_IMG_WORDS + CN_HIST;
constexpr int CN_STAT_WORDS = 16 + 3 * CN_HIST;      // k_canon_encode (part 1) -> k_canon_trees: models, seed, quirk counts, escape kinds, the three histograms

struct CanonPersist {
    uint32_t hist[3][CN_HIST];
    uint32_t tab[3][CN_HIST];
    uint32_t img[3][CN_IMG_WORDS];
    unsigned long long totalBits[3];
    uint32_t imgBits[3];
    uint32_t maxLen[3];
    uint32_t maxKind[3];                        // largest escape kind seen (bounds the bits of one value)
    uint32_t nGap[3];
    int32_t model[3];
    uint32_t seed;
    uint32_t flags;                             // bit0 any null, bit1 any valid, bit2 some cell differs from cell 0
    uint32_t waveSum[ENC_WAVES];
    unsigned long long sumStart;
    uint32_t nStart;
    uint32_t pmLock;                            // cn_pm_acquire
    uint32_t lbBytes[3];                        // lower bound of a candidate's packing from its histogram alone (phase B)
};

struct CanonTrees {
    CanonScratch tree[3];
    CanonPM pm;                                 // shared: see CanonPM
};
union CanonUnion {
    uint32_t histR[3][CN_HIST * HIST_R];        // phase A
    CanonTrees b;                               // phase B
};

// upper bound of the bits one value can take given the longest code and the largest escape kind
__device__ __forceinline__ uint32_t cn_elem_max_bits(uint32_t maxLen, uint32_t maxKind)
{
    if (maxKind == 0u || maxKind == 7u) return maxLen;
    if (maxKind <= 3u) return maxLen + maxKind * (maxLen + 2u);
    return maxLen + 3u * (maxLen + 8u);
}

// stream elements [sBegin, sEnd) of `model`: the general packer (border segments, huge codes)
__device__ void cpack_generic(int model, const uint32_t *__restrict__ tile, uint32_t nR, uint32_t nC, uint32_t seed,
                              const uint32_t *tab, uint32_t elemMaxBits, uint32_t sBegin, uint32_t sEnd, uint32_t *win,
                              uint32_t *__restrict__ out32, uint32_t *waveSum, PackState &ps)
{
    const uint32_t tid = threadIdx.x;
    uint32_t E = 4;
    while (E > 1 && (uint64_t)ENC_THREADS * E * elemMaxBits > (uint64_t)(WIN_WORDS - 2) * 32u) E >>= 1;
    uint32_t active = ENC_THREADS;
    if ((uint64_t)ENC_THREADS * elemMaxBits > (uint64_t)(WIN_WORDS - 2) * 32u)
        active = max(1u, (uint32_t)(((uint64_t)(WIN_WORDS - 2) * 32u) / elemMaxBits));
    const uint32_t chunkElems = active * E;
    for (uint32_t chunk = sBegin; chunk < sEnd; chunk += chunkElems) {
        uint32_t xs[4];
        uint32_t myBits = 0;
        const uint32_t s0 = chunk + tid * E;
        const uint32_t cEnd = min(sEnd, chunk + chunkElems);
#pragma unroll
        for (uint32_t e = 0; e < 4; e++) {
            xs[e] = 0;
            const uint32_t s = s0 + e;
            if (e < E && tid < active && s < cEnd) {
                const uint32_t idx = gf_stream_cell(model, nR, nC, s);
                const uint32_t r = idx / nC, c = idx - r * nC;
                xs[e] = cell_residual(model, tile, nC, idx, r, c, seed);
                myBits += cn_value_bits(tab, xs[e]);
            }
        }
        uint32_t total;
        const uint32_t excl = block_excl_scan(myBits, waveSum, &total);
        if (myBits) {
            BitSink sink;
            sink.init(win, ps.bitBase + excl - ps.wordBase * 32u);
#pragma unroll
            for (uint32_t e = 0; e < 4; e++) {
                const uint32_t s = s0 + e;
                if (e < E && tid < active && s < cEnd) cn_value_emit(sink, tab, xs[e]);
            }
            sink.finish();
        }
        __syncthreads();
        ps.bitBase += total;
        window_flush(win, out32, ps);
    }
}

// the short head of a Linear / Triangle stream in the fast case, one element per thread and chunk, inlined (pack_head of
// gvrs_encode.hip: a call into cpack_generic costs every tile the callee's register saves in scratch memory)
template <int MODEL>
__device__ __forceinline__ void cpack_head(const uint32_t *__restrict__ tile, uint32_t nR, uint32_t nC, uint32_t seed,
                                           const uint32_t *tab, uint32_t sEnd, uint32_t *win, uint32_t *__restrict__ out32,
                                           uint32_t *waveSum, PackState &ps)
{
    const uint32_t tid = threadIdx.x;
    for (uint32_t chunk = 0; chunk < sEnd; chunk += ENC_THREADS) {
        const uint32_t s = chunk + tid;
        uint32_t x = 0, myBits = 0;
        if (s < sEnd) {
            const uint32_t idx = gf_stream_cell(MODEL, nR, nC, s);
            const uint32_t r = idx / nC, c = idx - r * nC;
            x = cell_residual(MODEL, tile, nC, idx, r, c, seed);
            myBits = cn_value_bits(tab, x);
        }
        uint32_t total;
        const uint32_t excl = block_excl_scan(myBits, waveSum, &total);
        if (myBits) {
            BitSink sink;
            sink.init(win, ps.bitBase + excl - ps.wordBase * 32u);
            cn_value_emit(sink, tab, x);
            sink.finish();
        }
        __syncthreads();
        ps.bitBase += total;
        window_flush(win, out32, ps);
    }
}

// the main segment of a model's stream = flat scan over the cells with an emit mask
template <int MODEL>
__device__ void cpack_flat(const uint32_t *__restrict__ tile, uint32_t nC, uint32_t nCells, uint32_t seed,
                           const uint32_t *tab, uint32_t *win, uint32_t *__restrict__ out32, uint32_t *waveSum,
                           PackState &ps, uint32_t cellBegin = 0, uint32_t cellEnd = 0xFFFFFFFFu)
{
    // cells [cellBegin, cellEnd): cellBegin a multiple of CPT, cellEnd a multiple of CPT or the end of the tile
    const uint32_t tid = threadIdx.x;
    cellEnd = min(cellEnd, nCells);
    uint32_t c0 = (cellBegin + tid * CPT) % nC;
    const uint32_t cStep = STEP_CELLS % nC;
    for (uint32_t base = cellBegin; base < cellEnd; base += STEP_CELLS) {
        const uint32_t i0 = base + tid * CPT;
        uint32_t cl[CPT], xs[CPT];
        uint32_t myBits = 0, wide = 0;
        if (i0 < cellEnd) {
            Cells8 Q;
            load_cells8_wave(tile, nC, nCells, i0, Q);
            uint32_t c = c0;
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                bool emit;
                const uint32_t x = flat_residual<MODEL>(Q, j, i0 + j, c, nC, nCells, seed, &emit);
                const bool narrow = x + 128u < 256u;
                const uint32_t e = emit ? (narrow ? tab[x + 128u] : 0u) : 0u;
                cl[j] = e;
                xs[j] = x;
                myBits += e >> 16;
                if (emit && !narrow) wide |= 1u << j;
                if (++c == nC) c = 0;
            }
            if (wide) {                                               // values outside a byte (rare on terrain)
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if ((wide >> j) & 1u) myBits += cn_value_bits(tab, xs[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) { cl[j] = 0; xs[j] = 0; }
        }
        c0 += cStep;
        if (c0 >= nC) c0 -= nC;

        uint32_t total;
        const uint32_t excl = block_excl_scan(myBits, waveSum, &total);
        if (myBits) {
            BitSink sink;
            sink.init(win, ps.bitBase + excl - ps.wordBase * 32u);
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                if ((wide >> j) & 1u) cn_value_emit(sink, tab, xs[j]);
                else sink.put32(cl[j] & 0xffffu, cl[j] >> 16);
            }
            sink.finish();
        }
        __syncthreads();
        ps.bitBase += total;
        window_flush(win, out32, ps);
    }
}

// cpack_flat over cells [cellBegin, cellEnd) with wave-private bit windows (gvrs_encode_common.h: wave_windows_*); false =
// a wave's share did not fit, nothing was written
template <int MODEL>
__device__ bool cpack_flat_waves(const uint32_t *__restrict__ tile, uint32_t nC, uint32_t nCells, uint32_t seed,
                                 const uint32_t *tab, uint32_t *win, uint32_t *__restrict__ out32, uint32_t *waveSum,
                                 PackState &ps, uint32_t slotWords, uint32_t cellBegin, uint32_t cellEnd)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = gf_wave_id();
    cellEnd = min(cellEnd, nCells);
    const uint32_t carryWord = wave_windows_begin(win, waveSum);
    const uint32_t quarter = (((cellEnd - cellBegin + ENC_WAVES - 1) / ENC_WAVES) + CPT - 1) / CPT * CPT;
    const uint32_t segBegin = min(cellEnd, cellBegin + wave * quarter), segEnd = min(cellEnd, segBegin + quarter);
    uint32_t *wwin = win + wave * WAVE_WIN;
    uint32_t bits = 0;
    bool fits = true;
    uint32_t c0 = (segBegin + lane * CPT) % nC;
    const uint32_t cStep = (64u * CPT) % nC;
    for (uint32_t base = segBegin; base < segEnd; base += 64u * CPT) {
        const uint32_t i0 = base + lane * CPT;
        uint32_t cl[CPT], xs[CPT];
        uint32_t myBits = 0, wide = 0;
        if (i0 < segEnd) {
            Cells8 Q;
            load_cells8_wave(tile, nC, nCells, i0, Q);
            uint32_t c = c0;
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                bool emit;
                const uint32_t x = flat_residual<MODEL>(Q, j, i0 + j, c, nC, nCells, seed, &emit);
                const bool narrow = x + 128u < 256u;
                const uint32_t e = emit ? (narrow ? tab[x + 128u] : 0u) : 0u;
                cl[j] = e;
                xs[j] = x;
                myBits += e >> 16;
                if (emit && !narrow) wide |= 1u << j;
                if (++c == nC) c = 0;
            }
            if (wide) {                                               // values outside a byte (rare on terrain)
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if ((wide >> j) & 1u) myBits += cn_value_bits(tab, xs[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) { cl[j] = 0; xs[j] = 0; }
        }
        c0 += cStep;
        if (c0 >= nC) c0 -= nC;

        const uint32_t incl = gf_wave_incl_scan(myBits);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (bits + total > WAVE_WIN_BITS) { fits = false; break; }   // wave-uniform
        // the usual lane (eight values inside a byte, two groups of four codes of at most 32 bits each) joins its codes in
        // registers (GF_JOIN8_OR, gvrs_encode_common.h); the others go through the bit sink
#define GF_LN(j) (cl[j] >> 16)
#define GF_CD(j) (cl[j] & 0xffffu)
        const uint32_t n01 = GF_LN(0) + GF_LN(1), n45 = GF_LN(4) + GF_LN(5);
        const uint32_t n0 = n01 + GF_LN(2) + GF_LN(3), n1 = n45 + GF_LN(6) + GF_LN(7);
        static_assert(CPT == 8, "the register path joins eight codes");
        if (wide == 0u && n0 <= 32u && n1 <= 32u) {
            GF_JOIN8_OR(wwin, bits + incl - myBits, GF_CD, GF_LN, n01, n45, n0);
#undef GF_LN
#undef GF_CD
        } else if (myBits) {
            BitSink sink;
            sink.init(wwin, bits + incl - myBits);
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                if ((wide >> j) & 1u) cn_value_emit(sink, tab, xs[j]);
                else sink.put32(cl[j] & 0xffffu, cl[j] >> 16);
            }
            sink.finish();
        }
        bits += total;
    }
    return wave_windows_end(win, waveSum, carryWord, bits, fits, out32, slotWords, ps);
}

// the flat scan of a tile in as many cell ranges as its bit count asks for (see pack_flat_ranges in gvrs_encode.hip)
template <int MODEL>
__device__ __forceinline__ void cpack_flat_ranges(const uint32_t *__restrict__ tile, uint32_t nC, uint32_t nCells, uint32_t seed,
                                                  const uint32_t *tab, uint32_t *win, uint32_t *__restrict__ out32, uint32_t *waveSum,
                                                  PackState &ps, uint32_t slotWords, uint32_t textBits)
{
    const uint64_t want = ((uint64_t)textBits + (textBits >> 2)) / ENC_WAVES;           // a wave's share, with 25 % slack
    const uint32_t nRanges = (uint32_t)min((uint64_t)1024, want / WAVE_WIN_BITS + 1u);
    const uint32_t per = (((nCells + nRanges - 1) / nRanges) + (CPT * ENC_WAVES) - 1) / (CPT * ENC_WAVES) * (CPT * ENC_WAVES);
    for (uint32_t b = 0; b < nCells; b += per) {
        if (!cpack_flat_waves<MODEL>(tile, nC, nCells, seed, tab, win, out32, waveSum, ps, slotWords, b, b + per))
            cpack_flat<MODEL>(tile, nC, nCells, seed, tab, win, out32, waveSum, ps, b, b + per);
    }
}

// (round 6) cpack_flat_waves for a stream WITHOUT wide values (the winner's largest escape kind is 0: every residual a byte) from the
// tile's byte plane of raw row differences (GfEncodeArgs::plane; see pack_plane_waves in gvrs_encode.hip): a residual's byte, bit 7
// flipped, is its symbol (CanonicalHuffman.java:223-231: value + 128); the head of a Linear or Triangle stream rides in front of
// wave 0's share of the first range; the plane's words are asked for a turn ahead.  A cell the flat scan does not emit takes the
// table's last, empty entry.
constexpr uint32_t CN_NO_SYMBOL = CN_HIST - 1;                    // (an index behind the 261 symbols: the packer clears the entry)
static_assert(CN_HIST - 1 > CN_SYMS, "the canonical table needs a spare entry behind its symbols");

__device__ __forceinline__ void cemit8_codes(uint32_t *wwin, uint32_t pos, const uint32_t (&cl)[CPT], uint32_t myBits)
{
#define GF_LN(j) (cl[j] >> 16)
#define GF_CD(j) (cl[j] & 0xffffu)
    const uint32_t n01 = GF_LN(0) + GF_LN(1), n45 = GF_LN(4) + GF_LN(5);
    const uint32_t n0 = n01 + GF_LN(2) + GF_LN(3), n1 = n45 + GF_LN(6) + GF_LN(7);
    if (n0 <= 32u && n1 <= 32u) {
        GF_JOIN8_OR(wwin, pos, GF_CD, GF_LN, n01, n45, n0);
#undef GF_LN
#undef GF_CD
    } else if (myBits) {
        BitSink sink;
        sink.init(wwin, pos);
#pragma unroll
        for (int j = 0; j < CPT; j++) sink.put32(cl[j] & 0xffffu, cl[j] >> 16);
        sink.finish();
    }
}

template <int MODEL>
__device__ bool cpack_plane_waves(const uint8_t *__restrict__ plane, uint32_t nC, uint32_t nCells, const uint32_t *tab, uint32_t *win,
                                  uint32_t *__restrict__ out32, uint32_t *waveSum, PackState &ps, uint32_t slotWords, uint32_t cellBegin,
                                  uint32_t cellEnd, uint32_t headElems)
{
    static_assert(MODEL >= 1 && MODEL <= 3, "the byte plane serves the three predictors");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = gf_wave_id();
    cellEnd = min(cellEnd, nCells);
    const uint32_t quarter = (((cellEnd - cellBegin + ENC_WAVES - 1) / ENC_WAVES) + CPT - 1) / CPT * CPT;
    const uint32_t segBegin = min(cellEnd, cellBegin + wave * quarter), segEnd = min(cellEnd, segBegin + quarter);
    const uint32_t lastI0 = (nCells - 1u) & ~(uint32_t)(CPT - 1);
    PlaneWords ahead = plane_load<MODEL>(plane, nC, min(segBegin + lane * CPT, lastI0));
    const bool withHead = MODEL != 1 && headElems != 0u && wave == 0u;     // (wave-uniform)
    uint32_t hb[2] = {0u, 0u};
    if constexpr (MODEL != 1) {
        if (withHead) plane_head_bytes<MODEL>(plane, nC, lane * CPT, headElems, hb);
    }
    const uint32_t carryWord = wave_windows_begin(win, waveSum);
    uint32_t *wwin = win + wave * WAVE_WIN;
    uint32_t bits = 0;
    bool fits = true;
    if constexpr (MODEL != 1) {
        if (withHead) {
            for (uint32_t h = 0; h < headElems; h += 64u * CPT) {
                if (h) plane_head_bytes<MODEL>(plane, nC, h + lane * CPT, headElems, hb);
                uint32_t cl[CPT], myBits = 0;
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    const uint32_t b = (hb[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    const uint32_t e = tab[h + lane * CPT + (uint32_t)j < headElems ? (b ^ 0x80u) : CN_NO_SYMBOL];
                    cl[j] = e;
                    myBits += e >> 16;
                }
                const uint32_t incl = gf_wave_incl_scan(myBits);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (bits + total > WAVE_WIN_BITS) { fits = false; break; }
                cemit8_codes(wwin, bits + incl - myBits, cl, myBits);
                bits += total;
            }
        }
    }
    uint32_t c0 = (segBegin + lane * CPT) % nC;
    const uint32_t cStep = (64u * CPT) % nC;
    for (uint32_t base = segBegin; base < segEnd && fits; base += 64u * CPT) {
        const uint32_t i0 = base + lane * CPT;
        uint32_t cl[CPT], myBits = 0;
        const PlaneWords now = ahead;
        ahead = plane_load<MODEL>(plane, nC, min(i0 + 64u * CPT, lastI0));
        if (i0 < segEnd) {
            uint32_t rb[2];
            plane_residual_bytes<MODEL>(now, nC, i0, rb);
            // the cells of the eight that the flat scan emits (nC >= 8: at most one of them starts a row), as in pack_plane_waves
            uint32_t em = 0xffu;
            const uint32_t left = nCells - i0;
            if (left < (uint32_t)CPT) em = (1u << left) - 1u;
            const uint32_t kz = c0 == 0u ? 0u : min(nC - c0, 16u);
            if constexpr (MODEL == 1) {
                if (i0 == 0u) em &= ~1u;
            } else if constexpr (MODEL == 2) {
                em &= ~(3u << kz);
                if (c0 == 1u) em &= ~1u;
            } else {
                em &= ~(1u << kz);
                if (i0 < nC) em &= ~((1u << min((uint32_t)CPT, nC - i0)) - 1u);
            }
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const uint32_t b = (rb[j >> 2] >> (8 * (j & 3))) & 0xffu;
                const uint32_t e = tab[(em >> j) & 1u ? (b ^ 0x80u) : CN_NO_SYMBOL];
                cl[j] = e;
                myBits += e >> 16;
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) cl[j] = 0;
        }
        c0 += cStep;
        if (c0 >= nC) c0 -= nC;
        const uint32_t incl = gf_wave_incl_scan(myBits);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (bits + total > WAVE_WIN_BITS) { fits = false; break; }   // wave-uniform
        cemit8_codes(wwin, bits + incl - myBits, cl, myBits);
        bits += total;
    }
    return wave_windows_end(win, waveSum, carryWord, bits, fits, out32, slotWords, ps);
}

// ... in as many cell ranges as the text's bit count asks for; false: a range did not fit (nothing of it was written, ps stands behind
// the ranges before it, *resume = its first cell: the caller goes on from there through the tile)
template <int MODEL>
__device__ __forceinline__ bool cpack_plane_ranges(const uint8_t *__restrict__ plane, uint32_t nC, uint32_t nCells, const uint32_t *tab,
                                                   uint32_t *win, uint32_t *__restrict__ out32, uint32_t *waveSum, PackState &ps,
                                                   uint32_t slotWords, uint32_t textBits, uint32_t headElems, uint32_t *resume)
{
    const uint64_t want = ((uint64_t)textBits + (textBits >> 2)) / ENC_WAVES;           // a wave's share, with 25 % slack
    const uint32_t nRanges = (uint32_t)min((uint64_t)1024, want / WAVE_WIN_BITS + 1u);
    constexpr uint32_t UNIT = CPT * ENC_WAVES;
    uint32_t per = (((nCells + nRanges - 1) / nRanges) + UNIT - 1) / UNIT * UNIT;
    for (uint32_t b = 0; b < nCells;) {
        if (cpack_plane_waves<MODEL>(plane, nC, nCells, tab, win, out32, waveSum, ps, slotWords, b, b + per, b == 0u ? headElems : 0u)) {
            b += per;
        } else {
            if (per <= 8u * UNIT) { *resume = b; return false; }
            per = (per / 2u + UNIT - 1u) / UNIT * UNIT;
        }
    }
    return true;
}

// workgroups per CU: the histogram + table kernel runs six (25.8 KB of LDS since the package-merge scratch is shared; with the
// two-turn phase B of round 3 the bound has to say so: left at four the compiler took 110 VGPRs, 1.03 against 1.23 ms),
// the pack kernel runs six (80 VGPRs; eight, at 64 VGPRs, cost 0.3 ms with the wave-private windows)
#ifndef GF_CN_AB_WGS
#define GF_CN_AB_WGS 6
#endif
#ifndef GF_CN_PACK_WGS
#define GF_CN_PACK_WGS 7        // sweep 6 / 7 / 8 after the head packer was inlined: encode 1.277 / 1.259 / 1.273 ms
#endif
constexpr int CN_AB_WGS = GF_CN_AB_WGS, CN_PACK_WGS = GF_CN_PACK_WGS;

// PART 0: phases A and B and the selection in one kernel.  PART 1 (round 4, as the legacy encoder's): phase A alone, the histograms
// to GfEncodeArgs::encStats; k_canon_trees builds the code tables with a wave per tile.  In the one-kernel form cn_build -- sort,
// tree rounds, depths, run-length tokens, the meta tree, the image: a long chain of dependent steps -- ran on one wave per candidate
// while the workgroup's other waves held their slots idle.
struct CanonPersistA {
    uint32_t maxKind[3];
    uint32_t nGap[3];
    int32_t model[3];
    uint32_t seed;
    uint32_t flags;
    unsigned long long sumStart;
    uint32_t nStart;
    uint32_t pmLock;
};
union CanonUnionA {
    uint32_t histR[3][CN_HIST * HIST_R];
};
#ifndef GF_CN_A_WGS
#define GF_CN_A_WGS 7
#endif

// PLANE (PART 1 only, round 6): the byte plane of raw row differences is written for k_canon_pack (GfEncodeArgs::plane, as
// k_huffman_encode<true, 1, true> of the legacy codec: the Differencing residual of every cell as one byte; bit 3 of the flags: some
// row difference is no byte -- the plane does not hold the tile)
template <int PART = 0, bool PLANE = false>
__global__ __launch_bounds__(ENC_THREADS, PART == 1 ? GF_CN_A_WGS : CN_AB_WGS) void k_canon_encode(GfEncodeArgs a)
{
    static_assert(!PLANE || PART == 1, "the byte plane belongs to the part-1 kernel");
    __shared__ std::conditional_t<PART == 1, CanonPersistA, CanonPersist> P;
    __shared__ std::conditional_t<PART == 1, CanonUnionA, CanonUnion> S;

    const int tid = threadIdx.x, lane = tid & 63, wave = (int)gf_wave_id();
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;

    GF_FOR_WG_TILE(t, a.nTiles) {                                         // no tile loop: see gvrs_kernels.h
        const uint32_t *__restrict__ tile = reinterpret_cast<const uint32_t *>(a.values) + t * (size_t)nCells;
        uint32_t *__restrict__ out32 = reinterpret_cast<uint32_t *>(a.out + t * a.slotStride);

        // ---------------- phase A: null / uniform scan + three histograms ----------------
        for (int i = tid; i < 3 * CN_HIST * HIST_R; i += ENC_THREADS) (&S.histR[0][0])[i] = 0;
        if (tid == 0) { P.flags = 0; P.sumStart = 0; P.nStart = 0; P.pmLock = 0; }
        if (tid < 3) { P.maxKind[tid] = 0; P.model[tid] = 0; P.nGap[tid] = 0; }
        __syncthreads();

        const bool triOk = nR >= 2 && nC >= 2;
        const uint32_t rep = (uint32_t)lane & (HIST_R - 1);
        const uint32_t v0 = tile[0];
        uint32_t myFlags = 0, mk1 = 0, mk2 = 0, mk3 = 0, gap1 = 0, gap2 = 0, gap3 = 0;
        // all symbols of residual x into histogram h; returns the escape kind, counts the quirk values
        auto addHist = [&](uint32_t *h, uint32_t x, uint32_t &gaps) -> uint32_t {
            if (x + 128u < 256u) {
                atomicAdd(h + (x + 128u) * HIST_R, 1u);
                return 0u;
            }
            uint32_t kind;
            const uint32_t target = cn_classify_count(x, &kind);
            atomicAdd(h + target * HIST_R, 1u);
            if (kind >= 1u && kind <= 3u) atomicAdd(h + CN_ESC2 * HIST_R, kind);
            else if (kind >= 4u && kind <= 6u) atomicAdd(h + CN_ESC1 * HIST_R, kind - 3u);
            if (cn_is_gap(x)) { gaps++; kind = 6u; }
            return kind == 7u ? 0u : kind;
        };
        {
            uint32_t *const h0 = &S.histR[0][rep], *const h1 = &S.histR[1][rep], *const h2 = &S.histR[2][rep];
            uint32_t c0 = ((uint32_t)tid * CPT) % nC;
            const uint32_t cStep = STEP_CELLS % nC;
            for (uint32_t i0 = (uint32_t)tid * CPT; i0 < nCells; i0 += STEP_CELLS) {
                Cells8 Q;
                load_cells8_wave(tile, nC, nCells, i0, Q);
                uint32_t c = c0;
                // (round 6, as the legacy encoder's phase A since round 4) the residuals of the thread's eight cells first; where every
                // one of the wave's is a byte -- every turn of a terrain tile -- the turn is 24 additions to the histograms and
                // nothing else: no "is it a byte?" branch per residual, no classification
                uint32_t D1[CPT], D2[CPT], D3[CPT];
                uint32_t wide = 0;
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    const uint32_t idx = i0 + j;
                    const uint32_t v = Q.cur[j];
                    const bool real = idx < nCells;
                    myFlags |= real ? (((v == GF_NULL_CODE) ? 1u : 2u) | (v != v0 ? 4u : 0u)) : 0u;
                    const uint32_t W = j > 0 ? Q.cur[j - 1] : Q.wm1;
                    const uint32_t WW = j > 1 ? Q.cur[j - 2] : (j == 1 ? Q.wm1 : Q.wm2);
                    const uint32_t N = Q.up[j];
                    const uint32_t NW = j > 0 ? Q.up[j - 1] : Q.upm1;
                    // the seed cell and the padding behind the tile count as residual 0; bin 128 is corrected below
                    const bool counted = real && idx > 0;
                    const uint32_t d = v - (c > 0 ? W : N);
                    D1[j] = counted ? d : 0u;
                    D2[j] = counted ? (c >= 2 ? v - (2u * W - WW) : d) : 0u;
                    D3[j] = (counted && triOk) ? ((idx >= nC && c > 0) ? v - (W + N - NW) : d) : 0u;
                    wide = max(wide, max(D1[j] + 128u, max(D2[j] + 128u, D3[j] + 128u)));
                    if (++c == nC) c = 0;
                }
                if (__all(wide <= 255u)) {
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        atomicAdd(h0 + (D1[j] + 128u) * HIST_R, 1u);
                        atomicAdd(h1 + (D2[j] + 128u) * HIST_R, 1u);
                        if (triOk) atomicAdd(h2 + (D3[j] + 128u) * HIST_R, 1u);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        mk1 = max(mk1, addHist(h0, D1[j], gap1));
                        mk2 = max(mk2, addHist(h1, D2[j], gap2));
                        if (triOk) mk3 = max(mk3, addHist(h2, D3[j], gap3));
                        if (PLANE && D1[j] + 128u > 255u) myFlags |= 8u;
                    }
                }
                [[maybe_unused]] uint32_t (&rowDiff)[CPT] = D1;
                if constexpr (PLANE) {
                    const uint32_t p01 = __builtin_amdgcn_perm(rowDiff[1], rowDiff[0], 0x0c0c0400u), p23 = __builtin_amdgcn_perm(rowDiff[3], rowDiff[2], 0x0c0c0400u);
                    const uint32_t p45 = __builtin_amdgcn_perm(rowDiff[5], rowDiff[4], 0x0c0c0400u), p67 = __builtin_amdgcn_perm(rowDiff[7], rowDiff[6], 0x0c0c0400u);
                    GfU2 w;
                    w.x = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                    w.y = __builtin_amdgcn_perm(p67, p45, 0x05040100u);
                    *reinterpret_cast<GfU2 *>(a.plane + t * a.planeStride + i0) = w;
                }
                c0 += cStep;
                if (c0 >= nC) c0 -= nC;
            }
        }
        if (myFlags) atomicOr(&P.flags, myFlags);
        if (mk1) atomicMax(&P.maxKind[0], mk1);
        if (mk2) atomicMax(&P.maxKind[1], mk2);
        if (mk3) atomicMax(&P.maxKind[2], mk3);
        if (gap1) atomicAdd(&P.nGap[0], gap1);
        if (gap2) atomicAdd(&P.nGap[1], gap2);
        if (gap3) atomicAdd(&P.nGap[2], gap3);
        __syncthreads();
        const uint32_t flags = P.flags;
        const bool anyNull = flags & 1u, anyValid = flags & 2u, varied = flags & 4u;
        uint32_t forcedZeros = ((nCells + CPT - 1) / CPT) * CPT - nCells + 1u;

        int32_t early = 99;
        if (!anyValid) early = GF_K_DECLINED;                                  // CodecCanonHuffman.java:85-87 -> null
        else if (!varied) early = GF_K_OK;                                     // uniform shortcut :89-110
        else if (4ull * nCells + 8ull >= (1ull << 22)) early = GF_K_ERR_UNSUPPORTED;   // 22-bit counts in the tree keys
        else if (!anyNull && nC < 2 && (a.predictorMask & 2)) early = GF_K_ERR_BOUNDS;  // Linear: values[1] AIOOBE
        else if (!anyNull && !triOk && (a.predictorMask & 4)) early = GF_K_ERR_ARG;     // Triangle -1 -> encode throws :183
        if (early != 99) {
            if (tid == 0) {
                uint32_t n = 0;
                if (early == GF_K_OK) {
                    if (a.slotStride >= 8) {
                        out32[0] = ((uint32_t)a.codecIndex & 0xffu) | (v0 << 16);
                        out32[1] = v0 >> 16;
                        n = 6;
                    } else {
                        early = GF_K_OVERFLOW;
                        n = 6;
                    }
                }
                a.lengths[t] = n;
                a.status[t] = early;
                if (a.predictors) a.predictors[t] = 0;
                if (PART == 1) (a.encStats + t * (size_t)CN_STAT_WORDS)[7] = 0u;          // nothing for k_canon_trees
            }
            __syncthreads();
            continue;
        }

        if (anyNull) {
            // ---- nulls path: seed (PredictorModelDifferencingWithNulls.java:181-208), then one histogram ----
            forcedZeros = 0;
            for (int i = tid; i < 3 * CN_HIST * HIST_R; i += ENC_THREADS) (&S.histR[0][0])[i] = 0;
            long long mySum = 0;
            uint32_t myCnt = 0;
            for (uint32_t idx = tid; idx < nCells; idx += ENC_THREADS) {
                const uint32_t v = tile[idx];
                if (v == GF_NULL_CODE) continue;
                const uint32_t r = idx / nC, c = idx - r * nC;
                bool flag;
                if (c > 0) flag = tile[idx - 1] == GF_NULL_CODE;
                else flag = r == 0 ? true : tile[idx - nC] == GF_NULL_CODE;
                if (flag) { mySum += (int32_t)v; myCnt++; }
            }
            if (myCnt) {
                atomicAdd(&P.sumStart, (unsigned long long)mySum);
                atomicAdd(&P.nStart, myCnt);
            }
            if (tid < 3) { P.maxKind[tid] = 0; P.nGap[tid] = 0; }
            __syncthreads();
            if (tid == 0) {
                const double avg = (double)(long long)P.sumStart / (double)P.nStart;
                double f = floor(avg + 0.5);
                int32_t s;
                if (f >= 2147483647.0) s = 2147483647;
                else if (f <= -2147483648.0) s = (int32_t)0x80000000;
                else s = (int32_t)f;
                P.seed = (uint32_t)s;
                P.model[0] = (a.predictorMask & 8) ? 4 : 0;
                P.model[1] = 0;
                P.model[2] = 0;
            }
            __syncthreads();
            const uint32_t seed = P.seed;
            uint32_t mk = 0, gaps = 0;
            for (uint32_t idx = tid; idx < nCells; idx += ENC_THREADS) {
                const uint32_t r = idx / nC, c = idx - r * nC;
                const uint32_t x = cell_residual(4, tile, nC, idx, r, c, seed);
                mk = max(mk, addHist(&S.histR[0][rep], x, gaps));
            }
            if (mk) atomicMax(&P.maxKind[0], mk);
            if (gaps) atomicAdd(&P.nGap[0], gaps);
            __syncthreads();
        } else if (tid == 0) {
            P.seed = v0;
            P.model[0] = (a.predictorMask & 1) ? 1 : 0;
            P.model[1] = (a.predictorMask & 2) ? 2 : 0;
            P.model[2] = ((a.predictorMask & 4) && triOk) ? 3 : 0;
        }

        // reduce the replicas; the end-of-text symbol counts 1 (CanonicalHuffman.java:355)
        for (int i = tid; i < 3 * CN_HIST; i += ENC_THREADS) {
            const int p = i / CN_HIST, s = i - p * CN_HIST;
            const uint32_t *h = &S.histR[p][0] + (size_t)s * HIST_R;
            uint32_t sum = 0;
#pragma unroll
            for (int k = 0; k < HIST_R; k++) sum += h[k];
            if (s == 128 && sum >= forcedZeros) sum -= forcedZeros;
            if (s == CN_EOT) sum = 1;
            if (s >= CN_SYMS) sum = 0;
            if constexpr (PART == 1) (a.encStats + t * (size_t)CN_STAT_WORDS + 16)[i] = sum;
            else P.hist[p][s] = sum;
        }
        if constexpr (PART == 1) {
            __syncthreads();                     // the models, the seed, the quirk counts and the escape kinds are in place
            if (tid < 3) {
                uint32_t *stat = a.encStats + t * (size_t)CN_STAT_WORDS;
                stat[tid] = (uint32_t)P.model[tid];
                stat[4 + tid] = P.nGap[tid];
                stat[8 + tid] = P.maxKind[tid];
                if (tid == 0) {
                    stat[3] = P.seed;
                    stat[7] = 1u;
                    stat[11] = (PLANE && !(flags & 9u)) ? 1u : 0u;       // the byte plane holds this tile (no null cell, every row difference a byte)
                }
            }
            __syncthreads();
            continue;
        } else {
        for (int i = tid; i < 3 * CN_IMG_WORDS; i += ENC_THREADS) (&P.img[0][0])[i] = 0;
        __syncthreads();                         // histR dead from here: S.tree may be written

        // ---------------- phase B: code tables of the candidates, one wave each ----------------
        // Only the shortest packing is written (CodecCanonHuffman.java:126-140), so code tables are built only for a predictor that
        // can still win -- the rule of the legacy encoder (gvrs_encode.hip, phase B), round 3.  No prefix code, length-limited or
        // not, spends fewer bits on a text than its zero-order entropy; the raw bits of the escapes are known from the counts:
        // 48 + N H + 2 n(ESC2) + 8 n(ESC1) bits (the serialised tables counted as nothing) is a LOWER BOUND of a candidate's
        // packing from its histogram alone.  The candidate with the smallest bound builds first; the others only if their bound
        // does not already lose against its exact size (the earlier predictor wins ties, :133).  A tile with a value inside the
        // -8333608 quirk (nGap: its correction can shorten the text) builds all three as before.
        const bool prune = P.nGap[0] == 0u && P.nGap[1] == 0u && P.nGap[2] == 0u;
        int firstP = -1;
        if (prune) {
            if (wave < 3 && P.model[wave] != 0) {
                const int p = wave;
                double sumCLogC = 0.0;
                uint32_t N = 0;
                for (int e = lane; e < CN_SYMS; e += 64) {
                    const uint32_t cnt = P.hist[p][e];
                    N += cnt;
                    if (cnt > 1) sumCLogC += (double)cnt * (double)__log2f((float)cnt);
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    N += gf_lane_xor(N, d);
                    sumCLogC += __shfl_xor(sumCLogC, d, 64);
                }
                if (lane == 0) {
                    // (less a margin for the single-precision logarithms: 2^-22 relative on sums of at most N log2 N, and the cast)
                    double text = (double)N * (double)__log2f((float)N) - sumCLogC;
                    text -= 64.0 + (double)N * (1.0 / 4096.0);
                    const double bits = 48.0 + (text > 0.0 ? text : 0.0) + 2.0 * (double)P.hist[p][CN_ESC2] + 8.0 * (double)P.hist[p][CN_ESC1];
                    P.lbBytes[p] = (uint32_t)(bits * 0.125);                  // floor: a lower bound stays one
                }
            }
            __syncthreads();
            for (int p = 0; p < 3; p++)
                if (P.model[p] != 0 && (firstP < 0 || P.lbBytes[p] < P.lbBytes[firstP])) firstP = p;
        }
#pragma unroll 1
        for (int stage = 0; stage < 2; stage++) {
            bool mine = wave < 3 && P.model[wave] != 0;
            if (prune) {
                mine = mine && (stage == 0 ? wave == firstP : wave != firstP);
                if (mine && stage == 1) {
                    const uint64_t firstBytes = (P.totalBits[firstP] + 7) >> 3;
                    const bool lost = wave < firstP ? (uint64_t)P.lbBytes[wave] > firstBytes : (uint64_t)P.lbBytes[wave] >= firstBytes;
                    if (lost) {
                        if (lane == 0) P.model[wave] = 0;                     // not a candidate any more
                        mine = false;
                    }
                }
            } else if (stage == 1) mine = false;                              // (all of them were built in the first turn)
            if (mine) {
                const int p = wave;
                const CanonBuilt B = cn_build(S.b.tree[p], S.b.pm, &P.pmLock, P.hist[p], P.nGap[p], P.tab[p], P.img[p], lane);
                if (lane == 0) {
                    P.imgBits[p] = B.imgBits;
                    P.maxLen[p] = B.maxLen;
                    P.totalBits[p] = 48ull + B.imgBits + B.textBits;
                }
            }
            __syncthreads();
        }

        // ---------------- phase C: pick the shortest, pack it ----------------
        int best = -1;
        uint64_t bestBytes = ~0ull;
        for (int p = 0; p < 3; p++) {
            if (P.model[p] == 0) continue;
            const uint64_t bytes = (P.totalBits[p] + 7) >> 3;
            if (bytes < bestBytes) { bestBytes = bytes; best = p; }          // strict: CodecCanonHuffman.java:133
        }
        if (best < 0) {
            if (tid == 0) {
                a.lengths[t] = 0;
                a.status[t] = GF_K_DECLINED;
                if (a.predictors) a.predictors[t] = 0;
            }
            __syncthreads();
            continue;
        }
        const int model = P.model[best];
        if (tid == 0) {
            a.lengths[t] = (uint32_t)min(bestBytes, (uint64_t)0xffffffffu);
            a.status[t] = bestBytes > a.slotStride ? GF_K_OVERFLOW : GF_K_OK;
            if (a.predictors) a.predictors[t] = (uint8_t)model;
        }
        if (bestBytes > a.slotStride) { __syncthreads(); continue; }

        {
            // the pack phase runs as k_canon_pack with its own register budget and occupancy: leave it the selection
            uint32_t *rec = a.packRecs + t * (size_t)CN_PACK_REC_WORDS;
            if (tid == 0) {
                rec[0] = (uint32_t)model;
                rec[1] = P.seed;
                rec[2] = P.imgBits[best];
                rec[3] = P.maxLen[best];
                rec[4] = P.maxKind[best];
                rec[5] = (uint32_t)min(P.totalBits[best] - 48ull - P.imgBits[best], (unsigned long long)0xFFFFFFFFu);   // text bits
                rec[6] = 0u;                                                                                             // (no byte plane in the one-kernel form)
            }
            for (int i = tid; i < CN_IMG_WORDS; i += ENC_THREADS) rec[8 + i] = P.img[best][i];
            for (int i = tid; i < CN_HIST; i += ENC_THREADS) rec[8 + CN_IMG_WORDS + i] = P.tab[best][i];
            __syncthreads();
        }
        }                                        // PART != 1
    }
}

// ------------------------------------------------------------------------------------------------
// k_canon_trees: phase B and the selection of k_canon_encode with ONE WAVE PER TILE (round 4), four tiles to a workgroup, which
// share the package-merge scratch under its lock as the candidates' waves of the one-kernel form do.  Same steps in the same order:
// lower bounds and pruning where no value lies in the -8333608 quirk, cn_build per candidate (its histogram read from the statistics
// record), strictly shortest wins, the earlier predictor wins a tie; the best candidate so far lies in the tile's record.
// ------------------------------------------------------------------------------------------------
#ifndef GF_CN_TREES_WAVES
#define GF_CN_TREES_WAVES 4      // tiles (= waves) per workgroup: 28.6 KB of LDS and five workgroups per CU at four
#endif
constexpr int CN_TW = GF_CN_TREES_WAVES;
struct CanonTreesShared {
    CanonScratch tree[CN_TW];
    uint32_t tab[CN_TW][CN_HIST];
    uint32_t img[CN_TW][CN_IMG_WORDS];
    CanonPM pm;
    uint32_t pmLock;
};

__global__ __launch_bounds__(64 * CN_TW, CN_TW == 4 ? 5 : CN_TW == 2 ? 9 : 3) void k_canon_trees(GfEncodeArgs a)
{
    __shared__ CanonTreesShared S;
    const int tid = threadIdx.x, lane = tid & 63, w = (int)gf_wave_id();
    if (tid == 0) S.pmLock = 0;
    __syncthreads();
    const size_t t = ((size_t)blockIdx.x + (size_t)blockIdx.y * gridDim.x) * (size_t)CN_TW + (size_t)w;
    if (t >= a.nTiles) return;
    const uint32_t *__restrict__ stat = a.encStats + t * (size_t)CN_STAT_WORDS;
    if (stat[7] == 0u) return;                                           // declined, uniform or refused by k_canon_encode
    int model[3];
    uint32_t nGap[3], maxKind[3];
#pragma unroll
    for (int p = 0; p < 3; p++) { model[p] = (int)stat[p]; nGap[p] = stat[4 + p]; maxKind[p] = stat[8 + p]; }
    const uint32_t seed = stat[3];
    const bool prune = nGap[0] == 0u && nGap[1] == 0u && nGap[2] == 0u;
    uint32_t lbBytes[3] = {0, 0, 0};
    int firstP = -1;
    if (prune) {
#pragma unroll
        for (int p = 0; p < 3; p++) {
            if (model[p] == 0) continue;
            const uint32_t *hist = stat + 16 + p * CN_HIST;
            double sumCLogC = 0.0;
            uint32_t N = 0;
            for (int e = lane; e < CN_SYMS; e += 64) {
                const uint32_t cnt = hist[e];
                N += cnt;
                if (cnt > 1) sumCLogC += (double)cnt * (double)__log2f((float)cnt);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                N += gf_lane_xor(N, d);
                sumCLogC += __shfl_xor(sumCLogC, d, 64);
            }
            double text = (double)N * (double)__log2f((float)N) - sumCLogC;
            text -= 64.0 + (double)N * (1.0 / 4096.0);
            const double bits = 48.0 + (text > 0.0 ? text : 0.0) + 2.0 * (double)hist[CN_ESC2] + 8.0 * (double)hist[CN_ESC1];
            lbBytes[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits * 0.125));
            if (firstP < 0 || lbBytes[p] < lbBytes[firstP]) firstP = p;
        }
    }
    uint32_t *rec = a.packRecs + t * (size_t)CN_PACK_REC_WORDS;
    int best = -1;
    uint64_t bestBytes = ~0ull, firstBytes = 0;
    uint32_t bImgBits = 0, bMaxLen = 0;
    unsigned long long bTotalBits = 0;
    for (int k = 0; k < 3; k++) {
        // with pruning: the candidate of the smallest bound first, then the others in order; without: all three in order
        int p = k;
        if (prune) {
            if (firstP < 0) break;
            p = firstP;
            if (k > 0) {
                p = k - 1;
                if (p >= firstP) p++;
            }
        }
        if (model[p] == 0) continue;
        if (prune && k > 0) {
            const bool lost = p < firstP ? (uint64_t)lbBytes[p] > firstBytes : (uint64_t)lbBytes[p] >= firstBytes;
            if (lost) continue;
        }
        for (int i = lane; i < CN_IMG_WORDS; i += 64) S.img[w][i] = 0;
        __builtin_amdgcn_wave_barrier();
        const CanonBuilt B = cn_build(S.tree[w], S.pm, &S.pmLock, stat + 16 + p * CN_HIST, nGap[p], S.tab[w], S.img[w], lane);
        const unsigned long long totalBits = 48ull + B.imgBits + B.textBits;
        const uint64_t bytes = (totalBits + 7) >> 3;
        if (prune && k == 0) firstBytes = bytes;
        __builtin_amdgcn_wave_barrier();
        if (best < 0 || bytes < bestBytes || (bytes == bestBytes && p < best)) {
            best = p;
            bestBytes = bytes;
            bImgBits = B.imgBits;
            bMaxLen = B.maxLen;
            bTotalBits = totalBits;
            for (int i = lane; i < CN_IMG_WORDS; i += 64) rec[8 + i] = S.img[w][i];
            for (int i = lane; i < CN_HIST; i += 64) rec[8 + CN_IMG_WORDS + i] = S.tab[w][i];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (best < 0) {
        if (lane == 0) {
            a.lengths[t] = 0;
            a.status[t] = GF_K_DECLINED;
            if (a.predictors) a.predictors[t] = 0;
        }
        return;
    }
    if (lane == 0) {
        a.lengths[t] = (uint32_t)min(bestBytes, (uint64_t)0xffffffffu);
        a.status[t] = bestBytes > a.slotStride ? GF_K_OVERFLOW : GF_K_OK;
        if (a.predictors) a.predictors[t] = (uint8_t)model[best];
        rec[0] = (uint32_t)model[best];
        rec[1] = seed;
        rec[2] = bImgBits;
        rec[3] = bMaxLen;
        rec[4] = maxKind[best];
        rec[5] = (uint32_t)min(bTotalBits - 48ull - bImgBits, (unsigned long long)0xFFFFFFFFu);   // text bits
        rec[6] = stat[11];                                                                       // the byte plane holds the tile
    }
}

// k_canon_pack: phase C of the encoder as its own kernel (see k_huffman_pack in gvrs_encode.hip for the measurements)
struct CanonPackShared {
    uint32_t tab[CN_HIST];
    uint32_t img[CN_IMG_WORDS];
    uint32_t waveSum[ENC_WAVES];
};

__global__ __launch_bounds__(ENC_THREADS, CN_PACK_WGS) void k_canon_pack(GfEncodeArgs a)
{
    __shared__ CanonPackShared P;
    __shared__ __attribute__((aligned(16))) uint32_t win[WIN_WORDS + WIN_SLACK];

    const int tid = threadIdx.x;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;

    GF_FOR_WG_TILE(t, a.nTiles) {                                         // no tile loop: see gvrs_kernels.h
        // (round 6) the tile's status and length and everything its record holds are asked for at once (as k_huffman_pack: one behind
        // the other they were three round trips to memory before the first bit was packed)
        const uint32_t *rec = a.packRecs + t * (size_t)CN_PACK_REC_WORDS;
        constexpr int IMG_PER = (CN_IMG_WORDS + ENC_THREADS - 1) / ENC_THREADS, TAB_PER = (CN_HIST + ENC_THREADS - 1) / ENC_THREADS;
        const int32_t tileStatus = a.status[t];
        const uint32_t tileLen = a.lengths[t];
        uint32_t rv[8], imgv[IMG_PER], tabv[TAB_PER];
#pragma unroll
        for (int k = 0; k < 8; k++) rv[k] = rec[k];
#pragma unroll
        for (int k = 0; k < IMG_PER; k++) imgv[k] = rec[8 + min(tid + k * ENC_THREADS, CN_IMG_WORDS - 1)];
#pragma unroll
        for (int k = 0; k < TAB_PER; k++) tabv[k] = rec[8 + CN_IMG_WORDS + min(tid + k * ENC_THREADS, CN_HIST - 1)];
        // declined, overflow: nothing to write; a uniform tile (exactly the 6 header bytes) was written by k_canon_encode
        if (tileStatus != GF_K_OK || tileLen <= 6u) continue;
        const uint32_t *__restrict__ tile = reinterpret_cast<const uint32_t *>(a.values) + t * (size_t)nCells;
        uint32_t *__restrict__ out32 = reinterpret_cast<uint32_t *>(a.out + t * a.slotStride);
        const int model = (int)rv[0];
        const uint32_t seed = rv[1], imgBits = rv[2], maxLen = rv[3], maxKind = rv[4], textBits = rv[5];
        // a stream without wide values of a tile whose byte plane k_canon_encode<1, true> has left: the plane instead of the tile
        const bool fromPlane = a.plane && rv[6] != 0u && maxKind == 0u && model >= 1 && model <= 3 && nC >= (uint32_t)CPT;
#pragma unroll
        for (int k = 0; k < IMG_PER; k++)
            if (tid + k * ENC_THREADS < CN_IMG_WORDS) P.img[tid + k * ENC_THREADS] = imgv[k];
#pragma unroll
        for (int k = 0; k < TAB_PER; k++) {
            const int i = tid + k * ENC_THREADS;
            if (i < CN_HIST) P.tab[i] = (fromPlane && i == (int)CN_NO_SYMBOL) ? 0u : tabv[k];
        }
        __syncthreads();

        // header (6 bytes = 48 bits) + code tables image shifted behind it
        const uint32_t headEnd = 48u + imgBits;
        const uint32_t headWords = (headEnd + 31u) >> 5;
        for (int i = tid; i < WIN_WORDS + WIN_SLACK; i += ENC_THREADS) {
            uint32_t w = 0;
            if (i < (int)headWords) {
                // image bit b lives at packing bit 48 + b: word i takes image bits [32 i - 48, 32 i - 16)
                const uint32_t *img = P.img;
                if (i == 0) w = ((uint32_t)a.codecIndex & 0xffu) | ((uint32_t)model << 8) | (seed << 16);
                else if (i == 1) w = (seed >> 16) | (img[0] << 16);
                else {
                    const uint32_t lo = img[i - 2], hi = i - 1 < CN_IMG_WORDS ? img[i - 1] : 0u;
                    w = (lo >> 16) | (hi << 16);
                }
            }
            win[i] = w;
        }
        __syncthreads();

        const uint32_t nStream = gf_stream_len(model, nR, nC);
        const uint32_t *tab = P.tab;
        const uint32_t elemMaxBits = max(1u, cn_elem_max_bits(maxLen, maxKind));
        PackState ps;
        ps.bitBase = headEnd;
        ps.wordBase = 0;
        window_flush(win, out32, ps);
        if (nStream > 0) {
            const bool fast = (uint64_t)STEP_CELLS * elemMaxBits <= (uint64_t)(WIN_WORDS - 2) * 32u && nC >= 2;
            uint32_t resume = 0;
            bool done = false;
            if (fast && fromPlane) {
                const uint8_t *__restrict__ plane = a.plane + t * a.planeStride;
                const uint32_t slotWords = (uint32_t)(a.slotStride >> 2);
                if (model == 1) done = cpack_plane_ranges<1>(plane, nC, nCells, tab, win, out32, P.waveSum, ps, slotWords, textBits, 0u, &resume);
                else if (model == 2) done = cpack_plane_ranges<2>(plane, nC, nCells, tab, win, out32, P.waveSum, ps, slotWords, textBits, 2u * nR - 1u, &resume);
                else done = cpack_plane_ranges<3>(plane, nC, nCells, tab, win, out32, P.waveSum, ps, slotWords, textBits, nC - 1u + nR - 1u, &resume);
                if (!done && resume != 0u) {
                    // (a range that overran its windows after all: the rest of the cells from the tile, the slow way)
                    if (model == 1) cpack_flat<1>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, resume, nCells);
                    else if (model == 2) cpack_flat<2>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, resume, nCells);
                    else cpack_flat<3>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, resume, nCells);
                    done = true;
                }
            }
            if (done) {
                // (packed from the plane)
            } else if (!fast) {
                cpack_generic(model, tile, nR, nC, seed, tab, elemMaxBits, 0u, nStream, win, out32, P.waveSum, ps);
            } else if (model == 1) {
                cpack_flat_ranges<1>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, (uint32_t)(a.slotStride >> 2), textBits);
            } else if (model == 2) {
                cpack_head<2>(tile, nR, nC, seed, tab, 2u * nR - 1u, win, out32, P.waveSum, ps);
                cpack_flat_ranges<2>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, (uint32_t)(a.slotStride >> 2), textBits);
            } else if (model == 3) {
                cpack_head<3>(tile, nR, nC, seed, tab, nC - 1u + nR - 1u, win, out32, P.waveSum, ps);
                cpack_flat_ranges<3>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, (uint32_t)(a.slotStride >> 2), textBits);
            } else {
                cpack_flat_ranges<4>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, (uint32_t)(a.slotStride >> 2), textBits);
            }
        }
        // end-of-text symbol (CanonicalHuffman.java:278), then the tail of the window
        if (tid == 0) {
            const uint32_t e = tab[CN_EOT];
            BitSink sink;
            sink.init(win, ps.bitBase - ps.wordBase * 32u);
            sink.put32(e & 0xffffu, e >> 16);
            sink.finish();
        }
        ps.bitBase += tab[CN_EOT] >> 16;
        __syncthreads();
        {
            const uint32_t remBits = ps.bitBase - ps.wordBase * 32u;
            const uint32_t remWords = (remBits + 31u) >> 5;
            const uint32_t slotWords = (uint32_t)(a.slotStride >> 2);
            for (uint32_t j = tid; j < remWords; j += ENC_THREADS)
                if (ps.wordBase + j < slotWords) out32[ps.wordBase + j] = win[j];
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t gf_launch_canon_encode(const GfEncodeArgs &a, hipStream_t stream)
{
    if (a.nTiles == 0) return hipSuccess;
    const dim3 grid = gf_tile_grid(a.nTiles);
    if (!a.packRecs) return hipErrorInvalidValue;
    if (a.encStats) {
        // the histograms, then the code tables with a wave per tile (four tiles to a workgroup)
        if (a.plane) hipLaunchKernelGGL((k_canon_encode<1, true>), grid, dim3(ENC_THREADS), 0, stream, a);
        else hipLaunchKernelGGL(k_canon_encode<1>, grid, dim3(ENC_THREADS), 0, stream, a);
        hipLaunchKernelGGL(k_canon_trees, gf_tile_grid((a.nTiles + CN_TW - 1) / CN_TW), dim3(64 * CN_TW), 0, stream, a);
    } else {
        hipLaunchKernelGGL(k_canon_encode<0>, grid, dim3(ENC_THREADS), 0, stream, a);
    }
    hipLaunchKernelGGL(k_canon_pack, grid, dim3(ENC_THREADS), 0, stream, a);
    return hipGetLastError();
}

size_t gf_canon_pack_rec_words() { return (size_t)CN_PACK_REC_WORDS; }
size_t gf_canon_stat_words() { return (size_t)CN_STAT_WORDS; }
