// gvrs_encode.hip -- CodecHuffman.encode for a batch of tiles, one workgroup per tile.
//
// Replaces (reference, core/src/main/java/org/gridfour/):
//   compress/CodecHuffman.java:70-130          null scan, 3 predictor passes, keep shortest
//   compress/PredictorModel{Differencing,Linear,Triangle,DifferencingWithNulls}.java encode
//   compress/CodecM32.java:257-311             signed varint
//   compress/HuffmanEncoder.java:124-305       histogram, tree, serialisation, text
//   io/BitOutputStore.java:205-288             LSB-first bit order
//
// Two kernels: k_huffman_encode runs phases A and B and the selection and leaves the winner's code table and tree
// image in a per-tile record; k_huffman_pack runs phase C from it (each with its own register budget, see below).
// Phases of a workgroup (256 threads = 4 waves) on one tile
//   A  one flat pass over the tile, 8 consecutive cells per thread (two 16-byte loads + the row
//      above + 3 halo words, all issued before use): residuals of all three predictors, M32
//      bytes, three 256-bin histograms in LDS (replicated to spread the atomics).  The common
//      single-byte residual is straight-line code; multi-byte residuals take a masked side path.
//   B  B1 waves 0..2 bitonic-sort the used symbols of one predictor each, B2 the sequential
//      tree merges run SIMT (lane p of wave 0 = predictor p), B3 waves 0..2 derive codes, the
//      serialised header+tree image and the exact bit total of their predictor
//   C  the shortest candidate wins (ties: D, L, T order, CodecHuffman.java:107).  Its stream is
//      [short border segment] + [flat scan of the cells with an emit mask], which is exactly
//      the order the reference emits residuals in; code lengths are prefix-summed across the
//      workgroup and each thread ORs its bits into an LDS window that is flushed to the
//      tile's output slot with coalesced dword stores (tile re-read hits L2).
// HBM traffic per cell: 4 B read + c B written; everything else stays on chip.
// This integer pipeline is bound by dependent-instruction latency at the occupancy LDS and registers allow, not by
// bandwidth.

#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <type_traits>

#include "gvrs_kernels.h"
#include "gvrs_encode_layout.h"

namespace {

#include "gvrs_encode_common.h"


// (the diagnostic flavour dumps the trees in the full layout from both kernels)
#ifdef GF_DIAG
template <bool FAST> using EncTree = GfHuffTree;
#else
template <bool FAST> using EncTree = std::conditional_t<FAST, GfHuffTreeSlim, GfHuffTree>;
#endif
template <bool FAST>
union EncScratchT {
    uint32_t histR[3][256 * HIST_R];            // phase A
    EncTree<FAST> tree[3];                      // phase B
};


// reduced histogram of predictor slot p (see EncPersist)
__device__ __forceinline__ uint32_t *enc_hist(EncPersist &P, int p)
{
#ifdef GF_ENC_HIST_SEPARATE
    return P.hist[p];
#else
    return reinterpret_cast<uint32_t *>(P.tab[p]);
#endif
}

// first M32 byte of residual x and whether it is the whole encoding (CodecM32.java:257-283)
__device__ __forceinline__ uint32_t m32_first_byte(uint32_t x, bool *single)
{
    const bool isNull = x == GF_NULL_CODE;
    const bool one = (x + 126u) <= 252u || isNull;       // -126..126, or Integer.MIN_VALUE -> 0x80
    *single = one;
    const uint32_t intro = (int32_t)x < 0 ? 0x81u : 0x7fu;
    return one ? (isNull ? 0x80u : (x & 0xffu)) : intro;
}


// Huffman tree of one predictor by data-parallel rounds (one wave).  K holds the live nodes in
// list order as keys (count << 9 | tie), 4 per lane, dead slots = 0xFFFFFFFF.  Each round pairs up
// all nodes whose count is below x0 + x1 -- exactly the next merges of the sequential algorithm
// of HuffmanEncoder.java:165-194, none of which can be affected by a branch created in the same
// round -- then re-sorts.  tie: leaf = 256 + sorted index, branch k = 254 - k, so that newer
// branches precede older ones and leaves of equal count (:175-193).  ~15 rounds instead of
// ~150 dependent merges.  Writes T.parent / T.left / T.nl.  Needs total count < 2^23.
// A round's P new nodes and the nodes it left, both in ascending order, merged by RANK (round 4): a node that stays moves down by
// the 2 P places of the pairs and up by the number of new nodes below it, a new node goes behind the old and the new ones below it;
// every key is written to its place in `scr` (LDS, 64 NR words) and read back in order.  The keys are distinct (count, then a tie
// field no two nodes share), so this is the permutation a sort of the keys makes.  For a round of at most eight pairs -- ten of the
// fifteen rounds of a terrain tile's tree -- it is 30 to 130 instructions where the bitonic network over 64 NR keys is 100 to 580.
// (It buys nothing where the wave that builds the tree is alone with its latencies -- the one-kernel form --, and a third of
// k_huffman_trees, whose waves are all busy.)
template <int NR>
__device__ __forceinline__ void wave_huff_merge_few(uint32_t (&K)[4], uint32_t P, uint32_t Lold, int lane, uint32_t *scr)
{
    // the new keys are elements 0, 2, .. 14 (P <= 8): lanes of K[0].  Equal sums make a LATER pair's node the smaller key -- its tie
    // field counts down --, so a new node counts the new nodes below it as well.
    uint32_t below[NR], newRank = 0, newBelow = 0;
    const bool isNew = (uint32_t)lane < 2u * P && !(lane & 1);
    bool old[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uint32_t e = (uint32_t)(r * 64 + lane);
        below[r] = 0;
        old[r] = e >= 2u * P && e < Lold;
    }
#pragma unroll
    for (uint32_t i = 0; i < 8u; i++) {
        if (i < P) {                                               // wave-uniform
            const uint32_t Ni = (uint32_t)__builtin_amdgcn_readlane((int)K[0], (int)(2u * i));
            uint32_t cnt = 0;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                below[r] += (old[r] && Ni < K[r]) ? 1u : 0u;
                cnt += (uint32_t)__popcll(__ballot(old[r] && K[r] < Ni));
            }
            newBelow += (isNew && Ni < K[0]) ? 1u : 0u;
            newRank = (uint32_t)lane == 2u * i ? cnt : newRank;
        }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uint32_t e = (uint32_t)(r * 64 + lane);
        if (old[r]) scr[e - 2u * P + below[r]] = K[r];
    }
    if (isNew) scr[newRank + newBelow] = K[0];
    const uint32_t Lnew = Lold - P;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uint32_t e = (uint32_t)(r * 64 + lane);
        K[r] = e < Lnew ? scr[e] : 0xFFFFFFFFu;
    }
}

// scr: 256 words of LDS of the wave's own that nothing else needs during the rounds, or null (every round a sort)
template <class Tree>
__device__ __forceinline__ void wave_huff_rounds(Tree &T, uint32_t (&K)[4], int n, int lane, uint32_t *scr = nullptr)
{
    uint32_t L = (uint32_t)n, kbase = 0;
    const uint32_t un = (uint32_t)n;
    while (L > 1) {
        const uint32_t s0 = ((uint32_t)__builtin_amdgcn_readlane((int)K[0], 0) >> 9) +
                            ((uint32_t)__builtin_amdgcn_readlane((int)K[0], 1) >> 9);
        uint32_t t = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) t += (uint32_t)__popcll(__ballot((K[r] >> 9) < s0));
        const uint32_t P = t >> 1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if ((uint32_t)(r * 64) < 2u * P) {                      // wave-uniform
                const uint32_t e = (uint32_t)(r * 64 + lane);
                const bool inPair = e < 2u * P;
                const uint32_t mine = K[r];
                const uint32_t other = gf_lane_xor(mine, 1);
                const uint32_t tieM = mine & 511u, tieO = other & 511u;
                const uint32_t idM = tieM >= 256u ? tieM - 256u : un + (254u - tieM);
                const uint32_t idO = tieO >= 256u ? tieO - 256u : un + (254u - tieO);
                const uint32_t k = kbase + (e >> 1);
                const uint32_t parent = un + k;
                const bool right = e & 1u;
                if (inPair) {
                    T.parent[idM] = (uint16_t)(parent | (right ? 0x8000u : 0u));
                    if (!right) {
                        const uint32_t nlL = idM < un ? 1u : T.nl[idM];
                        const uint32_t nlR = idO < un ? 1u : T.nl[idO];
                        T.left[k] = (uint16_t)idM;
                        T.nl[parent] = (uint16_t)(nlL + nlR);
                    }
                    K[r] = right ? 0xFFFFFFFFu : ((((mine >> 9) + (other >> 9)) << 9) | (254u - k));
                }
            }
        }
        kbase += P;
        const uint32_t Lold = L;
        L -= P;
        if (scr && P <= 8u) {
            if (Lold > 128) wave_huff_merge_few<4>(K, P, Lold, lane, scr);
            else if (Lold > 64) wave_huff_merge_few<2>(K, P, Lold, lane, scr);
            else wave_huff_merge_few<1>(K, P, Lold, lane, scr);
        } else if (Lold > 128) wave_bitonic_sort<4>(K, lane);
        else if (Lold > 64) wave_bitonic_sort<2>(K, lane);
        else wave_bitonic_sort<1>(K, lane);
    }
    if (lane == 0 && n >= 1) T.parent[2 * n - 2] = 0xFFFF;         // root
}

// diagnostics: cycle stamps per phase, dump of the on-chip tables, phase ablation -- only in the -DGF_DIAG flavour of the
// library (tools/); the shipping kernels carry none of it
#ifdef GF_DIAG
#define GF_STAMP(i)                                                                       \
    do {                                                                                  \
        if (a.debug && tid == 0)                                                          \
            (a.debug + t * (size_t)GF_ENC_DEBUG_WORDS + GF_ENC_DEBUG_WORDS - 16)[i] =    \
                (uint32_t)__builtin_amdgcn_s_memtime();                                   \
    } while (0)
// (inside phase B a tree is built by the wave of its predictor: the stamp is that wave's, the last one to pass wins)
#define GF_STAMP_WAVE(i)                                                                  \
    do {                                                                                  \
        if (a.debug && lane == 0)                                                         \
            (a.debug + t * (size_t)GF_ENC_DEBUG_WORDS + GF_ENC_DEBUG_WORDS - 16)[i] =    \
                (uint32_t)__builtin_amdgcn_s_memtime();                                   \
    } while (0)
#else
#define GF_STAMP(i) do { } while (0)
#define GF_STAMP_WAVE(i) do { } while (0)
#endif

// The continuation bytes of a wide M32 value (CodecM32.java:283-311) without the general form's loop and switch for the lengths
// terrain has: two bytes -- |x| - 127 as one byte --, three -- |x| - 255 as two 7-bit groups.  Returns the byte count n; b1 / b2 are
// set for n <= 3, longer values go through gf_m32_byte.
__device__ __forceinline__ uint32_t m32_wide_bytes(uint32_t x, uint32_t *b1, uint32_t *b2)
{
    const uint32_t mag = (int32_t)x < 0 ? 0u - x : x;
    if (mag <= 254u) { *b1 = mag - 127u; *b2 = 0; return 2u; }
    if (mag <= 16638u) {
        const uint32_t d = mag - 255u;
        *b1 = 0x80u | (d >> 7);
        *b2 = d & 0x7fu;
        return 3u;
    }
    *b1 = 0;
    *b2 = 0;
    return (uint32_t)gf_m32_len(x);
}

// stream elements [sBegin, sEnd) of `model`, any residual size: the general (slow) packer
// (The packers are real calls: the pack state travels by value and comes back as the result -- a reference parameter of
// a function that is not inlined is a stack object, i.e. scratch memory.)
__device__ PackState pack_generic(int model, const uint32_t *__restrict__ tile, uint32_t nR, uint32_t nC, uint32_t seed,
                                  const uint64_t *tab, uint32_t elemMaxBits, uint32_t sBegin, uint32_t sEnd,
                                  uint32_t *win, uint32_t *__restrict__ out32, uint32_t *waveSum, PackState ps)
{
    const uint32_t tid = threadIdx.x;
    // elements per chunk such that a chunk can never overflow the window
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
                const uint32_t x = cell_residual(model, tile, nC, idx, r, c, seed);
                xs[e] = x;
                const int n = gf_m32_len(x);
                for (int k = 0; k < n; k++) myBits += (uint32_t)(tab[gf_m32_byte(x, n, k)] >> 56);
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
                if (e < E && tid < active && s < cEnd) {
                    const uint32_t x = xs[e];
                    const int n = gf_m32_len(x);
                    for (int k = 0; k < n; k++) {
                        const uint64_t cl = tab[gf_m32_byte(x, n, k)];
                        sink.put(cl & 0x00ffffffffffffffull, (uint32_t)(cl >> 56));
                    }
                }
            }
            sink.finish();
        }
        __syncthreads();
        ps.bitBase += total;
        window_flush(win, out32, ps);
    }
    return ps;
}

// the short head of a Linear / Triangle stream (first column and first row(s): nR + nC elements or so) in the fast case,
// one element per thread and chunk, INLINED: as a call into pack_generic it cost every tile the callee's register saves --
// twelve VGPRs per lane stored to and reloaded from scratch, a third of a gigabyte per launch on the bench batch.
// Requires 256 * elemMaxBits <= window bits (implied by the kernels' `fast` condition).
template <int MODEL>
__device__ __forceinline__ PackState pack_head(const uint32_t *__restrict__ tile, uint32_t nR, uint32_t nC, uint32_t seed,
                                               const uint64_t *tab, uint32_t sEnd, uint32_t *win, uint32_t *__restrict__ out32,
                                               uint32_t *waveSum, PackState ps)
{
    const uint32_t tid = threadIdx.x;
    for (uint32_t chunk = 0; chunk < sEnd; chunk += ENC_THREADS) {
        const uint32_t s = chunk + tid;
        uint32_t x = 0, myBits = 0;
        int n = 0;
        if (s < sEnd) {
            const uint32_t idx = gf_stream_cell(MODEL, nR, nC, s);
            const uint32_t r = idx / nC, c = idx - r * nC;
            x = cell_residual(MODEL, tile, nC, idx, r, c, seed);
            n = gf_m32_len(x);
            for (int k = 0; k < n; k++) myBits += (uint32_t)(tab[gf_m32_byte(x, n, k)] >> 56);
        }
        uint32_t total;
        const uint32_t excl = block_excl_scan(myBits, waveSum, &total);
        if (myBits) {
            BitSink sink;
            sink.init(win, ps.bitBase + excl - ps.wordBase * 32u);
            for (int k = 0; k < n; k++) {
                const uint64_t cl = tab[gf_m32_byte(x, n, k)];
                sink.put(cl & 0x00ffffffffffffffull, (uint32_t)(cl >> 56));
            }
            sink.finish();
        }
        __syncthreads();
        ps.bitBase += total;
        window_flush(win, out32, ps);
    }
    return ps;
}

// the main segment of a model's stream = flat scan over the cells with an emit mask; fast path
// for residuals whose codes fit the window at CPT cells per thread
template <int MODEL>
__device__ PackState pack_flat(const uint32_t *__restrict__ tile, uint32_t nC, uint32_t nCells, uint32_t seed,
                               const uint64_t *tab, uint32_t *win, uint32_t *__restrict__ out32, uint32_t *waveSum,
                               PackState ps, uint32_t cellBegin = 0, uint32_t cellEnd = 0xFFFFFFFFu)
{
    // cells [cellBegin, cellEnd): cellBegin a multiple of CPT, cellEnd a multiple of CPT or the end of the tile
    const uint32_t tid = threadIdx.x;
    cellEnd = min(cellEnd, nCells);
    uint32_t c0 = (cellBegin + tid * CPT) % nC;
    const uint32_t cStep = STEP_CELLS % nC;
    for (uint32_t base = cellBegin; base < cellEnd; base += STEP_CELLS) {
        const uint32_t i0 = base + tid * CPT;
        uint64_t cl[CPT];
        uint32_t xs[CPT];
        uint32_t myBits = 0, multi = 0;
        if (i0 < cellEnd) {
            Cells8 Q;
            load_cells8_wave(tile, nC, nCells, i0, Q);
            uint32_t c = c0;
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                bool emit, single;
                const uint32_t x = flat_residual<MODEL>(Q, j, i0 + j, c, nC, nCells, seed, &emit);
                const uint32_t b0 = m32_first_byte(x, &single);
                const uint64_t e = emit ? tab[b0] : 0ull;
                cl[j] = e;
                xs[j] = x;
                myBits += (uint32_t)(e >> 56);
                if (emit && !single) multi |= 1u << j;
                if (++c == nC) c = 0;
            }
            if (multi) {                                              // continuation bytes (rare)
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    if ((multi >> j) & 1u) {
                        const uint32_t x = xs[j];
                        const int n = gf_m32_len(x);
                        for (int k = 1; k < n; k++) myBits += (uint32_t)(tab[gf_m32_byte(x, n, k)] >> 56);
                    }
                }
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
                sink.put(cl[j] & 0x00ffffffffffffffull, (uint32_t)(cl[j] >> 56));
                if ((multi >> j) & 1u) {
                    const uint32_t x = xs[j];
                    const int n = gf_m32_len(x);
                    for (int k = 1; k < n; k++) {
                        const uint64_t e = tab[gf_m32_byte(x, n, k)];
                        sink.put(e & 0x00ffffffffffffffull, (uint32_t)(e >> 56));
                    }
                }
            }
            sink.finish();
        }
        __syncthreads();
        ps.bitBase += total;
        window_flush(win, out32, ps);
    }
    return ps;
}

// The same segment with wave-private bit windows: every wave packs one contiguous quarter of the cells into its own
// quarter of the LDS window, starting at bit 0, with a wave-level scan and no workgroup barrier inside the loop; the
// four bit strings are then shifted into place (they follow one another in the stream) and written out.  Returns
// false -- nothing written, ps untouched -- when a quarter does not fit its window; the caller then uses pack_flat.
struct PackStateOk {
    PackState ps;
    bool ok;
};

template <int MODEL, bool PLAIN>
__device__ PackStateOk pack_flat_waves(const uint32_t *__restrict__ tile, uint32_t nC, uint32_t nCells, uint32_t seed,
                                       const uint64_t *tab, uint32_t *win, uint32_t *__restrict__ out32, uint32_t *waveSum,
                                       PackState ps, uint32_t slotWords, uint32_t cellBegin, uint32_t cellEnd)
{
    // cells [cellBegin, cellEnd) as for pack_flat; on return the window again holds the partial last word at win[0]
    // (the wave index as a SCALAR: what derives from it -- the wave's cell range, its window, the loop bounds -- then lives in
    // scalar registers and the loop is a scalar loop; as `tid >> 6` the compiler has to treat all of it as per-lane values)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = GF_UNI(tid >> 6);
    cellEnd = min(cellEnd, nCells);
    const uint32_t carryWord = wave_windows_begin(win, waveSum);
    const uint32_t quarter = (((cellEnd - cellBegin + ENC_WAVES - 1) / ENC_WAVES) + CPT - 1) / CPT * CPT;
    const uint32_t segBegin = min(cellEnd, cellBegin + wave * quarter), segEnd = min(cellEnd, segBegin + quarter);
    uint32_t *wwin = win + wave * WAVE_WIN;
    const uint32_t capBits = WAVE_WIN_BITS;
    uint32_t bits = 0;
    bool fits = true;
    uint32_t c0 = (segBegin + lane * CPT) % nC;
    const uint32_t cStep = (64u * CPT) % nC;
    for (uint32_t base = segBegin; base < segEnd; base += 64u * CPT) {
        const uint32_t i0 = base + lane * CPT;
        // PLAIN (round 4): the record says that every value of the stream is one plain M32 byte (k_huffman_encode looked at the
        // winner's symbols): the byte is the residual's low eight bits, there are no continuation bytes to look for and nothing
        // of a residual to keep across the scan
        uint64_t cl[CPT];
        uint32_t xs[PLAIN ? 1 : CPT];
        uint32_t myBits = 0, multi = 0, hard = 0;
        if (i0 < segEnd) {
            Cells8 Q;
            load_cells8_wave(tile, nC, nCells, i0, Q);
            uint32_t c = c0;
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                bool emit, single = true;
                const uint32_t x = flat_residual<MODEL>(Q, j, i0 + j, c, nC, nCells, seed, &emit);
                const uint32_t b0 = PLAIN ? (x & 0xffu) : m32_first_byte(x, &single);
                const uint64_t e = emit ? tab[b0] : 0ull;
                cl[j] = e;
                if (!PLAIN) xs[j] = x;
                myBits += (uint32_t)(e >> 56);
                if (!PLAIN && emit && !single) multi |= 1u << j;
                if (++c == nC) c = 0;
            }
            if (!PLAIN && multi) {                                    // continuation bytes
                // (round 5) a value of two or three bytes -- what rough terrain has -- becomes ONE code here: the codes of its bytes
                // joined in stream order, at most 56 bits, in the cell's (code, length) pair; the emission below then sees eight
                // codes whatever their origin.  Longer values, or joins beyond 56 bits, stay `hard`: first byte in the pair, the
                // rest through the bit sink.
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    if ((multi >> j) & 1u) {
                        const uint32_t x = xs[PLAIN ? 0 : j];
                        uint32_t b1, b2;
                        const uint32_t n = m32_wide_bytes(x, &b1, &b2);
                        const uint32_t l0 = (uint32_t)(cl[j] >> 56);
                        if (n <= 3u) {
                            const uint64_t e1 = tab[b1], e2 = n == 3u ? tab[b2] : 0ull;
                            const uint32_t l1 = (uint32_t)(e1 >> 56), l2 = (uint32_t)(e2 >> 56), lt = l0 + l1 + l2;
                            myBits += l1 + l2;
                            if (lt <= 56u) {
                                const uint64_t c = (cl[j] & 0x00ffffffffffffffull) | ((e1 & 0x00ffffffffffffffull) << l0) |
                                                   ((e2 & 0x00ffffffffffffffull) << (l0 + l1));
                                cl[j] = c | ((uint64_t)lt << 56);
                            } else {
                                hard |= 1u << j;
                            }
                        } else {
                            hard |= 1u << j;
                            for (uint32_t k = 1; k < n; k++) myBits += (uint32_t)(tab[gf_m32_byte(x, (int)n, (int)k)] >> 56);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) cl[j] = 0;
#pragma unroll
            for (int j = 0; j < (PLAIN ? 1 : CPT); j++) xs[j] = 0;
        }
        c0 += cStep;
        if (c0 >= nC) c0 -= nC;

        const uint32_t incl = gf_wave_incl_scan(myBits);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (bits + total > capBits) { fits = false; break; }         // wave-uniform
        // the usual lane -- eight one-byte values whose codes make two groups of four of at most 32 bits each -- joins its codes
        // in registers (GF_JOIN8_OR, gvrs_encode_common.h).  Round 5: a lane with longer codes -- a rare symbol, a joined
        // multi-byte value -- joins them in PAIRS of at most 64 bits and ORs every pair into the window as three words
        // (GF_JOIN2_OR64): on rough terrain some lane of nearly every wave is such a lane, and through the bit sink (a branch or
        // two per code, the whole wave waiting) it made the packer 2.2 times the smooth batch's.  What is left to the sink: values
        // of four bytes and more, joins beyond 56 bits, a pair beyond 64.
#define GF_LN(j) ((uint32_t)(cl[j] >> 56))
#define GF_CD(j) ((uint32_t)cl[j])
        const uint32_t n01 = GF_LN(0) + GF_LN(1), n45 = GF_LN(4) + GF_LN(5);
        const uint32_t n23 = GF_LN(2) + GF_LN(3), n67 = GF_LN(6) + GF_LN(7);
        const uint32_t n0 = n01 + n23, n1 = n45 + n67;
        static_assert(CPT == 8, "the register path joins eight codes");
        if ((PLAIN || multi == 0u) && n0 <= 32u && n1 <= 32u) {
            GF_JOIN8_OR(wwin, bits + incl - myBits, GF_CD, GF_LN, n01, n45, n0);
#undef GF_LN
#undef GF_CD
#ifdef GF_PACK_NO_PAIRS                                             // (experiment builds: tools/ab.sh)
        } else if (false) {
#else
        } else if (hard == 0u && max(max(n01, n23), max(n45, n67)) <= 64u) {
#endif
            uint32_t pos = bits + incl - myBits;
#pragma unroll
            for (int q = 0; q < CPT; q += 2) {
                const uint32_t la = (uint32_t)(cl[q] >> 56), lb = (uint32_t)(cl[q + 1] >> 56);
                const uint64_t ca = cl[q] & 0x00ffffffffffffffull, cb = cl[q + 1] & 0x00ffffffffffffffull;
                GF_JOIN2_OR64(wwin, pos, ca, la, cb, lb);
                pos += la + lb;
            }
        } else if (myBits) {
            BitSink sink;
            sink.init(wwin, bits + incl - myBits);
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                sink.put(cl[j] & 0x00ffffffffffffffull, (uint32_t)(cl[j] >> 56));
                if (!PLAIN && ((hard >> j) & 1u)) {
                    const uint32_t x = xs[PLAIN ? 0 : j];
                    uint32_t b1, b2;
                    const uint32_t n = m32_wide_bytes(x, &b1, &b2);
                    if (n <= 3u) {
                        const uint64_t e1 = tab[b1];
                        sink.put(e1 & 0x00ffffffffffffffull, (uint32_t)(e1 >> 56));
                        if (n == 3u) {
                            const uint64_t e2 = tab[b2];
                            sink.put(e2 & 0x00ffffffffffffffull, (uint32_t)(e2 >> 56));
                        }
                    } else {
                        for (uint32_t k = 1; k < n; k++) {
                            const uint64_t e = tab[gf_m32_byte(x, (int)n, (int)k)];
                            sink.put(e & 0x00ffffffffffffffull, (uint32_t)(e >> 56));
                        }
                    }
                }
            }
            sink.finish();
        }
        bits += total;
    }
    PackStateOk r;
    r.ok = wave_windows_end(win, waveSum, carryWord, bits, fits, out32, slotWords, ps);
    r.ps = ps;
    return r;
}

// Eight (code, length) pairs of one-byte values at bit `pos` of a wave's window: the register join, the join in pairs, the bit sink
// (see pack_flat_waves).
__device__ __forceinline__ void emit8_codes(uint32_t *wwin, uint32_t pos, const uint64_t (&cl)[CPT], uint32_t myBits)
{
#define GF_LN(j) ((uint32_t)(cl[j] >> 56))
#define GF_CD(j) ((uint32_t)cl[j])
    const uint32_t n01 = GF_LN(0) + GF_LN(1), n45 = GF_LN(4) + GF_LN(5);
    const uint32_t n23 = GF_LN(2) + GF_LN(3), n67 = GF_LN(6) + GF_LN(7);
    const uint32_t n0 = n01 + n23, n1 = n45 + n67;
    if (n0 <= 32u && n1 <= 32u) {
        GF_JOIN8_OR(wwin, pos, GF_CD, GF_LN, n01, n45, n0);
#undef GF_LN
#undef GF_CD
    } else if (max(max(n01, n23), max(n45, n67)) <= 64u) {
#pragma unroll
        for (int q = 0; q < CPT; q += 2) {
            const uint32_t la = (uint32_t)(cl[q] >> 56), lb = (uint32_t)(cl[q + 1] >> 56);
            const uint64_t ca = cl[q] & 0x00ffffffffffffffull, cb = cl[q + 1] & 0x00ffffffffffffffull;
            GF_JOIN2_OR64(wwin, pos, ca, la, cb, lb);
            pos += la + lb;
        }
    } else if (myBits) {
        BitSink sink;
        sink.init(wwin, pos);
#pragma unroll
        for (int j = 0; j < CPT; j++) sink.put(cl[j] & 0x00ffffffffffffffull, (uint32_t)(cl[j] >> 56));
        sink.finish();
    }
}

// (round 6) pack_flat_waves for a PLAIN stream from the tile's byte plane of raw row differences (GfEncodeArgs::plane) -- cells
// [cellBegin, cellEnd) through the wave-private windows -- and, in front of the tile's first range (headElems != 0), the head of a
// Linear or Triangle stream: first row, first column(s), which are plane bytes as they stand (PredictorModelLinear.java:113-126,
// PredictorModelTriangle.java:114-127).  Wave 0 takes the head as turns of its own in front of its share of the cells, eight
// elements per lane like any turn: no pass of the workgroup over the head, no barriers, no flush for it.  A tile's time in the packer
// was five round trips to memory one behind the other (status; record; two chunks of the head; the first turn's cells) around nine
// turns of work; here the head's bytes and the first turn's words are asked for together, before the windows are cleared, and every
// later turn's words a turn ahead (five registers per lane; the tile's nineteen left no room for that).
template <int MODEL>
__device__ PackStateOk pack_plane_waves(const uint8_t *__restrict__ plane, uint32_t nC, uint32_t nCells, const uint64_t *tab, uint32_t *win,
                                        uint32_t *__restrict__ out32, uint32_t *waveSum, PackState ps, uint32_t slotWords,
                                        uint32_t cellBegin, uint32_t cellEnd, uint32_t headElems)
{
    static_assert(MODEL >= 1 && MODEL <= 3, "the byte plane serves the three predictors");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = GF_UNI(tid >> 6);
    cellEnd = min(cellEnd, nCells);
    const uint32_t quarter = (((cellEnd - cellBegin + ENC_WAVES - 1) / ENC_WAVES) + CPT - 1) / CPT * CPT;
    const uint32_t segBegin = min(cellEnd, cellBegin + wave * quarter), segEnd = min(cellEnd, segBegin + quarter);
    const uint32_t lastI0 = (nCells - 1u) & ~(uint32_t)(CPT - 1);
    // what memory has to give first: the first turn's words and, on wave 0, the head's bytes
    PlaneWords ahead = plane_load<MODEL>(plane, nC, min(segBegin + lane * CPT, lastI0));
    const bool withHead = MODEL != 1 && headElems != 0u && wave == 0u;     // (wave-uniform)
    uint32_t hb[2] = {0x80808080u, 0x80808080u};
    if constexpr (MODEL != 1) {
        if (withHead) plane_head_bytes<MODEL>(plane, nC, lane * CPT, headElems, hb);
    }
    const uint32_t carryWord = wave_windows_begin(win, waveSum);
    uint32_t *wwin = win + wave * WAVE_WIN;
    const uint32_t capBits = WAVE_WIN_BITS;
    uint32_t bits = 0;
    bool fits = true;
    if constexpr (MODEL != 1) {
        if (withHead) {
            for (uint32_t h = 0; h < headElems; h += 64u * CPT) {       // (one turn up to 512 elements: tiles of up to 256 rows)
                if (h) plane_head_bytes<MODEL>(plane, nC, h + lane * CPT, headElems, hb);
                uint64_t cl[CPT];
                uint32_t myBits = 0;
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    const uint64_t e = tab[(hb[j >> 2] >> (8 * (j & 3))) & 0xffu];
                    cl[j] = e;
                    myBits += (uint32_t)(e >> 56);
                }
                const uint32_t incl = gf_wave_incl_scan(myBits);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (bits + total > capBits) { fits = false; break; }     // wave-uniform
                emit8_codes(wwin, bits + incl - myBits, cl, myBits);
                bits += total;
            }
        }
    }
    uint32_t c0 = (segBegin + lane * CPT) % nC;
    const uint32_t cStep = (64u * CPT) % nC;
    for (uint32_t base = segBegin; base < segEnd && fits; base += 64u * CPT) {
        const uint32_t i0 = base + lane * CPT;
        uint64_t cl[CPT];
        uint32_t myBits = 0;
        const PlaneWords now = ahead;
        ahead = plane_load<MODEL>(plane, nC, min(i0 + 64u * CPT, lastI0));
        if (i0 < segEnd) {
            uint32_t rb[2];
            plane_residual_bytes<MODEL>(now, nC, i0, rb);
            // which of the eight cells the flat scan emits, as a bit mask (nC >= 8: at most one of them starts a row): not the
            // cells behind the tile; Differencing: not the seed; Linear: not the first two cells of a row; Triangle: not the first
            // row, not the first cell of a row.  A cell that is not emitted gets the byte 0x80, which no plain stream holds and
            // whose table entry the packer has emptied: eight table reads without a condition.
            const uint32_t kz = c0 == 0u ? 0u : min(nC - c0, 16u);              // the place of the cell that starts a row (>= 8: none)
            // (the wave's turn as a whole -- scalars: its first cell and the tile's shape)
            const bool edgeTurn = base + 64u * CPT > nCells || (MODEL == 3 && base < nC) || (MODEL == 1 && base == 0u);
            if (edgeTurn) {
                uint32_t em = 0xffu;
                const uint32_t left = nCells - i0;
                if (left < (uint32_t)CPT) em = (1u << left) - 1u;
                if constexpr (MODEL == 1) {
                    if (i0 == 0u) em &= ~1u;
                } else if constexpr (MODEL == 2) {
                    em &= ~(3u << kz);
                    if (c0 == 1u) em &= ~1u;
                } else {
                    em &= ~(1u << kz);
                    if (i0 < nC) em &= ~((1u << min((uint32_t)CPT, nC - i0)) - 1u);
                }
                const uint32_t y0 = __umul24(em & 15u, 0x00204081u) & 0x01010101u, y1 = __umul24((em >> 4) & 15u, 0x00204081u) & 0x01010101u;
                constexpr uint32_t H = 0x80808080u;
                const uint32_t m0 = (H - y0) ^ H, m1 = (H - y1) ^ H;           // a byte of ones per emitted cell (no borrow crosses a byte)
                rb[0] = (rb[0] & m0) | (0x80808080u & ~m0);
                rb[1] = (rb[1] & m1) | (0x80808080u & ~m1);
            } else if constexpr (MODEL != 1) {
                // a turn inside the tile (all but its first and last): the cells that start a row are all there is to leave out -- a
                // byte's mask shifted to its place (Linear: two bytes, and the second cell of a row that started in the lane before)
                unsigned long long m = kz < (uint32_t)CPT ? (MODEL == 2 ? 0xffffull : 0xffull) << (8u * kz) : 0ull;
                if (MODEL == 2 && c0 == 1u) m |= 0xffull;
                const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
                rb[0] = (rb[0] & ~m0) | (0x80808080u & m0);
                rb[1] = (rb[1] & ~m1) | (0x80808080u & m1);
            }
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const uint64_t e = tab[(rb[j >> 2] >> (8 * (j & 3))) & 0xffu];
                cl[j] = e;
                myBits += (uint32_t)(e >> 56);
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) cl[j] = 0;
        }
        c0 += cStep;
        if (c0 >= nC) c0 -= nC;
        const uint32_t incl = gf_wave_incl_scan(myBits);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (bits + total > capBits) { fits = false; break; }             // wave-uniform
        emit8_codes(wwin, bits + incl - myBits, cl, myBits);
        bits += total;
    }
    PackStateOk r;
    r.ok = wave_windows_end(win, waveSum, carryWord, bits, fits, out32, slotWords, ps);
    r.ps = ps;
    return r;
}

// Two kernels per batch: k_huffman_encode (phases A and B and the selection) and k_huffman_pack (phase C).  Fused into
// one kernel the phases shared one register budget (96 VGPRs at five workgroups per CU, 428 bytes of scratch per lane,
// the calls into the packers saving and restoring two dozen registers); apart, they run at six workgroups per CU
// (80 VGPRs) resp. eight (64 VGPRs, with the wave-private windows of pack_flat_waves): 1.63 -> 1.40 ms on the ETOPO1-shaped
// batch (sweep: 4..8 workgroups per CU each).
#ifndef GF_ENC_PACK_WGS
#define GF_ENC_PACK_WGS 8        // sweep: 8 -> 1.137 ms (15 spilled registers: 0.65 GB of scratch traffic), 5 -> 1.208 ms, no spills
#endif
constexpr int ENC_AB_WGS = ENC_THREADS == 256 ? 6 : 1, ENC_PACK_WGS = ENC_THREADS == 256 ? GF_ENC_PACK_WGS : 1;

constexpr int GF_K_RETRY = 0x7fff0002;          // internal: the fast kernel leaves this tile to the general one

// Two instantiations of one body.  FAST is what a batch of terrain tiles consists of: no null cells, symbol counts below
// 2^23 (the in-register sort and the data-parallel tree rounds).  A tile with nulls (its own predictor, seed from a mean,
// a second histogram pass) is marked GF_K_RETRY and left to the general instantiation, which with a.retryFlag touches only
// marked tiles and returns at once when there are none.  The general body is twice the size of the fast one and carries the
// register pressure of the rare paths into every tile's allocation.
// PART 0: phases A and B and the selection in one kernel (the general kernel, the one-tile-per-call build, the diagnostic flavour).
// PART 1 (round 4, CodecHuffman batches): phase A alone -- the histograms and what else the trees need go to GfEncodeArgs::encStats,
// and k_huffman_trees takes it from there with ONE WAVE PER TILE.  In the one-kernel form the tree of a tile is built by one wave
// (sort, tree rounds, codes: a chain of dependent steps, 42 % of a tile's time) while the workgroup's other three hold their wave
// slots idle: a CU had fewer than three such chains in flight; the tree kernel keeps thirty-two.
// EncPersistA: what phase A keeps of EncPersist -- the part-1 kernel carries no code tables and no tree images, 12.4 instead of
// 19.6 KB of LDS per workgroup.
struct EncPersistA {
    uint32_t maxN[3];
    uint32_t nM32[3];
    int32_t model[3];
    uint32_t seed;
    uint32_t flags;
    unsigned long long sumStart;
    uint32_t nStart;
};
#ifndef GF_ENC_HIST_R_A
#define GF_ENC_HIST_R_A 4        // histogram replicas of the part-1 kernel (sweep: see DESIGN.md)
#endif
union EncScratchA {
    uint32_t histR[3][256 * GF_ENC_HIST_R_A];
};

#ifndef GF_ENC_A_WGS
#define GF_ENC_A_WGS 7           // workgroups per CU the part-1 kernel is compiled for: 72 VGPRs, no scratch (8: 64 VGPRs and 28 bytes of scratch per lane; measured 0.689 / 0.714 / 0.705 ms per encode with 7 / 8 / 6)
#endif
// PLANE (PART 1 only, round 6): the byte plane of raw row differences is written (GfEncodeArgs::plane is not null).  A template
// parameter and not a test of the pointer: behind a branch around the store the compiler waits for the store itself (the counter of
// outstanding memory operations is one for loads and stores and counts in order; at the join it cannot tell which path was taken).
template <bool FAST, int PART = 0, bool PLANE = false>
__global__ __launch_bounds__(ENC_THREADS, PART == 1 && ENC_THREADS == 256 && GF_ENC_HIST_R_A <= 4 ? GF_ENC_A_WGS : ENC_AB_WGS) void k_huffman_encode(GfEncodeArgs a)
{
    static_assert(!PLANE || PART == 1, "the byte plane belongs to the part-1 kernel");
    __shared__ std::conditional_t<PART == 1, EncPersistA, EncPersist> P;
    __shared__ std::conditional_t<PART == 1, EncScratchA, EncScratchT<FAST>> S;
    constexpr int HR = PART == 1 ? GF_ENC_HIST_R_A : HIST_R;              // histogram replicas

    const int tid = threadIdx.x, lane = tid & 63, wave = (int)gf_wave_id();
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    if constexpr (!FAST) {
        if (a.retryFlag && *a.retryFlag == 0u) return;               // the fast kernel finished every tile
    } else {
        // (round 5) the count of tiles the packer leaves to k_huffman_pack_rare starts at zero: cleared here, a kernel boundary
        // before the packer adds to it, instead of by a memset node in front of every encode (4 us that found nothing to do)
        if (a.retryFlag && !a.lean && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) a.retryFlag[1] = 0u;
    }

    GF_FOR_TILES(t, a.nTiles, FAST) {                                     // the fast kernel: one tile per workgroup, no loop
        if constexpr (!FAST) {
            if (a.retryFlag && a.status[t] != GF_K_RETRY) continue;
        }
        const uint32_t *__restrict__ tile = reinterpret_cast<const uint32_t *>(a.values) + t * (size_t)nCells;

        GF_STAMP(0);
        // ---------------- phase A: null scan + three histograms ----------------
        for (int i = tid; i < 3 * 256 * HR; i += ENC_THREADS) (&S.histR[0][0])[i] = 0;
        if (tid == 0) { P.flags = 0; P.sumStart = 0; P.nStart = 0; }
        if (tid < 3) { P.maxN[tid] = 1; P.model[tid] = 0; P.nM32[tid] = 0; }
        __syncthreads();

        const bool triOk = nR >= 2 && nC >= 2;
        const uint32_t rep = (uint32_t)lane & (HR - 1);
        uint32_t myFlags = 0, maxN1 = 1, maxN2 = 1, maxN3 = 1;
        {
            uint32_t *const h0 = &S.histR[0][rep], *const h1 = &S.histR[1][rep], *const h2 = &S.histR[2][rep];
            uint32_t c0 = ((uint32_t)tid * CPT) % nC;
            const uint32_t cStep = STEP_CELLS % nC;
            for (uint32_t i0 = (uint32_t)tid * CPT; i0 < nCells; i0 += STEP_CELLS) {
                Cells8 Q;
#ifdef GF_ENC_HALO_LOADS                                                 // (experiment builds: the halo words as three loads per lane)
                load_cells8(tile, nC, nCells, i0, Q);
#else
                load_cells8_wave(tile, nC, nCells, i0, Q);
#endif
                uint32_t c = c0;
                // a tile with a null cell takes the nulls predictor alone, with a histogram of its own (below): once a lane of the wave
                // has seen one, the three histograms of this pass are not needed any more -- only the scan for valid cells goes on
                // (wave-local: a wave's lanes cover 512 consecutive cells per turn, so a block of nulls reaches every wave within a turn or two)
                if (__any((myFlags & 1u) != 0u)) {
#pragma unroll
                    for (int j = 0; j < CPT; j++)
                        if (i0 + j < nCells) myFlags |= Q.cur[j] == GF_NULL_CODE ? 1u : 2u;
                    continue;
                }
                // The residuals of the three predictors for the thread's eight cells first (the seed cell and the padding behind
                // the tile count as residual 0 in every histogram; bin 0 is corrected after the reduction), then the histograms.
                // Round 4: where every lane of the wave is inside the tile (no seed cell, no first row, no padding), holds no null
                // cell and all of its 24 residuals are plain one-byte values (-126..126: the M32 byte is the residual's low byte,
                // CodecM32.java:257-283) -- every turn of a terrain tile but its first and last --, the turn is 24 additions to the
                // histograms and nothing else: no null / introducer / sign selects per residual, no "is there more" test per cell.
                uint32_t D1[CPT], D2[CPT], D3[CPT];
                const bool inside = i0 >= nC + 2u && i0 + (CPT - 1) < nCells;
                uint32_t widest = 0;                                 // max over the residuals of (d + 126) as unsigned: <= 252 = plain
                int32_t lowest = 0x7fffffff;                         // min over the cells: Integer.MIN_VALUE = a null cell
                if (__all(inside) && nC >= (uint32_t)CPT) {
                    // (round 4: phase A is bound by vector issue -- 290 instructions per wave and turn.  The three residuals from
                    // two raw differences, v - W along the row and N - NW along the row above: Linear's is the difference of two
                    // neighbouring raw differences, Triangle's the difference of the two rows'.  At most one of the eight cells
                    // starts a row (nC >= 8): k is its place, and "first / second cell of a row" is a compare of two constants'
                    // worth instead of a column counter per cell.)
                    const uint32_t toEnd = nC - c;
                    const uint32_t k = c == 0u ? 0u : (toEnd < (uint32_t)CPT ? toEnd : (uint32_t)CPT);
                    const bool second0 = c == 1u;
                    int32_t hi = -0x7fffffff - 1, lo = 0x7fffffff;
                    uint32_t rlPrev = Q.wm1 - Q.wm2;
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        const uint32_t v = Q.cur[j];
                        const uint32_t W = j > 0 ? Q.cur[j - 1] : Q.wm1;
                        const uint32_t N = Q.up[j];
                        const uint32_t NW = j > 0 ? Q.up[j - 1] : Q.upm1;
                        const uint32_t rl = v - W, ru = N - NW;
                        const bool first = (uint32_t)j == k;
                        const bool second = (uint32_t)j == k + 1u || (j == 0 && second0);
                        const uint32_t d = first ? v - N : rl;
                        D1[j] = d;
                        D2[j] = (first || second) ? d : rl - rlPrev;
                        D3[j] = first ? d : rl - ru;
                        rlPrev = rl;
                        hi = max(hi, max((int32_t)D1[j], max((int32_t)D2[j], (int32_t)D3[j])));
                        lo = min(lo, min((int32_t)D1[j], min((int32_t)D2[j], (int32_t)D3[j])));
                        lowest = min(lowest, (int32_t)v);
                    }
                    widest = (hi <= 126 && lo >= -126) ? 0u : 0xFFFFFFFFu;
                } else if (__all(inside)) {
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        const uint32_t v = Q.cur[j];
                        const uint32_t W = j > 0 ? Q.cur[j - 1] : Q.wm1;
                        const uint32_t WW = j > 1 ? Q.cur[j - 2] : (j == 1 ? Q.wm1 : Q.wm2);
                        const uint32_t N = Q.up[j];
                        const uint32_t NW = j > 0 ? Q.up[j - 1] : Q.upm1;
                        const uint32_t d = v - (c > 0 ? W : N);
                        D1[j] = d;
                        D2[j] = c >= 2 ? v - (2u * W - WW) : d;
                        D3[j] = c > 0 ? v - (W + N - NW) : d;
                        widest = max(widest, max(D1[j] + 126u, max(D2[j] + 126u, D3[j] + 126u)));
                        lowest = min(lowest, (int32_t)v);
                        if (++c == nC) c = 0;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        const uint32_t idx = i0 + j;
                        const uint32_t v = Q.cur[j];
                        const uint32_t W = j > 0 ? Q.cur[j - 1] : Q.wm1;
                        const uint32_t WW = j > 1 ? Q.cur[j - 2] : (j == 1 ? Q.wm1 : Q.wm2);
                        const uint32_t N = Q.up[j];
                        const uint32_t NW = j > 0 ? Q.up[j - 1] : Q.upm1;
                        const bool counted = idx < nCells && idx > 0;
                        const uint32_t d = v - (c > 0 ? W : N);
                        D1[j] = counted ? d : 0u;
                        D2[j] = counted ? (c >= 2 ? v - (2u * W - WW) : d) : 0u;
                        D3[j] = counted ? ((idx >= nC && c > 0) ? v - (W + N - NW) : d) : 0u;
                        if (++c == nC) c = 0;
                    }
                    widest = 0xFFFFFFFFu;                            // (the turns at the tile's ends take the general histogram code)
                }
                if constexpr (PLANE) {
                    // (round 6) the Differencing residual -- the raw row difference, column 0: the difference to the row above -- as one
                    // byte per cell for the packer (GfEncodeArgs::plane): eight bytes per lane, a wave's store is 512 consecutive bytes.
                    // The residuals of all three predictors are differences of these (k_huffman_pack, PLANE), as long as every byte IS
                    // its value: a turn that is not plain looks at its eight values (bit 2 of the flags: the plane does not hold the tile).
                    const uint32_t p01 = __builtin_amdgcn_perm(D1[1], D1[0], 0x0c0c0400u), p23 = __builtin_amdgcn_perm(D1[3], D1[2], 0x0c0c0400u);
                    const uint32_t p45 = __builtin_amdgcn_perm(D1[5], D1[4], 0x0c0c0400u), p67 = __builtin_amdgcn_perm(D1[7], D1[6], 0x0c0c0400u);
                    GfU2 w;
                    w.x = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                    w.y = __builtin_amdgcn_perm(p67, p45, 0x05040100u);
                    *reinterpret_cast<GfU2 *>(a.plane + t * a.planeStride + i0) = w;
                }
                if (__all(widest <= 252u && lowest != (int32_t)0x80000000)) {
                    myFlags |= 2u;
#ifdef GF_ENC_HIST_AGG
                    // (experiment build, round 6: the round-5 review's wave-level aggregation of the hot bins -- the value the first lane
                    // holds is counted once for all the lanes that hold it when they are sixteen or more; measured in profiles/HISTORY.md)
                    auto addAgg = [&](uint32_t *h, uint32_t d) {
                        d &= 0xffu;
                        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                        const unsigned long long m = __ballot(d == d0);
                        const uint32_t n = (uint32_t)__popcll(m);
                        if (n >= 16u) {
                            if (lane == 0) atomicAdd(h + d0 * HR, n);
                            if (d != d0) atomicAdd(h + d * HR, 1u);
                        } else {
                            atomicAdd(h + d * HR, 1u);
                        }
                    };
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        addAgg(h0, D1[j]);
                        addAgg(h1, D2[j]);
                        if (triOk) addAgg(h2, D3[j]);
                    }
#else
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        atomicAdd(h0 + (D1[j] & 0xffu) * HR, 1u);
                        atomicAdd(h1 + (D2[j] & 0xffu) * HR, 1u);
                        if (triOk) atomicAdd(h2 + (D3[j] & 0xffu) * HR, 1u);
                    }
#endif
                } else {
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        const uint32_t v = Q.cur[j];
                        myFlags |= i0 + j < nCells ? ((v == GF_NULL_CODE) ? 1u : 2u) : 0u;
                        const uint32_t d1 = D1[j], d2 = D2[j], d3 = D3[j];
                        if (PLANE && d1 + 128u > 255u) myFlags |= 4u;
                        // the first M32 byte of the three residuals without a branch; the continuation bytes of values that have any
                        // behind ONE branch per cell (three, one per predictor, cost the scalar unit more than the cell cost the SIMDs)
                        bool s1, s2, s3 = true;
                        atomicAdd(h0 + m32_first_byte(d1, &s1) * HR, 1u);
                        atomicAdd(h1 + m32_first_byte(d2, &s2) * HR, 1u);
                        if (triOk) atomicAdd(h2 + m32_first_byte(d3, &s3) * HR, 1u);
                        if (!(s1 && s2 && s3)) {
                            // the continuation bytes of a wide value (CodecM32.java:283-311).  Two and three bytes -- what terrain has
                            // -- without a loop: |x| - 127 as one byte, |x| - 255 as two 7-bit groups; longer ones (rare) by the
                            // general form.  (Round 4: the general form alone, a loop per value around a switch per byte, made phase A
                            // of the rough batch 3.7 times the plain one.)
                            auto rest = [&](uint32_t *h, uint32_t x) -> uint32_t {
                                const uint32_t mag = (int32_t)x < 0 ? 0u - x : x;
                                if (mag <= 254u) {
                                    atomicAdd(h + (mag - 127u) * HR, 1u);
                                    return 2u;
                                }
                                if (mag <= 16638u) {
                                    const uint32_t d = mag - 255u;
                                    atomicAdd(h + (0x80u | (d >> 7)) * HR, 1u);
                                    atomicAdd(h + (d & 0x7fu) * HR, 1u);
                                    return 3u;
                                }
                                const uint32_t n = (uint32_t)gf_m32_len(x);
                                for (uint32_t k = 1; k < n; k++) atomicAdd(h + gf_m32_byte(x, (int)n, (int)k) * HR, 1u);
                                return n;
                            };
                            if (!s1) maxN1 = max(maxN1, rest(h0, d1));
                            if (!s2) maxN2 = max(maxN2, rest(h1, d2));
                            if (!s3) maxN3 = max(maxN3, rest(h2, d3));
                        }
                    }
                }
                c0 += cStep;
                if (c0 >= nC) c0 -= nC;
            }
        }
        if (myFlags) atomicOr(&P.flags, myFlags);
        if (maxN1 > 1) atomicMax(&P.maxN[0], maxN1);
        if (maxN2 > 1) atomicMax(&P.maxN[1], maxN2);
        if (maxN3 > 1) atomicMax(&P.maxN[2], maxN3);
        __syncthreads();
        const uint32_t flags = P.flags;
        GF_STAMP(1);
#ifdef GF_DIAG
        if (a.phaseLimit == 1) { __syncthreads(); continue; }
#endif
        const bool anyNull = flags & 1u, anyValid = flags & 2u;
        // cells that were counted as residual 0 above: the seed cell + the padding of the last step
        uint32_t forcedZeros = ((nCells + CPT - 1) / CPT) * CPT - nCells + 1u;

        if (!anyValid) {                         // CodecHuffman.java:80-82 -> null
            if (tid == 0) {
                a.lengths[t] = 0;
                a.status[t] = GF_K_DECLINED;
                if (a.predictors) a.predictors[t] = 0;
                if (PART == 1) (a.encStats + t * (size_t)GF_ENC_STAT_WORDS)[7] = 0u;     // nothing for k_huffman_trees
            }
            __syncthreads();
            continue;
        }
        if (!anyNull && nC < 2 && (a.predictorMask & 2)) {   // PredictorModelLinear.java:113 indexes values[1]: AIOOBE
            if (tid == 0) {
                a.lengths[t] = 0;
                a.status[t] = GF_K_ERR_BOUNDS;
                if (a.predictors) a.predictors[t] = 0;
                if (PART == 1) (a.encStats + t * (size_t)GF_ENC_STAT_WORDS)[7] = 0u;
            }
            __syncthreads();
            continue;
        }

#ifdef GF_ENC_NULLS_RETRY
        if constexpr (FAST) {
            if (anyNull) {
                if (tid == 0) {
                    a.status[t] = GF_K_RETRY;
                    atomicOr(a.retryFlag, 1u);
                }
                __syncthreads();
                continue;
            }
        }
#endif
        if (anyNull) {
            // ---- nulls path: seed (PredictorModelDifferencingWithNulls.java:79-105), then one histogram ----
            // A cell's residual is v - prior, prior = left neighbour (column 0: the first cell of the row above), and the seed in
            // prior's place where prior is null (cell (0,0) included): only those "start" cells need the seed, which is the rounded
            // mean of exactly their values (:79-96).  So one pass over the tile (eight cells per thread as in phase A) sums the
            // start cells and counts everything else -- null cells as the byte 0x80 -- and a second one adds the start cells.
            forcedZeros = 0;
            for (int i = tid; i < 3 * 256 * HR; i += ENC_THREADS) (&S.histR[0][0])[i] = 0;
            __syncthreads();
            uint32_t *const h0 = &S.histR[0][rep];
            auto addHist0 = [&](uint32_t x) -> uint32_t {
                bool single;
                const uint32_t b0 = m32_first_byte(x, &single);
                atomicAdd(h0 + b0 * HR, 1u);
                uint32_t n = 1;
                if (!single) {
                    n = (uint32_t)gf_m32_len(x);
                    for (uint32_t k = 1; k < n; k++) atomicAdd(h0 + gf_m32_byte(x, (int)n, (int)k) * HR, 1u);
                }
                return n;
            };
            long long mySum = 0;
            uint32_t myCnt = 0, maxN = 1;
            {
                uint32_t c0 = ((uint32_t)tid * CPT) % nC;
                const uint32_t cStep = STEP_CELLS % nC;
                for (uint32_t i0 = (uint32_t)tid * CPT; i0 < nCells; i0 += STEP_CELLS) {
                    Cells8 Q;
                    load_cells8(tile, nC, nCells, i0, Q);
                    uint32_t c = c0;
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        const uint32_t idx = i0 + j, v = Q.cur[j];
                        const uint32_t prior = c > 0 ? (j > 0 ? Q.cur[j - 1] : Q.wm1) : Q.up[j];
                        if (idx < nCells) {
                            if (v == GF_NULL_CODE) addHist0(GF_NULL_CODE);
                            else if (idx == 0 || prior == GF_NULL_CODE) { mySum += (int32_t)v; myCnt++; }
                            else maxN = max(maxN, addHist0(v - prior));
                        }
                        if (++c == nC) c = 0;
                    }
                    c0 += cStep;
                    if (c0 >= nC) c0 -= nC;
                }
            }
            if (myCnt) {
                atomicAdd(&P.sumStart, (unsigned long long)mySum);
                atomicAdd(&P.nStart, myCnt);
            }
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
                P.maxN[0] = 1;
            }
            __syncthreads();
            const uint32_t seed = P.seed;
            if (myCnt) {                                     // the threads that met start cells meet them again
                uint32_t c0 = ((uint32_t)tid * CPT) % nC;
                const uint32_t cStep = STEP_CELLS % nC;
                for (uint32_t i0 = (uint32_t)tid * CPT; i0 < nCells; i0 += STEP_CELLS) {
                    Cells8 Q;
                    load_cells8(tile, nC, nCells, i0, Q);
                    uint32_t c = c0;
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        const uint32_t idx = i0 + j, v = Q.cur[j];
                        const uint32_t prior = c > 0 ? (j > 0 ? Q.cur[j - 1] : Q.wm1) : Q.up[j];
                        if (idx < nCells && v != GF_NULL_CODE && (idx == 0 || prior == GF_NULL_CODE)) maxN = max(maxN, addHist0(v - seed));
                        if (++c == nC) c = 0;
                    }
                    c0 += cStep;
                    if (c0 >= nC) c0 -= nC;
                }
            }
            atomicMax(&P.maxN[0], maxN);
            __syncthreads();
        } else if (tid == 0) {
            P.seed = tile[0];
            P.model[0] = (a.predictorMask & 1) ? 1 : 0;
            P.model[1] = (a.predictorMask & 2) ? 2 : 0;
            P.model[2] = ((a.predictorMask & 4) && triOk) ? 3 : 0;
        }

        // reduce the replicas
        for (int i = tid; i < 3 * 256; i += ENC_THREADS) {
            const uint32_t *h = &S.histR[0][0] + (size_t)i * HR;
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < HR; k++) s += h[k];
            if ((i & 255) == 0 && s >= forcedZeros) s -= forcedZeros;
            if constexpr (PART == 1) (a.encStats + t * (size_t)GF_ENC_STAT_WORDS + 16)[i] = s;
            else enc_hist(P, i >> 8)[i & 255] = s;
        }
        if constexpr (PART == 1) {
            __syncthreads();                     // the models, the seed and the longest values are in place
            if (tid < 3) {
                uint32_t *stat = a.encStats + t * (size_t)GF_ENC_STAT_WORDS;
                stat[tid] = (uint32_t)P.model[tid];
                stat[4 + tid] = P.maxN[tid];
                if (tid == 0) {
                    stat[3] = P.seed;
                    stat[7] = 1u;
                    stat[8] = (PLANE && !(flags & 5u)) ? 1u : 0u;              // the byte plane holds this tile (no null cell, no wide row difference)
                }
            }
            __syncthreads();
            continue;
        } else {
        for (int i = tid; i < 3 * IMG_WORDS; i += ENC_THREADS) (&P.img[0][0])[i] = 0;
        __syncthreads();                         // histR dead from here: S.tree may be written

        GF_STAMP(2);
        // ---------------- phase B: the three Huffman trees, one wave each ----------------
        // Only the shortest packing is written (CodecHuffman.java:106-119), so a tree is built only for a predictor that can
        // still win.  No prefix code spends fewer bits on a text than its zero-order entropy (Shannon), the header and the
        // serialised tree are known from the number of distinct symbols: a LOWER BOUND of every candidate's packing comes
        // from its histogram alone (a logarithm per symbol).  The candidate with the smallest bound builds its tree first;
        // once its exact size is known the others build theirs only if their bound does not already lose against it (the
        // earlier predictor wins ties, :107).  On terrain the Triangle residuals undercut the others by a tenth of the
        // packing: two of the three trees -- a third of the kernel's instructions -- are not built.  (Fast kernel only.)
        int firstP = -1;
        if constexpr (FAST) {
            if (wave < 3 && P.model[wave] != 0) {
                const int p = wave;
                double sumCLogC = 0.0;
                uint32_t n = 0, N = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t cnt = enc_hist(P, p)[r * 64 + lane];
                    n += cnt != 0;
                    N += cnt;
                    if (cnt > 1) sumCLogC += (double)cnt * (double)__log2f((float)cnt);
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    n += gf_lane_xor(n, d);
                    N += gf_lane_xor(N, d);
                    sumCLogC += __shfl_xor(sumCLogC, d, 64);
                }
                if (lane == 0) {
                    // bits: 80 of header, the tree (n == 1: 17; else 8 + 10 n - 1), the text >= N log2 N - sum c log2 c, less a
                    // margin for the single-precision logarithms (2^-22 relative on sums of at most N log2 N) and the cast
                    double text = n > 1 ? (double)N * (double)__log2f((float)N) - sumCLogC : 0.0;
                    text -= 64.0 + (double)N * (1.0 / 4096.0);
                    const double bits = 80.0 + (n == 1 ? 17.0 : 8.0 + 10.0 * (double)n - 1.0) + (text > 0.0 ? text : 0.0);
                    P.lbBytes[p] = (uint32_t)(bits * 0.125);                // floor: a lower bound stays one
                }
            }
            __syncthreads();
            for (int p = 0; p < 3; p++)
                if (P.model[p] != 0 && (firstP < 0 || P.lbBytes[p] < P.lbBytes[firstP])) firstP = p;
        }
        for (int stage = 0; stage < (FAST ? 2 : 1); stage++) {
        bool mine = wave < 3 && P.model[wave] != 0;
        if constexpr (FAST) {
            mine = mine && (stage == 0 ? wave == firstP : wave != firstP);
            if (mine && stage == 1) {
                const uint64_t firstBytes = (P.totalBits[firstP] + 7) >> 3;
                const bool lost = wave < firstP ? (uint64_t)P.lbBytes[wave] > firstBytes : (uint64_t)P.lbBytes[wave] >= firstBytes;
                if (lost) {
                    if (lane == 0) P.model[wave] = 0;                         // not a candidate any more
                    mine = false;
                }
            }
        }
        if (mine) {
            const int p = wave;
            EncTree<FAST> &T = S.tree[p];
            int n = 0;
            uint32_t nM32 = 0;
            // B1  sort the used symbols by (count asc, symbol asc)
            if (FAST || 6ull * nCells < (1ull << 23)) {
                // counts < 2^23: one 32-bit key (count << 8 | symbol) per symbol, 256 keys in 4 registers
                // per lane (element e = r*64 + lane), bitonic network
                uint32_t key[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t cnt = enc_hist(P, p)[r * 64 + lane];
                    key[r] = cnt ? ((cnt << 8) | (uint32_t)(r * 64 + lane)) : 0xFFFFFFFFu;
                    n += __popcll(__ballot(cnt != 0));
                    nM32 += cnt;
                }
                wave_bitonic_sort<4>(key, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int e = r * 64 + lane;
                    if (e < n) {
                        T.cnt[e] = key[r] >> 8;
                        T.sym[e] = (uint8_t)(key[r] & 0xffu);
                        T.nl[e] = 1;
                        key[r] = ((key[r] >> 8) << 9) | (uint32_t)(256 + e);
                    } else {
                        key[r] = 0xFFFFFFFFu;
                    }
                }
                if (lane == 0) T.n = n;
                GF_STAMP_WAVE(3);
                // B2  tree by data-parallel rounds
                wave_huff_rounds(T, key, n, lane);
            } else if constexpr (!FAST) {
                // huge tiles: compaction + rank sort on full 32-bit counts, sequential merge on one lane
                uint32_t *ccnt = &T.cnt[255];    // compacted counts (temp, branch area is free until the merge)
                uint16_t *csym = T.bq;           // compacted symbols (temp)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int s = lane + 64 * j;
                    const uint32_t cnt = enc_hist(P, p)[s];
                    const unsigned long long m = __ballot(cnt != 0);
                    const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
                    if (cnt != 0) { ccnt[pos] = cnt; csym[pos] = (uint16_t)s; }
                    n += __popcll(m);
                    nM32 += cnt;
                }
                __builtin_amdgcn_wave_barrier();
                uint32_t rk[4], myc[4];
                uint16_t mys[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = lane + 64 * j;
                    rk[j] = 0;
                    if (i < n) {
                        const uint32_t ci = ccnt[i];
                        myc[j] = ci;
                        mys[j] = csym[i];
                        uint32_t rank = 0;
                        for (int q = 0; q < n; q++) {
                            const uint32_t cq = ccnt[q];
                            rank += (cq < ci || (cq == ci && q < i)) ? 1u : 0u;
                        }
                        rk[j] = rank;
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = lane + 64 * j;
                    if (i < n) { T.cnt[rk[j]] = myc[j]; T.sym[rk[j]] = (uint8_t)mys[j]; }
                }
                if (lane == 0) T.n = n;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0 && n > 1) gf_huff_merge_t<false>(T, n, true);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) nM32 += gf_lane_xor(nM32, d);
            __builtin_amdgcn_wave_barrier();
            GF_STAMP_WAVE(4);
            // B3  header image, codes, serialised tree, exact bit totals
            uint32_t *img = P.img[p];
            if (lane == 0) {
                P.nM32[p] = nM32;
                // header, CodecHuffman.java:121-130 (LSB-first bit store == little-endian bytes)
                const uint32_t seed = P.seed;
                img[0] = ((uint32_t)a.codecIndex & 0xffu) | ((uint32_t)P.model[p] << 8) | (seed << 16);
                img[1] = (seed >> 16) | (nM32 << 16);
                img[2] = (nM32 >> 16) | ((n > 1 ? (uint32_t)(n - 1) : 0u) << 16);
            }
            __builtin_amdgcn_wave_barrier();
            unsigned long long textBits = 0;
            uint32_t maxLen = 0;
            bool nullByte = false;                                // the byte 0x80 is among the symbols
            if (n == 1) {
                // uniform special case, HuffmanEncoder.java:147-157: 8 zero bits, a 1 bit, the symbol
                nullByte = T.sym[0] == 0x80u;
                if (lane == 0) {
                    const uint32_t rec = 1u | ((uint32_t)T.sym[0] << 1);     // 9 bits at bit 88
                    atomicOr(&img[2], rec << 24);
                    atomicOr(&img[3], rec >> 8);
                    P.tab[p][T.sym[0]] = 0;
                }
            } else {
                for (int i = lane; i < n; i += 64) {
                    uint64_t code;
                    uint32_t pos;
                    const int len = gf_huff_leaf_code(T, i, &code, &pos);
                    const uint32_t sym = T.sym[i];
                    nullByte = nullByte || sym == 0x80u;
                    P.tab[p][sym] = ((uint64_t)len << 56) | code;
                    textBits += (unsigned long long)T.cnt[i] * (unsigned)len;
                    maxLen = max(maxLen, (uint32_t)len);
                    const uint32_t bit = 88u + pos;                           // record: 1, then 8 symbol bits
                    const uint64_t rec = (uint64_t)(1u | (sym << 1)) << (bit & 31u);
                    atomicOr(&img[bit >> 5], (uint32_t)rec);
                    if (rec >> 32) atomicOr(&img[(bit >> 5) + 1], (uint32_t)(rec >> 32));
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                textBits += __shfl_xor(textBits, d, 64);
                maxLen = max(maxLen, gf_lane_xor(maxLen, d));
            }
            if (lane == 0) {
                const uint32_t treeBits = n == 1 ? 17u : (8u + 10u * (uint32_t)n - 1u);
                P.treeEndBit[p] = 80u + treeBits;
                P.totalBits[p] = 80ull + treeBits + textBits;
                P.maxLen[p] = maxLen;
            }
            {
                // a stream of plain bytes -- every value one M32 byte (-126..126), none of them the null code: the packer then takes a
                // residual's low byte for its M32 form (k_huffman_pack, PLAIN)
                const bool anyNullByte = __any(nullByte) != 0;
                if (lane == 0) P.plain[p] = (!anyNullByte && P.maxN[p] == 1u) ? 1u : 0u;
            }
        }
        __syncthreads();
        }                                        // stage
        GF_STAMP(5);
#ifdef GF_DIAG
        if (a.debug) {                           // dump of the on-chip state
            uint32_t *dbg = a.debug + t * (size_t)GF_ENC_DEBUG_WORDS;
            const uint32_t *pw = reinterpret_cast<const uint32_t *>(&P);
            const uint32_t *tw = reinterpret_cast<const uint32_t *>(&S.tree[0]);
            for (uint32_t i = tid; i < sizeof(EncPersist) / 4; i += ENC_THREADS) dbg[i] = pw[i];
            for (uint32_t i = tid; i < 3 * sizeof(GfHuffTree) / 4; i += ENC_THREADS)
                dbg[sizeof(EncPersist) / 4 + i] = tw[i];
            __syncthreads();
        }
        if (a.phaseLimit == 2) continue;
#endif

        // ---------------- phase C: pick the shortest, pack it ----------------
        int best = -1;
        uint64_t bestBytes = ~0ull;
        for (int p = 0; p < 3; p++) {
            if (P.model[p] == 0) continue;
            const uint64_t bytes = (P.totalBits[p] + 7) >> 3;
            if (bytes < bestBytes) { bestBytes = bytes; best = p; }          // strict: CodecHuffman.java:107
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
            // the pack phase runs as k_huffman_pack, with its own register budget and occupancy: leave it the selection
            uint32_t *rec = a.packRecs + t * (size_t)GF_PACK_REC_WORDS;
            if (tid == 0) {
                rec[0] = (uint32_t)model;
                rec[1] = P.treeEndBit[best];
                rec[2] = P.seed;
                rec[3] = P.maxN[best];
                rec[4] = P.maxLen[best];
                rec[5] = (uint32_t)min(P.totalBits[best] - P.treeEndBit[best], (uint64_t)0xFFFFFFFFu);   // bits of the text
                rec[7] = P.plain[best];
            }
            for (int i = tid; i < GF_IMG_WORDS; i += ENC_THREADS) rec[8 + i] = P.img[best][i];
            const uint32_t *tw = reinterpret_cast<const uint32_t *>(P.tab[best]);
            for (int i = tid; i < 512; i += ENC_THREADS) rec[8 + 88 + i] = tw[i];
            __syncthreads();
        }
        }                                        // PART != 1
    }
}


#ifndef GF_ENC_VARIANT
// ------------------------------------------------------------------------------------------------
// k_huffman_trees: phase B and the selection of k_huffman_encode, ONE WAVE PER TILE (round 4; see k_huffman_encode, PART 1).
// The same steps in the same order: a lower bound of every candidate's packing from its histogram, the tree of the candidate with
// the smallest bound, then the others' only if their bound does not already lose against its exact size (CodecHuffman.java:100-119;
// the earlier predictor wins ties); the shortest packing's record for k_huffman_pack.  The histograms are read from the statistics
// record as they are needed (four bins per lane); of the code tables and tree images only the best so far and the one being built
// are kept, and one tree's arrays serve every candidate: 8.6 KB of LDS per wave, seventeen waves per CU.
// ------------------------------------------------------------------------------------------------
// (a tree's links without the leaves' counts and symbols, which stay in the registers of the lanes that sorted them)
struct TreeLinks {
    uint16_t parent[511];
    uint16_t left[255];
    uint16_t nl[511];
    int n;
};
struct TreesShared {
    uint64_t tab[256];                  // the candidate's code table (scratch of the sort and the rounds before it is written)
    uint32_t img[IMG_WORDS];
    TreeLinks T;
};
static_assert(sizeof(TreesShared) <= 5120, "k_huffman_trees: 32 waves per CU need 5,120 bytes of LDS or less per wave");

__global__ __launch_bounds__(64) void k_huffman_trees(GfEncodeArgs a)
{
    __shared__ TreesShared S;
    const int lane = threadIdx.x;
    GF_FOR_TILES(t, a.nTiles, true) {
        const uint32_t *__restrict__ stat = a.encStats + t * (size_t)GF_ENC_STAT_WORDS;
        if (stat[7] == 0u) continue;                                     // declined or refused by k_huffman_encode
        int model[3];
        uint32_t maxN[3];
#pragma unroll
        for (int p = 0; p < 3; p++) { model[p] = (int)stat[p]; maxN[p] = stat[4 + p]; }
        const uint32_t seed = stat[3];
        // lower bounds (see k_huffman_encode, phase B)
        uint32_t lbBytes[3] = {0, 0, 0};
        int firstP = -1;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            if (model[p] == 0) continue;                                  // wave-uniform
            double sumCLogC = 0.0;
            uint32_t n = 0, N = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t cnt = stat[16 + p * 256 + r * 64 + lane];
                n += cnt != 0;
                N += cnt;
                if (cnt > 1) sumCLogC += (double)cnt * (double)__log2f((float)cnt);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                n += gf_lane_xor(n, d);
                N += gf_lane_xor(N, d);
                sumCLogC += __shfl_xor(sumCLogC, d, 64);
            }
            double text = n > 1 ? (double)N * (double)__log2f((float)N) - sumCLogC : 0.0;
            text -= 64.0 + (double)N * (1.0 / 4096.0);
            const double bits = 80.0 + (n == 1 ? 17.0 : 8.0 + 10.0 * (double)n - 1.0) + (text > 0.0 ? text : 0.0);
            // (every lane holds the same sums; lane 0's value as in the one-kernel form)
            lbBytes[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits * 0.125));
            if (firstP < 0 || lbBytes[p] < lbBytes[firstP]) firstP = p;
        }
        // the candidates: firstP, then the others in predictor order; the best so far lies in the tile's record
        uint32_t *rec = a.packRecs + t * (size_t)GF_PACK_REC_WORDS;
        int best = -1;
        uint64_t bestBytes = ~0ull, firstBytes = 0;
        uint32_t bTreeEnd = 0, bMaxLen = 0, bPlain = 0;
        uint64_t bTotalBits = 0;
        for (int k = 0; k < 3 && firstP >= 0; k++) {
            // k = 0: firstP; k = 1, 2: the other two in order
            int p = firstP;
            if (k > 0) {
                p = k - 1;
                if (p >= firstP) p++;
                if (model[p] == 0) continue;
                const bool lost = p < firstP ? (uint64_t)lbBytes[p] > firstBytes : (uint64_t)lbBytes[p] >= firstBytes;
                if (lost) continue;
            }
            TreeLinks &T = S.T;
            uint64_t *tab = S.tab;
            uint32_t *img = S.img;
            for (int i = lane; i < IMG_WORDS; i += 64) img[i] = 0;
            int n = 0;
            uint32_t nM32 = 0;
            // B1  sort the used symbols by (count asc, symbol asc)
            uint32_t key[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t cnt = stat[16 + p * 256 + r * 64 + lane];
                key[r] = cnt ? ((cnt << 8) | (uint32_t)(r * 64 + lane)) : 0xFFFFFFFFu;
                n += __popcll(__ballot(cnt != 0));
                nM32 += cnt;
            }
            uint32_t *scr = reinterpret_cast<uint32_t *>(tab);            // (the code table is written in B3)
            if (n <= 128) {
                // the used bins moved together first (terrain uses 70 to 100 of the 256): a network over 128 keys instead of 256
                uint32_t at = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned long long m = __ballot(key[r] != 0xFFFFFFFFu);
                    if (key[r] != 0xFFFFFFFFu) scr[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key[r];
                    at += (uint32_t)__popcll(m);
                }
                key[0] = lane < n ? scr[lane] : 0xFFFFFFFFu;
                key[1] = 64 + lane < n ? scr[64 + lane] : 0xFFFFFFFFu;
                key[2] = 0xFFFFFFFFu;
                key[3] = 0xFFFFFFFFu;
                wave_bitonic_sort<2>(key, lane);
            } else {
                wave_bitonic_sort<4>(key, lane);
            }
            // leaf e = r * 64 + lane of the sorted order: its count and symbol stay with this lane
            uint32_t leafCnt[4], leafSym[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int e = r * 64 + lane;
                leafCnt[r] = key[r] >> 8;
                leafSym[r] = key[r] & 0xffu;
                if (e < n) {
                    T.nl[e] = 1;
                    key[r] = ((key[r] >> 8) << 9) | (uint32_t)(256 + e);
                } else {
                    key[r] = 0xFFFFFFFFu;
                }
            }
            if (lane == 0) T.n = n;
            // B2  tree by data-parallel rounds (small rounds merged by rank through the same scratch)
            wave_huff_rounds(T, key, n, lane, scr);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) nM32 += gf_lane_xor(nM32, d);
            __builtin_amdgcn_wave_barrier();
            // B3  header image, codes, serialised tree, exact bit totals
            if (lane == 0) {
                // header, CodecHuffman.java:121-130 (LSB-first bit store == little-endian bytes)
                img[0] = ((uint32_t)a.codecIndex & 0xffu) | ((uint32_t)model[p] << 8) | (seed << 16);
                img[1] = (seed >> 16) | (nM32 << 16);
                img[2] = (nM32 >> 16) | ((n > 1 ? (uint32_t)(n - 1) : 0u) << 16);
            }
            __builtin_amdgcn_wave_barrier();
            unsigned long long textBits = 0;
            uint32_t maxLen = 0;
            bool nullByte = false;                                // the byte 0x80 is among the symbols
            if (n == 1) {
                // uniform special case, HuffmanEncoder.java:147-157: 8 zero bits, a 1 bit, the symbol
                const uint32_t sym0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)leafSym[0]);
                nullByte = sym0 == 0x80u;
                if (lane == 0) {
                    const uint32_t r9 = 1u | (sym0 << 1);                     // 9 bits at bit 88
                    atomicOr(&img[2], r9 << 24);
                    atomicOr(&img[3], r9 >> 8);
                    tab[sym0] = 0;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = r * 64 + lane;
                    if (i < n) {
                        uint64_t code;
                        uint32_t pos;
                        const int len = gf_huff_leaf_code(T, i, &code, &pos);
                        const uint32_t sym = leafSym[r];
                        nullByte = nullByte || sym == 0x80u;
                        tab[sym] = ((uint64_t)len << 56) | code;
                        textBits += (unsigned long long)leafCnt[r] * (unsigned)len;
                        maxLen = max(maxLen, (uint32_t)len);
                        const uint32_t bit = 88u + pos;                       // record: 1, then 8 symbol bits
                        const uint64_t r9 = (uint64_t)(1u | (sym << 1)) << (bit & 31u);
                        atomicOr(&img[bit >> 5], (uint32_t)r9);
                        if (r9 >> 32) atomicOr(&img[(bit >> 5) + 1], (uint32_t)(r9 >> 32));
                    }
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                textBits += __shfl_xor(textBits, d, 64);
                maxLen = max(maxLen, gf_lane_xor(maxLen, d));
            }
            const uint32_t treeBits = n == 1 ? 17u : (8u + 10u * (uint32_t)n - 1u);
            const uint64_t totalBits = 80ull + treeBits + textBits;
            const uint64_t bytes = (totalBits + 7) >> 3;
            const bool anyNullByte = __any(nullByte) != 0;
            if (k == 0) firstBytes = bytes;
            __builtin_amdgcn_wave_barrier();
            // strictly shorter wins, the earlier predictor wins a tie (CodecHuffman.java:107; the candidates do not come in order)
            if (best < 0 || bytes < bestBytes || (bytes == bestBytes && p < best)) {
                best = p;
                bestBytes = bytes;
                bTreeEnd = 80u + treeBits;
                bTotalBits = totalBits;
                bMaxLen = maxLen;
                bPlain = (!anyNullByte && maxN[p] == 1u) ? 1u : 0u;
                // its tree image and code table to the record (the code table holds what the scratch left where no symbol lies:
                // the packer looks up symbols of the tile only)
                for (int i = lane; i < GF_IMG_WORDS; i += 64) rec[8 + i] = img[i];
                const uint32_t *tw = reinterpret_cast<const uint32_t *>(tab);
                for (int i = lane; i < 512; i += 64) rec[8 + 88 + i] = tw[i];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (best < 0) {
            if (lane == 0) {
                a.lengths[t] = 0;
                a.status[t] = GF_K_DECLINED;
                if (a.predictors) a.predictors[t] = 0;
            }
            continue;
        }
        if (lane == 0) {
            a.lengths[t] = (uint32_t)min(bestBytes, (uint64_t)0xffffffffu);
            a.status[t] = bestBytes > a.slotStride ? GF_K_OVERFLOW : GF_K_OK;
            if (a.predictors) a.predictors[t] = (uint8_t)model[best];
            rec[0] = (uint32_t)model[best];
            rec[1] = bTreeEnd;
            rec[2] = seed;
            rec[3] = maxN[best];
            rec[4] = bMaxLen;
            rec[5] = (uint32_t)min(bTotalBits - bTreeEnd, (uint64_t)0xFFFFFFFFu);   // bits of the text
            rec[7] = bPlain | (stat[8] ? 2u : 0u);                           // bit 1: the byte plane holds the tile (k_huffman_encode, PART 1)
        }
    }
}
#endif  // GF_ENC_VARIANT

// the flat scan of a tile through the wave-private windows, in as many cell ranges as its bit count asks for (a wave's
// share of a range must fit its quarter of the window); false: a range did not fit after all -- the tile is left to
// k_huffman_pack_rare, which packs it again from its first bit
template <int MODEL, bool PLAIN>
__device__ __forceinline__ bool pack_flat_ranges(const uint32_t *__restrict__ tile, uint32_t nC, uint32_t nCells, uint32_t seed,
                                                 const uint64_t *tab, uint32_t *win, uint32_t *__restrict__ out32, uint32_t *waveSum,
                                                 PackState &ps, uint32_t slotWords, uint32_t textBits)
{
    const uint32_t capBits = WAVE_WIN_BITS;
    const uint64_t want = ((uint64_t)textBits + (textBits >> 2)) / ENC_WAVES;           // a wave's share, with 25 % slack
    const uint32_t nRanges = (uint32_t)min((uint64_t)1024, want / capBits + 1u);
    constexpr uint32_t UNIT = CPT * ENC_WAVES;
    uint32_t per = (((nCells + nRanges - 1) / nRanges) + UNIT - 1) / UNIT * UNIT;
    // A range whose bits are spread unevenly over the waves' shares (a cliff in one quarter of it) overruns a window although the
    // range as a whole would fit: nothing was written then, and the range is packed again in two halves (round 4; such tiles -- most
    // of a rough surface's -- used to go to k_huffman_pack_rare, 0.2 ms per launch of the rough batch).
    for (uint32_t b = 0; b < nCells;) {
        const PackStateOk r = pack_flat_waves<MODEL, PLAIN>(tile, nC, nCells, seed, tab, win, out32, waveSum, ps, slotWords, b, b + per);
        if (r.ok) {
            ps = r.ps;
            b += per;
        } else {
            if (per <= 8u * UNIT) return false;
            per = (per / 2u + UNIT - 1u) / UNIT * UNIT;
        }
    }
    return true;
}

// ... and the same for a plain stream from the byte plane: the head of the stream rides in front of the first range
template <int MODEL>
__device__ __forceinline__ bool pack_plane_ranges(const uint8_t *__restrict__ plane, uint32_t nC, uint32_t nCells, const uint64_t *tab, uint32_t *win,
                                                  uint32_t *__restrict__ out32, uint32_t *waveSum, PackState &ps, uint32_t slotWords,
                                                  uint32_t textBits, uint32_t headElems)
{
    const uint32_t capBits = WAVE_WIN_BITS;
    const uint64_t want = ((uint64_t)textBits + (textBits >> 2)) / ENC_WAVES;           // a wave's share, with 25 % slack
    const uint32_t nRanges = (uint32_t)min((uint64_t)1024, want / capBits + 1u);
    constexpr uint32_t UNIT = CPT * ENC_WAVES;
    uint32_t per = (((nCells + nRanges - 1) / nRanges) + UNIT - 1) / UNIT * UNIT;
    for (uint32_t b = 0; b < nCells;) {
        const PackStateOk r = pack_plane_waves<MODEL>(plane, nC, nCells, tab, win, out32, waveSum, ps, slotWords, b, b + per, b == 0u ? headElems : 0u);
        if (r.ok) {
            ps = r.ps;
            b += per;
        } else {
            if (per <= 8u * UNIT) return false;
            per = (per / 2u + UNIT - 1u) / UNIT * UNIT;
        }
    }
    return true;
}

// k_huffman_pack: phase C of the encoder as its own kernel (see GfEncodeArgs::packRecs)
struct PackShared {
    uint64_t tab[256];
    uint32_t waveSum[ENC_WAVES];
};

// The packer proper has no calls and no stack: tiles whose codes do not fit the wave windows (a value of more than
// (WIN_WORDS - 2) * 32 / STEP_CELLS bits, a single column, a range that overran its window after all) are marked in word 6 of
// their record and packed by k_huffman_pack_rare behind it, through the general scans (pack_generic / pack_flat: calls with
// frames, 128 bytes of scratch per lane -- which the packer of every tile used to carry for them).
template <bool RARE>
__device__ __forceinline__ void huffman_pack_tiles(const GfEncodeArgs &a, PackShared &P, uint32_t *win)
{
    const int tid = threadIdx.x;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    GF_FOR_TILES(t, a.nTiles, !RARE) {                                    // the packer: no tile loop (see gvrs_kernels.h)
        // (round 6) everything the tile's record holds is asked for at once, the tile's status with it: as `status, then the record's
        // scalars, then as many words of the tree image as they say` these were three round trips to memory one behind the other --
        // 13 K of a tile's 58 K cycles in the packer (tools/phase_cycles_pack.py)
        uint32_t *rec = a.packRecs + t * (size_t)GF_PACK_REC_WORDS;
        constexpr int IMG_PER = (GF_IMG_WORDS + ENC_THREADS - 1) / ENC_THREADS, TAB_PER = (512 + ENC_THREADS - 1) / ENC_THREADS;
        const int32_t tileStatus = a.status[t];
        uint32_t rv[8], imgv[IMG_PER], tabv[TAB_PER];
#pragma unroll
        for (int k = 0; k < 8; k++) rv[k] = rec[k];
#pragma unroll
        for (int k = 0; k < IMG_PER; k++) imgv[k] = rec[8 + min(tid + k * ENC_THREADS, GF_IMG_WORDS - 1)];
#pragma unroll
        for (int k = 0; k < TAB_PER; k++) tabv[k] = rec[8 + 88 + min(tid + k * ENC_THREADS, 511)];
        if (tileStatus != GF_K_OK) continue;                              // declined, overflow: nothing to write
        if (RARE && rv[6] != 1u) continue;                                // (the same word in every thread)
        const uint32_t *__restrict__ tile = reinterpret_cast<const uint32_t *>(a.values) + t * (size_t)nCells;
        uint32_t *__restrict__ out32 = reinterpret_cast<uint32_t *>(a.out + t * a.slotStride);
        if (!RARE) GF_STAMP(6);
        const int model = (int)rv[0];
        const uint32_t treeEnd = rv[1], seed = rv[2], maxN = rv[3], maxLen = rv[4], textBits = rv[5], recFlags = rv[7];
        const uint32_t imgWords = (treeEnd + 31u) >> 5;
        for (int i = tid + IMG_PER * ENC_THREADS; i < WIN_WORDS + WIN_SLACK; i += ENC_THREADS) win[i] = 0u;
#pragma unroll
        for (int k = 0; k < IMG_PER; k++) {
            const int i = tid + k * ENC_THREADS;
            win[i] = i < (int)imgWords ? imgv[k] : 0u;                    // (IMG_PER * ENC_THREADS <= WIN_WORDS)
        }
        {
            // (0x80 is no symbol of a plain stream -- bit 0 of the record's flags --: the plane path gives that byte to the cells it does not emit)
            uint32_t *tw = reinterpret_cast<uint32_t *>(P.tab);
            const bool emptyNull = !RARE && (recFlags & 3u) == 3u;
#pragma unroll
            for (int k = 0; k < TAB_PER; k++) {
                const int i = tid + k * ENC_THREADS;
                if (i < 512) tw[i] = (emptyNull && (i >> 1) == 0x80) ? 0u : tabv[k];
            }
        }
        __syncthreads();
        if (!RARE) GF_STAMP(8);

        const uint32_t nStream = gf_stream_len(model, nR, nC);
        const uint64_t *tab = P.tab;
        const uint32_t elemMaxBits = max(1u, maxN * maxLen);
        const uint32_t slotWords = (uint32_t)(a.slotStride >> 2);
        PackState ps;
        ps.bitBase = treeEnd;
        ps.wordBase = 0;
        window_flush(win, out32, ps);            // header + tree image
        if (!RARE) GF_STAMP(9);
        bool done = true;
        if (nStream > 0 && maxLen > 0) {
            const bool fast = (uint64_t)STEP_CELLS * elemMaxBits <= (uint64_t)(WIN_WORDS - 2) * 32u && nC >= 2;
            if constexpr (RARE) {
                if (!fast) {
                    ps = pack_generic(model, tile, nR, nC, seed, tab, elemMaxBits, 0u, nStream, win, out32, P.waveSum, ps);
                } else if (model == 1) {
                    ps = pack_flat<1>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps);
                } else if (model == 2) {
                    ps = pack_head<2>(tile, nR, nC, seed, tab, 2u * nR - 1u, win, out32, P.waveSum, ps);
                    ps = pack_flat<2>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps);
                } else if (model == 3) {
                    ps = pack_head<3>(tile, nR, nC, seed, tab, nC - 1u + nR - 1u, win, out32, P.waveSum, ps);
                    ps = pack_flat<3>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps);
                } else {
                    ps = pack_flat<4>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps);
                }
            } else {
                const bool plain = (recFlags & 1u) != 0u && model != 4;       // (the same word in every thread)
                // (round 6) a plain stream of a tile whose byte plane phase A has left: the plane instead of the tile -- a byte per cell
                if (!fast) {
                    done = false;
                } else if (plain && a.plane && (recFlags & 2u) && nC >= (uint32_t)CPT) {
                    const uint8_t *__restrict__ plane = a.plane + t * a.planeStride;
                    if (model == 1) done = pack_plane_ranges<1>(plane, nC, nCells, tab, win, out32, P.waveSum, ps, slotWords, textBits, 0u);
                    else if (model == 2) done = pack_plane_ranges<2>(plane, nC, nCells, tab, win, out32, P.waveSum, ps, slotWords, textBits, 2u * nR - 1u);
                    else done = pack_plane_ranges<3>(plane, nC, nCells, tab, win, out32, P.waveSum, ps, slotWords, textBits, nC - 1u + nR - 1u);
                    GF_STAMP(10);
                } else if (model == 1) {
                    done = plain ? pack_flat_ranges<1, true>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, slotWords, textBits)
                                 : pack_flat_ranges<1, false>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, slotWords, textBits);
                } else if (model == 2) {
                    ps = pack_head<2>(tile, nR, nC, seed, tab, 2u * nR - 1u, win, out32, P.waveSum, ps);
                    done = plain ? pack_flat_ranges<2, true>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, slotWords, textBits)
                                 : pack_flat_ranges<2, false>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, slotWords, textBits);
                } else if (model == 3) {
                    ps = pack_head<3>(tile, nR, nC, seed, tab, nC - 1u + nR - 1u, win, out32, P.waveSum, ps);
                    done = plain ? pack_flat_ranges<3, true>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, slotWords, textBits)
                                 : pack_flat_ranges<3, false>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, slotWords, textBits);
                } else {
                    done = pack_flat_ranges<4, false>(tile, nC, nCells, seed, tab, win, out32, P.waveSum, ps, slotWords, textBits);
                }
            }
        }
        if (!RARE) {
            GF_STAMP(7);
            if (tid == 0) {
                rec[6] = done ? 0u : 1u;
                if (!done && a.retryFlag) atomicAdd(a.retryFlag + 1, 1u);  // (word 1: tiles left to k_huffman_pack_rare)
                if (!done && a.lean) a.status[t] = GF_K_RETRY;              // (no k_huffman_pack_rare behind this launch: the caller's business)
            }
        }
        if (done) {
            const uint32_t remBits = ps.bitBase - ps.wordBase * 32u;
            const uint32_t remWords = (remBits + 31u) >> 5;
            for (uint32_t j = tid; j < remWords; j += ENC_THREADS)
                if (ps.wordBase + j < slotWords) out32[ps.wordBase + j] = win[j];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(ENC_THREADS, ENC_PACK_WGS) void k_huffman_pack(GfEncodeArgs a)
{
    __shared__ PackShared P;
    __shared__ __attribute__((aligned(16))) uint32_t win[WIN_WORDS + WIN_SLACK];
    huffman_pack_tiles<false>(a, P, win);
}

#ifndef GF_ENC_VARIANT
__global__ __launch_bounds__(ENC_THREADS, 4) void k_huffman_pack_rare(GfEncodeArgs a)
{
    __shared__ PackShared P;
    __shared__ __attribute__((aligned(16))) uint32_t win[WIN_WORDS + WIN_SLACK];
    if (a.retryFlag && a.retryFlag[1] == 0u) return;                      // the packer finished every tile
    huffman_pack_tiles<true>(a, P, win);
}


// ------------------------------------------------------------------------------------------------
// k_m32_streams: the predictor -> CodecM32 stage on its own, for the codecs whose entropy stage is the host's zlib
// (CodecDeflate.java:157-199: every applicable predictor's M32 stream is deflated, the shortest packing wins).
// An M32 stream is what the packers above produce with the identity code table (every byte its own 8-bit code).
// ------------------------------------------------------------------------------------------------
struct M32Shared {
    uint64_t tab[256];
    alignas(16) uint32_t win[WIN_WORDS + WIN_SLACK];
    uint32_t waveSum[ENC_WAVES];
    uint32_t flags, seed, nStart;
    uint32_t nBytes[3];         // exact M32 bytes of the candidates
    unsigned long long sumStart;
};

__global__ __launch_bounds__(ENC_THREADS, 4) void k_m32_streams(GfM32Args a)
{
    __shared__ M32Shared S;
    const int tid = threadIdx.x;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    S.tab[tid] = (8ull << 56) | (uint64_t)tid;

    GF_FOR_WG_TILE(t, a.nTiles) {                                         // no tile loop: see gvrs_kernels.h
        const uint32_t *__restrict__ tile = reinterpret_cast<const uint32_t *>(a.values) + t * (size_t)nCells;
        if (tid == 0) { S.flags = 0; S.sumStart = 0; S.nStart = 0; }
        __syncthreads();
        uint32_t myFlags = 0;
        for (uint32_t i = tid; i < nCells; i += ENC_THREADS) myFlags |= tile[i] == GF_NULL_CODE ? 1u : 2u;
        if (myFlags) atomicOr(&S.flags, myFlags);
        __syncthreads();
        const bool anyNull = S.flags & 1u, anyValid = S.flags & 2u;
        const bool triOk = nR >= 2 && nC >= 2;
        int32_t early = 99;
        if (!anyValid) early = GF_K_DECLINED;                               // CodecDeflate.java:168-170 -> null
        else if (!anyNull && nC < 2) early = GF_K_ERR_BOUNDS;               // PredictorModelLinear indexes values[1]
        if (early != 99) {
            if (tid < 3) { a.lengths[t * 3 + tid] = 0; a.models[t * 3 + tid] = 0; }
            if (tid == 0) { a.status[t] = early; a.seeds[t] = 0; }
            __syncthreads();
            continue;
        }
        if (anyNull) {
            // seed of the nulls predictor (PredictorModelDifferencingWithNulls.java:79-105)
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
                atomicAdd(&S.sumStart, (unsigned long long)mySum);
                atomicAdd(&S.nStart, myCnt);
            }
            __syncthreads();
            if (tid == 0) {
                const double avg = (double)(long long)S.sumStart / (double)S.nStart;
                const double f = floor(avg + 0.5);
                int32_t sd;
                if (f >= 2147483647.0) sd = 2147483647;
                else if (f <= -2147483648.0) sd = (int32_t)0x80000000;
                else sd = (int32_t)f;
                S.seed = (uint32_t)sd;
            }
        } else if (tid == 0) {
            S.seed = tile[0];
        }
        if (tid < 3) S.nBytes[tid] = 0;
        __syncthreads();
        const uint32_t seed = S.seed;
        // exact stream lengths first (a candidate longer than its sub-slot is reported, not written)
        {
            uint32_t n1 = 0, n2 = 0, n3 = 0;
            if (anyNull) {
                for (uint32_t idx = tid; idx < nCells; idx += ENC_THREADS) {
                    const uint32_t r = idx / nC, c = idx - r * nC;
                    n1 += (uint32_t)gf_m32_len(cell_residual(4, tile, nC, idx, r, c, seed));
                }
            } else {
                uint32_t c0 = ((uint32_t)tid * CPT) % nC;
                const uint32_t cStep = STEP_CELLS % nC;
                for (uint32_t i0 = (uint32_t)tid * CPT; i0 < nCells; i0 += STEP_CELLS) {
                    Cells8 Q;
                    load_cells8(tile, nC, nCells, i0, Q);
                    uint32_t c = c0;
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        const uint32_t idx = i0 + j;
                        if (idx > 0 && idx < nCells) {                           // every cell but the seed has one residual per model
                            const uint32_t v = Q.cur[j];
                            const uint32_t W = j > 0 ? Q.cur[j - 1] : Q.wm1;
                            const uint32_t WW = j > 1 ? Q.cur[j - 2] : (j == 1 ? Q.wm1 : Q.wm2);
                            const uint32_t N = Q.up[j];
                            const uint32_t NW = j > 0 ? Q.up[j - 1] : Q.upm1;
                            const uint32_t d = v - (c > 0 ? W : N);
                            n1 += (uint32_t)gf_m32_len(d);
                            n2 += (uint32_t)gf_m32_len(c >= 2 ? v - (2u * W - WW) : d);
                            n3 += (uint32_t)gf_m32_len((idx >= nC && c > 0) ? v - (W + N - NW) : d);
                        }
                        if (++c == nC) c = 0;
                    }
                    c0 += cStep;
                    if (c0 >= nC) c0 -= nC;
                }
            }
            if (n1) atomicAdd(&S.nBytes[0], n1);
            if (n2) atomicAdd(&S.nBytes[1], n2);
            if (n3) atomicAdd(&S.nBytes[2], n3);
        }
        __syncthreads();
        bool overflow = false;
        for (int p = 0; p < 3; p++) {
            const int model = anyNull ? (p == 0 ? 4 : 0) : (p == 2 ? (triOk ? 3 : 0) : p + 1);
            if (model == 0) {
                if (tid == 0) { a.lengths[t * 3 + p] = 0; a.models[t * 3 + p] = 0; }
                continue;
            }
            if (S.nBytes[p] + 8u > a.subStride) {                            // does not fit (the flushes write whole words)
                overflow = true;
                if (tid == 0) { a.lengths[t * 3 + p] = S.nBytes[p]; a.models[t * 3 + p] = (uint8_t)model; }
                continue;
            }
            uint32_t *__restrict__ out32 = reinterpret_cast<uint32_t *>(a.out + (t * 3 + (size_t)p) * a.subStride);
            for (int i = tid; i < WIN_WORDS + WIN_SLACK; i += ENC_THREADS) S.win[i] = 0;
            __syncthreads();
            PackState ps;
            ps.bitBase = 0;
            ps.wordBase = 0;
            const uint32_t nStream = gf_stream_len(model, nR, nC);
            const uint32_t emb = 48;                                        // six bytes at most per value
            if (nStream > 0) {
                if (nC < 2) {
                    ps = pack_generic(model, tile, nR, nC, seed, S.tab, emb, 0u, nStream, S.win, out32, S.waveSum, ps);
                } else if (model == 1) {
                    ps = pack_flat<1>(tile, nC, nCells, seed, S.tab, S.win, out32, S.waveSum, ps);
                } else if (model == 2) {
                    ps = pack_generic(2, tile, nR, nC, seed, S.tab, emb, 0u, 2u * nR - 1u, S.win, out32, S.waveSum, ps);
                    ps = pack_flat<2>(tile, nC, nCells, seed, S.tab, S.win, out32, S.waveSum, ps);
                } else if (model == 3) {
                    ps = pack_generic(3, tile, nR, nC, seed, S.tab, emb, 0u, nC - 1u + nR - 1u, S.win, out32, S.waveSum, ps);
                    ps = pack_flat<3>(tile, nC, nCells, seed, S.tab, S.win, out32, S.waveSum, ps);
                } else {
                    ps = pack_flat<4>(tile, nC, nCells, seed, S.tab, S.win, out32, S.waveSum, ps);
                }
            }
            {
                const uint32_t remBits = ps.bitBase - ps.wordBase * 32u;
                const uint32_t remWords = (remBits + 31u) >> 5;
                for (uint32_t j = tid; j < remWords; j += ENC_THREADS)
                    out32[ps.wordBase + j] = S.win[j];
            }
            const uint32_t nBytes = ps.bitBase >> 3;
            if (tid == 0) { a.lengths[t * 3 + p] = nBytes; a.models[t * 3 + p] = (uint8_t)model; }
            __syncthreads();
        }
        if (tid == 0) { a.status[t] = overflow ? GF_K_OVERFLOW : GF_K_OK; a.seeds[t] = seed; }
        __syncthreads();
    }
}

#endif  // GF_ENC_VARIANT

}  // namespace

#ifdef GF_ENC_VARIANT
// The one-tile-per-call build (-DGF_ENC_THREADS=1024 -DGF_ENC_VARIANT, round 4): a workgroup alone on the chip is a chain of
// latencies -- phase A a round trip to memory per turn, the packer likewise --, and sixteen waves take a 120x150 tile in three turns
// instead of nine (84 -> 76 us per call).  Only the two kernels a tile usually needs; what they leave behind keeps GF_K_RETRY
// (GfEncodeArgs::lean).
hipError_t gf_launch_huffman_encode_lean_t1024(const GfEncodeArgs &a, hipStream_t stream)
{
    if (a.nTiles == 0) return hipSuccess;
    const size_t nCells = (size_t)a.nRows * (size_t)a.nCols;
    if (!a.packRecs || !a.retryFlag || !a.lean || 6ull * nCells >= (1ull << 23)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_huffman_encode<true>, gf_tile_grid(a.nTiles), dim3(ENC_THREADS), 0, stream, a);
    hipLaunchKernelGGL(k_huffman_pack, gf_tile_grid(a.nTiles), dim3(ENC_THREADS), 0, stream, a);
    return hipGetLastError();
}
#else
hipError_t gf_launch_huffman_encode(const GfEncodeArgs &a, hipStream_t stream)
{
    if (a.nTiles == 0) return hipSuccess;
    if (!a.packRecs) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(a.nTiles < 65536 * 16 ? a.nTiles : 65536 * 16);
    const size_t nCells = (size_t)a.nRows * (size_t)a.nCols;
    // word 0: tiles for k_huffman_encode<false> (experiment builds only), word 1: for k_huffman_pack_rare -- zeroed by the fast
    // kernel's first workgroup, or here where that kernel does not run or sets word 0 itself
#ifndef GF_ENC_NULLS_RETRY
    if (a.retryFlag && !a.lean && 6ull * nCells >= (1ull << 23)) {
#else
    if (a.retryFlag && !a.lean) {
#endif
        const hipError_t e = hipMemsetAsync(a.retryFlag, 0, 8, stream);
        if (e != hipSuccess) return e;
    }
    if (a.retryFlag && 6ull * nCells < (1ull << 23)) {
#ifdef GF_DIAG
        static const bool diagSplit = getenv("GF_DIAG_SPLIT") != nullptr;      // the three-kernel form with the packer's stamps (tools/phase_cycles_pack.py)
        if (diagSplit && a.encStats && !a.lean) {
#else
        if (a.encStats && !a.lean) {
#endif
            // the histograms, then the trees with a wave per tile (the diagnostic flavour keeps the one-kernel form its stamps describe)
            if (a.plane) hipLaunchKernelGGL((k_huffman_encode<true, 1, true>), gf_tile_grid(a.nTiles), dim3(ENC_THREADS), 0, stream, a);
            else hipLaunchKernelGGL((k_huffman_encode<true, 1>), gf_tile_grid(a.nTiles), dim3(ENC_THREADS), 0, stream, a);
            hipLaunchKernelGGL(k_huffman_trees, gf_tile_grid(a.nTiles), dim3(64), 0, stream, a);
        } else
        hipLaunchKernelGGL(k_huffman_encode<true>, gf_tile_grid(a.nTiles), dim3(ENC_THREADS), 0, stream, a);
#ifdef GF_ENC_NULLS_RETRY
        // (only this experiment build's fast kernel leaves tiles behind: the shipping one takes every tile of up to 2^23 / 6 cells, and
        // the general kernel's launch -- 4-5 us to find nothing to do -- went in round 4)
        hipLaunchKernelGGL(k_huffman_encode<false>, dim3(grid < 2048 ? grid : 2048), dim3(ENC_THREADS), 0, stream, a);
#endif
    } else {
        GfEncodeArgs g = a;
        g.retryFlag = nullptr;
        hipLaunchKernelGGL(k_huffman_encode<false>, dim3(grid), dim3(ENC_THREADS), 0, stream, g);
    }
    // (round 6: the packer with a wave or two per tile -- a 64- and a 128-thread build of this file, 4 KB of window per wave -- was
    // measured: 0.192 / 0.176 ms against 0.176 with four waves on the bench batch, 0.42 / 0.36 against 0.32 on 200x200 tiles)
    hipLaunchKernelGGL(k_huffman_pack, gf_tile_grid(a.nTiles), dim3(ENC_THREADS), 0, stream, a);
    if (!a.lean) hipLaunchKernelGGL(k_huffman_pack_rare, dim3(grid < 1024 ? grid : 1024), dim3(ENC_THREADS), 0, stream, a);
    return hipGetLastError();
}

hipError_t gf_launch_m32_streams(const GfM32Args &a, hipStream_t stream)
{
    if (a.nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_m32_streams, gf_tile_grid(a.nTiles), dim3(ENC_THREADS), 0, stream, a);
    return hipGetLastError();
}
#endif  // GF_ENC_VARIANT
