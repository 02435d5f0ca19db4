// gvrs_encode.hip -- CodecHuffman.encode for a batch of tiles, one workgroup per tile.
//
// Replaces (reference, core/src/main/java/org/gridfour/):
//   compress/CodecHuffman.java:70-130          null scan, 3 predictor passes, keep shortest
//   compress/PredictorModel{Differencing,Linear,Triangle,DifferencingWithNulls}.java encode
//   compress/CodecM32.java:257-311             signed varint
//   compress/HuffmanEncoder.java:124-305       histogram, tree, serialisation, text
//   io/BitOutputStore.java:205-288             LSB-first bit order
//
// Phases of a workgroup (256 threads = 4 waves) on one tile
//   A  one pass over the tile: residuals of all three predictors per cell, M32 byte
//      lengths, three 256-bin histograms in LDS (replicated 8x to spread atomics)
//   B  waves 0..2 build one Huffman tree each (rank sort of the used symbols, O(1)-pop
//      merge of huff_build.h on lane 0, per-leaf code + pre-order position in parallel),
//      giving code tables, the serialised header+tree image and the exact bit total
//   C  the shortest candidate wins (ties: D, L, T order, CodecHuffman.java:107); its
//      residuals are recomputed in stream order (tile re-read hits L2), code lengths are
//      prefix-summed across the workgroup and each thread ORs its bits into an LDS window
//      that is flushed to the tile's output slot with coalesced dword stores.
// HBM traffic per cell: 4 B read + c B written; everything else stays on chip.

#include <hip/hip_runtime.h>

#include "gvrs_kernels.h"
#include "gvrs_encode_layout.h"

namespace {

constexpr int ENC_THREADS = GF_ENC_THREADS;
constexpr int ENC_WAVES = GF_ENC_WAVES;
constexpr int HIST_R = 8;                       // histogram replicas
constexpr int IMG_WORDS = GF_IMG_WORDS;
constexpr int WIN_WORDS = 2048;                 // bit-pack window (8 KB)
constexpr int WIN_SLACK = 8;

union EncScratch {
    uint32_t histR[3][256 * HIST_R];            // phase A
    GfHuffTree tree[3];                         // phase B
    uint32_t win[WIN_WORDS + WIN_SLACK];        // phase C
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan over the workgroup; *total = sum over all threads
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *waveSum, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) waveSum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < ENC_WAVES; w++) {
        uint32_t s = waveSum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// residual of one cell for a model; idx = r*nC + c
__device__ __forceinline__ uint32_t cell_residual(int model, const uint32_t *__restrict__ tile, uint32_t nC,
                                                  uint32_t idx, uint32_t r, uint32_t c, uint32_t seed)
{
    const uint32_t v = tile[idx];
    switch (model) {
    case 1: return v - (c > 0 ? tile[idx - 1] : tile[idx - nC]);
    case 2:
        if (c >= 2) return v - (2u * tile[idx - 1] - tile[idx - 2]);
        return v - (c == 1 ? tile[idx - 1] : tile[idx - nC]);
    case 3:
        if (r == 0) return v - tile[idx - 1];
        if (c == 0) return v - tile[idx - nC];
        return v - (tile[idx - 1] + tile[idx - nC] - tile[idx - nC - 1]);
    default: {
        // PredictorModelDifferencingWithNulls.java:109-131: prior = left neighbour, or the
        // first cell of the previous row at a row start; the seed replaces a null prior.
        if (v == GF_NULL_CODE) return GF_NULL_CODE;
        uint32_t prior;
        if (c > 0) prior = tile[idx - 1];
        else prior = r > 0 ? tile[idx - nC] : GF_NULL_CODE;
        if (prior == GF_NULL_CODE) prior = seed;
        return v - prior;
    }
    }
}

// Bitonic sort of 256 32-bit keys held 4 per lane (element e = r*64 + lane), ascending.
__device__ __forceinline__ void wave_bitonic_sort256(uint32_t (&k)[4], int lane)
{
#pragma unroll
    for (int size = 2; size <= 256; size <<= 1) {
#pragma unroll
        for (int j = size >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int rj = j >> 6;           // partner lives in another register of the same lane
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if ((r & rj) == 0) {
                        const int r2 = r | rj;
                        const bool asc = ((r * 64) & size) == 0;
                        const uint32_t lo = min(k[r], k[r2]), hi = max(k[r], k[r2]);
                        k[r] = asc ? lo : hi;
                        k[r2] = asc ? hi : lo;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t other = (uint32_t)__shfl_xor((int)k[r], j, 64);
                    const bool asc = (((r * 64) | lane) & size) == 0;
                    const bool lower = (lane & j) == 0;
                    k[r] = (lower == asc) ? min(k[r], other) : max(k[r], other);
                }
            }
        }
    }
}

// diagnostic: cycle stamps per phase (only when a debug buffer is attached)
#define GF_STAMP(i)                                                                       \
    do {                                                                                  \
        if (a.debug && tid == 0)                                                          \
            (a.debug + t * (size_t)GF_ENC_DEBUG_WORDS + GF_ENC_DEBUG_WORDS - 16)[i] =    \
                (uint32_t)__builtin_amdgcn_s_memtime();                                   \
    } while (0)

struct BitSink {
    uint32_t *win;
    uint64_t acc;
    uint32_t nacc;      // valid bits in acc (< 32 between puts)
    uint32_t wp;        // window word the low bits of acc belong to
    bool first;         // the first word is shared with the previous thread

    __device__ __forceinline__ void init(uint32_t *w, uint32_t bitPos)
    {
        win = w;
        wp = bitPos >> 5;
        nacc = bitPos & 31u;
        acc = 0;
        first = true;
    }
    __device__ __forceinline__ void flushWord()
    {
        uint32_t lo = (uint32_t)acc;
        if (first) { atomicOr(&win[wp], lo); first = false; }
        else win[wp] = lo;
        wp++;
        acc >>= 32;
        nacc -= 32;
    }
    // len <= 32
    __device__ __forceinline__ void put32(uint32_t code, uint32_t len)
    {
        acc |= (uint64_t)code << nacc;
        nacc += len;
        if (nacc >= 32) flushWord();
    }
    __device__ __forceinline__ void put(uint64_t code, uint32_t len)
    {
        if (len > 32) {
            put32((uint32_t)code, 32);
            put32((uint32_t)(code >> 32), len - 32);
        } else {
            put32((uint32_t)code, len);
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (nacc > 0) atomicOr(&win[wp], (uint32_t)acc);
    }
};

__global__ __launch_bounds__(ENC_THREADS) void k_huffman_encode(GfEncodeArgs a)
{
    __shared__ EncPersist P;
    __shared__ EncScratch S;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;

    for (size_t t = blockIdx.x; t < a.nTiles; t += gridDim.x) {
        const uint32_t *__restrict__ tile = reinterpret_cast<const uint32_t *>(a.values) + t * (size_t)nCells;
        uint32_t *__restrict__ out32 = reinterpret_cast<uint32_t *>(a.out + t * a.slotStride);

        GF_STAMP(0);
        // ---------------- phase A: null scan + three histograms ----------------
        for (int i = tid; i < 3 * 256 * HIST_R; i += ENC_THREADS) (&S.histR[0][0])[i] = 0;
        if (tid == 0) { P.flags = 0; P.sumStart = 0; P.nStart = 0; }
        if (tid < 3) { P.maxN[tid] = 0; P.model[tid] = 0; P.nM32[tid] = 0; }
        __syncthreads();

        const bool triOk = nR >= 2 && nC >= 2;
        const uint32_t rep = (uint32_t)lane & (HIST_R - 1);
        uint32_t myFlags = 0, maxN1 = 0, maxN2 = 0, maxN3 = 0;
        {
            // Each thread takes quads of 4 consecutive cells (flat index), two quads per iteration, and
            // issues every load of the iteration before using any of them: 16-byte loads of the cells and
            // of the row above (4-byte aligned only), plus the three halo words.
            auto addHist = [&](int p, uint32_t d) -> uint32_t {
                const int n = gf_m32_len(d);
                if (n == 1) {
                    atomicAdd(&S.histR[p][gf_m32_byte(d, 1, 0) * HIST_R + rep], 1u);
                } else {
                    for (int k = 0; k < n; k++) atomicAdd(&S.histR[p][gf_m32_byte(d, n, k) * HIST_R + rep], 1u);
                }
                return (uint32_t)n;
            };
            struct Quad {
                uint32_t cur[4], up[4], wm1, wm2, upm1;
            };
            auto loadQuad = [&](uint32_t i0, Quad &Q) {
                if (i0 + 3 < nCells) {
                    const GfU4 v = *reinterpret_cast<const GfU4 *>(tile + i0);
                    Q.cur[0] = v.x; Q.cur[1] = v.y; Q.cur[2] = v.z; Q.cur[3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) Q.cur[j] = i0 + j < nCells ? tile[i0 + j] : 0u;
                }
                Q.wm1 = i0 >= 1 ? tile[i0 - 1] : 0u;
                Q.wm2 = i0 >= 2 ? tile[i0 - 2] : 0u;
                if (i0 >= nC && i0 + 3 < nCells) {
                    const GfU4 v = *reinterpret_cast<const GfU4 *>(tile + (i0 - nC));
                    Q.up[0] = v.x; Q.up[1] = v.y; Q.up[2] = v.z; Q.up[3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) Q.up[j] = (i0 + j >= nC && i0 + j < nCells) ? tile[i0 + j - nC] : 0u;
                }
                Q.upm1 = i0 >= nC + 1 ? tile[i0 - nC - 1] : 0u;
            };
            auto doQuad = [&](uint32_t i0, const Quad &Q) {
                uint32_t r = i0 / nC, c = i0 - r * nC;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t idx = i0 + j;
                    if (idx < nCells) {
                        const uint32_t v = Q.cur[j];
                        myFlags |= (v == GF_NULL_CODE) ? 1u : 2u;
                        if (idx > 0) {
                            const uint32_t W = j > 0 ? Q.cur[j - 1] : Q.wm1;
                            const uint32_t WW = j > 1 ? Q.cur[j - 2] : (j == 1 ? Q.wm1 : Q.wm2);
                            const uint32_t N = Q.up[j];
                            const uint32_t NW = j > 0 ? Q.up[j - 1] : Q.upm1;
                            maxN1 = max(maxN1, addHist(0, gf_res_differencing(r, c, v, W, N)));
                            maxN2 = max(maxN2, addHist(1, gf_res_linear(r, c, v, W, WW, N)));
                            if (triOk) maxN3 = max(maxN3, addHist(2, gf_res_triangle(r, c, v, W, N, NW)));
                        }
                    }
                    if (++c >= nC) { c = 0; r++; }
                }
            };
            const uint32_t nQuads = (nCells + 3) >> 2;
            for (uint32_t q = tid; q < nQuads; q += 2 * ENC_THREADS) {
                Quad Q0, Q1;
                const uint32_t q1 = q + ENC_THREADS;
                loadQuad(q << 2, Q0);
                if (q1 < nQuads) loadQuad(q1 << 2, Q1);
                doQuad(q << 2, Q0);
                if (q1 < nQuads) doQuad(q1 << 2, Q1);
            }
        }
        if (myFlags) atomicOr(&P.flags, myFlags);
        atomicMax(&P.maxN[0], maxN1);
        atomicMax(&P.maxN[1], maxN2);
        atomicMax(&P.maxN[2], maxN3);
        __syncthreads();
        const uint32_t flags = P.flags;
        GF_STAMP(1);
        if (a.phaseLimit == 1) { __syncthreads(); continue; }
        const bool anyNull = flags & 1u, anyValid = flags & 2u;

        if (!anyValid) {                         // CodecHuffman.java:80-82 -> null
            if (tid == 0) {
                a.lengths[t] = 0;
                a.status[t] = GF_K_DECLINED;
                if (a.predictors) a.predictors[t] = 0;
            }
            __syncthreads();
            continue;
        }
        if (!anyNull && nC < 2 && (a.predictorMask & 2)) {   // PredictorModelLinear.java:113 indexes values[1]: AIOOBE
            if (tid == 0) {
                a.lengths[t] = 0;
                a.status[t] = GF_K_ERR_BOUNDS;
                if (a.predictors) a.predictors[t] = 0;
            }
            __syncthreads();
            continue;
        }

        if (anyNull) {
            // ---- nulls path: seed (PredictorModelDifferencingWithNulls.java:79-105), then one histogram ----
            for (int i = tid; i < 3 * 256 * HIST_R; i += ENC_THREADS) (&S.histR[0][0])[i] = 0;
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
                P.maxN[0] = 0;
            }
            __syncthreads();
            const uint32_t seed = P.seed;
            uint32_t maxN = 0;
            for (uint32_t idx = tid; idx < nCells; idx += ENC_THREADS) {
                const uint32_t r = idx / nC, c = idx - r * nC;
                const uint32_t x = cell_residual(4, tile, nC, idx, r, c, seed);
                const int n = gf_m32_len(x);
                maxN = max(maxN, (uint32_t)n);
                for (int k = 0; k < n; k++) atomicAdd(&S.histR[0][gf_m32_byte(x, n, k) * HIST_R + rep], 1u);
            }
            atomicMax(&P.maxN[0], maxN);
            __syncthreads();
        } else if (tid == 0) {
            P.seed = tile[0];
            P.model[0] = (a.predictorMask & 1) ? 1 : 0;
            P.model[1] = (a.predictorMask & 2) ? 2 : 0;
            P.model[2] = ((a.predictorMask & 4) && triOk) ? 3 : 0;
        }

        // reduce the replicas (registers first: hist aliases nothing, histR is read-only here)
        for (int i = tid; i < 3 * 256; i += ENC_THREADS) {
            const uint32_t *h = &S.histR[0][0] + (size_t)i * HIST_R;
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < HIST_R; k++) s += h[k];
            P.hist[i >> 8][i & 255] = s;
        }
        for (int i = tid; i < 3 * IMG_WORDS; i += ENC_THREADS) (&P.img[0][0])[i] = 0;
        __syncthreads();                         // histR dead from here: S.tree may be written

        GF_STAMP(2);
        // ---------------- phase B: the three Huffman trees ----------------
        // B1  waves 0..2: sort the used symbols of predictor p by (count asc, symbol asc)
        if (wave < 3 && P.model[wave] != 0) {
            const int p = wave;
            GfHuffTree &T = S.tree[p];
            int n = 0;
            uint32_t nM32 = 0;
            if (6ull * nCells < (1ull << 24)) {
                // counts < 2^24: one 32-bit key (count << 8 | symbol) per symbol, 256 keys in 4 registers
                // per lane (element e = r*64 + lane), bitonic network: 36 compare-exchange steps
                uint32_t key[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t cnt = P.hist[p][r * 64 + lane];
                    key[r] = cnt ? ((cnt << 8) | (uint32_t)(r * 64 + lane)) : 0xFFFFFFFFu;
                    n += __popcll(__ballot(cnt != 0));
                    nM32 += cnt;
                }
                wave_bitonic_sort256(key, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int e = r * 64 + lane;
                    if (e < n) { T.cnt[e] = key[r] >> 8; T.sym[e] = (uint8_t)(key[r] & 0xffu); }
                }
            } else {
                // huge tiles: compaction + rank sort on full 32-bit counts
                uint32_t *ccnt = &T.cnt[255];    // compacted counts (temp, branch area is free until the merge)
                uint16_t *csym = T.bq;           // compacted symbols (temp)
                for (int j = 0; j < 4; j++) {
                    const int s = lane + 64 * j;
                    const uint32_t cnt = P.hist[p][s];
                    const unsigned long long m = __ballot(cnt != 0);
                    const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
                    if (cnt != 0) { ccnt[pos] = cnt; csym[pos] = (uint16_t)s; }
                    n += __popcll(m);
                    nM32 += cnt;
                }
                __builtin_amdgcn_wave_barrier();
                uint32_t rk[4], myc[4];
                uint16_t mys[4];
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
                for (int j = 0; j < 4; j++) {
                    const int i = lane + 64 * j;
                    if (i < n) { T.cnt[rk[j]] = myc[j]; T.sym[rk[j]] = (uint8_t)mys[j]; }
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) nM32 += __shfl_xor(nM32, d, 64);
            if (lane == 0) {
                T.n = n;
                P.nM32[p] = nM32;
            }
        }
        __syncthreads();
        GF_STAMP(3);
        // B2  SIMT over trees: lane p of wave 0 runs the sequential merge of predictor p (per-lane LDS
        //     addressing); a scalar-unit formulation would be bound by the CU's single SALU
        if (wave == 0 && lane < 3 && P.model[lane] != 0) {
            GfHuffTree &T = S.tree[lane];
            const int n = T.n;
            if (n > 1) gf_huff_merge_t<false>(T, n, true);
        }
        __syncthreads();
        GF_STAMP(4);
        // B3  waves 0..2: header image, codes, serialised tree, exact bit totals
        if (wave < 3 && P.model[wave] != 0) {
            const int p = wave;
            GfHuffTree &T = S.tree[p];
            const int n = T.n;
            const uint32_t nM32 = P.nM32[p];
            uint32_t *img = P.img[p];
            if (lane == 0) {
                // header, CodecHuffman.java:121-130 (LSB-first bit store == little-endian bytes)
                const uint32_t seed = P.seed;
                img[0] = ((uint32_t)a.codecIndex & 0xffu) | ((uint32_t)P.model[p] << 8) | (seed << 16);
                img[1] = (seed >> 16) | (nM32 << 16);
                img[2] = (nM32 >> 16) | ((n > 1 ? (uint32_t)(n - 1) : 0u) << 16);
            }
            __builtin_amdgcn_wave_barrier();
            unsigned long long textBits = 0;
            uint32_t maxLen = 0;
            if (n == 1) {
                // uniform special case, HuffmanEncoder.java:147-157: 8 zero bits, a 1 bit, the symbol
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
                maxLen = max(maxLen, (uint32_t)__shfl_xor(maxLen, d, 64));
            }
            if (lane == 0) {
                const uint32_t treeBits = n == 1 ? 17u : (8u + 10u * (uint32_t)n - 1u);
                P.treeEndBit[p] = 80u + treeBits;
                P.totalBits[p] = 80ull + treeBits + textBits;
                P.maxLen[p] = maxLen;
            }
        }
        __syncthreads();
        GF_STAMP(5);
        if (a.debug) {                           // diagnostic dump of the on-chip state (tests/tools only)
            uint32_t *dbg = a.debug + t * (size_t)GF_ENC_DEBUG_WORDS;
            const uint32_t *pw = reinterpret_cast<const uint32_t *>(&P);
            const uint32_t *tw = reinterpret_cast<const uint32_t *>(&S.tree[0]);
            for (uint32_t i = tid; i < sizeof(EncPersist) / 4; i += ENC_THREADS) dbg[i] = pw[i];
            for (uint32_t i = tid; i < 3 * sizeof(GfHuffTree) / 4; i += ENC_THREADS)
                dbg[sizeof(EncPersist) / 4 + i] = tw[i];
        }
        __syncthreads();                         // trees dead from here: S.win may be written
        if (a.phaseLimit == 2) continue;

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

        const uint32_t treeEnd = P.treeEndBit[best];
        const uint32_t imgWords = (treeEnd + 31u) >> 5;
        for (int i = tid; i < WIN_WORDS + WIN_SLACK; i += ENC_THREADS) S.win[i] = i < (int)imgWords ? P.img[best][i] : 0u;
        __syncthreads();

        const uint32_t nStream = gf_stream_len(model, nR, nC);
        const uint32_t seed = P.seed;
        const uint64_t *__restrict__ tab = P.tab[best];
        // elements per chunk such that a chunk can never overflow the window
        const uint32_t elemMaxBits = max(1u, P.maxN[best] * P.maxLen[best]);
        uint32_t E = 4;
        while (E > 1 && (uint64_t)ENC_THREADS * E * elemMaxBits > (uint64_t)(WIN_WORDS - 2) * 32u) E >>= 1;
        uint32_t active = ENC_THREADS;
        if ((uint64_t)ENC_THREADS * elemMaxBits > (uint64_t)(WIN_WORDS - 2) * 32u)
            active = max(1u, (uint32_t)(((uint64_t)(WIN_WORDS - 2) * 32u) / elemMaxBits));
        const uint32_t chunkElems = active * E;

        uint32_t bitBase = treeEnd;              // next free bit of the packing (absolute)
        uint32_t wordBase = 0;                   // words already flushed to global
        // moves the completed words of the window to the output slot and slides the window
        auto flush = [&]() {
            const uint32_t fullWords = (bitBase >> 5) - wordBase;
            for (uint32_t j = tid; j < fullWords; j += ENC_THREADS) out32[wordBase + j] = S.win[j];
            const uint32_t partial = S.win[fullWords];
            __syncthreads();
            for (uint32_t j = tid; j <= fullWords; j += ENC_THREADS) S.win[j] = 0;
            __syncthreads();
            if (tid == 0) S.win[0] = partial;
            wordBase += fullWords;
            __syncthreads();
        };
        GF_STAMP(6);
        flush();                                 // header + tree image
        for (uint32_t chunk = 0; chunk < nStream; chunk += chunkElems) {
            // pass 1: residuals + their bit counts
            uint32_t xs[4];
            uint32_t myBits = 0;
            const uint32_t s0 = chunk + (uint32_t)tid * E;
            const uint32_t sEnd = min(nStream, chunk + chunkElems);
#pragma unroll
            for (uint32_t e = 0; e < 4; e++) {
                xs[e] = 0;
                const uint32_t s = s0 + e;
                if (e < E && (uint32_t)tid < active && s < sEnd) {
                    const uint32_t idx = gf_stream_cell(model, nR, nC, s);
                    const uint32_t r = idx / nC, c = idx - r * nC;
                    const uint32_t x = cell_residual(model, tile, nC, idx, r, c, seed);
                    xs[e] = x;
                    const int n = gf_m32_len(x);
                    for (int k = 0; k < n; k++) myBits += (uint32_t)(tab[gf_m32_byte(x, n, k)] >> 56);
                }
            }
            uint32_t total;
            const uint32_t excl = block_excl_scan(myBits, P.waveSum, &total);
            // pass 2: emit
            if (myBits) {
                BitSink sink;
                sink.init(S.win, bitBase + excl - wordBase * 32u);
#pragma unroll
                for (uint32_t e = 0; e < 4; e++) {
                    const uint32_t s = s0 + e;
                    if (e < E && (uint32_t)tid < active && s < sEnd) {
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
            bitBase += total;
            flush();
        }
        GF_STAMP(7);
        // tail: whatever is left in the window (also covers nStream == 0 / uniform tiles)
        {
            const uint32_t remBits = bitBase - wordBase * 32u;
            const uint32_t remWords = (remBits + 31u) >> 5;
            const uint32_t slotWords = (uint32_t)(a.slotStride >> 2);
            for (uint32_t j = tid; j < remWords; j += ENC_THREADS)
                if (wordBase + j < slotWords) out32[wordBase + j] = S.win[j];
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t gf_launch_huffman_encode(const GfEncodeArgs &a, hipStream_t stream)
{
    if (a.nTiles == 0) return hipSuccess;
    const unsigned grid = (unsigned)(a.nTiles < 65536 * 16 ? a.nTiles : 65536 * 16);
    hipLaunchKernelGGL(k_huffman_encode, dim3(grid), dim3(ENC_THREADS), 0, stream, a);
    return hipGetLastError();
}
