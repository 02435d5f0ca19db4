// gvrs_decode.hip -- CodecHuffman.decode for a batch of tiles, one workgroup per tile.
//
// Replaces (reference, core/src/main/java/org/gridfour/):
//   compress/CodecHuffman.java:133-169        header, Huffman decode, predictor decode
//   compress/HuffmanDecoder.java:65-187       tree parse, bit-serial symbol decode
//   compress/CodecM32.java:327-356            varint decode
//   compress/PredictorModel*.java decode      running sums
//   io/BitInputStore.java:112-210             LSB-first bit order
//
// The format has no synchronisation points, so both variable-length layers (Huffman
// codes over bits, M32 values over bytes) are parsed with the same self-synchronising
// scheme: the stream is cut into fixed-size subsequences, every thread parses its
// subsequence from a guessed start, then start positions are corrected from the
// predecessor's end position until nothing changes (codes resynchronise after a few
// symbols, so this takes 2-3 rounds; the worst case is still correct, just serial).
// A prefix sum of the per-subsequence symbol counts then tells every thread where its
// output goes.
//
// Phases of a workgroup (256 threads) on one tile
//   0  header + tree parse (thread 0, from an LDS copy of the first 344 bytes),
//      11-bit decode LUT built by all threads
//   1  Huffman text -> M32 bytes (LDS, or the per-workgroup global spill buffer for
//      tiles whose M32 stream exceeds the LDS budget)
//   2  M32 bytes -> residuals, scattered to their cells of the output tile
//   3  predictor inverse in place: int32 wrap-around prefix sums (column 0 chain, then
//      row scans; Linear = double scan, Triangle = column scans then row scans)

#include <hip/hip_runtime.h>

#include "gvrs_kernels.h"
#include "huff_build.h"

namespace {

constexpr int DEC_THREADS = 256;
constexpr int DEC_WAVES = DEC_THREADS / 64;
constexpr int LUT_BITS = 11;
constexpr int MAXQ = 512;                      // subsequences per chain
constexpr int HEAD_BYTES = 344;                // 10 header + 1 + ceil(2559/8) tree bytes, rounded up

struct DecShared {
    uint16_t lut[1 << LUT_BITS];               // short: (len << 8) | sym ; long: 0x8000 | node
    uint16_t child0[512];                      // 0xFFFF marks a leaf
    uint16_t child1[512];
    uint8_t leafSym[512];
    uint8_t childCount[512];
    uint16_t stack[260];
    uint32_t qs[MAXQ];                         // subsequence start
    uint32_t qe[MAXQ];                         // subsequence end (start of the next one)
    uint32_t qn[MAXQ];                         // symbols in the subsequence, later exclusive prefix
    uint8_t qdirty[MAXQ];
    uint32_t head[HEAD_BYTES / 4 + 2];         // +2 words of slack for the 64-bit window reads
    uint32_t waveSum[DEC_WAVES];
    uint32_t textStart;                        // bit offset of the Huffman text in the packing
    int32_t parseStatus;
    int32_t uniformSym;                        // >= 0: single-symbol encoding
    uint32_t chainEnd;                         // position after the last needed symbol
    uint32_t chainTotal;
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

// 64 bits of the blob starting at absolute bit position `bit` (LSB-first); reads beyond
// the buffer return zeros
__device__ __forceinline__ uint64_t peek64(const uint32_t *__restrict__ w32, uint64_t nWords, uint64_t bit)
{
    const uint64_t wi = bit >> 5;
    const uint32_t sh = (uint32_t)bit & 31u;
    const uint32_t w0 = wi < nWords ? w32[wi] : 0u;
    const uint32_t w1 = wi + 1 < nWords ? w32[wi + 1] : 0u;
    const uint32_t w2 = wi + 2 < nWords ? w32[wi + 2] : 0u;
    uint64_t lo = ((uint64_t)w1 << 32) | w0;
    lo >>= sh;
    if (sh) lo |= (uint64_t)w2 << (64u - sh);
    return lo;
}

struct HuffStep {
    const uint32_t *w32;
    uint64_t nWords;
    uint64_t base;                 // absolute bit position of packing bit 0
    const DecShared *S;
    // decodes the symbol at packing bit `pos`; returns the position of the next symbol
    __device__ __forceinline__ uint32_t operator()(uint32_t pos, uint32_t *sym) const
    {
        uint64_t w = peek64(w32, nWords, base + pos);
        const uint32_t e = S->lut[(uint32_t)w & ((1u << LUT_BITS) - 1u)];
        if (!(e & 0x8000u)) {
            *sym = e & 0xffu;
            return pos + (e >> 8);
        }
        uint32_t node = e & 0x7fffu;
        uint32_t d = LUT_BITS;
        uint32_t p = pos;
        while (S->child0[node] != 0xFFFFu) {
            const uint32_t bit = (uint32_t)(w >> d) & 1u;
            node = bit ? S->child1[node] : S->child0[node];
            if (++d == 64) {
                p += 64;
                d = 0;
                w = peek64(w32, nWords, base + p);
            }
        }
        *sym = S->leafSym[node];
        return p + d;
    }
};

struct M32Step {
    const uint8_t *m;
    uint32_t n;                    // bytes available
    // length of the value at byte `pos` (CodecM32.java:327-356); returns next position
    __device__ __forceinline__ uint32_t operator()(uint32_t pos, uint32_t *val) const
    {
        const uint32_t b0 = m[pos];
        uint32_t p = pos + 1;
        if (b0 != 0x7fu && b0 != 0x81u) {
            *val = b0 == 0x80u ? GF_NULL_CODE : (uint32_t)(int32_t)(int8_t)b0;
            return p;
        }
        uint32_t delta = 0;
        const uint32_t base[5] = {127u, 255u, 16639u, 2113791u, 270549247u};
        uint32_t v = 0;
        bool done = false;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (!done) {
                const uint32_t smp = p < n ? m[p] : 0u;
                p++;
                delta = (delta << 7) | (smp & 0x7fu);
                if (!(smp & 0x80u)) {
                    v = b0 == 0x81u ? (0u - delta - base[i]) : (delta + base[i]);
                    done = true;
                }
            }
        }
        *val = done ? v : delta;
        return p;
    }
};

// Self-synchronising parse of [start, end) cut into Q subsequences of S units.  On return
// qs[q] = true start of subsequence q, qn[q] = EXCLUSIVE prefix of the symbol counts,
// S.chainTotal = number of symbols that start before `end`.
template <class Step>
__device__ void resolve_chain(DecShared &S, const Step &step, uint32_t start, uint32_t end, uint32_t unit, uint32_t Q)
{
    const int tid = threadIdx.x;
    for (uint32_t q = tid; q < Q; q += DEC_THREADS) {
        S.qs[q] = start + q * unit;
        S.qdirty[q] = 1;
    }
    __syncthreads();
    for (;;) {
        for (uint32_t q = tid; q < Q; q += DEC_THREADS) {
            if (S.qdirty[q]) {
                uint32_t pos = S.qs[q], cnt = 0, dummy;
                const uint32_t limit = min(end, start + (q + 1) * unit);
                while (pos < limit) {
                    pos = step(pos, &dummy);
                    cnt++;
                }
                S.qe[q] = pos;
                S.qn[q] = cnt;
                S.qdirty[q] = 0;
            }
        }
        __syncthreads();
        int changed = 0;
        for (uint32_t q = tid; q < Q; q += DEC_THREADS) {
            if (q > 0) {
                const uint32_t ns = S.qe[q - 1];
                if (ns != S.qs[q]) {
                    S.qs[q] = ns;
                    S.qdirty[q] = 1;
                    changed = 1;
                }
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    // exclusive prefix sum of qn over q (Q <= MAXQ = 2 per thread)
    const uint32_t per = (Q + DEC_THREADS - 1) / DEC_THREADS;
    uint32_t local[MAXQ / DEC_THREADS];
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t q = tid * per + j;
        local[j] = q < Q ? S.qn[q] : 0u;
        sum += local[j];
    }
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t incl = wave_incl_scan(sum, lane);
    if (lane == 63) S.waveSum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < DEC_WAVES; w++) {
        if (w < wave) base += S.waveSum[w];
        tot += S.waveSum[w];
    }
    uint32_t run = base + incl - sum;
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t q = tid * per + j;
        if (q < Q) S.qn[q] = run;
        run += local[j];
    }
    if (tid == 0) S.chainTotal = tot;
    __syncthreads();
}

// wave-wide inclusive scan of one row segment with carry; returns the new carry
__device__ __forceinline__ uint32_t row_scan_segment(uint32_t x, uint32_t carry, int lane, uint32_t *outv)
{
    const uint32_t incl = wave_incl_scan(x, lane) + carry;
    *outv = incl;
    return __shfl(incl, 63, 64);
}

__global__ __launch_bounds__(DEC_THREADS) void k_huffman_decode(GfDecodeArgs a)
{
    __shared__ DecShared S;
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsM32[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const uint32_t *__restrict__ w32 = reinterpret_cast<const uint32_t *>(a.blob);
    const uint64_t nWords = (a.blobBytes + 3) >> 2;

    for (size_t t = blockIdx.x; t < a.nTiles; t += gridDim.x) {
        const uint64_t off = a.offsets ? a.offsets[t] : (uint64_t)t * a.slotStride;
        const uint32_t len = a.lengths[t];
        uint32_t *o = reinterpret_cast<uint32_t *>(a.values) + t * (size_t)nCells;
        const uint8_t *__restrict__ pk = a.blob + off;

        if (len < 10 || off + len > a.blobBytes) {       // BitInputStore would run out / AIOOBE on the header
            if (tid == 0) a.status[t] = GF_K_ERR_BOUNDS;
            __syncthreads();
            continue;
        }

        // ---------------- phase 0: header + tree ----------------
        {
            uint8_t *hb = reinterpret_cast<uint8_t *>(S.head);
            const uint32_t nh = min(len, (uint32_t)HEAD_BYTES);
            for (uint32_t i = tid; i < HEAD_BYTES; i += DEC_THREADS) hb[i] = i < nh ? pk[i] : 0;
        }
        __syncthreads();
        const uint8_t *hb = reinterpret_cast<const uint8_t *>(S.head);
        const int model = hb[1];
        const uint32_t seed = (uint32_t)hb[2] | ((uint32_t)hb[3] << 8) | ((uint32_t)hb[4] << 16) | ((uint32_t)hb[5] << 24);
        const uint32_t nM32 = (uint32_t)hb[6] | ((uint32_t)hb[7] << 8) | ((uint32_t)hb[8] << 16) | ((uint32_t)hb[9] << 24);
        const uint32_t nStream = gf_stream_len(model, nR, nC);
        int32_t early = GF_K_OK;
        if (model < 1 || model > 4) early = GF_K_ERR_FORMAT;            // CodecHuffman.java:155-169
        else if ((int32_t)nM32 < 0) early = GF_K_ERR_BOUNDS;            // NegativeArraySizeException
        else if ((uint64_t)nM32 > 6ull * nCells) early = GF_K_ERR_FORMAT; // no encoder emits this
        else if ((model == 2 && nC < 2)) early = GF_K_ERR_BOUNDS;       // PredictorModelLinear.java:80 output[1]
        else if (nM32 < nStream) early = GF_K_ERR_BOUNDS;               // M32 reads run off codeM32s
        if (early != GF_K_OK) {
            if (tid == 0) a.status[t] = early;
            __syncthreads();
            continue;
        }

        if (wave == 0) {
            // HuffmanDecoder.decodeTree, HuffmanDecoder.java:65-161.  Executed wave-uniformly by all
            // lanes of wave 0 (scalar loop, see GF_UNI in huff_build.h); lane 0 does the stores.
            const bool writer = lane == 0;
            uint32_t bp = 80;
            const uint32_t totalBits = len * 8u;
            auto getBits = [&](uint32_t nb) -> uint32_t {       // nb <= 9
                const uint32_t wi = bp >> 5, sh = bp & 31u;
                uint64_t w = 0;
                if (wi + 1 < HEAD_BYTES / 4 + 2) w = ((uint64_t)GF_UNI(S.head[wi + 1]) << 32) | GF_UNI(S.head[wi]);
                bp += nb;
                return (uint32_t)(w >> sh) & ((1u << nb) - 1u);
            };
            int32_t st = GF_K_OK;
            int32_t uniformSym = -1;
            const uint32_t nLeaves = getBits(8) + 1;
            const uint32_t rootBit = getBits(1);
            uint32_t nodes = 1;
            if (rootBit == 1) {
                uniformSym = (int32_t)getBits(8);
            } else {
                uint32_t leaves = 0;
                int sp = 0;
                if (writer) {
                    S.stack[0] = 0;
                    S.childCount[0] = 0;
                    S.child0[0] = 0;
                    S.child1[0] = 0;
                }
                while (leaves < nLeaves) {
                    const uint32_t parent = GF_UNI(S.stack[sp]);
                    const uint32_t cc = GF_UNI(S.childCount[parent]);
                    if (nodes >= 2 * nLeaves || nodes >= 511) { st = GF_K_ERR_BOUNDS; break; }
                    const uint32_t id = nodes++;
                    if (writer) {
                        if (cc == 0) S.child0[parent] = (uint16_t)id;
                        else S.child1[parent] = (uint16_t)id;
                        S.childCount[parent] = (uint8_t)(cc + 1);
                    }
                    if (getBits(1)) {
                        const uint32_t sym = getBits(8);
                        if (writer) {
                            S.leafSym[id] = (uint8_t)sym;
                            S.child0[id] = 0xFFFFu;
                            S.child1[id] = 0xFFFFu;
                            S.childCount[id] = 2;
                        }
                        leaves++;
                        if (leaves == nLeaves) break;
                        while (sp >= 0 && GF_UNI(S.childCount[GF_UNI(S.stack[sp])]) == 2) sp--;
                        if (sp < 0) { st = GF_K_ERR_BOUNDS; break; }
                    } else {
                        sp++;
                        if (sp > (int)nLeaves || sp >= 259) { st = GF_K_ERR_BOUNDS; break; }
                        if (writer) {
                            S.childCount[id] = 0;
                            S.child0[id] = 0;
                            S.child1[id] = 0;
                            S.stack[sp] = (uint16_t)id;
                        }
                    }
                }
                if (st == GF_K_OK) {
                    // every branch must have both children, otherwise the reference walks garbage
                    bool bad = false;
                    for (uint32_t k = lane; k < nodes; k += 64) bad |= S.childCount[k] != 2;
                    if (__ballot(bad) != 0ull) st = GF_K_ERR_FORMAT;
                }
            }
            if (st == GF_K_OK && bp > totalBits) st = GF_K_ERR_BOUNDS;   // read past end of data
            if (writer) {
                S.uniformSym = uniformSym;
                S.textStart = bp;
                S.parseStatus = st;
            }
        }
        __syncthreads();
        if (S.parseStatus != GF_K_OK) {
            if (tid == 0) a.status[t] = S.parseStatus;
            __syncthreads();
            continue;
        }

        if (a.phaseLimit == 1) continue;
        uint8_t *m32 = nM32 <= a.ldsM32Bytes ? ldsM32 : a.workspace + (size_t)blockIdx.x * a.workspaceStride;
        int32_t tileStatus = GF_K_OK;

        // ---------------- phase 1: Huffman text -> M32 bytes ----------------
        if (S.uniformSym >= 0) {
            const uint8_t sym = (uint8_t)S.uniformSym;
            for (uint32_t i = tid; i < nM32; i += DEC_THREADS) m32[i] = sym;
            __syncthreads();
        } else {
            // LUT: walk the tree with the low bits of every 11-bit index
            for (uint32_t e = tid; e < (1u << LUT_BITS); e += DEC_THREADS) {
                uint32_t node = 0, d = 0;
                while (d < LUT_BITS && S.child0[node] != 0xFFFFu) {
                    node = ((e >> d) & 1u) ? S.child1[node] : S.child0[node];
                    d++;
                }
                S.lut[e] = S.child0[node] == 0xFFFFu ? (uint16_t)((d << 8) | S.leafSym[node]) : (uint16_t)(0x8000u | node);
            }
            __syncthreads();
            const uint32_t textStart = S.textStart, endBit = len * 8u;
            HuffStep step{w32, nWords, off * 8ull, &S};
            const uint32_t textBits = endBit - textStart;
            uint32_t unit = (textBits + MAXQ - 1) / MAXQ;
            unit = max(128u, (unit + 31u) & ~31u);
            const uint32_t Q = max(1u, (textBits + unit - 1) / unit);
            resolve_chain(S, step, textStart, endBit, unit, Q);
            if (S.chainTotal < nM32) tileStatus = GF_K_ERR_BOUNDS;       // ran out of bits
            if (tid == 0) S.chainEnd = 0;
            __syncthreads();
            for (uint32_t q = tid; q < Q; q += DEC_THREADS) {
                uint32_t pos = S.qs[q], k = S.qn[q], sym;
                const uint32_t limit = min(endBit, textStart + (q + 1) * unit);
                while (pos < limit && k < nM32) {
                    pos = step(pos, &sym);
                    m32[k++] = (uint8_t)sym;
                    if (k == nM32) S.chainEnd = pos;
                }
            }
            __syncthreads();
            if (tileStatus == GF_K_OK && S.chainEnd > endBit) tileStatus = GF_K_ERR_BOUNDS;  // last code ran past the end
        }
        if (tileStatus != GF_K_OK) {
            if (tid == 0) a.status[t] = tileStatus;
            __syncthreads();
            continue;
        }

        if (a.phaseLimit == 2) continue;
        // ---------------- phase 2: M32 bytes -> residuals at their cells ----------------
        {
            M32Step step{m32, nM32};
            uint32_t unit = (nM32 + MAXQ - 1) / MAXQ;
            unit = max(16u, unit);
            const uint32_t Q = max(1u, (nM32 + unit - 1) / unit);
            resolve_chain(S, step, 0u, nM32, unit, Q);
            if (tid == 0) S.chainEnd = 0;
            __syncthreads();
            if (S.chainTotal < nStream) tileStatus = GF_K_ERR_BOUNDS;    // predictor reads past codeM32s
            for (uint32_t q = tid; q < Q; q += DEC_THREADS) {
                uint32_t pos = S.qs[q], k = S.qn[q], val;
                const uint32_t limit = min(nM32, (q + 1) * unit);
                while (pos < limit && k < nStream) {
                    pos = step(pos, &val);
                    o[gf_stream_cell(model, nR, nC, k)] = val;
                    k++;
                    if (k == nStream) S.chainEnd = pos;
                }
            }
            __syncthreads();
            if (tileStatus == GF_K_OK && S.chainEnd > nM32) tileStatus = GF_K_ERR_BOUNDS;   // last value truncated
        }
        if (tileStatus != GF_K_OK) {
            if (tid == 0) a.status[t] = tileStatus;
            __syncthreads();
            continue;
        }

        if (a.phaseLimit == 3) continue;
        // ---------------- phase 3: predictor inverse (wrap-around prefix sums) ----------------
        if (model != 4) {
            // Triangle: column sums of the interior residuals first (needs row 0 still as residuals)
            if (model == 3) {
                for (uint32_t c = 1 + tid; c < nC; c += DEC_THREADS) {
                    uint32_t acc = o[c];
                    for (uint32_t r = 1; r < nR; r++) {
                        acc += o[r * nC + c];
                        o[r * nC + c] = acc;
                    }
                }
            }
            // column 0 chain: o[r][0] = seed + sum of the column-0 residuals (all three models)
            if (wave == 0) {
                uint32_t carry = seed;
                if (lane == 0) o[0] = seed;
                for (uint32_t r0 = 1; r0 < nR; r0 += 64) {
                    const uint32_t r = r0 + lane;
                    const uint32_t x = r < nR ? o[r * nC] : 0u;
                    uint32_t v;
                    carry = row_scan_segment(x, carry, lane, &v);
                    if (r < nR) o[r * nC] = v;
                }
            }
            __syncthreads();
            // rows
            for (uint32_t r = wave; r < nR; r += DEC_WAVES) {
                uint32_t *row = o + (size_t)r * nC;
                if (model == 2) {
                    // second column, then c[k] = 2b - a + res  <=>  first differences are a running sum
                    uint32_t a0 = row[0];
                    uint32_t b0 = row[1] + a0;                          // residual of (r,1) is relative to (r,0)
                    uint32_t carryD = b0 - a0, carryV = b0;
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) row[1] = b0;
                    for (uint32_t c0 = 2; c0 < nC; c0 += 64) {
                        const uint32_t c = c0 + lane;
                        const uint32_t x = c < nC ? row[c] : 0u;
                        uint32_t d, v;
                        carryD = row_scan_segment(x, carryD, lane, &d);
                        carryV = row_scan_segment(c < nC ? d : 0u, carryV, lane, &v);
                        if (c < nC) row[c] = v;
                    }
                } else {
                    uint32_t carry = row[0];
                    for (uint32_t c0 = 1; c0 < nC; c0 += 64) {
                        const uint32_t c = c0 + lane;
                        const uint32_t x = c < nC ? row[c] : 0u;
                        uint32_t v;
                        carry = row_scan_segment(x, carry, lane, &v);
                        if (c < nC) row[c] = v;
                    }
                }
            }
        } else {
            // PredictorModelDifferencingWithNulls.java:137-166: column 0 first (row starts depend on the
            // first cell of the previous row), then every row on its own
            if (wave == 0) {
                // wave-uniform scalar loop (all lanes hold the same state; lane 0 stores)
                uint32_t prior = seed;
                bool nullFlag = true;
                for (uint32_t r = 0; r < nR; r++) {
                    const uint32_t test = GF_UNI(o[(size_t)r * nC]);
                    uint32_t first = GF_NULL_CODE;
                    if (test != GF_NULL_CODE) {
                        first = (nullFlag ? seed : prior) + test;
                        if (lane == 0) o[(size_t)r * nC] = first;
                    }
                    // row start of the next row: prior = first cell of this row, flag by VALUE (:162-163)
                    prior = first;
                    nullFlag = first == GF_NULL_CODE;
                }
            }
            __syncthreads();
            for (uint32_t r = tid; r < nR; r += DEC_THREADS) {
                uint32_t *row = o + (size_t)r * nC;
                uint32_t prior = row[0];
                // inside a row the flag follows the residual just decoded, not the value
                bool nullFlag;
                {
                    // (r,0) was null iff its residual was the null code; its final value is then the null code too
                    nullFlag = prior == GF_NULL_CODE;
                }
                for (uint32_t c = 1; c < nC; c++) {
                    const uint32_t test = row[c];
                    if (test == GF_NULL_CODE) {
                        nullFlag = true;
                    } else {
                        if (nullFlag) { nullFlag = false; prior = seed; }
                        prior += test;
                        row[c] = prior;
                    }
                }
            }
        }
        if (tid == 0) a.status[t] = GF_K_OK;
        __syncthreads();
    }
}

}  // namespace

uint32_t gf_huffman_decode_lds_m32(int nRows, int nCols)
{
    // typical M32 streams are ~1.0-1.1 bytes per cell; larger ones spill to the workspace
    size_t cells = (size_t)nRows * (size_t)nCols;
    size_t want = cells + cells / 4 + 1024;
    if (want < 16384) want = 16384;
    if (want > 98304) want = 98304;
    return (uint32_t)((want + 15) & ~(size_t)15);
}

unsigned gf_huffman_decode_grid(size_t nTiles)
{
    const size_t cap = 256 * 8;                    // workgroups resident on the chip, upper bound
    return (unsigned)(nTiles < cap ? (nTiles ? nTiles : 1) : cap);
}

hipError_t gf_launch_huffman_decode(const GfDecodeArgs &a, hipStream_t stream, unsigned grid)
{
    if (a.nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_huffman_decode, dim3(grid), dim3(DEC_THREADS), a.ldsM32Bytes, stream, a);
    return hipGetLastError();
}
