// gvrs_decode_common.h -- device helpers shared by the decode kernels: workgroup scans, stream-order
// to cell mapping, the in-place predictor inverse.  Include inside the kernel file's anonymous namespace
// after defining DEC_THREADS / DEC_WAVES.
#pragma once

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
    (void)lane;
    return gf_wave_incl_scan(v);
}

// exclusive scan over the workgroup (two barriers); *total = sum over all threads
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *waveSum, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = (int)gf_wave_id();
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) waveSum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < DEC_WAVES; w++) {
        const uint32_t s = waveSum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// Cell of stream element k of a predictor's stream (PredictorModel*.java: the order the encoder walks the tile in), as ONE
// formula whose parameters are worked out once per tile (wave-uniform, in SGPRs) -- so the per-value code has no branch on the
// model and none on "border or interior": written as a switch over the model with per-lane ifs inside, the mapping cost ~50
// scalar instructions per call for its exec masks and its dispatch, in the value loops of the canonical decoders (round 3).
//   k <  hA         : k + hAdd                    (row 0 of Triangle / Differencing / the identity; element 0 of Linear)
//   k <  nHead      : u = k - hA:  (1 + (u >> sh)) * nC + (u & msk)
//                     (Linear: elements 2r-1, 2r are cells (r,0), (r,1);  Triangle: column 0 from row 1 on)
//   else            : t = k - nHead, r = t / w:   base + t + r * colBase      (= (r + rowBase) * nC + colBase + t - r * w,
//                     w = nC - colBase cells of a row in the interior, colBase = 2 for Linear, 1 for Triangle)
// t / w is a multiply-high by a 33-bit reciprocal (Granlund & Montgomery: m = floor(2^32 (2^s - w) / w) + 1, s = ceil(log2 w);
// q = mulhi(t, m); r = (((t - q) >> min(s, 1)) + q) >> max(s - 1, 0)): exact for every 32-bit t, no division fallback.
struct GfCellMap {
    uint32_t nC, hA, hAdd, nHead, sh, msk, base, colBase, m, s1, s2;
    __device__ __forceinline__ static GfCellMap make(int model, uint32_t nR, uint32_t nC)
    {
        GfCellMap c;
        const bool linear = model == 2, tri = model == 3;
        c.nC = nC;
        c.hAdd = model == 4 ? 0u : 1u;
        c.hA = linear ? 1u : tri ? nC - 1u : 0xFFFFFFFFu;
        c.nHead = linear ? 2u * nR - 1u : tri ? nC + nR - 2u : 0xFFFFFFFFu;
        c.sh = linear ? 1u : 0u;
        c.msk = linear ? 1u : 0u;
        c.colBase = linear ? 2u : 1u;
        c.base = (linear ? 0u : nC) + c.colBase;
        const uint32_t w = nC > c.colBase ? nC - c.colBase : 1u;
        const uint32_t s = w > 1u ? 32u - (uint32_t)__builtin_clz(w - 1u) : 0u;
        c.m = (uint32_t)(((uint64_t)((1ull << s) - w) << 32) / w) + 1u;
        c.s1 = s ? 1u : 0u;
        c.s2 = s - c.s1;
        return c;
    }
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const
    {
        const uint32_t u = k - hA;
        const uint32_t border = (1u + (u >> sh)) * nC + (u & msk);
        const uint32_t t = k - nHead;                            // (of no meaning before the interior)
        const uint32_t q = __umulhi(t, m);
        const uint32_t r = (((t - q) >> s1) + q) >> s2;
        const uint32_t interior = base + t + r * colBase;
        return k < hA ? k + hAdd : k < nHead ? border : interior;
    }
};

// wave-wide inclusive scan of one row segment with carry; returns the new carry
__device__ __forceinline__ uint32_t row_scan_segment(uint32_t x, uint32_t carry, int lane, uint32_t *outv)
{
    const uint32_t incl = wave_incl_scan(x, lane) + carry;
    *outv = incl;
    return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}

constexpr int ROW_BATCH = DEC_THREADS > 256 ? 4 : 8;   // rows a wave keeps in flight in the row scans (the builds with more waves per tile are held to 64 VGPRs)

// Predictor inverse in place (residuals at their cells -> values), int32 wrap-around prefix sums:
// PredictorModelDifferencing.java:145-167, PredictorModelLinear.java:66-101, PredictorModelTriangle.java:62-98,
// PredictorModelDifferencingWithNulls.java:137-166 (and the decodeInt twins).  Whole workgroup; o[0] need not
// hold the seed.  stamp (optional) receives a cycle stamp after the column-0 chain.
constexpr int COL_BATCH = DEC_THREADS > 256 ? 8 : 16;  // rows of a column in flight in the Triangle column sums

__device__ __forceinline__ void gf_predictor_inverse(int model, uint32_t seed, uint32_t *o, uint32_t nR, uint32_t nC,
                                                     uint32_t *stamp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)gf_wave_id();
    if (model != 4) {
        // Triangle: column sums of the interior residuals first (needs row 0 still as residuals)
        if (model == 3) {
            for (uint32_t c = 1 + tid; c < nC; c += DEC_THREADS) {
                uint32_t acc = o[c];
                for (uint32_t r = 1; r < nR; r += COL_BATCH) {
                    uint32_t x[COL_BATCH];
#pragma unroll
                    for (int j = 0; j < COL_BATCH; j++) x[j] = r + j < nR ? o[(size_t)(r + j) * nC + c] : 0u;
#pragma unroll
                    for (int j = 0; j < COL_BATCH; j++) {
                        acc += x[j];
                        if (r + j < nR) o[(size_t)(r + j) * nC + c] = acc;
                    }
                }
            }
        }
        // column 0 chain: o[r][0] = seed + sum of the column-0 residuals (all three models)
        if (wave == 0) {
            uint32_t carry = seed;
            if (lane == 0) o[0] = seed;
            for (uint32_t r0 = 1; r0 < nR; r0 += 64) {
                const uint32_t r = r0 + lane;
                const uint32_t x = r < nR ? o[(size_t)r * nC] : 0u;
                uint32_t v;
                carry = row_scan_segment(x, carry, lane, &v);
                if (r < nR) o[(size_t)r * nC] = v;
            }
        }
        __syncthreads();
        if (stamp && tid == 0) *stamp = (uint32_t)__builtin_amdgcn_s_memtime();
        // rows: each wave keeps ROW_BATCH rows in flight
        for (uint32_t rb = (uint32_t)wave * ROW_BATCH; rb < nR; rb += DEC_WAVES * ROW_BATCH) {
            uint32_t carryV[ROW_BATCH], carryD[ROW_BATCH];
            uint32_t cStart = 1;
#pragma unroll
            for (int b = 0; b < ROW_BATCH; b++) {
                const uint32_t r = rb + b;
                carryV[b] = 0;
                carryD[b] = 0;
                if (r < nR) {
                    uint32_t *row = o + (size_t)r * nC;
                    if (model == 2) {
                        // second column, then c[k] = 2b - a + res  <=>  first differences are a running sum
                        const uint32_t a0 = row[0];
                        const uint32_t b0 = row[1] + a0;            // residual of (r,1) is relative to (r,0)
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) row[1] = b0;
                        carryD[b] = b0 - a0;
                        carryV[b] = b0;
                    } else {
                        carryV[b] = row[0];
                    }
                }
            }
            if (model == 2) cStart = 2;
            uint32_t xn[ROW_BATCH];                                 // the next chunk is loaded while this one is scanned
#pragma unroll
            for (int b = 0; b < ROW_BATCH; b++)
                xn[b] = (rb + b < nR && cStart + lane < nC) ? o[(size_t)(rb + b) * nC + cStart + lane] : 0u;
            for (uint32_t c0 = cStart; c0 < nC; c0 += 64) {
                const uint32_t c = c0 + lane;
                uint32_t x[ROW_BATCH];
#pragma unroll
                for (int b = 0; b < ROW_BATCH; b++) x[b] = xn[b];
#pragma unroll
                for (int b = 0; b < ROW_BATCH; b++)
                    xn[b] = (rb + b < nR && c + 64u < nC) ? o[(size_t)(rb + b) * nC + c + 64u] : 0u;
#pragma unroll
                for (int b = 0; b < ROW_BATCH; b++) {
                    uint32_t v;
                    if (model == 2) {
                        uint32_t d;
                        carryD[b] = row_scan_segment(x[b], carryD[b], lane, &d);
                        carryV[b] = row_scan_segment(c < nC ? d : 0u, carryV[b], lane, &v);
                    } else {
                        carryV[b] = row_scan_segment(x[b], carryV[b], lane, &v);
                    }
                    if (rb + b < nR && c < nC) o[(size_t)(rb + b) * nC + c] = v;
                }
            }
        }
    } else {
        // PredictorModelDifferencingWithNulls.java:137-166.  Inside a row the flag follows the RESIDUAL just decoded (null code or
        // not), at a row start it follows the VALUE of the previous row's first cell (:162-163) -- and a sum may come out as
        // the null code, Integer.MIN_VALUE, without any null residual (damaged input only: the encoder never sees that value
        // in a valid cell).  So nothing here may ask a value whether "it was null":
        //   pass 1, a thread per row: everything behind the row's first null residual (from column 1 on when (r,0) itself is
        //           null) -- restarts from the seed, independent of column 0; the stretch before it stays as residuals;
        //   pass 2, wave 0: the column-0 chain over the rows.  Where a sum comes out as the null code, the row's first stretch
        //           is finished right there, continuing from that sum as the reference does;
        //   pass 3, a thread per row: the first stretch of the rows whose first VALUE is not the null code (the others are
        //           complete: a null residual has no first stretch, the rare other case was finished in pass 2).
        for (uint32_t r = tid; r < nR; r += DEC_THREADS) {
            uint32_t *row = o + (size_t)r * nC;
            uint32_t c = 1;
            if (row[0] != GF_NULL_CODE)
                while (c < nC && row[c] != GF_NULL_CODE) c++;
            uint32_t prior = seed;
            bool nullFlag = true;
            for (; c < nC; c++) {
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
        __syncthreads();
        if (wave == 0) {
            // wave-uniform scalar loop (all lanes hold the same state; lane 0 stores)
            uint32_t prior = seed;
            bool nullFlag = true;
            for (uint32_t r = 0; r < nR; r++) {
                const uint32_t test = GF_UNI(o[(size_t)r * nC]);
                uint32_t first = GF_NULL_CODE;
                if (test != GF_NULL_CODE) {
                    first = (nullFlag ? seed : prior) + test;
                    if (lane == 0) {
                        uint32_t *row = o + (size_t)r * nC;
                        row[0] = first;
                        if (first == GF_NULL_CODE) {
                            uint32_t p = first;
                            for (uint32_t c = 1; c < nC && row[c] != GF_NULL_CODE; c++) {
                                p += row[c];
                                row[c] = p;
                            }
                        }
                    }
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
            if (prior == GF_NULL_CODE) continue;
            for (uint32_t c = 1; c < nC; c++) {
                const uint32_t test = row[c];
                if (test == GF_NULL_CODE) break;
                prior += test;
                row[c] = prior;
            }
        }
    }
}
