// gvrs_lsop.hip -- the LSOP12 codec (optimal 12-coefficient linear predictor) for a batch of tiles on gfx950.
//
// Reference (core/src/main/java/org/gridfour/):
//   lsop/LsOptimalPredictor12.java:109-292   encode: initialiser stream, coefficients, interior stream
//   lsop/LsOptimalPredictor12.java:311-383   computeCoefficients: 13x13 normal equations with a Lagrange row
//   util/jama/LUDecomposition.java:70-134, 253-284   Crout LU with partial pivoting, solve
//   lsop/LsEncoder12.java:122-219, lsop/LsHeader.java:210-265   container: header + canonical Huffman of the two
//       integer streams in one bit store (compression type 2)
//   lsop/LsDecoder12.java:94-160, 186-221, 311-383   decode
//
// Four kernels, all one tile per workgroup (k_lsop_reconstruct: one tile per WAVE):
//   k_lsop_predict      tile -> seed, 12 float coefficients, residual ints [initialisers | interior]
//   k_canon_pack2       header + CanonicalHuffman(initialisers) + CanonicalHuffman(interior) -> packing
//   k_lsop_unpack2      packing -> seed, coefficients, residual ints (two canonical-Huffman streams back to back)
//   k_lsop_reconstruct  residual ints -> tile: prefix sums for the border cells, then the interior as a
//                       slope-3 wavefront (cell (r,c) needs (r-1,c+2) and (r-2,c+2): step = c + 3r)
//
// Floating point: the normal equations are accumulated in FP64.  When max|v|^2 * cells < 2^53 every partial sum
// is an exact integer, so any order (and FMA) gives the reference's bits: lanes split the cells, waves split the
// 104 accumulators.  Otherwise 104 threads own one accumulator each and add in the reference's scan order with
// separately rounded multiply and add.  LU, solve and the float32 prediction follow the reference's operation
// order exactly (no contraction: the library is built with -ffp-contract=off).

#include <hip/hip_runtime.h>

#include "gvrs_kernels.h"
#include "gvrs_encode_layout.h"
#include "huff_build.h"

namespace {

#include "gvrs_encode_common.h"
#include "gvrs_canon_common.h"

// ------------------------------------------------------------------------------------------------
// shared by predict and reconstruct
// ------------------------------------------------------------------------------------------------

// StrictMath.round(float): closest int, ties toward +infinity, NaN -> 0, saturating
__device__ __forceinline__ int32_t lsop_round(float p)
{
    if (p != p) return 0;
    const double f = floor((double)p + 0.5);
    if (f >= 2147483647.0) return 2147483647;
    if (f <= -2147483648.0) return (int32_t)0x80000000;
    return (int32_t)f;
}

// the same value without branches: v_cvt_i32_f64 saturates and turns NaN into 0 by itself
__device__ __forceinline__ int32_t lsop_round_sat(float p)
{
    const double f = floor((double)p + 0.5);
    int32_t r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// The same value from single-precision steps only (round 5; FP64 instructions issue at a quarter of the rate and the interior
// stream rounds once per cell): floor(p) is exact, and so is p - floor(p) -- for |p| >= 1 the two lie within a factor of two of
// each other (Sterbenz); for 0 <= p < 1 it is p itself; for -1 < p < 0 it is p + 1, which rounds only where p is so small that
// both the rounded and the true value are far above one half -- hence floor(p + 0.5) = floor(p) + [p - floor(p) >= 0.5] with no
// rounding anywhere.  v_cvt_i32_f32 saturates and turns NaN into 0 (an infinite p: inf - inf = NaN fails the compare).
__device__ __forceinline__ int32_t lsop_round_f32(float p)
{
    const float fl = floorf(p);
    const float up = (p - fl) >= 0.5f ? fl + 1.0f : fl;       // (fl + 1 is exact below 2^24; beyond that p is an integer and p - fl = 0)
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(up));
    return r;
}

// u1*z1 + ... + u12*z12 in float32, left to right (LsOptimalPredictor12.java:254-267)
__device__ __forceinline__ float lsop_predict12(const float *u, const int32_t *v, uint32_t idx, uint32_t nC)
{
    float p = u[0] * (float)v[idx - 1];
    p = p + u[1] * (float)v[idx - nC - 1];
    p = p + u[2] * (float)v[idx - nC];
    p = p + u[3] * (float)v[idx - nC + 1];
    p = p + u[4] * (float)v[idx - nC + 2];
    p = p + u[5] * (float)v[idx - 2];
    p = p + u[6] * (float)v[idx - nC - 2];
    p = p + u[7] * (float)v[idx - 2 * nC - 2];
    p = p + u[8] * (float)v[idx - 2 * nC - 1];
    p = p + u[9] * (float)v[idx - 2 * nC];
    p = p + u[10] * (float)v[idx - 2 * nC + 1];
    p = p + u[11] * (float)v[idx - 2 * nC + 2];
    return p;
}

// initialiser element k -> its cell and whether it is a plain difference (kind 0: left, 1: up) or a triangle
// residual (kind 2); order of LsOptimalPredictor12.java:143-209
__device__ __forceinline__ uint32_t lsop_init_cell(uint32_t k, uint32_t nR, uint32_t nC, uint32_t *kind)
{
    if (k < nC - 1u) { *kind = 0; return k + 1u; }
    k -= nC - 1u;
    if (k < nR - 1u) { *kind = 1; return (k + 1u) * nC; }
    k -= nR - 1u;
    *kind = 2;
    if (k < nC - 1u) return nC + k + 1u;
    k -= nC - 1u;
    if (k < nR - 2u) return (k + 2u) * nC + 1u;
    k -= nR - 2u;
    return (2u + (k >> 1)) * nC + nC - 2u + (k & 1u);
}

__device__ __forceinline__ uint32_t lsop_n_init(uint32_t nR, uint32_t nC) { return 4u * nR + 2u * nC - 9u; }
__device__ __forceinline__ uint32_t lsop_n_interior(uint32_t nR, uint32_t nC) { return (nR - 2u) * (nC - 4u); }

// ------------------------------------------------------------------------------------------------
// k_lsop_predict
// ------------------------------------------------------------------------------------------------

// the 104 accumulators: pairs (i, j) with i <= j over z0..z12 plus z13 = 1 (the plain sums), without (13,13)
struct PairTab {
    int8_t i[104], j[104];
};
constexpr PairTab make_pairs()
{
    PairTab t{};
    int p = 0;
    for (int i = 0; i < 13; i++)
        for (int j = i; j < 14; j++) { t.i[p] = (int8_t)i; t.j[p] = (int8_t)j; p++; }
    return t;
}
__device__ constexpr PairTab PAIRS = make_pairs();

struct LsopShared {
    double G[104];
    int32_t C32[1024];         // lsop_gram_mfma: the 32 x 32 digit Gram matrix
    float u[12];
    uint32_t maxAbs;
    int32_t status;
    int32_t off[14];
};

// z_i of the cell at idx as an offset into the tile (LsOptimalPredictor12.java:322-334)
__device__ __forceinline__ int32_t lsop_z_offset(int i, int32_t nC)
{
    switch (i) {
    case 0: return 0;
    case 1: return -1;
    case 2: return -nC - 1;
    case 3: return -nC;
    case 4: return -nC + 1;
    case 5: return -nC + 2;
    case 6: return -2;
    case 7: return -nC - 2;
    case 8: return -2 * nC - 2;
    case 9: return -2 * nC - 1;
    case 10: return -2 * nC;
    case 11: return -2 * nC + 1;
    default: return -2 * nC + 2;
    }
}

// fast path: wave W accumulates pairs [26 W, 26 W + 26) over the cells its lanes own
template <int W>
__device__ __forceinline__ void lsop_gram_wave(const int32_t *__restrict__ v, uint32_t nC, uint32_t nInt, double *G, int lane)
{
    double acc[26];
#pragma unroll
    for (int q = 0; q < 26; q++) acc[q] = 0.0;
    const uint32_t wI = nC - 4u;
    uint32_t r = (uint32_t)lane / wI, c = (uint32_t)lane - r * wI;          // row / column of cell e, kept incrementally
    for (uint32_t e = (uint32_t)lane; e < nInt; e += 64) {
        const int32_t *p = v + (size_t)(r + 2u) * nC + (c + 2u);
        const int32_t n = (int32_t)nC;
        double z[14];
        z[0] = (double)p[0];
        z[1] = (double)p[-1];
        z[2] = (double)p[-n - 1];
        z[3] = (double)p[-n];
        z[4] = (double)p[-n + 1];
        z[5] = (double)p[-n + 2];
        z[6] = (double)p[-2];
        z[7] = (double)p[-n - 2];
        z[8] = (double)p[-2 * n - 2];
        z[9] = (double)p[-2 * n - 1];
        z[10] = (double)p[-2 * n];
        z[11] = (double)p[-2 * n + 1];
        z[12] = (double)p[-2 * n + 2];
        z[13] = 1.0;
#pragma unroll
        for (int q = 0; q < 26; q++) acc[q] = __fma_rn(z[PAIRS.i[W * 26 + q]], z[PAIRS.j[W * 26 + q]], acc[q]);   // exact: see the guard
        c += 64u;
        while (c >= wI) { c -= wI; r++; }
    }
#pragma unroll
    for (int q = 0; q < 26; q++) {
        double a = acc[q];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) G[W * 26 + q] = a;
    }
}

// The same sums with the tile's rows passing through a four-row ring in LDS: a row is loaded from HBM once, by the whole
// workgroup and coalesced, one row ahead of its use, and the 13 neighbours of a cell are LDS reads.  (Read straight from
// global memory, every one of the four waves issues 13 scattered loads per cell group: 2.71 -> 2.45 ms for the kernel on the
// ETOPO1-shaped batch, 3.39 -> 2.51 ms on 256x256 tiles; two rows per barrier with the cells of the pair dealt flat to the
// lanes measured the same.)  Wave w still owns the pairs [26 w, 26 w + 26); one barrier per row.
template <int W>
__device__ __forceinline__ void lsop_gram_fma(const double *z, double *acc)
{
#pragma unroll
    for (int q = 0; q < 26; q++) acc[q] = __fma_rn(z[PAIRS.i[W * 26 + q]], z[PAIRS.j[W * 26 + q]], acc[q]);   // exact: see the guard
}

constexpr uint32_t LSOP_RING_MAXC = 256;         // widest row the ring path takes (one prefetch register per thread)

// Every wave of the workgroup runs this loop (each its own instantiation: only the 26 pairs differ) with the same trip
// counts, so the barriers inside match up.
template <int W>
__device__ __forceinline__ void lsop_gram_rows(const int32_t *__restrict__ v, int32_t *ring, uint32_t nR, uint32_t nC, double *G, int tid)
{
    const int lane = tid & 63;
    double acc[26];
#pragma unroll
    for (int q = 0; q < 26; q++) acc[q] = 0.0;
    for (uint32_t i = (uint32_t)tid; i < 3u * nC; i += 256u) ring[i] = v[i];          // rows 0..2 -> slots 0..2
    __syncthreads();
    for (uint32_t r = 2; r < nR; r++) {
        const bool more = r + 1u < nR && (uint32_t)tid < nC;
        const int32_t pre = more ? v[(size_t)(r + 1u) * nC + (uint32_t)tid] : 0;
        const int32_t *r0 = ring + (r & 3u) * nC, *r1 = ring + ((r - 1u) & 3u) * nC, *r2 = ring + ((r - 2u) & 3u) * nC;
#pragma unroll 1
        for (uint32_t c = 2u + (uint32_t)lane; c < nC - 2u; c += 64u) {
            double z[14];
            z[0] = (double)r0[c];
            z[1] = (double)r0[c - 1];
            z[2] = (double)r1[c - 1];
            z[3] = (double)r1[c];
            z[4] = (double)r1[c + 1];
            z[5] = (double)r1[c + 2];
            z[6] = (double)r0[c - 2];
            z[7] = (double)r1[c - 2];
            z[8] = (double)r2[c - 2];
            z[9] = (double)r2[c - 1];
            z[10] = (double)r2[c];
            z[11] = (double)r2[c + 1];
            z[12] = (double)r2[c + 2];
            z[13] = 1.0;
            lsop_gram_fma<W>(z, acc);
        }
        if (more) ring[((r + 1u) & 3u) * nC + (uint32_t)tid] = pre;     // the slot of row r - 3
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 26; q++) {
        double a = acc[q];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) G[W * 26 + q] = a;
    }
}

// The same sums on the MATRIX pipe (round 4).  The normal equations are the Gram matrix Z^T Z of Z = (cells x 14) with
// z0..z12 and a column of ones (LsOptimalPredictor12.java:335-342): the one dense contraction of this code base.
// Under the exactness guard every z is an integer of |z| <= 32,639, i.e. two balanced base-256 digits  z = 256 hi + lo,
// lo = (int8) z, hi = (z + 128) >> 8, both in -128..127, and
//   SUM z_i z_j = 65536 SUM hi_i hi_j + 256 (SUM hi_i lo_j + SUM lo_i hi_j) + SUM lo_i lo_j :
// the 27 digit columns (13 lo, 13 hi, the ones) are one int8 operand of v_mfma_i32_32x32x32_i8 -- THE SAME register quadruple as
// A and as B: lane l holds column l & 31 for the sixteen cells 16 (l >> 5) .. + 15 of a group of 32 cells of a tile row, which is
// A[i][k] and B[k][i] at once -- and 32 cells are 27 x 27 x 32 exact integer multiply-adds in ONE instruction of 32 cycles, with
// int32 accumulators (128 x 128 x cells < 2^31 for cells < 2^17) that are recombined in int64 and converted to double once: the same
// values, bit for bit, as the scan-order FP64 sums (all of which are exact integers below 2^53 under the guard).  The rows pass
// through the four-row LDS ring of lsop_gram_rows; the groups of a tile are dealt round-robin to the four waves, which add
// their 32 x 32 accumulators up in LDS.  (FP64 FMA form: 26 fused multiply-adds per wave and 64 cells on each of the four
// SIMDs; v_mfma_f64_16x16x4_f64 has the vector rate on gfx950 and lost.)
typedef int LsV4i __attribute__((ext_vector_type(4)));
typedef int LsV16i __attribute__((ext_vector_type(16)));
constexpr uint32_t LSOP_MFMA_MAX_ABS = 32639u;             // (z + 128) >> 8 <= 127
__device__ __forceinline__ void lsop_gram_mfma(const int32_t *__restrict__ v, int32_t *ring, int32_t *C32, uint32_t nR, uint32_t nC,
                                               double *G, int tid)
{
    const int lane = tid & 63;
    const uint32_t wave = gf_wave_id();
    const uint32_t col = (uint32_t)lane & 31u, h = (uint32_t)lane >> 5;
    // what this lane's column is: digit `shift` of z_zi (13: the ones, 14: nothing)
    const uint32_t zi = col < 13u ? col : col < 26u ? col - 13u : col == 26u ? 13u : 14u;
    const uint32_t bias = (col >= 13u && col < 26u) ? 128u : 0u, shift = (col >= 13u && col < 26u) ? 8u : 0u;
    // z_i = v(r + dr, c + dc) (lsop_z_offset)
    int dr = 0, dc = 0;
    switch (zi) {
    case 1: dc = -1; break;
    case 2: dr = -1; dc = -1; break;
    case 3: dr = -1; break;
    case 4: dr = -1; dc = 1; break;
    case 5: dr = -1; dc = 2; break;
    case 6: dc = -2; break;
    case 7: dr = -1; dc = -2; break;
    case 8: dr = -2; dc = -2; break;
    case 9: dr = -2; dc = -1; break;
    case 10: dr = -2; break;
    case 11: dr = -2; dc = 1; break;
    case 12: dr = -2; dc = 2; break;
    default: break;
    }
    // Sixteen cells of a column become four operand words: the digit is byte 0 of v (lo) or byte 1 of v + 128 (hi), picked out of
    // two values at a time by v_perm_b32 with a per-lane selector; the ones column and the five idle columns are constants.
    const uint32_t selPair = 0x0c0c0000u | ((4u + (shift >> 3)) << 8) | (shift >> 3);     // byte s of the 2nd source, byte s of the 1st
    const uint32_t constWord = zi == 13u ? 0x01010101u : 0u;
    const bool isConst = zi >= 13u;
    LsV16i acc = {};
    for (uint32_t i = (uint32_t)tid; i < 1024u; i += 256u) C32[i] = 0;
    for (uint32_t i = (uint32_t)tid; i < 3u * nC; i += 256u) ring[i] = v[i];          // rows 0..2 -> slots 0..2
    __syncthreads();
    const uint32_t wI = nC - 4u, nGroups = (wI + 31u) >> 5;
    uint32_t turn = 0;                                                                // groups so far, over the rows: dealt to the waves
    for (uint32_t r = 2; r < nR; r++) {
        const bool more = r + 1u < nR && (uint32_t)tid < nC;
        const int32_t pre = more ? v[(size_t)(r + 1u) * nC + (uint32_t)tid] : 0;
        const int32_t *row = ring + ((r + (uint32_t)dr) & 3u) * nC + dc;               // the ring row this lane's z lives in, shifted
        for (uint32_t g = 0; g < nGroups; g++, turn++) {
            if ((turn & 3u) != wave) continue;                                        // (wave-uniform)
            const uint32_t c0 = 2u + 32u * g + 16u * h;                               // the lane's first cell of the group
            const int32_t *src = row + c0;                                            // (the ring has room behind its last row for the
                                                                                      //  reads of a row's last group: their bytes are masked)
            LsV4i x;
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) {
                const uint32_t t0 = (uint32_t)src[4u * q] + bias, t1 = (uint32_t)src[4u * q + 1u] + bias;
                const uint32_t t2 = (uint32_t)src[4u * q + 2u] + bias, t3 = (uint32_t)src[4u * q + 3u] + bias;
                const uint32_t p01 = __builtin_amdgcn_perm(t1, t0, selPair), p23 = __builtin_amdgcn_perm(t3, t2, selPair);
                const uint32_t w = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                x[q] = (int)(isConst ? constWord : w);
            }
            if (g + 1u == nGroups) {                                                  // (wave-uniform) the row's last group: cells behind it
                const uint32_t nValid = c0 < nC - 2u ? min(16u, nC - 2u - c0) : 0u;
#pragma unroll
                for (uint32_t q = 0; q < 4u; q++) {
                    const uint32_t have = nValid > 4u * q ? min(4u, nValid - 4u * q) : 0u;
                    x[q] &= (int)(have >= 4u ? 0xFFFFFFFFu : (1u << (8u * have)) - 1u);
                }
            }
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, x, acc, 0, 0, 0);
        }
        if (more) ring[((r + 1u) & 3u) * nC + (uint32_t)tid] = pre;     // the slot of row r - 3
        __syncthreads();
    }
    // the four waves' accumulators into one 32 x 32 matrix (result register q of lane l: row (q & 3) + 8 (q >> 2) + 4 (l >> 5), column l & 31)
#pragma unroll
    for (uint32_t q = 0; q < 16u; q++) atomicAdd(&C32[((q & 3u) + 8u * (q >> 2) + 4u * h) * 32u + col], acc[q]);
    __syncthreads();
    if (tid < 104) {
        const int i = PAIRS.i[tid], j = PAIRS.j[tid];
        long long sum;
        if (j == 13) sum = 256ll * C32[(13 + i) * 32 + 26] + C32[i * 32 + 26];           // the plain sums: against the ones
        else
            sum = 65536ll * C32[(13 + i) * 32 + 13 + j] + 256ll * ((long long)C32[(13 + i) * 32 + j] + C32[i * 32 + 13 + j]) + C32[i * 32 + j];
        G[tid] = (double)sum;
    }
}

__device__ __forceinline__ double lsop_bcast(double x, int k)           // value of lane k (k wave-uniform)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), k), hi = __builtin_amdgcn_readlane(__double2hiint(x), k);
    return __hiloint2double(hi, lo);
}

// JAMA's LU (LUDecomposition.java:70-134) and solve (:253-284) of the 13x13 system of
// LsOptimalPredictor12.computeCoefficients :353-378 on one wave: lane i holds row i in registers.  The reference's
// inner loop  s += LU[i][k] * LUcolj[k]  runs over k in the same order here, for all rows in lockstep (row i stops at
// k = min(i, j)); the column value of row k is final exactly when step k needs it.  Every multiply and add is
// rounded separately, as in Java.  Writes S.u / S.status.
template <class Shared>                                // LsopShared / LsopShared16: G, u, status
__device__ __forceinline__ void lsop_lu_solve_wave(Shared &S, int lane)
{
    auto Cij = [&](int i, int j) -> double {              // c[i][j], symmetric (:345-349)
        if (i > j) { const int q = i; i = j; j = q; }
        return S.G[i * 14 - (i * (i - 1)) / 2 + (j - i)];
    };
    auto Si = [&](int i) -> double { return S.G[i * 14 - (i * (i - 1)) / 2 + (13 - i)]; };
    const int row = lane < 13 ? lane : 12;                // lanes >= 13 shadow row 12 (results unused)
    double r[13], x;
#pragma unroll
    for (int j = 0; j < 12; j++) r[j] = row < 12 ? Cij(row + 1, j + 1) : Si(j + 1);
    r[12] = row < 12 ? Si(row + 1) : 0.0;
    x = row < 12 ? Cij(0, row + 1) : Si(0);               // right-hand side, permuted along with the rows
    bool singular = false;
#pragma unroll
    for (int j = 0; j < 13; j++) {
        double colv = r[j], s = 0.0;
#pragma unroll
        for (int k = 0; k < j; k++) {
            const double ck = lsop_bcast(__dsub_rn(colv, s), k);      // LUcolj[k] after its own update
            if (lane > k) s = __dadd_rn(s, __dmul_rn(r[k], ck));
        }
        colv = __dsub_rn(colv, s);
        r[j] = colv;
        // pivot: first row >= j with the largest |value| (strict > in scan order)
        int p = j;
        double ap = fabs(lsop_bcast(colv, j));
#pragma unroll
        for (int i = j + 1; i < 13; i++) {
            const double ai = fabs(lsop_bcast(colv, i));
            if (ai > ap) { p = i; ap = ai; }
        }
        p = __builtin_amdgcn_readfirstlane(p);
        if (p != j) {
            const int partner = lane == j ? p : lane == p ? j : lane;
#pragma unroll
            for (int c = 0; c < 13; c++) r[c] = __shfl(r[c], partner, 64);
            x = __shfl(x, partner, 64);
        }
        const double djj = lsop_bcast(r[j], j);
        if (djj != 0.0) { if (lane > j) r[j] = __ddiv_rn(r[j], djj); }
        else singular = true;
    }
    if (singular) {
        if (lane == 0) S.status = GF_K_DECLINED;          // "Matrix is singular." -> computeCoefficients returns null
        return;
    }
#pragma unroll
    for (int k = 0; k < 13; k++) {                        // L y = b(piv)
        const double xk = lsop_bcast(x, k);
        if (lane > k) x = __dsub_rn(x, __dmul_rn(xk, r[k]));
    }
#pragma unroll
    for (int k = 12; k >= 0; k--) {                       // U x = y
        if (lane == k) x = __ddiv_rn(x, r[k]);
        const double xk = lsop_bcast(x, k);
        if (lane < k) x = __dsub_rn(x, __dmul_rn(xk, r[k]));
    }
    if (lane < 12) S.u[lane] = (float)x;
}

struct GfLsopPredictArgs {
    const int32_t *values;
    int32_t *residuals;        // per tile resStride ints: [initialisers | interior]
    size_t resStride;
    uint32_t *coefs;           // per tile 16 words: seed, 12 float bit patterns, 3 spare
    int32_t *status;
    size_t nTiles;
    int nRows, nCols;
    int retryOnly;             // 1: behind k_lsop_predict16 -- only the tiles that kernel marked GF_K_LSOP_RETRY
};
constexpr int GF_K_LSOP_RETRY_ = 0x7fff0011;     // (= GF_K_LSOP_RETRY below)

#ifndef GF_LSOP_PREDICT_WGS
#define GF_LSOP_PREDICT_WGS 4
#endif
#ifndef GF_LSOP_PACK2_WGS
#define GF_LSOP_PACK2_WGS 7      // sweep 4 / 6 / 7 workgroups per CU (22 KB of LDS: 7 at most): LSOP12 encode 3.46 / 3.33 / 3.28 ms
#endif
__global__ __launch_bounds__(256, GF_LSOP_PREDICT_WGS) void k_lsop_predict(GfLsopPredictArgs a)
{
    __shared__ LsopShared S;
    extern __shared__ __attribute__((aligned(16))) int32_t lsopRing[];      // four rows of the tile (lsop_gram_rows)
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)gf_wave_id();
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const uint32_t nInit = lsop_n_init(nR, nC), nInt = lsop_n_interior(nR, nC);

    GF_FOR_WG_TILE(t, a.nTiles) {                                         // no tile loop: see gvrs_kernels.h
        if (a.retryOnly && a.status[t] != GF_K_LSOP_RETRY_) continue;
        const int32_t *__restrict__ v = a.values + t * (size_t)nCells;
        int32_t *__restrict__ res = a.residuals + t * a.resStride;
        if (tid == 0) { S.maxAbs = 0; S.status = GF_K_OK; }
        __syncthreads();

        // initialiser stream (:143-209) and max |v|
        for (uint32_t k = tid; k < nInit; k += 256) {
            uint32_t kind;
            const uint32_t idx = lsop_init_cell(k, nR, nC, &kind);
            const uint32_t x = (uint32_t)v[idx];
            uint32_t r;
            if (kind == 0) r = x - (uint32_t)v[idx - 1];
            else if (kind == 1) r = x - (uint32_t)v[idx - nC];
            else r = x - ((uint32_t)v[idx - 1] + (uint32_t)v[idx - nC] - (uint32_t)v[idx - nC - 1]);
            res[k] = (int32_t)r;
        }
        uint32_t m = 0;
        for (uint32_t i = tid; i < nCells; i += 256) {
            const int32_t x = v[i];
            m = max(m, x < 0 ? 0u - (uint32_t)x : (uint32_t)x);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, gf_lane_xor(m, o));
        if (lane == 0) atomicMax(&S.maxAbs, m);
        __syncthreads();

        // normal equations (:335-342)
        const double bound = (double)S.maxAbs * (double)S.maxAbs * (double)nInt;
        if (bound < 9007199254740992.0) {
            // Exact sums: lanes split the cells, the four waves split the 104 accumulators (26 each), FMA.
            // (v_mfma_f64_16x16x4_f64 was measured here: 64-cycle issue for 1024 MACs of which 416 are needed --
            //  6.7 ms against 3.7 ms for this form; FP64 MFMA has the vector rate on gfx950, so padding loses.)
            if (nC <= LSOP_RING_MAXC && S.maxAbs <= LSOP_MFMA_MAX_ABS && nInt < (1u << 17)) {
                lsop_gram_mfma(v, lsopRing, S.C32, nR, nC, S.G, tid);
            } else if (nC <= LSOP_RING_MAXC) {
                const int w = (int)GF_UNI(wave);                    // (a scalar: each wave runs ONE of the four loops)
                if (w == 0) lsop_gram_rows<0>(v, lsopRing, nR, nC, S.G, tid);
                else if (w == 1) lsop_gram_rows<1>(v, lsopRing, nR, nC, S.G, tid);
                else if (w == 2) lsop_gram_rows<2>(v, lsopRing, nR, nC, S.G, tid);
                else lsop_gram_rows<3>(v, lsopRing, nR, nC, S.G, tid);
            } else {
                if (wave == 0) lsop_gram_wave<0>(v, nC, nInt, S.G, lane);
                else if (wave == 1) lsop_gram_wave<1>(v, nC, nInt, S.G, lane);
                else if (wave == 2) lsop_gram_wave<2>(v, nC, nInt, S.G, lane);
                else lsop_gram_wave<3>(v, nC, nInt, S.G, lane);
            }
        } else if (tid < 104) {
            // inexact sums: one accumulator per thread, the reference's scan order, multiply and add rounded separately
            const int pi = PAIRS.i[tid], pj = PAIRS.j[tid];
            const int32_t oi = lsop_z_offset(pi, (int32_t)nC), oj = pj == 13 ? 0 : lsop_z_offset(pj, (int32_t)nC);
            double acc = 0.0;
            for (uint32_t r = 2; r < nR; r++) {
                const int32_t *row = v + (size_t)r * nC;
                for (uint32_t c = 2; c < nC - 2u; c++) {
                    const double zi = (double)row[(int32_t)c + oi];
                    const double zj = pj == 13 ? 1.0 : (double)row[(int32_t)c + oj];
                    acc = __dadd_rn(acc, pj == 13 ? zi : __dmul_rn(zi, zj));
                }
            }
            S.G[tid] = acc;
        }
        __syncthreads();

        // 13x13 bordered system, LU, solve (:353-378; LUDecomposition.java:70-134, 253-284) by wave 0
        if (wave == 0) lsop_lu_solve_wave(S, lane);
        __syncthreads();
        if (S.status != GF_K_OK) {
            if (tid == 0) a.status[t] = S.status;
            __syncthreads();
            continue;
        }
        float u[12];
#pragma unroll
        for (int i = 0; i < 12; i++) u[i] = S.u[i];
        if (tid < 16) a.coefs[t * 16 + tid] = tid == 0 ? (uint32_t)v[0] : tid <= 12 ? __float_as_uint(S.u[tid - 1]) : 0u;

        // interior stream (:248-272)
        const uint32_t wI = nC - 4u;
        {
            // four cells per thread in flight (their 52 loads are independent); row / column kept incrementally
            constexpr int U = 4;
            const uint32_t dr = 256u / wI, dc = 256u - dr * wI;
            uint32_t r = (uint32_t)tid / wI, c = (uint32_t)tid - r * wI;
            for (uint32_t e0 = tid; e0 < nInt; e0 += 256u * U) {
                uint32_t idx[U];
                float p[U];
#pragma unroll
                for (int k = 0; k < U; k++) {
                    const bool in = e0 + 256u * k < nInt;
                    idx[k] = in ? (r + 2u) * nC + c + 2u : 2u * nC + 2u;      // a harmless cell for the idle slots
                    r += dr;
                    c += dc;
                    if (c >= wI) { c -= wI; r++; }
                }
#pragma unroll
                for (int k = 0; k < U; k++) p[k] = lsop_predict12(u, v, idx[k], nC);
#pragma unroll
                for (int k = 0; k < U; k++) {
                    const uint32_t e = e0 + 256u * k;
                    if (e < nInt) res[nInit + e] = (int32_t)((uint32_t)v[idx[k]] - (uint32_t)lsop_round(p[k]));
                }
            }
        }
        if (tid == 0) a.status[t] = GF_K_OK;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// k_lsop_predict16 (round 5): the encoder's first kernel for terrain-sized tiles, with the tile ON THE CHIP.
//
// k_lsop_predict reads a tile three and a half times (the scan for max |v|, the rows through its LDS ring for the normal
// equations, thirteen scattered loads per cell for the interior residuals: 2.9 GB of HBM reads for 0.93 GB of tiles) and hands
// the residuals to k_canon_pack2 as int32 (0.94 GB written, read back twice for the histograms and for the text).  Here:
//   * the tile is read ONCE, coalesced, and kept in LDS as its two base-256 digit planes -- every tile the matrix-pipe form of
//     the normal equations takes (|v| <= 32,639) is two bytes per cell: 36 KB for 120 x 150 --; the max scan, the initialiser
//     residuals, the Gram matrix and the interior residuals all read LDS;
//   * the residuals go out as int16 (half the bytes, read once: see below), and
//   * the two histograms of CanonicalHuffman.countSymbols (:352-418) are counted here, while the residuals are in registers,
//     and handed over as a 2 KB record per tile: k_canon_pack2 starts at the code tables and reads the residuals once.
// A tile this kernel cannot take (|v| beyond the matrix-pipe guard, a residual beyond a halfword) is marked GF_K_LSOP_RETRY
// and goes through k_lsop_predict and the int32 form of k_canon_pack2, which run behind it and touch only such tiles.
// Same sums, same coefficients, same residuals: every operand of the normal equations is exact (see lsop_gram_mfma), the
// prediction is LsOptimalPredictor12's float32 sum in its order (:254-267).
// ------------------------------------------------------------------------------------------------
#ifndef GF_LSOP_PREDICT16_WGS
#define GF_LSOP_PREDICT16_WGS 4
#endif
constexpr int GF_K_LSOP16 = 0x7fff0010;          // internal: predicted by k_lsop_predict16 (residuals int16, histogram record written)
constexpr int GF_K_LSOP_RETRY = GF_K_LSOP_RETRY_; // internal: left to k_lsop_predict
#ifndef GF_LSOP_HR16
#define GF_LSOP_HR16 2
#endif
constexpr int LSOP_HR16 = GF_LSOP_HR16;          // replicas of the interior's histogram in k_lsop_predict16
constexpr int LSOP_HIST_REC_WORDS = 2 * CN_HIST + 8;   // per tile: the two histograms, then maxKind[2], nGap[2], 4 spare

struct LsopShared16 {
    double G[104];
    union {
        int32_t C32[27 * 32];                    // lsop_gram_mfma16: the digit Gram matrix, the 27 rows that hold digits
        struct {                                 // ... and afterwards the residual histograms
            uint32_t hist0[CN_HIST];
            uint32_t hist1[CN_HIST * LSOP_HR16];
        };
    };
    float u[12];
    uint32_t maxAbs, maxRes;
    uint32_t maxKind[2];
    int32_t status;
};

struct GfLsopPredict16Args {
    const int32_t *values;
    int16_t *residuals;        // per tile resStride halfwords: [initialisers | interior]
    size_t resStride;
    uint32_t *coefs;           // per tile 16 words: seed, 12 float bit patterns, 3 spare
    int32_t *status;           // GF_K_LSOP16 / GF_K_DECLINED / GF_K_LSOP_RETRY
    uint32_t *hist;            // per tile LSOP_HIST_REC_WORDS
    size_t nTiles;
    int nRows, nCols;
};

// halfword layout of a tile's residuals between k_lsop_predict16 and k_canon_pack2<true>: initialisers at 0, the interior on the
// next 16-byte boundary
__device__ __forceinline__ uint32_t lsop_interior_offset16(uint32_t nInit) { return (nInit + 7u) & ~7u; }

// The tile in LDS as its two balanced base-256 DIGIT PLANES (a byte per cell each): lo = (int8) v, hi = (v + 128) >> 8,
// v = 256 hi + lo -- the digits the matrix pipe multiplies (lsop_gram_mfma).  Kept as halfwords, every cell's digits were picked
// apart again for each of the 26 operand columns it belongs to: 16 reads, 16 adds and 12 byte permutes per MFMA, 1.58 of the
// kernel's 2.3 ms; from a plane a lane's sixteen cells are sixteen consecutive BYTES: five words and four v_alignbyte.
struct LsopPlanes {
    const int8_t *lo, *hi;
};
__device__ __forceinline__ int32_t lsop_plane_value(const LsopPlanes &P, uint32_t idx) { return ((int32_t)P.hi[idx] << 8) + (int32_t)P.lo[idx]; }

// the prediction of lsop_predict12 from the planes
__device__ __forceinline__ float lsop_predict12_p(const float *u, const LsopPlanes &P, uint32_t idx, uint32_t nC)
{
    auto V = [&](uint32_t i) -> float { return (float)lsop_plane_value(P, i); };
    float p = u[0] * V(idx - 1);
    p = p + u[1] * V(idx - nC - 1);
    p = p + u[2] * V(idx - nC);
    p = p + u[3] * V(idx - nC + 1);
    p = p + u[4] * V(idx - nC + 2);
    p = p + u[5] * V(idx - 2);
    p = p + u[6] * V(idx - nC - 2);
    p = p + u[7] * V(idx - 2 * nC - 2);
    p = p + u[8] * V(idx - 2 * nC - 1);
    p = p + u[9] * V(idx - 2 * nC);
    p = p + u[10] * V(idx - 2 * nC + 1);
    p = p + u[11] * V(idx - 2 * nC + 2);
    return p;
}

// lsop_gram_mfma with the whole tile in LDS as digit planes: no ring, no barrier per row -- the groups of 32 cells of all rows are
// dealt round-robin to the four waves.  Operand layout and the recombination are lsop_gram_mfma's: lane l holds column l & 31
// (13 lo digits, 13 hi digits, the ones, five idle) for the sixteen cells 16 (l >> 5) .. + 15 of the group.
template <int THREADS>
__device__ __forceinline__ void lsop_gram_mfma16(const LsopPlanes &P, int32_t *C32, uint32_t nR, uint32_t nC, double *G, int tid)
{
    const int lane = tid & 63;
    const uint32_t wave = gf_wave_id();
    const uint32_t col = (uint32_t)lane & 31u, h = (uint32_t)lane >> 5;
    const uint32_t zi = col < 13u ? col : col < 26u ? col - 13u : col == 26u ? 13u : 14u;
    int dr = 0, dc = 0;
    switch (zi) {
    case 1: dc = -1; break;
    case 2: dr = -1; dc = -1; break;
    case 3: dr = -1; break;
    case 4: dr = -1; dc = 1; break;
    case 5: dr = -1; dc = 2; break;
    case 6: dc = -2; break;
    case 7: dr = -1; dc = -2; break;
    case 8: dr = -2; dc = -2; break;
    case 9: dr = -2; dc = -1; break;
    case 10: dr = -2; break;
    case 11: dr = -2; dc = 1; break;
    case 12: dr = -2; dc = 2; break;
    default: break;
    }
    const int8_t *plane = (col >= 13u && col < 26u) ? P.hi : P.lo;                    // (the constant columns read the lo plane and drop it)
    const uint32_t constWord = zi == 13u ? 0x01010101u : 0u;
    const bool isConst = zi >= 13u;
    LsV16i acc = {};
    for (uint32_t i = (uint32_t)tid; i < 27u * 32u; i += (uint32_t)THREADS) C32[i] = 0;
    __syncthreads();
    const uint32_t wI = nC - 4u, nGroups = (wI + 31u) >> 5, nTurns = (nR - 2u) * nGroups;
    uint32_t r = 2u, g = wave;                                                        // turn = (r - 2) nGroups + g
    while (g >= nGroups) { g -= nGroups; r++; }
    for (uint32_t turn = wave; turn < nTurns; turn += (uint32_t)(THREADS / 64)) {
        const uint32_t c0 = 2u + 32u * g + 16u * h;                                   // the lane's first cell of the group
        const uint32_t byteAt = (uint32_t)((int)(r * nC) + dr * (int)nC + dc) + c0;   // ... in its plane
        const uint32_t *w = reinterpret_cast<const uint32_t *>(plane + (byteAt & ~3u));
        const uint32_t sh = byteAt & 3u;
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
        LsV4i x;
        x[0] = (int)(isConst ? constWord : __builtin_amdgcn_alignbyte(w1, w0, sh));
        x[1] = (int)(isConst ? constWord : __builtin_amdgcn_alignbyte(w2, w1, sh));
        x[2] = (int)(isConst ? constWord : __builtin_amdgcn_alignbyte(w3, w2, sh));
        x[3] = (int)(isConst ? constWord : __builtin_amdgcn_alignbyte(w4, w3, sh));
        if (g + 1u == nGroups) {                                                      // (wave-uniform) the row's last group
            const uint32_t nValid = c0 < nC - 2u ? min(16u, nC - 2u - c0) : 0u;
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) {
                const uint32_t have = nValid > 4u * q ? min(4u, nValid - 4u * q) : 0u;
                x[q] &= (int)(have >= 4u ? 0xFFFFFFFFu : (1u << (8u * have)) - 1u);
            }
        }
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, x, acc, 0, 0, 0);
        g += (uint32_t)(THREADS / 64);
        while (g >= nGroups) { g -= nGroups; r++; }
    }
#pragma unroll
    for (uint32_t q = 0; q < 16u; q++) {
        const uint32_t row = (q & 3u) + 8u * (q >> 2) + 4u * h;
        if (row < 27u) atomicAdd(&C32[row * 32u + col], acc[q]);
    }
    __syncthreads();
    if (tid < 104) {
        const int i = PAIRS.i[tid], j = PAIRS.j[tid];
        long long sum;
        if (j == 13) sum = 256ll * C32[(13 + i) * 32 + 26] + C32[i * 32 + 26];
        else
            sum = 65536ll * C32[(13 + i) * 32 + 13 + j] + 256ll * ((long long)C32[(13 + i) * 32 + j] + C32[i * 32 + 13 + j]) + C32[i * 32 + j];
        G[tid] = (double)sum;
    }
}

// OUT32: the same kernel behind gf_lsop12_predict_dev -- residuals as int32 in the public layout ([initialisers | interior], resStride
// ints per tile), no histograms, status GF_K_OK; only the tiles the matrix pipe cannot take are left to k_lsop_predict.
// THREADS: 256 (tiles of up to ~26 K cells: four workgroups per CU) or 1,024 (larger tiles -- BASELINE config 5's 256 x 256: the planes are
// 128 KB, one workgroup of sixteen waves per CU, the same sixteen waves a CU holds of the small form).
template <bool OUT32, int THREADS>
__global__ __launch_bounds__(THREADS, GF_LSOP_PREDICT16_WGS) void k_lsop_predict16(GfLsopPredict16Args a)
{
    __shared__ LsopShared16 S;
    extern __shared__ __attribute__((aligned(16))) int8_t lsopPlanes[];    // the tile's two digit planes, a byte per cell each (+ 64
                                                                           // behind either: the reads of a row's last group, masked)
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)gf_wave_id();
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const uint32_t nInit = lsop_n_init(nR, nC), nInt = lsop_n_interior(nR, nC);
    const uint32_t planeStride = (nCells + 64u + 15u) & ~15u;
    int8_t *const loP = lsopPlanes, *const hiP = lsopPlanes + planeStride;
    const LsopPlanes PL{loP, hiP};
    auto V = [&](uint32_t idx) -> uint32_t { return (uint32_t)lsop_plane_value(PL, idx); };

    GF_FOR_WG_TILE(t, a.nTiles) {
        const int32_t *__restrict__ v = a.values + t * (size_t)nCells;
        int16_t *__restrict__ res = a.residuals + t * a.resStride;                    // (OUT32: resStride counts halfwords here too)
        int32_t *__restrict__ res32 = reinterpret_cast<int32_t *>(res);
        if (tid == 0) { S.maxAbs = 0; S.maxRes = 0; S.status = GF_K_OK; S.maxKind[0] = 0; S.maxKind[1] = 0; }
        __syncthreads();

        // the tile, once: HBM -> halfwords in LDS, max |v| on the way (four cells per thread and turn where the tile starts on a
        // 16-byte boundary)
        uint32_t m = 0;
        if ((nCells & 3u) == 0u) {
            const GfU4 *v4 = reinterpret_cast<const GfU4 *>(v);
            uint32_t *lo32 = reinterpret_cast<uint32_t *>(loP), *hi32 = reinterpret_cast<uint32_t *>(hiP);
            for (uint32_t i = tid; i < (nCells >> 2); i += (uint32_t)THREADS) {
                const GfU4 q = v4[i];
                auto mag = [](uint32_t x) -> uint32_t { return (int32_t)x < 0 ? 0u - x : x; };
                m = max(max(m, mag(q.x)), max(max(mag(q.y), mag(q.z)), mag(q.w)));
                // byte 0 of the four cells; byte 1 of the four cells + 128
                lo32[i] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(q.w, q.z, 0x0c0c0400u), __builtin_amdgcn_perm(q.y, q.x, 0x0c0c0400u), 0x05040100u);
                const uint32_t hx = q.x + 128u, hy = q.y + 128u, hz = q.z + 128u, hw = q.w + 128u;
                hi32[i] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(hw, hz, 0x0c0c0501u), __builtin_amdgcn_perm(hy, hx, 0x0c0c0501u), 0x05040100u);
            }
        } else {
            for (uint32_t i = tid; i < nCells; i += (uint32_t)THREADS) {
                const int32_t x = v[i];
                m = max(m, x < 0 ? 0u - (uint32_t)x : (uint32_t)x);
                loP[i] = (int8_t)x;
                hiP[i] = (int8_t)((x + 128) >> 8);
            }
        }
        if (tid < 64) { loP[nCells + (uint32_t)tid] = 0; hiP[nCells + (uint32_t)tid] = 0; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, gf_lane_xor(m, o));
        if (lane == 0) atomicMax(&S.maxAbs, m);
        __syncthreads();
        if (S.maxAbs > LSOP_MFMA_MAX_ABS) {                               // (what a halfword and the matrix pipe's digits hold)
            if (tid == 0) a.status[t] = GF_K_LSOP_RETRY;
            __syncthreads();
            continue;
        }

        // normal equations (:335-342) on the matrix pipe, 13 x 13 system (:353-378) by wave 0
        lsop_gram_mfma16<THREADS>(PL, S.C32, nR, nC, S.G, tid);
        __syncthreads();
        if (wave == 0) lsop_lu_solve_wave(S, lane);
        __syncthreads();

        if (S.status != GF_K_OK) {
            if (tid == 0) a.status[t] = S.status;
            __syncthreads();
            continue;
        }
        float u[12];
#pragma unroll
        for (int i = 0; i < 12; i++) u[i] = S.u[i];
        if (tid < 16) a.coefs[t * 16 + tid] = tid == 0 ? (uint32_t)v[0] : tid <= 12 ? __float_as_uint(S.u[tid - 1]) : 0u;
        for (int i = tid; i < CN_HIST * (1 + LSOP_HR16); i += THREADS) S.hist0[i] = 0;          // (C32 is done with; hist1 follows hist0)
        __syncthreads();

        // a residual into its stream's histogram as CanonicalHuffman.countSymbols classifies it (:352-418; a halfword has kinds 0-4)
        // (the interior's histogram in LSOP_HR16 replicas, replica-major -- a tile's residuals crowd a dozen bins around zero,
        // and lanes that add to the same word take turns --; the few hundred initialisers in one)
        const uint32_t rep = (uint32_t)lane & (LSOP_HR16 - 1);
        uint32_t mk0 = 0, mk1 = 0, maxRes = 0;
        auto count = [&](uint32_t *hist, uint32_t x, uint32_t &mk, bool replicated) {
            uint32_t *h = hist + (replicated ? rep * CN_HIST : 0u);
            if (x + 128u < 256u) { atomicAdd(h + (x + 128u), 1u); return; }
            uint32_t kind;
            const uint32_t target = cn_classify_count(x, &kind);
            atomicAdd(h + target, 1u);
            if (kind >= 1u && kind <= 3u) atomicAdd(h + CN_ESC2, kind);
            else if (kind >= 4u && kind <= 6u) atomicAdd(h + CN_ESC1, kind - 3u);
            mk = max(mk, kind == 7u ? 0u : kind);
        };
        auto magOf = [](uint32_t x) -> uint32_t { return (int32_t)x < 0 ? ~x : x; };   // x fits a halfword iff magOf(x) <= 32767

        // initialiser stream (:143-209)
        for (uint32_t k = tid; k < nInit; k += THREADS) {
            uint32_t kind;
            const uint32_t idx = lsop_init_cell(k, nR, nC, &kind);
            const uint32_t x = V(idx);
            uint32_t r;
            if (kind == 0) r = x - V(idx - 1);
            else if (kind == 1) r = x - V(idx - nC);
            else r = x - (V(idx - 1) + V(idx - nC) - V(idx - nC - 1));
            if constexpr (OUT32) res32[k] = (int32_t)r;
            else {
                res[k] = (int16_t)r;
                maxRes = max(maxRes, magOf(r));
                count(S.hist0, r, mk0, false);
            }
        }

        // interior stream (:248-272): EIGHT neighbouring elements per thread and turn -- inside a row (all but the one item in 19
        // that straddles a row's end) their 104 neighbours are 34 distinct cells: ten of the row, twelve of each row above, read
        // and converted once --, one 16-byte store (the interior starts on a 16-byte boundary of the halfword layout)
        {
            GfU4 *__restrict__ ri128 = reinterpret_cast<GfU4 *>(res + lsop_interior_offset16(nInit));
            const uint32_t wI = nC - 4u;
            const uint32_t magicW = (uint32_t)(((1ull << 32) + wI - 1u) / wI);          // e / wI = umulhi(e, magicW) for e wI < 2^32
            auto residualOf = [&](uint32_t e) -> uint32_t {
                const uint32_t r = __umulhi(e, magicW), c = e - r * wI;
                const uint32_t idx = (r + 2u) * nC + c + 2u;
                return V(idx) - (uint32_t)lsop_round_f32(lsop_predict12_p(u, PL, idx, nC));
            };
            constexpr int IT = 8;                                                        // elements per item
            const uint32_t nItems = (nInt + IT - 1u) / IT;
            for (uint32_t k = (uint32_t)tid; k < nItems; k += (uint32_t)THREADS) {
                const uint32_t e = (uint32_t)IT * k;
                const uint32_t r = __umulhi(e, magicW), c = e - r * wI;
                uint32_t x[IT];
                if (c + (IT - 1u) < wI) {                                                // (then e + IT - 1 < nInt as well)
                    const uint32_t i0 = (r + 2u) * nC + c, i1 = i0 - nC, i2 = i1 - nC;   // column c = the first cell's column - 2
                    float a[IT + 2], b[IT + 4], d[IT + 4];
                    int32_t own[IT];
#pragma unroll
                    for (int q = 0; q < IT + 2; q++) {
                        const int32_t val = lsop_plane_value(PL, i0 + (uint32_t)q);
                        a[q] = (float)val;
                        if (q >= 2) own[q - 2] = val;
                    }
#pragma unroll
                    for (int q = 0; q < IT + 4; q++) { b[q] = (float)lsop_plane_value(PL, i1 + (uint32_t)q); d[q] = (float)lsop_plane_value(PL, i2 + (uint32_t)q); }
#pragma unroll
                    for (int q = 0; q < IT; q++) {
                        // the cell at column c + 2 + q: z1 = W, z2 = NW, z3 = N, z4 = NE, z5 = NEE, z6 = WW, z7 = NWW, z8..z12 = the row two above
                        float pr = u[0] * a[q + 1];
                        pr = pr + u[1] * b[q + 1];
                        pr = pr + u[2] * b[q + 2];
                        pr = pr + u[3] * b[q + 3];
                        pr = pr + u[4] * b[q + 4];
                        pr = pr + u[5] * a[q];
                        pr = pr + u[6] * b[q];
                        pr = pr + u[7] * d[q];
                        pr = pr + u[8] * d[q + 1];
                        pr = pr + u[9] * d[q + 2];
                        pr = pr + u[10] * d[q + 3];
                        pr = pr + u[11] * d[q + 4];
                        x[q] = (uint32_t)own[q] - (uint32_t)lsop_round_f32(pr);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < IT; q++) x[q] = e + (uint32_t)q < nInt ? residualOf(e + (uint32_t)q) : 0u;
                }
                if constexpr (OUT32) {
                    int32_t *ro = res32 + nInit + e;
                    if (e + (uint32_t)IT <= nInt) {
                        GfU4 w0, w1;
                        w0.x = x[0]; w0.y = x[1]; w0.z = x[2]; w0.w = x[3];
                        w1.x = x[4]; w1.y = x[5]; w1.z = x[6]; w1.w = x[7];
                        *reinterpret_cast<GfU4 *>(ro) = w0;
                        *reinterpret_cast<GfU4 *>(ro + 4) = w1;
                    } else {
#pragma unroll
                        for (int q = 0; q < IT; q++)
                            if (e + (uint32_t)q < nInt) ro[q] = (int32_t)x[q];
                    }
                    continue;
                }
#pragma unroll
                for (int q = 0; q < IT; q++) {
                    if (e + (uint32_t)q < nInt) {
                        maxRes = max(maxRes, magOf(x[q]));
                        count(S.hist1, x[q], mk1, true);
                    }
                }
                GfU4 w;
                w.x = (x[0] & 0xffffu) | (x[1] << 16);
                w.y = (x[2] & 0xffffu) | (x[3] << 16);
                w.z = (x[4] & 0xffffu) | (x[5] << 16);
                w.w = (x[6] & 0xffffu) | (x[7] << 16);
                ri128[k] = w;                                                            // (the region has room for the padding halfwords)
            }
        }
        if constexpr (OUT32) {
            if (tid == 0) a.status[t] = GF_K_OK;
            __syncthreads();
            continue;
        }
        if (mk0) atomicMax(&S.maxKind[0], mk0);
        if (mk1) atomicMax(&S.maxKind[1], mk1);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) maxRes = max(maxRes, gf_lane_xor(maxRes, o));
        if (lane == 0) atomicMax(&S.maxRes, maxRes);
        __syncthreads();

        // the record for k_canon_pack2: the histograms as its own first pass leaves them (end-of-text counted once, nothing
        // beyond the alphabet), the largest escape kind per stream, no values in the -8333608 gap (a halfword holds none)
        uint32_t *rec = a.hist + t * (size_t)LSOP_HIST_REC_WORDS;
        for (int i = tid; i < 2 * CN_HIST; i += THREADS) {
            const int p = i / CN_HIST, sym = i - p * CN_HIST;
            uint32_t sum = 0;
            if (p == 0) sum = S.hist0[sym];
            else {
#pragma unroll
                for (int k = 0; k < LSOP_HR16; k++) sum += S.hist1[k * CN_HIST + sym];
            }
            if (sym == CN_EOT) sum = 1;
            if (sym >= CN_SYMS) sum = 0;
            rec[i] = sum;
        }
        if (tid < 8) rec[2 * CN_HIST + tid] = tid < 2 ? S.maxKind[tid] : 0u;
        if (tid == 0) a.status[t] = S.maxRes > 32767u ? GF_K_LSOP_RETRY : GF_K_LSOP16;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// k_canon_pack2: header + two canonical-Huffman streams in one bit store
// ------------------------------------------------------------------------------------------------

struct Pack2Persist {
    uint32_t hist[2][CN_HIST];
    uint32_t tab[2][CN_HIST];
    uint32_t img[2][CN_IMG_WORDS];
    unsigned long long textBits[2];
    uint32_t imgBits[2];
    uint32_t maxLen[2];
    uint32_t maxKind[2];
    uint32_t nGap[2];
    uint32_t waveSum[ENC_WAVES];
    uint32_t pmLock;                            // cn_pm_acquire
};

struct Pack2Trees {
    CanonScratch tree[2];
    CanonPM pm;                                 // shared: see CanonPM
};
union Pack2Union {
    uint32_t histR[2][CN_HIST * HIST_R];
    Pack2Trees b;
    alignas(16) uint32_t win[WIN_WORDS + WIN_SLACK];
};

__device__ __forceinline__ uint32_t p2_elem_max_bits(uint32_t maxLen, uint32_t maxKind)
{
    if (maxKind == 0u) return maxLen;
    if (maxKind <= 3u) return maxLen + maxKind * (maxLen + 2u);
    return maxLen + 3u * (maxLen + 8u);
}

// text of one stream held as an int array
// elements [begin, end) of a residual array through wave-private bit windows (gvrs_encode_common.h: wave_windows_*), eight
// consecutive elements per lane; false = a wave's share did not fit its window, nothing was written
template <class T>                                     // int32_t, or int16_t behind k_lsop_predict16
__device__ bool p2_pack_array_waves(const T *__restrict__ arr, uint32_t begin, uint32_t end, const uint32_t *tab,
                                    uint32_t *win, uint32_t *__restrict__ out32, uint32_t *waveSum, PackState &ps,
                                    uint32_t slotWords)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = gf_wave_id();
    const uint32_t carryWord = wave_windows_begin(win, waveSum);
    const uint32_t quarter = (((end - begin + ENC_WAVES - 1) / ENC_WAVES) + CPT - 1) / CPT * CPT;
    const uint32_t segBegin = min(end, begin + wave * quarter), segEnd = min(end, segBegin + quarter);
    uint32_t *wwin = win + wave * WAVE_WIN;
    uint32_t bits = 0;
    bool fits = true;
    for (uint32_t base = segBegin; base < segEnd; base += 64u * CPT) {
        const uint32_t i0 = base + lane * CPT;
        uint32_t cl[CPT], xs[CPT];
        uint32_t myBits = 0, wide = 0;
#pragma unroll
        for (int j = 0; j < CPT; j++) xs[j] = i0 + j < segEnd ? (uint32_t)(int32_t)arr[i0 + j] : 0u;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const bool emit = i0 + j < segEnd;
            const uint32_t x = xs[j];
            const bool narrow = x + 128u < 256u;
            const uint32_t e = emit ? (narrow ? tab[x + 128u] : 0u) : 0u;
            cl[j] = e;
            myBits += e >> 16;
            if (emit && !narrow) wide |= 1u << j;
        }
        if (wide) {
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if ((wide >> j) & 1u) myBits += cn_value_bits(tab, xs[j]);
        }
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

template <class T>
__device__ void p2_pack_array(const T *__restrict__ arr, uint32_t n, const uint32_t *tab, uint32_t elemMaxBits,
                              uint32_t *win, uint32_t *__restrict__ out32, uint32_t *waveSum, PackState &ps)
{
    const uint32_t tid = threadIdx.x;
    uint32_t E = 4;
    while (E > 1 && (uint64_t)ENC_THREADS * E * elemMaxBits > (uint64_t)(WIN_WORDS - 2) * 32u) E >>= 1;
    uint32_t active = ENC_THREADS;
    if ((uint64_t)ENC_THREADS * elemMaxBits > (uint64_t)(WIN_WORDS - 2) * 32u)
        active = max(1u, (uint32_t)(((uint64_t)(WIN_WORDS - 2) * 32u) / elemMaxBits));
    const uint32_t chunkElems = active * E;
    for (uint32_t chunk = 0; chunk < n; chunk += chunkElems) {
        uint32_t xs[4];
        uint32_t myBits = 0;
        const uint32_t s0 = chunk + tid * E;
        const uint32_t cEnd = min(n, chunk + chunkElems);
#pragma unroll
        for (uint32_t e = 0; e < 4; e++) {
            xs[e] = 0;
            const uint32_t s = s0 + e;
            if (e < E && tid < active && s < cEnd) {
                xs[e] = (uint32_t)(int32_t)arr[s];
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

// appends nbits of the image (a bit string starting at bit 0 of img) and then the end-of-text code eot to the window
__device__ void p2_append_bits(const uint32_t *img, uint32_t nbits, uint32_t *win, uint32_t *__restrict__ out32, PackState &ps)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t words = (nbits + 31u) >> 5;
    const uint32_t base = ps.bitBase - ps.wordBase * 32u;
    for (uint32_t i = tid; i < words; i += ENC_THREADS) {
        const uint32_t nb = min(32u, nbits - i * 32u);
        const uint32_t w = nb < 32u ? img[i] & ((1u << nb) - 1u) : img[i];
        cn_img_or(win, base + i * 32u, w, nb);
    }
    __syncthreads();
    ps.bitBase += nbits;
    window_flush(win, out32, ps);
}

// ------------------------------------------------------------------------------------------------
// k_lsop_value_crc: LsHeader.computeChecksum (lsop/LsHeader.java:391-406) -- the CRC-32C (util/GridfourCRC32C.java:160-185) of a
// tile's values as little-endian bytes, i.e. of the tile as it lies in memory -- for LsEncoder12.setValueChecksumEnabled
// (:117-119, default off).  One wave per tile: lane l takes the l-th of 64 equal runs of cells byte by byte through the
// polynomial's table (in LDS, made by the workgroup), and the runs' checksums are joined by the CRC's linearity:
//   crc(A || B) = crc(A) * x^(8 |B|)  xor  crc(B)    in GF(2)[x] modulo the (reflected) polynomial,
// so the tile's checksum is the XOR over the lanes of crc(run) * x^(8 bytes behind the run).  The multiplications are 32 steps
// of shift-and-conditional-xor each, the powers by square and multiply.  Word 13 of the tile's coefficient record receives it.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t CRC32C_POLY = 0x82F63B78u;
__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b)        // a * b mod P, operands and result bit-reflected
{
    uint32_t p = 0;
#pragma unroll 1
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        p ^= (a & m) ? b : 0u;
        b = (b & 1u) ? (b >> 1) ^ CRC32C_POLY : b >> 1;
    }
    return p;
}
__device__ __forceinline__ uint32_t crc_xpow8n(uint32_t nBytes)               // x^(8 nBytes) mod P
{
    uint32_t p = 0x80000000u, sq = 0x00800000u;                               // x^0; x^8
#pragma unroll 1
    for (uint32_t n = nBytes; n; n >>= 1) {
        if (n & 1u) p = crc_mulmod(sq, p);
        sq = crc_mulmod(sq, sq);
    }
    return p;
}
__global__ __launch_bounds__(256) void k_lsop_value_crc(const int32_t *__restrict__ values, uint32_t nCells, size_t nTiles,
                                                        const int32_t *__restrict__ inStatus, uint32_t *__restrict__ coefs)
{
    __shared__ uint32_t table[256];
    {
        uint32_t c = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? CRC32C_POLY : 0u);
        table[threadIdx.x] = c;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const size_t t = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t >= nTiles || (inStatus && inStatus[t] != GF_K_OK)) return;
    const uint32_t *__restrict__ v = reinterpret_cast<const uint32_t *>(values) + t * (size_t)nCells;
    const uint32_t per = (nCells + 63u) / 64u, begin = min(nCells, lane * per), end = min(nCells, begin + per);
    uint32_t crc = 0xffffffffu;
    for (uint32_t i = begin; i < end; i++) {
        uint32_t w = v[i];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            crc = table[(crc ^ w) & 0xffu] ^ (crc >> 8);
            w >>= 8;
        }
    }
    crc ^= 0xffffffffu;                                                       // the run's own checksum (an empty run: 0)
    uint32_t part = crc_mulmod(crc_xpow8n((nCells - end) * 4u), crc);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part ^= gf_lane_xor(part, o);
    if (lane == 0) coefs[t * 16 + 13] = part;
}

struct GfPack2Args {
    const int32_t *residuals;  // per tile resStride ints: stream 0 (n0), stream 1 (n1)
    size_t resStride;
    const uint32_t *coefs;     // per tile 16 words: seed, 12 float bit patterns
    const int32_t *inStatus;   // status of the predict stage (tiles it declined are passed through)
    uint8_t *out;
    size_t slotStride;
    uint32_t *lengths;
    int32_t *status;
    size_t nTiles;
    uint32_t n0, n1;
    int codecIndex;
    int valueChecksum;         // LsEncoder12.setValueChecksumEnabled: word 13 of the coefficient record goes behind the header
    const uint32_t *hist;      // R16: the histogram records of k_lsop_predict16 (LSOP_HIST_REC_WORDS per tile)
};

constexpr uint32_t LSOP_HEADER_BYTES = 55;     // LsHeader.packHeader :219-222 for the canonical container: 7 + 12*4 (+ 4: the checksum)

// R16 (round 5): the tiles k_lsop_predict16 took -- residuals as halfwords, the histograms already counted (a.hist): the kernel
// starts at the code tables and reads the residuals once.  The int32 instantiation takes every other tile (predicted by
// k_lsop_predict: inStatus GF_K_OK; or declined) and leaves the halfword tiles alone.
template <bool R16>
__global__ __launch_bounds__(ENC_THREADS, GF_LSOP_PACK2_WGS) void k_canon_pack2(GfPack2Args a)
{
    __shared__ Pack2Persist P;
    __shared__ Pack2Union S;
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)gf_wave_id();
    using ResT = std::conditional_t<R16, int16_t, int32_t>;

    GF_FOR_WG_TILE(t, a.nTiles) {                                         // no tile loop: see gvrs_kernels.h
        const int32_t inStatus = a.inStatus[t];
        if (R16 ? inStatus != GF_K_LSOP16 : inStatus == GF_K_LSOP16) continue;            // the other instantiation's tile
        if (inStatus != (R16 ? GF_K_LSOP16 : GF_K_OK)) {
            if (tid == 0) { a.lengths[t] = 0; a.status[t] = inStatus; }
            __syncthreads();
            continue;
        }
        const ResT *__restrict__ res = reinterpret_cast<const ResT *>(a.residuals) + t * (R16 ? 2 * a.resStride : a.resStride);
        uint32_t *__restrict__ out32 = reinterpret_cast<uint32_t *>(a.out + t * a.slotStride);

        if (tid < 2) { P.maxKind[tid] = 0; P.nGap[tid] = 0; }
        if (tid == 0) P.pmLock = 0;
        if constexpr (R16) {
            const uint32_t *rec = a.hist + t * (size_t)LSOP_HIST_REC_WORDS;
            __syncthreads();
            for (int i = tid; i < 2 * CN_HIST; i += ENC_THREADS) (&P.hist[0][0])[i] = rec[i];
            if (tid < 2) P.maxKind[tid] = rec[2 * CN_HIST + tid];
        } else {
        for (int i = tid; i < 2 * CN_HIST * HIST_R; i += ENC_THREADS) (&S.histR[0][0])[i] = 0;
        __syncthreads();
        const uint32_t rep = (uint32_t)lane & (HIST_R - 1);
        for (int sidx = 0; sidx < 2; sidx++) {
            const ResT *arr = sidx == 0 ? res : res + a.n0;
            const uint32_t n = sidx == 0 ? a.n0 : a.n1;
            uint32_t *h = &S.histR[sidx][rep];
            uint32_t mk = 0, gaps = 0;
            for (uint32_t i0 = tid; i0 < n; i0 += 4u * ENC_THREADS) {       // four independent loads in flight
                uint32_t xv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) xv[k] = i0 + k * ENC_THREADS < n ? (uint32_t)(int32_t)arr[i0 + k * ENC_THREADS] : 0u;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (i0 + k * ENC_THREADS >= n) break;
                    const uint32_t x = xv[k];
                    if (x + 128u < 256u) { atomicAdd(h + (x + 128u) * HIST_R, 1u); continue; }
                    uint32_t kind;
                    const uint32_t target = cn_classify_count(x, &kind);
                    atomicAdd(h + target * HIST_R, 1u);
                    if (kind >= 1u && kind <= 3u) atomicAdd(h + CN_ESC2 * HIST_R, kind);
                    else if (kind >= 4u && kind <= 6u) atomicAdd(h + CN_ESC1 * HIST_R, kind - 3u);
                    if (cn_is_gap(x)) { gaps++; kind = 6u; }
                    mk = max(mk, kind == 7u ? 0u : kind);
                }
            }
            if (mk) atomicMax(&P.maxKind[sidx], mk);
            if (gaps) atomicAdd(&P.nGap[sidx], gaps);
        }
        __syncthreads();
        for (int i = tid; i < 2 * CN_HIST; i += ENC_THREADS) {
            const int p = i / CN_HIST, s = i - p * CN_HIST;
            const uint32_t *h = &S.histR[p][0] + (size_t)s * HIST_R;
            uint32_t sum = 0;
#pragma unroll
            for (int k = 0; k < HIST_R; k++) sum += h[k];
            if (s == CN_EOT) sum = 1;
            if (s >= CN_SYMS) sum = 0;
            P.hist[p][s] = sum;
        }
        }
        for (int i = tid; i < 2 * CN_IMG_WORDS; i += ENC_THREADS) (&P.img[0][0])[i] = 0;
        __syncthreads();
        if (wave < 2) {
            const CanonBuilt B = cn_build(S.b.tree[wave], S.b.pm, &P.pmLock, P.hist[wave], P.nGap[wave], P.tab[wave], P.img[wave], lane);
            if (lane == 0) { P.imgBits[wave] = B.imgBits; P.maxLen[wave] = B.maxLen; P.textBits[wave] = B.textBits; }
        }
        __syncthreads();

        const uint32_t headerBytes = LSOP_HEADER_BYTES + (a.valueChecksum ? 4u : 0u);
        const unsigned long long totalBits = 8ull * headerBytes + P.imgBits[0] + P.textBits[0] + P.imgBits[1] + P.textBits[1];
        const unsigned long long bytes = (totalBits + 7) >> 3;
        if (tid == 0) {
            a.lengths[t] = (uint32_t)min(bytes, 0xffffffffull);
            a.status[t] = bytes > a.slotStride ? GF_K_OVERFLOW : GF_K_OK;
        }
        if (bytes > a.slotStride) { __syncthreads(); continue; }

        // header bytes (LsHeader.packHeader :224-246): codec index, type 2 | revision flag, 12, seed, 12 floats
        for (int i = tid; i < WIN_WORDS + WIN_SLACK; i += ENC_THREADS) S.win[i] = 0;
        __syncthreads();
        if (tid < 14) {
            const uint32_t *cf = a.coefs + t * 16;
            // byte stream: [codec][0x42, 0xC2 with the value checksum][12] then 13 little-endian words (seed, coefficients) from
            // byte 3 on, and the checksum as a 14th (LsHeader.packHeader :245-262)
            if (tid == 0) atomicOr(&S.win[0], ((uint32_t)a.codecIndex & 0xffu) | ((a.valueChecksum ? 0xC2u : 0x42u) << 8) | (12u << 16));
            if (tid < 13 || a.valueChecksum) {
                const uint32_t w = cf[tid];
                atomicOr(&S.win[tid], w << 24);
                atomicOr(&S.win[tid + 1], w >> 8);
            }
        }
        __syncthreads();
        PackState ps;
        ps.bitBase = 8u * headerBytes;
        ps.wordBase = 0;
        window_flush(S.win, out32, ps);
        for (int sidx = 0; sidx < 2; sidx++) {
            const ResT *arr = sidx == 0 ? res : res + (R16 ? lsop_interior_offset16(a.n0) : a.n0);
            const uint32_t n = sidx == 0 ? a.n0 : a.n1;
            p2_append_bits(P.img[sidx], P.imgBits[sidx], S.win, out32, ps);
            const uint32_t emb = max(1u, p2_elem_max_bits(P.maxLen[sidx], P.maxKind[sidx]));
            {
                // in as many element ranges as the text's bit count asks for; a range that does not fit goes the old way
                const unsigned long long tb = P.textBits[sidx];
                const uint64_t want = (tb + (tb >> 2)) / ENC_WAVES;
                const uint32_t nRanges = (uint32_t)min((uint64_t)1024, want / WAVE_WIN_BITS + 1u);
                const uint32_t per = (((n + nRanges - 1) / nRanges) + (CPT * ENC_WAVES) - 1) / (CPT * ENC_WAVES) * (CPT * ENC_WAVES);
                for (uint32_t b = 0; b < n; b += per) {
                    const uint32_t e2 = min(n, b + per);
                    if (!p2_pack_array_waves(arr, b, e2, P.tab[sidx], S.win, out32, P.waveSum, ps, (uint32_t)(a.slotStride >> 2)))
                        p2_pack_array(arr + b, e2 - b, P.tab[sidx], emb, S.win, out32, P.waveSum, ps);
                }
            }
            const uint32_t e = P.tab[sidx][CN_EOT];
            if (tid == 0) cn_img_or(S.win, ps.bitBase - ps.wordBase * 32u, e & 0xffffu, e >> 16);
            __syncthreads();
            ps.bitBase += e >> 16;
            window_flush(S.win, out32, ps);
        }
        {
            const uint32_t remBits = ps.bitBase - ps.wordBase * 32u;
            const uint32_t remWords = (remBits + 31u) >> 5;
            const uint32_t slotWords = (uint32_t)(a.slotStride >> 2);
            for (uint32_t j = tid; j < remWords; j += ENC_THREADS)
                if (ps.wordBase + j < slotWords) out32[ps.wordBase + j] = S.win[j];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// k_lsop_reconstruct: one tile per wave
// ------------------------------------------------------------------------------------------------

struct GfLsopReconArgs {
    const int32_t *residuals;
    size_t resStride;
    const uint32_t *coefs;
    const int32_t *inStatus;   // may be null; tiles with a status other than GF_K_OK are skipped
    int32_t *values;
    int32_t *status;
    size_t nTiles;
    int nRows, nCols;
    bool planes;               // word GF_LSOP_FMT_WORD of a tile's coefficient record is valid: 1 = its interior residuals are a byte
                               // plane (k_lsop_reconstruct_plane's tile; the other kernels pass it by)
};

__global__ __launch_bounds__(256) void k_lsop_reconstruct_global(GfLsopReconArgs a)
{
    const int lane = threadIdx.x & 63;
    const size_t wavesPerGrid = (size_t)gridDim.x * 4, wid = (size_t)blockIdx.x * 4 + gf_wave_id();
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const uint32_t nInit = lsop_n_init(nR, nC);
    const uint32_t wI = nC - 4u;

    for (size_t t = wid; t < a.nTiles; t += wavesPerGrid) {
        if (a.planes && a.coefs[t * 16 + GF_LSOP_FMT_WORD] == 1u) continue;
        if (a.inStatus && a.inStatus[t] != GF_K_OK) {
            if (lane == 0) a.status[t] = a.inStatus[t];
            continue;
        }
        const int32_t *__restrict__ res = a.residuals + t * a.resStride;
        const int32_t *__restrict__ inter = res + nInit;
        int32_t *v = a.values + t * (size_t)nCells;
        const uint32_t *cf = a.coefs + t * 16;
        const uint32_t seed = cf[0];
        float u[12];
#pragma unroll
        for (int i = 0; i < 12; i++) u[i] = __uint_as_float(cf[1 + i]);

        // LsDecoder12.unpackInitializers :186-221 as prefix sums
        // row 0
        {
            uint32_t carry = seed;
            if (lane == 0) v[0] = (int32_t)seed;
            for (uint32_t c0 = 1; c0 < nC; c0 += 64) {
                const uint32_t c = c0 + lane;
                const uint32_t x = c < nC ? (uint32_t)res[c - 1] : 0u;
                const uint32_t incl = gf_wave_incl_scan(x) + carry;
                if (c < nC) v[c] = (int32_t)incl;
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        // column 0
        {
            uint32_t carry = seed;
            for (uint32_t r0 = 1; r0 < nR; r0 += 64) {
                const uint32_t r = r0 + lane;
                const uint32_t x = r < nR ? (uint32_t)res[nC - 1 + r - 1] : 0u;
                const uint32_t incl = gf_wave_incl_scan(x) + carry;
                if (r < nR) v[(size_t)r * nC] = (int32_t)incl;
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // row 1: v[1][c] - v[0][c] is a running sum of the triangle residuals, starting from v[1][0] - v[0][0]
        {
            uint32_t carry = (uint32_t)v[nC] - (uint32_t)v[0];
            const uint32_t base = nC - 1u + nR - 1u;
            for (uint32_t c0 = 1; c0 < nC; c0 += 64) {
                const uint32_t c = c0 + lane;
                const uint32_t x = c < nC ? (uint32_t)res[base + c - 1] : 0u;
                const uint32_t incl = gf_wave_incl_scan(x) + carry;
                if (c < nC) v[nC + c] = (int32_t)(incl + (uint32_t)v[c]);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // column 1, rows 2..: v[r][1] - v[r][0] is a running sum, starting from v[1][1] - v[1][0]
        {
            uint32_t carry = (uint32_t)v[nC + 1] - (uint32_t)v[nC];
            const uint32_t base = 2u * (nC - 1u) + nR - 1u;
            for (uint32_t r0 = 2; r0 < nR; r0 += 64) {
                const uint32_t r = r0 + lane;
                const uint32_t x = r < nR ? (uint32_t)res[base + r - 2] : 0u;
                const uint32_t incl = gf_wave_incl_scan(x) + carry;
                if (r < nR) v[(size_t)r * nC + 1] = (int32_t)(incl + (uint32_t)v[(size_t)r * nC]);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();

        // interior and the last two columns: LsDecoder12.unpackInterior :311-383 as a wavefront, step = c + 3 r
        const uint32_t tailBase = 2u * (nC - 1u) + nR - 1u + nR - 2u;
        const uint32_t sEnd = (nC - 1u) + 3u * (nR - 1u);
        for (uint32_t s = 8; s <= sEnd; s++) {
            for (uint32_t r0 = 2; r0 < nR; r0 += 64) {
                const uint32_t r = r0 + lane;
                if (r < nR && s >= 3u * r + 2u && s - 3u * r <= nC - 1u) {
                    const uint32_t c = s - 3u * r;
                    const uint32_t idx = r * nC + c;
                    uint32_t val;
                    if (c <= nC - 3u) {
                        const int32_t est = lsop_round(lsop_predict12(u, v, idx, nC));
                        val = (uint32_t)est + (uint32_t)inter[(r - 2u) * wI + (c - 2u)];
                    } else {
                        const uint32_t x = (uint32_t)res[tailBase + 2u * (r - 2u) + (c - (nC - 2u))];
                        val = x + ((uint32_t)v[idx - 1] + (uint32_t)v[idx - nC] - (uint32_t)v[idx - nC - 1]);
                    }
                    v[idx] = (int32_t)val;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) a.status[t] = GF_K_OK;
    }
}


// The same reconstruction, one tile per 64-thread workgroup, for global memory that only ever sees whole-row pieces.
//
// A wave walks the tile in bands of 64 rows, one lane per row; lane l is 3 columns behind lane l-1 (step s: lane l works
// on column c = s - 3 l, columns 0 and 1 included: there the "computed" value is the border value, so that every row
// looks the same to the row below).
//
//   * the neighbourhood lives in registers: a lane keeps columns c-2..c+2 of the row above (A) and of the row two above
//     (B) as two 5-register windows that shift by one per step.  The value entering A is what lane l-1 produced in the
//     previous step, the value entering B is the middle of lane l-1's own A window -- both arrive with one DPP
//     wave_shr:1 each; lane 0 takes them from the full-row LDS buffers that lanes 62 / 63 of the previous band filled.
//   * residuals come in and values go out through one LDS staging ring per row (64 slots, slot = s & 63, a value
//     overwrites the residual it was made from).  The ring is filled and drained a 32-step ROUND at a time and 16 rows
//     per 8 steps, cooperatively: a half-wave moves the 32 consecutive residuals / values of one row (128 contiguous
//     bytes of global memory) per instruction.  Residuals of round R+1 are requested while round R runs and land in the
//     slots that round R-1's values have just left.
__device__ __forceinline__ uint32_t lsop_from_lane_above(uint32_t lane0Value, uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0Value, (int)x, 0x138, 0xf, 0xf, false);   // wave_shr:1
}

#ifndef RECON_WAVES_PER_SIMD
#define RECON_WAVES_PER_SIMD 4
#endif
#ifndef RECON_ROUND
#define RECON_ROUND 16
#endif
template <int ROUND>
__global__ __launch_bounds__(64, RECON_WAVES_PER_SIMD) void k_lsop_reconstruct(GfLsopReconArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t reconLds[];
    constexpr uint32_t RING = 2u * ROUND;                  // slots per staging ring: two rounds
    constexpr uint32_t RS = RING + 1u;                     // words per ring (+1: the lanes of a step hit different banks)
    constexpr uint32_t RPI = 64u / ROUND;                  // rows per cooperative instruction
    constexpr uint32_t SUBS = ROUND / 8u;                  // 8-step sub-rounds per round
    constexpr uint32_t SUBROWS = 64u / SUBS;               // rows staged per sub-round
    const int lane = threadIdx.x;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const uint32_t nInit = lsop_n_init(nR, nC);
    const uint32_t wI = nC - 4u;
    uint32_t *stage = reconLds;                            // [64][RS]
    uint32_t *rows = reconLds + 64u * RS;                  // 2 full rows: the two rows above the band
    const uint32_t half = (uint32_t)lane / ROUND, j32 = (uint32_t)lane % ROUND;     // row within the instruction, step within the round

    // (a grid-stride loop, although the launcher gives every tile its own workgroup: without the loop -- GF_FOR_WG_TILE -- the
    // compiler allocates 79 instead of 103 registers and the kernel is slower, 1.28 against 1.18 ms on the bench batch)
    for (size_t t = blockIdx.x; t < a.nTiles; t += gridDim.x) {
        if (a.planes && a.coefs[t * 16 + GF_LSOP_FMT_WORD] == 1u) continue;
        if (a.inStatus && a.inStatus[t] != GF_K_OK) {
            if (lane == 0) a.status[t] = a.inStatus[t];
            continue;
        }
        const int32_t *__restrict__ res = a.residuals + t * a.resStride;
        const int32_t *__restrict__ inter = res + nInit;
        int32_t *__restrict__ v = a.values + t * (size_t)nCells;
        const uint32_t *cf = a.coefs + t * 16;
        const uint32_t seed = cf[0];
        float u[12];
#pragma unroll
        for (int i = 0; i < 12; i++) u[i] = __uint_as_float(cf[1 + i]);
        // the two rows above the band (offsets into rows[]).  Lanes 62 / 63 write the rows above the NEXT band into the same
        // buffers: they are 186 / 189 columns behind lane 0, which reads (ahead of its own column) what they overwrite later
        const uint32_t prev0 = 0, prev1 = nC;

        // rows 0 and 1 as prefix sums (LsDecoder12.unpackInitializers :186-221): to global memory and to prev0 / prev1
        {
            uint32_t carry = seed;
            if (lane == 0) { v[0] = (int32_t)seed; rows[prev0] = seed; }
            for (uint32_t c0 = 1; c0 < nC; c0 += 64) {
                const uint32_t c = c0 + lane;
                const uint32_t x = c < nC ? (uint32_t)res[c - 1] : 0u;
                const uint32_t incl = gf_wave_incl_scan(x) + carry;
                if (c < nC) { v[c] = (int32_t)incl; rows[prev0 + c] = incl; }
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        const uint32_t v10 = seed + (uint32_t)res[nC - 1u];                    // v[1][0]
        if (lane == 0) { v[nC] = (int32_t)v10; rows[prev1] = v10; }
        uint32_t v11 = 0;                                                      // v[1][1]
        {
            uint32_t carry = v10 - seed;
            const uint32_t base = nC - 1u + nR - 1u;
            for (uint32_t c0 = 1; c0 < nC; c0 += 64) {
                const uint32_t c = c0 + lane;
                const uint32_t x = c < nC ? (uint32_t)res[base + c - 1] : 0u;
                const uint32_t incl = gf_wave_incl_scan(x) + carry;
                const uint32_t val = incl + (c < nC ? rows[prev0 + c] : 0u);
                if (c < nC) { v[nC + c] = (int32_t)val; rows[prev1 + c] = val; }
                if (c0 == 1) v11 = (uint32_t)__builtin_amdgcn_readlane((int)val, 0);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        // columns 0 and 1 of the rows below are running sums too; they are carried from band to band
        uint32_t carry0 = v10, carry1 = v11 - v10;
        const uint32_t base0 = nC - 1u, base1 = 2u * (nC - 1u) + nR - 1u, tailBase = base1 + nR - 2u;

        for (uint32_t r0 = 2; r0 < nR; r0 += 64) {
            const uint32_t r = r0 + lane;
            const bool rowValid = r < nR;
            const uint32_t nBand = min(64u, nR - r0);
            uint32_t colv0, colv1, t0 = 0, t1 = 0;
            {
                const uint32_t x0 = rowValid ? (uint32_t)res[base0 + r - 1u] : 0u;
                const uint32_t i0 = gf_wave_incl_scan(x0) + carry0;
                carry0 = (uint32_t)__builtin_amdgcn_readlane((int)i0, 63);
                const uint32_t x1 = rowValid ? (uint32_t)res[base1 + r - 2u] : 0u;
                const uint32_t i1 = gf_wave_incl_scan(x1) + carry1;
                carry1 = (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
                colv0 = i0;
                colv1 = i1 + i0;
                if (rowValid) {
                    t0 = (uint32_t)res[tailBase + 2u * (r - 2u)];
                    t1 = (uint32_t)res[tailBase + 2u * (r - 2u) + 1u];
                }
            }
            const bool feeds = lane >= 62;                                     // rows r0+62, r0+63 are the next band's rows above
            const uint32_t feedBase = lane == 62 ? prev0 : prev1;
            const uint32_t sEnd = 3u * (nBand - 1u) + nC - 1u;

            // cooperative piece k of a group of rows: row = rowBase + RPI k + half, the ROUND steps of `round`
            auto pieceLoad = [&](uint32_t round, uint32_t row) -> uint32_t {   // the residual that (row, step) will consume
                const int32_t e = (int32_t)(round * ROUND + j32) - 3 * (int32_t)row - 2;
                uint32_t x = 0;
                if (row < nBand && e >= 0 && e < (int32_t)wI) x = (uint32_t)inter[(size_t)(r0 + row - 2u) * wI + (uint32_t)e];
                return x;
            };
            auto pieceSlot = [&](uint32_t round, uint32_t row) -> uint32_t { return row * RS + ((round & 1u) * ROUND + j32); };
            auto pieceStore = [&](uint32_t round, uint32_t row) {              // the value that (row, step) produced
                const int32_t c = (int32_t)(round * ROUND + j32) - 3 * (int32_t)row;
                if (row < nBand && c >= 0 && c < (int32_t)nC) v[(size_t)(r0 + row) * nC + (uint32_t)c] = (int32_t)stage[pieceSlot(round, row)];
            };

            // round 0 is loaded up front
#pragma unroll 4
            for (uint32_t k = 0; k < ROUND; k++) {
                const uint32_t row = RPI * k + half;
                stage[pieceSlot(0, row)] = pieceLoad(0, row);
            }

            // lane 0 starts at column 0 with no steps before it: columns 0 and 1 of the rows above are put into its windows
            // here (a lane further down starts at column -3 l and has picked them up by the time it reaches column 0)
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = rows[prev1], a4 = rows[prev1 + 1u];
            uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = rows[prev0], b4 = rows[prev0 + 1u], z1 = 0, z6 = 0;
            uint32_t ld[8];
            bool pending = false;
            uint32_t pendRound = 0, pendRow = 0;
            const uint32_t nRounds = sEnd / ROUND + 1u;
            for (uint32_t round = 0; round < nRounds; round++) {
                for (uint32_t q = 0; q < SUBS; q++) {
                    const uint32_t rowBase = SUBROWS * q;
                    if (pending) {                                             // requested 8 steps ago
#pragma unroll
                        for (uint32_t k = 0; k < 8; k++) stage[pieceSlot(pendRound, pendRow + RPI * k + half)] = ld[k];
                        pending = false;
                    }
                    if (round >= 1u) {
#pragma unroll
                        for (uint32_t k = 0; k < 8; k++) pieceStore(round - 1u, rowBase + RPI * k + half);
                    }
                    if (round + 1u < nRounds) {
#pragma unroll
                        for (uint32_t k = 0; k < 8; k++) ld[k] = pieceLoad(round + 1u, rowBase + RPI * k + half);
                        pending = true;
                        pendRound = round + 1u;
                        pendRow = rowBase;
                    }
                    const uint32_t sFirst = round * ROUND + q * 8u;
                    if (sFirst > sEnd) continue;
                    // everything the 8 steps read from LDS, up front: column c + 2 of the two rows above lane 0 and the
                    // lane's own 8 residuals (each slot is read here before the step that overwrites it)
                    uint32_t ra[8], rb[8], rsv[8];
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t cc = min(sFirst + k + 2u, nC - 1u);
                        ra[k] = rows[prev1 + cc];
                        rb[k] = rows[prev0 + cc];
                        rsv[k] = stage[(uint32_t)lane * RS + ((sFirst + k) & (RING - 1u))];
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) asm volatile("" : "+v"(ra[k]), "+v"(rb[k]), "+v"(rsv[k]));   // read here, not at the use
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t s = sFirst + k;
                        if (s > sEnd) break;
                        const int32_t c = (int32_t)s - 3 * lane;
                        // what enters the two windows: column c + 2 of the row above and of the row two above
                        const uint32_t na = lsop_from_lane_above(ra[k], z1);
                        const uint32_t nb = lsop_from_lane_above(rb[k], a2);
                        a0 = a1; a1 = a2; a2 = a3; a3 = a4; a4 = na;
                        b0 = b1; b1 = b2; b2 = b3; b3 = b4; b4 = nb;
                        float p = u[0] * (float)(int32_t)z1;
                        p = p + u[1] * (float)(int32_t)a1;
                        p = p + u[2] * (float)(int32_t)a2;
                        p = p + u[3] * (float)(int32_t)a3;
                        p = p + u[4] * (float)(int32_t)a4;
                        p = p + u[5] * (float)(int32_t)z6;
                        p = p + u[6] * (float)(int32_t)a0;
                        p = p + u[7] * (float)(int32_t)b0;
                        p = p + u[8] * (float)(int32_t)b1;
                        p = p + u[9] * (float)(int32_t)b2;
                        p = p + u[10] * (float)(int32_t)b3;
                        p = p + u[11] * (float)(int32_t)b4;
                        const uint32_t interior = (uint32_t)lsop_round_sat(p) + rsv[k];               // LsDecoder12 :311-351
                        const uint32_t tail = (c == (int32_t)nC - 2 ? t0 : t1) + (z1 + a2 - a1);      // :353-383
                        const uint32_t border = c == 0 ? colv0 : colv1;
                        const uint32_t val = c < 2 ? border : (c <= (int32_t)nC - 3 ? interior : tail);
                        const bool act = rowValid && c >= 0 && c < (int32_t)nC;
                        if (act) {
                            stage[(uint32_t)lane * RS + (s & (RING - 1u))] = val;
                            if (feeds) rows[feedBase + (uint32_t)c] = val;
                        }
                        z6 = act ? z1 : z6;
                        z1 = act ? val : z1;
                    }
                }
            }
            if (pending) {                                                     // (nothing is pending after the last round)
                pending = false;
            }
            // the last round's values
#pragma unroll 4
            for (uint32_t k = 0; k < ROUND; k++) pieceStore(nRounds - 1u, RPI * k + half);
        }
        if (lane == 0) a.status[t] = GF_K_OK;
    }
}


// The same wavefront as ONE pipeline down the tile (round 3) instead of a fill and a drain per band of 64 rows: lane l takes rows
// 2 + l, 2 + 64 + l, ... one after the other, and starts the next one P = max(nC, 208) steps (rounded up to a round) after the
// last: lane 63 starts 189 steps into a period, lane 0 reads eight columns ahead in the row buffers and two more are kept between
// them (189 + 2 + 8 = 199 <= P, i.e. 208 as a multiple of the 16-step round).  A band costs P steps instead of 189 + nC: 517
// instead of 648 for the 120 x 150 tiles of the ETOPO1-shaped batch, 1,213 instead of 1,780 for 256 x 256.  Everything a step does is the band
// kernel's; what changes is bookkeeping: a lane's column wraps at P (its row index goes up by 64), the per-row constants of the
// next rows are worked out by the whole wave when lane 0 starts a row and taken over by a lane when it gets there, lane 0 alone
// re-arms its windows from the two row buffers, and the staging pieces tell rows apart by whether a step lies before or behind a
// row's start inside the current period.  For tiles of more than one band and at least 16 columns (the launcher decides).
// (the body: tiles tFirst, tFirst + tStride, ... -- k_lsop_reconstruct_plane's first workgroups run it beside the plane tiles, round 6)
template <int ROUND>
__device__ __forceinline__ void lsop_reconstruct_pipe_tiles(const GfLsopReconArgs &a, uint32_t *const reconLds, const size_t tFirst,
                                                            const size_t tStride)
{
    constexpr uint32_t RING = 2u * ROUND;
    constexpr uint32_t RS = RING + 1u;
    constexpr uint32_t RPI = 64u / ROUND;
    constexpr uint32_t SUBS = ROUND / 8u;
    constexpr uint32_t SUBROWS = 64u / SUBS;
    const int lane = threadIdx.x;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const uint32_t nInit = lsop_n_init(nR, nC);
    const uint32_t wI = nC - 4u;
    uint32_t *stage = reconLds;
    uint32_t *rows = reconLds + 64u * RS;
    const uint32_t half = (uint32_t)lane / ROUND, j32 = (uint32_t)lane % ROUND;
    const uint32_t nPh = (nR - 2u + 63u) / 64u;                              // rows per lane
    // steps between a lane's rows: lane 63 starts its row 189 steps into a period, and lane 0 reads column c + 2 of that row for
    // the eight steps of a group at once, from the row buffers: 189 + 2 + 8 <= P
    static_assert(189 + 2 + 8 <= 208, "the period's floor: lane 63's start, the row buffers' safety distance, lane 0's read-ahead");
    const uint32_t P = ((max(nC, 208u) + ROUND - 1u) / ROUND) * ROUND;
    const uint32_t nLast = nR - 2u - 64u * (nPh - 1u);
    const uint32_t sEnd = (nPh - 1u) * P + 3u * (nLast - 1u) + nC - 1u;       // the last step that produces a value
    const uint32_t nRounds = sEnd / ROUND + 1u;

    for (size_t t = tFirst; t < a.nTiles; t += tStride) {
        if (a.planes && a.coefs[t * 16 + GF_LSOP_FMT_WORD] == 1u) continue;
        if (a.inStatus && a.inStatus[t] != GF_K_OK) {
            if (lane == 0) a.status[t] = a.inStatus[t];
            continue;
        }
        const int32_t *__restrict__ res = a.residuals + t * a.resStride;
        const int32_t *__restrict__ inter = res + nInit;
        int32_t *__restrict__ v = a.values + t * (size_t)nCells;
        const uint32_t *cf = a.coefs + t * 16;
        const uint32_t seed = cf[0];
        float u[12];
#pragma unroll
        for (int i = 0; i < 12; i++) u[i] = __uint_as_float(cf[1 + i]);
        const uint32_t prev0 = 0, prev1 = nC;

        // rows 0 and 1 as in the band kernel
        {
            uint32_t carry = seed;
            if (lane == 0) { v[0] = (int32_t)seed; rows[prev0] = seed; }
            for (uint32_t c0 = 1; c0 < nC; c0 += 64) {
                const uint32_t c = c0 + lane;
                const uint32_t x = c < nC ? (uint32_t)res[c - 1] : 0u;
                const uint32_t incl = gf_wave_incl_scan(x) + carry;
                if (c < nC) { v[c] = (int32_t)incl; rows[prev0 + c] = incl; }
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        const uint32_t v10 = seed + (uint32_t)res[nC - 1u];
        if (lane == 0) { v[nC] = (int32_t)v10; rows[prev1] = v10; }
        uint32_t v11 = 0;
        {
            uint32_t carry = v10 - seed;
            const uint32_t base = nC - 1u + nR - 1u;
            for (uint32_t c0 = 1; c0 < nC; c0 += 64) {
                const uint32_t c = c0 + lane;
                const uint32_t x = c < nC ? (uint32_t)res[base + c - 1] : 0u;
                const uint32_t incl = gf_wave_incl_scan(x) + carry;
                const uint32_t val = incl + (c < nC ? rows[prev0 + c] : 0u);
                if (c < nC) { v[nC + c] = (int32_t)val; rows[prev1 + c] = val; }
                if (c0 == 1) v11 = (uint32_t)__builtin_amdgcn_readlane((int)val, 0);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        uint32_t carry0 = v10, carry1 = v11 - v10;
        const uint32_t base0 = nC - 1u, base1 = 2u * (nC - 1u) + nR - 1u, tailBase = base1 + nR - 2u;

        // (row, step) of a staging piece -> the row of the tile and its column: step S belongs to period pg = S / P (the same for a
        // whole round: P is a multiple of ROUND); row `row` starts 3 row steps into a period, what lies before that is the end of
        // its row of the period before
        uint32_t relOf[3] = {0u, 0u, ROUND % P}, pgOf[3] = {0u, 0u, ROUND / P};      // first step mod P / period of rounds round - 1, round, round + 1
        auto pieceRC = [&](uint32_t round, uint32_t rel, uint32_t pg, uint32_t row, uint32_t &r, int32_t &c) {
            (void)round;
            const int32_t tq = (int32_t)(rel + j32) - 3 * (int32_t)row;
            const bool before = tq < 0;
            c = before ? tq + (int32_t)P : tq;
            const int32_t ph = (int32_t)pg - (before ? 1 : 0);
            r = ph >= 0 ? 2u + 64u * (uint32_t)ph + row : 0xFFFFFFFFu;
        };
        auto pieceLoad = [&](uint32_t round, uint32_t rel, uint32_t pg, uint32_t row) -> uint32_t {
            uint32_t r;
            int32_t c;
            pieceRC(round, rel, pg, row, r, c);
            const int32_t e = c - 2;
            uint32_t x = 0;
            if (r < nR && e >= 0 && e < (int32_t)wI) x = (uint32_t)inter[(size_t)(r - 2u) * wI + (uint32_t)e];
            return x;
        };
        auto pieceSlot = [&](uint32_t round, uint32_t row) -> uint32_t { return row * RS + ((round & 1u) * ROUND + j32); };
        auto pieceStore = [&](uint32_t round, uint32_t rel, uint32_t pg, uint32_t row) {
            uint32_t r;
            int32_t c;
            pieceRC(round, rel, pg, row, r, c);
            if (r < nR && c < (int32_t)nC) v[(size_t)r * nC + (uint32_t)c] = (int32_t)stage[pieceSlot(round, row)];
        };

#pragma unroll 4
        for (uint32_t k = 0; k < ROUND; k++) {
            const uint32_t row = RPI * k + half;
            stage[pieceSlot(0, row)] = pieceLoad(0, 0u, 0u, row);
        }

        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = rows[prev1], a4 = rows[prev1 + 1u];
        uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = rows[prev0], b4 = rows[prev0 + 1u], z1 = 0, z6 = 0;
        uint32_t colv0 = 0, colv1 = 0, t0 = 0, t1 = 0;            // of the lane's current row
        uint32_t pc0 = 0, pc1 = 0, pt0 = 0, pt1 = 0;              // of its next one (taken over when the lane gets there)
        int32_t cBase = -3 * lane;                                // the lane's column at the next step (wraps at P)
        uint32_t ph = 0;                                          // its row: 2 + 64 ph + lane
        uint32_t nextStart = 0, pn = 0;                           // lane 0 starts row 2 + 64 pn at step nextStart
        const bool feeds = lane >= 62;
        const uint32_t feedBase = lane == 62 ? prev0 : prev1;
        uint32_t ld[8];
        bool pending = false;
        uint32_t pendRound = 0, pendRow = 0;
        for (uint32_t round = 0; round < nRounds; round++) {
            for (uint32_t q = 0; q < SUBS; q++) {
                const uint32_t rowBase = SUBROWS * q;
                const uint32_t sFirst = round * ROUND + q * 8u;
                // a lane whose column 0 lies in these eight steps takes over the border values of its next row (nothing of the
                // row it is finishing needs them); its tail residuals one group later (the row's own tail is behind it then)
                {
                    const bool wrapSoon = (cBase <= 0 && cBase + 8 > 0) || cBase + 8 > (int32_t)P;
                    colv0 = wrapSoon ? pc0 : colv0;
                    colv1 = wrapSoon ? pc1 : colv1;
                    const bool justWrapped = cBase >= 0 && cBase < 8;
                    t0 = justWrapped ? pt0 : t0;
                    t1 = justWrapped ? pt1 : t1;
                }
                if (sFirst == nextStart && pn < nPh) {                        // lane 0 starts a row: the next rows' constants
                    const uint32_t r = 2u + 64u * pn + (uint32_t)lane;
                    const bool rowValidN = r < nR;
                    const uint32_t x0 = rowValidN ? (uint32_t)res[base0 + r - 1u] : 0u;
                    const uint32_t i0 = gf_wave_incl_scan(x0) + carry0;
                    carry0 = (uint32_t)__builtin_amdgcn_readlane((int)i0, 63);
                    const uint32_t x1 = rowValidN ? (uint32_t)res[base1 + r - 2u] : 0u;
                    const uint32_t i1 = gf_wave_incl_scan(x1) + carry1;
                    carry1 = (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
                    pc0 = i0;
                    pc1 = i1 + i0;
                    pt0 = rowValidN ? (uint32_t)res[tailBase + 2u * (r - 2u)] : 0u;
                    pt1 = rowValidN ? (uint32_t)res[tailBase + 2u * (r - 2u) + 1u] : 0u;
                    if (pn == 0u) { colv0 = pc0; colv1 = pc1; t0 = pt0; t1 = pt1; }
                    else {
                        // the lanes that reach their column 0 within this group (0, 1, 2) start rows of THIS period: the take-over
                        // above handed them the constants of the period before
                        const bool wrapSoon = (cBase <= 0 && cBase + 8 > 0) || cBase + 8 > (int32_t)P;
                        colv0 = wrapSoon ? pc0 : colv0;
                        colv1 = wrapSoon ? pc1 : colv1;
                        if (lane == 0) {
                            // its windows: columns 0 and 1 of the two rows above.  (Only these: lane 1 is three columns behind
                            // on its old row and still takes lane 0's last value and the middle of its window at this step.)
                            a3 = rows[prev1]; a4 = rows[prev1 + 1u];
                            b3 = rows[prev0]; b4 = rows[prev0 + 1u];
                            t0 = pt0; t1 = pt1;                               // (its old row ended nC steps into the period before)
                        }
                    }
                    pn++;
                    nextStart += P;
                }
                if (pending) {
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) stage[pieceSlot(pendRound, pendRow + RPI * k + half)] = ld[k];
                    pending = false;
                }
                if (round >= 1u) {
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) pieceStore(round - 1u, relOf[0], pgOf[0], rowBase + RPI * k + half);
                }
                if (round + 1u < nRounds) {
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) ld[k] = pieceLoad(round + 1u, relOf[2], pgOf[2], rowBase + RPI * k + half);
                    pending = true;
                    pendRound = round + 1u;
                    pendRow = rowBase;
                }
                if (sFirst <= sEnd) {
                    // lane 0's column in this group (its windows take column c + 2 of the two rows above from the row buffers)
                    const uint32_t c00 = sFirst - (nextStart - P);
                    uint32_t ra[8], rb[8], rsv[8];
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t cc = min(c00 + k + 2u, nC - 1u);
                        ra[k] = rows[prev1 + cc];
                        rb[k] = rows[prev0 + cc];
                        rsv[k] = stage[(uint32_t)lane * RS + ((sFirst + k) & (RING - 1u))];
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) asm volatile("" : "+v"(ra[k]), "+v"(rb[k]), "+v"(rsv[k]));
                    const bool validCur = 2u + 64u * ph + (uint32_t)lane < nR, validNext = 2u + 64u * (ph + 1u) + (uint32_t)lane < nR;
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t s = sFirst + k;
                        if (s > sEnd) break;
                        const int32_t cw = cBase + (int32_t)k;
                        const bool wrapped = cw >= (int32_t)P;
                        const int32_t c = wrapped ? cw - (int32_t)P : cw;
                        const bool rowValid = wrapped ? validNext : validCur;
                        const uint32_t na = lsop_from_lane_above(ra[k], z1);
                        const uint32_t nb = lsop_from_lane_above(rb[k], a2);
                        a0 = a1; a1 = a2; a2 = a3; a3 = a4; a4 = na;
                        b0 = b1; b1 = b2; b2 = b3; b3 = b4; b4 = nb;
                        float p = u[0] * (float)(int32_t)z1;
                        p = p + u[1] * (float)(int32_t)a1;
                        p = p + u[2] * (float)(int32_t)a2;
                        p = p + u[3] * (float)(int32_t)a3;
                        p = p + u[4] * (float)(int32_t)a4;
                        p = p + u[5] * (float)(int32_t)z6;
                        p = p + u[6] * (float)(int32_t)a0;
                        p = p + u[7] * (float)(int32_t)b0;
                        p = p + u[8] * (float)(int32_t)b1;
                        p = p + u[9] * (float)(int32_t)b2;
                        p = p + u[10] * (float)(int32_t)b3;
                        p = p + u[11] * (float)(int32_t)b4;
                        const uint32_t interior = (uint32_t)lsop_round_sat(p) + rsv[k];
                        const uint32_t tail = (c == (int32_t)nC - 2 ? t0 : t1) + (z1 + a2 - a1);
                        const uint32_t border = c == 0 ? colv0 : colv1;
                        const uint32_t val = c < 2 ? border : (c <= (int32_t)nC - 3 ? interior : tail);
                        const bool act = rowValid && c >= 0 && c < (int32_t)nC;
                        if (act) {
                            stage[(uint32_t)lane * RS + (s & (RING - 1u))] = val;
                            if (feeds) rows[feedBase + (uint32_t)c] = val;
                        }
                        z6 = act ? z1 : z6;
                        z1 = act ? val : z1;
                    }
                }
                cBase += 8;
                if (cBase >= (int32_t)P) { cBase -= (int32_t)P; ph++; }
            }
            relOf[0] = relOf[1]; pgOf[0] = pgOf[1];
            relOf[1] = relOf[2]; pgOf[1] = pgOf[2];
            relOf[2] += ROUND;
            if (relOf[2] >= P) { relOf[2] -= P; pgOf[2]++; }
        }
        // (after the loop relOf[1] / pgOf[1] belong to round nRounds, relOf[0] / pgOf[0] to the last round)
#pragma unroll 4
        for (uint32_t k = 0; k < ROUND; k++) pieceStore(nRounds - 1u, relOf[0], pgOf[0], RPI * k + half);
        if (lane == 0) a.status[t] = GF_K_OK;
    }
}


template <int ROUND>
__global__ __launch_bounds__(64, RECON_WAVES_PER_SIMD) void k_lsop_reconstruct_pipe(GfLsopReconArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t reconLds[];
    lsop_reconstruct_pipe_tiles<ROUND>(a, reconLds, blockIdx.x, gridDim.x);
}


// ------------------------------------------------------------------------------------------------
// k_lsop_reconstruct_plane (round 6): the pipeline of k_lsop_reconstruct_pipe for tiles whose residuals k_lsop_unpack2 left as a byte
// plane in pipeline order (gvrs_kernels.h: GfLsopPlaneGeom).  What the old kernel spent per step -- 185 instructions, of which the
// arithmetic of LsDecoder12.java:311-383 is about fifty -- went into moving data between the order the memory wants (rows) and the
// order the pipeline wants (a lane a row, three steps apart): sixteen residual pieces in and sixteen value pieces out per round,
// each with its own (row, column) arithmetic, through an LDS ring.  Here
//   * TWO TILES SHARE A WAVE, thirty-two lanes each: a pipeline of 64 lanes is 189 steps deep, so a lane cannot start a new row
//     more often than every 208 steps -- on a 150-column tile its lanes idle for a quarter of every period and half of the first
//     and the last one (517 steps per tile); with 32 lanes a period is max(nC, 112) steps: 693 steps per PAIR of such tiles;
//   * a lane's sixteen residuals of a round are one 16-byte load of the plane, taken apart with v_bfe_i32 / SDWA;
//   * EVERY cell of rows 2.. is "something the lane already has + a byte of the plane": an interior cell the rounded prediction +
//     its residual (:311-351); the two tail cells and column 1 the triangle z1 + a2 - a1 + their residual (:353-383, :204-221);
//     column 0 the cell above + its residual (:196-202) -- k_lsop_unpack2 puts those four initialisers of a row into the plane's
//     holes at the row's ends, so there are no per-row constants to work out, keep or hand over;
//   * the values wait in LDS (a word per lane and step, conflict-free) and leave every SECOND round as a lane's 128 bytes in a row:
//     eight 16-byte stores to its own row (a quad that a row's end or start cuts goes out cell by cell).  Written four cells at a
//     time as they came the same kernel took 0.87 ms instead of 0.41 without any store: 262,000 rows are open at once on the chip,
//     a 128-byte line of each of them a quarter written for thousands of cycles -- the L2 (32 MB in all) wrote 2.7 GB back for 0.93;
//   * a neighbour is converted to float and multiplied by the coefficients it will meet ONCE, when it enters the window (packed
//     multiplies with the value on both halves), not in each of the five steps it takes part in; the rounding is the encoder's
//     lsop_round_f32 (single precision, exact);
//   * a lane's state is not guarded: what a lane computes while it is between two rows (or before its first) is never stored and
//     reaches only cells of the lane below that are not stored either; which of a round's sixteen steps is a row's column 0 and
//     which its column nC - 2 is worked out once per round, not the column per step.
// ------------------------------------------------------------------------------------------------
#ifndef RECON_PLANE_WAVES
#define RECON_PLANE_WAVES 4
#endif
typedef float lsop_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint32_t lsop_lds_u32;
constexpr uint32_t RP_STAGE_STRIDE = 33;      // words per lane of the value stage: 32 steps + 1 (lanes of a step hit different banks)

__global__ __launch_bounds__(64, RECON_PLANE_WAVES) void k_lsop_reconstruct_plane(GfLsopReconArgs a, GfLsopPlaneGeom g, uint32_t nOld)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t rpLds[];
    // The first nOld workgroups take the tiles that are NOT planes (a residual beyond a byte: a handful of a terrain batch, if any)
    // through the old pipeline, beside the plane tiles: in a launch of their own behind this one, three such tiles of 256 x 256 were
    // 0.46 ms -- one wave's walk down a tile is a chain of 1,200 steps whatever else the chip does
    if (blockIdx.x < nOld) {
        lsop_reconstruct_pipe_tiles<RECON_ROUND>(a, rpLds, blockIdx.x, nOld);
        return;
    }
    constexpr uint32_t L = GF_LSOP_PLANE_LANES;
    const uint32_t lane = threadIdx.x, lam = lane & (L - 1u), half = lane / L;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const int32_t P = (int32_t)g.P;
    // per tile of the pair: [FRONT + P + 32][2] words {row r - 2, row r - 1} of lane 0's row by column; then the wave's value stage
    uint32_t *const rowsAB = rpLds + half * g.rowsWords + 2u * GF_LSOP_PLANE_FRONT;
    lsop_lds_u32 *const stageAll = (lsop_lds_u32 *)(rpLds + 2u * g.rowsWords);
    lsop_lds_u32 *const stage = stageAll + lane * RP_STAGE_STRIDE;
    uint32_t *const desc = rpLds + 2u * g.rowsWords + 64u * RP_STAGE_STRIDE;             // [64][2]: see the even rounds
    const bool head = lam == 0u, feeds = lam >= L - 2u;
    const uint32_t nRounds = (g.nBlocks + 1u) & ~1u;                        // (an even number: the values leave every second round)

    for (size_t pair = blockIdx.x - nOld; 2 * pair < a.nTiles; pair += gridDim.x - nOld) {
        // ---- the two tiles' rows 0 and 1, a tile after the other by the whole wave ----
        bool okT[2];
#pragma unroll
        for (uint32_t hh = 0; hh < 2; hh++) {
            const size_t t = 2 * pair + hh;
            okT[hh] = t < a.nTiles && a.coefs[t * 16 + GF_LSOP_FMT_WORD] == 1u;     // (else: int32 residuals, the kernels above)
            if (okT[hh] && a.inStatus && a.inStatus[t] != GF_K_OK) {
                if (lane == 0) a.status[t] = a.inStatus[t];
                okT[hh] = false;
            }
            if (!okT[hh]) continue;
            const int32_t *__restrict__ res = a.residuals + t * a.resStride;
            int32_t *__restrict__ v = a.values + t * (size_t)nCells;
            uint32_t *const rows = rpLds + hh * g.rowsWords + 2u * GF_LSOP_PLANE_FRONT;
            const uint32_t seed = a.coefs[t * 16];
            // rows 0 and 1 as prefix sums (LsDecoder12.unpackInitializers :186-221): to global memory and to lane 0's row buffers
            {
                uint32_t carry = seed;
                if (lane == 0) { v[0] = (int32_t)seed; rows[0] = seed; }
                for (uint32_t c0 = 1; c0 < nC; c0 += 64) {
                    const uint32_t c = c0 + lane;
                    const uint32_t x = c < nC ? (uint32_t)res[c - 1] : 0u;
                    const uint32_t incl = gf_wave_incl_scan(x) + carry;
                    if (c < nC) { v[c] = (int32_t)incl; rows[2u * c] = incl; }
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
            }
            const uint32_t v10 = seed + (uint32_t)res[nC - 1u];                    // v[1][0]
            if (lane == 0) { v[nC] = (int32_t)v10; rows[1] = v10; }
            {
                uint32_t carry = v10 - seed;
                const uint32_t base = nC - 1u + nR - 1u;
                for (uint32_t c0 = 1; c0 < nC; c0 += 64) {
                    const uint32_t c = c0 + lane;
                    const uint32_t x = c < nC ? (uint32_t)res[base + c - 1] : 0u;
                    const uint32_t incl = gf_wave_incl_scan(x) + carry;
                    const uint32_t val = incl + (c < nC ? rows[2u * c] : 0u);
                    if (c < nC) { v[nC + c] = (int32_t)val; rows[2u * c + 1u] = val; }
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
            }
        }
        if (!okT[0] && !okT[1]) continue;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        // ---- per lane: its tile (a half without a tile computes on the other one's and stores nothing) ----
        const bool tileOk = half == 0u ? okT[0] : okT[1];
        const size_t tMine = 2 * pair + (tileOk ? half : (okT[0] ? 0u : 1u));
        const uint8_t *__restrict__ plane = reinterpret_cast<const uint8_t *>(a.residuals + tMine * a.resStride + g.offWords);
        const uint32_t tileHalf = (uint32_t)(tMine - 2 * pair);
        int32_t *const vPair = a.values + 2 * pair * (size_t)nCells;
        lsop_f2 u43, u21, u6_, uBA, u98, u7_, u05;
        {
            const uint32_t *cf = a.coefs + tMine * 16;
            float u[12];
#pragma unroll
            for (int i = 0; i < 12; i++) u[i] = __uint_as_float(cf[1 + i]);
            u43 = lsop_f2{u[4], u[3]}; u21 = lsop_f2{u[2], u[1]}; u6_ = lsop_f2{u[6], 0.0f};
            uBA = lsop_f2{u[11], u[10]}; u98 = lsop_f2{u[9], u[8]}; u7_ = lsop_f2{u[7], 0.0f};
            u05 = lsop_f2{u[0], u[5]};
        }
        GfU4 chunk = *reinterpret_cast<const GfU4 *>(plane + lam * 16u);

        // products of the neighbours with the coefficients they will meet, by age (0: entered in this step).  Row above: a value is
        // a4 (u4) when it enters, then a3 (u3), a2 (u2), a1 (u1), a0 (u6); two rows above: b4 (u11) ... b0 (u7)
        lsop_f2 A43n = {0, 0}, A43o = {0, 0}, A21[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, B43n = {0, 0}, B43o = {0, 0}, B21[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        float A6[5] = {0, 0, 0, 0, 0}, B7[5] = {0, 0, 0, 0, 0};
        lsop_f2 Z = {0, 0};                                // {u0 z1, u5 z1}: the second one is next step's u5 z6
        float z6p = 0;
        uint32_t a1i = 0, a2i = 0, a3i = 0, a4i = 0, z1i = 0;
        // a neighbour of each of the two rows above enters (wave_shr:1; lanes 0 and 32 take theirs from the row buffers)
        auto enter = [&](uint32_t raK, uint32_t rbK) {
            uint32_t na = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)z1i, 0x138, 0xf, 0xf, true);      // wave_shr:1
            uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a2i, 0x138, 0xf, 0xf, true);
            na = head ? raK : na;
            nb = head ? rbK : nb;
            a1i = a2i; a2i = a3i; a3i = a4i; a4i = na;
            const float xa = (float)(int32_t)na, xb = (float)(int32_t)nb;
            A43o = A43n; A43n = u43 * lsop_f2{xa, xa};
            A21[3] = A21[2]; A21[2] = A21[1]; A21[1] = A21[0]; A21[0] = u21 * lsop_f2{xa, xa};
            A6[4] = A6[3]; A6[3] = A6[2]; A6[2] = A6[1]; A6[1] = A6[0]; A6[0] = u6_.x * xa;
            B43o = B43n; B43n = uBA * lsop_f2{xb, xb};
            B21[3] = B21[2]; B21[2] = B21[1]; B21[1] = B21[0]; B21[0] = u98 * lsop_f2{xb, xb};
            B7[4] = B7[3]; B7[3] = B7[2]; B7[2] = B7[1]; B7[1] = B7[0]; B7[0] = u7_.x * xb;
        };
        // lane 0's first row: columns 0 and 1 of rows 1 and 0 enter as they do in the two steps before a later row's start
        {
            const GfU4 w = *reinterpret_cast<const GfU4 *>(rowsAB);
            enter(w.y, w.x);
            enter(w.w, w.z);
        }
        int32_t cB = -3 * (int32_t)lam;                    // the lane's column at the round's first step (negative: not started)
        uint32_t ph = 0;                                   // its row: 2 + 32 ph + lam
        asm volatile("" : "+v"(chunk.x), "+v"(chunk.y), "+v"(chunk.z), "+v"(chunk.w));      // (waited for here, not inside the loop: see k == 2)
        for (uint32_t R = 0; R < nRounds; R++) {
            // the next round's residuals (the last rounds ask for the last block again)
            GfU4 nxt = *reinterpret_cast<const GfU4 *>(plane + ((size_t)min(R + 1u, g.nBlocks - 1u) * L + lam) * 16u);
            // lane 0's column (a multiple of 16: it starts its rows with a round); its rows above at columns c0 + 2 .. c0 + 17 -- in the
            // round before it starts a row the last two are that row's columns 0 and 1
            const int32_t c0 = __builtin_amdgcn_readfirstlane(cB);
            const uint32_t cHi = c0 + 16 == P ? 0u : (uint32_t)(c0 + 16);
            const bool wrapSoon = cB > P - 16;
            // the steps of this round at which the lane is in column 0 / in column nC - 2 (anything else: none of the sixteen)
            const int32_t kB0 = cB <= 1 ? -cB : P - cB, kT0 = (int32_t)nC - 2 - cB;
            // where lanes 30 / 31 leave their values for lane 0's next row: they begin a row at step 10 / 13 of a round (3 l mod 16)
            lsop_lds_u32 *feedLo = (lsop_lds_u32 *)(rowsAB + 2 * cB + (int32_t)(lam - (L - 2u)));
            lsop_lds_u32 *feedHi = wrapSoon ? feedLo - 2 * P : feedLo;
            lsop_lds_u32 *feedMid = lam == L - 2u ? feedHi : feedLo;
            const bool odd = (R & 1u) != 0u;
            lsop_lds_u32 *st = odd ? stage + 16 : stage;
            asm volatile("" : "+v"(feedLo), "+v"(feedHi), "+v"(feedMid), "+v"(st));    // (a register each per round, the step in the offset field)
            if (!odd) {
                // where the lane's 32 values of this round and the next belong: its column now, its row (cells from the pair's first
                // tile), whether that row and its next one exist -- for the lanes that will store them
                const uint32_t r = 2u + L * ph + lam;
                GfU2 d;
                d.x = (uint32_t)cB;
                d.y = (tileHalf * nCells + r * nC) | (tileOk && r < nR ? 1u << 30 : 0u) | (tileOk && r + L < nR ? 1u << 31 : 0u);
                *reinterpret_cast<GfU2 *>(desc + 2u * lane) = d;
            }
            uint32_t ra[4], rb[4];
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                if ((k & 3u) == 0u) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; j += 2) {
                        const uint32_t col = k + j == 14u ? cHi : (uint32_t)(c0 + 2 + (int32_t)(k + j));
                        const GfU4 w = *reinterpret_cast<const GfU4 *>(rowsAB + 2 * (int32_t)col);
                        rb[j] = w.x; ra[j] = w.y; rb[j + 1] = w.z; ra[j + 1] = w.w;
                    }
                }
                enter(ra[k & 3u], rb[k & 3u]);
                float p = Z.x;                      // u0 z1
                p = p + A21[3].y;                   // u1 a1
                p = p + A21[2].x;                   // u2 a2
                p = p + A43o.y;                     // u3 a3
                p = p + A43n.x;                     // u4 a4
                p = p + z6p;                        // u5 z6
                p = p + A6[4];                      // u6 a0
                p = p + B7[4];                      // u7 b0
                p = p + B21[3].y;                   // u8 b1
                p = p + B21[2].x;                   // u9 b2
                p = p + B43o.y;                     // u10 b3
                p = p + B43n.x;                     // u11 b4
                const uint32_t word = k < 4 ? chunk.x : k < 8 ? chunk.y : k < 12 ? chunk.z : chunk.w;
                const int32_t r8 = (int32_t)(word << (24u - 8u * (k & 3u))) >> 24;
                const bool isB0 = kB0 == (int32_t)k, isB1 = kB0 == (int32_t)k - 1, isT0 = kT0 == (int32_t)k, isT1 = kT0 == (int32_t)k - 1;
                uint32_t base = (uint32_t)lsop_round_f32(p);                               // LsDecoder12 :311-351
                if (isB1 || isT0 || isT1) base = z1i + a2i - a1i;                          // :353-383, :204-221
                if (isB0) base = a2i;                                                      // :196-202
                const uint32_t val = base + (uint32_t)r8;
                z6p = Z.y;
                Z = u05 * lsop_f2{(float)(int32_t)val, (float)(int32_t)val};
                z1i = val;
                st[k] = val;
                if (feeds) (k < 10 ? feedLo : k < 13 ? feedMid : feedHi)[2 * (int32_t)k] = val;
                // (the next round's residuals have arrived before this round's stores are issued: loads and stores share a counter,
                // and a wait for the load behind the stores would be a wait for the stores)
                if (k == 2u) asm volatile("" : "+v"(nxt.x), "+v"(nxt.y), "+v"(nxt.z), "+v"(nxt.w));
            }
            if (odd) {
                // the wave's 64 x 32 values of the two rounds to their rows: EIGHT LANES TAKE A ROW'S 128 BYTES (a store instruction: eight
                // rows), four cells each -- the memory sees whole lines or two pieces of a line, not sixty-four 16-byte requests per
                // instruction (that form, a lane storing its own row: 0.84 ms where the kernel without stores takes 0.42: the L2's
                // request rate, not its bytes).  Whole quads where the row covers them, single cells at its ends
                const uint32_t q = lane & 7u;
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    const uint32_t rho = 8u * i + (lane >> 3);
                    const GfU2 d = *reinterpret_cast<const GfU2 *>(desc + 2u * rho);
                    const int32_t cq = (int32_t)d.x + 4 * (int32_t)q;
                    const bool vCurQ = (d.y >> 30) & 1u, vNextQ = (d.y >> 31) != 0u;
                    int32_t *const vrowQ = vPair + (d.y & 0x3fffffffu);
#ifdef GF_RP_NO_STORES                                      // (experiment builds: what the kernel costs without its stores)
                    const bool full = false, part = false;
#else
                    const bool full = vCurQ && cq >= 0 && cq + 3 < (int32_t)nC;
                    const bool part = !full && ((vCurQ && cq + 3 >= 0 && cq < (int32_t)nC) || (vNextQ && cq + 3 >= P));
#endif
                    if (full || part) {
                        const lsop_lds_u32 *src = stageAll + rho * RP_STAGE_STRIDE + 4u * q;
                        GfU4 x;
                        x.x = src[0]; x.y = src[1]; x.z = src[2]; x.w = src[3];
                        if (full) *reinterpret_cast<GfU4 *>(vrowQ + cq) = x;
                        else {
                            const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int32_t cj = cq + j;
                                const bool wr = cj >= P;
                                const bool ok = wr ? vNextQ : (vCurQ && cj >= 0 && cj < (int32_t)nC);   // (a new row's first columns: inside it)
                                int32_t *const at = wr ? vrowQ + (size_t)L * nC + (cj - P) : vrowQ + cj;
                                if (ok) *at = (int32_t)xs[j];
                            }
                        }
                    }
                }
            }
            chunk = nxt;
            cB += 16;
            if (cB >= P) { cB -= P; ph++; }
        }
        if (head && tileOk) a.status[2 * pair + half] = GF_K_OK;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

static int gf_lsop_predict16_threads(int nRows, int nCols);

hipError_t gf_launch_lsop_predict(const int32_t *values, int32_t *residuals, size_t resStride, uint32_t *coefs,
                                  int32_t *status, size_t nTiles, int nRows, int nCols, hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    if (gf_lsop_predict16_eligible(nRows, nCols)) {
        // terrain-sized tiles: the tile in LDS as digit planes (k_lsop_predict16<true>), k_lsop_predict behind it for what it leaves
        GfLsopPredict16Args p{values, reinterpret_cast<int16_t *>(residuals), 2 * resStride, coefs, status, nullptr, nTiles, nRows, nCols};
        const size_t dyn16 = 2 * (((size_t)nRows * (size_t)nCols + 64 + 15) & ~(size_t)15);
        static GfDynLdsOptIn opt32;
        if (gf_lsop_predict16_threads(nRows, nCols) == 1024) {
            static GfDynLdsOptIn opt32w;
            const hipError_t e = gf_opt_in_dyn_lds((k_lsop_predict16<true, 1024>), dyn16, opt32w);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_lsop_predict16<true, 1024>), gf_tile_grid(nTiles), dim3(1024), dyn16, stream, p);
        } else {
            const hipError_t e = gf_opt_in_dyn_lds((k_lsop_predict16<true, 256>), dyn16, opt32);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_lsop_predict16<true, 256>), gf_tile_grid(nTiles), dim3(256), dyn16, stream, p);
        }
        GfLsopPredictArgs b{values, residuals, resStride, coefs, status, nTiles, nRows, nCols, 1};
        hipLaunchKernelGGL(k_lsop_predict, gf_tile_grid(nTiles), dim3(256), ((size_t)4 * (size_t)nCols + 40) * 4, stream, b);
        return hipGetLastError();
    }
    GfLsopPredictArgs a{values, residuals, resStride, coefs, status, nTiles, nRows, nCols, 0};
    const size_t dyn = (size_t)nCols <= LSOP_RING_MAXC ? ((size_t)4 * (size_t)nCols + 40) * 4 : 0;   // (+ 40 ints: lsop_gram_mfma reads a
                                                                                                        // row's last group of sixteen past its end)
    hipLaunchKernelGGL(k_lsop_predict, gf_tile_grid(nTiles), dim3(256), dyn, stream, a);
    return hipGetLastError();
}

hipError_t gf_launch_canon_pack2(const int32_t *residuals, size_t resStride, const uint32_t *coefs, const int32_t *inStatus,
                                 uint8_t *out, size_t slotStride, uint32_t *lengths, int32_t *status, size_t nTiles,
                                 uint32_t n0, uint32_t n1, int codecIndex, hipStream_t stream, int valueChecksum, const uint32_t *hist16)
{
    if (nTiles == 0) return hipSuccess;
    GfPack2Args a{residuals, resStride, coefs, inStatus, out, slotStride, lengths, status, nTiles, n0, n1, codecIndex, valueChecksum, hist16};
    // hist16: k_lsop_predict16 ran first -- its tiles through the halfword instantiation, what it left to k_lsop_predict through the other
    if (hist16) hipLaunchKernelGGL(k_canon_pack2<true>, gf_tile_grid(nTiles), dim3(ENC_THREADS), 0, stream, a);
    hipLaunchKernelGGL(k_canon_pack2<false>, gf_tile_grid(nTiles), dim3(ENC_THREADS), 0, stream, a);
    return hipGetLastError();
}

// the tile sizes k_lsop_predict16 takes: the halfword tile and its static LDS in a third of a CU's 160 KB, rows the matrix-pipe
// form accepts, interiors whose int32 digit sums cannot overflow
// 256 threads while three workgroups or more fit a CU's LDS, 1,024 (one workgroup per CU) for the larger tiles the LDS still holds
static int gf_lsop_predict16_threads(int nRows, int nCols)
{
    const size_t nCells = (size_t)nRows * (size_t)nCols;
    return 2 * ((nCells + 64 + 15) & ~(size_t)15) + sizeof(LsopShared16) <= 53 * 1024 ? 256 : 1024;
}
bool gf_lsop_predict16_eligible(int nRows, int nCols)
{
    const size_t nCells = (size_t)nRows * (size_t)nCols, nInt = (size_t)(nRows - 2) * (size_t)(nCols - 4);
    return nRows >= 6 && nCols >= 6 && (size_t)nCols <= LSOP_RING_MAXC && nInt < (1u << 17) &&
           2 * ((nCells + 64 + 15) & ~(size_t)15) + sizeof(LsopShared16) <= 150 * 1024;
}
size_t gf_lsop_hist_rec_words() { return LSOP_HIST_REC_WORDS; }

// k_lsop_predict16, then k_lsop_predict for the tiles it marked GF_K_LSOP_RETRY (none on terrain)
hipError_t gf_launch_lsop_predict16(const int32_t *values, int32_t *residuals, size_t resStride, uint32_t *coefs, int32_t *status,
                                    uint32_t *hist, size_t nTiles, int nRows, int nCols, hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    GfLsopPredict16Args a{values, reinterpret_cast<int16_t *>(residuals), 2 * resStride, coefs, status, hist, nTiles, nRows, nCols};
    const size_t dyn = 2 * (((size_t)nRows * (size_t)nCols + 64 + 15) & ~(size_t)15);      // the two digit planes
    static GfDynLdsOptIn opt;
    if (gf_lsop_predict16_threads(nRows, nCols) == 1024) {
        static GfDynLdsOptIn optW;
        const hipError_t e = gf_opt_in_dyn_lds((k_lsop_predict16<false, 1024>), dyn, optW);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_lsop_predict16<false, 1024>), gf_tile_grid(nTiles), dim3(1024), dyn, stream, a);
    } else {
        const hipError_t e = gf_opt_in_dyn_lds((k_lsop_predict16<false, 256>), dyn, opt);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_lsop_predict16<false, 256>), gf_tile_grid(nTiles), dim3(256), dyn, stream, a);
    }
    GfLsopPredictArgs b{values, residuals, resStride, coefs, status, nTiles, nRows, nCols, 1};
    const size_t dynB = (size_t)nCols <= LSOP_RING_MAXC ? ((size_t)4 * (size_t)nCols + 40) * 4 : 0;
    hipLaunchKernelGGL(k_lsop_predict, gf_tile_grid(nTiles), dim3(256), dynB, stream, b);
    return hipGetLastError();
}

hipError_t gf_launch_lsop_value_crc(const int32_t *values, size_t nCells, size_t nTiles, const int32_t *inStatus, uint32_t *coefs,
                                    hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    if (nCells >= (1ull << 30)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_lsop_value_crc, dim3((unsigned)((nTiles + 3) / 4)), dim3(256), 0, stream, values, (uint32_t)nCells, nTiles,
                       inStatus, coefs);
    return hipGetLastError();
}

hipError_t gf_launch_lsop_reconstruct(const int32_t *residuals, size_t resStride, const uint32_t *coefs, const int32_t *inStatus,
                                      int32_t *values, int32_t *status, size_t nTiles, int nRows, int nCols,
                                      hipStream_t stream, bool planes)
{
    if (nTiles == 0) return hipSuccess;
    const GfLsopPlaneGeom g = gf_lsop_plane_geom((uint32_t)nRows, (uint32_t)nCols);
#ifndef GF_RP_LDS_PAD
#define GF_RP_LDS_PAD 0
#endif
    planes = planes && g.ok;
    GfLsopReconArgs a{residuals, resStride, coefs, inStatus, values, status, nTiles, nRows, nCols, planes};
    const size_t dyn = (64 * (2 * RECON_ROUND + 1) + 2 * (size_t)nCols) * 4;      // 64 staging rings + 2 row buffers
    const bool pipe = dyn <= 96 * 1024 && nRows > 66 && nCols >= 16;               // more than one band of 64 rows: one pipeline down the tile
    if (planes) {
        // the tiles whose interior residuals lie as byte planes (k_lsop_unpack2 says which: terrain, nearly all of them), two to a
        // wave; the others through the old pipeline by the launch's first workgroups where the shape is that kernel's, else by the
        // kernels below
        const size_t dynP = std::max<size_t>(g.ldsBytes + GF_RP_LDS_PAD, pipe ? dyn : 0);      // (GF_RP_LDS_PAD: experiment builds, fewer waves per CU)
        static GfDynLdsOptIn optPlane;
        const hipError_t e = gf_opt_in_dyn_lds(k_lsop_reconstruct_plane, dynP, optPlane);
        if (e != hipSuccess) return e;
        const size_t pairs = (nTiles + 1) / 2;                   // two tiles to a wave
        // (a workgroup per tile, as in that kernel's own launch: one that finds a plane tile leaves at once -- a batch may as well
        // consist of tiles with wide residuals only)
        const unsigned nOld = pipe ? (unsigned)std::min<size_t>(nTiles, 65536 * 16) : 0u;
        const unsigned gridP = nOld + (unsigned)(pairs < 65536 * 16 ? pairs : 65536 * 16);
        hipLaunchKernelGGL(k_lsop_reconstruct_plane, dim3(gridP), dim3(64), dynP, stream, a, g, nOld);
        if (pipe) return hipGetLastError();
    }
    if (dyn <= 96 * 1024) {
        static GfDynLdsOptIn opt;
        const hipError_t e = gf_opt_in_dyn_lds(k_lsop_reconstruct<RECON_ROUND>, dyn, opt);
        if (e != hipSuccess) return e;
        const unsigned grid = (unsigned)(nTiles < 65536 * 16 ? nTiles : 65536 * 16);
        if (pipe) {
            static GfDynLdsOptIn optPipe;
            const hipError_t e2 = gf_opt_in_dyn_lds(k_lsop_reconstruct_pipe<RECON_ROUND>, dyn, optPipe);
            if (e2 != hipSuccess) return e2;
            hipLaunchKernelGGL(k_lsop_reconstruct_pipe<RECON_ROUND>, dim3(grid), dim3(64), dyn, stream, a);
        } else hipLaunchKernelGGL(k_lsop_reconstruct<RECON_ROUND>, dim3(grid), dim3(64), dyn, stream, a);
    } else {
        const size_t wgs = (nTiles + 3) / 4;
        const unsigned grid = (unsigned)(wgs < 65536 * 16 ? wgs : 65536 * 16);
        hipLaunchKernelGGL(k_lsop_reconstruct_global, dim3(grid), dim3(256), 0, stream, a);   // very wide tiles
    }
    return hipGetLastError();
}
