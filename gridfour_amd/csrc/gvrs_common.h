// gvrs_common.h -- shared host/device helpers of the HIP GVRS codec.
//
// Everything here is integer arithmetic on uint32_t (Java int wrap-around).
// Reference paths are relative to core/src/main/java/org/gridfour/.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define GF_HD __host__ __device__ __forceinline__
#else
#define GF_HD inline
#endif

#define GF_NULL_CODE 0x80000000u

// 16-byte load from an address that is only 4-byte aligned (gfx950 global loads allow it)
struct __attribute__((packed, aligned(4))) GfU4 {
    uint32_t x, y, z, w;
};

struct __attribute__((packed, aligned(4))) GfU2 {
    uint32_t x, y;
};

// ---- CodecM32 (compress/CodecM32.java:257-311) -------------------------
// Number of M32 bytes of a residual (1..6).  Thresholds :105-111.
GF_HD int gf_m32_len(uint32_t x)
{
    if (x == GF_NULL_CODE) return 1;
    uint32_t a = ((int32_t)x < 0) ? (0u - x) : x;
    if (a <= 126u) return 1;
    if (a <= 254u) return 2;
    if (a <= 16638u) return 3;
    if (a <= 2113790u) return 4;
    if (a <= 270549246u) return 5;
    return 6;
}

// Byte k (0-based) of the n-byte M32 form of x.
//   n == 1 : the value itself (two's complement byte; 0x80 for the null code)
//   n  > 1 : byte 0 = introducer 0x7f / 0x81, then (abs - base[n-2]) in
//            big-endian 7-bit groups, continuation bit 0x80 on all but the last
GF_HD uint32_t gf_m32_byte(uint32_t x, int n, int k)
{
    if (n == 1) return x == GF_NULL_CODE ? 0x80u : (x & 0xffu);
    bool neg = (int32_t)x < 0;
    if (k == 0) return neg ? 0x81u : 0x7fu;
    uint32_t a = neg ? (0u - x) : x;
    uint32_t base = n == 2 ? 127u : n == 3 ? 255u : n == 4 ? 16639u : n == 5 ? 2113791u : 270549247u;
    uint32_t d = a - base;
    int shift = 7 * (n - 1 - k);              // k = n-1 -> 0
    uint32_t b = (d >> shift) & 0x7fu;
    return (k == n - 1) ? b : (b | 0x80u);
}

// ---- predictor residuals (compress/PredictorModel*.java) ---------------
// v = cell, W = (r,c-1), WW = (r,c-2), N = (r-1,c), NW = (r-1,c-1); callers pass
// any value for neighbours that do not exist.  Cell (0,0) has no residual.
GF_HD uint32_t gf_res_differencing(int r, int c, uint32_t v, uint32_t W, uint32_t N)
{
    (void)r;
    return c > 0 ? v - W : v - N;             // PredictorModelDifferencing.java:120-137
}
GF_HD uint32_t gf_res_linear(int r, int c, uint32_t v, uint32_t W, uint32_t WW, uint32_t N)
{
    (void)r;
    if (c >= 2) return v - (2u * W - WW);     // PredictorModelLinear.java:128-141
    return c == 1 ? v - W : v - N;            // :113-126
}
GF_HD uint32_t gf_res_triangle(int r, int c, uint32_t v, uint32_t W, uint32_t N, uint32_t NW)
{
    if (r == 0) return v - W;                 // PredictorModelTriangle.java:114-119
    if (c == 0) return v - N;                 // :121-127
    return v - (W + N - NW);                  // :130-142
}

// ---- stream order <-> cell (the order the reference emits residuals) ---
// number of residuals in the stream of a model
GF_HD uint32_t gf_stream_len(int model, uint32_t nR, uint32_t nC)
{
    return model == 4 ? nR * nC : nR * nC - 1u;
}

// cell index (row-major) of stream element s
GF_HD uint32_t gf_stream_cell(int model, uint32_t nR, uint32_t nC, uint32_t s)
{
    switch (model) {
    case 1: return s + 1u;
    case 2: {
        if (s == 0) return 1u;
        uint32_t seedLen = 2u * nR - 1u;      // 1 + 2*(nR-1)
        if (s < seedLen) {
            uint32_t t = s - 1u;
            return (1u + (t >> 1)) * nC + (t & 1u);
        }
        uint32_t t = s - seedLen, w = nC - 2u;
        uint32_t r = t / w;
        return r * nC + 2u + (t - r * w);
    }
    case 3: {
        if (s < nC - 1u) return s + 1u;
        uint32_t t = s - (nC - 1u);
        if (t < nR - 1u) return (t + 1u) * nC;
        t -= nR - 1u;
        uint32_t w = nC - 1u;
        uint32_t r = t / w;
        return (r + 1u) * nC + 1u + (t - r * w);
    }
    default: return s;
    }
}

#if defined(__HIPCC__)
// Value of lane (lane ^ j), j a compile-time power of two: DPP for j <= 8 (quad permutes, row mirrors and rotates:
// one or two VALU moves), ds_swizzle for 16, the LDS crossbar only for 32.  A __shfl_xor costs a ds_bpermute
// (> 100 cycles of latency) at every stage of a sorting network; 18 of the 21 stages of a 64-key bitonic sort have j <= 8.
__device__ __forceinline__ uint32_t gf_lane_xor(uint32_t v, int j)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int x = (int)v;
    switch (j) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);            // quad_perm [1,0,3,2]
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);            // quad_perm [2,3,0,1]
    case 4: {                                                                                      // i ^ 7, then reverse each quad
        const int h = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);                   // row_half_mirror
        return (uint32_t)__builtin_amdgcn_update_dpp(0, h, 0x1B, 0xf, 0xf, true);                 // quad_perm [3,2,1,0]
    }
    case 8: return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, true);            // row_ror:8
    case 16: return (uint32_t)__builtin_amdgcn_ds_swizzle(x, 0x401F);                             // bit mode: xor 0x10
    default: return (uint32_t)__shfl_xor(x, j, 64);
    }
#else
    (void)j;
    return v;                                                                                      // host pass of hipcc: never executed
#endif
}

// The wave's index inside its workgroup as a SCALAR (v_readfirstlane): written `threadIdx.x >> 6` the compiler has to treat it,
// and every loop bound, address and branch that derives from it, as per-lane values -- vector registers and exec-mask loops
// where scalar registers and scalar branches do (seen in the ISA of the canonical decoder's row loops, round 3).
__device__ __forceinline__ uint32_t gf_wave_id()
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
    return 0;                                                                                      // host pass of hipcc: never executed
#endif
}

// Wave64 inclusive prefix sum with DPP row shifts/broadcasts: 6 VALU steps, no LDS crossbar
// (a __shfl_up ladder costs 6 dependent ds_bpermute round trips).
__device__ __forceinline__ uint32_t gf_wave_incl_scan(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);    // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);    // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);    // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);    // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return (uint32_t)x;
#else
    return v;                                                          // host pass of hipcc: never executed
#endif
}
#endif
