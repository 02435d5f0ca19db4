// gvrs_float.hip -- CodecFloat byte planes on the GPU.
//
// Replaces (reference, core/src/main/java/org/gridfour/compress/CodecFloat.java):
//   encodeFloats :328-369   raw IEEE-754 bits -> sign-bit plane (LSB first), exponent plane, three
//                           mantissa byte planes, each mantissa plane byte-delta coded (:300-313:
//                           within a row b[k]-b[k-1]; the first cell of a row is relative to the
//                           first cell of the previous row, row 0 to 0)
//   decodeFloats :395-458   the inverse (decodeDeltas :315-325 = byte-wise running sums)
// The Deflate stage of the five planes (doDeflate :268-283) stays on the host (libz) because its
// bytes are only defined by zlib itself; see gvrs_api.hip gf_float_*.
//
// Plane buffer of a tile (same layout as the oracle): [sign: ceil(n/8)] [exponent: n] [m1: n]
// [m2: n] [m3: n], tiles at a fixed stride.  Pure streaming: 4 B/cell in, 4.125 B/cell out.

#include <hip/hip_runtime.h>

#include "gvrs_kernels.h"
#include "gvrs_common.h"

namespace {

constexpr int FLT_THREADS = 256;

// a dword of four plane bytes at any byte address (the planes of a tile follow one another without padding)
struct __attribute__((packed, aligned(1))) PlaneWord { uint32_t v; };
struct __attribute__((packed, aligned(16))) Cells4 { uint32_t x, y, z, w; };

__global__ __launch_bounds__(FLT_THREADS) void k_float_planes_encode(const uint32_t *__restrict__ raw, uint8_t *__restrict__ planes,
                                                                    size_t planeStride, size_t nTiles, int nRows, int nCols)
{
    const uint32_t nC = (uint32_t)nCols, n = (uint32_t)nRows * nC;
    const uint32_t nSign = (n + 7u) >> 3;
    const int lane = threadIdx.x & 63;
    GF_FOR_WG_TILE(t, nTiles) {                                           // no tile loop: see gvrs_kernels.h
        const uint32_t *__restrict__ c = raw + t * (size_t)n;
        uint8_t *pSign = planes + t * planeStride;
        uint8_t *pExp = pSign + nSign, *pM1 = pExp + n, *pM2 = pM1 + n, *pM3 = pM2 + n;
        if ((nC & 3u) == 0 && (((uintptr_t)c) & 15u) == 0) {
            // four cells of one row per lane: one 16-byte load, one dword store per byte plane; the sign bits of 32
            // cells are gathered over 8 lanes into one dword
            for (uint32_t i0 = threadIdx.x * 4u; i0 < ((n + 31u) & ~31u); i0 += FLT_THREADS * 4u) {
                const bool in = i0 < n;
                Cells4 q{0u, 0u, 0u, 0u};
                uint32_t p = 0;
                if (in) {
                    q = *reinterpret_cast<const Cells4 *>(c + i0);
                    const uint32_t col = i0 % nC;
                    p = col > 0 ? c[i0 - 1] : (i0 >= nC ? c[i0 - nC] : 0u);
                }
                const uint32_t v0 = q.x, v1 = q.y, v2 = q.z, v3 = q.w;
                uint32_t sg = ((v0 >> 31) | ((v1 >> 31) << 1) | ((v2 >> 31) << 2) | ((v3 >> 31) << 3)) << (4u * (lane & 7));
                sg |= gf_lane_xor(sg, 1);
                sg |= gf_lane_xor(sg, 2);
                sg |= gf_lane_xor(sg, 4);
                if (in) {
                    if ((lane & 7) == 0) {
                        if (i0 + 32u <= n) reinterpret_cast<PlaneWord *>(pSign + (i0 >> 3))->v = sg;
                        else for (uint32_t b = 0; i0 + 8u * b < n; b++) pSign[(i0 >> 3) + b] = (uint8_t)(sg >> (8u * b));
                    }
                    auto bytes4 = [](uint32_t a, uint32_t b, uint32_t cc, uint32_t d) -> uint32_t {
                        return (a & 0xffu) | ((b & 0xffu) << 8) | ((cc & 0xffu) << 16) | (d << 24);
                    };
                    reinterpret_cast<PlaneWord *>(pExp + i0)->v = bytes4(v0 >> 23, v1 >> 23, v2 >> 23, v3 >> 23);
                    reinterpret_cast<PlaneWord *>(pM1 + i0)->v =
                        bytes4(((v0 >> 16) & 0x7fu) - ((p >> 16) & 0x7fu), ((v1 >> 16) & 0x7fu) - ((v0 >> 16) & 0x7fu),
                               ((v2 >> 16) & 0x7fu) - ((v1 >> 16) & 0x7fu), ((v3 >> 16) & 0x7fu) - ((v2 >> 16) & 0x7fu));
                    reinterpret_cast<PlaneWord *>(pM2 + i0)->v =
                        bytes4((v0 >> 8) - (p >> 8), (v1 >> 8) - (v0 >> 8), (v2 >> 8) - (v1 >> 8), (v3 >> 8) - (v2 >> 8));
                    reinterpret_cast<PlaneWord *>(pM3 + i0)->v = bytes4(v0 - p, v1 - v0, v2 - v1, v3 - v2);
                }
            }
            continue;
        }
        // any shape: whole waves walk 64 consecutive cells so that one ballot yields 8 sign bytes
        for (uint32_t base = gf_wave_id() * 64u; base < n; base += FLT_THREADS) {
            const uint32_t i = base + lane;
            const bool in = i < n;
            const uint32_t v = in ? c[i] : 0u;
            const unsigned long long sb = __ballot(in && (v >> 31));
            if (lane < 8 && base + 8u * lane < n) pSign[(base >> 3) + lane] = (uint8_t)(sb >> (8 * lane));
            if (in) {
                const uint32_t col = i % nC;
                // prior cell of the delta rule: left neighbour, or the first cell of the previous row
                const uint32_t p = col > 0 ? c[i - 1] : (i >= nC ? c[i - nC] : 0u);
                pExp[i] = (uint8_t)(v >> 23);
                pM1[i] = (uint8_t)(((v >> 16) & 0x7fu) - ((p >> 16) & 0x7fu));
                pM2[i] = (uint8_t)((v >> 8) - (p >> 8));
                pM3[i] = (uint8_t)(v - p);
            }
        }
    }
}

__global__ __launch_bounds__(FLT_THREADS) void k_float_planes_decode(const uint8_t *__restrict__ planes, uint32_t *__restrict__ raw,
                                                                    size_t planeStride, size_t nTiles, int nRows, int nCols)
{
    __shared__ uint32_t col0[3][1024];          // decoded first cells of the rows (three mantissa planes), by chunk
    const uint32_t nR = (uint32_t)nRows, nC = (uint32_t)nCols, n = nR * nC;
    const uint32_t nSign = (n + 7u) >> 3;
    const int lane = threadIdx.x & 63, wave = (int)gf_wave_id();
    GF_FOR_WG_TILE(t, nTiles) {                                           // no tile loop: see gvrs_kernels.h
        const uint8_t *pSign = planes + t * planeStride;
        const uint8_t *pExp = pSign + nSign, *pM1 = pExp + n, *pM2 = pM1 + n, *pM3 = pM2 + n;
        uint32_t *o = raw + t * (size_t)n;
        // rows are processed in chunks of 1024: column-0 chain of the chunk (wave 0, running sums mod 256), then
        // every wave scans its rows
        uint32_t carry1 = 0, carry2 = 0, carry3 = 0;            // first cell of the row before the chunk
        for (uint32_t r0 = 0; r0 < nR; r0 += 1024) {
            const uint32_t rows = min(1024u, nR - r0);
            if (wave == 0) {
                for (uint32_t rb = 0; rb < rows; rb += 64) {
                    const uint32_t r = r0 + rb + lane;
                    const bool in = rb + lane < rows;
                    uint32_t a = in ? pM1[(size_t)r * nC] : 0u, b = in ? pM2[(size_t)r * nC] : 0u, d = in ? pM3[(size_t)r * nC] : 0u;
                    a = gf_wave_incl_scan(a) + carry1;
                    b = gf_wave_incl_scan(b) + carry2;
                    d = gf_wave_incl_scan(d) + carry3;
                    if (in) { col0[0][rb + lane] = a & 0xffu; col0[1][rb + lane] = b & 0xffu; col0[2][rb + lane] = d & 0xffu; }
                    carry1 = (uint32_t)__builtin_amdgcn_readlane((int)a, 63) & 0xffu;
                    carry2 = (uint32_t)__builtin_amdgcn_readlane((int)b, 63) & 0xffu;
                    carry3 = (uint32_t)__builtin_amdgcn_readlane((int)d, 63) & 0xffu;
                }
            }
            __syncthreads();
            carry1 = col0[0][rows - 1];
            carry2 = col0[1][rows - 1];
            carry3 = col0[2][rows - 1];
            const bool quads = (nC & 3u) == 0 && (((uintptr_t)o) & 15u) == 0;
            for (uint32_t rr = wave; quads && rr < rows; rr += FLT_THREADS / 64) {
                // four columns per lane: a dword per plane, byte-wise running sums inside the lane, one wave scan of the
                // lane totals per plane, one 16-byte store
                const uint32_t r = r0 + rr;
                const size_t rowOff = (size_t)r * nC;
                uint32_t c1 = 0, c2 = 0, c3 = 0;
                for (uint32_t cb = 0; cb < nC; cb += 256) {
                    const uint32_t cc = cb + 4u * lane;
                    const bool in = cc < nC;
                    const size_t i = rowOff + cc;
                    uint32_t w1 = 0, w2 = 0, w3 = 0, we = 0, sb = 0;
                    if (in) {
                        w1 = reinterpret_cast<const PlaneWord *>(pM1 + i)->v;
                        w2 = reinterpret_cast<const PlaneWord *>(pM2 + i)->v;
                        w3 = reinterpret_cast<const PlaneWord *>(pM3 + i)->v;
                        we = reinterpret_cast<const PlaneWord *>(pExp + i)->v;
                        sb = ((uint32_t)pSign[i >> 3] >> (i & 7)) & 0xfu;          // i is a multiple of 4: one byte holds the four bits
                    }
                    if (cc == 0) {                                                 // first cell of the row: already decoded
                        w1 = (w1 & ~0xffu) | col0[0][rr];
                        w2 = (w2 & ~0xffu) | col0[1][rr];
                        w3 = (w3 & ~0xffu) | col0[2][rr];
                    }
                    auto sums = [](uint32_t w, uint32_t *a, uint32_t *b, uint32_t *cc2, uint32_t *d) {
                        *a = w & 0xffu;
                        *b = *a + ((w >> 8) & 0xffu);
                        *cc2 = *b + ((w >> 16) & 0xffu);
                        *d = *cc2 + (w >> 24);
                    };
                    uint32_t a0, a1, a2, a3, b0, b1, b2, b3, d0, d1, d2, d3;
                    sums(w1, &a0, &a1, &a2, &a3);
                    sums(w2, &b0, &b1, &b2, &b3);
                    sums(w3, &d0, &d1, &d2, &d3);
                    const uint32_t ia = gf_wave_incl_scan(a3), ib = gf_wave_incl_scan(b3), id = gf_wave_incl_scan(d3);
                    const uint32_t ea = ia - a3 + c1, eb = ib - b3 + c2, ed = id - d3 + c3;   // sums of everything to the left
                    c1 += (uint32_t)__builtin_amdgcn_readlane((int)ia, 63);
                    c2 += (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
                    c3 += (uint32_t)__builtin_amdgcn_readlane((int)id, 63);
                    if (in) {
                        auto cell = [&](uint32_t k, uint32_t a, uint32_t b, uint32_t d) -> uint32_t {
                            return (((sb >> k) & 1u) << 31) | (((we >> (8u * k)) & 0xffu) << 23) | (((a + ea) & 0x7fu) << 16) |
                                   (((b + eb) & 0xffu) << 8) | ((d + ed) & 0xffu);
                        };
                        Cells4 q;
                        q.x = cell(0, a0, b0, d0);
                        q.y = cell(1, a1, b1, d1);
                        q.z = cell(2, a2, b2, d2);
                        q.w = cell(3, a3, b3, d3);
                        *reinterpret_cast<Cells4 *>(o + i) = q;
                    }
                }
            }
            for (uint32_t rr = wave; !quads && rr < rows; rr += FLT_THREADS / 64) {
                const uint32_t r = r0 + rr;
                const size_t rowOff = (size_t)r * nC;
                uint32_t c1 = 0, c2 = 0, c3 = 0;                // running sums carried along the row
                for (uint32_t cb = 0; cb < nC; cb += 64) {
                    const uint32_t cc = cb + lane;
                    const bool in = cc < nC;
                    uint32_t a, b, d;
                    if (cc == 0) { a = col0[0][rr]; b = col0[1][rr]; d = col0[2][rr]; }       // already decoded
                    else { a = in ? pM1[rowOff + cc] : 0u; b = in ? pM2[rowOff + cc] : 0u; d = in ? pM3[rowOff + cc] : 0u; }
                    a = gf_wave_incl_scan(a) + c1;
                    b = gf_wave_incl_scan(b) + c2;
                    d = gf_wave_incl_scan(d) + c3;
                    c1 = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
                    c2 = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
                    c3 = (uint32_t)__builtin_amdgcn_readlane((int)d, 63);
                    if (in) {
                        const size_t i = rowOff + cc;
                        const uint32_t s = (pSign[i >> 3] >> (i & 7)) & 1u;
                        o[i] = (s << 31) | ((uint32_t)pExp[i] << 23) | ((a & 0x7fu) << 16) | ((b & 0xffu) << 8) | (d & 0xffu);
                    }
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace

hipError_t gf_launch_float_planes_encode(const uint32_t *raw, uint8_t *planes, size_t planeStride, size_t nTiles, int nRows,
                                         int nCols, hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_float_planes_encode, gf_tile_grid(nTiles), dim3(FLT_THREADS), 0, stream, raw, planes, planeStride, nTiles, nRows, nCols);
    return hipGetLastError();
}

hipError_t gf_launch_float_planes_decode(const uint8_t *planes, uint32_t *raw, size_t planeStride, size_t nTiles, int nRows,
                                         int nCols, hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_float_planes_decode, gf_tile_grid(nTiles), dim3(FLT_THREADS), 0, stream, planes, raw, planeStride, nTiles, nRows, nCols);
    return hipGetLastError();
}
