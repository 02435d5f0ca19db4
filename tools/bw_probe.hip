// bw_probe: what does HBM give a kernel shaped like the encoder's phase A?  933 MB of int32 "tiles" (12,960 x 18,000 cells), one
// 256-thread workgroup per tile, eight cells per lane and turn.  (a) the tile read once with two 16-byte loads per lane; (b) the
// same + the row above (two more 16-byte loads at a 600-byte offset: L1 / L2 hits) + three halo words = phase A's loads; (c) = (b) + an
// 8-byte store per lane (the byte plane); (d) = (a) + that store.  Prints ms and TB/s of the bytes that must come from / go to HBM.
// build: hipcc -O3 --offload-arch=gfx950 tools/bw_probe.hip -o tools/bin/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) U2 { uint32_t x, y; };
template <int MODE, int WGS>
__global__ __launch_bounds__(256, WGS) void k(const uint32_t *__restrict__ v, uint8_t *__restrict__ plane, uint32_t *sink, uint32_t nCells, uint32_t nC)
{
    const uint32_t *tile = v + (size_t)blockIdx.x * nCells;
    uint32_t acc = 0;
    for (uint32_t i0 = threadIdx.x * 8; i0 + 7 < nCells; i0 += 2048) {
        uint32_t s = i0;
        if (MODE != 4) {
            const U4 a = *reinterpret_cast<const U4 *>(tile + i0), b = *reinterpret_cast<const U4 *>(tile + i0 + 4);
            s = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
        }
        if (MODE == 5 || MODE == 7) {                                  // the row above from memory only for the lanes whose row above lies in front of the wave's span
            if (i0 >= nC + 2 && (threadIdx.x & 63u) < 19u) {
                const U4 c = *reinterpret_cast<const U4 *>(tile + i0 - nC), d = *reinterpret_cast<const U4 *>(tile + i0 - nC + 4);
                s ^= c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w ^ tile[i0 - 1] ^ tile[i0 - 2] ^ tile[i0 - nC - 1];
            }
        }
        if (MODE == 6) {
            if (i0 >= nC + 2) {
                const U4 c = *reinterpret_cast<const U4 *>(tile + i0 - nC), d = *reinterpret_cast<const U4 *>(tile + i0 - nC + 4);
                s ^= c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
            }
        }
        if (MODE == 1 || MODE == 2) {
            if (i0 >= nC + 2) {
                const U4 c = *reinterpret_cast<const U4 *>(tile + i0 - nC), d = *reinterpret_cast<const U4 *>(tile + i0 - nC + 4);
                s ^= c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w ^ tile[i0 - 1] ^ tile[i0 - 2] ^ tile[i0 - nC - 1];
            }
        }
        if (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 7) {
            U2 w; w.x = s; w.y = s * 3u;
            *reinterpret_cast<U2 *>(plane + (size_t)blockIdx.x * nCells + i0) = w;
        }
        acc += s;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int MODE, int WGS>
void run(const char *name, const uint32_t *v, uint8_t *plane, uint32_t *sink, int nT, uint32_t nCells, double bytes)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<MODE, WGS>), dim3(nT), dim3(256), 0, 0, v, plane, sink, nCells, 150u);
    hipEventRecord(e0);
    const int N = 20;
    for (int i = 0; i < N; i++) hipLaunchKernelGGL((k<MODE, WGS>), dim3(nT), dim3(256), 0, 0, v, plane, sink, nCells, 150u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= N;
    printf("{\"case\": \"%s\", \"wgs_per_cu\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", name, WGS, ms, bytes / (ms * 1e-3) / 1e12);
}
// does a buffer that one kernel wrote come back faster to the next one when it is small (the 256 MB Infinity Cache in front of HBM)?
__global__ __launch_bounds__(256) void k_fill(uint4 *p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ __launch_bounds__(256) void k_sum(const uint4 *__restrict__ p, size_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
static void cache_probe(uint32_t *sink)
{
    uint4 *buf; hipMalloc(&buf, (size_t)1 << 30);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb = 16; mb <= 1024; mb *= 2) {
        const size_t n16 = mb * (1 << 20) / 16;
        float rd = 0, wr = 0;
        const int N = 10;
        for (int i = 0; i < N + 2; i++) {
            float ms;
            hipEventRecord(e0); hipLaunchKernelGGL(k_fill, dim3(256 * 16), dim3(256), 0, 0, buf, n16); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (i >= 2) wr += ms;
            hipEventRecord(e0); hipLaunchKernelGGL(k_sum, dim3(256 * 16), dim3(256), 0, 0, buf, n16, sink); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (i >= 2) rd += ms;
        }
        printf("{\"case\": \"write then read\", \"MB\": %zu, \"write_TBps\": %.3f, \"read_TBps\": %.3f}\n", mb, mb * 1.048576e6 / (wr / N * 1e-3) / 1e12, mb * 1.048576e6 / (rd / N * 1e-3) / 1e12);
    }
    hipFree(buf);
}
int main()
{
    const int nT = 12960; const uint32_t nCells = 18000;
    uint32_t *v, *sink; uint8_t *plane;
    hipMalloc(&v, (size_t)nT * nCells * 4); hipMalloc(&plane, (size_t)nT * nCells + 64); hipMalloc(&sink, 64);
    hipMemset(v, 1, (size_t)nT * nCells * 4);
    const double rd = (double)nT * nCells * 4, wr = (double)nT * nCells;
    run<0, 8>("read tile once", v, plane, sink, nT, nCells, rd);
    run<0, 4>("read tile once", v, plane, sink, nT, nCells, rd);
    run<1, 8>("phase A's loads", v, plane, sink, nT, nCells, rd);
    run<1, 7>("phase A's loads", v, plane, sink, nT, nCells, rd);
    run<2, 7>("phase A's loads + plane store", v, plane, sink, nT, nCells, rd + wr);
    run<3, 8>("read once + plane store", v, plane, sink, nT, nCells, rd + wr);
    run<4, 8>("plane store alone", v, plane, sink, nT, nCells, wr);
    run<5, 7>("tile + row above for 19 lanes of 64 + halo", v, plane, sink, nT, nCells, rd);
    run<6, 7>("tile + row above, no halo words", v, plane, sink, nT, nCells, rd);
    run<7, 7>("tile + row above for 19 lanes + halo + plane store", v, plane, sink, nT, nCells, rd + wr);
    cache_probe(sink);
    return 0;
}
