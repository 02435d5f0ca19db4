// Calibration of the L2 memory-side counters on gfx950: the same 1 GiB buffer is read once by four kernels that differ
// only in the width of the per-lane load (4, 8, 16 bytes, coalesced across the wave) and once with 4-byte loads whose
// wave covers two 128-byte row pieces at unrelated addresses (the access shape of k_lsop_reconstruct's staging), and
// written once with 4- and 16-byte stores.  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`: the ratio
// counter x 1024 / 2^30 per kernel says whether the counter halves that access shape.
//   hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr size_t BYTES = 1ull << 30;
template <typename T>
__global__ void k_read(const T *__restrict__ in, uint32_t *out, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = in[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += w[k];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_read_rows(const uint32_t *__restrict__ in, uint32_t *out, size_t n)      // half-waves on rows 73 KB apart
{
    uint32_t acc = 0;
    const size_t rowWords = 18000, nRows = n / rowWords;
    const unsigned half = (threadIdx.x & 63) >> 5, j = threadIdx.x & 31;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave * 2; r + 1 < nRows; r += nWaves * 2)
        for (size_t c = 0; c + 32 <= rowWords; c += 32) acc += in[(r + half) * rowWords + c + j + 3];
    if (acc == 0x12345678u) out[0] = acc;
}
template <typename T>
__global__ void k_write(T *out, size_t n)
{
    T v;
    uint32_t *w = reinterpret_cast<uint32_t *>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) w[k] = threadIdx.x + k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}
int main()
{
    void *buf;
    uint32_t *out;
    if (hipMalloc(&buf, BYTES + 4096) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, BYTES);
    const dim3 grid(256 * 16), block(256);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_read<uint32_t>, grid, block, 0, 0, (const uint32_t *)buf, out, BYTES / 4);
        hipLaunchKernelGGL(k_read<uint2>, grid, block, 0, 0, (const uint2 *)buf, out, BYTES / 8);
        hipLaunchKernelGGL(k_read<uint4>, grid, block, 0, 0, (const uint4 *)buf, out, BYTES / 16);
        hipLaunchKernelGGL(k_read_rows, grid, block, 0, 0, (const uint32_t *)buf, out, BYTES / 4);
        hipLaunchKernelGGL(k_write<uint32_t>, grid, block, 0, 0, (uint32_t *)buf, BYTES / 4);
        hipLaunchKernelGGL(k_write<uint4>, grid, block, 0, 0, (uint4 *)buf, BYTES / 16);
    }
    const hipError_t e = hipDeviceSynchronize();
    printf("fetch_calib: %s, %zu bytes per kernel\n", hipGetErrorString(e), BYTES);
    return e == hipSuccess ? 0 : 1;
}
