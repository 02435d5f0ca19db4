// microbenchmark: integer VALU issue rate per SIMD at 1..8 waves per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ void k(uint32_t *out, int iters)
{
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = a | 9, f = a * 11, g = a + 13, h = a ^ 17;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (KIND == 0) { a += b; c += d; e += f; g += h; b ^= a; d ^= c; f ^= e; h ^= g; }                 // 8 independent-ish int adds/xors
            if (KIND == 1) { a = a > b ? c : a; c = c > d ? e : c; e = e > f ? g : e; g = g > h ? a : g; b += 1; d += 1; f += 1; h += 1; }   // cmp + cndmask
            if (KIND == 2) { a = __builtin_amdgcn_alignbit(a, b, c & 31); c = (c >> 3) & 63; e = (e << 2) | f; g = g * 3 + h; b += a; d += c; f += e; h += g; }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0) reinterpret_cast<uint64_t *>(out + (1 << 20))[blockIdx.x] = t1 - t0;
}
int main()
{
    uint32_t *d;
    (void)hipMalloc(&d, (1 << 24) + (1 << 20));
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int kind = 0; kind < 3; kind++)
        for (int wpc : {4, 8, 16, 32}) {          // waves per CU -> 1, 2, 4, 8 per SIMD
            const int threads = 256, blocksPerCu = wpc / 4, iters = 2000;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(cus * blocksPerCu), dim3(threads), 0, 0, d, iters);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(cus * blocksPerCu), dim3(threads), 0, 0, d, iters);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(cus * blocksPerCu), dim3(threads), 0, 0, d, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // instructions per wave: iters * 16 * ~8 (kind 0: 8; kind 1: 4 cmp + 4 cndmask + 4 add = 12; kind 2: ~12)
            const double perWave = (double)iters * 16 * (kind == 0 ? 8 : 12);
            const double wavesPerSimd = wpc / 4.0;
            const double cyc = ms * 1e-3 * 2.4e9;
            printf("kind %d waves/SIMD %.0f: %.3f ms, ~%.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", kind, wavesPerSimd, ms,
                   cyc / (perWave * wavesPerSimd));
        }
    return 0;
}
