// mfma_i8_gram_check.hip -- the operand and result lane maps of v_mfma_i32_32x32x32_i8 as k_lsop_predict uses it (round 4): the
// Gram matrix D^T D of a 32-row x 32-column int8 block with ONE register quadruple as both operands.  Lane l holds column l & 31 of
// D for the sixteen rows 16 (l >> 5) .. + 15; the result register r of lane l is C[row (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col l & 31]
// (cdna_hip_programming.md: the C/D map is dtype-independent).  Checked with exact integer data against the host's sums.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_i8_gram_check.hip -o tools/bin/mfma_i8_gram_check && tools/bin/mfma_i8_gram_check
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k(const int8_t *D, int *C)          // D[k][col], 32 x 32, row-major; C[32][32]
{
    const int l = threadIdx.x, col = l & 31, h = l >> 5;
    v4i x;
    for (int q = 0; q < 4; q++) {
        unsigned w = 0;
        for (int b = 0; b < 4; b++) w |= (unsigned)(unsigned char)D[(16 * h + 4 * q + b) * 32 + col] << (8 * b);
        x[q] = (int)w;
    }
    v16i acc = {};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, x, acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + col] = acc[r];
}

int main()
{
    int8_t hD[1024];
    int hC[1024];
    srand(7);
    for (int i = 0; i < 1024; i++) hD[i] = (int8_t)(rand() % 256 - 128);
    int8_t *dD;
    int *dC;
    if (hipMalloc(&dD, 1024) != hipSuccess || hipMalloc(&dC, 4096) != hipSuccess) return 2;
    (void)hipMemcpy(dD, hD, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dD, dC);
    (void)hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            int s = 0;
            for (int kk = 0; kk < 32; kk++) s += (int)hD[kk * 32 + i] * (int)hD[kk * 32 + j];
            if (s != hC[i * 32 + j]) bad++;
        }
    printf("v_mfma_i32_32x32x32_i8 Gram check: %d of 1024 entries differ\n", bad);
    return bad ? 1 : 0;
}
