// issue_rate.hip -- what a CU's issue ports sustain (gfx950): vector instructions per SIMD, scalar instructions and branches on the
// CU's scalar unit, at 1, 2, 4 and 8 waves per SIMD, every CU of the chip busy.  (Round 4: bench.py's second roofline,
// `roofline_issue`, is priced with these figures; tools/valu_rate.hip measured three vector mixes through the compiler and
// guessed their instruction counts.)
//
// The measured streams are INLINE ASSEMBLY, so the instruction counts are exact: every kind is a block of UNROLL identical
// groups inside a counted loop (the loop's own s_add / s_cmp / s_cbranch are counted with it).  Time is taken with HIP events
// around the launch and turned into cycles at 2.4 GHz (the clock bench.py prices with: what matters is time per instruction).
//
//   hipcc -O3 --offload-arch=gfx950 tools/issue_rate.hip -o tools/bin/issue_rate && tools/bin/issue_rate
//
// Output, one line per kind and occupancy: cycles per wave-instruction and SIMD (vector kinds), per wave-instruction and CU
// (scalar kinds) -- i.e. the reciprocal throughput of that port with that many waves feeding it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum { K_VADD = 0, K_VCMPSEL, K_VOP3, K_SADD, K_SCMP_NOT_TAKEN, K_SCMP_TAKEN, K_RFL_SALU, K_MIX_2V1S, K_MIX_1V1S, K_VDPP, K_COUNT };
static const char *kNames[K_COUNT] = {
    "v_add_u32 x8 independent",      "v_cmp_gt_u32 + v_cndmask_b32 pairs", "v_alignbit / v_bfe_u32 / v_lshl_or_b32 / v_mad_u32_u24",
    "s_add_u32 x8 independent",      "s_cmp_lg_u32 + s_cbranch_scc1 (not taken)", "s_cmp_eq_u32 + s_cbranch_scc1 (taken, to the next instruction)",
    "v_readfirstlane_b32 -> s_add_u32 (dependent)", "2 v_add_u32 : 1 s_add_u32 interleaved", "1 v_add_u32 : 1 s_add_u32 interleaved",
    "v_add_u32_dpp row_shr:1 chain of 4 + s_nop 1 each"};
// vector / scalar / branch instructions of ONE group of each kind
static const int kVec[K_COUNT] = {8, 8, 8, 0, 0, 0, 4, 8, 4, 4};
static const int kSca[K_COUNT] = {0, 0, 0, 8, 4, 4, 4, 4, 4, 4};
static const int kBr[K_COUNT] = {0, 0, 0, 0, 4, 4, 0, 0, 0, 0};

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters)
{
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = a | 9, f = a * 11, g = a + 13, h = a ^ 17;
    uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
    for (int i = 0; i < iters; i++) {
        if (KIND == K_VADD)
            asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n"
                               "v_add_u32 %1, %1, %0\n v_add_u32 %3, %3, %2\n v_add_u32 %5, %5, %4\n v_add_u32 %7, %7, %6\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        if (KIND == K_VCMPSEL)
            asm volatile(REP16("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_gt_u32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %4, vcc\n"
                               "v_cmp_gt_u32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %6, vcc\n v_cmp_gt_u32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %0, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
                         :
                         : "vcc");
        if (KIND == K_VOP3)
            asm volatile(REP16("v_alignbit_b32 %0, %0, %1, %2\n v_bfe_u32 %2, %2, 3, 6\n v_lshl_or_b32 %4, %4, 2, %5\n v_mad_u32_u24 %6, %6, 3, %7\n"
                               "v_alignbit_b32 %1, %1, %0, %3\n v_bfe_u32 %3, %3, 3, 6\n v_lshl_or_b32 %5, %5, 2, %4\n v_mad_u32_u24 %7, %7, 3, %6\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        if (KIND == K_SADD)
            asm volatile(REP16("s_add_u32 %0, %0, %1\n s_add_u32 %2, %2, %3\n s_add_u32 %4, %4, %5\n s_add_u32 %6, %6, %7\n"
                               "s_add_u32 %1, %1, %0\n s_add_u32 %3, %3, %2\n s_add_u32 %5, %5, %4\n s_add_u32 %7, %7, %6\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)
                         :
                         : "scc");
        if (KIND == K_SCMP_NOT_TAKEN)        // s0 == s0: "not equal" is false, the branch falls through
            asm volatile(REP16("s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 1f\n 1: s_cmp_lg_u32 %1, %1\n s_cbranch_scc1 2f\n 2: s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 3f\n"
                               "3: s_cmp_lg_u32 %1, %1\n s_cbranch_scc1 4f\n 4:\n")
                         : "+s"(s0), "+s"(s1)
                         :
                         : "scc");
        if (KIND == K_SCMP_TAKEN)
            asm volatile(REP16("s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n 1: s_cmp_eq_u32 %1, %1\n s_cbranch_scc1 2f\n 2: s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 3f\n"
                               "3: s_cmp_eq_u32 %1, %1\n s_cbranch_scc1 4f\n 4:\n")
                         : "+s"(s0), "+s"(s1)
                         :
                         : "scc");
        if (KIND == K_RFL_SALU)
            asm volatile(REP16("v_readfirstlane_b32 %4, %0\n s_add_u32 %5, %5, %4\n v_readfirstlane_b32 %6, %1\n s_add_u32 %7, %7, %6\n"
                               "v_readfirstlane_b32 %4, %2\n s_add_u32 %5, %5, %4\n v_readfirstlane_b32 %6, %3\n s_add_u32 %7, %7, %6\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                         :
                         : "scc");
        if (KIND == K_MIX_2V1S)
            asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n s_add_u32 %8, %8, %9\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n s_add_u32 %10, %10, %11\n"
                               "v_add_u32 %1, %1, %0\n v_add_u32 %3, %3, %2\n s_add_u32 %9, %9, %8\n v_add_u32 %5, %5, %4\n v_add_u32 %7, %7, %6\n s_add_u32 %11, %11, %10\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                         :
                         : "scc");
        if (KIND == K_MIX_1V1S)
            asm volatile(REP16("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_add_u32 %2, %2, %3\n s_add_u32 %6, %6, %7\n"
                               "v_add_u32 %1, %1, %0\n s_add_u32 %5, %5, %4\n v_add_u32 %3, %3, %2\n s_add_u32 %7, %7, %6\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                         :
                         : "scc");
        if (KIND == K_VDPP)                  // the wave scans of the kernels: a dependent DPP step needs two wait states
            asm volatile(REP16("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
}

template <int KIND>
static float run(uint32_t *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return best;
}

int main()
{
    uint32_t *d;
    if (hipMalloc(&d, (size_t)1 << 26) != hipSuccess) return 1;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double clock = 2.4e9;
    const int iters = 400;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_hz_assumed\": %.0f, \"kinds\": [\n", p.name, cus, clock);
    for (int kind = 0; kind < K_COUNT; kind++) {
        printf("  {\"kind\": \"%s\", \"vector_per_group\": %d, \"scalar_per_group\": %d, \"branch_per_group\": %d, \"rates\": [", kNames[kind],
               kVec[kind], kSca[kind], kBr[kind]);
        int first = 1;
        for (int wpc : {4, 8, 16, 32}) {     // waves per CU -> 1, 2, 4, 8 per SIMD
            const int blocks = cus * (wpc / 4);
            float ms = 0;
            switch (kind) {
            case K_VADD: ms = run<K_VADD>(d, blocks, iters); break;
            case K_VCMPSEL: ms = run<K_VCMPSEL>(d, blocks, iters); break;
            case K_VOP3: ms = run<K_VOP3>(d, blocks, iters); break;
            case K_SADD: ms = run<K_SADD>(d, blocks, iters); break;
            case K_SCMP_NOT_TAKEN: ms = run<K_SCMP_NOT_TAKEN>(d, blocks, iters); break;
            case K_SCMP_TAKEN: ms = run<K_SCMP_TAKEN>(d, blocks, iters); break;
            case K_RFL_SALU: ms = run<K_RFL_SALU>(d, blocks, iters); break;
            case K_MIX_2V1S: ms = run<K_MIX_2V1S>(d, blocks, iters); break;
            case K_MIX_1V1S: ms = run<K_MIX_1V1S>(d, blocks, iters); break;
            default: ms = run<K_VDPP>(d, blocks, iters); break;
            }
            // per wave: iters x (16 groups + the loop's s_add, s_cmp, s_cbranch)
            const double groups = (double)iters * 16.0;
            const double vec = groups * kVec[kind], sca = groups * (kSca[kind] + kBr[kind]) + 3.0 * iters;
            const double cyc = ms * 1e-3 * clock;
            const double perSimdVec = vec > 0 ? cyc / (vec * (wpc / 4.0)) : 0.0;      // cycles per vector wave-instruction and SIMD
            const double perCuSca = sca > 0 ? cyc / (sca * wpc) : 0.0;                // cycles per scalar-side wave-instruction and CU
            printf("%s{\"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_vector_instr_per_simd\": %.3f, \"cycles_per_scalar_instr_per_cu\": %.3f}",
                   first ? "" : ", ", wpc / 4, ms, perSimdVec, perCuSca);
            first = 0;
        }
        printf("]}%s\n", kind + 1 < K_COUNT ? "," : "");
    }
    printf("]}\n");
    return 0;
}
