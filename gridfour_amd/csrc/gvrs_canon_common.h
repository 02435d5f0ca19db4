// gvrs_canon_common.h -- device side of Gridfour's canonical-Huffman entropy stage
// (compress/canonicalHuffman/*.java), shared by the CodecCanonHuffman and LSOP kernels.
// Include inside the kernel file's anonymous namespace after gvrs_encode_common.h.
//
// One WAVE builds the complete code description of one integer stream from its 260-bin histogram:
//   cn_build():  code lengths with the reference's tree (TreeBuilder.java:75-188: leaves sorted by
//                count asc / symbol DESC, branch inserted before the first node with count >= its
//                own) via the data-parallel rounds of gvrs_encode.hip; PackageMerge.java:91-175 when
//                a length exceeds 15; canonical codes (TreeBuilder.java:283-300); the run-length
//                tokens of the 260 lengths (LengthEncoder.java:86-166); the 20-symbol meta tree;
//                and the serialised code tables (CanonicalHuffman.java:285-343) as a bit image.
// Reference paths are relative to core/src/main/java/org/gridfour/.
#pragma once

constexpr int CN_SYMS = 260;                     // CanonicalHuffman.java:74-80
constexpr int CN_NULL = 256;
constexpr int CN_ESC1 = 257;                     // one raw byte follows
constexpr int CN_ESC2 = 258;                     // two raw bits follow
constexpr int CN_EOT = 259;
constexpr int CN_HIST = 264;                     // histogram row (padded)
constexpr int CN_META = 20;                      // LengthEncoder.SYMBOL_SET_SIZE + 1 (end-of-text)
constexpr int CN_MAXLEN = 15;                    // LengthEncoder.MAX_STANDARD_SYMBOL
constexpr int CN_IMG_WORDS = 192;                // 1 + 20*12 + 260*22 = 5961 bits at most -> 187 words
constexpr uint32_t CN_DEAD = 0xFFFFFFFFu;

// value -> (target symbol, escape kind) as CanonicalHuffman.countSymbols :352-418 classifies it.
// kind: 0 none, 1..3 = that many 2-bit escapes, 4..6 = 1..3 one-byte escapes, 7 = null symbol
__device__ __forceinline__ uint32_t cn_classify_count(uint32_t x, uint32_t *kind)
{
    const int32_t s = (int32_t)x;
    if (x + 128u < 256u) { *kind = 0; return x + 128u; }
    if (x + 512u < 1024u) { *kind = 1; return (uint32_t)((s >> 2) + 128); }
    if (x + 2048u < 4096u) { *kind = 2; return (uint32_t)((s >> 4) + 128); }
    if (x + 8192u < 16384u) { *kind = 3; return (uint32_t)((s >> 6) + 128); }
    if (x + 32768u < 65536u) { *kind = 4; return (uint32_t)((s >> 8) + 128); }
    if (x == GF_NULL_CODE) { *kind = 7; return (uint32_t)CN_NULL; }
    if (x + 8388608u < 16777216u) { *kind = 5; return (uint32_t)((s >> 16) + 128); }
    *kind = 6;
    return (uint32_t)((s >> 24) + 128);
}

// the text loop of CanonicalHuffman.encode :203-276 tests -8333608 (sic) where countSymbols tests
// -8388608: values in [-8388608, -8333609] are COUNTED as two-byte escapes but WRITTEN as three-byte
// ones.  Reproduced: cn_is_gap() marks them, the emit side classifies them as kind 6.
__device__ __forceinline__ bool cn_is_gap(uint32_t x) { return x + 8388608u < 8388608u - 8333608u; }

__device__ __forceinline__ uint32_t cn_classify_emit(uint32_t x, uint32_t *kind)
{
    if (cn_is_gap(x)) { *kind = 6; return (uint32_t)(((int32_t)x >> 24) + 128); }
    return cn_classify_count(x, kind);
}

// per-wave scratch of cn_build (LDS)
struct CanonScratch {
    uint16_t parent[2 * CN_SYMS];        // tree: leaves 0..n-1 in sorted order, branches n..2n-2
    uint16_t symOf[CN_HIST];             // symbol of sorted leaf i
    uint32_t cntOf[CN_HIST];             // count of sorted leaf i
    uint8_t len[CN_HIST];                // code length per text symbol
    uint8_t tokCode[CN_HIST], tokRun[CN_HIST];
    uint16_t runStart[CN_HIST + 1];
    uint32_t metaCnt[CN_META + 4];
    uint8_t metaLen[CN_META + 4];
    uint32_t metaTab[CN_META + 4];       // (len << 16) | bit-reversed code
    uint8_t mtokCode[CN_META + 4], mtokRun[CN_META + 4];
    uint32_t blc[16];
};

// Scratch of the package-merge (codes that would be longer than CN_MAXLEN bits: rare): one level's items and the next level's
// package counts.  ONE per workgroup, shared by the waves that build the code tables side by side and taken under a lock
// (cn_pm_acquire): a scratch per tree cost the canonical encoder 8.8 KB of LDS, i.e. its fifth workgroup per CU.
struct CanonPM {
    uint32_t pmM[2 * CN_SYMS];           // bit 31 = package, low bits = count
    uint32_t pmPC[CN_SYMS];
    uint32_t pmMask[CN_MAXLEN][(2 * CN_SYMS + 31) / 32];
    uint32_t pmB[CN_MAXLEN];
};

// the lock word lives outside the scratch unions (zeroed with the per-tile state); one wave at a time holds it
__device__ __forceinline__ void cn_pm_acquire(uint32_t *lock, int lane)
{
    if (lane == 0) {
        while (atomicCAS(lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
}
__device__ __forceinline__ void cn_pm_release(uint32_t *lock, int lane)
{
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) atomicExch(lock, 0u);
}

// Tree by data-parallel rounds (see wave_huff_rounds in gvrs_encode.hip for why this reproduces the
// reference's linked-list merge).  K: live nodes in list order, key = count << 10 | tie, NREG*64 slots,
// dead = CN_DEAD; tie: leaf = 512 + sorted index, branch k = 510 - k.  Needs total count < 2^22 - 1.
template <int NREG>
__device__ __forceinline__ void cn_rounds(uint16_t *parent, uint32_t (&K)[8], int n, int lane)
{
    uint32_t L = (uint32_t)n, kbase = 0;
    const uint32_t un = (uint32_t)n;
    while (L > 1) {
        const uint32_t s0 = ((uint32_t)__builtin_amdgcn_readlane((int)K[0], 0) >> 10) +
                            ((uint32_t)__builtin_amdgcn_readlane((int)K[0], 1) >> 10);
        uint32_t t = 0;
#pragma unroll
        for (int r = 0; r < NREG; r++) t += (uint32_t)__popcll(__ballot((K[r] >> 10) < s0));
        const uint32_t P = t >> 1;
#pragma unroll
        for (int r = 0; r < NREG; r++) {
            if ((uint32_t)(r * 64) < 2u * P) {                      // wave-uniform
                const uint32_t e = (uint32_t)(r * 64 + lane);
                const uint32_t mine = K[r];
                const uint32_t other = gf_lane_xor(mine, 1);
                const uint32_t tieM = mine & 1023u;
                const uint32_t idM = tieM >= 512u ? tieM - 512u : un + (510u - tieM);
                const uint32_t k = kbase + (e >> 1);
                const bool right = e & 1u;
                if (e < 2u * P) {
                    parent[idM] = (uint16_t)((un + k) | (right ? 0x8000u : 0u));
                    K[r] = right ? CN_DEAD : ((((mine >> 10) + (other >> 10)) << 10) | (510u - k));
                }
            }
        }
        kbase += P;
        const uint32_t Lold = L;
        L -= P;
        if (NREG > 4 && Lold > 256) wave_bitonic_sort<8>(K, lane);
        else if (NREG > 2 && Lold > 128) wave_bitonic_sort<4>(K, lane);
        else if (NREG > 1 && Lold > 64) wave_bitonic_sort<2>(K, lane);
        else wave_bitonic_sort<1>(K, lane);
    }
    if (lane == 0 && n >= 1) parent[2 * n - 2] = 0xFFFF;            // root
}

// PackageMerge.merge(15, sortNodes): leaves already in (count asc, index asc) order in S.cntOf[0..nb).
// Level d = merge of the base items with the packages (sums of consecutive pairs) of level d-1, base
// items first on equal counts (:131-143); phase 2 (:151-165) walks the levels from the deepest one,
// looking at the first n items: a base item among them gets one more bit, the packages among them decide
// n for the level above.  Base items keep their order in every level, so "the base items among the first
// n" is a prefix [0, b_d) and the length of leaf i is the number of levels with i < b_d.
__device__ __forceinline__ void cn_package_merge(CanonScratch &S, CanonPM &M, uint32_t *pmLock, int nb, uint8_t *lenSorted, int lane)
{
    cn_pm_acquire(pmLock, lane);
    const uint32_t unb = (uint32_t)nb;
    uint32_t nPair = 0;                                  // packages feeding the current level
    for (int d = 0; d < CN_MAXLEN; d++) {
        const uint32_t len = unb + nPair;
        const uint32_t words = (len + 31u) >> 5;
        for (uint32_t w = (uint32_t)lane; w < words; w += 64) M.pmMask[d][w] = 0;
        __builtin_amdgcn_wave_barrier();
        // base item i goes behind the packages with a smaller count; package j behind the base items with count <= its own
        for (uint32_t i = (uint32_t)lane; i < unb; i += 64) {
            const uint32_t c = S.cntOf[i];
            uint32_t lo = 0, hi = nPair;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (M.pmPC[mid] < c) lo = mid + 1; else hi = mid; }
            M.pmM[i + lo] = c;
        }
        for (uint32_t j = (uint32_t)lane; j < nPair; j += 64) {
            const uint32_t c = M.pmPC[j];
            uint32_t lo = 0, hi = unb;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (S.cntOf[mid] <= c) lo = mid + 1; else hi = mid; }
            M.pmM[j + lo] = c | 0x80000000u;
            atomicOr(&M.pmMask[d][(j + lo) >> 5], 1u << ((j + lo) & 31u));
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t nNext = len >> 1;
        uint32_t pc[(CN_SYMS + 63) / 64];
#pragma unroll
        for (int q = 0; q < (CN_SYMS + 63) / 64; q++) {
            const uint32_t j = (uint32_t)(q * 64 + lane);
            pc[q] = j < nNext ? (M.pmM[2 * j] & 0x7fffffffu) + (M.pmM[2 * j + 1] & 0x7fffffffu) : 0u;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < (CN_SYMS + 63) / 64; q++) {
            const uint32_t j = (uint32_t)(q * 64 + lane);
            if (j < nNext) M.pmPC[j] = pc[q];
        }
        nPair = nNext;
        __builtin_amdgcn_wave_barrier();
    }
    // phase 2, wave-uniform
    uint32_t n = 2u * unb - 2u;
    for (int d = CN_MAXLEN - 1; d >= 0; d--) {
        uint32_t merged = 0;
        for (uint32_t w0 = 0; w0 < (n + 31u) >> 5; w0 += 64) {
            const uint32_t w = w0 + (uint32_t)lane;
            uint32_t m = (w << 5) < n ? M.pmMask[d][w] : 0u;
            if ((w << 5) < n && n - (w << 5) < 32u) m &= (1u << (n - (w << 5))) - 1u;
            uint32_t pcnt = (uint32_t)__popc(m);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) pcnt += gf_lane_xor(pcnt, o);
            merged += pcnt;
        }
        if (lane == 0) M.pmB[d] = n - merged;
        n = 2u * merged;
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = (uint32_t)lane; i < unb; i += 64) {
        uint32_t bits = 0;
        for (int d = 0; d < CN_MAXLEN; d++) bits += i < M.pmB[d] ? 1u : 0u;
        lenSorted[i] = (uint8_t)bits;
    }
    __builtin_amdgcn_wave_barrier();
    cn_pm_release(pmLock, lane);
}

// Code lengths of an alphabet of nSym (<= 64*NREG) symbols from counts cnt[] (LDS) into lenOut[sym].
// Returns the number of used symbols; *maxLenOut = longest code.  At least two symbols must be used
// (the callers guarantee it: the end-of-text symbol always counts 1).
template <int NREG>
__device__ __forceinline__ int cn_code_lengths(CanonScratch &S, CanonPM &M, uint32_t *pmLock, const uint32_t *cnt, int nSym,
                                               uint8_t *lenOut, int lane, uint32_t *maxLenOut)
{
    uint32_t K[8];
    int n = 0;
#ifndef GF_CN_NO_COMPACT
    if constexpr (NREG > 1) {
        // The symbols in use first, side by side (round 5): a terrain tile uses 60 to 130 of the 260, and the sorting network over the
        // 512 slots that hold them all is 45 stages of eight registers -- a fifth of cn_build's instructions -- where 128 slots take
        // 28 stages of two.  (The keys carry their symbol, so where a key stands before the sort does not matter.)
        uint32_t *tmp = S.cntOf;                            // (filled from the sorted keys below)
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < NREG; r++) {
            const int e = r * 64 + lane;
            if (r * 64 < nSym) {                            // (wave-uniform)
                const uint32_t c = e < nSym ? cnt[e] : 0u;
                if (e < nSym) lenOut[e] = 0;
                const unsigned long long m = __ballot(c != 0);
                // TreeBuilder.java:124-128: count ascending, symbol DESCENDING
                if (c) tmp[n + __popcll(m & lt)] = (c << 9) | (uint32_t)(511 - e);
                n += __popcll(m);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; r++) K[r] = (r < NREG && r * 64 + lane < n) ? tmp[r * 64 + lane] : CN_DEAD;
        __builtin_amdgcn_wave_barrier();
        if (n <= 64) wave_bitonic_sort<1>(K, lane);
        else if (n <= 128) wave_bitonic_sort<2>(K, lane);
        else if (NREG <= 4 || n <= 256) wave_bitonic_sort<(NREG < 4 ? NREG : 4)>(K, lane);
        else wave_bitonic_sort<NREG>(K, lane);
    } else
#endif
    {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            K[r] = CN_DEAD;
            if (r < NREG) {
                const int e = r * 64 + lane;
                const uint32_t c = e < nSym ? cnt[e] : 0u;
                if (e < nSym) lenOut[e] = 0;
                // TreeBuilder.java:124-128: count ascending, symbol DESCENDING
                K[r] = c ? ((c << 9) | (uint32_t)(511 - e)) : CN_DEAD;
                n += __popcll(__ballot(c != 0));
            }
        }
        wave_bitonic_sort<NREG>(K, lane);
    }
#pragma unroll
    for (int r = 0; r < NREG; r++) {
        const int e = r * 64 + lane;
        if (e < n) {
            S.symOf[e] = (uint16_t)(511u - (K[r] & 511u));
            S.cntOf[e] = K[r] >> 9;
            K[r] = ((K[r] >> 9) << 10) | (uint32_t)(512 + e);
        } else {
            K[r] = CN_DEAD;
        }
    }
    __builtin_amdgcn_wave_barrier();
    cn_rounds<NREG>(S.parent, K, n, lane);
    __builtin_amdgcn_wave_barrier();
    // TreeBuilder.establishCodeLengths :193-274: depth of every leaf
    uint32_t depth[8], maxLen = 0;
#pragma unroll
    for (int r = 0; r < NREG; r++) {
        const int e = r * 64 + lane;
        depth[r] = 0;
        if (e < n) {
            uint32_t node = (uint32_t)e, d = 0;
            for (int guard = 0; guard < 2 * CN_SYMS; guard++) {
                const uint32_t p = S.parent[node];
                if (p == 0xFFFFu) break;
                node = p & 0x7fffu;
                d++;
            }
            depth[r] = d;
            maxLen = max(maxLen, d);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) maxLen = max(maxLen, gf_lane_xor(maxLen, o));
#ifdef GF_CN_FORCE_PM                                                 // stress build of the shared package-merge scratch (tools/)
    if (n >= 2) {
#else
    if (maxLen > (uint32_t)CN_MAXLEN) {                              // TreeBuilder.java:173-178
#endif
        uint8_t *lenSorted = reinterpret_cast<uint8_t *>(S.parent);  // the tree is no longer needed
        cn_package_merge(S, M, pmLock, n, lenSorted, lane);
        maxLen = 0;
#pragma unroll
        for (int r = 0; r < NREG; r++) {
            const int e = r * 64 + lane;
            if (e < n) { depth[r] = lenSorted[e]; maxLen = max(maxLen, depth[r]); }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) maxLen = max(maxLen, gf_lane_xor(maxLen, o));
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < NREG; r++) {
        const int e = r * 64 + lane;
        if (e < n) lenOut[S.symOf[e]] = (uint8_t)depth[r];
    }
    __builtin_amdgcn_wave_barrier();
    *maxLenOut = maxLen;
    return n;
}

// Canonical codes from lengths (TreeBuilder.populateCanonicalCodes :283-300, HuffmanCodeBits.java:47-64):
// symbols ordered by (length, symbol), consecutive code values, shifted left when the length grows.
// tab[sym] = (len << 16) | code bit-reversed (the stream takes the code most significant bit first).
template <int NREG>
__device__ __forceinline__ void cn_assign_codes(CanonScratch &S, const uint8_t *len, int nSym, uint32_t *tab, int lane)
{
    uint32_t L[8], rank[8];
#pragma unroll
    for (int r = 0; r < NREG; r++) {
        const int e = r * 64 + lane;
        L[r] = e < nSym ? len[e] : 0u;
        rank[r] = 0;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t l = 1; l <= (uint32_t)CN_MAXLEN; l++) {
        uint32_t running = 0;
#pragma unroll
        for (int r = 0; r < NREG; r++) {
            const unsigned long long m = __ballot(L[r] == l);
            if (L[r] == l) rank[r] = running + (uint32_t)__popcll(m & lt);
            running += (uint32_t)__popcll(m);
        }
        if (lane == 0) S.blc[l] = running;
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t first[16];
    uint32_t code = 0;
    first[0] = 0;
#pragma unroll
    for (int l = 1; l <= CN_MAXLEN; l++) {
        first[l] = code;
        code = (code + S.blc[l]) << 1;
    }
#pragma unroll
    for (int r = 0; r < NREG; r++) {
        const int e = r * 64 + lane;
        if (e < nSym) {
            uint32_t f = 0;
#pragma unroll
            for (int l = 1; l <= CN_MAXLEN; l++) f = L[r] == (uint32_t)l ? first[l] : f;
            const uint32_t c = f + rank[r];
            tab[e] = L[r] ? ((L[r] << 16) | (__brev(c) >> (32u - L[r]))) : 0u;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// LengthEncoder.encodeLengths :86-166 over len[0..n): maximal runs, one lane per run.
//   zero run of m:      floor(m/138) x ZERO7(127), then r = m % 138: r >= 11 ZERO7(r-11); 3..10 ZERO3(r-3); 2 -> 0,0; 1 -> 0
//   non-zero run of m:  the value, then for the other r = m-1: floor(r/6) x PREV(3), then rr = r % 6:
//                       rr >= 3 PREV(rr-3); 2 -> v,v; 1 -> v
// (the greedy scan of the reference, solved per run).  Returns the number of tokens.
__device__ __forceinline__ int cn_rle(CanonScratch &S, const uint8_t *len, int n, uint8_t *tokCode, uint8_t *tokRun, int lane)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nRuns = 0;
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        const bool st = e < n && (e == 0 || len[e] != len[e - 1]);
        const unsigned long long m = __ballot(st);
        if (st) S.runStart[nRuns + __popcll(m & lt)] = (uint16_t)e;
        nRuns += __popcll(m);
    }
    if (lane == 0) S.runStart[nRuns] = (uint16_t)n;
    __builtin_amdgcn_wave_barrier();
    int base = 0;
    for (int k0 = 0; k0 < nRuns; k0 += 64) {
        const int k = k0 + lane;
        uint32_t v = 0, m = 0, cnt = 0;
        if (k < nRuns) {
            const uint32_t s = S.runStart[k];
            m = S.runStart[k + 1] - s;
            v = len[s];
            if (v == 0) {
                const uint32_t r = m % 138u;
                cnt = m / 138u + (r >= 3u ? 1u : r);
            } else {
                const uint32_t r = m - 1u, rr = r % 6u;
                cnt = 1u + r / 6u + (rr >= 3u ? 1u : rr);
            }
        }
        const uint32_t incl = wave_incl_scan(cnt, lane);
        uint32_t o = (uint32_t)base + incl - cnt;
        if (k < nRuns) {
            if (v == 0) {
                for (uint32_t q = 0; q < m / 138u; q++) { tokCode[o] = 18; tokRun[o] = 127; o++; }
                const uint32_t r = m % 138u;
                if (r >= 11u) { tokCode[o] = 18; tokRun[o] = (uint8_t)(r - 11u); }
                else if (r >= 3u) { tokCode[o] = 17; tokRun[o] = (uint8_t)(r - 3u); }
                else for (uint32_t q = 0; q < r; q++) { tokCode[o] = 0; tokRun[o] = 0; o++; }
            } else {
                tokCode[o] = (uint8_t)v; tokRun[o] = 0; o++;
                const uint32_t r = m - 1u, rr = r % 6u;
                for (uint32_t q = 0; q < r / 6u; q++) { tokCode[o] = 16; tokRun[o] = 3; o++; }
                if (rr >= 3u) { tokCode[o] = 16; tokRun[o] = (uint8_t)(rr - 3u); }
                else for (uint32_t q = 0; q < rr; q++) { tokCode[o] = (uint8_t)v; tokRun[o] = 0; o++; }
            }
        }
        base += (int)__builtin_amdgcn_readlane((int)incl, 63);
    }
    __builtin_amdgcn_wave_barrier();
    return base;
}

__device__ __forceinline__ uint32_t cn_run_bits(uint32_t code) { return code == 16u ? 2u : code == 17u ? 3u : code == 18u ? 7u : 0u; }

// ORs `nbits` (<= 32) bits of v into the LDS image at bit position pos
__device__ __forceinline__ void cn_img_or(uint32_t *img, uint32_t pos, uint32_t v, uint32_t nbits)
{
    if (nbits == 0) return;
    const uint64_t x = (uint64_t)v << (pos & 31u);
    atomicOr(&img[pos >> 5], (uint32_t)x);
    if ((pos & 31u) + nbits > 32u) atomicOr(&img[(pos >> 5) + 1], (uint32_t)(x >> 32));
}

struct CanonBuilt {
    uint32_t imgBits;        // bits of the serialised code tables (image starts at bit 0 of img)
    uint32_t maxLen;         // longest text code
    unsigned long long textBits;   // text + end-of-text symbol
};

// Everything CanonicalHuffman.encode :191-200 + buildCodeLengthTree :285-343 decide, for one stream, by
// one wave.  hist[CN_HIST]: symbol counts INCLUDING the end-of-text count of 1; nGap: values hit by the
// -8333608 quirk.  img must be zeroed (CN_IMG_WORDS).  tab[CN_SYMS] receives the code table.
__device__ __forceinline__ CanonBuilt cn_build(CanonScratch &S, CanonPM &M, uint32_t *pmLock, const uint32_t *hist, uint32_t nGap,
                                               uint32_t *tab, uint32_t *img, int lane)
{
    CanonBuilt B;
    uint32_t maxLen, metaMax;
    cn_code_lengths<8>(S, M, pmLock, hist, CN_SYMS, S.len, lane, &maxLen);
    cn_assign_codes<8>(S, S.len, CN_SYMS, tab, lane);
    const int nTok = cn_rle(S, S.len, CN_SYMS, S.tokCode, S.tokRun, lane);
    // meta alphabet: token codes + end-of-text (count 1, never written)  :289-297
    if (lane < CN_META) S.metaCnt[lane] = lane == CN_META - 1 ? 1u : 0u;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nTok; i += 64) atomicAdd(&S.metaCnt[S.tokCode[i]], 1u);
    __builtin_amdgcn_wave_barrier();
    cn_code_lengths<1>(S, M, pmLock, S.metaCnt, CN_META, S.metaLen, lane, &metaMax);
    cn_assign_codes<1>(S, S.metaLen, CN_META, S.metaTab, lane);
    const int nMtok = cn_rle(S, S.metaLen, CN_META, S.mtokCode, S.mtokRun, lane);
    // image: reserved bit 0 (:306), raw 5-bit meta tokens (LengthEncoder.java:169-195), coded text-length tokens (:322-342)
    uint32_t pos = 1;
    {
        const uint32_t c = lane < nMtok ? S.mtokCode[lane] : 0u;
        const uint32_t nb = lane < nMtok ? 5u + cn_run_bits(c) : 0u;
        const uint32_t incl = wave_incl_scan(nb, lane);
        if (lane < nMtok) cn_img_or(img, pos + incl - nb, c | ((uint32_t)S.mtokRun[lane] << 5), nb);
        pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    for (int i0 = 0; i0 < nTok; i0 += 64) {
        const int i = i0 + lane;
        uint32_t nb = 0, v = 0;
        if (i < nTok) {
            const uint32_t c = S.tokCode[i];
            const uint32_t e = S.metaTab[c];
            const uint32_t cl = e >> 16;
            v = (e & 0xffffu) | ((uint32_t)S.tokRun[i] << cl);
            nb = cl + cn_run_bits(c);
        }
        const uint32_t incl = wave_incl_scan(nb, lane);
        if (i < nTok) cn_img_or(img, pos + incl - nb, v, nb);
        pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    // exact size of the text: every counted symbol at its code length, raw escape bits, the quirk's extra escape
    unsigned long long bits = 0;
    for (int e = lane; e < CN_SYMS; e += 64) bits += (unsigned long long)hist[e] * S.len[e];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bits += __shfl_xor(bits, o, 64);
    bits += 2ull * hist[CN_ESC2] + 8ull * hist[CN_ESC1];
    bits += (unsigned long long)nGap * (unsigned long long)(S.len[CN_ESC1] + 8u + S.len[127]) -
            (unsigned long long)nGap * (unsigned long long)S.len[0];
    B.imgBits = pos;
    B.maxLen = maxLen;
    B.textBits = bits;
    __builtin_amdgcn_wave_barrier();
    return B;
}

// bits of one value in the text (emit-side classification)
__device__ __forceinline__ uint32_t cn_value_bits(const uint32_t *tab, uint32_t x)
{
    if (x + 128u < 256u) return tab[x + 128u] >> 16;
    uint32_t kind;
    const uint32_t target = cn_classify_emit(x, &kind);
    uint32_t b = tab[target] >> 16;
    if (kind >= 1u && kind <= 3u) b += kind * ((tab[CN_ESC2] >> 16) + 2u);
    else if (kind >= 4u && kind <= 6u) b += (kind - 3u) * ((tab[CN_ESC1] >> 16) + 8u);
    return b;
}

// appends one value to the sink (CanonicalHuffman.java:203-276)
__device__ __forceinline__ void cn_value_emit(BitSink &sink, const uint32_t *tab, uint32_t x)
{
    if (x + 128u < 256u) {
        const uint32_t e = tab[x + 128u];
        sink.put32(e & 0xffffu, e >> 16);
        return;
    }
    uint32_t kind;
    const uint32_t target = cn_classify_emit(x, &kind);
    const uint32_t e = tab[target];
    sink.put32(e & 0xffffu, e >> 16);
    if (kind >= 1u && kind <= 3u) {
        const uint32_t esc = tab[CN_ESC2], el = esc >> 16;
        for (int k = (int)kind - 1; k >= 0; k--) sink.put32((esc & 0xffffu) | (((x >> (2 * k)) & 3u) << el), el + 2u);
    } else if (kind >= 4u && kind <= 6u) {
        const uint32_t esc = tab[CN_ESC1], el = esc >> 16;
        for (int k = (int)kind - 4; k >= 0; k--) sink.put32((esc & 0xffffu) | (((x >> (8 * k)) & 0xffu) << el), el + 8u);
    }
}
