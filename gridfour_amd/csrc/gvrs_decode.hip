// gvrs_decode.hip -- CodecHuffman.decode for a batch of tiles, one workgroup per tile.
//
// Replaces (reference, core/src/main/java/org/gridfour/):
//   compress/CodecHuffman.java:133-169        header, Huffman decode, predictor decode
//   compress/HuffmanDecoder.java:65-187       tree parse, bit-serial symbol decode
//   compress/CodecM32.java:327-356            varint decode
//   compress/PredictorModel*.java decode      running sums
//   io/BitInputStore.java:112-210             LSB-first bit order
//
// The format has no synchronisation points, so both variable-length layers (Huffman codes
// over bits, M32 values over bytes) are parsed with the same self-synchronising scheme: the
// stream is cut into fixed-size subsequences, every thread parses its subsequence from a
// guessed start, then start positions are corrected from the predecessor's end position
// until nothing changes (codes resynchronise after a few symbols, so this takes 2-3 rounds;
// the worst case is still correct, just serial).  A prefix sum of the per-subsequence symbol
// counts then tells every thread where its output goes.
//
// Phases of a workgroup (256 threads) on one tile
//   0  header + tree: the pre-order serialisation is walked by k_huffman_parse_trees, a pre-pass kernel
//      with one lane per tile, which leaves per-leaf (code, length, symbol) records in HBM (the scalar
//      walk of one wave, parse_tree_wave, remains for trees at data-dependent positions: LSOP legacy
//      containers); the workgroup loads the records, marks the second-level prefixes with a block scan
//      and fills the 10-bit two-symbol decode LUT from the leaf table
//   1  Huffman text -> M32 bytes in LDS (global spill buffer for oversized tiles); every thread
//      keeps a 96-bit window of the text in registers and refills it one dword at a time
//   2  M32 bytes -> residuals: value starts are marked in a bitmap, ranked by a popcount prefix
//      sum, and decoded one byte position per thread so that stores to the output tile are
//      coalesced (consecutive bytes are consecutive cells for single-byte values)
//   3  predictor inverse in place (L2-resident): int32 wrap-around prefix sums -- column-0
//      chain, then row scans (Linear = double scan, Triangle = column sums then row scans),
//      with several independent loads in flight per thread

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "gvrs_kernels.h"
#include "huff_build.h"

namespace {

constexpr int GF_K_SKIP = 0x7fff0001;           // internal: a diagnostic phase limit ended the tile early
// The file is compiled three times: as is (256 threads per workgroup, two Huffman cursors per thread), and with
// -DGF_DEC_THREADS=512 / 1024 -DGF_DEC_VARIANT -DGF_DEC_MAXQ=512 / 1024 (one cursor per thread) for the tile sizes whose LDS
// footprint lets those builds put more waves on a CU (decodeBatchDev weighs them).  A variant object exports
// gf_launch_huffman_decode_t<threads> and gf_huffman_decode_lds_per_wg_t<threads> only.
#ifndef GF_DEC_THREADS
#define GF_DEC_THREADS 256
#endif
#ifdef GF_DEC_VARIANT
#define GF_DEC_CAT2(a, b) a##b
#define GF_DEC_CAT(a, b) GF_DEC_CAT2(a, b)
#define gf_launch_huffman_decode GF_DEC_CAT(gf_launch_huffman_decode_t, GF_DEC_THREADS)
#define gf_huffman_decode_lds_per_wg GF_DEC_CAT(gf_huffman_decode_lds_per_wg_t, GF_DEC_THREADS)
#define gf_launch_huffman_decode_canon GF_DEC_CAT(gf_launch_huffman_decode_canon_t, GF_DEC_THREADS)
#endif
constexpr int DEC_THREADS = GF_DEC_THREADS;
constexpr int DEC_WAVES = DEC_THREADS / 64;
constexpr int LUT_BITS = 10;                    // first-level window: most symbol PAIRS of terrain data fit 10 bits
#ifndef GF_DEC_MAXQ
#define GF_DEC_MAXQ (2 * GF_DEC_THREADS)
#endif
constexpr int MAXQ = GF_DEC_MAXQ;                // subsequences per chain: two per thread (advanced in lockstep) or one
// second argument of __launch_bounds__: waves per SIMD the register allocation must leave room for.  The 512-thread build is
// held to 64 VGPRs (it needs 49): four of its workgroups fill a CU's 32 wave slots where the LDS footprint allows four
// (tiles up to about 120x150) -- round 3: k_huffman_decode on the ETOPO1-shaped batch 1.22 -> 1.07 ms against the 256-thread
// build at four workgroups (16 waves) per CU; tools/occupancy_sweep.sh: the 256-thread build gains 14 % from a fifth and
// 8 % from a sixth workgroup per CU, which LDS does not allow it at this tile size.
#ifndef GF_DEC_WGS
#define GF_DEC_WGS (GF_DEC_THREADS == 256 ? 4 : 8)
#endif
// the general and the analysis instantiations (several times the code, 128 VGPRs): 16 waves per CU in either build
#define GF_DEC_WGS_GENERAL (GF_DEC_THREADS == 256 ? 4 : 4)
constexpr int HEAD_WORDS = 88;                 // 10 header + 1 + ceil(2559/8) tree bytes = 332 -> 83 words, + slack
constexpr int MAX_DEPTH = 63;                  // code length limit of the register tree parser

constexpr uint32_t GF_TREE_HAS_INTRODUCER = 0x100u;   // word 3 of a tree record, above the longest code length: some leaf is 0x7f / 0x81
constexpr uint32_t GF_TREE_HAS_NULL = 0x200u;         // ... some leaf is 0x80 (the null code)

constexpr int L2_MAX_BITS = 8;                 // second-level LUT: up to 8 more bits (codes of 11..18 bits)
constexpr int L2_ENTRIES = 2048;               // shared by all second-level tables: 2048 >> l2bits tables of
                                               // 2^l2bits entries, l2bits = min(8, longest code - 11) per tile

// a dword at any byte address (global memory)
struct __attribute__((packed, aligned(1))) PackedWord { uint32_t v; };

struct DecShared {
    uint32_t lut[1 << LUT_BITS];               // sym1 | sym2 << 8 | len1 << 16 | (len1 + len2) << 22 (== len1 << 22: one symbol);
                                               // bit 31 | sub-table: the code is longer than the window
    unsigned long long leafCode[256];          // per leaf, in pre-order: path bits root->leaf, first step in bit 0
    uint8_t leafLen[256];
    uint8_t leafSym[256];
    uint8_t shortLeaf[64];                     // leaves with code length <= 5 (filled cooperatively)
    uint32_t qs[MAXQ];                         // subsequence start
    uint32_t qe[MAXQ];                         // subsequence end (start of the next one)
    uint32_t qn[MAXQ];                         // symbols in the subsequence, later exclusive prefix
    uint8_t qdirty[MAXQ];
    uint32_t head[HEAD_WORDS];
    uint32_t waveSum[DEC_WAVES];
    uint32_t fusedTot[2 * 3 * DEC_WAVES];      // m32_to_tile: wave totals of the chunk scans, double-buffered
    uint32_t nRedo[2];                         // fast_sync_pass: subsequences to redo, by round parity
    uint32_t carry;
    uint32_t textStart;                        // bit offset of the Huffman text in the packing
    int32_t parseStatus;
    int32_t uniformSym;                        // >= 0: single-symbol encoding
    uint32_t nLeaves, nShort, nSub, l2bits, maxLen;
    uint32_t chainEnd;                         // position after the last needed symbol
    uint32_t chainTotal;
    uint32_t dense;                            // M32 stream too dense in multi-byte values for local start resolution
    // An INCOMPLETE tree (damaged input: the leaf count was reached while branch nodes still waited for children;
    // HuffmanDecoder.decodeTree :87-120 returns it as it is): skipLen = length of the LAST leaf's path (0: the tree is
    // complete), skipLo/Hi = that path.  Every 0 step on it is a branch whose right child was never read; the reference's
    // decode loop :179-185 finds 0 in such a child slot, lands on the root again and goes on without a symbol.
    uint32_t skipLen, skipLo, skipHi;
    uint32_t symKinds;                         // GF_TREE_HAS_* of the tree record (fast kernel only)
    uint32_t poolOverflow;                     // fast_sync_pass with a symbol pool: a subsequence had more symbols than its share holds
};

#include "gvrs_decode_common.h"

// A first-level entry with bit 31 set stands for codes longer than the window: bits 0-15 = the prefix's second-level table (its
// number among the tile's; may lie beyond the tables there is room for), bits 16-23 = the first leaf with that prefix.  A
// second-level entry is (length << 8) | symbol, or LUT2_SEARCH | first leaf with that (longer) prefix: the code is too long for the
// table and is looked up among the leaves from there (0xFFFF: never filled -- damaged input --, the whole leaf table).
constexpr uint32_t LUT_SUB_MASK = 0xffffu, LUT_FIRST_SHIFT = 16u, LUT2_SEARCH = 0x8000u;
// lookup entry of one symbol
__device__ __forceinline__ uint32_t lut_single(uint32_t sym, uint32_t len) { return sym | (len << 16) | (len << 22); }   // len <= 63

// ---- cursors: sequential readers of the two variable-length layers ----

// Huffman symbols over the packing's bit stream.  The packing is staged in LDS once per tile
// (coalesced copy); a cursor keeps three words of it in registers (w0..w2, bit offset sh into w0),
// fetches one more word per 32 bits consumed -- a word ahead of its use -- and forms the 32-bit
// decode window with one v_alignbit_b32.  Everything on the per-symbol path is 32-bit arithmetic
// and the only waits are on LDS.  (Reading the text straight from global memory stalls a whole
// wave on nearly every symbol: some lane is always refilling.)  prepare() / lookup() / advance()
// are separate so that a thread can overlap the LDS lookups of two independent subsequences.
// Packings that do not fit the LDS text buffer use the same code with a global pointer.
template <class TextPtr>
struct HuffCursorT {
    TextPtr base32;                // word that holds packing bit 0 (LDS copy or global)
    uint32_t nW;                   // words readable from base32
    uint32_t sh0;                  // position of packing bit 0 inside base32[0]
    const DecShared *S;
    const uint16_t *lut2;          // second-level table: (len << 8) | sym ; 0xFFFF = search the leaf table.  It lives in the
                                   // dynamic LDS area that phase 2 reuses for its bitmap (disjoint lifetimes)
    uint32_t pos;                  // packing-relative bit position of the next symbol
    uint32_t sh;                   // (pos + sh0) & 31 : offset of the next symbol inside w0
    uint32_t wi;                   // index of w0
    uint32_t w0, w1, w2;

    __device__ __forceinline__ uint32_t ld(uint32_t i) const { return i < nW ? base32[i] : 0u; }
    __device__ __forceinline__ void seek(uint32_t p)
    {
        pos = p;
        const uint32_t a = p + sh0;
        wi = a >> 5;
        sh = a & 31u;
        w0 = ld(wi);
        w1 = ld(wi + 1);
        w2 = ld(wi + 2);
    }
    // 32 bits of the text starting at the next symbol
    __device__ __forceinline__ uint32_t prepare() const { return __builtin_amdgcn_alignbit(w1, w0, sh); }
    __device__ __forceinline__ uint32_t lookup(uint32_t w) const { return S->lut[w & ((1u << LUT_BITS) - 1u)]; }
    __device__ __forceinline__ void advance(uint32_t len)
    {
        pos += len;
        sh += len;
        while (sh >= 32u) {                       // once per ~7 symbols per lane
            sh -= 32u;
            w0 = w1;
            w1 = w2;
            wi++;
            w2 = ld(wi + 2);
        }
    }
    // codes longer than the window: second level, then (longer than that, or out of tables) the leaf table.
    // Returns a single-symbol entry.
    __device__ __forceinline__ uint32_t resolve_long(uint32_t e, uint32_t w32v) const
    {
        const uint32_t l2 = S->l2bits, sub = e & LUT_SUB_MASK;
        uint32_t e16 = sub < ((uint32_t)L2_ENTRIES >> l2) ? lut2[(sub << l2) | ((w32v >> LUT_BITS) & ((1u << l2) - 1u))]
                                                           : (uint32_t)(LUT2_SEARCH | ((e >> LUT_FIRST_SHIFT) & 0xffu));
        if (e16 & LUT2_SEARCH) {
            // 64 bits of text for the leaf-table search (rare), from the first leaf that shares the code's known prefix
            const uint64_t lo = ((uint64_t)w1 << 32) | w0;
            uint64_t w = lo >> sh;
            if (sh) w |= (uint64_t)w2 << (64u - sh);
            const uint32_t n = S->nLeaves;
            const uint32_t first = e16 == 0xFFFFu ? 0u : e16 & 0xffu;
            e16 = 0xFFFFu;
            for (uint32_t i = first; i < n; i++) {
                const uint32_t cl = S->leafLen[i];
                const uint64_t mask = cl >= 64 ? ~0ull : ((1ull << cl) - 1ull);
                if (cl > LUT_BITS && (w & mask) == S->leafCode[i]) {
                    e16 = (cl << 8) | S->leafSym[i];
                    break;
                }
            }
            if (e16 == 0xFFFFu) e16 = (1u << 8);   // cannot happen for a complete tree; keep moving
        }
        return lut_single(e16 & 0xffu, e16 >> 8);
    }
    // decodes one symbol, advances
    __device__ __forceinline__ uint32_t next()
    {
        const uint32_t w = prepare();
        uint32_t e = lookup(w);
        if (e & 0x80000000u) e = resolve_long(e, w);
        advance((e >> 16) & 63u);
        return e & 0xffu;
    }
};

// M32 values over a byte buffer (CodecM32.java:327-356).  Two dwords of the buffer (the next 8
// bytes, 4-byte aligned loads) are kept in registers; length detection is 32-bit bit twiddling.
struct M32Cursor {
    const uint8_t *m;              // 4-byte aligned, readable up to the next multiple of 4 beyond n
    uint32_t n;                    // bytes available
    uint32_t pos;
    uint32_t base;                 // byte index of d0's first byte (multiple of 4)
    uint32_t d0, d1, d2;           // bytes base .. base+11

    __device__ __forceinline__ uint32_t ld4b(uint32_t i) const
    {
        if (i >= n) return 0u;
        uint32_t v = *reinterpret_cast<const uint32_t *>(m + i);
        if (i + 4 > n) v &= (1u << ((n - i) * 8u)) - 1u;          // bytes at or beyond n read as zero
        return v;
    }
    __device__ __forceinline__ void seek(uint32_t p)
    {
        pos = p;
        base = p & ~3u;
        d0 = ld4b(base);
        d1 = ld4b(base + 4);
        d2 = ld4b(base + 8);
    }
    // bytes pos..pos+3 in *lo, pos+4..pos+7 in *hi
    __device__ __forceinline__ void prepare(uint32_t *lo, uint32_t *hi)
    {
        while (pos >= base + 4) {
            d0 = d1;
            d1 = d2;
            base += 4;
            d2 = ld4b(base + 8);
        }
        const uint32_t bsh = (pos - base) * 8u;
        *lo = __builtin_amdgcn_alignbit(d1, d0, bsh);
        *hi = __builtin_amdgcn_alignbit(d2, d1, bsh);
    }
    // number of bytes of the value whose bytes 0..3 are lo and 4..7 are hi
    __device__ static __forceinline__ uint32_t length(uint32_t lo, uint32_t hi)
    {
        const uint32_t b0 = lo & 0xffu;
        if (b0 != 0x7fu && b0 != 0x81u) return 1u;
        // introducer: payload bytes 1..5 carry a continuation bit, at most 5 of them (:335 loop bound)
        const uint32_t stopLo = ~lo & 0x80808000u;               // payload bytes 1..3
        if (stopLo) return 1u + ((uint32_t)__builtin_ctz(stopLo) >> 3);      // bit 15 -> 2, 23 -> 3, 31 -> 4
        const uint32_t stopHi = ~hi & 0x00008080u;               // payload bytes 4..5
        return stopHi ? 5u + ((uint32_t)__builtin_ctz(stopHi) >> 3) : 6u;    // bit 7 -> 5, 15 -> 6
    }
    __device__ __forceinline__ uint32_t next()
    {
        uint32_t lo, hi;
        prepare(&lo, &hi);
        pos += length(lo, hi);
        return 0;
    }
};

// value encoded by bytes 0..3 (lo) and 4..7 (hi) (CodecM32.java:327-356); *len = its byte count
__device__ __forceinline__ uint32_t m32_value(uint32_t lo, uint32_t hi, uint32_t *len)
{
    const uint32_t b0 = lo & 0xffu;
    if (b0 != 0x7fu && b0 != 0x81u) {
        *len = 1;
        return b0 == 0x80u ? GF_NULL_CODE : (uint32_t)(int32_t)(int8_t)b0;
    }
    const uint32_t n = M32Cursor::length(lo, hi);
    *len = n;
    // payload bytes p1..p(n-1), big-endian 7-bit groups
    const uint32_t p1 = (lo >> 8) & 0x7fu, p2 = (lo >> 16) & 0x7fu, p3 = (lo >> 24) & 0x7fu;
    const uint32_t p4 = hi & 0x7fu, p5 = (hi >> 8) & 0x7fu;
    uint32_t delta, base, last;
    if (n == 2) { delta = p1; base = 127u; last = lo >> 8; }
    else if (n == 3) { delta = (p1 << 7) | p2; base = 255u; last = lo >> 16; }
    else if (n == 4) { delta = (p1 << 14) | (p2 << 7) | p3; base = 16639u; last = lo >> 24; }
    else if (n == 5) { delta = (p1 << 21) | (p2 << 14) | (p3 << 7) | p4; base = 2113791u; last = hi; }
    else { delta = (p1 << 28) | (p2 << 21) | (p3 << 14) | (p4 << 7) | p5; base = 270549247u; last = hi >> 8; }
    if (last & 0x80u) return delta;                // five continuation bytes: the reference returns delta
    return b0 == 0x81u ? (0u - delta - base) : (delta + base);
}

// advance two cursors of the same kind by one symbol each where active; the two table lookups
// are issued back to back so that their latencies overlap
template <class TextPtr>
__device__ __forceinline__ void step2(HuffCursorT<TextPtr> &c0, bool r0, uint32_t lim0, uint32_t room0, uint32_t *s0, uint32_t *n0,
                                      HuffCursorT<TextPtr> &c1, bool r1, uint32_t lim1, uint32_t room1, uint32_t *s1, uint32_t *n1)
{
    // lim: the second symbol of a pair is taken only if it starts before lim; room: symbols still wanted (>= 1)
    const uint32_t w0 = c0.prepare(), w1 = c1.prepare();
    uint32_t e0 = c0.lookup(w0), e1 = c1.lookup(w1);
    if ((e0 | e1) & 0x80000000u) {
        if (e0 & 0x80000000u) e0 = c0.resolve_long(e0, w0);
        if (e1 & 0x80000000u) e1 = c1.resolve_long(e1, w1);
    }
    const uint32_t a0 = (e0 >> 16) & 63u, t0 = e0 >> 22, a1 = (e1 >> 16) & 63u, t1 = e1 >> 22;
    const bool two0 = t0 != a0 && c0.pos + a0 < lim0 && room0 > 1u, two1 = t1 != a1 && c1.pos + a1 < lim1 && room1 > 1u;
    c0.advance(r0 ? (two0 ? t0 : a0) : 0u);
    c1.advance(r1 ? (two1 ? t1 : a1) : 0u);
    *s0 = e0 & 0xffffu;
    *s1 = e1 & 0xffffu;
    *n0 = two0 ? 2u : 1u;
    *n1 = two1 ? 2u : 1u;
}
__device__ __forceinline__ void step2(M32Cursor &c0, bool r0, uint32_t, uint32_t, uint32_t *s0, uint32_t *n0, M32Cursor &c1, bool r1,
                                      uint32_t, uint32_t, uint32_t *s1, uint32_t *n1)
{
    uint32_t lo0, hi0, lo1, hi1;
    c0.prepare(&lo0, &hi0);
    c1.prepare(&lo1, &hi1);
    c0.pos += r0 ? M32Cursor::length(lo0, hi0) : 0u;
    c1.pos += r1 ? M32Cursor::length(lo1, hi1) : 0u;
    *s0 = 0;
    *s1 = 0;
    *n0 = 1;
    *n1 = 1;
}

// Self-synchronising parse of [start, end) cut into Q <= 2*DEC_THREADS subsequences of `unit`.  On
// return qs[q] = true start of subsequence q, qn[q] = EXCLUSIVE prefix of the symbol counts,
// S.chainTotal = number of symbols that start before `end`.
//
// Subsequence q > 0 first parses a warm-up stretch of `warm` units that ends at its boundary: the
// codes resynchronise inside it with high probability, so the first position it reaches at or
// beyond the boundary is already the true start, and its end and count are final after ONE pass.
// Starts are then checked against the predecessor's end; only a mismatch (rare) marks a
// subsequence dirty for another, barrier-synchronised, round.  Every thread owns subsequences
// tid and tid + DEC_THREADS and advances them in lockstep (their lookups overlap).
template <int OWNER, class Cursor>                 // OWNER: see huffman_to_m32
__device__ void resolve_chain(DecShared &S, Cursor cur, uint32_t start, uint32_t end, uint32_t unit, uint32_t Q,
                              uint32_t warm, uint32_t *dbg = nullptr)
{
    const uint32_t tid = threadIdx.x;
    uint32_t rounds = 0;
    const uint32_t tBegin = (uint32_t)__builtin_amdgcn_s_memtime();
    const uint32_t q0 = tid, q1 = tid + DEC_THREADS;
    bool first = true;
    if (q0 < Q) S.qdirty[q0] = 1;
    if (q1 < Q) S.qdirty[q1] = 1;
    __syncthreads();
    for (;;) {
        {
            Cursor c0 = cur, c1 = cur;
            const bool d0 = q0 < Q && S.qdirty[q0], d1 = q1 < Q && S.qdirty[q1];
            const uint32_t b0 = start + q0 * unit, b1 = start + q1 * unit;       // boundaries
            const uint32_t lim0 = min(end, b0 + unit), lim1 = min(end, b1 + unit);
            uint32_t cnt0 = 0, cnt1 = 0;
            uint32_t s0 = first ? b0 : S.qs[q0], s1 = first ? b1 : S.qs[q1];
            if (first) {
                // warm-up: run from (boundary - warm) up to the boundary, keep only the landing position
                const uint32_t w0 = (q0 > 0 && b0 - start >= warm) ? b0 - warm : (q0 > 0 ? start : b0);
                const uint32_t w1 = b1 - start >= warm ? b1 - warm : start;
                c0.seek(d0 ? w0 : end);
                c1.seek(d1 ? w1 : end);
                for (;;) {
                    const uint32_t wl0 = min(b0, end), wl1 = min(b1, end);
                    const bool r0 = d0 && c0.pos < wl0, r1 = d1 && c1.pos < wl1;
                    if (!r0 && !r1) break;
                    uint32_t u0, u1, n0, n1;
                    step2(c0, r0, wl0, 2u, &u0, &n0, c1, r1, wl1, 2u, &u1, &n1);
                }
                s0 = c0.pos;
                s1 = c1.pos;
            } else {
                c0.seek(d0 ? s0 : end);
                c1.seek(d1 ? s1 : end);
            }
            for (;;) {
                const bool r0 = d0 && c0.pos < lim0, r1 = d1 && c1.pos < lim1;
                if (!r0 && !r1) break;
                uint32_t u0, u1, n0, n1;
                step2(c0, r0, lim0, 2u, &u0, &n0, c1, r1, lim1, 2u, &u1, &n1);
                cnt0 += r0 ? n0 : 0u;
                cnt1 += r1 ? n1 : 0u;
            }
            if (d0) { S.qs[q0] = s0; S.qe[q0] = c0.pos; S.qn[q0] = cnt0; S.qdirty[q0] = 0; }
            if (d1) { S.qs[q1] = s1; S.qe[q1] = c1.pos; S.qn[q1] = cnt1; S.qdirty[q1] = 0; }
            first = false;
        }
        __syncthreads();
        rounds++;
        if (dbg && tid == 0 && rounds == 1) dbg[1] = (uint32_t)__builtin_amdgcn_s_memtime() - tBegin;
        int changed = 0;
        if (q0 > 0 && q0 < Q) {
            const uint32_t ns = S.qe[q0 - 1];
            if (ns != S.qs[q0]) { S.qs[q0] = ns; S.qdirty[q0] = 1; changed = 1; }
        }
        if (q1 < Q) {
            const uint32_t ns = S.qe[q1 - 1];
            if (ns != S.qs[q1]) { S.qs[q1] = ns; S.qdirty[q1] = 1; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }
    // exclusive prefix sum of qn in subsequence order: thread t sums q = 2t, 2t+1
    const uint32_t qa = 2 * tid, qb = 2 * tid + 1;
    const uint32_t na = qa < Q ? S.qn[qa] : 0u, nb = qb < Q ? S.qn[qb] : 0u;
    uint32_t tot;
    const uint32_t run = block_excl_scan(na + nb, S.waveSum, &tot);
    if (qa < Q) S.qn[qa] = run;
    if (qb < Q) S.qn[qb] = run + na;
    if (tid == 0) S.chainTotal = tot;
    if (dbg && tid == 0) dbg[0] = rounds;
    __syncthreads();
}

// phase 1 body: Huffman text -> nM32 bytes at m32; returns GF_K_OK or the Java error it mirrors.  OWNER: one copy per
// kernel -- the function is not inlined, and a copy shared between kernels loses the address-space knowledge the compiler
// propagates from a single caller (flat accesses and pointer tests on the per-symbol path; measured 2.01 -> 2.10 ms)
template <int OWNER, class TextPtr>
__device__ int32_t huffman_to_m32(DecShared &S, HuffCursorT<TextPtr> cur, uint32_t textStart, uint32_t endBit,
                                  uint32_t nM32, uint8_t *m32, uint32_t *dbg, uint32_t warmBits)
{
    const uint32_t tid = threadIdx.x;
    int32_t status = GF_K_OK;
    const uint32_t textBits = endBit - textStart;
    uint32_t unit = (textBits + MAXQ - 1) / MAXQ;
    unit = max(128u, (unit + 31u) & ~31u);
    const uint32_t Q = max(1u, (textBits + unit - 1) / unit);
    resolve_chain<OWNER>(S, cur, textStart, endBit, unit, Q, warmBits, dbg);   // warm-up: 128 bits = about 25 symbols by default
    if (dbg && tid == 0) dbg[-7] = (uint32_t)__builtin_amdgcn_s_memtime();      // stamp 4
    if (S.chainTotal < nM32) status = GF_K_ERR_BOUNDS;                   // ran out of bits
    if (tid == 0) S.chainEnd = 0;
    __syncthreads();
    {
        const uint32_t q0 = tid, q1 = tid + DEC_THREADS;
        HuffCursorT<TextPtr> c0 = cur, c1 = cur;
        const bool d0 = q0 < Q, d1 = q1 < Q;
        uint32_t k0 = d0 ? S.qn[q0] : nM32, k1 = d1 ? S.qn[q1] : nM32;
        const uint32_t lim0 = min(endBit, textStart + (q0 + 1) * unit), lim1 = min(endBit, textStart + (q1 + 1) * unit);
        c0.seek(d0 ? S.qs[q0] : endBit);
        c1.seek(d1 ? S.qs[q1] : endBit);
        for (;;) {
            const bool r0 = d0 && c0.pos < lim0 && k0 < nM32, r1 = d1 && c1.pos < lim1 && k1 < nM32;
            if (!r0 && !r1) break;
            uint32_t u0, u1, n0, n1;
            step2(c0, r0, lim0, nM32 - k0, &u0, &n0, c1, r1, lim1, nM32 - k1, &u1, &n1);
            if (r0) {
                m32[k0] = (uint8_t)u0;
                if (n0 == 2u) m32[k0 + 1] = (uint8_t)(u0 >> 8);
                k0 += n0;
                if (k0 == nM32) S.chainEnd = c0.pos;
            }
            if (r1) {
                m32[k1] = (uint8_t)u1;
                if (n1 == 2u) m32[k1 + 1] = (uint8_t)(u1 >> 8);
                k1 += n1;
                if (k1 == nM32) S.chainEnd = c1.pos;
            }
        }
    }
    __syncthreads();
    if (status == GF_K_OK && S.chainEnd > endBit) status = GF_K_ERR_BOUNDS;      // last code ran past the end
    return status;
}


// ---------------------------------------------------------------------------------------------------------------
// The two Huffman passes of the fast kernel.  Same results as resolve_chain + the write loop of huffman_to_m32 (same
// subsequences, same start / end / count per subsequence, same bytes), written for instruction count: the generic
// cursor above refills its window in a divergent loop around a conditional global load and resolves codes longer than
// the window in nested divergent branches -- some lane of a wave is in each of them at nearly every step, so a wave paid
// about 240 instructions per step of its two cursors.  Here
//   * pass 1 (synchronisation) reads the text from an LDS copy -- the M32 buffer is still empty then -- padded with
//     zero words, so the window refill is an unconditional prefetch plus three selects;
//   * pass 2 (write) cannot keep that copy (it fills the M32 buffer), so each cursor loads eight words of its
//     subsequence into registers at once and shifts them down as bits are consumed; a block of steps ends when some
//     lane has used up its registers (all lanes consume bits at about the same rate);
//   * codes longer than the window take ONE branch per step for both cursors, with the second-level lookup straight-line
//     inside it (the leaf-table search behind that stays a loop: it runs for codes beyond 18 bits only).
// Requires code lengths <= 32 (longer codes need Fib(34) symbols in a tile; such tiles go to the general kernel).
struct FastHuff {
    const DecShared *S;
    const uint16_t *lut2;
    const unsigned long long *leafCodeG;           // the leaves' path bits in the tile's tree record (global memory): the LDS copy
                                                   // is given up after build_lut -- the count table of pass 1 lies over it
    uint32_t l2bits, nSub;
};

// entry of a code longer than the window: second level, then (rare) the leaf table; w = 32 text bits at the code,
// t0 | t1 << 32 | ... = the text from the code on (for the search)
__device__ __forceinline__ uint32_t fh_resolve(const FastHuff &H, uint32_t e, uint32_t w, uint32_t x0, uint32_t x1, uint32_t x2,
                                               uint32_t sh)
{
    const uint32_t sub = e & LUT_SUB_MASK;
    uint32_t e16 = LUT2_SEARCH | ((e >> LUT_FIRST_SHIFT) & 0xffu);
    if (sub < H.nSub) e16 = H.lut2[(sub << H.l2bits) | ((w >> LUT_BITS) & ((1u << H.l2bits) - 1u))];
    if (e16 & LUT2_SEARCH) {
        // the leaf table, from the first leaf that shares the code's known prefix (the leaves are in pre-order: those that share a
        // prefix lie side by side).  Round 4: the search used to start at leaf 0 and reads the path bits from the tile's record in
        // global memory -- a tile of flat ground with a few hundred rare symbols (codes of 19 bits and more) spent five million
        // cycles here, and the launch waited for it.
        const uint64_t lo = ((uint64_t)x1 << 32) | x0;
        uint64_t t = lo >> sh;
        if (sh) t |= (uint64_t)x2 << (64u - sh);
        const uint32_t n = H.S->nLeaves;
        const uint32_t first = e16 == 0xFFFFu ? 0u : e16 & 0xffu;
        e16 = 0xFFFFu;
        for (uint32_t i = first; i < n; i++) {
            const uint32_t cl = H.S->leafLen[i];
            const uint64_t mask = cl >= 64 ? ~0ull : ((1ull << cl) - 1ull);
            if (cl > LUT_BITS && (t & mask) == H.leafCodeG[i]) {
                e16 = (cl << 8) | H.S->leafSym[i];
                break;
            }
        }
        if (e16 == 0xFFFFu) e16 = (1u << 8);       // cannot happen for a complete tree; keep moving
    }
    return lut_single(e16 & 0xffu, e16 >> 8);
}

// Pass 1 only needs to know WHERE codes start and HOW MANY there are, not which symbols they are: its table gives, for the ten
// text bits at a code start, the bits and the number of ALL the codes that lie completely inside them (bits 0-3 / 4-7: up to
// ten one-bit codes) next to the length of the first one (bits 8-13); bit 15: the first code is longer than the window.  On
// terrain data a step consumes about eight bits whatever the code lengths are (the two-symbol table of pass 2 gives two
// codes at most: three bits per step where the residuals are small).  256 x 8 bytes: the table lies over DecShared::leafCode.
constexpr uint32_t CNT_LONG = 0x8000u;
__device__ __forceinline__ void build_count_table(DecShared &S, uint16_t *cnt16)
{
    for (uint32_t x = threadIdx.x; x < (1u << LUT_BITS); x += DEC_THREADS) {
        const uint32_t e = S.lut[x];
        uint32_t v = CNT_LONG;
        if (!(e & 0x80000000u)) {
            const uint32_t l1 = (e >> 16) & 63u;
            uint32_t pos = l1, ns = 1u;
            for (;;) {
                // the bits behind the codes found so far, zero-extended: an entry whose code fits the bits that are left was
                // decided by those bits alone (prefix code)
                const uint32_t e2 = S.lut[x >> pos];
                const uint32_t l = (e2 >> 16) & 63u;
                if ((e2 & 0x80000000u) || pos + l > (uint32_t)LUT_BITS) break;
                pos += l;
                ns++;
            }
            v = (l1 << 8) | (ns << 4) | pos;
        }
        cnt16[x] = (uint16_t)v;
    }
}

#ifndef GF_DEC_MIN_UNIT
#define GF_DEC_MIN_UNIT 128                         // bits of a subsequence of the fast Huffman pass, at least
#endif
constexpr uint32_t SHORT5_READY = 0xFFFFFFFFu;      // DecShared::nShort: the entries of the codes of up to five bits stand in S.qs[0..31]
constexpr uint32_t FAST_TEXT_PAD = 8;              // zero words behind the LDS copy of the text
#ifndef GF_DEC_EARLY_TXT
#define GF_DEC_EARLY_TXT 6
#endif
constexpr int EARLY_TXT = GF_DEC_EARLY_TXT;         // words per thread of the packing that k_huffman_decode<FAST> stages at the top of a tile

constexpr int NCUR = MAXQ / DEC_THREADS;           // cursors per thread: subsequences tid, tid + DEC_THREADS, ...

// Pass 1.  Round 1: every subsequence from its warm-up start.  A subsequence whose start then differs from its predecessor's end
// is decoded again from that end -- a handful per tile (a few codes in a thousand take longer than the warm-up to fall into
// step), yet a round costs every wave that owns one of them a full pass.  So from round 2 on the subsequences to redo are
// listed and dealt out to the lanes of as few waves as possible (wave 0 takes the first 64 * NCUR of them); the other
// waves go straight to the barrier.  Rounds repeat until no start moves.
//
// A step (round 3): the cursor is ONE register, the bit address of the next code inside the LDS copy of the text.  The two text
// words at it are read afresh (one ds_read2_b32), the window formed by v_alignbit, the count table answers with all the codes
// inside the window.  Codes are taken one at a time only within ten bits of a border (the warm-up's end, the subsequence's
// end: the first code at or behind a border belongs to the other side) and where the first code is longer than the window.
// With 32 waves on a CU the two dependent LDS round trips per step are covered by the other waves; what the kernel is short of
// is issue slots (PMC, profiles/r03_*: VALU 60 %, scalar unit 70 % busy), and the step is 28 instructions instead of 42.
//
// With a symbol POOL (round 4) the pass is the only decode of the text: behind its warm-up a subsequence is decoded through the
// two-symbol table and its symbols go, four to a word, to its share of the pool -- poolWords words per subsequence, word-major, in global
// memory (the tile's own output area, which nothing needs before the rows are written) --; when the counts are known the symbols
// are moved to their places in the M32 buffer (pool_to_m32) and the write pass, a second decode of every bit, is not run.
template <int OWNER>
__device__ void fast_sync_pass(DecShared &S, const FastHuff H, const uint16_t *cnt16, const uint32_t *txt, uint32_t sh0, uint32_t start,
                               uint32_t end, uint32_t unit, uint32_t Q, uint32_t warm, uint32_t *dbg, uint32_t *pool = nullptr,
                               const uint32_t poolWords = 0)
{
    const uint32_t tid = threadIdx.x;
    uint16_t *list = reinterpret_cast<uint16_t *>(S.qdirty);      // subsequences to redo (the flags themselves are not used here)
    constexpr uint32_t LIST_CAP = MAXQ / 2;                        // what does not fit waits for the next round
#ifdef GF_DIAG
    uint32_t rounds = 0;
#else
    (void)dbg;
#endif
    if (tid == 0) { S.nRedo[0] = 0; S.nRedo[1] = 0; S.poolOverflow = 0; }
    // one round: cursor i of this thread decodes subsequence qv[i] (>= Q: none)
    auto runRound = [&](const bool first, const uint32_t (&qv)[NCUR]) {
        bool d[NCUR];
        uint32_t bA[NCUR], limA[NCUR], a[NCUR], sPos[NCUR], cnt[NCUR];       // positions as bit addresses in txt (+ sh0)
#pragma unroll
        for (int i = 0; i < NCUR; i++) {
            const uint32_t q = qv[i];
            d[i] = q < Q;
            const uint32_t b = d[i] ? start + q * unit : end;                   // boundary
            bA[i] = b + sh0;
            limA[i] = min(end, b + unit) + sh0;
            // first round: from a warm-up stretch in front of the boundary (the first position reached at or beyond the
            // boundary is the start); later rounds: from the predecessor's end
            uint32_t p;
            if (first) p = (q > 0 && b - start >= warm) ? b - warm : (q > 0 ? start : b);
            else p = d[i] ? S.qe[q - 1u] : end;                                 // listed subsequences have q > 0
            a[i] = (d[i] ? p : end) + sh0;
            cnt[i] = 0;
        }
        // one step of every cursor towards its border: the codes of the window, or ONE code within ten bits of the border and
        // where the first code is longer than the window.  The loops are wave-uniform (no exec-mask bookkeeping: a cursor that
        // has arrived adds zero), and the warm-up is a loop of its own: it counts nothing and has no second border to watch.
        auto step = [&](const uint32_t (&border)[NCUR], auto counting) {
            uint32_t x[NCUR], e[NCUR], l1[NCUR], anyLong = 0;
#pragma unroll
            for (int i = 0; i < NCUR; i++) {
                const uint32_t wi = a[i] >> 5;
                x[i] = __builtin_amdgcn_alignbit(txt[wi + 1u], txt[wi], a[i]);  // the shift takes the low five bits of a
                e[i] = cnt16[x[i] & ((1u << LUT_BITS) - 1u)];
                anyLong |= e[i];
            }
#pragma unroll
            for (int i = 0; i < NCUR; i++) l1[i] = (e[i] >> 8) & 63u;
            if (__any((anyLong & CNT_LONG) != 0u)) {
#pragma unroll
                for (int i = 0; i < NCUR; i++)
                    if (e[i] & CNT_LONG) {
                        const uint32_t wi = a[i] >> 5;
                        const uint32_t e32 = fh_resolve(H, S.lut[x[i] & ((1u << LUT_BITS) - 1u)], x[i], txt[wi], txt[wi + 1u], txt[wi + 2u],
                                                        a[i] & 31u);
                        l1[i] = (e32 >> 16) & 63u;
                    }
            }
#pragma unroll
            for (int i = 0; i < NCUR; i++) {
                const bool live = a[i] < border[i];
                const bool single = (e[i] & CNT_LONG) || a[i] + (uint32_t)LUT_BITS > border[i];
                if constexpr (decltype(counting)::value) cnt[i] += live ? (single ? 1u : ((e[i] >> 4) & 15u)) : 0u;
                a[i] += live ? (single ? l1[i] : (e[i] & 15u)) : 0u;
            }
        };
        auto anyBefore = [&](const uint32_t (&border)[NCUR]) {
            bool any = false;
#pragma unroll
            for (int i = 0; i < NCUR; i++) any = any || a[i] < border[i];
            return __any(any) != 0;
        };
        while (anyBefore(bA)) step(bA, std::false_type{});
#pragma unroll
        for (int i = 0; i < NCUR; i++) sPos[i] = a[i] - sh0;                    // the first code at or beyond the boundary
        if (!pool) {
            while (anyBefore(limA)) step(limA, std::true_type{});
        } else {
            // the subsequence itself, symbols kept: one or two codes per step (the second one only if it STARTS before the
            // border: the same rule as above, so positions and counts are the same), bytes gathered in a register pair and
            // stored a word at a time
            // (the word being filled is stored after EVERY step, complete or not -- a later store of the same word only adds bytes --
            // so there is no "word full?" branch and no epilogue; word-major addresses: the lanes of a wave are at about the same
            // word, their stores fall into a line or two; a cursor that has filled its share, or overrun it, stores to the
            // dump row behind the pool: an idle step of a cursor with exactly 4 poolWords symbols must not touch its last word)
            uint32_t lo[NCUR];
#pragma unroll
            for (int i = 0; i < NCUR; i++) lo[i] = 0;
            const uint32_t dumpRow = poolWords;
            while (anyBefore(limA)) {
                uint32_t x[NCUR], e[NCUR], anyLong = 0;
#pragma unroll
                for (int i = 0; i < NCUR; i++) {
                    const uint32_t wi = a[i] >> 5;
                    x[i] = __builtin_amdgcn_alignbit(txt[wi + 1u], txt[wi], a[i]);
                    e[i] = S.lut[x[i] & ((1u << LUT_BITS) - 1u)];
                    anyLong |= e[i];
                }
                if (__any((anyLong & 0x80000000u) != 0u)) {
#pragma unroll
                    for (int i = 0; i < NCUR; i++)
                        if (e[i] & 0x80000000u) {
                            const uint32_t wi = a[i] >> 5;
                            e[i] = fh_resolve(H, e[i], x[i], txt[wi], txt[wi + 1u], txt[wi + 2u], a[i] & 31u);
                        }
                }
#pragma unroll
                for (int i = 0; i < NCUR; i++) {
                    const bool live = a[i] < limA[i];
                    const uint32_t a1 = (e[i] >> 16) & 63u, t2 = e[i] >> 22;
                    const bool two = t2 != a1 && a[i] + a1 < limA[i];
                    const uint32_t k = live ? (two ? 2u : 1u) : 0u;
                    const uint32_t syms = live ? (two ? e[i] & 0xffffu : e[i] & 0xffu) : 0u;
                    const uint32_t at = cnt[i] & 3u;
                    lo[i] |= syms << (8u * at);                                 // (a second byte behind byte 3 drops out: it opens the next word)
                    if (d[i]) pool[min(cnt[i] >> 2, dumpRow) * Q + qv[i]] = lo[i];
                    lo[i] = at + k >= 4u ? (at == 3u ? syms >> 8 : 0u) : lo[i];
                    cnt[i] += k;
                    a[i] += live ? (two ? t2 : a1) : 0u;
                }
            }
#pragma unroll
            for (int i = 0; i < NCUR; i++) {
                // (the word a second byte may have opened in the very last step)
                if (d[i] && (cnt[i] & 3u)) pool[min(cnt[i] >> 2, dumpRow) * Q + qv[i]] = lo[i];
                if (d[i] && cnt[i] > 4u * poolWords) S.poolOverflow = 1u;
            }
        }
#pragma unroll
        for (int i = 0; i < NCUR; i++)
            if (d[i]) { S.qs[qv[i]] = sPos[i]; S.qe[qv[i]] = a[i] - sh0; S.qn[qv[i]] = cnt[i]; }
    };
#ifdef GF_DIAG
    const uint32_t tBegin = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    {
        uint32_t qv[NCUR];
#pragma unroll
        for (int i = 0; i < NCUR; i++) qv[i] = tid + i * DEC_THREADS;
        runRound(true, qv);
    }
#ifdef GF_DIAG
    if (dbg && tid == 0) dbg[1] = (uint32_t)__builtin_amdgcn_s_memtime() - tBegin;      // round 1, wave 0
#endif
    for (uint32_t round = 1;; round++) {
        __syncthreads();                                                        // the round's starts and ends are in place
#ifdef GF_DIAG
        rounds++;
        if (dbg && tid == 0 && round == 1) dbg[3] = (uint32_t)__builtin_amdgcn_s_memtime() - tBegin;   // ... all waves
#endif
        if (tid == 0) S.nRedo[(round + 1u) & 1u] = 0;                           // the next round's counter (last read before this barrier)
#pragma unroll
        for (int i = 0; i < NCUR; i++) {
            const uint32_t q = tid + i * DEC_THREADS;
            if (q > 0 && q < Q && S.qe[q - 1] != S.qs[q]) {
                const uint32_t slot = atomicAdd(&S.nRedo[round & 1u], 1u);
                if (slot < LIST_CAP) list[slot] = (uint16_t)q;
            }
        }
        __syncthreads();
        const uint32_t nList = min(S.nRedo[round & 1u], LIST_CAP);
        if (nList == 0) break;
        uint32_t qv[NCUR];
#pragma unroll
        for (int i = 0; i < NCUR; i++) {
            const uint32_t slot = gf_wave_id() * (64u * NCUR) + (uint32_t)i * 64u + (tid & 63u);
            qv[i] = slot < nList ? list[slot] : 0xFFFFFFFFu;
        }
        runRound(false, qv);
    }
    // exclusive prefix sum of qn in subsequence order: thread t sums q = NCUR t .. NCUR t + NCUR - 1
    uint32_t n[NCUR], sum = 0;
#pragma unroll
    for (int i = 0; i < NCUR; i++) {
        const uint32_t q = NCUR * tid + i;
        n[i] = q < Q ? S.qn[q] : 0u;
        sum += n[i];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan(sum, S.waveSum, &tot);
#pragma unroll
    for (int i = 0; i < NCUR; i++) {
        const uint32_t q = NCUR * tid + i;
        if (q < Q) S.qn[q] = run;
        run += n[i];
    }
    if (tid == 0) S.chainTotal = tot;
#ifdef GF_DIAG
    if (dbg && tid == 0) { dbg[0] = rounds; dbg[4] = (uint32_t)__builtin_amdgcn_s_memtime() - tBegin; }   // the whole pass
#endif
    __syncthreads();
}

struct RCur {                                      // pass 2: eight words of the text in registers
    uint32_t pos, sh, r0, r1, r2, r3, r4, r5, r6, r7;
};
// words [wi, wi + 8) of the packing's text, wi = the word that holds bit p; beyond the readable words: the last one again
// (bits behind the packing never reach a result: a code that needs them ends behind the packing, which is an error)
__device__ __forceinline__ void rcur_load(RCur &c, const uint32_t *__restrict__ base32, uint32_t nW, uint32_t sh0, uint32_t p)
{
    const uint32_t a = p + sh0, wi = a >> 5;
    c.pos = p;
    c.sh = a & 31u;
    if (wi + 8u <= nW) {
        const GfU4 x = *reinterpret_cast<const GfU4 *>(base32 + wi), y = *reinterpret_cast<const GfU4 *>(base32 + wi + 4);
        c.r0 = x.x; c.r1 = x.y; c.r2 = x.z; c.r3 = x.w;
        c.r4 = y.x; c.r5 = y.y; c.r6 = y.z; c.r7 = y.w;
    } else {
        const uint32_t last = nW - 1u;
        c.r0 = base32[min(wi, last)];
        c.r1 = base32[min(wi + 1u, last)];
        c.r2 = base32[min(wi + 2u, last)];
        c.r3 = base32[min(wi + 3u, last)];
        c.r4 = base32[min(wi + 4u, last)];
        c.r5 = base32[min(wi + 5u, last)];
        c.r6 = base32[min(wi + 6u, last)];
        c.r7 = base32[min(wi + 7u, last)];
    }
}
__device__ __forceinline__ void rcur_advance(RCur &c, uint32_t len)                           // len <= 32
{
    c.pos += len;
    const uint32_t sh = c.sh + len;
    const bool ge = sh >= 32u;
    c.r0 = ge ? c.r1 : c.r0;
    c.r1 = ge ? c.r2 : c.r1;
    c.r2 = ge ? c.r3 : c.r2;
    c.r3 = ge ? c.r4 : c.r3;
    c.r4 = ge ? c.r5 : c.r4;
    c.r5 = ge ? c.r6 : c.r5;
    c.r6 = ge ? c.r7 : c.r6;
    c.sh = sh & 31u;
}
// bits a cursor may consume from one load: the step after them still finds its window, the second-level index and the
// 64 bits of a leaf search in r0..r2 without shifting in a word that was never loaded
constexpr uint32_t RCUR_BUDGET = 256u - 31u - 96u;

// The pool's symbols to their places: S.qn holds the exclusive prefix of the subsequences' counts, so subsequence q's symbols are bytes
// [qn[q], qn[q + 1]) of the M32 stream (what lies beyond nM32 is not wanted).  A thread reads its share of the pool eight words at
// a time; up to three head bytes bring it to a word boundary of the stream, then whole words go out (the pool's words re-aligned by
// v_alignbyte), then up to three tail bytes -- the words at either end are shared with the neighbouring subsequences.  What is not
// wanted goes to a spare word behind the stream (no branch per store).
__device__ __forceinline__ void pool_to_m32(DecShared &S, const uint32_t *pool, uint32_t poolWords, uint32_t Q, uint32_t nM32, uint8_t *m32,
                                            uint32_t spare)
{
    const uint32_t tid = threadIdx.x;
    uint32_t *m32w = reinterpret_cast<uint32_t *>(m32);
#pragma unroll
    for (int i = 0; i < NCUR; i++) {
        const uint32_t q = tid + (uint32_t)i * DEC_THREADS;
        uint32_t base = 0, n = 0;
        if (q < Q) {
            base = S.qn[q];
            n = (q + 1u < Q ? S.qn[q + 1u] : S.chainTotal) - base;
            n = base < nM32 ? min(n, nM32 - base) : 0u;
        }
        const uint32_t *src = pool + (q < Q ? q : 0u);               // word w of subsequence q: pool[w Q + q]
        const uint32_t h = min(n, (0u - base) & 3u);             // head bytes
        const uint32_t nb = (n - h) >> 2, r = (n - h) & 3u;       // whole words, tail bytes
        const uint32_t w0 = (base + h) >> 2;                      // the stream word the first whole word goes to
        uint32_t tailWord = 0, first = 0;
        for (uint32_t c = 0; c < poolWords; c += 8u) {
            if (!__any(4u * c < n)) break;
            uint32_t w[9];
#pragma unroll
            for (uint32_t j = 0; j < 9u; j++) w[j] = c + j < poolWords ? src[(c + j) * Q] : 0u;
            if (c == 0u) first = w[0];
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) {
                const uint32_t k = c + j;                         // the k-th whole word: pool bytes h + 4 k ..
                const uint32_t out = __builtin_amdgcn_alignbyte(w[j + 1u], w[j], h);
                m32w[k < nb ? w0 + k : spare >> 2] = out;
                tailWord = k == nb ? out : tailWord;
            }
        }
        // head: pool bytes 0 .. h - 1 to stream bytes base ..; tail: the first r bytes of tailWord behind the whole words
#pragma unroll
        for (uint32_t j = 0; j < 3u; j++) {
            m32[j < h ? base + j : spare] = (uint8_t)(first >> (8u * j));
            m32[j < r ? base + h + 4u * nb + j : spare] = (uint8_t)(tailWord >> (8u * j));
        }
    }
}

template <int OWNER>
__device__ int32_t huffman_to_m32_fast(DecShared &S, const uint32_t *__restrict__ base32, uint32_t nW, uint32_t sh0,
                                       uint32_t *txt, uint32_t pkWords, const uint16_t *lut2, const unsigned long long *leafCodeG,
                                       uint32_t textStart, uint32_t endBit, uint32_t nM32, uint8_t *m32, uint32_t *dbg,
                                       const uint32_t warmBits, const int diagLimit = 0, uint32_t *pool = nullptr, uint32_t poolBytes = 0,
                                       uint32_t spareByte = 0)
{
    const uint32_t tid = threadIdx.x;
    int32_t status = GF_K_OK;
    const uint32_t textBits = endBit - textStart;
    uint32_t unit = (textBits + MAXQ - 1) / MAXQ;
    unit = max((uint32_t)GF_DEC_MIN_UNIT, (unit + 31u) & ~31u);
    const uint32_t Q = max(1u, (textBits + unit - 1) / unit);
    FastHuff H;
    H.S = &S;
    H.lut2 = lut2;
    H.leafCodeG = leafCodeG;
    H.l2bits = S.l2bits;
    H.nSub = S.nSub;
    // the count table of pass 1, over the leaves' path bits (build_lut, which read them, ended with a barrier)
    uint16_t *cnt16 = reinterpret_cast<uint16_t *>(S.leafCode);
    static_assert(sizeof(S.leafCode) >= (sizeof(uint16_t) << LUT_BITS), "count table does not fit over leafCode");
    build_count_table(S, cnt16);
    // the packing into LDS (txt = the M32 buffer, not yet in use), zero words behind it
#ifdef GF_DIAG
    const uint32_t tStage = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    // (the first EARLY_TXT words per thread are there already: the kernel asked for them at the top of the tile)
    for (uint32_t i = tid + (uint32_t)EARLY_TXT * DEC_THREADS; i < pkWords + FAST_TEXT_PAD; i += DEC_THREADS) txt[i] = (i < pkWords && i < nW) ? base32[i] : 0u;
    __syncthreads();
#ifdef GF_DIAG
    if (dbg && tid == 0) dbg[2] = (uint32_t)__builtin_amdgcn_s_memtime() - tStage;
#endif
#ifdef GF_DIAG
    if (diagLimit == 7) return GF_K_SKIP;                                        // (count table built, text staged)
#endif
    // the symbol pool: the share of a subsequence is what the tile's output area gives each of the Q, at most 128 symbols (more
    // than a 160-bit subsequence of terrain holds three times over); below 64 the two passes below
    // (one row of the area is the dump row of fast_sync_pass)
    uint32_t poolWords = pool ? min(33u, (poolBytes / Q) >> 2) : 0u;
    poolWords = poolWords < 17u ? 0u : poolWords - 1u;
    fast_sync_pass<OWNER>(S, H, cnt16, txt, sh0, textStart, endBit, unit, Q, warmBits, dbg, poolWords ? pool : nullptr, poolWords);
#ifdef GF_DIAG
    if (diagLimit == 8) return GF_K_SKIP;                                        // (+ synchronisation pass)
    if (dbg && tid == 0) dbg[-7] = (uint32_t)__builtin_amdgcn_s_memtime();      // stamp 4
#else
    (void)dbg;
#endif
    if (S.chainTotal < nM32) status = GF_K_ERR_BOUNDS;                   // ran out of bits
#ifdef GF_DEC_POOL_FORCE_OVERFLOW                                        // (experiment builds: the fall-back behind a pool that overflowed)
    if (false) {
#else
    if (poolWords && !S.poolOverflow) {
#endif
        // (fast_sync_pass ended with a barrier: every reader of the LDS text is done, the pool is written)
        pool_to_m32(S, pool, poolWords, Q, nM32, m32, spareByte);
        // the last code the stream needs may run past the end of the packing: only the text's very last code can, and it is the
        // nM32-th exactly when the text holds no more than that
        if (status == GF_K_OK && S.chainTotal == nM32 && S.qe[Q - 1u] > endBit) status = GF_K_ERR_BOUNDS;
        __syncthreads();
        return status;
    }
    if (tid == 0) S.chainEnd = 0;
    __syncthreads();                                                     // every reader of the LDS text is done
    {
        bool d[NCUR];
        uint32_t k[NCUR], lim[NCUR];
        RCur c[NCUR];
#pragma unroll
        for (int i = 0; i < NCUR; i++) {
            const uint32_t q = tid + i * DEC_THREADS;
            d[i] = q < Q;
            k[i] = d[i] ? S.qn[q] : nM32;
            lim[i] = min(endBit, textStart + (q + 1) * unit);
            c[i].pos = d[i] ? S.qs[q] : endBit;
        }
        for (;;) {
            bool live[NCUR], anyLive = false;
#pragma unroll
            for (int i = 0; i < NCUR; i++) {
                live[i] = d[i] && c[i].pos < lim[i] && k[i] < nM32;
                anyLive = anyLive || live[i];
            }
            if (!__any(anyLive)) break;
            uint32_t blk[NCUR];
#pragma unroll
            for (int i = 0; i < NCUR; i++) {
                rcur_load(c[i], base32, nW, sh0, live[i] ? c[i].pos : 0u);
                if (!live[i]) c[i].pos = endBit;
                blk[i] = c[i].pos;
            }
            for (;;) {
                bool r[NCUR], any = false;
#pragma unroll
                for (int i = 0; i < NCUR; i++) {
                    r[i] = live[i] && c[i].pos < lim[i] && k[i] < nM32 && c[i].pos - blk[i] <= RCUR_BUDGET;
                    any = any || r[i];
                }
                if (!__any(any)) break;
                uint32_t x[NCUR], e[NCUR], anyLong = 0;
#pragma unroll
                for (int i = 0; i < NCUR; i++) {
                    x[i] = __builtin_amdgcn_alignbit(c[i].r1, c[i].r0, c[i].sh);
                    e[i] = S.lut[x[i] & ((1u << LUT_BITS) - 1u)];
                    anyLong |= e[i];
                }
                if (anyLong & 0x80000000u) {
#pragma unroll
                    for (int i = 0; i < NCUR; i++)
                        if (e[i] & 0x80000000u) e[i] = fh_resolve(H, e[i], x[i], c[i].r0, c[i].r1, c[i].r2, c[i].sh);
                }
#pragma unroll
                for (int i = 0; i < NCUR; i++) {
                    const uint32_t a1 = (e[i] >> 16) & 63u, t2 = e[i] >> 22;
                    const bool two = t2 != a1 && c[i].pos + a1 < lim[i] && nM32 - k[i] > 1u;
                    rcur_advance(c[i], r[i] ? (two ? t2 : a1) : 0u);
                    if (r[i]) {
                        m32[k[i]] = (uint8_t)e[i];
                        if (two) m32[k[i] + 1] = (uint8_t)(e[i] >> 8);
                        k[i] += two ? 2u : 1u;
                        if (k[i] == nM32) S.chainEnd = c[i].pos;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (status == GF_K_OK && S.chainEnd > endBit) status = GF_K_ERR_BOUNDS;      // last code ran past the end
    return status;
}


// lane `lane` of `old` becomes the wave-uniform `value` (this hipcc has no v_writelane builtin; a compare and a select
// do the same without inline assembly)
__device__ __forceinline__ int gf_writelane(int value, int lane, int old)
{
    return (int)(threadIdx.x & 63u) == lane ? value : old;
}

// HuffmanDecoder.decodeTree (HuffmanDecoder.java:65-161), run by ONE wave.  S.head holds the packing words around the
// serialised tree; the tree starts at bit relBit of S.head (= bit absBit of the packing, totalBits long).  Leaves go to
// S.leaf*, sub-table markers to S.lut, the position of the first code to S.textStart.
__device__ void parse_tree_wave(DecShared &S, uint32_t relBit, uint32_t absBit, uint32_t totalBits)
{
    // HuffmanDecoder.decodeTree (HuffmanDecoder.java:65-161) as a wave-uniform scalar loop: the
    // serialised tree (<= 83 dwords) and the node stack live in VGPRs and are read with
    // v_readlane, so a node costs a few dozen scalar instructions and no LDS round trip.
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t hw0 = S.head[lane];
    const uint32_t hw1 = lane < HEAD_WORDS - 64 ? S.head[64 + lane] : 0u;
    auto word = [&](uint32_t wi) -> uint32_t {
        if (wi >= HEAD_WORDS) return 0u;
        return (uint32_t)__builtin_amdgcn_readlane((int)(wi < 64 ? hw0 : hw1), (int)(wi & 63u));
    };
    // scalar bit buffer: `buf` holds the next `have` bits of the serialised tree, LSB first
    const uint32_t fw = relBit >> 5, fs = relBit & 31u;
    uint64_t buf = (((uint64_t)word(fw + 1u) << 32) | word(fw)) >> fs;
    uint32_t have = 64u - fs, wnext = fw + 2u, bp = absBit;
    auto refill = [&]() {
        if (have <= 32) {
            buf |= (uint64_t)word(wnext) << have;
            have += 32;
            wnext++;
        }
    };
    auto take = [&](uint32_t nb) -> uint32_t {          // nb <= 9, needs have >= nb
        const uint32_t v = (uint32_t)buf & ((1u << nb) - 1u);
        buf >>= nb;
        have -= nb;
        bp += nb;
        return v;
    };
    const bool writer = lane == 0;
    int32_t st = GF_K_OK;
    int32_t uniformSym = -1;
    const uint32_t nLeaves = take(8) + 1;
    const uint32_t rootBit = take(1);
    uint32_t nShort = 0, nSub = 0, maxLen = 1;
    uint32_t skipLen = 0;
    unsigned long long skipPath = 0;
    if (rootBit == 1) {
        uniformSym = (int32_t)take(8);
    } else {
        // Pre-order walk without building nodes.  `c` is the code of the node about to be read as an integer of
        // L bits, most significant bit = first step (0 left, 1 right).  One iteration per leaf: a run of z branch
        // records (zero bits) appends z zeros to the code; the leaf that follows takes the code as it stands; the
        // next node is the right sibling of the deepest ancestor still on its left side: strip the trailing ones,
        // turn the zero in front of them into a one.
        uint64_t c = 0;
        uint32_t L = 1;                                              // the root's left child
        uint32_t leaves = 0, records = 0;
        bool complete = false;
        // leaf records gather in registers, lane = leaf index mod 64 (v_writelane from the scalar loop), and go to LDS 64
        // at a time with every lane storing its own: a predicated single-lane LDS store per field and leaf was the
        // larger part of the loop
        int recLo = 0, recHi = 0, recLS = 0;
        auto flush = [&](uint32_t base, uint32_t count) {            // leaves base .. base + count - 1 are in the registers
            const bool mine = lane < count;
            const uint32_t len = (uint32_t)recLS >> 8;
            if (mine) {
                S.leafCode[base + lane] = ((unsigned long long)(uint32_t)recHi << 32) | (uint32_t)recLo;
                S.leafLen[base + lane] = (uint8_t)len;
                S.leafSym[base + lane] = (uint8_t)recLS;
            }
            const unsigned long long shortMask = __ballot(mine && len <= 5u);
            if (mine && len <= 5u) {
                const uint32_t slot = nShort + (uint32_t)__popcll(shortMask & ((1ull << lane) - 1ull));
                S.shortLeaf[slot & 63u] = (uint8_t)(base + lane);
            }
            nShort += (uint32_t)__popcll(shortMask);
        };
        while (leaves < nLeaves) {
            refill();
            if (records > 511) { st = GF_K_ERR_BOUNDS; break; }
            uint32_t z = buf ? (uint32_t)__builtin_ctzll(buf) : 64u;
            z = min(z, have);
            if (z) {
                // decodeTree's arrays: stack = new int[nLeaves + 1] (a branch of code length l is pushed at index l) and
                // nodeIndex = new int[6 nLeaves] (three ints per node, the root included): damage that asks for more ends in
                // ArrayIndexOutOfBoundsException there
                if (L - 1u + z > nLeaves || records + z > 2u * nLeaves - 1u) { st = GF_K_ERR_BOUNDS; break; }
                if (L - 1u + z > MAX_DEPTH) { st = GF_K_ERR_FORMAT; break; }   // see DESIGN.md (unsupported depth)
                if (L <= (uint32_t)LUT_BITS && L + z > (uint32_t)LUT_BITS) {
                    // a branch at depth LUT_BITS on this path: codes below it continue into a second-level table,
                    // indexed by the first LUT_BITS bits of the path (first step in bit 0)
                    const uint32_t prefix = (uint32_t)((c << z) >> (L + z - LUT_BITS));
                    if (writer) S.lut[__brev(prefix) >> (32 - LUT_BITS)] = 0x80000000u | (leaves << LUT_FIRST_SHIFT) | nSub;   // (leaves: the next leaf's index)
                    nSub++;
                }
                c <<= z;                                             // z <= 63 here
                L += z;
                records += z;
                buf >>= z;
                have -= z;
                bp += z;
                if (have == 0u || !(buf & 1ull)) continue;           // the run continues beyond the buffered bits
                refill();
            }
            if (records + 1u > 2u * nLeaves - 1u) { st = GF_K_ERR_BOUNDS; break; }      // nodeIndex is full (see above)
            const uint32_t rec = take(9);
            const uint32_t sym = rec >> 1;
            const uint32_t clen = L;
            records++;
            {
                const unsigned long long path = __brevll(c) >> (64u - L);   // path bits, first step in bit 0
                const int ln = (int)(leaves & 63u);
                recLo = gf_writelane((int)(uint32_t)path, ln, recLo);
                recHi = gf_writelane((int)(uint32_t)(path >> 32), ln, recHi);
                recLS = gf_writelane((int)((clen << 8) | sym), ln, recLS);
            }
            maxLen = max(maxLen, clen);
            leaves++;
            if ((leaves & 63u) == 0u) flush(leaves - 64u, 64u);
            const uint32_t t1 = ~c ? (uint32_t)__builtin_ctzll(~c) : 64u;   // trailing ones
            if (leaves == nLeaves) { complete = t1 >= L; break; }
            if (t1 >= L) { st = GF_K_ERR_BOUNDS; break; }            // tree complete but leaves missing
            c = (c >> t1) | 1ull;
            L -= t1;
        }
        if (leaves & 63u) flush(leaves & ~63u, leaves & 63u);
        // all leaves read.  An incomplete tree (some open branch still on its left child) is what the reference's decoder
        // walks with a fall-back to the root: huffman_serial_skips
        if (st == GF_K_OK && !complete && leaves == nLeaves) {
            skipLen = L;
            skipPath = __brevll(c) >> (64u - L);
        }
    }
    if (st == GF_K_OK && bp > totalBits) st = GF_K_ERR_BOUNDS;   // read past end of data
    if (writer) {
        S.skipLen = skipLen;
        S.skipLo = (uint32_t)skipPath;
        S.skipHi = (uint32_t)(skipPath >> 32);
        S.uniformSym = uniformSym;
        S.textStart = bp;
        S.parseStatus = st;
        S.nLeaves = nLeaves;
        S.nShort = nShort;
        const uint32_t l2 = maxLen > LUT_BITS ? min((uint32_t)L2_MAX_BITS, maxLen - LUT_BITS) : 1u;
        S.l2bits = l2;
        S.nSub = min(nSub, (uint32_t)L2_ENTRIES >> l2);
        S.maxLen = maxLen;
    }
}

// Lookup tables from the leaf table, by the whole workgroup; ends with a barrier.
template <class Lut2Ptr>
__device__ __forceinline__ void build_lut(DecShared &S, Lut2Ptr lut2)
{
    const uint32_t tid = threadIdx.x;
    // LUT from the leaf table: a leaf with a code of <= LUT_BITS bits owns 2^(LUT_BITS-len) first-level entries,
    // a longer one (up to LUT_BITS + l2bits) owns entries of its prefix's second-level table
    {
        const uint32_t nLeaves = S.nLeaves;
        const uint32_t nSub = S.nSub;
        const uint32_t l2 = S.l2bits;
        for (uint32_t x = tid; x < (nSub << l2); x += DEC_THREADS) lut2[x] = 0xFFFFu;
        if ((uint32_t)tid < nLeaves) {
            const uint32_t cl = S.leafLen[tid];
            if (cl > 5 && cl <= LUT_BITS) {
                const uint32_t e = lut_single(S.leafSym[tid], cl);
                for (uint32_t x = (uint32_t)S.leafCode[tid]; x < (1u << LUT_BITS); x += 1u << cl) S.lut[x] = e;
            }
        }
        if (S.nShort == SHORT5_READY) {                       // (leaf records from a pre-pass: the short codes' table is in S.qs)
            for (uint32_t x = tid; x < (1u << LUT_BITS); x += DEC_THREADS) {
                const uint32_t e = S.qs[x & 31u];
                if (e) S.lut[x] = e;
            }
        }
        const uint32_t nShort = S.nShort == SHORT5_READY ? 0u : min(S.nShort, 64u);
        for (uint32_t j = 0; j < nShort; j++) {             // few, large fills: all threads together
            const uint32_t i = S.shortLeaf[j];
            const uint32_t cl = S.leafLen[i];
            const uint32_t e = lut_single(S.leafSym[i], cl);
            for (uint32_t x = (uint32_t)S.leafCode[i] + ((uint32_t)tid << cl); x < (1u << LUT_BITS);
                 x += (uint32_t)DEC_THREADS << cl)
                S.lut[x] = e;
        }
        __syncthreads();                                     // lut2 cleared, first level complete
        // pair up: where the code behind an entry's symbol is short enough to lie inside the window too, the
        // entry yields both symbols (its first-symbol fields stay as they are, so in-place update is safe)
        for (uint32_t x = tid; x < (1u << LUT_BITS); x += DEC_THREADS) {
            const uint32_t e = S.lut[x];
            if (!(e & 0x80000000u)) {
                const uint32_t l1 = (e >> 16) & 63u;
                const uint32_t e2 = S.lut[x >> l1];                // the following bits, zero-extended
                const uint32_t l2b = (e2 >> 16) & 63u;
                if (!(e2 & 0x80000000u) && l1 + l2b <= (uint32_t)LUT_BITS)
                    S.lut[x] = (e & 0x003F00FFu) | ((e2 & 0xffu) << 8) | ((l1 + l2b) << 22);
            }
        }
        if ((uint32_t)tid < nLeaves) {
            const uint32_t cl = S.leafLen[tid];
            if (cl > LUT_BITS && cl <= LUT_BITS + l2) {
                const uint64_t code = S.leafCode[tid];
                const uint32_t subIdx = S.lut[(uint32_t)code & ((1u << LUT_BITS) - 1u)] & LUT_SUB_MASK;
                if (subIdx < nSub) {
                    uint16_t *sub = &lut2[subIdx << l2];
                    const uint16_t e = (uint16_t)((cl << 8) | S.leafSym[tid]);
                    for (uint32_t x = (uint32_t)(code >> LUT_BITS); x < (1u << l2); x += 1u << (cl - LUT_BITS))
                        sub[x] = e;
                }
            } else if (cl > LUT_BITS + l2) {
                // too long for the second level too: the first leaf of those that share its LUT_BITS + l2 bits (they lie side by
                // side) leaves its index in their slot, where the search starts
                const uint64_t code = S.leafCode[tid];
                const uint32_t subIdx = S.lut[(uint32_t)code & ((1u << LUT_BITS) - 1u)] & LUT_SUB_MASK;
                const uint64_t pmask = (1ull << (LUT_BITS + l2)) - 1ull;
                const bool first = tid == 0 || S.leafLen[tid - 1] <= LUT_BITS + l2 || ((S.leafCode[tid - 1] ^ code) & pmask) != 0ull;
                if (subIdx < nSub && first)
                    lut2[(subIdx << l2) | ((uint32_t)(code >> LUT_BITS) & ((1u << l2) - 1u))] = (uint16_t)(LUT2_SEARCH | (uint32_t)tid);
            }
        }
    }
    __syncthreads();
}

// CodecM32.decode (CodecM32.java:327-356) of the first nStream values of the byte stream m32[0..nM32), by the whole
// workgroup: value k goes to o[map(k)].  bm / wb: start bitmap and its rank bases (same memory space as m32).
// Returns GF_K_OK, or GF_K_ERR_BOUNDS where the reference's reads run off the stream.
struct CellMapPredictor {
    GfCellMap map;
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return map(k); }
};
struct CellMapIdentity {
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return k; }
};

// Marks the bytes of m32[0..nM32) that start a value (bitmap bm) and ranks them (wb[w] = number of starts before bitmap
// word w); S.carry = number of values in the stream.  Whole workgroup; ends with a barrier.
template <class M32Ptr, class BmPtr>
__device__ __forceinline__ void m32_mark_starts(DecShared &S, M32Ptr m32, uint32_t nM32, BmPtr bm, BmPtr wb)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t bmWords = (nM32 + 31u) >> 5;
    M32Cursor cur;
    cur.m = m32;
    cur.n = nM32;
    cur.pos = 0;
    cur.base = 0;
    cur.d0 = cur.d1 = cur.d2 = 0;
    const uint32_t *m32w = reinterpret_cast<const uint32_t *>(m32);
    if (tid == 0) { S.chainEnd = 0; S.dense = 0; S.nRedo[0] = 0; }
    constexpr uint32_t OPEN_CAP = MAXQ;                 // open words are listed in S.qs (free between the Huffman passes and here)
    __syncthreads();
    // Local resolution, 32 bytes of the stream (one bitmap word) per thread, by bit operations on masks -- no walk over the bytes.
    // I: bytes that may be an introducer (0x7f / 0x81), H: bytes with the continuation bit.  A true introducer at p makes p+1 a
    // payload byte, and p+2 .. p+5 as long as the bytes before them carry the continuation bit (CodecM32.java:327-356: at most
    // five payload bytes): cover(T) below, a handful of shifts for all 64 positions of the window at once.  Which candidates
    // are true introducers depends on the starts (T = S & I) and the starts on them (S = ~cover(T)): iterate from "every
    // candidate is one" -- the iterates close in on the solution from both sides and settle from the left, one nested
    // candidate at least per turn; on terrain data candidates stand alone and the second turn confirms the first.  The window
    // is the thread's word and the word before it, entered at an anchor: a byte with no candidate among the five before it is
    // certainly a start (no value is longer than six bytes).  No anchor in the 27 bytes before the word, or no fixed point
    // after eight turns (a stream dense in multi-byte values): the chain resolution below.
    // (Round 2 walked the values from an anchor per DWORD, a divergent loop over a byte cursor that some lane of every wave was in:
    // with the value decode below about half of the kernel's instructions.)
    {
        // 0x80 in every byte that IS 0x7f or 0x81 -- exactly: the masks decide which bytes a value covers, so the usual
        // (v - 0x01010101) & ~v test will not do here (its borrow marks a 0x01 byte above a zero byte: 0x80 behind 0x81, 0x7e
        // behind 0x7f; tools/soak.py, seed 30031003, found the packing that has 0x81 0x80 behind a six-byte value).  Per byte:
        // (v & 0x7f) + 0x7f carries into bit 7 iff the low seven bits are not all zero; or v: bit 7 set iff the byte is not zero
        auto isZero = [](uint32_t v) -> uint32_t { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };
        auto cand = [&](uint32_t x) -> uint32_t { return isZero(x ^ 0x7F7F7F7Fu) | isZero(x ^ 0x81818181u); };
        auto nib4 = [](uint32_t f) -> uint32_t {      // 0x80-per-byte flags -> 4 bits (the partial products do not collide)
            return (((f >> 7) * 0x00204081u) >> 21) & 0xfu;
        };
        const uint32_t turns = (bmWords + DEC_THREADS - 1u) / DEC_THREADS;
        for (uint32_t turn = 0; turn < turns; turn++) {
            const uint32_t w = turn * DEC_THREADS + tid;
            if (w < bmWords) {
                uint32_t I = 0, H = 0;
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) {
                    const uint32_t d = m32w[8u * w + j];
                    I |= nib4(cand(d)) << (4u * j);
                    H |= nib4(d & 0x80808080u) << (4u * j);
                }
                const uint32_t left = nM32 - 32u * w, valid = left >= 32u ? 0xFFFFFFFFu : (1u << left) - 1u;   // stale bytes beyond nM32
                bm[w] = I & valid;
                wb[w] = H & valid;
            }
        }
        __syncthreads();
        // the masks give way to the starts word by word; a word needs its left neighbour's masks: last turn first
        for (uint32_t turn = turns; turn-- > 0u;) {
            const uint32_t w = turn * DEC_THREADS + tid;
            uint32_t starts = 0;
            if (w < bmWords) {
                unsigned long long I64 = ((unsigned long long)bm[w] << 32) | (w ? bm[w - 1u] : 0u);
                const unsigned long long H64 = ((unsigned long long)wb[w] << 32) | (w ? wb[w - 1u] : 0u);
                // Anchors (round 4): the bytes that NO candidate of the window could cover even if every one of them were a true
                // introducer -- a byte inside a value lies behind that value's introducer with nothing but continuation bytes in
                // between, so a byte no candidate reaches that way starts a value.  Positions 0..4 of the window do not see all of
                // the five bytes before them and do not count (the word before the first one is empty: byte 0 of the stream is an
                // anchor).  (Round 3 asked for five bytes without any candidate in front of an anchor: a run of three-byte values --
                // the border residuals of a tile whose rows have little to do with each other -- has none for hundreds of bytes,
                // and every tile of the rough workload had words left open.)
                unsigned long long reach;
                {
                    unsigned long long c = I64 << 1;
                    reach = c;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        c = (c & H64) << 1;
                        reach |= c;
                    }
                }
                const unsigned long long anchors = ~reach & ~0x1Full & 0x1FFFFFFFFull;                  // up to my first byte (bit 32)
                bool settled = false;
                unsigned long long Sx = 0;
                if (anchors) {
                    I64 &= ~0ull << (63u - (uint32_t)__builtin_clzll(anchors));
                    unsigned long long T = I64;
                    for (int it = 0; it < 8 && !settled; it++) {
                        unsigned long long c = T << 1, cover = c;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            c = (c & H64) << 1;
                            cover |= c;
                        }
                        Sx = ~cover;
                        const unsigned long long Tn = Sx & I64;
                        settled = Tn == T;
                        T = Tn;
                    }
                }
                const uint32_t left = nM32 - 32u * w, valid = left >= 32u ? 0xFFFFFFFFu : (1u << left) - 1u;
                starts = (uint32_t)(Sx >> 32) & valid;
                if (!settled) {
                    // round 4: the word is left OPEN (no start marked) and listed; the words around it that did settle are
                    // final, and an open word is walked from the end of its left neighbour's last value below.  (The whole
                    // tile used to fall back to the chain resolution for one such word: a steep slope next to scree in the
                    // rough workload cost the tile ten times its usual time.)
                    starts = 0;
                    const uint32_t slot = atomicAdd(&S.nRedo[0], 1u);
                    if (slot < OPEN_CAP) S.qs[slot] = w;
                }
            }
            __syncthreads();                            // every mask of this turn has been read
            if (w < bmWords) bm[w] = starts;
        }
    }
    __syncthreads();
    {
        const uint32_t nOpen = S.nRedo[0];
        if (nOpen > OPEN_CAP || 8u * nOpen > bmWords + 56u || ((bmWords + 31u) >> 5) > (uint32_t)MAXQ) {
            if (tid == 0) S.dense = 1;                  // dense in multi-byte values all over: the chain resolution
        } else if (nOpen) {
            // A value is at most six bytes, so every 32-byte word holds a start: an open word whose left neighbour is final
            // begins where that neighbour's last value ends, and is walked from there, a lane per word.  Runs of open words
            // take a round per word of the run.
            uint32_t *openBits = S.qe;                  // bit w: word w is still open
            for (uint32_t i = tid; i < (bmWords + 31u) >> 5; i += DEC_THREADS) openBits[i] = 0;
            __syncthreads();
            for (uint32_t e = tid; e < nOpen; e += DEC_THREADS) atomicOr(&openBits[S.qs[e] >> 5], 1u << (S.qs[e] & 31u));
            __syncthreads();
            for (;;) {
                int progress = 0;
                uint32_t doneW = 0xFFFFFFFFu, doneMask = 0;
                for (uint32_t e = tid; e < nOpen; e += DEC_THREADS) {   // (nOpen <= DEC_THREADS as a rule: one word per thread)
                    const uint32_t w = S.qs[e];
                    if (!((openBits[w >> 5] >> (w & 31u)) & 1u)) continue;
                    if (w > 0u && ((openBits[(w - 1u) >> 5] >> ((w - 1u) & 31u)) & 1u)) continue;   // its neighbour first
                    M32Cursor c = cur;
                    uint32_t pos = 32u * w;
                    const uint32_t before = w > 0u ? bm[w - 1u] : 0u;
                    if (before) {
                        c.seek(32u * (w - 1u) + 31u - (uint32_t)__builtin_clz(before));
                        c.next();
                        pos = max(pos, c.pos);
                    }
                    const uint32_t limit = min(nM32, 32u * w + 32u);
                    uint32_t mask = 0;
                    if (pos < limit) {
                        c.seek(pos);
                        while (c.pos < limit) {
                            mask |= 1u << (c.pos & 31u);
                            c.next();
                        }
                    }
                    doneW = w;
                    doneMask = mask;
                    progress = 1;
                    break;                                              // one word per thread and round
                }
                const int any = __syncthreads_or(progress);             // every neighbour's state has been read
                if (doneW != 0xFFFFFFFFu) {
                    bm[doneW] = doneMask;
                    atomicAnd(&openBits[doneW >> 5], ~(1u << (doneW & 31u)));
                }
                __syncthreads();
                if (!any) break;
            }
        }
    }
    __syncthreads();
    if (S.dense) {
        // chain resolution over the bytes (same scheme as the Huffman text), then mark the starts
        for (uint32_t w = tid; w < bmWords; w += DEC_THREADS) bm[w] = 0;
        uint32_t unit = (nM32 + MAXQ - 1) / MAXQ;
        unit = max(16u, unit);
        const uint32_t Q = max(1u, (nM32 + unit - 1) / unit);
        resolve_chain<0>(S, cur, 0u, nM32, unit, Q, 8u);   // warm-up: 8 bytes
        for (uint32_t q = tid; q < Q; q += DEC_THREADS) {
            const uint32_t limit = min(nM32, (q + 1) * unit);
            M32Cursor c = cur;
            c.seek(S.qs[q]);
            uint32_t word = c.pos >> 5, mask = 0;
            while (c.pos < limit) {
                const uint32_t w = c.pos >> 5;
                if (w != word) {
                    if (mask) atomicOr(&bm[word], mask);
                    word = w;
                    mask = 0;
                }
                mask |= 1u << (c.pos & 31u);
                c.next();
            }
            if (mask) atomicOr(&bm[word], mask);
        }
        __syncthreads();
    }
    // rank base of every bitmap word (exclusive popcount prefix)
    if (tid == 0) S.carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < bmWords; base += DEC_THREADS) {
        const uint32_t w = base + tid;
        const uint32_t pc = w < bmWords ? (uint32_t)__popc(bm[w]) : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(pc, S.waveSum, &tot);
        if (w < bmWords) wb[w] = S.carry + ex;
        __syncthreads();
        if (tid == 0) S.carry += tot;
        __syncthreads();
    }
}

template <class M32Ptr, class BmPtr, class Map>
__device__ __forceinline__ int32_t m32_to_values(DecShared &S, M32Ptr m32, uint32_t nM32, BmPtr bm, BmPtr wb, uint32_t nStream,
                                                 const Map map, uint32_t *o)
{
    const uint32_t tid = threadIdx.x;
    int32_t tileStatus = GF_K_OK;
    const uint32_t *m32w = reinterpret_cast<const uint32_t *>(m32);
    const uint32_t nDw = (nM32 + 3u) >> 2;
    m32_mark_starts(S, m32, nM32, bm, wb);
    if (S.carry < nStream) tileStatus = GF_K_ERR_BOUNDS;         // predictor reads past codeM32s
    // four byte positions per thread and step: consecutive bytes are (mostly) consecutive cells,
    // so the stores of a wave are coalesced.  12 bytes of the buffer cover every value that
    // starts in the thread's dword.
    for (uint32_t dw = tid; dw < nDw; dw += DEC_THREADS) {
        const uint32_t i0 = dw << 2;
        const uint32_t bits = (bm[i0 >> 5] >> (i0 & 31u)) & 0xfu;
        if (bits) {
            const uint32_t word = bm[i0 >> 5];
            uint32_t k = wb[i0 >> 5] + (uint32_t)__popc(word & ((1u << (i0 & 31u)) - 1u));
            uint32_t d0 = m32w[dw];
            // Fast path (most dwords of terrain data): four value starts, none an introducer (0x7f / 0x81) or the
            // null code (0x80) -> four sign-extended bytes; if their cells are neighbours, one 16-byte store.
            {
                const uint32_t p7 = d0 ^ 0x7F7F7F7Fu, p1 = d0 ^ 0x81818181u, p0 = d0 ^ 0x80808080u;
                const uint32_t special = (((p7 - 0x01010101u) & ~p7) | ((p1 - 0x01010101u) & ~p1) | ((p0 - 0x01010101u) & ~p0)) &
                                         0x80808080u;
                if (bits == 0xfu && !special && k + 3u < nStream) {
                    const uint32_t v0 = (uint32_t)(int32_t)(int8_t)(d0 & 0xffu), v1 = (uint32_t)((int32_t)(d0 << 16) >> 24),
                                   v2 = (uint32_t)((int32_t)(d0 << 8) >> 24), v3 = (uint32_t)((int32_t)d0 >> 24);
                    const uint32_t c0 = map(k);
                    const uint32_t c3 = map(k + 3u);
                    if (c3 - c0 == 3u) {
                        GfU4 q;
                        q.x = v0; q.y = v1; q.z = v2; q.w = v3;
                        *reinterpret_cast<GfU4 *>(o + c0) = q;
                    } else {
                        o[c0] = v0;
                        o[map(k + 1u)] = v1;
                        o[map(k + 2u)] = v2;
                        o[c3] = v3;
                    }
                    continue;
                }
            }
            // bytes i0 .. i0+11, zero beyond nM32
            uint32_t d1 = dw + 1 < nDw ? m32w[dw + 1] : 0u, d2 = dw + 2 < nDw ? m32w[dw + 2] : 0u;
            if (i0 + 12 > nM32) {
                const uint32_t valid = nM32 - i0;            // 1..11 bytes
                if (valid < 4) d0 &= (1u << (valid * 8u)) - 1u;
                if (valid < 8) d1 &= valid > 4 ? (1u << ((valid - 4u) * 8u)) - 1u : 0u;
                d2 &= valid > 8 ? (1u << ((valid - 8u) * 8u)) - 1u : 0u;
            }
    #pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                if ((bits >> j) & 1u) {
                    if (k < nStream) {
                        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, 8u * j);
                        const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, 8u * j);
                        uint32_t vlen;
                        const uint32_t val = m32_value(lo, hi, &vlen);
                        o[map(k)] = val;
                        if (k == nStream - 1) S.chainEnd = i0 + j + vlen;
                    }
                    k++;
                }
            }
        }
    }
    __syncthreads();
    if (tileStatus == GF_K_OK && S.chainEnd > nM32) tileStatus = GF_K_ERR_BOUNDS;   // last value truncated
    return tileStatus;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused phases 2 + 3: M32 bytes -> cell values, every cell stored to HBM exactly once.
//
// The predictor inverses are prefix sums (int32 wrap-around), so they run on the residuals while these are still on the
// chip.  With F = running sum of the residuals in STREAM order (the order the reference emits them, row-major inside
// the interior), a row's own prefix is F minus F at the row start, and what is left per model is a per-row constant:
//   Differencing (PredictorModelDifferencing.java:145-167)   v(r,c) = C(r) + F(r,c) - F(r,0)
//   Linear       (PredictorModelLinear.java:66-101)          v(r,c) = v(r,1) + (c-1) (d1(r) - F1ex(r)) + F2(r,c) - F2ex(r),
//                F2 = running sum of F1 (the second differences are summed twice), d1 = v(r,1) - v(r,0)
//   Triangle     (PredictorModelTriangle.java:62-98)         v(r,c) = v(r-1,c) + c0(r) + F(r,c) - Fex(r)   (2-D prefix sum)
// C(r) = column 0 (a running sum over the rows of the column-0 residuals), Fex = F just before the row's first interior
// element.  The border residuals (column 0; row 0 for Triangle; column 1 for Linear) sit at stream positions known in
// closed form and are fetched once per tile by select() on the start bitmap.
//
// The M32 bytes are swept in chunks of DEC_THREADS dwords (<= 4 values per thread): decode, workgroup scan, F into a ring
// in LDS (the decode tables of phase 1 are dead by now: the ring and three per-row arrays alias them); the rows a chunk
// completes are then finished from the ring -- lane = consecutive cells of a row, so every store instruction writes
// whole contiguous lines -- and for Triangle the column recurrence is carried in a register per column.
constexpr uint32_t SCR_WORDS = offsetof(DecShared, waveSum) / 4;  // lut .. head: free after phase 1 (the header fields are in registers)
constexpr uint32_t FUSED_CHUNK = 4u * DEC_THREADS;                // values a chunk can hold at most

// Segmented inclusive scan over the wave (DPP): s = sum of the lane's segment up to the lane, f != 0 = a segment head lies at or
// before the lane inside the scanned range (its value is the head's: heads hand their mark down their segment).  A lane that
// has a head of its own takes nothing from the left.
__device__ __forceinline__ void gf_wave_seg_incl_scan(uint32_t &s, uint32_t &f)
{
#define GF_SEG_STEP(ctrl, rowmask, bc)                                                          \
    {                                                                                           \
        const uint32_t ps = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, ctrl, rowmask, 0xf, bc); \
        const uint32_t pf = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, ctrl, rowmask, 0xf, bc); \
        const bool open = f == 0u;                                                              \
        s += open ? ps : 0u;                                                                    \
        f = open ? pf : f;                                                                      \
    }
    GF_SEG_STEP(0x111, 0xf, true)      // row_shr:1
    GF_SEG_STEP(0x112, 0xf, true)      // row_shr:2
    GF_SEG_STEP(0x114, 0xf, true)      // row_shr:4
    GF_SEG_STEP(0x118, 0xf, true)      // row_shr:8
    GF_SEG_STEP(0x142, 0xa, false)     // row_bcast:15 -> rows 1, 3
    GF_SEG_STEP(0x143, 0xc, false)     // row_bcast:31 -> rows 2, 3
#undef GF_SEG_STEP
}
// (s, f) of the lane before (lane 0: nothing)
__device__ __forceinline__ void gf_wave_shr1(uint32_t &s, uint32_t &f)
{
    s = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x138, 0xf, 0xf, false);   // wave_shr:1
    f = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x138, 0xf, 0xf, false);
}
// a then b of one sequence
__device__ __forceinline__ void gf_seg_combine(uint32_t as, uint32_t af, uint32_t bs, uint32_t bf, uint32_t &rs, uint32_t &rf)
{
    rs = bf ? bs : as + bs;
    rf = bf ? bf : af;
}

struct FusedPlan {
    uint32_t ring;          // ring entries (0 = tile shape not eligible)
    bool endBarrier;        // the ring is too short to overlap the next chunk's writes with this chunk's reads
};
__device__ __forceinline__ FusedPlan fused_plan(uint32_t nR, uint32_t nC, int model)
{
    FusedPlan p;
    p.ring = 0;
    p.endBarrier = false;
    // nC <= 2 * DEC_THREADS: two column registers per thread (Triangle); the products behind the reciprocal divisions stay
    // below 2^32 by a wide margin at these sizes.  Row arrays behind the ring: two, a third one for Linear
    const uint32_t rowWords = (model == 2 ? 3u : 2u) * nR;
    if (model == 4) {
        // the nulls predictor keeps no ring (m32_to_tile_nulls): two row arrays and the scan's wave totals
        if (nR >= 1u && nC >= 2u && 2u * nR + 4u * DEC_WAVES <= SCR_WORDS) p.ring = 1u;
        return p;
    }
    if (nR < 2u || nC < 4u || nC > 2u * DEC_THREADS || nR > 4096u || rowWords + nC + FUSED_CHUNK + 1u > SCR_WORDS) return p;
    const uint32_t avail = SCR_WORDS - rowWords;
    p.ring = avail;
    p.endBarrier = avail < 2u * FUSED_CHUNK + nC;
    return p;
}

// byte position of value start number k (k < number of starts)
template <class BmPtr>
__device__ __forceinline__ uint32_t m32_select(BmPtr bm, BmPtr wb, uint32_t bmWords, uint32_t k)
{
    // every value has at least one byte, so its word is >= k / 32 -- and close to it unless multi-byte values are frequent:
    // gallop from there, then bisect
    uint32_t lo = min(k >> 5, bmWords - 1u), hi = bmWords, step = 1u;
    while (lo + step < bmWords) {
        if (wb[lo + step] > k) { hi = lo + step; break; }
        lo += step;
        step <<= 1;
    }
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (wb[mid] <= k) lo = mid;
        else hi = mid;
    }
    uint32_t x = bm[lo], j = k - wb[lo], pos = 0;
    uint32_t c = (uint32_t)__popc(x & 0xffffu);
    if (j >= c) { j -= c; pos += 16u; x >>= 16; }
    c = (uint32_t)__popc(x & 0xffu);
    if (j >= c) { j -= c; pos += 8u; x >>= 8; }
    c = (uint32_t)__popc(x & 0xfu);
    if (j >= c) { j -= c; pos += 4u; x >>= 4; }
    c = (uint32_t)__popc(x & 0x3u);
    if (j >= c) { j -= c; pos += 2u; x >>= 2; }
    if (j >= (x & 1u)) pos += 1u;
    return (lo << 5) + pos;
}

// the value that starts at byte p of the stream; *len = its byte count
template <class M32Ptr>
__device__ __forceinline__ uint32_t m32_value_at(M32Ptr m32, uint32_t nM32, uint32_t p, uint32_t *len)
{
    M32Cursor c;
    c.m = m32;
    c.n = nM32;
    c.seek(p);
    uint32_t lo, hi;
    c.prepare(&lo, &hi);
    return m32_value(lo, hi, len);
}

template <class M32Ptr, class BmPtr>
__device__ __forceinline__ int32_t m32_to_tile(DecShared &S, M32Ptr m32, uint32_t nM32, BmPtr bm, BmPtr wb, const int model,
                                               const uint32_t seed, const uint32_t nR, const uint32_t nC, const uint32_t nStream,
                                               const FusedPlan plan, uint32_t *__restrict__ o, uint32_t *stamps, const uint32_t diagFlags)
{
#ifdef GF_DIAG
#define GF_TSTAMP(i) do { if (stamps && threadIdx.x == 0) stamps[i] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define GF_TSTAMP(i) do { (void)stamps; } while (0)
#endif
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = gf_wave_id();
    const uint32_t *m32w = reinterpret_cast<const uint32_t *>(m32);
    const uint32_t nDw = (nM32 + 3u) >> 2, bmWords = (nM32 + 31u) >> 5;
    m32_mark_starts(S, m32, nM32, bm, wb);
    const uint32_t nValues = S.carry;
    if (nValues < nStream) return GF_K_ERR_BOUNDS;               // predictor reads past codeM32s
    GF_TSTAMP(6);
#ifdef GF_DIAG
    if ((diagFlags & 0xffu) == 4u) return GF_K_OK;               // diagnostic: instruction counts of the parts (tools/pmc_phases_dec.sh)
#endif

    uint32_t *scr = reinterpret_cast<uint32_t *>(&S);
    const uint32_t RING = plan.ring;
    uint32_t *ring = scr, *rowA = scr + RING, *rowB = rowA + nR, *rowC = rowB + nR;
    const uint32_t magicR = (uint32_t)(((1ull << 32) + RING - 1u) / RING);
    auto ringIdx = [&](uint32_t t) -> uint32_t { return t - __umulhi(t, magicR) * RING; };   // t * RING < 2^32

    // interior stream: element k of the stream is interior element t = k + tOff (t < 0: border), rows of W elements
    const uint32_t W = model == 1 ? nC : model == 2 ? nC - 2u : nC - 1u;
    const uint32_t nB = model == 1 ? 0u : model == 2 ? 2u * nR - 1u : nC + nR - 2u;
    const uint32_t tOff = model == 1 ? 1u : 0u - nB;
    const uint32_t nRowsI = model == 3 ? nR - 1u : nR;
    const uint32_t magicW = (uint32_t)(((1ull << 32) + W - 1u) / W);
    const uint32_t magicC = (uint32_t)(((1ull << 32) + nC - 1u) / nC);

    // Up to four values of one dword of the stream: slot j = the value that starts at byte j (bits: the dword's start flags, k: rank
    // of its first start).  Every byte is taken as a one-byte value first (a slot that starts no value is masked out by its
    // flag); the slots whose byte is an introducer or the null code -- a few per cent on terrain data, but some lane of every
    // wave has one -- are then redone one at a time: the loop runs as often as the wave's worst dword has such bytes, once as a rule.
    auto decodeDword = [&](const uint32_t dw, uint32_t &bits, uint32_t &k, uint32_t &v0, uint32_t &v1, uint32_t &v2, uint32_t &v3) {
        const uint32_t i0 = dw << 2;
        const uint32_t word = bm[i0 >> 5];
        bits = (word >> (i0 & 31u)) & 0xfu;
        k = wb[i0 >> 5] + (uint32_t)__popc(word & ((1u << (i0 & 31u)) - 1u));
        uint32_t d0 = m32w[dw];
        v0 = (uint32_t)(int32_t)(int8_t)(d0 & 0xffu);
        v1 = (uint32_t)((int32_t)(d0 << 16) >> 24);
        v2 = (uint32_t)((int32_t)(d0 << 8) >> 24);
        v3 = (uint32_t)((int32_t)d0 >> 24);
        const uint32_t p7 = d0 ^ 0x7F7F7F7Fu, p1 = d0 ^ 0x81818181u, p0 = d0 ^ 0x80808080u;
        const uint32_t special = (((p7 - 0x01010101u) & ~p7) | ((p1 - 0x01010101u) & ~p1) | ((p0 - 0x01010101u) & ~p0)) & 0x80808080u;
        uint32_t redo = bits & ((((special >> 7) * 0x00204081u) >> 21) & 0xfu);
        if (redo) {
            uint32_t d1 = dw + 1 < nDw ? m32w[dw + 1] : 0u, d2 = dw + 2 < nDw ? m32w[dw + 2] : 0u;
            if (i0 + 12 > nM32) {
                const uint32_t valid = nM32 - i0;        // 1..11 bytes
                if (valid < 4) d0 &= (1u << (valid * 8u)) - 1u;
                if (valid < 8) d1 &= valid > 4 ? (1u << ((valid - 4u) * 8u)) - 1u : 0u;
                d2 &= valid > 8 ? (1u << ((valid - 8u) * 8u)) - 1u : 0u;
            }
            do {
                const uint32_t j = (uint32_t)__builtin_ctz(redo);
                redo &= redo - 1u;
                const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, 8u * j);
                const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, 8u * j);
                uint32_t vlen;
                const uint32_t val = m32_value(lo, hi, &vlen);
                v0 = j == 0u ? val : v0;
                v1 = j == 1u ? val : v1;
                v2 = j == 2u ? val : v2;
                v3 = j == 3u ? val : v3;
                if (k + (uint32_t)__popc(bits & ((1u << j) - 1u)) == nStream - 1u) S.chainEnd = i0 + j + vlen;
            } while (redo);
        }
    };

    if (model == 4) {
        // ---- PredictorModelDifferencingWithNulls.decode (:137-166), fused: one store per cell, no second pass over the tile ----
        // The stream holds every cell in row-major order; a value is the sum of the residuals since the last RESTART plus what
        // the restart starts from: behind a null residual the seed; at a row's first cell the first cell of the row before --
        // unless that VALUE is the null code (:162-163), then the seed again (row 0: the seed).  So: a segmented scan in stream
        // order.  Elements: null residual = head (kind 1: the seed follows), sum 0; first cell of a row = head (kind 3: the row's
        // base follows), sum = its residual; others = no head.  Heads hand their kind down their segment, so every cell knows
        // what its sum is to be added to; the row bases B(r) come from the column-0 chain, which is worked out first (its
        // residuals are elements r nC of the stream: m32_select).  A sum may come out as the null code without any null residual
        // (damaged input only); inside a row that changes nothing (the flag follows the residual there), on the column-0 chain
        // it restarts the NEXT row from the seed: the chain is checked for it and redone the reference's way, serially, then.
        uint32_t *x0 = scr, *rowBase = scr + nR, *wtot = scr + 2u * nR;              // column-0 residuals, B(r), wave totals of the scan
        constexpr uint32_t K_NULL = 1u, K_ROW = 3u;
        for (uint32_t r = tid; r < nR; r += DEC_THREADS) {
            uint32_t vlen;
            x0[r] = m32_value_at(m32, nM32, m32_select(bm, wb, bmWords, r * nC), &vlen);
        }
        __syncthreads();
        if (wave == 0) {
            // C(r) = value of cell (r, 0); B(r) = what row r starts from
            uint32_t cs = 0, cf = 0;                                                  // carry: sum and kind of the running segment
            bool odd = false;
            for (uint32_t rb = 0; rb < nR; rb += 64u) {
                const uint32_t r = rb + lane;
                const uint32_t x = r < nR ? x0[r] : GF_NULL_CODE;
                const bool isNull = x == GF_NULL_CODE;
                uint32_t es = isNull ? 0u : x, ef = isNull ? K_NULL : 0u;
                gf_wave_seg_incl_scan(es, ef);
                uint32_t is, ifl;
                gf_seg_combine(cs, cf, es, ef, is, ifl);                             // inclusive, with the rows before
                // C(r): a segment of the chain starts from the seed (the rows before the first null included: carry kind 0)
                const uint32_t C = isNull ? GF_NULL_CODE : seed + is;
                odd = odd || (!isNull && C == GF_NULL_CODE && r < nR);
                // B(r + 1) = C(r) unless that is the null code
                if (r + 1u < nR) rowBase[r + 1u] = C == GF_NULL_CODE ? seed : C;
                cs = (uint32_t)__builtin_amdgcn_readlane((int)is, 63);
                cf = (uint32_t)__builtin_amdgcn_readlane((int)ifl, 63);
            }
            if (lane == 0) rowBase[0] = seed;
            if (__any(odd)) {
                // a sum hit the null code: the reference's loop over the rows, one lane
                if (lane == 0) {
                    uint32_t prior = seed;
                    bool nullFlag = true;
                    for (uint32_t r = 0; r < nR; r++) {
                        rowBase[r] = nullFlag ? seed : prior;
                        const uint32_t x = x0[r];
                        prior = x == GF_NULL_CODE ? GF_NULL_CODE : rowBase[r] + x;
                        nullFlag = prior == GF_NULL_CODE;
                    }
                }
            }
        }
        __syncthreads();
        uint32_t carryS = 0, carryF = 0;
        const uint32_t nIterN = (nDw + DEC_THREADS - 1u) / DEC_THREADS;
        for (uint32_t it = 0; it < nIterN; it++) {
            const uint32_t dw = it * DEC_THREADS + tid;
            uint32_t bits = 0, k = 0, v[4] = {0, 0, 0, 0};
            if (dw < nDw) decodeDword(dw, bits, k, v[0], v[1], v[2], v[3]);
            // the thread's elements in order: position, row / column, element of the scan; local inclusive scan
            uint32_t t[4], ls[4], lf[4];
            bool ok[4], nul[4];
            uint32_t kk = k, rs = 0, rf = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                t[j] = kk;
                ok[j] = ((bits >> j) & 1u) && kk < nStream;
                kk += (bits >> j) & 1u;
                nul[j] = v[j] == GF_NULL_CODE;
                const uint32_t r = __umulhi(t[j], magicC), c = t[j] - r * nC;
                const uint32_t es = ok[j] && !nul[j] ? v[j] : 0u;
                const uint32_t ef = !ok[j] ? 0u : nul[j] ? K_NULL : c == 0u ? K_ROW : 0u;
                gf_seg_combine(rs, rf, es, ef, rs, rf);
                ls[j] = rs;
                lf[j] = rf;
            }
            uint32_t ws = rs, wf = rf;
            gf_wave_seg_incl_scan(ws, wf);
            uint32_t *wt = wtot + (it & 1u) * (2u * DEC_WAVES);                      // double-buffered by chunk parity
            if (lane == 63u) { wt[wave] = ws; wt[DEC_WAVES + wave] = wf; }
            gf_wave_shr1(ws, wf);                                                    // exclusive: what lies before the thread in its wave
            __syncthreads();
            uint32_t bs = carryS, bf = carryF;                                       // ... before the wave
            {
                uint32_t cs = carryS, cf = carryF;
#pragma unroll
                for (uint32_t w = 0; w < (uint32_t)DEC_WAVES; w++) {
                    if (w == wave) { bs = cs; bf = cf; }
                    gf_seg_combine(cs, cf, wt[w], wt[DEC_WAVES + w], cs, cf);
                }
                carryS = cs;
                carryF = cf;
            }
            gf_seg_combine(bs, bf, ws, wf, bs, bf);                                  // before the thread
            uint32_t out[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t fs, ff;
                gf_seg_combine(bs, bf, ls[j], lf[j], fs, ff);
                const uint32_t r = __umulhi(t[j], magicC);
                const uint32_t base = ff == K_ROW ? rowBase[min(r, nR - 1u)] : seed;
                out[j] = nul[j] ? GF_NULL_CODE : base + fs;
            }
            if (bits == 0xfu && t[3] < nStream) {
                GfU4 q;
                q.x = out[0]; q.y = out[1]; q.z = out[2]; q.w = out[3];
                *reinterpret_cast<GfU4 *>(o + t[0]) = q;                             // four cells in a row: lanes side by side
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (ok[j]) o[t[j]] = out[j];
            }
        }
        __syncthreads();
        return S.chainEnd > nM32 ? GF_K_ERR_BOUNDS : GF_K_OK;                        // last value truncated
    }

    // ---- borders: per-row constants (and row 0 of Triangle) ----
    // Linear and Triangle emit their border residuals FIRST (2 nR - 1, nC + nR - 2 elements): the head of the stream is decoded a
    // dword per thread like any chunk, and every border value goes to its place in LDS by its rank -- row 0 of Triangle into the
    // (still unused) ring, the column-0 / column-1 residuals into the row arrays.  (Round 2 fetched each of them through
    // m32_select, a gallop and a bisection per value: 5 K of the kernel's 90 K instructions per tile.)  Differencing has its
    // column 0 spread over the stream and keeps the selects.
    uint32_t colPrev0 = 0, colPrev1 = 0;
    {
        if (model != 1) {
            for (uint32_t base = 0; base < nDw; base += DEC_THREADS) {
                // values before this turn's first byte (wave-uniform); the turns end where the border does
                const uint32_t kBase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wb[base >> 3]);
                if (kBase >= nB) break;
                const uint32_t dw = base + tid;
                // (a wave whose first byte already lies behind the border has nothing to pick up: on an ETOPO1-shaped tile the
                // border is the first 300 bytes of the stream, i.e. the dwords of two of the eight waves)
                const uint32_t dwWave = base + wave * 64u;
                const bool waveIn = dwWave < nDw && (uint32_t)__builtin_amdgcn_readfirstlane((int)wb[min(dwWave >> 3, bmWords - 1u)]) < nB;
                if (waveIn && dw < nDw) {
                    uint32_t bits, k, v[4];
                    decodeDword(dw, bits, k, v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) {
                        if (((bits >> j) & 1u) && k < nB) {
                            if (model == 3) {
                                if (k < nC - 1u) ring[k + 1u] = v[j];              // cell (0, k + 1)
                                else rowB[k - (nC - 2u)] = v[j];                   // cell (r, 0), r = k - (nC - 2)
                            } else if (k == 0u) rowB[0] = v[j];                    // Linear: cell (0, 1)
                            else if ((k - 1u) & 1u) rowB[1u + ((k - 1u) >> 1)] = v[j];   // cell (r, 1)
                            else rowA[1u + ((k - 1u) >> 1)] = v[j];                // cell (r, 0)
                        }
                        k += (bits >> j) & 1u;
                    }
                }
            }
            __syncthreads();
        }
        uint32_t vlen;
        if (model == 3) {
            // row 0: stream elements 0 .. nC-2 are cells (0,1) .. (0,nC-1), each relative to its left neighbour
            uint32_t carry = seed;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t c = tid + (uint32_t)u * DEC_THREADS;
                if (u == 1 && nC <= (uint32_t)DEC_THREADS) break;
                const uint32_t x = (c >= 1u && c < nC) ? ring[c] : 0u;
                uint32_t tot;
                const uint32_t v = carry + block_excl_scan(x, S.waveSum, &tot) + x;
                if (c < nC) o[c] = v;
                if (u == 0) colPrev0 = v;
                else colPrev1 = v;
                carry += tot;
            }
        }
        uint32_t carry = seed;
        for (uint32_t rb = 0; rb < nR; rb += DEC_THREADS) {
            const uint32_t r = rb + tid;
            uint32_t x0 = 0;
            if (r >= 1u && r < nR) {
                if (model == 1) x0 = m32_value_at(m32, nM32, m32_select(bm, wb, bmWords, r * nC - 1u), &vlen);
                else x0 = model == 3 ? rowB[r] : rowA[r];
            }
            uint32_t tot;
            const uint32_t cv = carry + block_excl_scan(x0, S.waveSum, &tot) + x0;
            if (r < nR) rowA[r] = cv;                            // column 0 (rowB: Triangle column-0 residual, Linear v(r,1) - v(r,0))
            carry += tot;
        }
        if (tid == 0) ring[model == 1 ? 0u : RING - 1u] = 0u;     // F before the first interior element
    }
    // (the scan barrier of the first chunk orders these stores before the first reads)

    GF_TSTAMP(7);
#ifdef GF_DIAG
    if ((diagFlags & 0xffu) == 5u) return GF_K_OK;
    const bool skipRows = (diagFlags & 0x40000u) != 0u;          // ... the chunk loop without the rows finished from the ring
#else
    constexpr bool skipRows = false;
#endif
    uint32_t carry1 = 0, carry2 = 0, rowsDone = 0;
    const uint32_t nIter = (nDw + DEC_THREADS - 1u) / DEC_THREADS;
    // Stage 2 of chunk i runs AFTER the scan barrier of chunk i + 1: that barrier also publishes chunk i's ring entries, so a
    // chunk costs one barrier.  The ring keeps two chunks and a row (fused_plan), so the writes of chunk i + 1 that follow the
    // same barrier never touch what stage 2 of chunk i still reads; with a shorter ring two more barriers frame stage 2.
    auto finishRows = [&](uint32_t chunk) {
        uint32_t kEnd;
        {
            const uint32_t pEnd = (chunk + 1u) * FUSED_CHUNK;       // multiple of 32
            kEnd = pEnd >= nM32 ? nValues : wb[pEnd >> 5];
            kEnd = min(kEnd, nStream);
        }
        const uint32_t tEnd = kEnd + tOff;
        uint32_t rowsNew = (int32_t)tEnd > 0 ? min(nRowsI, __umulhi(tEnd, magicW)) : 0u;
        rowsNew = (uint32_t)__builtin_amdgcn_readfirstlane((int)rowsNew);
        if (model == 1) {
            const uint32_t cellHi = rowsNew * nC;
            for (uint32_t cell = rowsDone * nC + tid; cell < cellHi; cell += DEC_THREADS) {
                const uint32_t r = __umulhi(cell, magicC);
                o[cell] = rowA[r] + ring[ringIdx(cell)] - ring[ringIdx(r * nC)];
            }
        } else if (model == 2) {
            const uint32_t cellHi = rowsNew * nC;
            for (uint32_t cell = rowsDone * nC + tid; cell < cellHi; cell += DEC_THREADS) {
                const uint32_t r = __umulhi(cell, magicC), c = cell - r * nC;
                const uint32_t cv = rowA[r], d1 = rowB[r];
                uint32_t v = cv;
                if (c == 1u) v = cv + d1;
                else if (c >= 2u) {
                    const uint32_t tS = r * W;
                    const uint32_t F1ex = r ? rowC[r - 1u] : 0u, F2ex = ring[ringIdx(tS + RING - 1u)];
                    v = cv + d1 + (c - 1u) * (d1 - F1ex) + ring[ringIdx(tS + c - 2u)] - F2ex;
                }
                o[cell] = v;
            }
        } else {
            // Triangle: v(r,c) = v(r-1,c) + K(r) + F(r,c), K(r) = column-0 residual of the row - F before its first interior element.
            // Column 0 needs no case of its own: its ring entry "c - 1" is the F before the row, so the same sum gives
            // v(r-1,0) + residual (wrap-around sums are associative).  The column recurrence is one add per row; what a row costs
            // is its bookkeeping, so that is pared down to a running ring position per lane (add, wrap), one LDS read, one
            // v_readlane for K -- the K of up to 64 rows are computed side by side in the lanes first -- one three-operand add
            // and the store.  FIN_ROWS rows go together with every LDS read issued before the first sum.
            // (Round 2 finished a row per iteration and waited for its three reads each time; two of them were the wave-uniform
            // row constants, fetched by every lane, behind a dozen scalar instructions of ring arithmetic per row.)
            constexpr uint32_t FIN_ROWS = 4;
            const uint32_t waveCol0 = tid & ~63u;
            for (uint32_t rb = rowsDone; rb < rowsNew; rb += 64u) {
                const uint32_t nb = min(64u, rowsNew - rb);
                uint32_t Kl = 0;
                if (waveCol0 < nC || (nC > (uint32_t)DEC_THREADS && waveCol0 + DEC_THREADS < nC)) {      // the wave holds a column
                    const uint32_t j = min(lane, nb - 1u), bj = ringIdx((rb + j) * W);
                    Kl = rowB[rb + j + 1u] - ring[bj ? bj - 1u : RING - 1u];
                }
                const uint32_t b0 = ringIdx(rb * W), bm1 = b0 ? b0 - 1u : RING - 1u;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t c = tid + (uint32_t)u * DEC_THREADS;
                    if (u == 1 && nC <= (uint32_t)DEC_THREADS) break;
                    if (c < nC) {
                        uint32_t cp = u == 0 ? colPrev0 : colPrev1;
                        uint32_t e = bm1 + c;                                   // ring position of the row's entry c - 1
                        e = min(e, e - RING);
                        uint32_t cell = (rb + 1u) * nC + c;
                        for (uint32_t j0 = 0; j0 < nb; j0 += FIN_ROWS) {
                            uint32_t a[FIN_ROWS];
#pragma unroll
                            for (uint32_t j = 0; j < FIN_ROWS; j++) {
                                a[j] = ring[e];
                                if (j0 + j + 1u < nb) {                          // (rows beyond the last: the last one again)
                                    e += W;
                                    e = min(e, e - RING);
                                }
                            }
#pragma unroll
                            for (uint32_t j = 0; j < FIN_ROWS; j++) {
                                if (j0 + j < nb) {
                                    cp += (uint32_t)__builtin_amdgcn_readlane((int)Kl, (int)(j0 + j)) + a[j];
                                    o[cell] = cp;
                                    cell += nC;
                                }
                            }
                        }
                        if (u == 0) colPrev0 = cp;
                        else colPrev1 = cp;
                    }
                }
            }
        }
        rowsDone = rowsNew;
    };
#ifdef GF_DIAG
    // wave 0's cycles in the parts of the chunk loop, summed over the chunks (diagFlags bit 17; they go where the stamps of the
    // synchronisation pass are otherwise)
    uint32_t tcA = 0, tcB = 0, tcC = 0, tcD = 0, tcLast = (uint32_t)__builtin_amdgcn_s_memtime();
#define GF_CSTAMP(acc) do { const uint32_t n_ = (uint32_t)__builtin_amdgcn_s_memtime(); acc += n_ - tcLast; tcLast = n_; } while (0)
#else
#define GF_CSTAMP(acc) do { } while (0)
#endif
    for (uint32_t it = 0; it < nIter; it++) {
        const uint32_t dw = it * DEC_THREADS + tid;
        // ---- decode: up to four values, slot j = the value that starts at byte j of the thread's dword ----
        uint32_t bits = 0, k = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (dw < nDw) decodeDword(dw, bits, k, v0, v1, v2, v3);
        // stream index of every slot; a slot counts if it holds an interior element of the stream
        const uint32_t b0 = bits & 1u, b1 = (bits >> 1) & 1u, b2 = (bits >> 2) & 1u, b3 = (bits >> 3) & 1u;
        const uint32_t t0 = k + tOff, t1 = t0 + b0, t2 = t1 + b1, t3 = t2 + b2;
        const uint32_t tLim = nStream + tOff;                    // interior elements: 0 <= t < tLim (t as int32)
        const bool ok0 = b0 && t0 < tLim, ok1 = b1 && t1 < tLim, ok2 = b2 && t2 < tLim, ok3 = b3 && t3 < tLim;
        const uint32_t f0 = ok0 ? v0 : 0u, f1 = f0 + (ok1 ? v1 : 0u), f2 = f1 + (ok2 ? v2 : 0u), f3 = f2 + (ok3 ? v3 : 0u);
        const uint32_t incl1 = gf_wave_incl_scan(f3);
        uint32_t exclN = 0, exclQ = 0;
        // wave totals, double-buffered by chunk parity so that one barrier per chunk suffices: [parity][quantity][wave]
        uint32_t *wt = S.fusedTot + (it & 1u) * (3u * DEC_WAVES);
        if (model == 2) {
            const uint32_t nv = (uint32_t)ok0 + (uint32_t)ok1 + (uint32_t)ok2 + (uint32_t)ok3;
            const uint32_t q = nv * (incl1 - f3) + (ok0 ? f0 : 0u) + (ok1 ? f1 : 0u) + (ok2 ? f2 : 0u) + (ok3 ? f3 : 0u);
            const uint32_t inclN = gf_wave_incl_scan(nv), inclQ = gf_wave_incl_scan(q);
            exclN = inclN - nv;
            exclQ = inclQ - q;
            if (lane == 63u) { wt[DEC_WAVES + wave] = inclN; wt[2 * DEC_WAVES + wave] = inclQ; }
        }
        if (lane == 63u) wt[wave] = incl1;
        GF_CSTAMP(tcA);
        __syncthreads();
        GF_CSTAMP(tcB);
        uint32_t base1 = carry1, base2 = carry2;                 // F1 / F2 before this wave's first element
        if (model == 2) {
            uint32_t c1 = carry1, c2 = carry2;
#pragma unroll
            for (uint32_t w = 0; w < (uint32_t)DEC_WAVES; w++) {
                if (w == wave) { base1 = c1; base2 = c2; }
                c2 += wt[DEC_WAVES + w] * c1 + wt[2 * DEC_WAVES + w];
                c1 += wt[w];
            }
            carry1 = c1;
            carry2 = c2;
        } else {
            // the waves' totals side by side in the first lanes of every wave, summed up by three DPP steps, picked by v_readlane
            // (one LDS read instead of a read and an add per wave)
            static_assert(DEC_WAVES <= 16, "row_shr steps stay inside a row of sixteen lanes");
            int x = (int)wt[lane & (uint32_t)(DEC_WAVES - 1)];
            x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);    // row_shr:1
            x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);    // row_shr:2
            x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);    // row_shr:4
            if (DEC_WAVES > 8) x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
            const uint32_t waveU = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
            base1 = carry1 + (waveU ? (uint32_t)__builtin_amdgcn_readlane(x, (int)waveU - 1) : 0u);
            carry1 += (uint32_t)__builtin_amdgcn_readlane(x, DEC_WAVES - 1);
        }
        {
            const uint32_t e1 = base1 + (incl1 - f3);            // F1 before this thread's first slot
            uint32_t F0 = e1 + f0, F1 = e1 + f1, F2 = e1 + f2, F3 = e1 + f3;
            if (model == 2) {
                uint32_t g = base2 + base1 * exclN + exclQ;      // F2 before this thread's first slot
                const uint32_t G0 = g + F0;
                g = ok0 ? G0 : g;
                const uint32_t G1 = g + F1;
                g = ok1 ? G1 : g;
                const uint32_t G2 = g + F2;
                g = ok2 ? G2 : g;
                const uint32_t G3 = g + F3;
                // F1 at the end of a row: the next row's F1ex
                if (ok0) { const uint32_t r = __umulhi(t0, magicW); if (t0 - r * W == W - 1u) rowC[r] = F0; }
                if (ok1) { const uint32_t r = __umulhi(t1, magicW); if (t1 - r * W == W - 1u) rowC[r] = F1; }
                if (ok2) { const uint32_t r = __umulhi(t2, magicW); if (t2 - r * W == W - 1u) rowC[r] = F2; }
                if (ok3) { const uint32_t r = __umulhi(t3, magicW); if (t3 - r * W == W - 1u) rowC[r] = F3; }
                F0 = G0; F1 = G1; F2 = G2; F3 = G3;
            }
            if (ok0) ring[ringIdx(t0)] = F0;
            if (ok1) ring[ringIdx(t1)] = F1;
            if (ok2) ring[ringIdx(t2)] = F2;
            if (ok3) ring[ringIdx(t3)] = F3;
        }
        if (!plan.endBarrier) {
            GF_CSTAMP(tcC);
            if (it > 0u && !skipRows) finishRows(it - 1u);
        } else {
            __syncthreads();
            GF_CSTAMP(tcC);
            if (!skipRows) finishRows(it);
#ifdef GF_DEC_END_BARRIER2
            __syncthreads();
#endif
            // (no barrier behind the rows -- round 5: what they read from the ring is overwritten by the NEXT chunk's ring writes,
            // and those stand behind that chunk's scan barrier, which no wave passes before every wave has finished these rows)
        }
        GF_CSTAMP(tcD);
    }
#ifdef GF_DIAG
    if (stamps && tid == 0 && (diagFlags & 0x20000u)) { stamps[12] = tcA; stamps[13] = tcB; stamps[14] = tcC; stamps[15] = tcD; }
#endif
    if (!plan.endBarrier) {
        __syncthreads();
        if (!skipRows) finishRows(nIter - 1u);
    }
    __syncthreads();
    return S.chainEnd > nM32 ? GF_K_ERR_BOUNDS : GF_K_OK;       // last value truncated
}

// ---------------------------------------------------------------------------------------------------------------
// The byte path (round 4): phases 2 + 3 for a tile whose tree holds NO introducer (0x7f / 0x81) and no null code (0x80).
//
// Every M32 value of such a text is one byte (CodecM32.java:327-356: a value is longer only behind an introducer), so byte j of
// the Huffman output IS element j of the predictor's stream, sign-extended -- no start marks, no ranks, no redo of special
// slots, and the element of cell (r, c) sits at an address known in closed form.  The pre-pass tells (GF_TREE_HAS_*, from the
// leaves it walks anyway); all of a terrain batch's tiles qualify, a tile with a steep step in it takes m32_to_tile as before.
// With the addresses known the inverse predictors (PredictorModelDifferencing.java:145-167, PredictorModelLinear.java:66-101,
// PredictorModelTriangle.java:62-98) need no stream-order scan and no ring either.  They are evaluated ROW BY ROW: a lane owns
// four neighbouring columns (nC <= 256: a wave spans a row) -- two LDS words and a v_alignbyte give it its four residual bytes,
// a prefix over the four, a DPP scan over the lanes' sums, one 16-byte store -- and the waves own blocks of rows:
//   Differencing  v(r,c) = C(r) + SUM(j = 1..c) res(r,j)                                C(r) = column 0, a running sum from the seed
//   Linear        v(r,c) = C(r) + c d1(r) + SUM(k = 2..c) SUM(j = 2..k) res(r,j)        d1(r) = v(r,1) - v(r,0): a double prefix sum
//   Triangle      v(r,c) = C(r) + SUM(i = 0..r) SUM(j = 1..c) res(i,j)                  four accumulators running down the rows; what
//                 the blocks above a wave's own add up to comes from a pre-pass over the bytes (column sums per block as pairs
//                 of 16-bit fields, then one row-wise prefix sum per block), through LDS
// Stream layouts (the order the encoders emit, see gf_stream_cell): Differencing -- cell i is byte i - 1; Linear -- byte 0 is
// (0,1), bytes 2r - 1 / 2r are (r,0) / (r,1), the interior of row r starts at 2 nR - 1 + r (nC - 2); Triangle -- row 0, then
// column 0, then the interior of row r >= 1 at nC + nR - 2 + (r - 1)(nC - 1).
// Scratch (over the dead decode tables): C(r), d1(r), the block sums.
constexpr uint32_t BYTE_MAX_COLS = 256;
__device__ __forceinline__ bool byte_path_eligible(int model, uint32_t nR, uint32_t nC, uint32_t ldsM32Bytes)
{
    const uint32_t RB = (nR + DEC_WAVES - 1u) / DEC_WAVES;
    // (a lane's four bytes may end three bytes behind the stream, and the word behind them is read too)
    return model >= 1 && model <= 3 && nR >= 2u && nC >= 4u && nC <= BYTE_MAX_COLS && RB <= 255u && nR * nC + 8u <= ldsM32Bytes &&
           ((2u * nR + 3u) & ~3u) + (model == 3 ? (uint32_t)DEC_WAVES * ((nC + 3u) & ~3u) : 0u) <= SCR_WORDS;
}

template <int MODEL>
__device__ __forceinline__ void m32_bytes_rows(DecShared &S, const uint8_t *m32, const uint32_t seed, const uint32_t nR, const uint32_t nC,
                                               uint32_t *__restrict__ o)
{
    const uint32_t lane = threadIdx.x & 63u, wave = gf_wave_id();
    uint32_t *scr = reinterpret_cast<uint32_t *>(&S);
    uint32_t *col0 = scr, *d1s = scr + nR, *T = scr + ((2u * nR + 3u) & ~3u);
    const uint32_t TS = (nC + 3u) & ~3u;
    const uint32_t *m32w = reinterpret_cast<const uint32_t *>(m32);
    const uint32_t j0 = 4u * lane;                                // the lane's columns: j0 .. j0 + 3
    const uint32_t jf = j0 < nC ? j0 : 0u;                        // (lanes without a column read lane 0's bytes: nothing behind the stream)
    auto sx = [](uint32_t b) -> uint32_t { return (uint32_t)(int32_t)(int8_t)b; };
    // ---- per-row constants: C(r), a running sum of the column-0 residuals, by the last wave (it has no pre-pass to do below),
    // d1(r) by the one before it
    if (wave == (uint32_t)DEC_WAVES - 1u) {
        uint32_t carry = seed;
        for (uint32_t rb = 0; rb < nR; rb += 64u) {
            const uint32_t r = rb + lane;
            uint32_t x = 0;
            if (r >= 1u && r < nR) x = sx(m32[MODEL == 3 ? nC - 2u + r : MODEL == 1 ? r * nC - 1u : 2u * r - 1u]);
            const uint32_t v = gf_wave_incl_scan(x) + carry;
            if (r < nR) col0[r] = v;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
        }
    } else if (MODEL == 2 && wave == (uint32_t)DEC_WAVES - 2u) {
        for (uint32_t r = lane; r < nR; r += 64u) d1s[r] = sx(m32[2u * r]);
    }
    // the rows of this wave, and where the bytes of a row lie: byte rowAt(r) + c belongs to cell (r, c) from column 1 on (Linear: 2)
    const uint32_t RB = (nR + DEC_WAVES - 1u) / DEC_WAVES;
    const uint32_t r0 = min(nR, wave * RB), r1 = min(nR, r0 + RB);
    const uint32_t rowStep = MODEL == 1 ? nC : MODEL == 2 ? nC - 2u : nC - 1u;
    auto rowAt = [&](uint32_t r) -> uint32_t {                    // (as if column 0 had a byte of its own; may be "-1" for row 0)
        if (MODEL == 1) return r * nC - 1u;
        if (MODEL == 2) return 2u * nR - 1u + r * (nC - 2u) - 2u;
        return r ? nC + nR - 2u + (r - 1u) * (nC - 1u) - 1u : 0u - 1u;
    };
    // the four bytes at rowAt + j0 .. + 3 as one word: a4 = rowAt + j0 + 4 (so that it is never negative)
    auto fetch = [&](uint32_t a4) -> uint32_t {
        const uint32_t wi = a4 >> 2;
        return __builtin_amdgcn_alignbyte(m32w[wi], m32w[wi - 1u], a4 & 3u);
    };
    // row 0 of Differencing / Triangle starts at byte "-1": lane 0's first word does not exist (its byte would be column 0)
    auto fetchRow0 = [&]() -> uint32_t { return __builtin_amdgcn_alignbyte(m32w[lane], lane ? m32w[lane - 1u] : 0u, 3u); };
    const bool peel = MODEL != 2 && r0 == 0u && r1 > 0u;         // (wave-uniform: wave 0)
    const uint32_t keep0 = lane == 0u ? 0u : 0xFFFFFFFFu;        // lane 0: the residuals of column 0 (Linear: and 1) count as zero
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;              // Triangle: SUM over the rows so far of the row prefixes
    if (MODEL == 3) {
        // ---- pre-pass: T[b][j] = SUM(rows r of block b) SUM(m = 1..j) res(r,m); the last block's is not needed
        if (wave + 1u < (uint32_t)DEC_WAVES && r1 > r0) {
            uint32_t ev = 0, od = 0;                              // bytes + 128 of columns j0, j0 + 2 / j0 + 1, j0 + 3 as 16-bit fields
            uint32_t r = r0;
            if (peel) {
                const uint32_t y = fetchRow0() ^ 0x80808080u;
                ev = y & 0x00FF00FFu;
                od = (y >> 8) & 0x00FF00FFu;
                r = 1u;
            }
            uint32_t a4 = rowAt(r) + jf + 4u;
            for (; r < r1; r++) {
                const uint32_t y = fetch(a4) ^ 0x80808080u;
                ev += y & 0x00FF00FFu;
                od += (y >> 8) & 0x00FF00FFu;
                a4 += rowStep;
            }
            const uint32_t bias = 128u * (r1 - r0);
            const uint32_t c0 = ((ev & 0xFFFFu) - bias) & keep0, c1 = (od & 0xFFFFu) - bias, c2 = (ev >> 16) - bias, c3 = (od >> 16) - bias;
            const uint32_t p1 = c0 + c1, p2 = p1 + c2, p3 = p2 + c3;
            const uint32_t ex = gf_wave_incl_scan(p3) - p3;
            if (j0 < TS) {
                uint32_t *Tw = T + wave * TS + j0;
                Tw[0] = ex + c0; Tw[1] = ex + p1; Tw[2] = ex + p2; Tw[3] = ex + p3;
            }
        }
    }
    __syncthreads();                                              // C(r), d1(r), T are in place
    if (MODEL == 3 && j0 < TS) {
        for (uint32_t w = 0; w < wave; w++) {
            const uint32_t *Tw = T + w * TS + j0;
            acc0 += Tw[0]; acc1 += Tw[1]; acc2 += Tw[2]; acc3 += Tw[3];
        }
    }
    // ---- the rows
    const uint32_t nFull = nC >> 2, rem = nC & 3u;               // lanes that store four cells; cells of the lane behind them
    const bool full = lane < nFull, part = lane == nFull && rem != 0u;
    struct Quad { uint32_t q0, q1, q2, q3; };
    auto sums = [&](uint32_t x, uint32_t left, uint32_t d1) -> Quad {
        const uint32_t v0 = sx(x) & keep0, v1 = MODEL == 2 ? sx(x >> 8) & keep0 : sx(x >> 8), v2 = sx(x >> 16), v3 = (uint32_t)((int32_t)x >> 24);
        const uint32_t p1 = v0 + v1, p2 = p1 + v2, p3 = p2 + v3;
        const uint32_t ex = gf_wave_incl_scan(p3) - p3;
        Quad q;
        if (MODEL == 2) {
            // the second prefix sum: within the lane, then over the lanes (a lane before adds its own and four times what lay before it)
            const uint32_t g1 = v0 + p1, g2 = g1 + p2, g3 = g2 + p3;
            const uint32_t t = g3 + 4u * ex;
            const uint32_t gx = gf_wave_incl_scan(t) - t;
            const uint32_t b = left + j0 * d1 + gx + ex;
            q.q0 = b + v0;
            q.q1 = b + d1 + ex + g1;
            q.q2 = b + 2u * (d1 + ex) + g2;
            q.q3 = b + 3u * (d1 + ex) + g3;
        } else if (MODEL == 3) {
            acc0 += ex + v0; acc1 += ex + p1; acc2 += ex + p2; acc3 += ex + p3;
            q.q0 = acc0 + left; q.q1 = acc1 + left; q.q2 = acc2 + left; q.q3 = acc3 + left;
        } else {
            const uint32_t b = left + ex;
            q.q0 = b + v0; q.q1 = b + p1; q.q2 = b + p2; q.q3 = b + p3;
        }
        return q;
    };
    auto put = [&](uint32_t r, const Quad &q) {
        uint32_t *dst = o + r * nC + j0;
        if (full) {
            GfU4 v;
            v.x = q.q0; v.y = q.q1; v.z = q.q2; v.w = q.q3;
            *reinterpret_cast<GfU4 *>(dst) = v;
        }
        if (part) {
            dst[0] = q.q0;
            if (rem >= 2u) dst[1] = q.q1;
            if (rem == 3u) dst[2] = q.q2;
        }
    };
    uint32_t r = r0;
    if (peel) {
        put(0u, sums(fetchRow0(), col0[0], 0u));
        r = 1u;
    }
    if (r < r1) {
        // two rows per turn -- their scans are independent chains: one fills the wait states the DPP steps of the other need -- and
        // the words of the next two are asked for before these are summed (behind the block's last row: that row again)
        const uint32_t last = r1 - 1u;
        uint32_t a4 = rowAt(r) + jf + 4u, rB = min(r + 1u, last);
        uint32_t a4B = a4 + (rB - r) * rowStep;
        uint32_t xA = fetch(a4), leftA = col0[r], dA = MODEL == 2 ? d1s[r] : 0u;
        uint32_t xB = fetch(a4B), leftB = col0[rB], dB = MODEL == 2 ? d1s[rB] : 0u;
        for (; r < r1; r += 2u) {
            const uint32_t x0 = xA, l0 = leftA, e0 = dA, x1 = xB, l1 = leftB, e1 = dB;
            const uint32_t nA = min(r + 2u, last), nB = min(r + 3u, last);
            a4 += (nA - r) * rowStep;
            a4B = a4 + (nB - nA) * rowStep;
            xA = fetch(a4);
            leftA = col0[nA];
            xB = fetch(a4B);
            leftB = col0[nB];
            if (MODEL == 2) { dA = d1s[nA]; dB = d1s[nB]; }
            const Quad qa = sums(x0, l0, e0);
            const Quad qb = sums(x1, l1, e1);                     // (of the last row once more where the block has no row r + 1)
            put(r, qa);
            if (r + 1u < r1) put(r + 1u, qb);
        }
    }
    __syncthreads();
}

template <class M32Ptr>
__device__ __forceinline__ void m32_bytes_to_tile(DecShared &S, M32Ptr m32, const int model, const uint32_t seed, const uint32_t nR,
                                                  const uint32_t nC, uint32_t *__restrict__ o)
{
    if (model == 3) m32_bytes_rows<3>(S, m32, seed, nR, nC, o);
    else if (model == 1) m32_bytes_rows<1>(S, m32, seed, nR, nC, o);
    else m32_bytes_rows<2>(S, m32, seed, nR, nC, o);
}

// The Huffman text of a tile whose serialised tree is INCOMPLETE (DecShared::skipLen), decoded the way the reference walks
// it (HuffmanDecoder.decode :179-185 over the node table of decodeTree :87-120): a step onto a child that was never read
// finds 0 in the table, which is the root's own slot, so the walk starts over at the root WITHOUT a symbol.  The missing
// children are the right children of the branches on the last leaf's path that were left by their left child: the codes
// (path prefix, then 1 where the path has 0).  No encoder writes such a tree; this is the decode of a damaged packing, done
// by one wave, a symbol at a time: the lanes hold the leaves (four each) and every step of the walk is one ballot.
// Ends with a barrier; returns the status (same in all threads); S.chainEnd = bit after the last symbol.
template <class M32Ptr>
__device__ int32_t huffman_serial_skips(DecShared &S, const uint32_t *__restrict__ base32, uint32_t nW, uint32_t sh0, uint32_t textStart,
                                        uint32_t endBit, uint32_t nM32, M32Ptr m32)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < 64u) {
        const uint32_t nLeaves = S.nLeaves;
        unsigned long long lc[4];
        uint32_t ll[4], ls[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t i = lane + 64u * j;
            lc[j] = i < nLeaves ? S.leafCode[i] : 0ull;
            ll[j] = i < nLeaves ? (uint32_t)S.leafLen[i] : 0xFFu;
            ls[j] = i < nLeaves ? (uint32_t)S.leafSym[i] : 0u;
        }
        const uint32_t skipLen = GF_UNI(S.skipLen);
        const unsigned long long skipPath = ((unsigned long long)GF_UNI(S.skipHi) << 32) | GF_UNI(S.skipLo);
        uint32_t pos = textStart, k = 0;
        int32_t st = GF_K_OK;
        while (k < nM32) {
            unsigned long long cur = 0;                              // path so far, first step in bit 0
            uint32_t len = 0;
            bool emitted = false, restart = false;
            while (!emitted && !restart) {
                if (pos >= endBit || len >= 64u) { st = GF_K_ERR_BOUNDS; break; }   // the bits end inside a code
                const uint32_t g = pos + sh0, wi = g >> 5;
                const uint32_t w = wi < nW ? GF_UNI(base32[wi]) : 0u;
                const uint32_t b = (w >> (g & 31u)) & 1u;
                pos++;
                cur |= (unsigned long long)b << len;
                len++;
                bool hit = false;
                uint32_t sym = 0;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++)
                    if (ll[j] == len && lc[j] == cur) { hit = true; sym = ls[j]; }
                const unsigned long long m = __ballot(hit);
                if (m) {
                    const uint32_t who = (uint32_t)__builtin_ctzll(m);
                    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)sym, (int)who);
                    if (lane == 0u) m32[k] = (uint8_t)v;
                    k++;
                    emitted = true;
                } else if (len <= skipLen && b == 1u && ((skipPath >> (len - 1u)) & 1ull) == 0ull &&
                           ((cur ^ skipPath) & ((1ull << (len - 1u)) - 1ull)) == 0ull) {
                    restart = true;                                  // a child that was never read: back to the root
                }
            }
            if (st != GF_K_OK) break;
        }
        if (lane == 0u) {
            S.chainEnd = pos;
            S.chainTotal = k;
            S.parseStatus = st;
        }
    }
    __syncthreads();
    const int32_t st = S.parseStatus;
    __syncthreads();
    return st;
}

// Diagnostics (cycle stamps per phase, phase ablation, warm-up sweep) exist only in the -DGF_DIAG build that tools/ use
// (gridfour_amd/build.py: libgvrs_hip_diag.so); the shipping kernels carry none of it.
#ifdef GF_DIAG
#define GF_DSTAMP(i)                                                                   \
    do {                                                                               \
        if (a.debug && tid == 0) (a.debug + t * 16)[i] = (uint32_t)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define GF_DPHASE_LIMIT(n, what) \
    if ((a.phaseLimit & 0xff) == (n)) what
#else
#define GF_DSTAMP(i) do { } while (0)
#define GF_DPHASE_LIMIT(n, what)
#endif

constexpr int GF_K_RETRY = 0x7fff0002;          // internal: the fast kernel leaves this tile to the general one

// One kernel body, three instantiations:
//   DEC_FAST     what a CodecHuffman batch consists of: tree records from the pre-pass, M32 stream in LDS, text read from
//                global memory, fused value decode + predictor inverse.  Anything else (stream larger than its LDS buffer,
//                nulls predictor, tile shapes the fused stage does not take) is marked GF_K_RETRY and left to
//   DEC_GENERAL  every container the codec accepts (CodecDeflate's raw M32 bytes, trees parsed in the kernel, spill
//                workspace, the unfused value pass + in-place inverse).  With a.retryFlag it touches only the tiles the fast
//                kernel marked, and returns at once when there are none.
//   DEC_ANALYZE  CodecHuffman.analyze: statistics instead of values.
// The general body is several times the size of the fast one; run for every tile it kept the instruction cache of a CU
// (shared by the workgroups of different phases) missing.
//   DEC_FAST_ROOMY  (round 5) the fast body once more for the tiles the pre-pass marked GF_TREE_ROOMY (M32 stream or packing
//                beyond the usual LDS buffer: tiles dense in multi-byte values), launched with LDS for two bytes per cell BESIDE
//                the fast run, on the other stream.  A few persistent workgroups that draw their tiles from the pre-pass's list
//                (GfDecodeArgs::roomyList, an atomic cursor): a tile-indexed grid would need a 55 KB hole in some CU's LDS for
//                every one of its thousands of workgroups, empty or not, while the fast run refills every 40 KB hole at once --
//                measured, such a run crawled beside the fast one and ended after it.  An instantiation of its own: the loop
//                around the body costs the one-tile-per-workgroup form 5-25 % (round 4).
//   DEC_FAST_CANON  (round 5) the fast body for CodecCanonHuffman packings (CodecCanonHuffman.java:163-195) whose code has no
//                escape, null or spare symbol: such a text is a prefix-coded string of the bytes value + 128 and an end-of-text
//                symbol (CanonicalHuffman.java:441-519), i.e. what the fast body decodes anyway -- one decode of the text into the
//                symbol pool, the byte path behind it -- once the 260 code lengths of the pre-pass (k_canon_parse_lengths) have been
//                turned into leaf records (canonical order IS pre-order).  The end-of-text symbol takes a byte value no symbol of the
//                tile uses, is asked for as value number nStream + 1 and looked for among the others afterwards.  A tile this run cannot
//                take, or finds anything unusual in, is left to k_canon_decode (GF_K_RETRY), which owns every status.
enum { DEC_GENERAL = 0, DEC_ANALYZE = 1, DEC_FAST = 2, DEC_FAST_ROOMY = 3, DEC_FAST_CANON = 4 };

template <int MODE>
__global__ __launch_bounds__(DEC_THREADS, MODE >= 2 ? GF_DEC_WGS : GF_DEC_WGS_GENERAL) void k_huffman_decode(GfDecodeArgs a)
{
    __shared__ DecShared S;
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsDyn[];
    constexpr bool ANALYZE = MODE == DEC_ANALYZE;
    constexpr bool ROOMY = MODE == DEC_FAST_ROOMY;
    constexpr bool CANON = MODE == DEC_FAST_CANON;
    constexpr bool FAST = MODE == DEC_FAST || ROOMY || CANON;
    constexpr int OWNER = MODE;                                       // (the Huffman passes: one copy per kernel, so that each inlines its own)

    const int tid = threadIdx.x, wave = (int)gf_wave_id();
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const uint32_t *__restrict__ w32 = reinterpret_cast<const uint32_t *>(a.blob);
    const uint64_t nWords = (a.blobBytes + 3) >> 2;
    if constexpr (MODE == DEC_GENERAL) {
        // (the last kernel of a CodecHuffman batch: the roomy list's count and cursor are zero again for the next pre-pass)
        if (a.retryFlag && a.ldsM32Roomy && blockIdx.x == 0 && tid == 0) {
            if (a.roomySeenHost) *a.roomySeenHost = 1u + a.retryFlag[2];
            a.retryFlag[2] = 0u;
            a.retryFlag[3] = 0u;
        }
        if (a.retryFlag && a.retryFlag[a.ldsM32Roomy ? 1 : 0] == 0u) return;   // the fast kernel decoded every tile
    }
    // the roomy run's tiles: entry i of the pre-pass's list, i drawn from the cursor (all threads call; ~0 = the list is exhausted)
    // (the draw travels through a word of S that is free between tiles -- a __shared__ word of its own would be LDS of EVERY
    // instantiation, and the 512-thread build's 120x150 workgroup sits exactly on its 40 KB step: four workgroups per CU, or three)
    auto roomy_next = [&]() -> size_t {
        __syncthreads();                                             // (the tile before is finished: S and the dynamic LDS are free)
        if (tid == 0) S.carry = atomicAdd(a.retryFlag + 3, 1u);
        __syncthreads();
        const uint32_t i = S.carry;
        return i < min(a.retryFlag[2], (uint32_t)a.nTiles) ? (size_t)a.roomyList[i] : ~(size_t)0;
    };

    for (size_t t = ROOMY ? roomy_next() : (size_t)blockIdx.x + (size_t)blockIdx.y * gridDim.x; t < a.nTiles;
         t = ROOMY ? roomy_next() : (MODE == DEC_FAST || CANON) ? ~(size_t)0 : t + gridDim.x) {
                                                                      // the fast kernel: one tile per workgroup, no loop (it never
                                                                      // touches the per-workgroup workspace; a second run that walks
                                                                      // the tiles with a small grid was measured in round 4: the
                                                                      // loop costs the first run 5 % at 120x150 and 25 % at 200x200)
        if constexpr (MODE == DEC_GENERAL) {
            if (a.retryFlag && a.status[t] != GF_K_RETRY) continue;
        }
        const uint64_t off = a.offsets ? a.offsets[t] : (uint64_t)t * a.slotStride;
        const uint32_t len = a.lengths[t];
        uint32_t *o = reinterpret_cast<uint32_t *>(a.values) + t * (size_t)nCells;
        const uint8_t *__restrict__ pk = a.blob + off;
        // The fast kernel's leaf records (round 4): where they lie depends on the tile's index alone, so they are asked for HERE, with
        // the packing's offset and length, and arrive while the header is read -- the tile's start was four dependent round trips
        // to memory (offset / length, header, record scalars, leaf arrays), now two.
#ifdef GF_DEC_NO_PREFETCH                                          // (experiment builds: tools/ab.sh)
        constexpr bool PRE = CANON;
#else
        constexpr bool PRE = FAST;
#endif
        unsigned long long preCode = 0;
        uint32_t preLen = 0, preSym = 0, preRec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // (and the header's twelve bytes, before the packing's length is known, where the blob has that many behind the offset)
        uint32_t preHead = 0;
        const bool headEarly = PRE && off + 12u <= a.blobBytes;
        if constexpr (PRE && CANON) {
            // (the pre-pass's record of a canonical packing: status, bit position of the text, then the 260 (+1) code lengths, a byte each;
            // thread s takes the length of symbol s, the first sixteen threads those of 256 .. 271 as well)
            if (headEarly && tid < 12) preHead = pk[tid];
            const uint32_t *recP = a.trees + t * GF_CANON_REC_WORDS;
            preRec[0] = recP[0];
            preRec[1] = recP[1];
            if (tid < 256) preLen = reinterpret_cast<const uint8_t *>(recP + 8)[tid];
            if (tid < 16) preSym = reinterpret_cast<const uint8_t *>(recP + 8)[256 + tid];
        } else if constexpr (PRE) {
            if (headEarly && tid < 12) preHead = pk[tid];
            const uint32_t *recP = a.trees + t * GF_TREE_REC_WORDS;
#pragma unroll
            for (int i = 0; i < 8; i++) preRec[i] = recP[i];              // (the same words in every thread: scalar loads)
            if (tid < 256) {
                preCode = reinterpret_cast<const unsigned long long *>(recP + 8)[tid];
                preLen = reinterpret_cast<const uint8_t *>(recP + 8 + 512)[tid];
                preSym = reinterpret_cast<const uint8_t *>(recP + 8 + 512 + 64)[tid];
            }
        }

        // (the canonical run: whatever it does not decode itself goes to k_canon_decode, which owns the statuses)
        auto canon_leave = [&]() {
            if (tid == 0) {
                a.status[t] = GF_K_RETRY;
                atomicOr(a.retryFlag, 1u);
            }
            __syncthreads();
        };
        if constexpr (CANON) {
            if (len < 12 || off + len > a.blobBytes) {
                canon_leave();
                continue;
            }
        }
        if (len < 10 || off + len > a.blobBytes) {       // BitInputStore would run out / AIOOBE on the header
            if (tid == 0) a.status[t] = GF_K_ERR_BOUNDS;
            __syncthreads();
            continue;
        }

        GF_DSTAMP(0);
        // The packing's text on its way into LDS (round 5): the fast Huffman pass reads it from a copy in the M32 buffer, which is idle
        // until then -- the first EARLY_TXT words per thread are asked for here, next to the leaf records, and arrive while the
        // header is read, instead of behind the lookup tables with the whole workgroup waiting for them (huffman_to_m32_fast
        // stages what is left).  A packing that turns out not to fit its buffer is marked below as before.
        if constexpr (FAST && EARLY_TXT > 0) {
            const uint64_t baseWord = (off * 8ull) >> 5;
            const uint32_t pkWords = (((uint32_t)(off * 8ull) & 31u) + len * 8u + 31u) >> 5;
            const uint32_t avail = (uint32_t)min((uint64_t)0xffffffffu, nWords - baseWord);
            const uint32_t cap = min(pkWords + FAST_TEXT_PAD, a.ldsM32Bytes >> 2);
            uint32_t *txt = reinterpret_cast<uint32_t *>(ldsDyn);
            uint32_t w[EARLY_TXT > 0 ? EARLY_TXT : 1];
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)EARLY_TXT; j++) {
                const uint32_t i = (uint32_t)tid + j * DEC_THREADS;
                w[j] = (i < pkWords && i < avail) ? w32[baseWord + i] : 0u;
            }
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)EARLY_TXT; j++) {
                const uint32_t i = (uint32_t)tid + j * DEC_THREADS;
                if (i < cap) txt[i] = w[j];
            }
        }
        // ---------------- phase 0: header + tree ----------------
        {
            uint8_t *hb = reinterpret_cast<uint8_t *>(S.head);
            const uint32_t nh = min(len, (uint32_t)(HEAD_WORDS * 4));
            constexpr uint32_t nHead = FAST ? 12u : (uint32_t)(HEAD_WORDS * 4);      // the fast kernel needs the 10 header bytes only
            if (headEarly) {
                asm volatile("" : "+v"(preHead));
                if (tid < 12) hb[tid] = (uint32_t)tid < nh ? (uint8_t)preHead : (uint8_t)0;
            } else {
                for (uint32_t i = tid; i < nHead; i += DEC_THREADS) hb[i] = i < nh ? pk[i] : 0;
            }
        }
        if constexpr (PRE) {
            // (the compiler would move the record loads down to where their values are used -- behind the header's barrier)
            uint32_t lo = (uint32_t)preCode, hi = (uint32_t)(preCode >> 32);
            asm volatile("" : "+v"(lo), "+v"(hi), "+v"(preLen), "+v"(preSym));
            asm volatile("" : "+v"(preRec[0]), "+v"(preRec[1]), "+v"(preRec[2]), "+v"(preRec[3]), "+v"(preRec[4]), "+v"(preRec[5]),
                              "+v"(preRec[6]), "+v"(preRec[7]));
            preCode = ((unsigned long long)hi << 32) | lo;
        }
        __syncthreads();
        const uint8_t *hb = reinterpret_cast<const uint8_t *>(S.head);
        const int model = CANON ? (int)(int8_t)hb[1] : (int)hb[1];
        const uint32_t seed = (uint32_t)hb[2] | ((uint32_t)hb[3] << 8) | ((uint32_t)hb[4] << 16) | ((uint32_t)hb[5] << 24);
        if constexpr (CANON) {
            // (CodecCanonHuffman's header is six bytes; the predictors of the byte path only, on a shape it takes, behind a readable table)
            if (model < 1 || model > 3 || !byte_path_eligible(model, nR, nC, a.ldsM32Bytes) || preRec[0] != (uint32_t)GF_K_OK) {
                canon_leave();
                continue;
            }
        }
        const uint32_t nStream = gf_stream_len(model, nR, nC);
        // (the canonical text: the stream's values and the end-of-text symbol behind them)
        const uint32_t nM32 = CANON ? nStream + 1u : (uint32_t)hb[6] | ((uint32_t)hb[7] << 8) | ((uint32_t)hb[8] << 16) | ((uint32_t)hb[9] << 24);
        int32_t early = GF_K_OK;
        if (model < 1 || model > 4) early = GF_K_ERR_FORMAT;            // CodecHuffman.java:155-169
        else if ((int32_t)nM32 < 0) early = GF_K_ERR_BOUNDS;            // NegativeArraySizeException
        else if ((uint64_t)nM32 > 6ull * nCells) early = GF_K_ERR_FORMAT; // no encoder emits this
        else if ((model == 2 && nC < 2)) early = GF_K_ERR_BOUNDS;       // PredictorModelLinear.java:80 output[1]
        else if (nM32 < nStream) early = GF_K_ERR_BOUNDS;               // M32 reads run off codeM32s
        const FusedPlan plan = fused_plan(nR, nC, model);
        if constexpr (FAST && !CANON) {
            // (first run: a tile of the roomy run is not touched -- not even its status, which the other run writes meanwhile;
            // the one-tile-per-call path has no second run: the tile keeps GF_K_RETRY and the caller sees to it)
            const uint32_t rec0 = PRE ? preRec[0] : (a.trees + t * GF_TREE_REC_WORDS)[0];
            const uint32_t rec3 = PRE ? preRec[3] : (a.trees + t * GF_TREE_REC_WORDS)[3];
            if (!ROOMY && a.ldsM32Roomy && !a.noRoomyRun && !a.lean && rec0 == (uint32_t)GF_K_OK && (rec3 & GF_TREE_ROOMY)) {
                __syncthreads();
                continue;
            }
            if (early == GF_K_OK && (nM32 > a.ldsM32Bytes || !plan.ring)) {
                early = GF_K_RETRY;
                if (tid == 0) atomicOr(a.retryFlag + (a.ldsM32Roomy ? 1 : 0), 1u);
            }
        }
        if constexpr (CANON) {
            if (early != GF_K_OK || nM32 > a.ldsM32Bytes) {
                canon_leave();
                continue;
            }
        }
        if (early != GF_K_OK) {
            if (tid == 0) a.status[t] = early;
            __syncthreads();
            continue;
        }

        GF_DSTAMP(1);
        if (!FAST && a.rawM32) {
            if (tid == 0) { S.parseStatus = len < 10ull + nM32 ? GF_K_ERR_BOUNDS : GF_K_OK; S.uniformSym = -1; S.textStart = 80; S.skipLen = 0; }
        } else if (FAST || a.trees) {
            // the tree was walked by k_huffman_parse_trees: fetch the leaf records, then mark the first-level entries
            // whose codes continue in a second-level table (one per distinct LUT_BITS-bit prefix among the longer codes,
            // numbered in pre-order) and list the short leaves -- in parallel, a leaf per thread
            const uint32_t *rec = a.trees + t * GF_TREE_REC_WORDS;
            // (the fast kernel has words 0..7 already: R)
            auto R = [&](int i) -> uint32_t { return PRE ? preRec[i] : rec[i]; };
            uint32_t nLeaves = R(1);
            bool mine = (uint32_t)tid < nLeaves && R(0) == (uint32_t)GF_K_OK && (int32_t)R(4) < 0;
            unsigned long long code = 0;
            uint32_t clen = 0, lsym = 0;
            uint32_t canonMaxLen = 0;
            // the short codes' table (round 5): the lookup entry of the leaf that owns a pattern of five text bits, 0 where a longer code
            // starts -- written by the leaves themselves below, read by build_lut (S.qs: free until the Huffman pass)
            if (tid < 32) S.qs[tid] = 0u;
            if constexpr (CANON) {
                // Leaf records from the 260 code lengths (CanonHuffTreeDecoder.java:68-95: symbols sorted by length, then symbol;
                // consecutive codes, shifted when the length grows), by the whole workgroup: a symbol's place is the number of
                // symbols with a shorter code plus those of its own length before it -- ballots inside the wave, a table of
                // per-wave counts across them.  The end-of-text symbol (259) is the last of its length.  Scratch: the bitmap area
                // behind the M32 buffer (the second-level table's place, not built yet; the buffer itself holds the text already).
                uint32_t *cw = reinterpret_cast<uint32_t *>(ldsDyn + a.ldsM32Bytes);   // [4][16] counts of the four waves of symbols; 64..67 first
                                                                       // unused byte per wave; 80..95 lengths of 256..271; 96 / 112:
                                                                       // first code / first place of a length; 128..191 places taken
                                                                       // by the waves before; 192: longest code
                const uint32_t lane = (uint32_t)tid & 63u;
                const uint32_t L = tid < 256 ? preLen : 0u;
                uint32_t within = 0;
                if (tid < 256) {                                       // (whole waves)
                    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
                    for (uint32_t l = 1; l <= 15u; l++) {
                        const unsigned long long m = __ballot(L == l);
                        within = L == l ? (uint32_t)__popcll(m & lt) : within;
                        if (lane == 0u) cw[(uint32_t)wave * 16u + l] = (uint32_t)__popcll(m);
                    }
                    const unsigned long long m0 = __ballot(L == 0u);
                    if (lane == 0u) {
                        cw[(uint32_t)wave * 16u] = 0u;
                        cw[64u + (uint32_t)wave] = m0 ? (uint32_t)wave * 64u + (uint32_t)__builtin_ctzll(m0) : 0xFFFFu;
                    }
                }
                if (tid < 16) cw[80 + tid] = preSym;
                __syncthreads();
                if (tid < 64) {
                    const uint32_t l = lane & 15u;
                    const uint32_t c0 = cw[l], c1 = cw[16u + l], c2 = cw[32u + l], c3 = cw[48u + l];
                    const uint32_t Le = cw[83];
                    const uint32_t esc = cw[80] | cw[81] | cw[82] | cw[84];            // null, the two escapes, the spare symbol 260
                    const uint32_t tot = (lane >= 1u && lane < 16u) ? c0 + c1 + c2 + c3 + (l == Le ? 1u : 0u) : 0u;
                    const uint32_t incl = gf_wave_incl_scan(tot);
                    uint32_t first = 0, kraft = 0;
#pragma unroll
                    for (uint32_t j = 1; j <= 15u; j++) {
                        const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)tot, (int)j);
                        first += j < l ? tj << (l - j) : 0u;
                        kraft += tj << (15u - j);
                    }
                    if (lane < 16u) {
                        cw[96u + l] = first;
                        cw[112u + l] = incl - tot;
                        cw[128u + l] = 0u;
                        cw[144u + l] = c0;
                        cw[160u + l] = c0 + c1;
                        cw[176u + l] = c0 + c1 + c2;
                    }
                    const uint32_t nL = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
                    const uint32_t unused = min(min(cw[64], cw[65]), min(cw[66], cw[67]));
                    const unsigned long long used = __ballot(tot != 0u);
                    const uint32_t mx = used ? 63u - (uint32_t)__builtin_clzll(used) : 0u;
                    const bool ok = esc == 0u && Le >= 1u && Le <= 15u && nL >= 2u && nL <= 256u && kraft == (1u << 15) && unused != 0xFFFFu;
                    // the end-of-text symbol's leaf: behind the plain symbols of its length
                    const uint32_t totE = (uint32_t)__builtin_amdgcn_readlane((int)tot, (int)(Le & 15u));
                    const uint32_t offE = (uint32_t)__builtin_amdgcn_readlane((int)(incl - tot), (int)(Le & 15u));
                    const uint32_t firstE = (uint32_t)__builtin_amdgcn_readlane((int)first, (int)(Le & 15u));
                    if (lane == 0u) {
                        S.parseStatus = ok ? GF_K_OK : GF_K_RETRY;
                        S.nLeaves = nL;
                        S.skipLo = (unused ^ 0x80u) & 0xffu;                           // the byte that stands for the end of the text
                        cw[192] = mx;
                        if (ok) {
                            const uint32_t rk = offE + totE - 1u, cd = firstE + totE - 1u;
                            S.leafCode[rk] = (unsigned long long)(__brev(cd) >> (32u - Le));
                            S.leafLen[rk] = (uint8_t)Le;
                            S.leafSym[rk] = (uint8_t)(unused ^ 0x80u);
                        }
                    }
                }
                __syncthreads();
                const bool okAll = S.parseStatus == GF_K_OK;
                if (tid < 256 && L != 0u && okAll) {
                    const uint32_t rk = cw[112u + L] + cw[128u + (uint32_t)wave * 16u + L] + within;
                    const uint32_t cd = cw[96u + L] + (rk - cw[112u + L]);
                    S.leafCode[rk] = (unsigned long long)(__brev(cd) >> (32u - L));    // first bit of the stream in bit 0
                    S.leafLen[rk] = (uint8_t)L;
                    S.leafSym[rk] = (uint8_t)((uint32_t)tid ^ 0x80u);                  // symbol s is the value s - 128, as a byte
                }
                canonMaxLen = cw[192];
                __syncthreads();
                nLeaves = okAll ? S.nLeaves : 0u;
                mine = (uint32_t)tid < nLeaves;
                if (mine) {
                    code = S.leafCode[tid];
                    clen = S.leafLen[tid];
                    lsym = S.leafSym[tid];
                }
            } else if (mine) {
                code = PRE ? preCode : reinterpret_cast<const unsigned long long *>(rec + 8)[tid];
                clen = PRE ? preLen : reinterpret_cast<const uint8_t *>(rec + 8 + 512)[tid];
                lsym = (uint8_t)(PRE ? preSym : reinterpret_cast<const uint8_t *>(rec + 8 + 512 + 64)[tid]);
                S.leafCode[tid] = code;
                S.leafLen[tid] = (uint8_t)clen;
                S.leafSym[tid] = (uint8_t)lsym;
            }
            __syncthreads();
            const uint32_t prefix = (uint32_t)code & ((1u << LUT_BITS) - 1u);
            const bool deep = mine && clen > (uint32_t)LUT_BITS;
            bool opens = deep;
            if (deep && tid > 0 && S.leafLen[tid - 1] > LUT_BITS)
                opens = ((uint32_t)S.leafCode[tid - 1] & ((1u << LUT_BITS) - 1u)) != prefix;
            uint32_t nSub;
            const uint32_t subIdx = block_excl_scan(opens ? 1u : 0u, S.waveSum, &nSub);
            if (opens) S.lut[prefix] = 0x80000000u | ((uint32_t)tid << LUT_FIRST_SHIFT) | subIdx;
            // a leaf with a code of up to five bits owns 2^(5 - length) of the thirty-two five-bit patterns (a prefix code: no two
            // leaves claim the same one).  Round 4 listed these leaves -- a second workgroup scan -- and build_lut went through the
            // list leaf by leaf, four dependent LDS reads per turn with every thread waiting: ten to fifteen turns per terrain tile.
            if (mine && clen <= 5u) {
                const uint32_t e = lut_single(lsym, clen);
                for (uint32_t k = 0; k < (1u << (5u - clen)); k++) S.qs[((uint32_t)code & 31u) | (k << clen)] = e;
            }
            constexpr uint32_t nShort = SHORT5_READY;
            if (tid == 0) {
                const uint32_t maxLen = CANON ? canonMaxLen : R(3) & 0xffu;
                const uint32_t l2 = maxLen > LUT_BITS ? min((uint32_t)L2_MAX_BITS, maxLen - LUT_BITS) : 1u;
                if constexpr (CANON) {
                    // (S.parseStatus, S.nLeaves and S.skipLo -- the end-of-text byte -- are set above)
                    S.symKinds = 0u;
                    S.uniformSym = -1;
                    S.skipLen = 0u;
                    S.textStart = R(1);
                    // (every long code in a second-level table: the search behind them reads legacy tree records)
                    if (nSub > ((uint32_t)L2_ENTRIES >> l2)) S.parseStatus = GF_K_RETRY;
                } else {
                    S.symKinds = R(3) & ~0xffu;
                    S.uniformSym = (int32_t)R(4);
                    S.skipLen = R(5);
                    S.skipLo = R(6);
                    S.skipHi = R(7);
                    S.textStart = R(2);
                    S.parseStatus = (int32_t)R(0);
                    S.nLeaves = nLeaves;
                }
                S.nShort = nShort;
                S.l2bits = l2;
                S.nSub = min(nSub, (uint32_t)L2_ENTRIES >> l2);
                S.maxLen = maxLen;
            }
        } else if (wave == 0) {
            if constexpr (!FAST) parse_tree_wave(S, 80u, 80u, len * 8u);
        }
        __syncthreads();
        if (S.parseStatus != GF_K_OK) {
            if (tid == 0) {
                a.status[t] = S.parseStatus;
                if constexpr (CANON) atomicOr(a.retryFlag, 1u);
            }
            __syncthreads();
            continue;
        }
        if constexpr (FAST) {
            // the fast Huffman passes want the packing (plus padding) in the M32 buffer and codes of at most 32 bits
            const uint32_t pkWords = (((uint32_t)(off * 8ull) & 31u) + len * 8u + 31u) >> 5;
            if (S.maxLen > 32u || S.skipLen != 0u || (pkWords + FAST_TEXT_PAD) * 4u > a.ldsM32Bytes) {
                if (tid == 0) {
                    a.status[t] = GF_K_RETRY;
                    atomicOr(a.retryFlag + (a.ldsM32Roomy ? 1 : 0), 1u);
                }
                __syncthreads();
                continue;
            }
        }
        GF_DSTAMP(2);
        GF_DPHASE_LIMIT(1, continue);
#ifdef GF_DIAG
        const uint32_t warmBits = (a.phaseLimit >> 8) ? (uint32_t)(a.phaseLimit >> 8) : 128u;   // experiment hook
        uint32_t *const dbg = a.debug ? a.debug + t * 16 + 11 : nullptr;
#else
        constexpr uint32_t warmBits = 128u;                           // about 25 symbols
        constexpr uint32_t *dbg = nullptr;
#endif

        // dynamic LDS / spill layout: [M32 bytes][start bitmap][bitmap rank base]; the bitmap area holds the second-level
        // lookup table while phase 1 runs
        const size_t bmRaw = 2 * (((size_t)a.ldsM32Bytes >> 5) + 2) * 4;
        const size_t bmArea = bmRaw > 2 * L2_ENTRIES ? bmRaw : 2 * L2_ENTRIES;
        uint16_t *lut2 = reinterpret_cast<uint16_t *>(ldsDyn + a.ldsM32Bytes);
        // Phases 1 and 2 run on the M32 buffer, its start bitmap and the rank bases -- in LDS, or in the workspace for
        // tiles whose stream does not fit.  The body is instantiated once per memory space: with a pointer that may be
        // either, every access would be a flat_* instruction (slow even when it lands in LDS).
        bool fused = false;                                           // the predictor inverse ran inside phase 2
        auto phases12 = [&](auto inLds) -> int32_t {
            uint8_t *m32;
            uint32_t *bm, *wb;
            if constexpr (decltype(inLds)::value) {
                m32 = ldsDyn;
                bm = reinterpret_cast<uint32_t *>(ldsDyn + a.ldsM32Bytes);
                wb = bm + (a.ldsM32Bytes >> 5) + 1;
            } else {
                uint8_t *ws = a.workspace + (size_t)blockIdx.x * a.workspaceStride;
                m32 = ws;
                const size_t cap = ((size_t)6 * nCells + 31) & ~(size_t)31;
                bm = reinterpret_cast<uint32_t *>(ws + cap);
                wb = bm + (cap >> 5) + 1;
            }
            int32_t tileStatus = GF_K_OK;

            // ---------------- phase 1: Huffman text -> M32 bytes ----------------
            if (!FAST && a.rawM32) {
                // the M32 bytes lie behind the header (the host inflated a CodecDeflate packing, CodecDeflate.java:141-147)
                for (uint32_t i = tid; i < nM32; i += DEC_THREADS) m32[i] = pk[10 + i];
                __syncthreads();
            } else if (S.uniformSym >= 0) {
                const uint8_t sym = (uint8_t)S.uniformSym;
                for (uint32_t i = tid; i < nM32; i += DEC_THREADS) m32[i] = sym;
                __syncthreads();
            } else if (!FAST && S.skipLen != 0u) {
                // an incomplete tree (damaged input): the reference's walk, symbol by symbol, by one wave
                const uint64_t baseWord = (off * 8ull) >> 5;
                tileStatus = huffman_serial_skips(S, w32 + baseWord, (uint32_t)min((uint64_t)0xffffffffu, nWords - baseWord),
                                                  (uint32_t)(off * 8ull) & 31u, S.textStart, len * 8u, nM32, m32);
            } else {
                build_lut(S, lut2);
                GF_DSTAMP(3);
                GF_DPHASE_LIMIT(6, return (int32_t)GF_K_SKIP);            // (diagnostic: header + tree records + lookup tables)
                const uint32_t textStart = S.textStart, endBit = len * 8u;
                const uint64_t baseWord = (off * 8ull) >> 5;
                const uint32_t sh0 = (uint32_t)(off * 8ull) & 31u;
                const uint32_t pkWords = (sh0 + endBit + 31u) >> 5;          // words that hold the packing
                if constexpr (FAST) {
                    tileStatus = huffman_to_m32_fast<OWNER>(S, w32 + baseWord, (uint32_t)min((uint64_t)0xffffffffu, nWords - baseWord),
                                                           sh0, reinterpret_cast<uint32_t *>(ldsDyn), pkWords, lut2,
                                                           // (the canonical run never searches the leaves: codes of fifteen bits at most)
                                                           reinterpret_cast<const unsigned long long *>(CANON ? a.trees : a.trees + t * GF_TREE_REC_WORDS + 8),
#ifdef GF_DIAG
                                                           textStart, endBit, nM32, m32, dbg, warmBits, a.phaseLimit & 0xff,
#else
                                                           textStart, endBit, nM32, m32, dbg, warmBits, 0,
#endif
                                                           // the symbol pool of the single-decode form: the tile's own output area (not
                                                           // in the one-tile-per-call path, whose output lies in host memory); the spare
                                                           // byte: in the second-level table's area behind the stream, dead by then
#ifdef GF_DEC_NO_POOL                                      // (experiment builds: tools/ab.sh)
                                                           nullptr, 0u, 0u);
#else
                                                           a.lean ? nullptr : o, nCells * 4u, a.ldsM32Bytes + 4u);
#endif
                } else if (pkWords * 4u <= a.ldsTextBytes) {
                    // stage the packing in LDS: one coalesced pass, then every symbol waits on LDS only
                    uint32_t *txt = reinterpret_cast<uint32_t *>(ldsDyn + a.ldsM32Bytes + bmArea);
                    const uint64_t avail = nWords - baseWord;
                    for (uint32_t i = tid; i < pkWords; i += DEC_THREADS) txt[i] = i < avail ? w32[baseWord + i] : 0u;
                    __syncthreads();
                    HuffCursorT<const uint32_t *> cur;
                    cur.base32 = txt;
                    cur.nW = pkWords;
                    cur.sh0 = sh0;
                    cur.S = &S;
                    cur.lut2 = lut2;
                    tileStatus = huffman_to_m32<OWNER>(S, cur, textStart, endBit, nM32, m32, dbg, warmBits);
                } else {
                    HuffCursorT<const uint32_t *> cur;
                    cur.base32 = w32 + baseWord;
                    cur.nW = (uint32_t)min((uint64_t)0xffffffffu, nWords - baseWord);
                    cur.sh0 = sh0;
                    cur.S = &S;
                    cur.lut2 = lut2;
                    tileStatus = huffman_to_m32<OWNER>(S, cur, textStart, endBit, nM32, m32, dbg, warmBits);
                }
            }
            if (tileStatus != GF_K_OK) return tileStatus;
            GF_DSTAMP(5);
            GF_DPHASE_LIMIT(2, return (int32_t)GF_K_SKIP);
            if constexpr (ANALYZE) {
                // CodecHuffman.analyze (CodecHuffman.java:172-199): what CodecStats.addToCounts / addCountsForM32 consume
                uint32_t *hist = S.lut;                                  // the lookup table is no longer needed
                for (uint32_t i = tid; i < 256; i += DEC_THREADS) hist[i] = 0;
                __syncthreads();
                for (uint32_t i = tid; i < nM32; i += DEC_THREADS) atomicAdd(&hist[m32[i]], 1u);
                if (a.pairCounts && nM32 >= 2u) {                        // sB[(prior << 8) | value]++ (CodecStats.java:150-156)
                    uint32_t *pc = a.pairCounts + (size_t)(model >= 0 && model < GF_PAIR_TABLES ? model : 0) * 65536u;
                    for (uint32_t i = tid + 1u; i < nM32; i += DEC_THREADS) atomicAdd(&pc[((uint32_t)m32[i - 1u] << 8) | m32[i]], 1u);
                }
                __syncthreads();
                uint32_t *rec = a.analysis + t * GF_ANALYSIS_WORDS;
                if (tid == 0) {
                    rec[0] = (uint32_t)model;
                    rec[1] = nM32;
                    rec[2] = S.textStart - 80u;                          // HuffmanDecoder.getBitsInTreeCount
                    rec[3] = len - 10u;
                    a.status[t] = GF_K_OK;
                }
                for (uint32_t i = tid; i < 256; i += DEC_THREADS) rec[4 + i] = hist[i];
                __syncthreads();
                return (int32_t)GF_K_SKIP;
            } else {
                // ---------------- phases 2 + 3 fused: M32 bytes -> values, one store per cell ----------------
                if constexpr (CANON && decltype(inLds)::value) {
                    // the text must be nStream values and THEN the end of the text (CanonicalHuffman.java:469-519 stops at that
                    // symbol wherever it stands, and a longer text overruns the reader's array): its byte -- one no value of this
                    // tile has -- nowhere before, and right behind them
                    const uint32_t eb = S.skipLo, eb4 = eb * 0x01010101u;
                    const uint32_t *mw = reinterpret_cast<const uint32_t *>(m32);
                    uint32_t bad = 0;
                    for (uint32_t i = tid; 4u * i < nStream; i += DEC_THREADS) {
                        const uint32_t v = mw[i] ^ eb4;
                        uint32_t z = ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;   // 0x80 in every byte that is zero
                        const uint32_t left = nStream - 4u * i;
                        if (left < 4u) z &= (1u << (8u * left)) - 1u;
                        bad |= z;
                    }
                    if (tid == 0 && m32[nStream] != (uint8_t)eb) bad = 1u;
                    if (__syncthreads_or((int)(bad != 0u))) return (int32_t)GF_K_RETRY;
                    fused = true;
                    m32_bytes_to_tile(S, m32, model, seed, nR, nC, o);
                    return (int32_t)GF_K_OK;
                }
                if constexpr (FAST && !CANON && decltype(inLds)::value) {
                    // a tree of one-byte values only (no introducer, no null code among its leaves): the byte path
                    if (!(GF_UNI(S.symKinds) & (GF_TREE_HAS_INTRODUCER | GF_TREE_HAS_NULL)) && byte_path_eligible(model, nR, nC, a.ldsM32Bytes)) {
                        fused = true;
                        GF_DPHASE_LIMIT(4, return (int32_t)GF_K_OK);
                        m32_bytes_to_tile(S, m32, model, seed, nR, nC, o);
                        return (int32_t)GF_K_OK;
                    }
                }
                if constexpr (decltype(inLds)::value) {
                    if (FAST || (plan.ring && model >= 1 && model <= 3)) {
                        fused = true;
                        #ifdef GF_DIAG
                        return m32_to_tile(S, m32, nM32, bm, wb, model, seed, nR, nC, nStream, plan, o, dbg ? dbg - 11 : nullptr, (uint32_t)a.phaseLimit);
#else
                        return m32_to_tile(S, m32, nM32, bm, wb, model, seed, nR, nC, nStream, plan, o, nullptr, 0u);
#endif
                    }
                }
                if constexpr (!FAST) {
                    // ---------------- phase 2: M32 bytes -> residuals at their cells ----------------
                    const CellMapPredictor map{GfCellMap::make(model, nR, nC)};
                    tileStatus = m32_to_values(S, m32, nM32, bm, wb, nStream, map, o);
                }
                return tileStatus;
            }
        };
        int32_t tileStatus;
        if constexpr (FAST) tileStatus = phases12(std::true_type{});
        else tileStatus = nM32 <= a.ldsM32Bytes ? phases12(std::true_type{}) : phases12(std::false_type{});
        if (tileStatus == (int32_t)GF_K_SKIP) continue;
        if constexpr (CANON) {
            if (tileStatus != GF_K_OK) {                 // (a text that ends early, a last code cut off, ...: k_canon_decode says what it is)
                canon_leave();
                continue;
            }
        }
        if (tileStatus != GF_K_OK) {
            if (tid == 0) a.status[t] = tileStatus;
            __syncthreads();
            continue;
        }
        GF_DSTAMP(8);
        GF_DPHASE_LIMIT(3, continue);

        // ---------------- phase 3: predictor inverse (wrap-around prefix sums), where phase 2 did not include it ----------------
        if constexpr (!FAST) {
            if (!fused) gf_predictor_inverse(model, seed, o, nR, nC, nullptr);
        }
        GF_DSTAMP(10);
        if (tid == 0) a.status[t] = GF_K_OK;
        __syncthreads();
    }
}

#ifndef GF_DEC_VARIANT
// ---------------------------------------------------------------------------------------------------------------
// LsDecoder12.decode for the containers that carry CodecM32 bytes (lsop/LsDecoder12.java:107-150): header (either
// revision, lsop/LsHeader.java:131-185), then the initialiser and interior M32 streams -- type 0: two legacy Huffman
// segments back to back in one bit store, the second starting at the bit after the first one's last code
// (LsDecoder12.java:116-124); type 1: two zlib streams, inflated by the host before the launch (rawM32).  Output:
// seed + coefficients and the residual ints that k_lsop_reconstruct (gvrs_lsop.hip) turns into the tile.
__global__ __launch_bounds__(DEC_THREADS, 4) void k_lsop_unpack_m32(GfLsopM32Args a)
{
    __shared__ DecShared S;
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsDyn[];

    const int tid = threadIdx.x, wave = (int)gf_wave_id();
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const uint32_t nInit = 4u * nR + 2u * nC - 9u, nInt = (nR - 2u) * (nC - 4u);
    const uint32_t *__restrict__ w32 = reinterpret_cast<const uint32_t *>(a.blob);
    const uint64_t nWords = (a.blobBytes + 3) >> 2;

    for (size_t t = blockIdx.x; t < a.nTiles; t += gridDim.x) {
        if (a.status[t] != GF_K_ERR_UNSUPPORTED) continue;               // decoded (or rejected) by k_lsop_unpack2
        const uint64_t off = a.offsets ? a.offsets[t] : (uint64_t)t * a.slotStride;
        const uint32_t len = a.lengths[t];
        const uint8_t *__restrict__ pk = a.blob + off;
        uint32_t *res = reinterpret_cast<uint32_t *>(a.residuals) + t * a.resStride;
        auto le32 = [&](uint32_t o) -> uint32_t {
            return (uint32_t)pk[o] | ((uint32_t)pk[o + 1] << 8) | ((uint32_t)pk[o + 2] << 16) | ((uint32_t)pk[o + 3] << 24);
        };

        // header: codec index, [type | flags], nCoef, seed, 12 floats, the two M32 byte counts, [type | flags]
        int32_t early = GF_K_OK;
        uint32_t o = 1, type = 0, nMI = 0, nMX = 0;
        bool checksum = false;
        if (len < 3 || off + len > a.blobBytes) early = GF_K_ERR_BOUNDS;
        else {
            const bool revised = pk[1] & 0x40;
            if (revised) { type = pk[1] & 0x0fu; checksum = pk[1] & 0x80; o = 2; }
            if (len < o + 1u + 52u + 8u + (revised ? 0u : 1u)) early = GF_K_ERR_BOUNDS;
            else if (pk[o] != 12) early = GF_K_ERR_FORMAT;                // u[11] would index out of bounds
            else {
                o += 53;
                nMI = le32(o);
                nMX = le32(o + 4);
                o += 8;
                if (!revised) { type = pk[o] & 0x0fu; checksum = pk[o] & 0x80; o++; }
                if (checksum) o += 4;                                     // value checksum: skipped
                if (o > len) early = GF_K_ERR_BOUNDS;
                else if (type == 2) early = GF_K_ERR_UNSUPPORTED;         // canonical text behind a legacy header: never written
                else if (type != 0 && !a.rawM32) early = GF_K_ERR_UNSUPPORTED;   // Deflate: needs an inflate pass first
                else if (nMI > 6u * nInit + 64u || nMX > 6u * nInt + 64u) early = GF_K_ERR_FORMAT;   // no encoder emits this
                else if (type != 0 && a.rawM32 == 1 && (uint64_t)o + nMI + nMX > len) early = GF_K_ERR_BOUNDS;
                else if (type != 0 && a.rawM32 == 2) {                    // inflated on the device (k_lsop_streams, k_inflate)
                    const int32_t sd = a.sideStatus[t];
                    if (sd < 0) early = sd;
                    else if (sd != 0) early = GF_K_ERR_UNSUPPORTED;
                    else if (a.inflStatus2[t] != GF_K_OK || a.produced2[t] < nMX) early = GF_K_ERR_FORMAT;
                }
            }
        }
        if (early != GF_K_OK) {
            if (tid == 0) a.status[t] = early;
            __syncthreads();
            continue;
        }
        if (tid < 13) a.coefs[t * 16 + tid] = le32((pk[1] & 0x40 ? 3u : 2u) + 4u * tid);

        uint32_t startBit = o * 8u;                                       // type 0: where the next Huffman segment begins
        uint32_t rawAt = a.rawM32 == 2 ? 0u : o;                          // type 1: where the next M32 stream begins
        const uint8_t *__restrict__ rawFrom = a.rawM32 == 2 ? a.rawSide + t * a.rawSideStride : pk;
        int32_t tileStatus = GF_K_OK;
        for (int seg = 0; seg < 2 && tileStatus == GF_K_OK; seg++) {
            const uint32_t nM32 = seg ? nMX : nMI, nVals = seg ? nInt : nInit;
            uint32_t *out = res + (seg ? nInit : 0u);
            auto segment = [&](auto inLds) -> int32_t {
                uint8_t *m32;
                uint32_t *bm, *wb;
                if constexpr (decltype(inLds)::value) {
                    m32 = ldsDyn;
                    bm = reinterpret_cast<uint32_t *>(ldsDyn + a.ldsM32Bytes);
                    wb = bm + (a.ldsM32Bytes >> 5) + 1;
                } else {
                    uint8_t *ws = a.workspace + (size_t)blockIdx.x * a.workspaceStride;
                    m32 = ws;
                    const size_t cap = ((size_t)6 * nCells + 31) & ~(size_t)31;
                    bm = reinterpret_cast<uint32_t *>(ws + cap);
                    wb = bm + (cap >> 5) + 1;
                }
                if (type != 0) {
                    for (uint32_t i = tid; i < nM32; i += DEC_THREADS) m32[i] = rawFrom[rawAt + i];
                    __syncthreads();
                } else {
                    // the serialised tree: stage the words around it, parse, build the tables
                    const uint32_t hb0 = (startBit >> 5) << 2;
                    {
                        uint8_t *hb = reinterpret_cast<uint8_t *>(S.head);
                        for (uint32_t i = tid; i < HEAD_WORDS * 4; i += DEC_THREADS) hb[i] = hb0 + i < len ? pk[hb0 + i] : 0;
                    }
                    __syncthreads();
                    if (wave == 0) parse_tree_wave(S, startBit & 31u, startBit, len * 8u);
                    __syncthreads();
                    if (S.parseStatus != GF_K_OK) return S.parseStatus;
                    if (S.uniformSym >= 0) {
                        const uint8_t sym = (uint8_t)S.uniformSym;
                        for (uint32_t i = tid; i < nM32; i += DEC_THREADS) m32[i] = sym;
                        if (tid == 0) S.chainEnd = S.textStart;          // no text (HuffmanDecoder.java:170-177)
                        __syncthreads();
                    } else if (S.skipLen != 0u) {
                        // an incomplete tree (damaged input): the reference's walk, symbol by symbol (huffman_serial_skips)
                        const uint64_t baseWord = (off * 8ull) >> 5;
                        const int32_t st = huffman_serial_skips(S, w32 + baseWord, (uint32_t)min((uint64_t)0xffffffffu, nWords - baseWord),
                                                                (uint32_t)(off * 8ull) & 31u, S.textStart, len * 8u, nM32, m32);
                        if (st != GF_K_OK) return st;
                        if (nM32 == 0 && tid == 0) S.chainEnd = S.textStart;
                        __syncthreads();
                    } else {
                        // second-level table: always in LDS, in the area the in-LDS bitmap uses later (as in k_huffman_decode)
                        uint16_t *lut2 = reinterpret_cast<uint16_t *>(ldsDyn + a.ldsM32Bytes);
                        build_lut(S, lut2);
                        const uint32_t textStart = S.textStart;
                        // the segment ends no later than nM32 codes of the longest length beyond its start
                        const uint32_t endBit = (uint32_t)min((uint64_t)len * 8u, (uint64_t)textStart + (uint64_t)nM32 * S.maxLen);
                        const uint64_t baseWord = (off * 8ull) >> 5;
                        HuffCursorT<const uint32_t *> cur;
                        cur.base32 = w32 + baseWord;
                        cur.nW = (uint32_t)min((uint64_t)0xffffffffu, nWords - baseWord);
                        cur.sh0 = (uint32_t)(off * 8ull) & 31u;
                        cur.S = &S;
                        cur.lut2 = lut2;
                        const int32_t st = huffman_to_m32<3>(S, cur, textStart, endBit, nM32, m32, nullptr, 128u);
                        if (st != GF_K_OK) return st;
                        if (nM32 == 0 && tid == 0) S.chainEnd = textStart;
                        __syncthreads();
                    }
                }
                const uint32_t segEnd = S.chainEnd;
                __syncthreads();                                          // m32_to_values reuses chainEnd
                const int32_t st = m32_to_values(S, m32, nM32, bm, wb, nVals, CellMapIdentity{}, out);
                startBit = segEnd;
                rawAt += nM32;
                return st;
            };
            tileStatus = nM32 <= a.ldsM32Bytes ? segment(std::true_type{}) : segment(std::false_type{});
        }
        if (tid == 0) a.status[t] = tileStatus;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Tree pre-pass: HuffmanDecoder.decodeTree (HuffmanDecoder.java:65-161) for a whole batch, ONE LANE PER TILE.  The walk
// is serial per tree (about a hundred leaves, a hundred instructions each); as a scalar loop inside the decode kernel it
// kept three of a workgroup's four waves idle for a sixth of the tile time.  Here 64 trees advance per wave instruction
// and the decode kernel starts from the leaf records.  Same walk, same checks and statuses as parse_tree_wave.
template <unsigned perWave>                                 // lanes of a wave that walk a tree each: gf_prepass_tiles_per_wave (as
                                                            // a constant: with one lane the compiler makes the walk scalar code)
__global__ __launch_bounds__(64) void k_huffman_parse_trees(const uint8_t *__restrict__ blob, size_t blobBytes,
                                                            const uint64_t *__restrict__ offsets, size_t slotStride,
                                                            const uint32_t *__restrict__ lengths, uint32_t *__restrict__ trees,
                                                            size_t nTiles, uint32_t *__restrict__ clearFlags, uint32_t fastBytes,
                                                            uint32_t roomyBytes, uint32_t *__restrict__ roomyList)
{
    static_assert(perWave == 1u || perWave == 64u, "a wave walks one tree or sixty-four");
    const uint32_t lane = threadIdx.x;
    // the decode kernels' two retry words (GfDecodeArgs::retryFlag), cleared here instead of by a launch of their own
    if (clearFlags && blockIdx.x == 0 && lane < 2u) clearFlags[lane] = 0u;
    const size_t t = perWave == 1u ? (size_t)blockIdx.x : (size_t)blockIdx.x * perWave + threadIdx.x;
    const bool inBatch = t < nTiles;                          // (the last wave's spare lanes help with the staging below)
    const uint64_t off = !inBatch ? 0ull : offsets ? offsets[t] : (uint64_t)t * slotStride;
    const uint32_t len = inBatch ? lengths[t] : 0u;
    const bool readable = inBatch && len >= 10 && off + len <= blobBytes;
    uint32_t *rec = trees + (inBatch ? t : 0) * GF_TREE_REC_WORDS;
    // The words a walk can ask for -- bytes 10 .. of the packing, zero from its end and from the end of the head (HEAD_WORDS) on --
    // are fetched before the walk, all loads in flight at once.  A wave per tile (small batches, one tile per call; round 4): the 64
    // lanes fetch the tile's words, the walk reads lanes (every refill of its bit buffer used to be a load of its own, some thirty
    // dependent round trips, 30 us for the one tile of a call).  Sixty-four tiles per wave: the wave fetches tile after tile, a
    // lane a word (coalesced), into a table in LDS that the lanes then read down their own column.
    constexpr uint32_t STAGE_WORDS = (HEAD_WORDS * 4 - 10 + 3) / 4;             // 86
    __shared__ uint32_t stage[perWave == 64u ? (STAGE_WORDS + 1) * 64 : 1];
    // word k of a packing at pk0 with `vis` visible bytes: never a byte beyond them is touched (the load is moved back to end with
    // the last visible byte and its bytes are shifted into place)
    // (in three steps, so that the loads of a turn are all in flight together: where to read -- a word that is not visible is read at
    // byte 0 of the packing and dropped --, the loads, the bytes into place.  The compiler moves a load whose value is used under
    // a condition into a branch of its own and waits for it there: the empty asm below uses the loaded words unconditionally.)
    auto staged_at = [](uint32_t vis, uint32_t k) -> uint32_t {
        const uint32_t i = 10u + 4u * k;
        return i < vis ? min(i, vis - 4u) : 0u;               // vis >= 10
    };
    auto staged_fix = [](uint32_t w, uint32_t vis, uint32_t k) -> uint32_t {
        const uint32_t i = 10u + 4u * k;
        return i < vis ? w >> (8u * (i - min(i, vis - 4u))) : 0u;           // (a shift of three bytes at most)
    };
    uint32_t stage0 = 0, stage1 = 0;
    if (blobBytes < 10) {                                    // no packing fits (and the staging below reads the blob's first bytes)
        if (inBatch && (perWave == 64u || lane == 0u)) rec[0] = (uint32_t)GF_K_ERR_BOUNDS;
        return;
    }
    if (perWave == 1u) {
        if (readable) {
            const uint32_t vis = min(len, (uint32_t)(HEAD_WORDS * 4));
            const uint8_t *__restrict__ pk0 = blob + off;
            stage0 = reinterpret_cast<const PackedWord *>(pk0 + staged_at(vis, lane))->v;
            stage1 = reinterpret_cast<const PackedWord *>(pk0 + staged_at(vis, 64u + lane))->v;
            asm volatile("" : "+v"(stage0), "+v"(stage1));
            stage0 = staged_fix(stage0, vis, lane);
            stage1 = staged_fix(stage1, vis, 64u + lane);
        }
        // (the wave stays whole: the walk below reads the staged words out of the other lanes' registers, which is only defined
        // while those lanes are active.  Everything the walk computes is wave-uniform -- scalar code --; lane 0 alone stores.)
    } else {
        const uint32_t visMine = readable ? min(len, (uint32_t)(HEAD_WORDS * 4)) : 0u;
        // (thirty-two tiles a turn: sixty-four loads in flight, then their stores -- a turn per tile waited for every load on its own)
        constexpr uint32_t TURN = 32;
        for (uint32_t j0 = 0; j0 < 64u; j0 += TURN) {
            uint32_t w0[TURN], w1[TURN], visJ[TURN];
#pragma unroll
            for (uint32_t u = 0; u < TURN; u++) {
                const uint32_t j = j0 + u;
                const uint32_t vis = (uint32_t)__builtin_amdgcn_readlane((int)visMine, (int)j);
                const uint64_t offJ = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), (int)j) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, (int)j);
                const uint8_t *__restrict__ pkJ = blob + (vis ? offJ : 0ull);      // (a tile that cannot be read: the blob's first bytes, dropped)
                visJ[u] = vis;
                w0[u] = reinterpret_cast<const PackedWord *>(pkJ + staged_at(vis, lane))->v;
                w1[u] = reinterpret_cast<const PackedWord *>(pkJ + staged_at(vis, 64u + lane))->v;   // (words 86 .. 127: beyond the head)
            }
#pragma unroll
            for (uint32_t u = 0; u < TURN; u += 8u)
                asm volatile("" : "+v"(w0[u]), "+v"(w0[u + 1]), "+v"(w0[u + 2]), "+v"(w0[u + 3]), "+v"(w0[u + 4]), "+v"(w0[u + 5]),
                                  "+v"(w0[u + 6]), "+v"(w0[u + 7]), "+v"(w1[u]), "+v"(w1[u + 1]), "+v"(w1[u + 2]), "+v"(w1[u + 3]),
                                  "+v"(w1[u + 4]), "+v"(w1[u + 5]), "+v"(w1[u + 6]), "+v"(w1[u + 7]));
#pragma unroll
            for (uint32_t u = 0; u < TURN; u++) {
                stage[lane * 64u + j0 + u] = staged_fix(w0[u], visJ[u], lane);
                if (lane < STAGE_WORDS - 64u) stage[(64u + lane) * 64u + j0 + u] = staged_fix(w1[u], visJ[u], 64u + lane);
            }
        }
        stage[STAGE_WORDS * 64u + lane] = 0u;                  // the row of zeros behind the head
        __syncthreads();
        if (!inBatch) return;
    }
    unsigned long long *codes = reinterpret_cast<unsigned long long *>(rec + 8);
    uint8_t *lens = reinterpret_cast<uint8_t *>(rec + 8 + 512), *syms = lens + 256;
    const bool writer = perWave == 64u || lane == 0u;        // a wave per tile: every lane walks, lane 0 stores
    if (len < 10 || off + len > blobBytes) {                 // the decode kernel rejects the tile before looking here
        if (writer) rec[0] = (uint32_t)GF_K_ERR_BOUNDS;
        return;
    }
    const uint8_t *__restrict__ pk = blob + off;
    // the walk sees what parse_tree_wave sees: the packing's bytes, zero beyond its end and beyond the staged head
    const uint32_t visible = min(len, (uint32_t)(HEAD_WORDS * 4));
    // (the exact walk, for the trees the fast one below turns down, reads the packing itself)
    auto ld32 = [&](uint32_t i) -> uint32_t {                // bytes i .. i+3 of the packing, little-endian
        if (perWave == 1u) {                                 // (i = 10 + 4 k: the staged word k)
            const uint32_t k = (i - 10u) >> 2;
            if (k >= 128u) return 0u;                        // beyond the staged head (= beyond `visible`)
            // (both registers are read at the lane and the pick is made afterwards: a select between them BEFORE the read would
            // pick per lane, not per word)
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)stage0, (int)(k & 63u));
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)stage1, (int)(k & 63u));
            return k < 64u ? lo : hi;
        }
        if (i + 4u <= visible) return reinterpret_cast<const PackedWord *>(pk + i)->v;
        uint32_t w = 0;
        for (uint32_t k = 0; k < 4; k++)
            if (i + k < visible) w |= (uint32_t)pk[i + k] << (8u * k);
        return w;
    };
    auto fetch = [&](uint32_t k) -> uint32_t {               // staged word k = ld32(10 + 4 k)
        if (perWave == 1u) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)stage0, (int)(k & 63u));
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)stage1, (int)(k & 63u));
            return k < 64u ? lo : (k < 128u ? hi : 0u);
        }
        return stage[min(k, STAGE_WORDS) * 64u + lane];
    };
    uint64_t buf = ((uint64_t)fetch(1) << 32) | fetch(0);    // packing bit 80 = byte 10
    uint32_t have = 64, next = 18, bp = 80;
    auto refill = [&]() {
        if (have <= 32) {
            buf |= (uint64_t)ld32(next) << have;
            have += 32;
            next += 4;
        }
    };
    auto take = [&](uint32_t nb) -> uint32_t {               // nb <= 9, needs have >= nb
        const uint32_t v = (uint32_t)buf & ((1u << nb) - 1u);
        buf >>= nb;
        have -= nb;
        bp += nb;
        return v;
    };
    int32_t st = GF_K_OK;
    int32_t uniformSym = -1;
    const uint32_t nLeaves = take(8) + 1;
    const uint32_t rootBit = take(1);
    uint32_t maxLen = 1;
    uint32_t skipLen = 0;
    unsigned long long skipPath = 0;
    // which kinds of M32 bytes the text can hold at all (GF_TREE_HAS_*): a tree without the introducers 0x7f / 0x81 codes
    // one-byte values only, so byte j of the Huffman output IS stream element j (CodecM32.java:327-356)
    uint32_t symKinds = 0;
    auto kindOf = [](uint32_t sym) -> uint32_t {
        return (sym == 0x7fu || sym == 0x81u) ? GF_TREE_HAS_INTRODUCER : sym == 0x80u ? GF_TREE_HAS_NULL : 0u;
    };
    // The walk of a WELL-FORMED tree first (round 4): the steps of the exact walk below with nothing in them that only a damaged
    // packing needs (no record counts, one depth test), written without a branch -- a turn takes the run of branch records at the
    // cursor and the leaf behind it, both under predicates; the next word of the packing comes from the staged table a turn before
    // it is used; the loop ends when no lane of the wave has a leaf left.  About 95 instructions per leaf against 185 of the exact
    // form (the pre-pass is one wave per SIMD and bound by exactly that chain): 0.090 -> 0.049 ms per 12,960 tiles.  It accepts a tree
    // only if its last leaf closes it and none before does; anything else (a tree that closes early, stays open, grows deeper than
    // a code register) is walked again by the exact form, which owns every status and the incomplete-tree record.
    bool walked = false;
    // A wave per tile (one tile per call, small batches; round 5): everything the walk computes is wave-uniform, i.e. scalar code, and
    // the predicated turn above all serves sixty-four DIFFERENT trees per instruction -- for one tree it was 95 dependent scalar
    // instructions per leaf, 25 us of a one-tile decode call's 66 on the device (profiles/r05_v3/single_tile_timeline.txt).  Here the same
    // steps with plain branches; the leaf records collect in LDS and leave together, a lane a leaf.  Same acceptance rule.
    // (round 6: the walk was a hundred scalar instructions per leaf -- 21 us of a one-tile decode call, 37 us of the 165 us of a
    // 1,024-tile batch, where four such walks share a CU's scalar unit.  What it no longer does per leaf: it does not tell the kinds of
    // M32 bytes apart (the lanes do that over the finished records), it does not switch the wave's lanes off and on around three LDS
    // stores (every lane stores the same record, four words in one store), it takes its minima on the scalar unit.)
    __shared__ __attribute__((aligned(16))) uint32_t leafLds[perWave == 1u ? 4 * 256 : 4];
#ifndef GF_PT_NO_SCAN                                               // (experiment builds: the walk alone)
    if constexpr (perWave == 1u) {
        // THE WALK AS SCANS (round 6).  What is sequential in a serialised tree is less than the walk makes it:
        //   * where the records start -- a branch is one bit, a leaf nine -- is a transducer with nine states (bits still to skip): a
        //     lane takes forty bits, works out backwards, for each of the nine states it may be entered in, the state it is left in
        //     (E[p] = bit p ? E[p + 9] : E[p + 1], nine of them in a register), a scalar chain hands the states from lane to lane
        //     (two v_readlane and a shift per lane in use), and every lane then walks its own forty bits from its true first record;
        //   * which leaf is which, and how many branch records stand in front of each, are prefix sums over the lanes;
        //   * only the code lengths are a recurrence over the LEAVES (a leaf's depth is its predecessor's, minus the ones its path
        //     ends in, plus the branch records between them): some twenty scalar instructions per leaf where the walk has eighty,
        //     the records written from registers afterwards, a lane a leaf.
        // Accepted under the walk's own rule: as many leaves and branch records as a tree of nLeaves leaves has, closed by its last
        // leaf and not before, no code deeper than a register; anything else goes on to the walk below.
        const uint32_t T = 10u * nLeaves + 7u;                        // bits of a well-formed tree behind packing bit 80
        // (batches only: four such waves share a CU's scalar unit there -- 1,024 tiles of 200 x 200: 30.3 -> 26.5 us; alone on the chip the
        // walk below is the shorter chain -- one tile per call: 69.3 us with it, 70.6-72.6 us with the scans)
        if (nTiles >= 64u && rootBit == 0u && nLeaves >= 2u && T <= 8u * (visible - 10u)) {
            uint32_t *tile = leafLds, *gbArr = leafLds + 128, *symArr = leafLds + 384;
            tile[lane] = stage0;
            tile[64u + lane] = stage1;
            __syncthreads();
            const uint32_t b0 = 9u + 40u * lane, w = b0 >> 5, sh = b0 & 31u;
            uint64_t cb = (((uint64_t)tile[w + 1u] << 32) | tile[w]) >> sh;
            if (sh > 16u) cb |= (uint64_t)tile[w + 2u] << (64u - sh);
            cb &= (1ull << 48) - 1ull;
            // the state a record walk leaves the lane's forty bits in, for each state it may enter them in
            // (nine nibbles: E[p + 1] .. E[p + 8] in wLo, E[p + 9] in wHi; 32-bit steps, the forty of them unrolled)
            uint32_t wLo = 0x76543210u, wHi = 8u;
            {
                const uint32_t cbLo = (uint32_t)cb, cbHi = (uint32_t)(cb >> 32);
#pragma unroll
                for (int p = 39; p >= 0; p--) {
                    const bool bit = p >= 32 ? ((cbHi >> (p - 32)) & 1u) != 0u : ((cbLo >> p) & 1u) != 0u;
                    const uint32_t e = bit ? wHi : wLo & 15u;
                    wHi = wLo >> 28;
                    wLo = (wLo << 4) | e;
                }
            }
            const uint32_t nLanes = (T - 9u + 39u) / 40u;             // lanes with bits of the tree
            uint32_t ent = 0;
            {
                uint32_t sState = 0;
                for (uint32_t l = 0; l < nLanes; l++) {
                    ent = l == lane ? sState : ent;
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)wLo, (int)l), hi = (uint32_t)__builtin_amdgcn_readlane((int)wHi, (int)l);
                    sState = sState < 8u ? (lo >> (4u * sState)) & 15u : hi & 15u;
                }
            }
            // the lane's own records: leaves (at most five: nine bits each) with the branch records in front of them in this lane
            uint32_t nLeaf = 0, nb = 0, nbPack = 0;
            uint64_t symPack = 0;
            {
                uint32_t p = ent;
                const uint32_t lim = lane < nLanes ? min(40u, T - b0) : 0u;
                while (__any(p < lim)) {
                    const bool live = p < lim;
                    const bool leaf = live && ((cb >> (p & 63u)) & 1ull);
                    const uint32_t sym = (uint32_t)(cb >> ((p + 1u) & 63u)) & 0xffu;
                    symPack |= leaf ? (uint64_t)sym << (8u * nLeaf) : 0ull;
                    nbPack |= leaf ? nb << (6u * nLeaf) : 0u;
                    nLeaf += leaf ? 1u : 0u;
                    nb += live && !leaf ? 1u : 0u;
                    p += leaf ? 9u : live ? 1u : 0u;
                }
            }
            const uint32_t leafIncl = gf_wave_incl_scan(nLeaf), nbIncl = gf_wave_incl_scan(nb);
            const uint32_t totalLeaves = (uint32_t)__builtin_amdgcn_readlane((int)leafIncl, 63);
            const uint32_t totalBranches = (uint32_t)__builtin_amdgcn_readlane((int)nbIncl, 63);
            if (totalLeaves == nLeaves && totalBranches == nLeaves - 2u) {
                const uint32_t leafBase = leafIncl - nLeaf, nbBase = nbIncl - nb;
#pragma unroll
                for (uint32_t j = 0; j < 5; j++)
                    if (j < nLeaf) {
                        gbArr[leafBase + j] = nbBase + ((nbPack >> (6u * j)) & 63u);
                        symArr[leafBase + j] = (uint32_t)(symPack >> (8u * j)) & 0xffu;
                    }
                __syncthreads();
                // the code lengths: a scalar recurrence over the leaves, sixty-four of them at a time (branch runs in a register's lanes)
                uint32_t cLo[4], cHi[4], lenR[4];
                uint64_t c = 0;
                uint32_t L = 1;
                const uint32_t depthCap = min(nLeaves, (uint32_t)MAX_DEPTH);
                bool bad = false, closed = false;
#pragma unroll
                for (uint32_t blk = 0; blk < 4; blk++) {
                    cLo[blk] = 0; cHi[blk] = 0; lenR[blk] = 0;
                    const uint32_t i = 64u * blk + lane;
                    const uint32_t zMine = i < nLeaves ? gbArr[i] - (i ? gbArr[i - 1u] : 0u) : 0u;
                    const uint32_t cnt = nLeaves > 64u * blk ? min(64u, nLeaves - 64u * blk) : 0u;
                    for (uint32_t k = 0; k < cnt && !bad && !closed; k++) {
                        const uint32_t z = (uint32_t)__builtin_amdgcn_readlane((int)zMine, (int)k);
                        if (L + z - 1u > depthCap) { bad = true; break; }
                        c <<= z;
                        L += z;
                        cLo[blk] = k == lane ? (uint32_t)c : cLo[blk];
                        cHi[blk] = k == lane ? (uint32_t)(c >> 32) : cHi[blk];
                        lenR[blk] = k == lane ? L : lenR[blk];
                        const uint32_t t1 = ~c ? (uint32_t)__builtin_ctzll(~c) : 64u;
                        if (t1 >= L) { closed = true; bad = 64u * blk + k + 1u != nLeaves; break; }
                        c = (c >> t1) | 1ull;
                        L -= t1;
                    }
                }
#ifndef GF_PT_FORCE_EXACT                                           // (test builds: every tree through the exact walk)
                if (!bad && closed) {
                    walked = true;
                    bp = 80u + T;
                    uint32_t fmax = 0;
                    bool intro = false, nul = false;
#pragma unroll
                    for (uint32_t blk = 0; blk < 4; blk++) {
                        const uint32_t i = 64u * blk + lane;
                        if (i < nLeaves) {
                            const uint32_t Li = lenR[blk], sym = symArr[i];
                            codes[i] = __brevll(((unsigned long long)cHi[blk] << 32) | cLo[blk]) >> (64u - Li);
                            lens[i] = (uint8_t)Li;
                            syms[i] = (uint8_t)sym;
                            fmax = max(fmax, Li);
                            intro = intro || sym == 0x7fu || sym == 0x81u;
                            nul = nul || sym == 0x80u;
                        }
                    }
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) fmax = max(fmax, (uint32_t)__shfl_xor((int)fmax, o));
                    maxLen = fmax;
                    symKinds = (__any(intro) ? GF_TREE_HAS_INTRODUCER : 0u) | (__any(nul) ? GF_TREE_HAS_NULL : 0u);
                }
#endif
            }
            __syncthreads();                                            // (the walk below writes the same LDS words)
        }
    }
#endif
    if constexpr (perWave == 1u) {
      if (!walked) {
        uint64_t fbuf = buf, c = 0;
        uint32_t fhave = have, nextW = (next - 10u) >> 2, fbp = bp, L = 1, leaves = 0, fmax = 1;
        const uint32_t depthCap = min(nLeaves, (uint32_t)MAX_DEPTH);
        bool closed = false, bad = rootBit != 0u;
        while (leaves < nLeaves && !bad && !closed) {
            if (fhave <= 32u) {
                fbuf |= (uint64_t)fetch(nextW) << fhave;
                fhave += 32u;
                nextW++;
            }
            uint32_t z = fbuf ? (uint32_t)__builtin_ctzll(fbuf) : 64u;
            z = z < fhave ? z : fhave;
            z = z < 63u ? z : 63u;
            c <<= z;
            L += z;
            fbuf >>= z;
            fhave -= z;
            fbp += z;
            if (L - 1u > depthCap) { bad = true; break; }
            if (fhave >= 9u && ((uint32_t)fbuf & 1u)) {
                const uint32_t sym = ((uint32_t)fbuf >> 1) & 0xffu;
                const uint64_t code = __brevll(c) >> (64u - L);
                GfU4 recW;
                recW.x = (uint32_t)code; recW.y = (uint32_t)(code >> 32); recW.z = L | (sym << 8); recW.w = 0u;
                reinterpret_cast<GfU4 *>(leafLds)[leaves] = recW;        // (by every lane: the same words to the same place)
                fbuf >>= 9;
                fhave -= 9u;
                fbp += 9u;
                fmax = fmax > L ? fmax : L;
                leaves++;
                const uint32_t t1 = ~c ? (uint32_t)__builtin_ctzll(~c) : 64u;
                if (t1 >= L) closed = true;
                else {
                    c = (c >> t1) | 1ull;
                    L -= t1;
                }
            }
        }
#ifndef GF_PT_FORCE_EXACT                                           // (test builds: every tree through the exact walk)
        if (!bad && closed && leaves == nLeaves) {
            walked = true;
            bp = fbp;
            maxLen = fmax;
            __syncthreads();                                        // (one wave: the records are in LDS)
            bool intro = false, nul = false;
            for (uint32_t i = lane; i < nLeaves; i += 64u) {
                const GfU4 r4 = reinterpret_cast<const GfU4 *>(leafLds)[i];
                codes[i] = ((unsigned long long)r4.y << 32) | r4.x;
                const uint32_t sym = r4.z >> 8;
                lens[i] = (uint8_t)r4.z;
                syms[i] = (uint8_t)sym;
                intro = intro || sym == 0x7fu || sym == 0x81u;
                nul = nul || sym == 0x80u;
            }
            symKinds = (__any(intro) ? GF_TREE_HAS_INTRODUCER : 0u) | (__any(nul) ? GF_TREE_HAS_NULL : 0u);
        }
#endif
      }
    } else {
        uint64_t fbuf = buf, c = 0;
        uint32_t fhave = have, nextW = (next - 10u) >> 2, fbp = bp, L = 1, leaves = 0, fmax = 1, fkinds = 0;
        uint32_t pre = fetch(nextW);
        const uint32_t depthCap = min(nLeaves, (uint32_t)MAX_DEPTH);
        bool closed = false, bad = rootBit != 0u;
        for (;;) {
            const bool live = leaves < nLeaves && !bad && !closed;
            if (!__any(live)) break;
            const bool top = fhave <= 32u;
            fbuf |= top ? (uint64_t)pre << (fhave & 63u) : 0ull;
            fhave += top ? 32u : 0u;
            nextW += top ? 1u : 0u;
            pre = fetch(nextW);
            uint32_t z = fbuf ? (uint32_t)__builtin_ctzll(fbuf) : 64u;
            z = live ? min(min(z, fhave), 63u) : 0u;           // (a run of 64 goes on in the next turn)
            c <<= z;
            L += z;
            fbuf >>= z;
            fhave -= z;
            fbp += z;
            bad = bad || L - 1u > depthCap;
            // the leaf behind the run, if its nine bits are in the buffer (else the next turn tops it up and finds a run of none)
            const bool leaf = live && !bad && fhave >= 9u && ((uint32_t)fbuf & 1u) != 0u;
            const uint32_t sym = ((uint32_t)fbuf >> 1) & 0xffu;
            if (leaf && writer) {
                codes[leaves] = __brevll(c) >> (64u - L);
                lens[leaves] = (uint8_t)L;
                syms[leaves] = (uint8_t)sym;
            }
            const uint32_t adv = leaf ? 9u : 0u;
            fbuf >>= adv;
            fhave -= adv;
            fbp += adv;
            const uint32_t kd = sym - 0x7fu;                   // 0x7f, 0x81: an introducer; 0x80: the null code
            fkinds |= (leaf && kd < 3u) ? ((kd & 1u) ? GF_TREE_HAS_NULL : GF_TREE_HAS_INTRODUCER) : 0u;
            fmax = leaf ? max(fmax, L) : fmax;
            leaves += leaf ? 1u : 0u;
            const uint32_t t1 = ~c ? (uint32_t)__builtin_ctzll(~c) : 64u;      // trailing ones
            const bool closes = leaf && t1 >= L;
            closed = closed || closes;
            const bool up = leaf && !closes;
            c = up ? (c >> (t1 & 63u)) | 1ull : c;
            L -= up ? t1 : 0u;
        }
#ifdef GF_PT_FORCE_EXACT                                            // (test builds: every tree through the exact walk)
        if (false) {
#else
        if (!bad && closed && leaves == nLeaves) {
#endif
            walked = true;
            bp = fbp;
            maxLen = fmax;
            symKinds = fkinds;
        }
    }
    if (walked) {
    } else if (rootBit == 1) {
        uniformSym = (int32_t)take(8);
        symKinds = kindOf((uint32_t)uniformSym);
    } else {
        uint64_t c = 0;
        uint32_t L = 1;
        uint32_t leaves = 0, records = 0;
        bool complete = false;
        while (leaves < nLeaves) {
            refill();
            if (records > 511) { st = GF_K_ERR_BOUNDS; break; }
            uint32_t z = buf ? (uint32_t)__builtin_ctzll(buf) : 64u;
            z = min(z, have);
            if (z) {
                if (L - 1u + z > nLeaves || records + z > 2u * nLeaves - 1u) { st = GF_K_ERR_BOUNDS; break; }   // as parse_tree_wave
                if (L - 1u + z > MAX_DEPTH) { st = GF_K_ERR_FORMAT; break; }
                c <<= z;
                L += z;
                records += z;
                buf >>= z;
                have -= z;
                bp += z;
                if (have == 0u || !(buf & 1ull)) continue;
                refill();
            }
            if (records + 1u > 2u * nLeaves - 1u) { st = GF_K_ERR_BOUNDS; break; }
            const uint32_t r9 = take(9);
            records++;
            if (writer) {
                codes[leaves] = __brevll(c) >> (64u - L);
                lens[leaves] = (uint8_t)L;
                syms[leaves] = (uint8_t)(r9 >> 1);
            }
            symKinds |= kindOf(r9 >> 1);
            maxLen = max(maxLen, L);
            leaves++;
            const uint32_t t1 = ~c ? (uint32_t)__builtin_ctzll(~c) : 64u;
            if (leaves == nLeaves) { complete = t1 >= L; break; }
            if (t1 >= L) { st = GF_K_ERR_BOUNDS; break; }
            c = (c >> t1) | 1ull;
            L -= t1;
        }
        if (st == GF_K_OK && !complete && leaves == nLeaves) {       // incomplete tree: see DecShared::skipLen
            skipLen = L;
            skipPath = __brevll(c) >> (64u - L);
        }
    }
    if (st == GF_K_OK && bp > len * 8u) st = GF_K_ERR_BOUNDS;
    if (!writer) return;
    // Which run of the fast decode kernel takes the tile (round 5): what that kernel needs in its M32 buffer is the stream (nM32
    // bytes: bytes 6..9 of the header, CodecHuffman.java:122-124) and, during the Huffman pass, the packing's words plus padding.
    // A tile that outgrows the usual buffer and fits the roomy one is marked here, before either run starts.
    uint32_t roomy = 0;
    if (roomyBytes > fastBytes && st == GF_K_OK && roomyList && clearFlags) {
        const uint32_t nM32 = reinterpret_cast<const PackedWord *>(pk + 6)->v;           // (len >= 10)
        const uint32_t pkWords = (((uint32_t)(off * 8ull) & 31u) + len * 8u + 31u) >> 5;
        const uint64_t need = max((uint64_t)nM32, ((uint64_t)pkWords + FAST_TEXT_PAD) * 4u);
        roomy = (need > fastBytes && need <= roomyBytes) ? GF_TREE_ROOMY : 0u;
        // ... and listed for the roomy run's workgroups (clearFlags[2]: the count, zero when the pre-pass starts: the general
        // decode kernel of the batch before left it so)
        // (the list has nTiles entries: a count that a batch which ended early left behind, or another batch of this context on
        // another stream, must not carry the writer beyond them -- the reader clamps to nTiles as well)
        if (roomy) {
            const uint32_t at = atomicAdd(clearFlags + 2, 1u);
            if (at < nTiles) roomyList[at] = (uint32_t)t;
        }
    }
    rec[0] = (uint32_t)st;
    rec[1] = nLeaves;
    rec[2] = bp;
    rec[3] = maxLen | symKinds | roomy;
    rec[4] = (uint32_t)uniformSym;
    rec[5] = skipLen;
    rec[6] = (uint32_t)skipPath;
    rec[7] = (uint32_t)(skipPath >> 32);
}

#endif  // GF_DEC_VARIANT

}  // namespace

#ifndef GF_DEC_VARIANT
uint32_t gf_huffman_decode_lds_m32(int nRows, int nCols)
{
    // typical M32 streams are ~1.0-1.1 bytes per cell; larger ones spill to the workspace
    size_t cells = (size_t)nRows * (size_t)nCols;
    size_t want = cells + cells / 8 + 512;
    if (want < 8192) want = 8192;
    if (want > 98304) want = 98304;             // with bitmap, rank bases and the static part of the 1024-thread build: 144 KB of the
                                                // CU's 160 (256x256 tiles: 74 KB of stream, one workgroup of 16 waves per CU)
    return (uint32_t)((want + 31) & ~(size_t)31);
}

#endif

// dynamic LDS bytes for a given M32 capacity: bytes + start bitmap + rank bases
static size_t decodeDynLds(uint32_t ldsM32Bytes, uint32_t ldsTextBytes)
{
    // the bitmap + rank area doubles as the second-level lookup table (L2_ENTRIES uint16) during phase 1
    const size_t bm = 2 * ((size_t)(ldsM32Bytes >> 5) + 2) * 4;
    size_t pad = 0;
#ifdef GF_DEC_LDS_PAD_ENV
    // experiment builds only (python -m gridfour_amd.build --variant NAME -DGF_DEC_LDS_PAD_ENV): unused dynamic LDS that takes
    // workgroups off a CU -- the occupancy sensitivity of the kernel on one tile shape (tools/occupancy_sweep.sh)
    if (const char *e = getenv("GF_DEC_LDS_PAD")) pad = (size_t)atol(e);
#endif
    return (size_t)ldsM32Bytes + (bm > 2 * L2_ENTRIES ? bm : 2 * L2_ENTRIES) + ldsTextBytes + pad;
}

#ifndef GF_DEC_VARIANT
uint32_t gf_huffman_decode_lds_text(int nRows, int nCols)
{
    // LDS copy of the packing text: measured on MI355X (ETOPO1-shaped batch) the copy makes a decode
    // pass ~25 % shorter but costs half the resident workgroups per CU, a net loss (5.3 vs 4.2 ms), so it
    // is off by default; the code path stays for tiles/LDS budgets where it wins.
    (void)nRows;
    (void)nCols;
    return 0;
}

unsigned gf_huffman_decode_grid(size_t nTiles)
{
    const size_t cap = 256 * 8;                    // workgroups resident on the chip, upper bound
    return (unsigned)(nTiles < cap ? (nTiles ? nTiles : 1) : cap);
}

#endif

// LDS bytes of one workgroup of this build for the launch a describes (static + dynamic): decodeBatchDev weighs the two builds
size_t gf_huffman_decode_lds_per_wg(const GfDecodeArgs &a)
{
    return sizeof(DecShared) + decodeDynLds(a.ldsM32Bytes, a.ldsTextBytes);
}

hipError_t gf_launch_huffman_decode(const GfDecodeArgs &a, hipStream_t stream, unsigned grid, const GfSideStream *side)
{
    if (a.nTiles == 0) return hipSuccess;
    const size_t dyn = decodeDynLds(a.ldsM32Bytes, a.ldsTextBytes);
    static GfDynLdsOptIn optG, optA, optF;
    hipError_t e;
    if (a.analysis) {
        if ((e = gf_opt_in_dyn_lds(k_huffman_decode<DEC_ANALYZE>, dyn, optA)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_huffman_decode<DEC_ANALYZE>, dim3(grid), dim3(DEC_THREADS), dyn, stream, a);
        return hipGetLastError();
    }
    if ((e = gf_opt_in_dyn_lds(k_huffman_decode<DEC_GENERAL>, dyn, optG)) != hipSuccess) return e;
    if (a.retryFlag) {
        // CodecHuffman batches: the fast kernel first; the general one picks up what that one marked (and returns at once
        // when nothing is marked)
        const size_t dynRoomy = a.ldsM32Roomy ? decodeDynLds(a.ldsM32Roomy, a.ldsTextBytes) : 0;
        if ((e = gf_opt_in_dyn_lds(k_huffman_decode<DEC_FAST>, dyn, optF)) != hipSuccess) return e;
        GfDecodeArgs f = a;
        if (a.lean) {
            // the one-tile-per-call path: the fast kernel alone (a tile it leaves behind keeps GF_K_RETRY; the caller sees to it)
            hipLaunchKernelGGL(k_huffman_decode<DEC_FAST>, gf_tile_grid(a.nTiles), dim3(DEC_THREADS), dyn, stream, f);
            return hipGetLastError();
        }
        if (!a.flagsCleared && (e = hipMemsetAsync(a.retryFlag, 0, 8, stream)) != hipSuccess) return e;
        // The roomy run -- the tiles the pre-pass marked GF_TREE_ROOMY, with LDS for two M32 bytes per cell; the workgroups of
        // the other tiles leave at once -- BESIDE the first run (round 5): it is a few hundred tiles of a rough batch at two
        // workgroups per CU, a chain of latencies that took 0.33 ms behind the first run's 1.2.  The roomy run stays on the
        // caller's stream, directly behind the pre-pass, and the FIRST run goes to the context's side stream: the roomy workgroups
        // must reach the CUs first -- once four workgroups of the first run hold a CU's LDS (4 x 40 KB), a 55 KB workgroup finds
        // no room until two of them end together, i.e. until the first run drains (measured: the other order gained 0.06 ms of
        // the 0.33).
        // (a small batch -- BASELINE config 2: 1,024 tiles, 0.18 ms per decode -- loses more to the two hand-overs between the
        // streams, ~10 us each, than the roomy run could hide: 0.183 -> 0.201 ms measured; there the runs follow one another)
        // ... and a batch whose predecessors on this context listed no tile for the roomy run (smooth terrain: the run is 7 us of empty
        // workgroups) keeps everything on one stream: the hand-overs were 15-20 us of its 0.70 ms.  The hint (roomySeenHost: 1 + the
        // count of the last batch whose general kernel has finished, 0 before the first) may be a batch or two old; either order of
        // the runs is correct for any data.
        const bool roomyLikely = !a.roomySeenHost || *(volatile const uint32_t *)a.roomySeenHost != 1u;
        const bool beside = a.ldsM32Roomy && side && side->stream && a.nTiles >= 4096 && roomyLikely;
        // (round 6) a SMALL batch whose predecessors listed no tile for the roomy run does without its launch (5 us of BASELINE config
        // 2's 165): should the pre-pass list a tile after all, the first run tries it, the general kernel takes it, and the next batch
        // knows.  What a stale hint costs (-DGF_DEC_FORCE_NO_ROOMY on the rough surface): 1,024 tiles of 120 x 150 0.304 -> 0.339 ms,
        // 1,300 0.350 -> 0.384, 3,000 0.530 -> 0.669 -- hence small batches only.  (For every batch, with a reduced grid for the run
        // where none is expected: a caller that queues a smooth batch and then rough ones without waiting had each of them draw
        // its 650 roomy tiles through 64 workgroups -- the default bench line's rough sub-record, 1.37 -> 2.38 ms; taken back.)
#ifdef GF_DEC_FORCE_NO_ROOMY                                        // (experiment builds)
        const bool noRoomy = a.ldsM32Roomy && a.nTiles < 4096;
#else
        const bool noRoomy = a.ldsM32Roomy && !roomyLikely && a.nTiles < 2048;
#endif
        f.noRoomyRun = noRoomy ? 1 : 0;
        GfDecodeArgs r = f;
        r.ldsM32Bytes = a.ldsM32Roomy;
        // (persistent workgroups: as many as the chip holds of them -- LDS in 1,280-byte steps, 256 CUs -- and never more than tiles)
        static GfDynLdsOptIn optR;
        unsigned roomyGrid = 1;
        if (a.ldsM32Roomy) {
            if ((e = gf_opt_in_dyn_lds(k_huffman_decode<DEC_FAST_ROOMY>, dynRoomy, optR)) != hipSuccess) return e;
            const size_t step = 1280, per = (sizeof(DecShared) + dynRoomy + 64 + step - 1) / step * step;
            // (the chip's CUs and a CU's LDS from the device, not from this file: advice of round 5)
            int dev = 0, cus = 256, ldsPerCu = 160 * 1024;
            if (hipGetDevice(&dev) == hipSuccess) {
                int v = 0;
                if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
                if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && v > 0) ldsPerCu = v;
            }
            (void)hipGetLastError();
            const size_t perCu = std::max<size_t>(1, std::min<size_t>((size_t)ldsPerCu / per, 2048 / DEC_THREADS));
            roomyGrid = (unsigned)std::min<size_t>(a.nTiles, perCu * (size_t)cus);
        }
        if (beside) {
            if ((e = hipEventRecord(side->fork, stream)) != hipSuccess) return e;            // (behind the pre-pass)
            if ((e = hipStreamWaitEvent(side->stream, side->fork, 0)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_huffman_decode<DEC_FAST_ROOMY>, dim3(roomyGrid), dim3(DEC_THREADS), dynRoomy, stream, r);
            hipLaunchKernelGGL(k_huffman_decode<DEC_FAST>, gf_tile_grid(a.nTiles), dim3(DEC_THREADS), dyn, side->stream, f);
            if ((e = hipEventRecord(side->join, side->stream)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(stream, side->join, 0)) != hipSuccess) return e;
        } else {
            if (a.ldsM32Roomy && !noRoomy) hipLaunchKernelGGL(k_huffman_decode<DEC_FAST_ROOMY>, dim3(roomyGrid), dim3(DEC_THREADS), dynRoomy, stream, r);
            hipLaunchKernelGGL(k_huffman_decode<DEC_FAST>, gf_tile_grid(a.nTiles), dim3(DEC_THREADS), dyn, stream, f);
        }
    }
    hipLaunchKernelGGL(k_huffman_decode<DEC_GENERAL>, dim3(grid), dim3(DEC_THREADS), dyn, stream, a);
    return hipGetLastError();
}

// The canonical run (DEC_FAST_CANON): a.trees = the records of k_canon_parse_lengths, a.retryFlag[0] = zero before the launch and
// non-zero behind it when some tile is left to k_canon_decode (status GF_K_RETRY).
hipError_t gf_launch_huffman_decode_canon(const GfDecodeArgs &a, hipStream_t stream)
{
    if (a.nTiles == 0) return hipSuccess;
    if (!a.retryFlag || !a.trees || a.ldsM32Roomy) return hipErrorInvalidValue;
    const size_t dyn = decodeDynLds(a.ldsM32Bytes, 0);
    static GfDynLdsOptIn opt;
    const hipError_t e = gf_opt_in_dyn_lds(k_huffman_decode<DEC_FAST_CANON>, dyn, opt);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_huffman_decode<DEC_FAST_CANON>, gf_tile_grid(a.nTiles), dim3(DEC_THREADS), dyn, stream, a);
    return hipGetLastError();
}

#ifndef GF_DEC_VARIANT
hipError_t gf_launch_lsop_unpack_m32(const GfLsopM32Args &a, hipStream_t stream, unsigned grid)
{
    if (a.nTiles == 0) return hipSuccess;
    const size_t dyn = decodeDynLds(a.ldsM32Bytes, 0);
    static GfDynLdsOptIn opt;
    hipError_t e = gf_opt_in_dyn_lds(k_lsop_unpack_m32, dyn, opt);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_lsop_unpack_m32, dim3(grid), dim3(DEC_THREADS), dyn, stream, a);
    return hipGetLastError();
}

hipError_t gf_launch_huffman_parse_trees(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, size_t slotStride,
                                         const uint32_t *lengths, uint32_t *trees, size_t nTiles, hipStream_t stream, uint32_t *clearFlags,
                                         uint32_t fastBytes, uint32_t roomyBytes, uint32_t *roomyList)
{
    if (nTiles == 0) return hipSuccess;
    if (gf_prepass_tiles_per_wave(nTiles) == 1u)
        hipLaunchKernelGGL(k_huffman_parse_trees<1>, dim3((unsigned)nTiles), dim3(64), 0, stream, blob, blobBytes, offsets, slotStride,
                           lengths, trees, nTiles, clearFlags, fastBytes, roomyBytes, roomyList);
    else
        hipLaunchKernelGGL(k_huffman_parse_trees<64>, dim3((unsigned)((nTiles + 63) / 64)), dim3(64), 0, stream, blob, blobBytes, offsets,
                           slotStride, lengths, trees, nTiles, clearFlags, fastBytes, roomyBytes, roomyList);
    return hipGetLastError();
}
#endif  // GF_DEC_VARIANT
