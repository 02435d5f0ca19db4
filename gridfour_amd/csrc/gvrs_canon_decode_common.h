// gvrs_canon_decode_common.h -- device side of CanonicalHuffman.decode (compress/canonicalHuffman/), shared by the
// CodecCanonHuffman and LSOP decode kernels.  Include inside the kernel file's anonymous namespace after
// gvrs_decode_common.h.  See gvrs_canon_decode.hip for the phases.
#pragma once

#include <type_traits>

constexpr int CN_SYMS = 260;
constexpr int CN_NULL = 256, CN_ESC1 = 257, CN_ESC2 = 258, CN_EOT = 259;
constexpr int CN_META = 20;
constexpr int CD_LUT_BITS = 11;
constexpr int CD_MAXQ = 512;
constexpr int CD_NCUR = CD_MAXQ / DEC_THREADS > 0 ? CD_MAXQ / DEC_THREADS : 1;   // subsequences per thread: tid, tid + DEC_THREADS, ...
constexpr uint32_t CD_WARM = 128;
constexpr uint32_t CD_END_EOT = 0xFFFFFFFFu;     // subsequence ended on the end-of-text symbol
constexpr uint32_t CD_END_BAD = 0xFFFFFFFEu;     // subsequence ran into an invalid code / the end of the data

struct CanonDec {
    uint32_t lut[1 << CD_LUT_BITS];             // see cd_entry(); 0 = longer than 11 bits (or no code)
    uint16_t symByOrder[320];                   // symbols ordered by (length, symbol)
    uint32_t first[16], count[16], offset[16];  // canonical code of the first symbol of a length, how many, where
    uint8_t len[320];
    uint8_t metaLen[160];
    uint16_t metaLut[256];                      // (len << 5) | symbol over 8 bits
    uint16_t metaSym[32];
    uint32_t mfirst[16], mcount[16], moffset[16];
    uint32_t qs[CD_MAXQ], qe[CD_MAXQ], qc[CD_MAXQ], qx[CD_MAXQ];   // start, end, values, bit after the end-of-text symbol
    uint32_t waveSum[DEC_WAVES];
    uint32_t textStart;
    int32_t parseStatus;
    uint32_t changed;
    uint32_t qStar;
    uint32_t carry;
    int32_t runStatus;
};

// The packing's text as 32-bit words, read either from the LDS copy (the normal case; an explicit LDS array so that
// the reads are ds_read, not flat loads) or, for packings larger than the copy, from the blob where it lies.
extern __shared__ __attribute__((aligned(16))) uint32_t cdLdsText[];

struct CdTextLds {
    uint32_t nWords;
    __device__ __forceinline__ uint32_t word(uint32_t i) const { return cdLdsText[i]; }
};
struct CdTextGlobal {
    const uint32_t *w;
    uint32_t nWords;
    __device__ __forceinline__ uint32_t word(uint32_t i) const { return i < nWords ? w[i] : 0u; }
};

// 32 stream bits at absolute bit position pos (LSB-first); the LDS copy is padded with two zero words
template <class Text>
__device__ __forceinline__ uint32_t cd_peek(const Text T, uint32_t pos)
{
    const uint32_t i = pos >> 5;
    return __builtin_amdgcn_alignbit(T.word(i + 1), T.word(i), pos & 31u);
}

// copies a packing (bits [0, endBit) of the word array starting at w32[word0]) into the LDS text, zero beyond its end
__device__ __forceinline__ void cd_stage_text(const uint32_t *__restrict__ w32, uint64_t word0, uint64_t nWords, uint32_t endBit,
                                              uint32_t needWords)
{
    for (uint32_t i = threadIdx.x; i < needWords; i += DEC_THREADS) {
        uint32_t w = word0 + i < nWords ? w32[word0 + i] : 0u;
        const uint32_t b0 = i * 32u;
        if (b0 + 32u > endBit) w = b0 >= endBit ? 0u : (w & ((1u << (endBit - b0)) - 1u));
        cdLdsText[i] = w;
    }
}

// canonical search over lengths [lmin, lmax]: c = next bits most-significant first (bit-reversed window)
__device__ __forceinline__ uint32_t cd_search(const uint32_t *first, const uint32_t *count, const uint32_t *offset,
                                              const uint16_t *symByOrder, uint32_t c, int lmin, int lmax, uint32_t *len)
{
    for (int l = lmin; l <= lmax; l++) {
        const uint32_t d = (c >> (32 - l)) - first[l];
        if (d < count[l]) {
            *len = (uint32_t)l;
            return symByOrder[offset[l] + d];
        }
    }
    *len = 0;
    return 0;
}

// lengths -> canonical decode tables (one wave): CanonHuffTreeDecoder.java:68-95 (sort by length, symbol;
// consecutive codes, shifted when the length grows)
template <int NREG>
__device__ __forceinline__ void cd_tables(const uint8_t *len, int nSym, uint32_t *first, uint32_t *count, uint32_t *offset,
                                          uint16_t *symByOrder, int lane, uint32_t *nUsed)
{
    uint32_t L[NREG], rank[NREG];
#pragma unroll
    for (int r = 0; r < NREG; r++) {
        const int e = r * 64 + lane;
        L[r] = e < nSym ? len[e] : 0u;
        rank[r] = 0;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t code = 0, off = 0;
    for (uint32_t l = 1; l <= 15; l++) {
        uint32_t running = 0;
#pragma unroll
        for (int r = 0; r < NREG; r++) {
            const unsigned long long m = __ballot(L[r] == l);
            if (L[r] == l) rank[r] = running + (uint32_t)__popcll(m & lt);
            running += (uint32_t)__popcll(m);
        }
        if (lane == 0) { first[l] = code; count[l] = running; offset[l] = off; }
#pragma unroll
        for (int r = 0; r < NREG; r++)
            if (L[r] == l) symByOrder[off + rank[r]] = (uint16_t)(r * 64 + lane);
        code = (code + running) << 1;
        off += running;
    }
    if (lane == 0) { first[0] = 0; count[0] = 0; offset[0] = 0; }
    *nUsed = off;
    __builtin_amdgcn_wave_barrier();
}

// The same tables for the 261 text lengths by the WHOLE workgroup (round 5): a symbol's place in the (length, symbol) order is the
// number of symbols with a shorter code plus those of its own length before it -- fifteen ballots inside each wave of 64 symbols, a
// table of per-wave counts across them, one wave that adds those up.  (cd_tables<5> on wave 0 was fifteen dependent turns with
// every other wave of the workgroup waiting: with the fetch of the pre-pass's record 25 K cycles of a tile's 200 K.)
// len: 320 entries in LDS, zero from the 261st on; scr: 192 words of LDS scratch.  All threads call; ends with a barrier.
// Returns the number of symbols that have a code.
__device__ __forceinline__ uint32_t cd_tables_wg(const uint8_t *len, uint32_t *first, uint32_t *count, uint32_t *offset,
                                                 uint16_t *symByOrder, uint32_t *scr)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    constexpr uint32_t TURNS = (320u + DEC_THREADS - 1u) / DEC_THREADS;      // symbols per thread: 2 with 256 threads, else 1
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t L[TURNS], within[TURNS];
#pragma unroll
    for (uint32_t u = 0; u < TURNS; u++) {
        const uint32_t s = u * DEC_THREADS + tid, vw = s >> 6;               // vw: the wave of 64 symbols this one belongs to (0 .. 4)
        L[u] = s < 320u ? len[s] : 0u;
        within[u] = 0;
        if ((s & ~63u) < 320u) {                                             // (wave-uniform)
#pragma unroll
            for (uint32_t l = 1; l <= 15u; l++) {
                const unsigned long long m = __ballot(L[u] == l);
                within[u] = L[u] == l ? (uint32_t)__popcll(m & lt) : within[u];
                if (lane == 0u) scr[vw * 16u + l] = (uint32_t)__popcll(m);
            }
        }
    }
    __syncthreads();
    if (tid < 64u) {
        const uint32_t l = lane & 15u;
        uint32_t tot = 0;
        if (lane >= 1u && lane < 16u) {
#pragma unroll
            for (uint32_t v = 0; v < 5u; v++) {
                const uint32_t c = scr[v * 16u + l];
                scr[80u + v * 16u + l] = tot;                                // places of this length taken by the waves before
                tot += c;
            }
        }
        const uint32_t incl = gf_wave_incl_scan(tot);
        uint32_t f = 0;
#pragma unroll
        for (uint32_t j = 1; j <= 14u; j++) {                                // first[l] = (first[l - 1] + count[l - 1]) << 1
            const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)tot, (int)j);
            f += j < l ? tj << (l - j) : 0u;
        }
        if (lane < 16u) {
            first[l] = lane ? f : 0u;
            count[l] = tot;
            offset[l] = lane ? incl - tot : 0u;
        }
        if (lane == 15u) scr[160] = incl;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < TURNS; u++) {
        const uint32_t s = u * DEC_THREADS + tid;
        if (L[u]) symByOrder[offset[L[u]] + scr[80u + (s >> 6) * 16u + L[u]] + within[u]] = (uint16_t)s;
    }
    const uint32_t nUsed = scr[160];
    __syncthreads();
    return nUsed;
}

struct CdTok {
    uint32_t sym, bits, raw;     // bits = code + raw bits; sym == 0xFFFF: no such code
};

// Lookup-table entry (32 bits): symbol (9 bits) | code length << 9 (4 bits) | escape kind << 13 (1: two raw bits
// follow, 2: a raw byte).  Where the window holds TWO complete codes of plain values (symbols < 256) the entry also
// carries the second one: symbol2 << 15 (8 bits) | both lengths << 23 (5 bits) | bit 31.  The fields of the first
// symbol are valid either way.
__device__ __forceinline__ uint32_t cd_entry(uint32_t sym, uint32_t cl)
{
    const uint32_t kind = sym == (uint32_t)CN_ESC2 ? 1u : sym == (uint32_t)CN_ESC1 ? 2u : 0u;
    return sym | (cl << 9) | (kind << 13);
}
__device__ __forceinline__ uint32_t cd_e_sym(uint32_t e) { return e & 511u; }
__device__ __forceinline__ uint32_t cd_e_len(uint32_t e) { return (e >> 9) & 15u; }
__device__ __forceinline__ uint32_t cd_e_extra(uint32_t e) { const uint32_t k = (e >> 13) & 3u; return (k & 1u) * 2u + (k >> 1) * 8u; }
__device__ __forceinline__ bool cd_e_pair(uint32_t e) { return e >> 31; }
__device__ __forceinline__ uint32_t cd_e_sym2(uint32_t e) { return (e >> 15) & 255u; }
__device__ __forceinline__ uint32_t cd_e_len12(uint32_t e) { return (e >> 23) & 31u; }

// entry of the code that starts at bit 0 of the 32-bit window w; 0x7FFFFFFF: no such code
__device__ __forceinline__ uint32_t cd_entry_of(const CanonDec &S, uint32_t w)
{
    uint32_t e = S.lut[w & ((1u << CD_LUT_BITS) - 1u)];
    if (!e) {                                             // longer than the table's 11 bits, or no such code
        uint32_t cl;
        const uint32_t sym = cd_search(S.first, S.count, S.offset, S.symByOrder, __brev(w), CD_LUT_BITS + 1, 15, &cl);
        e = cl ? cd_entry(sym, cl) : 0x7FFFFFFFu;
    }
    return e;
}

// the FIRST token of an entry: symbol, total bits (code + raw), raw bits
__device__ __forceinline__ CdTok cd_token_from(uint32_t e, uint32_t w)
{
    CdTok t;
    if (e == 0x7FFFFFFFu) { t.sym = 0xFFFFu; t.bits = 1; t.raw = 0; return t; }
    const uint32_t cl = cd_e_len(e), extra = cd_e_extra(e);
    t.sym = cd_e_sym(e);
    t.raw = (w >> cl) & ((1u << extra) - 1u);
    t.bits = cl + extra;
    return t;
}

__device__ __forceinline__ CdTok cd_token_of(const CanonDec &S, uint32_t w) { return cd_token_from(cd_entry_of(S, w), w); }

// Sequential reader: three words of the text in registers, the third fetched a word ahead of its use, the decode
// window formed with one v_alignbit_b32 -- the only latency on the per-token path is the table lookup.
template <class Text>
struct CdCur {
    uint32_t w0, w1, w2, sh, wi, pos;
    __device__ __forceinline__ void seek(const Text &T, uint32_t p)
    {
        pos = p;
        wi = p >> 5;
        sh = p & 31u;
        w0 = T.word(wi);
        w1 = T.word(wi + 1);
        w2 = T.word(wi + 2);
    }
    __device__ __forceinline__ uint32_t window() const { return __builtin_amdgcn_alignbit(w1, w0, sh); }
    __device__ __forceinline__ void advance(const Text &T, uint32_t n)     // n < 32
    {
        pos += n;
        sh += n;
        if (sh >= 32u) {
            sh -= 32u;
            w0 = w1;
            w1 = w2;
            wi++;
            w2 = T.word(wi + 2);
        }
    }
};

// decodes from the cursor's position: tokens until the next boundary; returns where it ended and how many
// values started on the way
template <class Text>
__device__ __forceinline__ void cd_run(const CanonDec &S, const Text &T, CdCur<Text> &cur, uint32_t bound, uint32_t endBit,
                                       uint32_t *endOut, uint32_t *cntOut, uint32_t *eotEnd)
{
    uint32_t cnt = 0, end;
    for (;;) {
        if (cur.pos >= bound) { end = cur.pos; break; }
        if (cur.pos >= endBit) { end = CD_END_BAD; break; }
        const uint32_t w = cur.window();
        const uint32_t e = cd_entry_of(S, w);
        if (cd_e_pair(e) && cur.pos + cd_e_len(e) < bound) {     // two plain values, the second one starts before the boundary
            cnt += 2u;
            cur.advance(T, cd_e_len12(e));
            continue;
        }
        const CdTok t = cd_token_from(e, w);
        if (t.sym == 0xFFFFu) { end = CD_END_BAD; break; }
        if (t.sym == (uint32_t)CN_EOT) { end = CD_END_EOT; *eotEnd = cur.pos + t.bits; break; }
        cnt += t.sym <= (uint32_t)CN_NULL ? 1u : 0u;
        cur.advance(T, t.bits);
    }
    *endOut = end;
    *cntOut = cnt;
}

// ---- token table of the synchronisation pass (round 3) ----
// Pass 1 only needs to know where tokens start and how many VALUES start on the way (a value = a symbol 0..256; the escapes
// behind it, with their raw bits, and the spare symbol 260 belong to it).  Over the CD_LUT_BITS-bit window at a token start the
// table gives: nb = bits of ALL the tokens whose CODE lies inside the window, raw bits of escapes included (so nb may reach
// past the window), ns = values among them; l1 / c1 = the same for the first token alone; CD_TOK_STOP: the first token is the
// end-of-text symbol, longer than the window or no code at all -- the caller takes that one through cd_entry_of.
// 16 bits: nb (5) | ns << 5 (4) | l1 << 9 (5) | c1 << 14 | stop << 15.
constexpr uint32_t CD_TOK_STOP = 0x8000u;
__device__ __forceinline__ void cd_build_tokens(const CanonDec &S, uint16_t *tok)
{
    for (uint32_t x = threadIdx.x; x < (1u << CD_LUT_BITS); x += DEC_THREADS) {
        uint32_t pos = 0, ns = 0, l1 = 0, c1 = 0, v = CD_TOK_STOP;
        while (pos < (uint32_t)CD_LUT_BITS) {
            const uint32_t e = S.lut[x >> pos];                     // the bits behind the tokens so far, zero-extended
            const uint32_t cl = cd_e_len(e), sym = cd_e_sym(e);
            if (!e || pos + cl > (uint32_t)CD_LUT_BITS || sym == (uint32_t)CN_EOT) break;   // not decided by the window's bits / the end
            const uint32_t bits = cl + cd_e_extra(e);
            if (pos == 0) { l1 = bits; c1 = sym <= (uint32_t)CN_NULL ? 1u : 0u; }
            ns += sym <= (uint32_t)CN_NULL ? 1u : 0u;
            pos += bits;
        }
        if (l1) v = min(pos, 31u) | (ns << 5) | (l1 << 9) | (c1 << 14);
        tok[x] = (uint16_t)v;
    }
}

// cd_run with the token table: same end, count and end-of-text position.  a = bit position in the LDS text.  The loop is
// wave-uniform (a lane that has arrived adds nothing); run == false: the lane takes no part.
__device__ __forceinline__ void cd_run_tok(const CanonDec &S, const uint16_t *tok, uint32_t a, uint32_t bound, uint32_t endBit, bool run,
                                           uint32_t *endOut, uint32_t *cntOut, uint32_t *eotEnd)
{
    const uint32_t border = min(bound, endBit);                     // no token of a group may start at or behind it
    uint32_t cnt = 0, end = 0;
    bool going = run;
    while (__any(going)) {
        const uint32_t i = a >> 5;
        const uint32_t w = __builtin_amdgcn_alignbit(cdLdsText[i + 1u], cdLdsText[i], a);
        const uint32_t t = tok[w & ((1u << CD_LUT_BITS) - 1u)];
        const bool arrived = a >= bound, out = !arrived && a >= endBit;
        uint32_t nb = 0, nv = 0;
        bool stopNow = going && (arrived || out);
        uint32_t endIf = arrived ? a : CD_END_BAD;
        if (going && !arrived && !out && (t & CD_TOK_STOP)) {       // end of text, a code longer than the window, no code at all
            const CdTok tk = cd_token_from(cd_entry_of(S, w), w);
            if (tk.sym == 0xFFFFu) { stopNow = true; endIf = CD_END_BAD; }
            else if (tk.sym == (uint32_t)CN_EOT) { stopNow = true; endIf = CD_END_EOT; *eotEnd = a + tk.bits; }
            else { nb = tk.bits; nv = tk.sym <= (uint32_t)CN_NULL ? 1u : 0u; }
        } else {
            const bool single = a + (uint32_t)CD_LUT_BITS > border;
            nv = single ? (t >> 14) & 1u : (t >> 5) & 15u;
            nb = single ? (t >> 9) & 31u : t & 31u;
        }
        const bool step = going && !stopNow;
        cnt += step ? nv : 0u;
        a += step ? nb : 0u;
        end = stopNow ? endIf : end;
        going = step;
    }
    *endOut = end;
    *cntOut = cnt;
}

// value sinks of cd_decode_stream: one(k, v) and quad(k0, four values), k0 a multiple of 4
struct CdArraySink {                      // value k -> dst[k], k < n
    int32_t *dst;
    uint32_t n;
    static constexpr bool kStaged = false;
    __device__ __forceinline__ void put(uint32_t k, uint32_t v, bool on = true) const { if (on) one(k, v); }
    __device__ __forceinline__ void one(uint32_t k, uint32_t v) const { if (k < n) dst[k] = (int32_t)v; }
    __device__ __forceinline__ void quad(uint32_t k0, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const
    {
        if (k0 + 3u < n) {
            GfU4 x;
            x.x = a; x.y = b; x.z = c; x.w = d;
            *reinterpret_cast<GfU4 *>(dst + k0) = x;
        } else { one(k0, a); one(k0 + 1u, b); one(k0 + 2u, c); one(k0 + 3u, d); }
    }
};

struct CdCellSink {                       // value k of a predictor's stream -> its cell of the tile
    uint32_t *o;
    GfCellMap map;                        // (worked out once per tile: no model dispatch per value)
    uint32_t nStream;
    __device__ __forceinline__ uint32_t cell(uint32_t k) const { return map(k); }
    bool enabled;                         // false: diagnostic ablation (no stores)
    __device__ __forceinline__ void one(uint32_t k, uint32_t v) const { if (k < nStream && enabled) o[cell(k)] = v; }
    // Staging (decode phase 2): a thread walks its own stretch of the stream, so the 64 lanes of a store instruction hit 64
    // different lines, four bytes each -- measured, that multiplies the HBM write traffic by four.  Values of -127..127 (one
    // byte; nearly all of them) are parked in LDS instead, a byte each, and expand() writes them out afterwards with
    // neighbouring lanes on neighbouring cells; the rare others (and whatever does not fit the stage) go straight to their
    // cell and leave the mark 0x80.  The stream is staged in two halves (subsequences 0..255, then 256..511).
    static constexpr bool kStaged = true;
    uint8_t *stA, *stB;                   // the stage: capA bytes at stA, then cap - capA bytes at stB (cap == 0: no staging)
    uint32_t capA, cap;
    uint32_t halfBase;                    // stream position of the half being decoded
    bool fuse;                            // the staged values stay where they are: cd_fused_triangle() turns them into the tile
    // (round 6) WINDOWS: k_lsop_unpack2's byte plane.  window != 0 asks cd_decode_stream, for a text whose every value is a byte
    // (a code without escapes and null, as many values as the reader expects), to stage the stream `window` values at a time -- each
    // subsequence is decoded once per window it reaches into -- and to call its `afterPass(first value, count)` behind each window
    // instead of writing anything to `o`; *windowed tells the caller which of the two happened
    uint32_t window = 0;
    uint32_t *windowed = nullptr;
    bool windowOn = false;                // (set by cd_decode_stream)
    __device__ __forceinline__ uint8_t *slot(uint32_t rel) const { return rel < capA ? stA + rel : stB + (rel - capA); }
    __device__ __forceinline__ void put(uint32_t k, uint32_t v, bool on = true) const
    {
        // (two flat predicated stores: nested, the conditions cost the scalar unit more than the stores cost the SIMDs)
        const uint32_t rel = k - halfBase;
        if (windowOn) {
            if (on && rel < window) *slot(rel) = (uint8_t)v;
            return;
        }
        const bool ok = on && k < nStream && enabled;
        const bool small = v + 127u <= 254u, staged = rel < cap;
        if (ok && staged) *slot(rel) = small ? (uint8_t)v : (uint8_t)0x80;
        if (ok && !(staged && small)) o[cell(k)] = v;
    }
    __device__ __forceinline__ void expand(uint32_t count) const        // the whole workgroup, between two barriers
    {
        const uint32_t n = min(count, cap);
        for (uint32_t rel = threadIdx.x; rel < n; rel += DEC_THREADS) {
            const uint32_t b = *slot(rel), k = halfBase + rel;
            if (b != 0x80u && k < nStream && enabled) o[cell(k)] = (uint32_t)(int32_t)(int8_t)b;
        }
    }
    __device__ __forceinline__ void quad(uint32_t k0, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const
    {
        if (k0 + 3u < nStream) {
            const uint32_t c0 = cell(k0), c3 = cell(k0 + 3u);
            if (c3 - c0 == 3u) {                                  // four neighbouring cells of one row
                GfU4 x;
                x.x = a; x.y = b; x.z = c; x.w = d;
                *reinterpret_cast<GfU4 *>(o + c0) = x;
                return;
            }
        }
        one(k0, a); one(k0 + 1u, b); one(k0 + 2u, c); one(k0 + 3u, d);
    }
};

// The Triangle predictor's inverse (PredictorModelTriangle.java:62-98) straight from the staged residuals (round 3): the value
// pass left every small residual of the stream as a byte in LDS (the others, and what the stage had no room for, at their cells
// in the tile, marked 0x80 in the stage), so the tile is written ONCE, whole rows by neighbouring lanes -- expand() + the in-place
// inverse wrote the residuals, read them back and wrote the values: 4.5 GB of HBM traffic per 0.93 GB of tiles on the
// ETOPO1-shaped batch (profiles/hbm_traffic.json, round 3 v1).
//   out[i][j] = res[i][j] + out[i][j-1] + out[i-1][j] - out[i-1][j-1]   (int32, wrapping)   unrolls to
//   out[i][j] = out[0][j] + out[i][0] - seed + SUM(r = 1..i) SUM(m = 1..j) res[r][m]:
// row 0 and column 0 are running sums of their residuals from the seed; the double sum is a row-wise prefix sum accumulated
// down the rows.  A lane owns FOUR neighbouring columns (nC <= 256: a wave spans a row): one 4-byte read of the stage, a prefix
// over its four values, a DPP scan over the lanes' sums, four accumulators that run down the rows, one 16-byte store.  The rows
// are dealt out in DEC_WAVES blocks, a wave each; the sums of the blocks above a wave's own come from a pre-pass -- column sums
// per block (the four bytes added up as two pairs of 16-bit fields), then ONE row-wise prefix sum per block -- through LDS.
// Row 0 and column 0 take part as rows / columns of zero residuals, so every wave stores whole rows.  Scratch: the lookup
// table (dead after the value pass).  (First version, a lane per column in 64-column segments: 14 K vector instructions per
// ETOPO1-shaped tile, as many as expand() + the inverse it replaced; this one: half of that.)
__device__ __forceinline__ bool cd_fuse_eligible(int model, uint32_t nR, uint32_t nC, uint32_t stageCap)
{
    return CD_NCUR == 1 && DEC_WAVES >= 2 && model == 3 && nR >= 2u && nC >= 2u && nC <= 256u && stageCap >= 8u &&
           (nR + DEC_WAVES - 1u) / DEC_WAVES <= 255u &&                       // rows per block: the 16-bit column sums of the pre-pass
           (uint32_t)(DEC_WAVES + 1) * nC + nR <= (1u << CD_LUT_BITS);
}
__device__ __forceinline__ void cd_fused_triangle(CanonDec &S, const CdCellSink &sink, uint32_t seed, uint32_t nR, uint32_t nC,
                                                  uint32_t *__restrict__ o)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = gf_wave_id();                           // (in an SGPR: the row loops and their addresses are scalar)
    uint32_t *row0 = S.lut, *col0 = row0 + nC, *T = col0 + nR;    // out[0][*], out[*][0], block sums [DEC_WAVES][nC]
    const uint32_t W = nC - 1u, H = nR - 1u, nHead = W + H;
    // residual of stream element k, whose cell is `cell`
    auto res = [&](uint32_t k, uint32_t cell) -> uint32_t {
        const uint32_t b = *sink.slot(min(k, sink.cap - 1u));
        uint32_t v = (uint32_t)(int32_t)(int8_t)b;
        if (b == 0x80u || k >= sink.cap) v = o[cell];             // (rare) wider than a byte, or beyond the stage
        return v;
    };
    // ---- the borders: running sums from the seed (wave 0: row 0, wave 1: column 0)
    if (wave < 2u) {
        const uint32_t n = wave == 0u ? W : H, k0 = wave == 0u ? 0u : W;
        uint32_t *dst = wave == 0u ? row0 : col0;
        uint32_t carry = seed;
        for (uint32_t c = 0; c < n; c += 64u) {
            const uint32_t e = c + lane;
            uint32_t v = 0;
            if (e < n) v = res(k0 + e, wave == 0u ? e + 1u : (e + 1u) * nC);
            v = gf_wave_incl_scan(v) + carry;
            if (e < n) dst[e + 1u] = v;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
        }
        if (lane == 0u) dst[0] = seed;
    }
    const uint32_t *stAw = reinterpret_cast<const uint32_t *>(sink.stA), *stBw = reinterpret_cast<const uint32_t *>(sink.stB);
    const uint32_t j0 = 4u * lane;                                // the lane's columns: j0 .. j0 + 3
    // the staged bytes of stream elements p0 + j0 .. + 3 (p0 wave-uniform); markers where the stage does not hold the four in
    // one piece (res() sorts those out)
    // (the second word of a read may lie one word behind the stage -- none of its bytes is used: the four bytes asked for end
    // inside the stage --: behind part A that is more of CanonDec, behind part B the word the callers keep out of the stage's
    // capacity for this, so every read stays inside the workgroup's LDS)
    auto fetch4 = [&](uint32_t p0) -> uint32_t {
        const uint32_t pEnd = p0 + 4u * 63u + 3u;                 // the last byte any lane asks for
        uint32_t x;
        if (pEnd < sink.capA || (p0 >= sink.capA && pEnd < sink.cap)) {         // (wave-uniform) one part of the stage holds them all
            const bool inA = pEnd < sink.capA;
            const uint32_t *w = inA ? stAw : stBw;
            const uint32_t off = (inA ? p0 : p0 - sink.capA) + j0;
            x = __builtin_amdgcn_alignbyte(w[(off >> 2) + 1u], w[off >> 2], off & 3u);
        } else {
            const uint32_t p = p0 + j0;
            const bool inA = p + 3u < sink.capA, inB = p >= sink.capA && p + 3u < sink.cap;
            const uint32_t *w = inA ? stAw : stBw;
            const uint32_t off = inA ? p : inB ? p - sink.capA : 0u;
            x = __builtin_amdgcn_alignbyte(w[(off >> 2) + 1u], w[off >> 2], off & 3u);
            x = (inA || inB) ? x : 0x80808080u;
        }
        return x;
    };
    auto hasMarker = [](uint32_t x) -> bool {                     // some byte is 0x80
        const uint32_t y = x ^ 0x80808080u;
        return ((y - 0x01010101u) & ~y & 0x80808080u) != 0u;
    };
    // ---- pre-pass: T[b][j] = SUM(rows r of block b) SUM(m = 1..j) res[r][m]
    const uint32_t RB = (nR + DEC_WAVES - 1u) / DEC_WAVES;
    const uint32_t r0 = min(nR, wave * RB), r1 = min(nR, r0 + RB);
    {
        uint32_t ev = 0, od = 0, corr[4] = {0u, 0u, 0u, 0u};      // bytes + 128 of columns j0, j0 + 2 / j0 + 1, j0 + 3 as 16-bit fields
        const uint32_t rFirst = max(r0, 1u);
        for (uint32_t r = rFirst; r < r1; r++) {
            const uint32_t kRow = nHead + (r - 1u) * W - 1u, cRow = r * nC;     // element of (r, j): kRow + j
            const uint32_t x = fetch4(kRow);
            if (hasMarker(x)) {                                   // (rare) the marker counts as -128 below
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t j = j0 + (uint32_t)i;
                    if (((x >> (8 * i)) & 0xffu) == 0x80u && j >= 1u && j < nC) corr[i] += res(kRow + j, cRow + j) + 128u;
                }
            }
            const uint32_t y = x ^ 0x80808080u;
            ev += y & 0x00FF00FFu;
            od += (y >> 8) & 0x00FF00FFu;
        }
        const uint32_t bias = r1 > rFirst ? 128u * (r1 - rFirst) : 0u;
        uint32_t c0 = (ev & 0xFFFFu) - bias + corr[0], c1 = (od & 0xFFFFu) - bias + corr[1];
        const uint32_t c2 = (ev >> 16) - bias + corr[2], c3 = (od >> 16) - bias + corr[3];
        if (lane == 0u) c0 = 0;                                   // column 0 has no residuals (the byte there is someone else's)
        const uint32_t p1 = c0 + c1, p2 = p1 + c2, p3 = p2 + c3;
        const uint32_t ex = gf_wave_incl_scan(p3) - p3;
        uint32_t *Tw = T + wave * nC + j0;
        if (j0 < nC) Tw[0] = ex + c0;
        if (j0 + 1u < nC) Tw[1] = ex + p1;
        if (j0 + 2u < nC) Tw[2] = ex + p2;
        if (j0 + 3u < nC) Tw[3] = ex + p3;
    }
    __syncthreads();
    // ---- the tile, row by row: acc = out[0][j] - seed + the double sum down to the row
    uint32_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t j = j0 + (uint32_t)i;
        uint32_t b = 0;
        if (j < nC) {
            b = row0[j] - seed;
            for (uint32_t w = 0; w < wave; w++) b += T[w * nC + j];
        }
        acc[i] = b;
    }
    for (uint32_t r = r0; r < r1; r++) {
        const uint32_t cRow = r * nC;
        const uint32_t left = col0[r];
        uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (r >= 1u) {
            const uint32_t kRow = nHead + (r - 1u) * W - 1u;
            const uint32_t x = fetch4(kRow);
            v0 = (uint32_t)(int32_t)(int8_t)x;
            v1 = (uint32_t)(int32_t)(int8_t)(x >> 8);
            v2 = (uint32_t)(int32_t)(int8_t)(x >> 16);
            v3 = (uint32_t)((int32_t)x >> 24);
            if (hasMarker(x)) {                                   // (rare)
                if ((x & 0xffu) == 0x80u && j0 >= 1u && j0 < nC) v0 = res(kRow + j0, cRow + j0);
                if (((x >> 8) & 0xffu) == 0x80u && j0 + 1u < nC) v1 = res(kRow + j0 + 1u, cRow + j0 + 1u);
                if (((x >> 16) & 0xffu) == 0x80u && j0 + 2u < nC) v2 = res(kRow + j0 + 2u, cRow + j0 + 2u);
                if ((x >> 24) == 0x80u && j0 + 3u < nC) v3 = res(kRow + j0 + 3u, cRow + j0 + 3u);
            }
            if (lane == 0u) v0 = 0;
        }
        const uint32_t p1 = v0 + v1, p2 = p1 + v2, p3 = p2 + v3;
        const uint32_t ex = gf_wave_incl_scan(p3) - p3;
        acc[0] += ex + v0;
        acc[1] += ex + p1;
        acc[2] += ex + p2;
        acc[3] += ex + p3;
        uint32_t *dst = o + cRow + j0;
        if (j0 + 3u < nC) {
            GfU4 q;
            q.x = acc[0] + left; q.y = acc[1] + left; q.z = acc[2] + left; q.w = acc[3] + left;
            *reinterpret_cast<GfU4 *>(dst) = q;
        } else {
            if (j0 < nC) dst[0] = acc[0] + left;
            if (j0 + 1u < nC) dst[1] = acc[1] + left;
            if (j0 + 2u < nC) dst[2] = acc[2] + left;
        }
    }
    __syncthreads();
}

// One canonical-Huffman stream (CanonicalHuffman.decode :441-519) starting at bit startBit of T, by the whole
// workgroup: code tables, subsequence synchronisation, then every value k handed to sink(k, value); values the
// text does not supply up to fillTo are handed over as 0.  More than maxValues values is the reference's
// ArrayIndexOutOfBounds.  Returns the tile status (same in all threads); *endPos = bit after the end-of-text symbol.
struct CdNoAfterPass {
    __device__ __forceinline__ void operator()(uint32_t, uint32_t) const {}
};
template <class Text, class Sink, class AfterPass = CdNoAfterPass>
__device__ __forceinline__ int32_t cd_decode_stream(CanonDec &S, const Text T, uint32_t startBit, uint32_t endBit,
                                                    uint32_t maxValues, uint32_t fillTo, Sink sink, uint32_t *endPos,
                                                    uint32_t *nValuesOut, uint32_t *stamps = nullptr,
                                                    const uint32_t *pre = nullptr, uint32_t preBase = 0, uint16_t *tok = nullptr,
                                                    int diagLimit = 0, const AfterPass afterPass = AfterPass())
{
    // diagLimit (diagnostic build only): return early after the tables (1), the synchronisation pass (2), the value pass (3)
    // tok: 4 KB of LDS for the token table of the synchronisation pass (LDS text only), or null: the cursor walk
#define CD_STAMP(i)                                                                        \
    do {                                                                                   \
        if (stamps && threadIdx.x == 0) stamps[i] = (uint32_t)__builtin_amdgcn_s_memtime(); \
    } while (0)
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)gf_wave_id();
    if (tid == 0) { S.parseStatus = GF_K_OK; S.runStatus = GF_K_OK; S.qStar = 0xFFFFFFFFu; S.carry = 0; }
    __syncthreads();
    // ---------------- phase 0: code tables ----------------
    if (pre) {
        // the code lengths were read by k_canon_parse_lengths (one lane per tile): pre = its record, preBase = the bit
        // position its packing-relative positions are counted from
        const uint8_t *pl = reinterpret_cast<const uint8_t *>(pre + 8);
        for (uint32_t e = (uint32_t)tid; e < 320u; e += DEC_THREADS) S.len[e] = e < 272u ? pl[e] : (uint8_t)0;
        if (tid == 0) { S.parseStatus = (int32_t)pre[0]; S.textStart = preBase + pre[1]; }
    } else if (wave == 0) {
        uint32_t pos = startBit + 1u;                           // reserved bit, CanonicalHuffman.java:451
        int32_t st = GF_K_OK;
        // LengthEncoder.readEncodedLengths :197-236
        {
            uint32_t k = 0, prior = 0;
            while (k < (uint32_t)CN_META && st == GF_K_OK) {
                if (pos + 12u > endBit + 32u) { st = GF_K_ERR_BOUNDS; break; }
                const uint32_t w = GF_UNI(cd_peek(T, pos));
                const uint32_t idx = w & 31u;
                pos += 5;
                if (idx <= 15u) {
                    if (lane == 0) S.metaLen[k] = (uint8_t)idx;
                    k++;
                    prior = idx;
                } else if (idx <= 18u) {
                    const uint32_t nb = idx == 16u ? 2u : idx == 17u ? 3u : 7u;
                    const uint32_t n = ((w >> 5) & ((1u << nb) - 1u)) + (idx == 18u ? 11u : 3u);
                    pos += nb;
                    if (idx != 16u) prior = 0;
                    if (k + n > (uint32_t)CN_META) { st = GF_K_ERR_BOUNDS; break; }       // symbols[k++] past the array
                    for (uint32_t j = (uint32_t)lane; j < n; j += 64) S.metaLen[k + j] = (uint8_t)prior;
                    k += n;
                }                                               // 19..31: ignored by the switch's default
                if (pos > endBit) st = GF_K_ERR_BOUNDS;
            }
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t nUsed = 0;
        if (st == GF_K_OK) {
            cd_tables<1>(S.metaLen, CN_META, S.mfirst, S.mcount, S.moffset, S.metaSym, lane, &nUsed);
            if (nUsed == 0) st = GF_K_ERR_BOUNDS;               // sortNodes[0] of an empty array
        }
        if (st == GF_K_OK) {
            for (uint32_t e = (uint32_t)lane; e < 256u; e += 64) {
                uint32_t cl;
                const uint32_t sym = cd_search(S.mfirst, S.mcount, S.moffset, S.metaSym, __brev(e), 1, 8, &cl);
                S.metaLut[e] = cl ? (uint16_t)((cl << 5) | sym) : 0;
            }
            for (uint32_t e = (uint32_t)lane; e < 320u; e += 64) S.len[e] = 0;
            __builtin_amdgcn_wave_barrier();
            // CanonHuffTreeDecoder.decodeTree :133-177
            uint32_t i = 0, prior = 0;
            while (i < (uint32_t)CN_SYMS && st == GF_K_OK) {
                if (pos >= endBit) { st = GF_K_ERR_BOUNDS; break; }
                const uint32_t w = GF_UNI(cd_peek(T, pos));
                uint32_t e = S.metaLut[w & 255u], cl, sym;
                if (e) { cl = e >> 5; sym = e & 31u; }
                else {
                    sym = cd_search(S.mfirst, S.mcount, S.moffset, S.metaSym, __brev(w), 9, 15, &cl);
                    if (cl == 0) { st = GF_K_ERR_BOUNDS; break; }  // walks into a missing node
                }
                sym = GF_UNI(sym);
                cl = GF_UNI(cl);
                pos += cl;
                if (sym <= 15u) {
                    if (lane == 0) S.len[i] = (uint8_t)sym;
                    i++;
                    prior = sym;
                } else if (sym <= 18u) {
                    const uint32_t nb = sym == 16u ? 2u : sym == 17u ? 3u : 7u;
                    const uint32_t n = ((w >> cl) & ((1u << nb) - 1u)) + (sym == 18u ? 11u : 3u);
                    pos += nb;
                    if (sym != 16u) prior = 0;
                    if (i + n > (uint32_t)CN_SYMS + 1u) { st = GF_K_ERR_BOUNDS; break; }   // int[N_SYMBOLS_TOTAL + 1]
                    for (uint32_t j = (uint32_t)lane; j < n; j += 64) S.len[i + j] = (uint8_t)prior;
                    i += n;
                } else {
                    i++;                                        // symbol 19 (meta end-of-text): no store, :170
                }
                if (pos > endBit) st = GF_K_ERR_BOUNDS;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) { S.parseStatus = st; S.textStart = pos; }
    }
    __syncthreads();
    if (S.parseStatus == GF_K_OK) {
        // (the lookup table's words are free until the table is built below)
        const uint32_t nUsed = cd_tables_wg(S.len, S.first, S.count, S.offset, S.symByOrder, S.lut);
        if (nUsed == 0) {
            __syncthreads();
            return GF_K_ERR_BOUNDS;
        }
    }
    CD_STAMP(1);                                  // code lengths read
    if (S.parseStatus != GF_K_OK) {
        const int32_t st = S.parseStatus;
        __syncthreads();
        return st;
    }
    {
        // The lookup table without a search per entry.  Canonical codes tile the code space in order of length: read most
        // significant bit first over CD_LUT_BITS bits, the codes of length l are the indices [B(l-1), B(l)) with
        // B(l) = (first[l] + count[l]) << (CD_LUT_BITS - l) = first[l + 1] << (CD_LUT_BITS - l - 1), B(0) = 0.  The bounds
        // and offset[l] - first[l] are wave-uniform, so the length of an entry is CD_LUT_BITS compare + select pairs on
        // scalar operands -- same entries as cd_search (the smallest length whose range holds the index), which read
        // first[] / count[] from LDS in a dependent loop for each of the eight entries of a thread.
        uint32_t bound[CD_LUT_BITS + 1], adj[CD_LUT_BITS + 1];
#pragma unroll
        for (int l = 1; l <= CD_LUT_BITS; l++) {
            bound[l] = GF_UNI((S.first[l] + S.count[l]) << (CD_LUT_BITS - l));
            adj[l] = GF_UNI(S.offset[l] - S.first[l]);
        }
        for (uint32_t e = tid; e < (1u << CD_LUT_BITS); e += DEC_THREADS) {
            const uint32_t m = __brev(e) >> (32 - CD_LUT_BITS);
            uint32_t cl = 0, at = 0;
#pragma unroll
            for (int l = CD_LUT_BITS; l >= 1; l--) {
                const bool in = m < bound[l];                       // bound[] never decreases: the last hit is the smallest l
                cl = in ? (uint32_t)l : cl;
                at = in ? adj[l] + (m >> (CD_LUT_BITS - l)) : at;
            }
            S.lut[e] = cl ? cd_entry(S.symByOrder[at], cl) : 0;
        }
    }
    __syncthreads();
    for (uint32_t x = tid; x < (1u << CD_LUT_BITS); x += DEC_THREADS) {     // pair up plain values that share a window
        const uint32_t e = S.lut[x];
        if (e && cd_e_sym(e) < 256u) {
            const uint32_t l1 = cd_e_len(e);
            const uint32_t e2 = S.lut[x >> l1];                               // the following bits, zero-extended
            if (e2 && cd_e_sym(e2) < 256u && l1 + cd_e_len(e2) <= (uint32_t)CD_LUT_BITS)
                S.lut[x] = (e & 0x7FFFu) | (cd_e_sym(e2) << 15) | ((l1 + cd_e_len(e2)) << 23) | 0x80000000u;
        }
    }
    __syncthreads();
    constexpr bool kLdsText = std::is_same<Text, CdTextLds>::value;
    if (!kLdsText) tok = nullptr;
    if (tok) {
        cd_build_tokens(S, tok);
        __syncthreads();
    }

    CD_STAMP(2);                                  // tables + LUT done
#ifdef GF_DIAG
    if (diagLimit == 1) { __syncthreads(); return GF_K_ERR_UNSUPPORTED; }
#endif
    // ---------------- phase 1: synchronise the subsequences, count their values ----------------
    const uint32_t T0 = S.textStart;
    const uint32_t span = endBit > T0 ? endBit - T0 : 1u;
    const uint32_t unit = max(128u, ((span + CD_MAXQ - 1) / CD_MAXQ + 31u) & ~31u);
    const uint32_t Q = (span + unit - 1) / unit;
    if (tok) {
        for (uint32_t q0 = 0; q0 < Q; q0 += DEC_THREADS) {      // (wave-uniform loops: every lane of a wave takes every turn)
            const uint32_t q = q0 + tid;
            const bool mine = q < Q;
            const uint32_t Bq = T0 + q * unit, Bn = min(endBit, Bq + unit);
            uint32_t a = Bq;
            if (mine && q > 0) a = Bq - T0 > CD_WARM ? Bq - CD_WARM : T0;      // warm-up: walk in from 128 bits before the boundary
            bool warm = mine && a < Bq;
            while (__any(warm)) {
                const uint32_t i = a >> 5;
                const uint32_t w = __builtin_amdgcn_alignbit(cdLdsText[i + 1u], cdLdsText[i], a);
                const uint32_t t = tok[w & ((1u << CD_LUT_BITS) - 1u)];
                uint32_t nb = a + (uint32_t)CD_LUT_BITS > Bq ? (t >> 9) & 31u : t & 31u;   // one token at a time close to the boundary
                bool jump = false;
                if (warm && (t & CD_TOK_STOP)) {
                    const CdTok tk = cd_token_from(cd_entry_of(S, w), w);
                    jump = tk.sym == 0xFFFFu || tk.sym == (uint32_t)CN_EOT;
                    nb = tk.bits;
                }
                a = !warm ? a : jump ? Bq : a + nb;
                warm = warm && a < Bq;
            }
            uint32_t e, c;
            cd_run_tok(S, tok, a, Bn == endBit ? 0xFFFFFFF0u : Bn, endBit, mine, &e, &c, &S.qx[mine ? q : 0]);
            if (mine) {
                S.qs[q] = a;
                S.qe[q] = e;
                S.qc[q] = c;
            }
        }
    } else
    for (uint32_t q = tid; q < Q; q += DEC_THREADS) {
        const uint32_t Bq = T0 + q * unit, Bn = min(endBit, Bq + unit);
        CdCur<Text> cur;
        if (q > 0) {                                            // warm-up: walk in from 128 bits before the boundary
            cur.seek(T, Bq - T0 > CD_WARM ? Bq - CD_WARM : T0);
            while (cur.pos < Bq) {
                const uint32_t w = cur.window();
                const uint32_t e = cd_entry_of(S, w);
                if (cd_e_pair(e) && cur.pos + cd_e_len(e) < Bq) { cur.advance(T, cd_e_len12(e)); continue; }
                const CdTok tk = cd_token_from(e, w);
                if (tk.sym == 0xFFFFu || tk.sym == (uint32_t)CN_EOT) { cur.seek(T, Bq); break; }
                cur.advance(T, tk.bits);
            }
        } else {
            cur.seek(T, Bq);
        }
        const uint32_t p = cur.pos;
        uint32_t e, c;
        cd_run(S, T, cur, Bn == endBit ? 0xFFFFFFF0u : Bn, endBit, &e, &c, &S.qx[q]);
        S.qs[q] = p;
        S.qe[q] = e;
        S.qc[q] = c;
    }
    __syncthreads();
    // Fix-up rounds: a subsequence whose start differs from its predecessor's end is decoded again from that end.  There are a
    // handful per tile, so they are LISTED (with the end they start from, as it stood when the round began) and dealt out to
    // the first lanes of wave 0 -- a wave that owned one of them used to run a whole pass for it while every other wave waited
    // at the barrier all the same.  The list (64 entries; what does not fit shows up again next round) lies over the meta
    // tables of phase 0.
    {
        constexpr uint32_t LIST_CAP = 64;
        uint32_t *listW = reinterpret_cast<uint32_t *>(S.metaLut);                    // 64 x 4 bytes
        uint16_t *listQ = reinterpret_cast<uint16_t *>(S.metaLut) + 2 * LIST_CAP;     // 64 x 2 bytes behind them
        static_assert(sizeof(S.metaLut) >= LIST_CAP * 6, "redo list does not fit over metaLut");
        for (uint32_t round = 0; round < (uint32_t)CD_MAXQ * 8u; round++) {
            if (tid == 0) S.changed = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < CD_NCUR; j++) {
                const uint32_t q = tid + j * DEC_THREADS;
                if (q >= 1 && q < Q) {
                    const uint32_t want = S.qe[q - 1];
                    if (want < CD_END_BAD && want != S.qs[q]) {
                        const uint32_t slot = atomicAdd(&S.changed, 1u);
                        if (slot < LIST_CAP) { listW[slot] = want; listQ[slot] = (uint16_t)q; }
                    }
                }
            }
            __syncthreads();
            const uint32_t nList = min(S.changed, LIST_CAP);
            if (nList == 0) break;
            if ((uint32_t)tid < nList) {
                const uint32_t q = listQ[tid], want = listW[tid];
                const uint32_t Bq = T0 + q * unit, Bn = min(endBit, Bq + unit);
                uint32_t e, c;
                // a value's escapes may carry the previous subsequence past this one's end: then it is empty
                if (want >= Bn && Bn != endBit) { e = want; c = 0; }
                else if (tok) cd_run_tok(S, tok, want, Bn == endBit ? 0xFFFFFFF0u : Bn, endBit, true, &e, &c, &S.qx[q]);
                else {
                    CdCur<Text> cur;
                    cur.seek(T, want);
                    cd_run(S, T, cur, Bn == endBit ? 0xFFFFFFF0u : Bn, endBit, &e, &c, &S.qx[q]);
                }
                S.qs[q] = want;
                S.qe[q] = e;
                S.qc[q] = c;
            }
            __syncthreads();
        }
    }
    CD_STAMP(3);                                  // synchronised
#ifdef GF_DIAG
    if (diagLimit == 2) { __syncthreads(); return GF_K_ERR_UNSUPPORTED; }
#endif
    // the true chain ends at the first subsequence that met the end-of-text symbol (or an error)
    for (uint32_t q = tid; q < Q; q += DEC_THREADS)
        if (S.qe[q] >= CD_END_BAD) atomicMin(&S.qStar, q);
    __syncthreads();
    const uint32_t qStar = S.qStar;
    int32_t tileStatus = GF_K_OK;
    if (qStar == 0xFFFFFFFFu || S.qe[qStar] != CD_END_EOT) tileStatus = GF_K_ERR_BOUNDS;   // no end-of-text: read past the data
    // exclusive prefix sum of the counts over the chain
    uint32_t base[CD_NCUR], myCount[CD_NCUR], firstHalf = 0;              // firstHalf: values of subsequences 0..DEC_THREADS-1 (uniform)
    {
        uint32_t running = 0;
#pragma unroll
        for (int j = 0; j < CD_NCUR; j++) {
            const uint32_t q = tid + j * DEC_THREADS;
            const uint32_t c = (q < Q && q <= qStar) ? S.qc[q] : 0u;
            myCount[j] = c;
            uint32_t tot;
            base[j] = running + block_excl_scan(c, S.waveSum, &tot);
            running += tot;
            if (j == 0) firstHalf = tot;
        }
        if (tid == 0) S.carry = running;
    }
    __syncthreads();
    const uint32_t nValues = S.carry;
    if (nValues > maxValues) tileStatus = GF_K_ERR_BOUNDS;      // text[iSymbol++] past the array
    if (tileStatus != GF_K_OK) {
        __syncthreads();
        return tileStatus;
    }

    CD_STAMP(4);                                  // counted
    // ---------------- phase 2: values to their sink ----------------
    // (a staged sink may lie over the sync arrays: with one subsequence per thread over all four of them -- what phase 2 needs
    // of them goes into registers first)
    const uint32_t eotEnd = S.qx[qStar];
    const bool plainOnly = GF_UNI((uint32_t)S.len[CN_NULL] | S.len[CN_ESC1] | S.len[CN_ESC2] | S.len[260]) == 0u;
    uint32_t myStart[CD_NCUR];
#pragma unroll
    for (int j = 0; j < CD_NCUR; j++) myStart[j] = tid + j * DEC_THREADS < Q ? S.qs[tid + j * DEC_THREADS] : 0u;
    if (CD_NCUR == 1) __syncthreads();
    uint32_t nPass = 1;
    if constexpr (Sink::kStaged) {
        sink.windowOn = sink.window != 0u && plainOnly && nValues == fillTo;
        if (sink.windowOn) nPass = (nValues + sink.window - 1u) / sink.window;
        if (sink.windowed && tid == 0) *sink.windowed = sink.windowOn ? 1u : 0u;
    }
    for (uint32_t pass = 0; pass < nPass; pass++)
#pragma unroll
    for (int j = 0; j < CD_NCUR; j++) {
        const uint32_t q = tid + j * DEC_THREADS;
        if constexpr (Sink::kStaged) sink.halfBase = sink.windowOn ? pass * sink.window : j == 0 ? 0u : min(firstHalf, nValues);
        {
            // One TOKEN per turn of a wave-uniform loop, everything by selects (round 3).  A value is PENDING from its symbol on until
            // the next token shows that no escape extends it any further (CanonicalHuffman.java:489-511): a plain symbol or the end
            // flushes it.  A subsequence owns the values whose symbol starts before its end; the escapes of its last value may carry
            // it past that end, and the escapes at its own start belong to the subsequence before (q == 0: an escape before any
            // value is text[-1]).  Two plain values that share a lookup go out together, the second one becomes the pending one.
            // (The loop used to be three nested per-lane loops -- values, escapes, pairs -- around a register cursor: the scalar
            // unit spent more instructions on their exec masks than the SIMDs on the values: 54 K SALU + 16 K branches per tile.)
            bool mine = q < Q && q <= qStar;
            if constexpr (Sink::kStaged) {
                // (a window's subsequences: those with a value in it)
                if (sink.windowOn) mine = mine && base[j] < sink.halfBase + sink.window && base[j] + myCount[j] > sink.halfBase;
            }
            const uint32_t Bq = T0 + q * unit, Bn = min(endBit, Bq + unit);
            const uint32_t bound = Bn == endBit ? 0xFFFFFFF0u : Bn;
            uint32_t k = base[j], a = mine ? (CD_NCUR == 1 ? myStart[j] : S.qs[q]) : 0u, v = 0;
            bool fin = !mine, pend = false, started = q != 0;
#ifndef GF_CD_NO_PLAIN_LOOP
            // A code without escapes, null and spare symbol (round 5; wave-uniform: the tile's code lengths say so): every symbol
            // is a complete value the moment it is read -- nothing pends, two values per lookup where the table pairs them.
            if (plainOnly) {
                while (__any(!fin)) {
                    const uint32_t w = cd_peek(T, a);
                    const uint32_t e = cd_entry_of(S, w);
                    const uint32_t sym = cd_e_sym(e), cl = cd_e_len(e);
                    const bool take = !fin && a < bound && e != 0x7FFFFFFFu && sym < 256u;   // (else: the end-of-text symbol, no code, the border)
                    const bool takePair = take && cd_e_pair(e) && a + cl < bound;
                    sink.put(k, sym - 128u, take);
                    sink.put(k + 1u, cd_e_sym2(e) - 128u, takePair);
                    k += (take ? 1u : 0u) + (takePair ? 1u : 0u);
                    a += takePair ? cd_e_len12(e) : take ? cl : 0u;
                    fin = fin || !take;
                }
            }
#endif
            while (__any(!fin)) {
                const uint32_t w = cd_peek(T, a);
                const uint32_t e = cd_entry_of(S, w);
                const bool inside = a < bound;
                const bool live = !fin && (inside || pend);
                const bool bad = e == 0x7FFFFFFFu;
                const uint32_t sym = cd_e_sym(e), cl = cd_e_len(e), extra = cd_e_extra(e);
                const bool isPlain = !bad && sym <= (uint32_t)CN_NULL, isEnd = bad || sym == (uint32_t)CN_EOT;
                const bool isEsc = !bad && (sym == (uint32_t)CN_ESC1 || sym == (uint32_t)CN_ESC2);
                const bool takePlain = live && isPlain && inside;
                const bool stop = live && (isEnd || (isPlain && !inside));
                const bool takeEsc = live && !isPlain && !isEnd;                 // an escape or the spare symbol 260
                const bool flush = pend && (takePlain || stop);
                sink.put(k, v, flush);
                k += flush ? 1u : 0u;
                const bool takePair = takePlain && cd_e_pair(e) && a + cl < bound;
                sink.put(k, sym - 128u, takePair);                                // complete: a value follows it, not an escape
                k += takePair ? 1u : 0u;
                if (takeEsc) {
                    const uint32_t raw = (w >> cl) & ((1u << extra) - 1u);
                    if (pend && isEsc) v = sym == (uint32_t)CN_ESC2 ? (v << 2) | raw : (v << 8) | raw;
                    if (!pend && !started && sym != 260u) S.runStatus = GF_K_ERR_BOUNDS;
                }
                if (takePlain) {
                    v = takePair ? cd_e_sym2(e) - 128u : sym == (uint32_t)CN_NULL ? GF_NULL_CODE : sym - 128u;
                    started = true;
                }
                pend = (pend || takePlain) && !stop;
                a += takePlain ? (takePair ? cd_e_len12(e) : cl) : takeEsc ? cl + extra : 0u;
                fin = fin || stop || !live;
            }
        }
        if constexpr (Sink::kStaged) {             // this half's small values wait in LDS: out with them, whole lines at a time
            __syncthreads();
            if (sink.windowOn) afterPass(sink.halfBase, min(sink.window, nValues - sink.halfBase));
            else if (!sink.fuse) sink.expand(j == 0 ? min(firstHalf, nValues) : nValues - min(firstHalf, nValues));
            __syncthreads();
        }
    }
    // a text shorter than its reader expects leaves zeros (fresh int[] in Java)
    for (uint32_t k = nValues + tid; k < fillTo; k += DEC_THREADS) {
        if constexpr (Sink::kStaged) {
            if (sink.fuse) sink.put(k, 0u);                       // (one subsequence per thread: halfBase is 0)
            else sink.one(k, 0u);
        } else sink.one(k, 0u);
    }
    __syncthreads();
    {
        const int32_t st = S.runStatus;
        __syncthreads();
        if (st != GF_K_OK) return st;
    }

    CD_STAMP(5);                                  // values written
#ifdef GF_DIAG
    if (diagLimit == 3) { __syncthreads(); return GF_K_ERR_UNSUPPORTED; }
#endif
    *endPos = eotEnd;
    *nValuesOut = nValues;
    __syncthreads();
    return GF_K_OK;
#undef CD_STAMP
}

struct __attribute__((packed, aligned(1))) CdPackedWord { uint32_t v; };     // a word of a packing at any byte address

// The code lengths of one canonical stream by ONE LANE (LengthEncoder.readEncodedLengths :197-236, CanonHuffTreeDecoder.decodeTree
// :133-177): the serial walk of k_canon_parse_lengths and k_lsop_head.  peek(pos) = 32 bits of the packing from bit pos; sMetaLen /
// sOrder / sLut: per-lane columns in LDS (CN_META, CN_META and 128 bytes per lane, element k of lane l at [k * 64 + l]); outLen: the
// record's 261 lengths, zero beforehand -- only non-zero lengths are stored, by `writer` lanes.  st: the lane's status so far (a lane
// that has failed before takes no part but stays in the wave's loops).  Same checks and statuses as phase 0 of cd_decode_stream.
template <class Peek>
__device__ __forceinline__ int32_t cd_lane_parse_lengths(const Peek &peek, int32_t st, uint32_t startBit, uint32_t endBit, uint32_t lane,
                                                         uint8_t *sMetaLen, uint8_t *sOrder, uint8_t *sLut, uint8_t *outLen, bool writer,
                                                         uint32_t *posOut, uint32_t *nonZeroOut)
{
    uint32_t pos = startBit + 1u;                             // reserved bit, CanonicalHuffman.java:451
    // LengthEncoder.readEncodedLengths :197-236: the 20 lengths of the meta alphabet
    for (uint32_t k = 0; k < (uint32_t)CN_META; k++) sMetaLen[k * 64 + lane] = 0;
    {
        uint32_t k = 0, prior = 0;
        while (k < (uint32_t)CN_META && st == GF_K_OK) {
            if (pos + 12u > endBit + 32u) { st = GF_K_ERR_BOUNDS; break; }
            const uint32_t w = peek(pos);
            const uint32_t idx = w & 31u;
            pos += 5;
            if (idx <= 15u) {
                sMetaLen[k * 64 + lane] = (uint8_t)idx;
                k++;
                prior = idx;
            } else if (idx <= 18u) {
                const uint32_t nb = idx == 16u ? 2u : idx == 17u ? 3u : 7u;
                const uint32_t n = ((w >> 5) & ((1u << nb) - 1u)) + (idx == 18u ? 11u : 3u);
                pos += nb;
                if (idx != 16u) prior = 0;
                if (k + n > (uint32_t)CN_META) { st = GF_K_ERR_BOUNDS; break; }
                for (uint32_t j = 0; j < n; j++) sMetaLen[(k + j) * 64 + lane] = (uint8_t)prior;
                k += n;
            }
            if (pos > endBit) st = GF_K_ERR_BOUNDS;
        }
    }
    // canonical tables of the meta code: symbols per length (8-bit counters packed in two words), symbols ordered by
    // (length, symbol)
    unsigned long long cntLo = 0, cntHi = 0;
    auto cnt8 = [&](uint32_t l) -> uint32_t { return (uint32_t)((l < 8 ? cntLo >> (8u * l) : cntHi >> (8u * (l - 8u))) & 0xffu); };
    uint32_t nUsed = 0;
    if (st == GF_K_OK) {
        for (uint32_t k = 0; k < (uint32_t)CN_META; k++) {
            const uint32_t l = sMetaLen[k * 64 + lane];
            if (l) {
                if (l < 8) cntLo += 1ull << (8u * l); else cntHi += 1ull << (8u * (l - 8u));
                nUsed++;
            }
        }
        if (nUsed == 0) st = GF_K_ERR_BOUNDS;                 // sortNodes[0] of an empty array
        // (a counting sort: where each length's symbols start, then the symbols in order -- a scan of the twenty per length was 300
        // turns)
        if (st == GF_K_OK) {
            unsigned long long atLo = 0, atHi = 0;            // first slot of length l, 8-bit fields as cntLo / cntHi
            uint32_t run = 0;
            for (uint32_t l = 1; l <= 15; l++) {
                if (l < 8) atLo |= (unsigned long long)run << (8u * l); else atHi |= (unsigned long long)run << (8u * (l - 8u));
                run += cnt8(l);
            }
            for (uint32_t k = 0; k < (uint32_t)CN_META; k++) {
                const uint32_t l = sMetaLen[k * 64 + lane];
                if (l) {
                    const uint32_t slot = (uint32_t)((l < 8 ? atLo >> (8u * l) : atHi >> (8u * (l - 8u))) & 0xffu);
                    sOrder[slot * 64 + lane] = (uint8_t)k;
                    if (l < 8) atLo += 1ull << (8u * l); else atHi += 1ull << (8u * (l - 8u));
                }
            }
        }
    }
    // The meta code's first-level table (round 4): the next seven bits of the stream -> (length << 5) | symbol for the codes of up to
    // seven bits, 0 for the longer ones (which take the canonical search below).  A lane's column of 128 bytes; canonical codes in
    // order of (length, symbol), first bit of a code = first bit of the stream = bit 0 of the index.
    // (Only for lengths that make a prefix code, Kraft sum <= 1 -- those a CanonicalHuffman encoder writes.  Damaged lengths that
    // over-subscribe the code space leave the table empty: the search alone decides what such a stream decodes to.)
    if (st == GF_K_OK) {
        for (uint32_t x = 0; x < 128u; x++) sLut[x * 64 + lane] = 0;
        uint32_t kraft = 0;
        for (uint32_t l = 1; l <= 15; l++) kraft += cnt8(l) << (15u - l);
        uint32_t code = 0, slot = 0;
        for (uint32_t l = 1; l <= 7 && kraft <= (1u << 15); l++) {
            const uint32_t n = cnt8(l);
            for (uint32_t d = 0; d < n; d++) {
                const uint32_t sym = sOrder[(slot + d) * 64 + lane];
                const uint32_t r = __brev(code + d) >> (32u - l);          // the code as the stream holds it
                const uint8_t e = (uint8_t)((l << 5) | sym);
                for (uint32_t x = r; x < 128u; x += 1u << l) sLut[x * 64 + lane] = e;
            }
            code = (code + n) << 1;
            slot += n;
        }
    }
    // CanonHuffTreeDecoder.decodeTree :133-177: the 260 (+1) code lengths
    // (round 4: one turn per token with its kinds -- a length, a run of the prior length, a run of zeros, the end-of-text symbol --
    // under predicates instead of branches; the lanes of the wave leave the loop together.  Every way out with an error is
    // GF_K_ERR_BOUNDS, and nothing is stored in the turn that finds one.)
    uint32_t nonZero = 0;
    {
        uint32_t i = 0, prior = 0;
        bool bad = false;
        for (;;) {
            const bool live = st == GF_K_OK && !bad && i < (uint32_t)CN_SYMS;
            if (!__any(live)) break;
            const bool atEnd = pos >= endBit;
            const uint32_t w = peek(pos);
            const uint32_t e = sLut[(w & 127u) * 64 + lane];
            uint32_t cl = e >> 5, sym = e & 31u;
            if (live && !atEnd && e == 0u) {                  // a code of more than seven bits (or none): the canonical search (cd_search)
                const uint32_t c = __brev(w);
                uint32_t code = 0, offs = 0;
                for (uint32_t l = 1; l <= 15; l++) {
                    const uint32_t n = cnt8(l);
                    const uint32_t d = (c >> (32u - l)) - code;
                    if (d < n) { cl = l; sym = sOrder[(offs + d) * 64 + lane]; break; }
                    code = (code + n) << 1;
                    offs += n;
                }
            }
            const bool found = live && !atEnd && cl != 0u;    // (cl == 0: the stream walks into a missing node)
            const bool isLit = sym <= 15u, isRun = sym - 16u < 3u;
            const uint32_t nb = sym == 16u ? 2u : sym == 17u ? 3u : 7u;
            const uint32_t n = isRun ? ((w >> cl) & ((1u << nb) - 1u)) + (sym == 18u ? 11u : 3u) : 1u;
            const bool fits = !isRun || i + n <= (uint32_t)CN_SYMS + 1u;
            const bool take = found && fits;
            const uint32_t val = isLit ? sym : (sym == 16u ? prior : 0u);         // symbol 19 (meta end-of-text): no store
            const uint32_t nStore = (take && val) ? n : 0u;
            if (nStore && writer) {
                outLen[i] = (uint8_t)val;
                for (uint32_t j = 1; j < nStore; j++) outLen[i + j] = (uint8_t)val;
            }
            nonZero += nStore;
            pos += found ? cl + (isRun ? nb : 0u) : 0u;
            prior = !take ? prior : isLit ? sym : (sym == 17u || sym == 18u) ? 0u : prior;
            i += take ? n : 0u;
            bad = bad || (live && (!take || pos > endBit));
        }
        if (bad) st = GF_K_ERR_BOUNDS;
    }
    *posOut = pos;
    *nonZeroOut = nonZero;
    return st;
}
