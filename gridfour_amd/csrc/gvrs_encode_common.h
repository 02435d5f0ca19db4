// gvrs_encode_common.h -- device helpers shared by the encode kernels (legacy Huffman, canonical
// Huffman, LSOP): workgroup scans, the 8-cell flat-scan loads with the three predictors' residuals, the
// in-register bitonic sort, the LDS bit window.  Include inside the kernel file's anonymous namespace.
// Reference paths are relative to core/src/main/java/org/gridfour/.
#pragma once

constexpr int ENC_THREADS = GF_ENC_THREADS;
constexpr int ENC_WAVES = GF_ENC_WAVES;
constexpr int HIST_R = 4;                       // histogram replicas
constexpr int IMG_WORDS = GF_IMG_WORDS;
constexpr int WIN_WORDS = ENC_WAVES < 4 ? 1024 * ENC_WAVES : 4096;   // bit-pack window: 4 KB per wave, 16 KB at most
constexpr int WIN_SLACK = 8;
constexpr int CPT = 8;                          // cells per thread per step of the flat scans
constexpr uint32_t STEP_CELLS = ENC_THREADS * CPT;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
    (void)lane;
    return gf_wave_incl_scan(v);
}

// exclusive scan over the workgroup; *total = sum over all threads
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *waveSum, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = (int)gf_wave_id();
    uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) waveSum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < ENC_WAVES; w++) {
        uint32_t s = waveSum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// residual of one cell for a model (generic form, used by the border segments and fallbacks)
__device__ __forceinline__ uint32_t cell_residual(int model, const uint32_t *__restrict__ tile, uint32_t nC,
                                                  uint32_t idx, uint32_t r, uint32_t c, uint32_t seed)
{
    const uint32_t v = tile[idx];
    switch (model) {
    case 1: return v - (c > 0 ? tile[idx - 1] : tile[idx - nC]);
    case 2:
        if (c >= 2) return v - (2u * tile[idx - 1] - tile[idx - 2]);
        return v - (c == 1 ? tile[idx - 1] : tile[idx - nC]);
    case 3:
        if (r == 0) return v - tile[idx - 1];
        if (c == 0) return v - tile[idx - nC];
        return v - (tile[idx - 1] + tile[idx - nC] - tile[idx - nC - 1]);
    default: {
        // PredictorModelDifferencingWithNulls.java:109-131: prior = left neighbour, or the
        // first cell of the previous row at a row start; the seed replaces a null prior.
        if (v == GF_NULL_CODE) return GF_NULL_CODE;
        uint32_t prior;
        if (c > 0) prior = tile[idx - 1];
        else prior = r > 0 ? tile[idx - nC] : GF_NULL_CODE;
        if (prior == GF_NULL_CODE) prior = seed;
        return v - prior;
    }
    }
}

// 8 consecutive cells starting at flat index i0 plus what their predictors need
struct Cells8 {
    uint32_t cur[CPT], up[CPT], wm1, wm2, upm1;
};

__device__ __forceinline__ void load_cells8(const uint32_t *__restrict__ tile, uint32_t nC, uint32_t nCells,
                                            uint32_t i0, Cells8 &Q)
{
    // every value lands in a scalar first and the struct is filled after the two paths have joined: with the members
    // assigned inside the branches the compiler kept three of them in a stack slot (12 bytes of scratch per lane)
    uint32_t c0, c1, c2, c3, c4, c5, c6, c7, u0, u1, u2, u3, u4, u5, u6, u7, wm1, wm2, upm1;
    if (i0 >= nC + 2 && i0 + (CPT - 1) < nCells) {       // interior: every word exists
        const GfU4 a = *reinterpret_cast<const GfU4 *>(tile + i0);
        const GfU4 b = *reinterpret_cast<const GfU4 *>(tile + i0 + 4);
        const GfU4 c = *reinterpret_cast<const GfU4 *>(tile + (i0 - nC));
        const GfU4 d = *reinterpret_cast<const GfU4 *>(tile + (i0 - nC) + 4);
        wm1 = tile[i0 - 1];
        wm2 = tile[i0 - 2];
        upm1 = tile[i0 - nC - 1];
        c0 = a.x; c1 = a.y; c2 = a.z; c3 = a.w; c4 = b.x; c5 = b.y; c6 = b.z; c7 = b.w;
        u0 = c.x; u1 = c.y; u2 = c.z; u3 = c.w; u4 = d.x; u5 = d.y; u6 = d.z; u7 = d.w;
    } else {
        auto cur = [&](uint32_t j) -> uint32_t { return i0 + j < nCells ? tile[i0 + j] : 0u; };
        auto up = [&](uint32_t j) -> uint32_t { return (i0 + j >= nC && i0 + j < nCells) ? tile[i0 + j - nC] : 0u; };
        c0 = cur(0); c1 = cur(1); c2 = cur(2); c3 = cur(3); c4 = cur(4); c5 = cur(5); c6 = cur(6); c7 = cur(7);
        u0 = up(0); u1 = up(1); u2 = up(2); u3 = up(3); u4 = up(4); u5 = up(5); u6 = up(6); u7 = up(7);
        wm1 = (i0 >= 1 && i0 - 1 < nCells) ? tile[i0 - 1] : 0u;
        wm2 = (i0 >= 2 && i0 - 2 < nCells) ? tile[i0 - 2] : 0u;
        upm1 = (i0 >= nC + 1 && i0 - nC - 1 < nCells) ? tile[i0 - nC - 1] : 0u;
    }
    Q.cur[0] = c0; Q.cur[1] = c1; Q.cur[2] = c2; Q.cur[3] = c3; Q.cur[4] = c4; Q.cur[5] = c5; Q.cur[6] = c6; Q.cur[7] = c7;
    Q.up[0] = u0; Q.up[1] = u1; Q.up[2] = u2; Q.up[3] = u3; Q.up[4] = u4; Q.up[5] = u5; Q.up[6] = u6; Q.up[7] = u7;
    Q.wm1 = wm1;
    Q.wm2 = wm2;
    Q.upm1 = upm1;
}

// The same for a WAVE whose lanes hold consecutive blocks of eight cells (lane l: i0 = i0 of lane 0 + 8 l -- the flat scans).
// Round 6: the three words in front of a block (W, WW and NW of its first cell) are the last words of the block of the lane before:
// a DPP move each (wave_shr:1); lane 0 alone loads its three.  As three dword loads per lane -- 64 addresses 32 bytes apart, sixteen
// cache lines per instruction -- they cost the memory pipeline more than the four 16-byte loads of the block itself
// (tools/bw_probe.hip: 0.208 -> 0.167 ms for phase A's loads over the bench batch; k_huffman_encode<true, 1> 0.323 -> 0.289 ms).
// (Lane 0's words through the scalar cache instead: the wait for them is a wait for every LDS operation in flight as well -- one
// counter --, which cost the rough batch's phase A, whose turns queue more histogram additions, 0.027 ms.)
__device__ __forceinline__ void load_cells8_wave(const uint32_t *__restrict__ tile, uint32_t nC, uint32_t nCells, uint32_t i0, Cells8 &Q)
{
    const bool interior = i0 >= nC + 2 && i0 + (CPT - 1) < nCells;
    if (__all(interior)) {
        const GfU4 a = *reinterpret_cast<const GfU4 *>(tile + i0);
        const GfU4 b = *reinterpret_cast<const GfU4 *>(tile + i0 + 4);
        const GfU4 c = *reinterpret_cast<const GfU4 *>(tile + (i0 - nC));
        const GfU4 d = *reinterpret_cast<const GfU4 *>(tile + (i0 - nC) + 4);
        uint32_t s1 = 0, s2 = 0, s3 = 0;
        if ((threadIdx.x & 63u) == 0u) {
            s1 = tile[i0 - 1];
            s2 = tile[i0 - 2];
            s3 = tile[i0 - nC - 1];
        }
        Q.cur[0] = a.x; Q.cur[1] = a.y; Q.cur[2] = a.z; Q.cur[3] = a.w; Q.cur[4] = b.x; Q.cur[5] = b.y; Q.cur[6] = b.z; Q.cur[7] = b.w;
        Q.up[0] = c.x; Q.up[1] = c.y; Q.up[2] = c.z; Q.up[3] = c.w; Q.up[4] = d.x; Q.up[5] = d.y; Q.up[6] = d.z; Q.up[7] = d.w;
        Q.wm1 = (uint32_t)__builtin_amdgcn_update_dpp((int)s1, (int)b.w, 0x138, 0xf, 0xf, false);      // wave_shr:1; lane 0 keeps its own
        Q.wm2 = (uint32_t)__builtin_amdgcn_update_dpp((int)s2, (int)b.z, 0x138, 0xf, 0xf, false);
        Q.upm1 = (uint32_t)__builtin_amdgcn_update_dpp((int)s3, (int)d.w, 0x138, 0xf, 0xf, false);
    } else {
        load_cells8(tile, nC, nCells, i0, Q);
    }
}

// Bitonic sort, ascending, of the 64*NREG 32-bit keys held in k[0..NREG) (element e = r*64 + lane).
template <int NREG, class Arr>
__device__ __forceinline__ void wave_bitonic_sort(Arr &k, int lane)
{
#pragma unroll
    for (int size = 2; size <= 64 * NREG; size <<= 1) {
#pragma unroll
        for (int j = size >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int rj = j >> 6;           // partner lives in another register of the same lane
#pragma unroll
                for (int r = 0; r < NREG; r++) {
                    if ((r & rj) == 0) {
                        const int r2 = r | rj;
                        const bool asc = ((r * 64) & size) == 0;
                        const uint32_t lo = min(k[r], k[r2]), hi = max(k[r], k[r2]);
                        k[r] = asc ? lo : hi;
                        k[r2] = asc ? hi : lo;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < NREG; r++) {
                    const uint32_t other = gf_lane_xor(k[r], j);
                    const bool asc = (((r * 64) | lane) & size) == 0;
                    const bool lower = (lane & j) == 0;
                    k[r] = (lower == asc) ? min(k[r], other) : max(k[r], other);
                }
            }
        }
    }
}

struct BitSink {
    uint32_t *win;
    uint32_t acc;       // bits of the current window word
    uint32_t nacc;      // valid bits in acc (< 32 between puts)
    uint32_t wp;        // window word acc belongs to
    bool first;         // the first word is shared with the previous thread

    __device__ __forceinline__ void init(uint32_t *w, uint32_t bitPos)
    {
        win = w;
        wp = bitPos >> 5;
        nacc = bitPos & 31u;
        acc = 0;
        first = true;
    }
    // len <= 32; (code, len) = (0, 0) is a no-op.  32-bit arithmetic only on the common path (64-bit shifts run at
    // a quarter of the rate): the bits that do not fit the current word are recovered in the (rarer) flush branch.
    __device__ __forceinline__ void put32(uint32_t code, uint32_t len)
    {
        acc |= code << nacc;                   // nacc < 32
        nacc += len;
        if (nacc >= 32u) {
            if (first) { atomicOr(&win[wp], acc); first = false; }
            else win[wp] = acc;
            wp++;
            nacc -= 32u;                       // bits of `code` that spilled into the next word
            acc = nacc ? code >> (len - nacc) : 0u;
        }
    }
    __device__ __forceinline__ void put(uint64_t code, uint32_t len)
    {
        if (len > 32) {
            put32((uint32_t)code, 32);
            put32((uint32_t)(code >> 32), len - 32);
        } else {
            put32((uint32_t)code, len);
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (nacc > 0) atomicOr(&win[wp], acc);
    }
};

// Eight (code, length) pairs whose two groups of four are at most 32 bits each, joined in registers (a tree of 32-bit shifts
// and ORs, no branch) and ORed into a zeroed bit window at bit position pbit as three words.  What the packers' usual lane does
// instead of eight BitSink::put calls with a branch or two each (a word filled up? the first word, shared with the lane
// before?) -- the scalar unit spent more instructions on those than the SIMDs on the codes (round 3).  Same bits at the same
// places.  A shift count of 32 only ever meets a zero operand: the codes behind a group that is full have no bits.  The window
// keeps two words behind what a wave may fill (WAVE_WIN_BITS).
#define GF_JOIN8_OR(wwin, pbit, CD, LN, n01, n45, n0)                                                              \
    do {                                                                                                            \
        const uint32_t p01_ = CD(0) | (CD(1) << (LN(0) & 31u)), p23_ = CD(2) | (CD(3) << (LN(2) & 31u));            \
        const uint32_t p45_ = CD(4) | (CD(5) << (LN(4) & 31u)), p67_ = CD(6) | (CD(7) << (LN(6) & 31u));            \
        const uint32_t q0_ = p01_ | (p23_ << ((n01) & 31u)), q1_ = p45_ | (p67_ << ((n45) & 31u));                  \
        const uint32_t lo_ = q0_ | ((n0) < 32u ? q1_ << (n0) : 0u);                                                 \
        const uint32_t hi_ = (n0) ? q1_ >> ((32u - (n0)) & 31u) : 0u;                                               \
        const uint32_t w_ = (pbit) >> 5, o_ = (pbit) & 31u, ro_ = (32u - o_) & 31u;                                 \
        atomicOr(&(wwin)[w_], lo_ << o_);                                                                           \
        atomicOr(&(wwin)[w_ + 1u], o_ ? (lo_ >> ro_) | (hi_ << o_) : hi_);                                          \
        atomicOr(&(wwin)[w_ + 2u], o_ ? hi_ >> ro_ : 0u);                                                           \
    } while (0)

// Two (code, length) pairs of at most 64 bits together (either may be empty), joined in a 64-bit register and ORed into a zeroed
// bit window at bit position pbit as three words.  A shift count of 64 or 32 only ever meets a guarded operand.
#define GF_JOIN2_OR64(wwin, pbit, ca, la, cb, lb)                                                                   \
    do {                                                                                                            \
        const uint64_t p_ = (ca) | ((la) < 64u ? (cb) << (la) : 0ull);                                              \
        const uint32_t w_ = (pbit) >> 5, o_ = (pbit) & 31u;                                                         \
        const uint64_t lo_ = p_ << o_;                                                                              \
        const uint32_t hi_ = o_ ? (uint32_t)(p_ >> (64u - o_)) : 0u;                                                \
        atomicOr(&(wwin)[w_], (uint32_t)lo_);                                                                       \
        atomicOr(&(wwin)[w_ + 1u], (uint32_t)(lo_ >> 32));                                                          \
        atomicOr(&(wwin)[w_ + 2u], hi_);                                                                            \
    } while (0)

// residual and emit mask of cell j of a Cells8 block for one model (flat-scan form)
template <int MODEL>
__device__ __forceinline__ uint32_t flat_residual(const Cells8 &Q, int j, uint32_t idx, uint32_t c, uint32_t nC,
                                                  uint32_t nCells, uint32_t seed, bool *emit)
{
    const uint32_t v = Q.cur[j];
    const uint32_t W = j > 0 ? Q.cur[j - 1] : Q.wm1;
    if (MODEL == 1) {
        *emit = idx >= 1 && idx < nCells;
        return v - (c > 0 ? W : Q.up[j]);                               // PredictorModelDifferencing.java:120-137
    } else if (MODEL == 2) {
        const uint32_t WW = j > 1 ? Q.cur[j - 2] : (j == 1 ? Q.wm1 : Q.wm2);
        *emit = c >= 2 && idx < nCells;
        return v - (2u * W - WW);                                       // PredictorModelLinear.java:128-141
    } else if (MODEL == 3) {
        const uint32_t NW = j > 0 ? Q.up[j - 1] : Q.upm1;
        *emit = idx >= nC && c >= 1 && idx < nCells;
        return v - (W + Q.up[j] - NW);                                  // PredictorModelTriangle.java:130-142
    } else {
        *emit = idx < nCells;
        uint32_t prior = c > 0 ? W : (idx >= nC ? Q.up[j] : GF_NULL_CODE);
        if (prior == GF_NULL_CODE) prior = seed;                        // ...DifferencingWithNulls.java:109-131
        return v == GF_NULL_CODE ? GF_NULL_CODE : v - prior;
    }
}

struct PackState {
    uint32_t bitBase;    // next free bit of the packing (absolute)
    uint32_t wordBase;   // words already flushed to the output slot
};

// moves the completed words of the window to the output slot and slides the window
// (call after a barrier that ends the emission into the window)
__device__ __forceinline__ void window_flush(uint32_t *win, uint32_t *__restrict__ out32, PackState &ps)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t fullWords = (ps.bitBase >> 5) - ps.wordBase;
    for (uint32_t j = tid; j < fullWords; j += ENC_THREADS) out32[ps.wordBase + j] = win[j];
    const uint32_t partial = win[fullWords];
    __syncthreads();
    for (uint32_t j = tid; j <= fullWords; j += ENC_THREADS) win[j] = j == 0 ? partial : 0u;
    ps.wordBase += fullWords;
    __syncthreads();
}

// ---- wave-private bit windows (the pack kernels) ----------------------------------------------------------------------
// Every wave packs a contiguous share of a cell range into its own quarter of the LDS window, from bit 0, with a wave-level
// scan and no workgroup barrier inside the loop; the four bit strings follow one another in the stream and are shifted into
// place at the end.  wave_windows_begin saves the stream's partial last word (win[0], the convention of window_flush) and
// clears the windows; wave_windows_end takes every wave's bit count (fits = false: its share did not fit), and either
// restores the window and returns false (nothing written, ps untouched: the caller packs the range the old way) or writes
// the full words, leaves the new partial word in win[0], advances ps and returns true.
// the whole window (and its slack) to zero, sixteen bytes per store (the window is 16-byte aligned: see the kernels' declarations)
__device__ __forceinline__ void window_clear(uint32_t *win)
{
    static_assert((WIN_WORDS + WIN_SLACK) % 4 == 0, "the window is cleared in 16-byte pieces");
    uint4 *w4 = reinterpret_cast<uint4 *>(win);
    for (uint32_t i = threadIdx.x; i < (uint32_t)(WIN_WORDS + WIN_SLACK) / 4u; i += ENC_THREADS) w4[i] = make_uint4(0u, 0u, 0u, 0u);
}

constexpr uint32_t WAVE_WIN = WIN_WORDS / ENC_WAVES;                 // words per wave
constexpr uint32_t WAVE_WIN_BITS = (WAVE_WIN - 2u) * 32u;            // what a wave may put into its window

__device__ __forceinline__ uint32_t wave_windows_begin(uint32_t *win, uint32_t *waveSum)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t carryWord = win[0];                               // bits of the stream so far in its last, partial word
    __syncthreads();
    window_clear(win);
    if (tid < ENC_WAVES) waveSum[tid] = 0xFFFFFFFFu;                 // = this wave's share did not fit
    __syncthreads();
    return carryWord;
}

__device__ __forceinline__ bool wave_windows_end(uint32_t *win, uint32_t *waveSum, uint32_t carryWord, uint32_t bits, bool fits,
                                                 uint32_t *__restrict__ out32, uint32_t slotWords, PackState &ps)
{
    // (the thread index worked out afresh -- lane by mbcnt, wave from the scalar -- instead of kept alive across the caller's
    // scan loop, where the register allocator had no room for it at 64 VGPRs and parked it in scratch)
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), wave = gf_wave_id();
    const uint32_t tid = wave * 64u + lane;
    if (lane == 0 && fits) waveSum[wave] = bits;
    __syncthreads();
    uint32_t L[ENC_WAVES];
    bool all = true;
#pragma unroll
    for (int w = 0; w < ENC_WAVES; w++) {
        L[w] = waveSum[w];
        all = all && L[w] != 0xFFFFFFFFu;
    }
    if (!all) {                                                       // back to the state the caller left: partial word, zeros
        __syncthreads();
        for (uint32_t i = tid; i < (uint32_t)(WIN_WORDS + WIN_SLACK); i += ENC_THREADS) win[i] = i == 0 ? carryWord : 0u;
        __syncthreads();
        return false;
    }
    // concatenate: the bit string of wave w lands at bit D[w] of the packing
    uint32_t D[ENC_WAVES + 1];
    D[0] = ps.bitBase;
#pragma unroll
    for (int w = 0; w < ENC_WAVES; w++) D[w + 1] = D[w] + L[w];
    // full words go out; the last, partial one stays in the window for whoever continues the stream
    // (round 6: "the one string that holds a word's first bit + whatever begins inside the word" instead of this loop over all
    // strings per word was measured: the string's start and length picked by the lane are select chains over D[] and L[], and
    // k_huffman_pack went from 0.176 to 0.189 ms)
    const uint32_t firstWord = ps.wordBase, endWord = D[ENC_WAVES] >> 5;
    uint32_t partial = 0;
    for (uint32_t J = firstWord + tid; J <= endWord; J += ENC_THREADS) {
        uint32_t val = J == firstWord ? carryWord : 0u;
#pragma unroll
        for (int w = 0; w < ENC_WAVES; w++) {
            const int32_t rel = (int32_t)(32u * J) - (int32_t)D[w];   // first bit of word J inside string w
            if (rel > -32 && rel < (int32_t)L[w]) {
                const uint32_t *src = win + w * WAVE_WIN;
                uint32_t x;
                if (rel >= 0) {
                    const uint32_t k = (uint32_t)rel >> 5, sh = (uint32_t)rel & 31u;
                    x = src[k] >> sh;
                    if (sh) x |= src[k + 1] << (32u - sh);            // bits beyond L[w] are zero
                } else {
                    x = src[0] << (uint32_t)(-rel);
                }
                val |= x;
            }
        }
        if (J == endWord) partial = val;
        else if (J < slotWords) out32[J] = val;
    }
    __syncthreads();
    window_clear(win);
    __syncthreads();
    if (((endWord - firstWord) % ENC_THREADS) == tid) win[0] = partial;      // the thread that computed word endWord
    ps.wordBase = endWord;
    ps.bitBase = D[ENC_WAVES];
    __syncthreads();
    return true;
}

// ---- the byte plane of raw row differences (GfEncodeArgs::plane; round 6): what the packers of the legacy and the canonical codec share ----
// (round 6) The low bytes of the residuals of eight consecutive cells i0 .. i0 + 7 (i0 a multiple of 8) of model 1, 2 or 3 from the
// tile's byte plane of raw row differences p (GfEncodeArgs::plane; every byte is its value exactly and the winner's values are plain
// bytes, so a residual's low byte is its M32 form): Differencing -- the plane's bytes; Linear -- a byte less its left neighbour
// (PredictorModelLinear.java:128-141: v - (2 W - WW) = (v - W) - (W - WW)); Triangle -- a byte less the byte above
// (PredictorModelTriangle.java:130-142: v - (W + N - NW) = (v - W) - (N - NW)).  Four bytes are subtracted at once (no carries
// between them); cells that the model does not emit in the flat scan (first row, first columns) give bytes nobody looks at.
// Words up to 12 bytes in front of the plane may be read (never used): the plane of tile 0 has the records' slack in front of it.
__device__ __forceinline__ uint32_t bytes_sub4(uint32_t x, uint32_t y)
{
    constexpr uint32_t H = 0x80808080u;
    return ((x | H) - (y & ~H)) ^ ((x ^ ~y) & H);
}
// (in two steps, so that the packer can ask for a turn's words one turn ahead: five registers instead of the nineteen of a turn's cells)
struct PlaneWords {
    uint32_t c0, c1;                    // the eight bytes at i0
    uint32_t x0, x1, x2;                // Linear: x0 = the word in front of them; Triangle: the three words that hold the bytes above
};
template <int MODEL>
__device__ __forceinline__ PlaneWords plane_load(const uint8_t *__restrict__ p, uint32_t nC, uint32_t i0)
{
    PlaneWords w;
    const uint32_t *pw = reinterpret_cast<const uint32_t *>(p + i0);
    const GfU2 cur = *reinterpret_cast<const GfU2 *>(pw);
    w.c0 = cur.x;
    w.c1 = cur.y;
    w.x0 = w.x1 = w.x2 = 0;
    if constexpr (MODEL == 2) {
        w.x0 = pw[-1];
    } else if constexpr (MODEL == 3) {
        int32_t o = (int32_t)i0 - (int32_t)nC;
        if (o < -8) o = -8;                                             // (the first row: nothing of it is emitted here)
        const uint32_t *uw = reinterpret_cast<const uint32_t *>(p + (o & ~3));
        w.x0 = uw[0];
        w.x1 = uw[1];
        w.x2 = uw[2];
    }
    return w;
}
template <int MODEL>
__device__ __forceinline__ void plane_residual_bytes(const PlaneWords &w, uint32_t nC, uint32_t i0, uint32_t (&rb)[2])
{
    if constexpr (MODEL == 1) {
        rb[0] = w.c0;
        rb[1] = w.c1;
    } else if constexpr (MODEL == 2) {
        rb[0] = bytes_sub4(w.c0, __builtin_amdgcn_alignbyte(w.c0, w.x0, 3u));
        rb[1] = bytes_sub4(w.c1, __builtin_amdgcn_alignbyte(w.c1, w.c0, 3u));
    } else {
        int32_t o = (int32_t)i0 - (int32_t)nC;
        if (o < -8) o = -8;
        const uint32_t sh = (uint32_t)o & 3u;
        rb[0] = bytes_sub4(w.c0, __builtin_amdgcn_alignbyte(w.x1, w.x0, sh));
        rb[1] = bytes_sub4(w.c1, __builtin_amdgcn_alignbyte(w.x2, w.x1, sh));
    }
}

template <int MODEL>
__device__ __forceinline__ uint32_t plane_head_cell(uint32_t s, uint32_t nC)
{
    if constexpr (MODEL == 2) {
        const uint32_t t = s - 1u;
        return s == 0u ? 1u : (1u + (t >> 1)) * nC + (t & 1u);
    } else {
        const uint32_t t = s - (nC - 1u);
        return s < nC - 1u ? s + 1u : (t + 1u) * nC;
    }
}
template <int MODEL>
__device__ __forceinline__ void plane_head_bytes(const uint8_t *__restrict__ p, uint32_t nC, uint32_t s0, uint32_t headElems, uint32_t (&hb)[2])
{
    uint32_t b[CPT];
#pragma unroll
    for (int j = 0; j < CPT; j++) {
        const uint32_t s = s0 + (uint32_t)j;
        const uint32_t v = p[s < headElems ? plane_head_cell<MODEL>(s, nC) : 0u];
        b[j] = s < headElems ? v : 0x80u;                                 // (0x80: the byte whose table entry is empty)
    }
    hb[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    hb[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
}

