// gvrs_inflate.hip -- RFC 1950 / 1951 (zlib) streams inflated on the GPU, one wave per stream.
//
// Replaces the java.util.zip.Inflater calls of the reference's Deflate-carrying decoders
//   compress/CodecDeflate.java:141-147        one stream of M32 bytes behind the 10-byte header
//   compress/CodecFloat.java:285-298, 395-458 five byte planes, each a zlib stream behind a 4-byte length
//   lsop/LsDecoder12.java:127-141             two streams of M32 bytes
// What an inflater produces is defined by the stream, not by the implementation, so the bytes equal the JDK's (and the
// host zlib's) by construction; tests/test_gpu_inflate.py holds the kernel to the host's zlib on every block type, on
// the reference's sample files and on corrupted streams.
//
// Decomposition: a deflate stream is one serial chain (every code's position depends on the lengths of all codes before
// it, every match on the bytes before it), so the parallelism is across streams -- a batch has one to five per tile,
// thousands per launch.  One wave owns a stream: all 64 lanes run the bit-serial parse in lockstep on wave-uniform state
// (bit buffer, positions), the lookup tables and the last WINDOW bytes of output live in the wave's slice of LDS, and the
// lanes work together where there is width: building the tables of a dynamic block, copying matches and stored blocks,
// flushing the window to HBM with coalesced stores, the Adler-32 of every flushed piece.
//
// Result per stream: bytes produced and GF_K_OK or GF_K_ERR_FORMAT -- what one call of Inflater.inflate(byte[]) on the whole
// input gives (zlib's inflate() with all input and `cap` bytes of room): it stops without error when the room or the input
// runs out, reports invalid data where zlib does, and checks the Adler-32 when the stream ends inside the room.
#include <hip/hip_runtime.h>

#include "gvrs_kernels.h"

namespace {

constexpr int INF_WAVES = 4;                       // waves (= streams) per workgroup
constexpr uint32_t LL_BITS = 10, D_BITS = 9;       // first-level lookup widths; longer codes walk the canonical tables
constexpr uint32_t FLUSH = 1024;                   // window bytes gathered before they go to HBM
constexpr uint32_t IN_WORDS = 128;                 // the wave's LDS copy of the compressed input: 512 bytes, refilled by all lanes

struct __attribute__((packed, aligned(1))) InfWord { uint32_t v; };

// per-wave tables (LDS)
struct InfTables {
    uint16_t ll[1 << LL_BITS];                     // sym | len << 9; 0 = no code here / longer than the window
    uint16_t dd[1 << D_BITS];
    uint16_t llSym[288], dSym[32];                 // symbols in canonical order (length, symbol)
    uint16_t llCount[16], dCount[16];              // codes per length
    uint8_t lens[320];                             // code lengths of the block being set up
    uint16_t cl[128];                              // the code-length code's table (7 bits)
    uint32_t ibuf[IN_WORDS];                       // the compressed input, a piece at a time
};

__constant__ uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// Canonical Huffman tables from n code lengths (RFC 1951 3.2.2), the whole wave: count per length, the symbols in canonical
// order, and the first-level table (code bits reversed: the stream delivers a code's first bit in the lowest position).
// Returns 0, or -1 for an over-subscribed or (not allowed) incomplete set -- zlib's inflate_table rules: an incomplete set
// is accepted only when it consists of a single 1-bit code, and never for the code-length code (clTable: `left > 0 && (type ==
// CODES || max != 1)`, inftrees.c); -2: no code at all (the caller does what zlib does with its invalid-code marker table).
__device__ int inf_build(const uint8_t *lens, uint32_t n, uint16_t *count, uint16_t *symOrder, uint16_t *table, uint32_t tbits,
                         uint32_t lane, bool clTable = false)
{
    // counts per length
    if (lane < 16) count[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; l++) cnt[l] = 0;
    for (uint32_t s0 = 0; s0 < n; s0 += 64) {
        const uint32_t s = s0 + lane;
        const uint32_t L = s < n ? lens[s] : 0u;
#pragma unroll
        for (uint32_t l = 1; l < 16; l++) cnt[l] += (uint32_t)__popcll(__ballot(L == l));
    }
    // Kraft sum, first code and first position of every length (uniform)
    uint32_t left = 1, code = 0, offs = 0, nCodes = 0, maxLen = 0;
    uint32_t first[16], start[16];
    bool over = false;
#pragma unroll
    for (uint32_t l = 1; l < 16; l++) {
        left <<= 1;
        if (cnt[l] > left) over = true;
        left -= over ? 0u : cnt[l];
        code = (code + (l > 1 ? cnt[l - 1] : 0u)) << 1;
        first[l] = code;
        start[l] = offs;
        offs += cnt[l];
        nCodes += cnt[l];
        if (cnt[l]) maxLen = l;
    }
    first[0] = start[0] = 0;
    if (lane < 16) count[lane] = (uint16_t)(lane ? cnt[lane] : 0u);
    for (uint32_t i = lane; i < (1u << tbits); i += 64) table[i] = 0;
    __builtin_amdgcn_wave_barrier();
    if (over) return -1;
    if (left > 0 && (clTable || !(nCodes == 1 && maxLen == 1))) return nCodes == 0 ? -2 : -1;    // -2: no codes at all (caller decides)
    // rank of every symbol among those of its length -> canonical position and code
    uint32_t seen[16];
#pragma unroll
    for (int l = 0; l < 16; l++) seen[l] = 0;
    for (uint32_t s0 = 0; s0 < n; s0 += 64) {
        const uint32_t s = s0 + lane;
        const uint32_t L = s < n ? lens[s] : 0u;
        uint32_t rank = 0, fst = 0, st = 0;
#pragma unroll
        for (uint32_t l = 1; l < 16; l++) {
            const unsigned long long m = __ballot(L == l);
            if (L == l) {
                rank = seen[l] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                fst = first[l];
                st = start[l];
            }
            seen[l] += (uint32_t)__popcll(m);
        }
        if (L) {
            symOrder[st + rank] = (uint16_t)s;
            if (L <= tbits) {
                const uint32_t c = fst + rank;
                const uint32_t r = __brev(c) >> (32u - L);
                const uint16_t e = (uint16_t)(s | (L << 9));
                for (uint32_t i = r; i < (1u << tbits); i += 1u << L) table[i] = e;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return 0;
}

struct InfState {
    const uint8_t *in;                             // the stream (any byte alignment)
    const uint32_t *in32;                          // the aligned dword that holds its first byte
    uint32_t mis;                                  // position of that byte inside the dword
    uint32_t *ibuf;                                // LDS: dwords [bufWord, bufWord + IN_WORDS) of in32
    uint32_t bufWord, lastWord;                    // first buffered dword; last dword that holds a byte of the stream
    uint32_t inLen, inPos;
    uint64_t bitbuf;
    uint32_t bitcnt;
    bool starved;                                  // a read wanted more bits than the input holds
};

// Four bytes of the stream from byte position p, through the LDS copy.  A miss refills the copy with coalesced loads: reading
// the input dword by dword straight from HBM cost a memory round trip every four bytes of input.
__device__ __forceinline__ uint32_t inf_load32(InfState &z, uint32_t p)
{
    const uint32_t g = z.mis + p, wi = g >> 2;
    if (wi < z.bufWord || wi + 1u >= z.bufWord + IN_WORDS) {
        __builtin_amdgcn_wave_barrier();
        z.bufWord = wi;
        const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
        for (uint32_t k = 0; k < IN_WORDS / 64u; k++) {
            const uint32_t w = wi + lane + 64u * k;
            z.ibuf[lane + 64u * k] = w <= z.lastWord ? z.in32[w] : 0u;
        }
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t d0 = uni(z.ibuf[wi - z.bufWord]), d1 = uni(z.ibuf[wi - z.bufWord + 1u]);
    return __builtin_amdgcn_alignbit(d1, d0, (g & 3u) * 8u);
}

// make sure `n` (<= 32) bits are buffered; bits behind the input read as zero and raise `starved` when they are consumed
__device__ __forceinline__ void inf_need(InfState &z, uint32_t n)
{
    (void)n;
    if (z.bitcnt <= 32 && z.inPos < z.inLen) {
        uint32_t w = inf_load32(z, z.inPos);
        uint32_t nb = z.inLen - z.inPos;
        if (nb >= 4u) nb = 4u;
        else w &= (1u << (8u * nb)) - 1u;
        z.bitbuf |= (uint64_t)w << z.bitcnt;
        z.bitcnt += 8u * nb;
        z.inPos += nb;
    }
}
__device__ __forceinline__ uint32_t inf_bits(InfState &z, uint32_t n)       // n <= 16
{
    inf_need(z, n);
    if (z.bitcnt < n) { z.starved = true; z.bitcnt = n; }
    const uint32_t v = (uint32_t)z.bitbuf & ((1u << n) - 1u);
    z.bitbuf >>= n;
    z.bitcnt -= n;
    return v;
}

// one symbol: first-level table, then the canonical walk for longer codes (puff's loop).  Returns the symbol, -1 for a bit
// pattern that is no code (incomplete set), -2 when the input ran out inside the code.
__device__ __forceinline__ int inf_sym(InfState &z, const uint16_t *table, uint32_t tbits, const uint16_t *count, const uint16_t *symOrder)
{
    inf_need(z, 15);
    const uint32_t e = uni(table[(uint32_t)z.bitbuf & ((1u << tbits) - 1u)]);
    if (e) {
        const uint32_t L = e >> 9;
        if (L > z.bitcnt) { z.starved = true; return -2; }
        z.bitbuf >>= L;
        z.bitcnt -= L;
        return (int)(e & 511u);
    }
    uint32_t code = 0, first = 0, index = 0;
    uint64_t b = z.bitbuf;
    for (uint32_t len = 1; len <= 15; len++) {
        code |= (uint32_t)b & 1u;
        b >>= 1;
        const uint32_t c = uni(count[len]);
        if (code < first + c) {
            if (len > z.bitcnt) { z.starved = true; return -2; }
            z.bitbuf >>= len;
            z.bitcnt -= len;
            return (int)uni(symOrder[index + (code - first)]);
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

__global__ __launch_bounds__(64 * INF_WAVES) void k_inflate(GfInflateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsDyn[];
    // everything that steers the parse is wave-uniform and is kept in scalar registers: values that come out of LDS or out
    // of thread ids go through v_readfirstlane (uni) so that the compiler knows
    const uint32_t lane = threadIdx.x & 63u, wave = uni(threadIdx.x >> 6);
    const uint32_t W = a.window;                                   // power of two
    uint8_t *mine = ldsDyn + (size_t)wave * (W + sizeof(InfTables));
    uint8_t *win = mine;
    InfTables &T = *reinterpret_cast<InfTables *>(mine + W);

    if (a.gate && *a.gate == 0u) return;                           // a batch without a single Deflate container
    for (size_t sIdx = (size_t)blockIdx.x * INF_WAVES + wave; sIdx < a.nStreams; sIdx += (size_t)gridDim.x * INF_WAVES) {
        const GfInflateStream S = a.streams[sIdx];
        uint8_t *__restrict__ out = a.outBase + S.outOffset;
        int32_t status = GF_K_OK;
        uint32_t pos = 0, flushed = 0;                             // bytes produced / bytes already in HBM
        uint32_t s1 = 1, s2 = 0;                                   // Adler-32 of the flushed bytes
        InfState z;
        z.in = a.inBase + S.inOffset;
        z.mis = (uint32_t)(reinterpret_cast<uintptr_t>(z.in) & 3u);
        z.in32 = reinterpret_cast<const uint32_t *>(z.in - z.mis);
        z.ibuf = T.ibuf;
        z.bufWord = 0xF0000000u;                                   // nothing buffered yet
        z.inLen = S.inLen;
        z.lastWord = S.inLen ? (z.mis + S.inLen - 1u) >> 2 : 0u;
        z.inPos = 0;
        z.bitbuf = 0;
        z.bitcnt = 0;
        z.starved = false;
        const uint32_t cap = S.outCap;

        // window -> HBM (and into the checksum): bytes [flushed, upTo)
        auto flush = [&](uint32_t upTo) {
            uint32_t sumB = 0;
            uint64_t sumW = 0;
            const uint32_t n = upTo - flushed;
            for (uint32_t i = lane; i < n; i += 64) {
                const uint32_t b = win[(flushed + i) & (W - 1u)];
                out[flushed + i] = (uint8_t)b;
                sumB += b;
                sumW += (uint64_t)(n - i) * b;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                sumB += (uint32_t)__shfl_xor((int)sumB, d, 64);
                sumW += (uint64_t)__shfl_xor((long long)sumW, d, 64);
            }
            // s2 += n * s1 + sum (n - i) b_i ; s1 += sum b_i   (mod 65521)
            s2 = (uint32_t)(((uint64_t)s2 + (uint64_t)n * s1 + sumW) % 65521u);
            s1 = (uint32_t)(((uint64_t)s1 + sumB) % 65521u);
            flushed = upTo;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // far matches read these bytes back (other lanes, L2)
        };

        // ---- zlib header (RFC 1950) ----
        if (S.inLen < 2) {
            z.starved = true;
        } else {
            const uint32_t cmf = inf_bits(z, 8), flg = inf_bits(z, 8);
            if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u) status = GF_K_ERR_FORMAT;
            // a header that asks for a preset dictionary is no error: zlib returns Z_NEED_DICT, Inflater.inflate 0 bytes with
            // needsDictionary() set -- nothing is inflated
            else if (flg & 0x20u) z.starved = true;
        }
        bool done = false;                                         // the stream's last block has ended
        while (status == GF_K_OK && !z.starved && !done && pos <= cap) {
            const uint32_t last = inf_bits(z, 1), type = inf_bits(z, 2);
            if (z.starved) break;
            if (type == 3u) { status = GF_K_ERR_FORMAT; break; }
            if (type == 0u) {
                // stored: to the byte boundary, LEN, ~LEN, bytes
                const uint32_t drop = z.bitcnt & 7u;
                z.bitbuf >>= drop;
                z.bitcnt -= drop;
                const uint32_t len = inf_bits(z, 16), nlen = inf_bits(z, 16);
                if (z.starved) break;
                if ((len ^ 0xffffu) != nlen) { status = GF_K_ERR_FORMAT; break; }
                // the bit buffer holds whole bytes now: hand them back to the byte position
                z.inPos -= z.bitcnt >> 3;
                z.bitbuf = 0;
                z.bitcnt = 0;
                uint32_t todo = len;
                while (todo) {
                    uint32_t n = min(todo, min(z.inLen - z.inPos, min(cap - pos, FLUSH)));
                    if (n == 0) break;
                    for (uint32_t i = lane; i < n; i += 64) win[(pos + i) & (W - 1u)] = z.in[z.inPos + i];
                    __builtin_amdgcn_wave_barrier();
                    pos += n;
                    z.inPos += n;
                    todo -= n;
                    if (pos - flushed >= FLUSH) flush(pos);
                }
                if (todo) {                                        // room or input ran out inside the block
                    if (z.inPos >= z.inLen) z.starved = true;
                    break;
                }
                done = last != 0u;
                continue;
            }
            // ---- code tables ----
            uint32_t nLL, nD;
            if (type == 1u) {
                nLL = 288;
                nD = 32;                                           // 30 and 31 are codes of the fixed set that must not occur
                for (uint32_t i = lane; i < 288; i += 64) T.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                if (lane < 32) T.lens[288 + lane] = 5;
                __builtin_amdgcn_wave_barrier();
            } else {
                nLL = inf_bits(z, 5) + 257u;
                nD = inf_bits(z, 5) + 1u;
                const uint32_t nCL = inf_bits(z, 4) + 4u;
                if (z.starved) break;
                if (nLL > 286u || nD > 30u) { status = GF_K_ERR_FORMAT; break; }
                if (lane < 19) T.lens[lane] = 0;
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = 0; i < nCL; i++) {
                    const uint32_t v = inf_bits(z, 3);
                    if (lane == 0) T.lens[CL_ORDER[i]] = (uint8_t)v;
                }
                __builtin_amdgcn_wave_barrier();
                if (z.starved) break;
                // the code-length code: its lengths sit in lens[0..19); tables into cl / (reused) dCount, dSym
                const int rcl = inf_build(T.lens, 19, T.dCount, T.dSym, T.cl, 7, lane, true);
                if (rcl == -2) {
                    // No code-length code at all: zlib's table for that is two invalid-code markers of one bit and value 0, and
                    // CODELENS takes `val < 16` before it looks at the marker -- every length reads as 0, a bit each; the block
                    // then fails on its missing end-of-block code, unless the input runs out first (no error then)
                    for (uint32_t i = 0; i < nLL + nD && !z.starved; i++) (void)inf_bits(z, 1);
                    if (z.starved) break;
                    status = GF_K_ERR_FORMAT;
                    break;
                }
                if (rcl != 0) { status = GF_K_ERR_FORMAT; break; }
                // the literal/length and distance code lengths, run-length coded (serial)
                uint32_t idx = 0;
                bool bad = false;
                uint32_t prev = 0;
                __builtin_amdgcn_wave_barrier();
                // lens[] is both the source of the CL table (already built) and the destination: rebuild from index 0
                while (idx < nLL + nD) {
                    const int sym = inf_sym(z, T.cl, 7, T.dCount, T.dSym);
                    if (sym < 0) { bad = sym == -1; break; }
                    if (sym < 16) {
                        if (lane == 0) T.lens[idx] = (uint8_t)sym;
                        prev = (uint32_t)sym;
                        idx++;
                    } else {
                        // zlib asks for the code AND its extra bits before it looks at either (inflate.c, CODELENS: NEEDBITS(here.bits
                        // + 2 / 3 / 7)): a stream that ends inside them waits for input, it is not "invalid bit length repeat"
                        uint32_t rep, val = 0;
                        if (sym == 16) rep = 3u + inf_bits(z, 2);
                        else if (sym == 17) rep = 3u + inf_bits(z, 3);
                        else rep = 11u + inf_bits(z, 7);
                        if (z.starved) break;
                        if (sym == 16) {
                            if (idx == 0) { bad = true; break; }
                            val = prev;
                        }
                        if (idx + rep > nLL + nD) { bad = true; break; }
                        for (uint32_t i = lane; i < rep; i += 64) T.lens[idx + i] = (uint8_t)val;
                        if (sym != 16) prev = 0;
                        idx += rep;
                    }
                    if (z.starved) break;
                }
                __builtin_amdgcn_wave_barrier();
                if (bad) { status = GF_K_ERR_FORMAT; break; }
                if (z.starved || idx < nLL + nD) { z.starved = true; break; }
                if (uni(T.lens[256]) == 0) { status = GF_K_ERR_FORMAT; break; }         // no end-of-block code
            }
            {
                const int r1 = inf_build(T.lens, nLL, T.llCount, T.llSym, T.ll, LL_BITS, lane);
                if (r1 != 0) { status = GF_K_ERR_FORMAT; break; }
                // distance lengths follow the literal/length ones; an empty distance set is legal (literals only)
                const int r2 = inf_build(T.lens + nLL, nD, T.dCount, T.dSym, T.dd, D_BITS, lane);
                if (r2 == -1) { status = GF_K_ERR_FORMAT; break; }
            }
            // ---- the block's symbols ----
            for (;;) {
                if (pos - flushed >= FLUSH) flush(pos);
                const int sym = inf_sym(z, T.ll, LL_BITS, T.llCount, T.llSym);
                if (sym < 0) {
                    if (sym == -1) status = GF_K_ERR_FORMAT;
                    break;
                }
                if (sym < 256) {
                    if (pos >= cap) { pos = cap + 1; break; }                            // no room: stop here, no error
                    if (lane == 0) win[pos & (W - 1u)] = (uint8_t)sym;
                    pos++;
                    continue;
                }
                if (sym == 256) {
                    done = last != 0u;
                    break;
                }
                if (sym > 285) { status = GF_K_ERR_FORMAT; break; }
                // base value and extra bits of the length / distance codes (RFC 1951 3.2.5) in closed form: the tables in
                // constant memory cost a memory round trip per match
                const uint32_t li = (uint32_t)sym - 257u;
                const uint32_t lx = li < 8u || li == 28u ? 0u : (li - 4u) >> 2;
                const uint32_t lbase = li < 8u ? li + 3u : li == 28u ? 258u : ((4u + (li & 3u)) << lx) + 3u;
                const uint32_t len = lbase + (lx ? inf_bits(z, lx) : 0u);
                if (z.starved) break;                              // (before the distance code is looked at: on zero-padded bits an
                                                                   // empty or one-code distance set would read as "invalid distance code")
                const int ds = inf_sym(z, T.dd, D_BITS, T.dCount, T.dSym);
                if (ds < 0) {
                    if (ds == -1) status = GF_K_ERR_FORMAT;
                    break;
                }
                if (ds > 29) { status = GF_K_ERR_FORMAT; break; }
                const uint32_t dx = ds < 4 ? 0u : ((uint32_t)ds - 2u) >> 1;
                const uint32_t dbase = ds < 4 ? (uint32_t)ds + 1u : ((2u + ((uint32_t)ds & 1u)) << dx) + 1u;
                const uint32_t dist = dbase + (dx ? inf_bits(z, dx) : 0u);
                if (z.starved) break;
                // zlib looks at the room before it looks at the distance (inflate.c, state MATCH: `if (left == 0) goto inf_leave`
                // comes first, "invalid distance too far back" after it): a match that arrives when the room is used up ends the
                // call without an error whatever its distance says
                if (pos >= cap) { pos = cap + 1; break; }
                if (dist > pos) { status = GF_K_ERR_FORMAT; break; }                     // too far back
                // copy `len` bytes from `dist` back; source and destination may overlap (then the pattern repeats): go in
                // pieces no longer than the distance, each piece with all lanes
                uint32_t n = min(len, cap - pos);
                const bool cut = n < len;
                __builtin_amdgcn_wave_barrier();
                // the LDS window holds the last W bytes; a source further back than that is read from the output in HBM, where
                // it has been since an earlier flush (at most FLUSH + 258 bytes are ever unflushed)
                const bool far = dist + 64u > W;
                // a distance below 64 repeats a pattern: once a piece of D bytes is copied, the last 2 D bytes are the pattern
                // twice, so the next piece can come from 2 D back -- 1, 2, 4 .. 64 bytes per step instead of `dist` each time
                uint32_t D = dist;
                while (n) {
                    const uint32_t piece = min(n, min(D, 64u));
                    uint32_t b = 0;
                    if (lane < piece)
                        b = far ? (uint32_t)__hip_atomic_load(out + (pos - D + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : (uint32_t)win[(pos - D + lane) & (W - 1u)];
                    __builtin_amdgcn_wave_barrier();
                    if (lane < piece) win[(pos + lane) & (W - 1u)] = (uint8_t)b;
                    __builtin_amdgcn_wave_barrier();
                    pos += piece;
                    n -= piece;
                    if (D < 64u) D += piece;
                    if (pos - flushed >= W - 512u) flush(pos);                           // (long matches in a small window)
                }
                if (cut) { pos = cap + 1; break; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        const bool full = pos > cap;                               // stopped for lack of room
        if (full) pos = cap;
        __builtin_amdgcn_wave_barrier();
        flush(pos);
        if (status == GF_K_OK && done && !full) {
            // the stream ended inside the room: its Adler-32 follows, big-endian, at the next byte boundary
            const uint32_t drop = z.bitcnt & 7u;
            z.bitbuf >>= drop;
            z.bitcnt -= drop;
            z.starved = false;
            const uint32_t b0 = inf_bits(z, 8), b1 = inf_bits(z, 8), b2 = inf_bits(z, 8), b3 = inf_bits(z, 8);
            if (!z.starved && ((b0 << 24) | (b1 << 16) | (b2 << 8) | b3) != ((s2 << 16) | s1)) status = GF_K_ERR_FORMAT;
        }
        if (lane == 0) {
            a.produced[sIdx] = pos;
            a.status[sIdx] = status;
            if (a.consumed) a.consumed[sIdx] = z.inPos - (z.bitcnt >> 3);      // Inflater.getTotalIn(): whole bytes not yet looked at stay
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- containers: where the zlib streams of a packing are, one thread per tile ------------------------------------
// CodecDeflate (CodecDeflate.java:108-155): 10-byte header (index, predictor, seed LE, nM32 LE), then ONE stream of nM32 bytes.
// The header goes to the front of the tile's raw container, the stream's output behind it: what k_huffman_decode's rawM32
// mode reads.  pre[t] = the status the reference's own checks give before inflating (GF_K_OK: go on).
__global__ void k_deflate_streams(const uint8_t *__restrict__ blob, size_t blobBytes, const uint64_t *__restrict__ offsets,
                                  size_t slotStride, const uint32_t *__restrict__ lengths, size_t tile0, size_t nTiles, uint32_t cells,
                                  uint8_t *__restrict__ raw, size_t rawStride, GfInflateStream *__restrict__ desc,
                                  int32_t *__restrict__ pre)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTiles) return;
    const size_t t = tile0 + i;
    const uint64_t off = offsets ? offsets[t] : (uint64_t)t * slotStride;
    const uint32_t len = lengths[t];
    GfInflateStream d;
    d.inOffset = off + 10;
    d.outOffset = i * rawStride + 10;
    d.inLen = 0;
    d.outCap = 0;
    int32_t st = GF_K_OK;
    if (len < 10 || off + len > blobBytes) {
        st = GF_K_ERR_BOUNDS;
    } else {
        const uint8_t *pk = blob + off;
        const uint32_t nM32 = (uint32_t)pk[6] | ((uint32_t)pk[7] << 8) | ((uint32_t)pk[8] << 16) | ((uint32_t)pk[9] << 24);
        if ((int32_t)nM32 < 0) st = GF_K_ERR_BOUNDS;                       // NegativeArraySizeException
        else if ((uint64_t)nM32 > 6ull * cells) st = GF_K_ERR_FORMAT;      // no encoder emits this
        else {
            for (int k = 0; k < 10; k++) raw[i * rawStride + k] = pk[k];
            d.inLen = len - 10;
            d.outCap = nM32;
        }
    }
    desc[i] = d;
    pre[i] = st;
}

// lengths of the raw containers for the decode kernel (0: the tile is out) and the status so far
__global__ void k_deflate_lengths(size_t nTiles, const GfInflateStream *__restrict__ desc, const uint32_t *__restrict__ produced,
                                  const int32_t *__restrict__ inflStatus, int32_t *__restrict__ pre, uint32_t *__restrict__ rawLengths,
                                  uint8_t *__restrict__ rawBase)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTiles) return;
    int32_t st = pre[i];
    // A stream that ends early (damaged input) fills only part of codeM32s; the rest of the Java array is zero
    // (new byte[nM32], CodecDeflate.java:139) and the predictor may read it.  Rare, so one thread clears it.
    if (st == GF_K_OK && inflStatus[i] == GF_K_OK && produced[i] > 0u && produced[i] < desc[i].outCap) {
        uint8_t *p = rawBase + desc[i].outOffset;
        for (uint32_t k = produced[i]; k < desc[i].outCap; k++) p[k] = 0;
    }
    if (st == GF_K_OK && inflStatus[i] != GF_K_OK) st = GF_K_ERR_FORMAT;        // DataFormatException -> IOException
    else if (st == GF_K_OK && produced[i] == 0) st = GF_K_DECLINED;             // inflate gave nothing: decode returns null (:143-154)
    pre[i] = st;
    rawLengths[i] = st == GF_K_OK ? 10u + desc[i].outCap : 0u;
}

// final status: the container's own where it failed, the decode kernel's otherwise
__global__ void k_merge_status(size_t nTiles, const int32_t *__restrict__ pre, const int32_t *__restrict__ decoded,
                               int32_t *__restrict__ status)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTiles) return;
    status[i] = pre[i] != GF_K_OK ? pre[i] : (decoded ? decoded[i] : GF_K_OK);
}

// CodecFloat (CodecFloat.java:395-458): two header bytes, then five times [int32 LE n][zlib stream of n bytes]: sign bits,
// exponent, three mantissa planes.  Five descriptors per tile; pre[t] = GF_K_ERR_BOUNDS where the framing runs off the packing.
__global__ void k_float_streams(const uint8_t *__restrict__ blob, size_t blobBytes, const uint64_t *__restrict__ offsets,
                                const uint32_t *__restrict__ lengths, size_t tile0, size_t nTiles, uint32_t cells, size_t planeStride,
                                GfInflateStream *__restrict__ desc, int32_t *__restrict__ pre)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTiles) return;
    const size_t t = tile0 + i;
    const uint64_t off = offsets[t];
    const uint32_t len = lengths[t];
    const uint32_t nSign = (cells + 7u) >> 3;
    int32_t st = off + len > blobBytes ? GF_K_ERR_BOUNDS : GF_K_OK;
    uint32_t p = 2, planeOff = 0;
    for (int k = 0; k < 5; k++) {
        GfInflateStream d;
        d.inOffset = 0;
        d.outOffset = i * planeStride + planeOff;
        d.inLen = 0;
        d.outCap = 0;
        const uint32_t pl = k == 0 ? nSign : cells;
        if (st == GF_K_OK) {
            if (p + 4u > len) st = GF_K_ERR_BOUNDS;
            else {
                const uint8_t *q = blob + off + p;
                const uint32_t zn = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
                p += 4;
                if ((uint64_t)p + zn > len) st = GF_K_ERR_BOUNDS;
                else {
                    d.inOffset = off + p;
                    d.inLen = zn;
                    d.outCap = pl;
                    p += zn;
                }
            }
        }
        desc[i * 5 + k] = d;
        planeOff += pl;
    }
    pre[i] = st;
}

// LSOP12 containers that carry Deflate (LsDecoder12.java:127-141): header (either revision, LsHeader.java:131-185) with the
// byte counts of the two M32 streams, then TWO zlib streams back to back -- the second one begins where the first one's
// Inflater stopped reading (Inflater.getTotalIn()).  Two passes of one thread per tile around two k_inflate launches:
// pass 0 describes stream one; pass 1 checks what came out of it and describes stream two.  Both streams of tile i go to
// raw + i * rawStride: [initialiser M32 bytes][interior M32 bytes].  side[i]: 0 = a Deflate container on its way, 1 = not one
// (or a header k_lsop_unpack_m32 rejects by itself), < 0 = the status of a failed first stream.
__global__ void k_lsop_streams(const uint8_t *__restrict__ blob, size_t blobBytes, const uint64_t *__restrict__ offsets, size_t slotStride,
                               const uint32_t *__restrict__ lengths, size_t nTiles, uint32_t nInit, uint32_t nInt, size_t rawStride,
                               int pass, const uint32_t *__restrict__ produced, const int32_t *__restrict__ inflStatus,
                               const uint32_t *__restrict__ consumed, GfInflateStream *__restrict__ desc, int32_t *__restrict__ side,
                               uint32_t *__restrict__ gate)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTiles) return;
    if (pass == 1 && *gate == 0u) return;                          // pass 0 found no Deflate container: nothing to place
    GfInflateStream d;
    d.inOffset = 0;
    d.outOffset = i * rawStride;
    d.inLen = 0;
    d.outCap = 0;
    if (pass == 1 && side[i] != 0) {
        desc[i] = d;
        return;
    }
    const uint64_t off = offsets ? offsets[i] : (uint64_t)i * slotStride;
    const uint32_t len = lengths[i];
    const uint8_t *__restrict__ pk = blob + off;
    auto le32 = [&](uint32_t o) -> uint32_t {
        return (uint32_t)pk[o] | ((uint32_t)pk[o + 1] << 8) | ((uint32_t)pk[o + 2] << 16) | ((uint32_t)pk[o + 3] << 24);
    };
    int32_t sd = 1;
    uint32_t o = 1, type = 0, nMI = 0, nMX = 0;
    if (len >= 3 && off + len <= blobBytes) {
        const bool revised = pk[1] & 0x40;
        bool checksum = false;
        if (revised) { type = pk[1] & 0x0fu; checksum = pk[1] & 0x80; o = 2; }
        if (len >= o + 1u + 52u + 8u + (revised ? 0u : 1u) && pk[o] == 12) {
            o += 53;
            nMI = le32(o);
            nMX = le32(o + 4);
            o += 8;
            if (!revised) { type = pk[o] & 0x0fu; checksum = pk[o] & 0x80; o++; }
            if (checksum) o += 4;
            if (o <= len && type != 0 && type != 2 && nMI <= 6u * nInit + 64u && nMX <= 6u * nInt + 64u) sd = 0;
        }
    }
    if (sd == 0) {
        if (pass == 0) {
            d.inOffset = off + o;
            d.inLen = len - o;
            d.outCap = nMI;
        } else if (inflStatus[i] != GF_K_OK || produced[i] < nMI) {
            sd = GF_K_ERR_FORMAT;                                  // what makes the first Inflater.inflate call fail or fall short
        } else {
            const uint32_t used = min(consumed[i], len - o);
            d.inOffset = off + o + used;
            d.outOffset = i * rawStride + nMI;
            d.inLen = len - o - used;
            d.outCap = nMX;
        }
    }
    desc[i] = d;
    side[i] = sd;
    if (pass == 0 && sd == 0) atomicAdd(gate, 1u);
}

__global__ void k_float_status(size_t nTiles, const int32_t *__restrict__ pre, const int32_t *__restrict__ inflStatus,
                               int32_t *__restrict__ status)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTiles) return;
    int32_t st = pre[i];
    for (int k = 0; k < 5 && st == GF_K_OK; k++)
        if (inflStatus[i * 5 + k] != GF_K_OK) st = GF_K_ERR_FORMAT;            // doInflate :285-298 -> RuntimeException
    status[i] = st;
}

}  // namespace

hipError_t gf_launch_deflate_streams(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, size_t slotStride,
                                     const uint32_t *lengths, size_t tile0, size_t nTiles, uint32_t cells, uint8_t *raw, size_t rawStride,
                                     GfInflateStream *desc, int32_t *pre, hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_deflate_streams, dim3((unsigned)((nTiles + 255) / 256)), dim3(256), 0, stream, blob, blobBytes, offsets, slotStride,
                       lengths, tile0, nTiles, cells, raw, rawStride, desc, pre);
    return hipGetLastError();
}

hipError_t gf_launch_deflate_lengths(size_t nTiles, const GfInflateStream *desc, const uint32_t *produced, const int32_t *inflStatus,
                                     int32_t *pre, uint32_t *rawLengths, uint8_t *rawBase, hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_deflate_lengths, dim3((unsigned)((nTiles + 255) / 256)), dim3(256), 0, stream, nTiles, desc, produced, inflStatus,
                       pre, rawLengths, rawBase);
    return hipGetLastError();
}

hipError_t gf_launch_merge_status(size_t nTiles, const int32_t *pre, const int32_t *decoded, int32_t *status, hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_status, dim3((unsigned)((nTiles + 255) / 256)), dim3(256), 0, stream, nTiles, pre, decoded, status);
    return hipGetLastError();
}

hipError_t gf_launch_float_streams(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, const uint32_t *lengths, size_t tile0,
                                   size_t nTiles, uint32_t cells, size_t planeStride, GfInflateStream *desc, int32_t *pre, hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_float_streams, dim3((unsigned)((nTiles + 255) / 256)), dim3(256), 0, stream, blob, blobBytes, offsets, lengths,
                       tile0, nTiles, cells, planeStride, desc, pre);
    return hipGetLastError();
}

// CodecFloat.decodeFloats uses ONE scratch array for its five planes (CodecFloat.java:397, 409-446) and decodes the mantissa
// deltas in place: behind a stream that ends early -- damaged input; no exception, doInflate only looks at the sign of the
// count -- a plane keeps what the plane before it left there (delta-DECODED bytes behind the mantissa planes).  The plane
// buffer here has a region per plane, zeroed before the inflate; for the rare tile with a short plane this kernel (one
// thread per tile) writes into each short plane's tail what the Java array would hold, so that the plane merge that follows
// sees the reference's bytes.
__global__ void k_float_short_planes(size_t nTiles, const int32_t *__restrict__ pre, const int32_t *__restrict__ inflStatus,
                                     const uint32_t *__restrict__ produced, uint8_t *__restrict__ planes, size_t planeStride, int nRows,
                                     int nCols)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTiles || pre[t] != GF_K_OK) return;
    const uint32_t n = (uint32_t)nRows * (uint32_t)nCols, nSign = (n + 7u) >> 3;
    bool any = false;
    for (int p = 0; p < 5; p++) {
        if (inflStatus[t * 5 + p] != GF_K_OK) return;                 // the tile fails anyway
        if (p > 0 && produced[t * 5 + p] < n) any = true;
    }
    if (!any) return;
    uint8_t *P0 = planes + t * planeStride, *P1 = P0 + nSign;
    // plane 1 (exponent) behind its end: the sign bytes, then the zeros of the fresh array
    for (uint32_t i = produced[t * 5 + 1]; i < n; i++) P1[i] = i < nSign ? P0[i] : (uint8_t)0;
    // planes 2..4 behind their ends: what decodeDeltas left of the plane before (plane 2: the exponent bytes as they are)
    for (int p = 2; p < 5; p++) {
        uint8_t *cur = P1 + (size_t)(p - 1) * n;
        const uint8_t *prev = cur - n;
        const uint32_t got = produced[t * 5 + p];
        if (got >= n) continue;
        if (p == 2) {
            for (uint32_t i = got; i < n; i++) cur[i] = prev[i];
        } else {
            // decodeDeltas :315-326 over the previous plane, keeping only the tail
            int prior = 0;
            uint32_t k = 0;
            for (int r = 0; r < nRows; r++) {
                int first = 0;
                for (int c = 0; c < nCols; c++, k++) {
                    prior += (int8_t)prev[k];
                    const uint8_t d = (uint8_t)prior;
                    prior = (int8_t)d;
                    if (c == 0) first = (int8_t)d;
                    if (k >= got) cur[k] = d;
                }
                prior = first;
            }
        }
    }
}

hipError_t gf_launch_float_short_planes(size_t nTiles, const int32_t *pre, const int32_t *inflStatus, const uint32_t *produced,
                                        uint8_t *planes, size_t planeStride, int nRows, int nCols, hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_float_short_planes, dim3((unsigned)((nTiles + 63) / 64)), dim3(64), 0, stream, nTiles, pre, inflStatus, produced,
                       planes, planeStride, nRows, nCols);
    return hipGetLastError();
}

hipError_t gf_launch_lsop_streams(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, size_t slotStride, const uint32_t *lengths,
                                  size_t nTiles, uint32_t nInit, uint32_t nInt, size_t rawStride, int pass, const uint32_t *produced,
                                  const int32_t *inflStatus, const uint32_t *consumed, GfInflateStream *desc, int32_t *side, uint32_t *gate,
                                  hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_lsop_streams, dim3((unsigned)((nTiles + 255) / 256)), dim3(256), 0, stream, blob, blobBytes, offsets, slotStride,
                       lengths, nTiles, nInit, nInt, rawStride, pass, produced, inflStatus, consumed, desc, side, gate);
    return hipGetLastError();
}

hipError_t gf_launch_float_status(size_t nTiles, const int32_t *pre, const int32_t *inflStatus, int32_t *status, hipStream_t stream)
{
    if (nTiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_float_status, dim3((unsigned)((nTiles + 255) / 256)), dim3(256), 0, stream, nTiles, pre, inflStatus, status);
    return hipGetLastError();
}

uint32_t gf_inflate_window(uint32_t maxOut)
{
    (void)maxOut;
    return 2048;                // > FLUSH + the longest match + 512: unflushed bytes never wrap; older bytes are read back from HBM
}

hipError_t gf_launch_inflate(const GfInflateArgs &a, hipStream_t stream)
{
    if (a.nStreams == 0) return hipSuccess;
    const size_t dyn = (size_t)INF_WAVES * (a.window + sizeof(InfTables));
    static GfDynLdsOptIn opt;
    const hipError_t e = gf_opt_in_dyn_lds(k_inflate, dyn, opt);
    if (e != hipSuccess) return e;
    const size_t blocks = (a.nStreams + INF_WAVES - 1) / INF_WAVES;
    const unsigned grid = (unsigned)(blocks < 256 * 16 ? blocks : 256 * 16);
    hipLaunchKernelGGL(k_inflate, dim3(grid), dim3(64 * INF_WAVES), dyn, stream, a);
    return hipGetLastError();
}
