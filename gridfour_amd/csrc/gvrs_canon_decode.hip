// gvrs_canon_decode.hip -- CodecCanonHuffman.decode for a batch of tiles on gfx950: persistent
// 256-thread workgroups, one tile at a time per workgroup.
//
// Reference (core/src/main/java/org/gridfour/compress/canonicalHuffman/):
//   CodecCanonHuffman.decode :163-195      6-byte header, uniform shortcut, predictor dispatch
//   CanonicalHuffman.decode :441-466, decodeText :469-519
//   LengthEncoder.readEncodedLengths :197-236, CanonHuffTreeDecoder :68-177
//   the predictors' decodeInt (compress/PredictorModel*.java)
//
// Phases
//   0  wave 0 reads the code tables (20 raw meta lengths, then the 260 coded text lengths); all threads
//      build an 11-bit lookup table by canonical search (no scatter), longer codes fall back to the search
//   1  the text is cut into <= 512 subsequences, two per thread; each one starts 128 bits early so that
//      it is synchronised with the true symbol sequence at its boundary almost always (self-synchronising
//      prefix code); barrier-synchronised fix-up rounds repair the rest; value counts are prefix-summed
//   2  every subsequence is decoded again, now writing each value (its target symbol and the escapes that
//      follow it, CanonicalHuffman.java:489-511) to the cell its stream position belongs to
//   3  predictor inverse in place (gvrs_decode_common.h)

#include <hip/hip_runtime.h>

#include "gvrs_kernels.h"
#include "huff_build.h"

namespace {

constexpr int DEC_THREADS = 256;
constexpr int DEC_WAVES = DEC_THREADS / 64;

#include "gvrs_decode_common.h"

#include "gvrs_canon_decode_common.h"

__global__ __launch_bounds__(DEC_THREADS, 4) void k_canon_decode(GfDecodeArgs a)
{
    __shared__ CanonDec S;

    const int tid = threadIdx.x;
    const uint32_t nR = (uint32_t)a.nRows, nC = (uint32_t)a.nCols, nCells = nR * nC;
    const uint32_t *__restrict__ w32 = reinterpret_cast<const uint32_t *>(a.blob);
    const uint64_t nWords = (a.blobBytes + 3) >> 2;
    const uint32_t capWords = a.ldsTextBytes >> 2;

    for (size_t t = blockIdx.x; t < a.nTiles; t += gridDim.x) {
        const uint64_t off = a.offsets ? a.offsets[t] : (uint64_t)t * a.slotStride;
        const uint32_t len = a.lengths[t];
        uint32_t *o = reinterpret_cast<uint32_t *>(a.values) + t * (size_t)nCells;
        const uint8_t *__restrict__ pk = a.blob + off;

        if (len < 6 || off + len > a.blobBytes) {                 // packing[1..5] -> ArrayIndexOutOfBounds
            if (tid == 0) a.status[t] = GF_K_ERR_BOUNDS;
            __syncthreads();
            continue;
        }
        const int predictor = (int8_t)pk[1];
        const uint32_t seed = (uint32_t)pk[2] | ((uint32_t)pk[3] << 8) | ((uint32_t)pk[4] << 16) | ((uint32_t)pk[5] << 24);
        if (predictor == 0 && len == 6) {                         // uniform tile, CodecCanonHuffman.java:171-176
            for (uint32_t i = tid; i < nCells; i += DEC_THREADS) o[i] = seed;
            if (tid == 0) a.status[t] = GF_K_OK;
            __syncthreads();
            continue;
        }
        if (predictor < 1 || predictor > 4 || (predictor == 2 && nC < 2)) {   // :208-209 IOException; Linear output[1]
            if (tid == 0) a.status[t] = predictor < 1 || predictor > 4 ? GF_K_ERR_FORMAT : GF_K_ERR_BOUNDS;
            __syncthreads();
            continue;
        }
        const int model = predictor;

        // the packing as words: aligned words of the blob, bit positions carry the misalignment
        const uint64_t word0 = off >> 2;
        const uint32_t bias = (uint32_t)(off & 3u) * 8u;
        const uint32_t endBit = bias + len * 8u;
        const uint32_t needWords = (endBit + 31u) / 32u + 4u;       // the readers look up to three words ahead
        const bool textInLds = needWords <= capWords;
        if (textInLds) cd_stage_text(w32, word0, nWords, endBit, needWords);
        const CdTextLds TL{needWords};
        const CdTextGlobal TG{w32 + word0, (uint32_t)min((uint64_t)needWords, nWords - word0)};   // huge packing: read in place
        __syncthreads();

        // ---------------- phases 0-2: the canonical-Huffman stream, values to their cells ----------------
        const uint32_t nStream = gf_stream_len(model, nR, nC);
        const bool useMagic = (uint64_t)nCells * nC < (1ull << 32);
        const uint32_t wMain = model == 2 ? (nC > 2 ? nC - 2u : 1u) : (nC > 1 ? nC - 1u : 1u);
        const uint32_t magic = (uint32_t)(((1ull << 32) + wMain - 1) / wMain);
        const CdCellSink sink{o, model, nR, nC, nStream, magic, useMagic && wMain > 1, !(a.phaseLimit & 0x100)};
        uint32_t endPos, nValues;
        uint32_t *stamps = a.debug ? a.debug + t * 16 : nullptr;
        if (stamps && tid == 0) stamps[0] = (uint32_t)__builtin_amdgcn_s_memtime();
        const int32_t st = textInLds ? cd_decode_stream(S, TL, bias + 48u, endBit, nCells, nStream, sink, &endPos, &nValues, stamps)
                                     : cd_decode_stream(S, TG, bias + 48u, endBit, nCells, nStream, sink, &endPos, &nValues, stamps);
        if (st != GF_K_OK) {
            if (tid == 0) a.status[t] = st;
            __syncthreads();
            continue;
        }

        // ---------------- phase 3: predictor inverse ----------------
        gf_predictor_inverse(model, seed, o, nR, nC, nullptr);
        if (stamps && tid == 0) stamps[6] = (uint32_t)__builtin_amdgcn_s_memtime();
        if (tid == 0) a.status[t] = GF_K_OK;
        __syncthreads();
    }
}

}  // namespace

uint32_t gf_canon_decode_lds_text(int nRows, int nCols)
{
    // LDS copy of a packing: typical terrain packs to 0.5-1 byte per cell; larger packings are read in place
    size_t cells = (size_t)nRows * (size_t)nCols;
    size_t want = cells - cells / 4 + 1024;
    if (want > 96 * 1024) want = 96 * 1024;
    return (uint32_t)((want + 31) & ~(size_t)31);
}

hipError_t gf_launch_canon_decode(const GfDecodeArgs &a, hipStream_t stream, unsigned grid)
{
    if (a.nTiles == 0) return hipSuccess;
    const size_t dyn = a.ldsTextBytes;
    static size_t maxDynSet = 0;
    if (dyn > maxDynSet) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_canon_decode),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) return e;
        maxDynSet = dyn;
    }
    hipLaunchKernelGGL(k_canon_decode, dim3(grid), dim3(DEC_THREADS), dyn, stream, a);
    return hipGetLastError();
}
