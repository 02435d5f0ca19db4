// gvrs_kernels.h -- launch interface between the C ABI (gvrs_api.hip) and the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

// per-tile status values written by the kernels; identical to gf_status in
// include/gvrs_hip_codec.h
#define GF_K_OK 0
#define GF_K_DECLINED 1
#define GF_K_OVERFLOW 2
#define GF_K_ERR_FORMAT (-1)
#define GF_K_ERR_BOUNDS (-2)
#define GF_K_ERR_ARG (-4)
#define GF_K_ERR_UNSUPPORTED (-7)

// One tile per workgroup and NO tile loop in the kernel.  A persistent grid-stride loop over tiles invites the compiler to hoist
// every thread-index and tile-shape term out of it; at the kernels' register caps those dozens of hoisted values are then
// spilled (VGPRs to scratch -- stored and reloaded once per tile, i.e. HBM traffic; SGPRs into lanes of reserved VGPRs).
// Measured on the bench batch: k_huffman_decode<fast> 118 -> 86 VGPRs, 72 -> 11 spilled SGPRs, 1.43 -> 1.31 ms;
// k_huffman_pack 14 -> 6 spilled VGPRs.  GF_FOR_WG_TILE is a `for` that runs at most once (so `continue` / `break` keep
// their meaning); gf_tile_grid spreads the tiles over x and y so that any tile count fits one launch.
#define GF_FOR_WG_TILE(t, nTiles) \
    for (size_t t = (size_t)blockIdx.x + (size_t)blockIdx.y * gridDim.x; t < (size_t)(nTiles); t = ~(size_t)0)
// the same with the choice left to a template parameter: a persistent grid-stride loop (1-D grid) when ONESHOT is false
#define GF_FOR_TILES(t, nTiles, ONESHOT) \
    for (size_t t = (size_t)blockIdx.x + (size_t)blockIdx.y * gridDim.x; t < (size_t)(nTiles); t = (ONESHOT) ? ~(size_t)0 : t + gridDim.x)
inline dim3 gf_tile_grid(size_t nTiles)
{
    const size_t gx = nTiles < (1u << 20) ? (nTiles ? nTiles : 1) : (1u << 20), gy = (nTiles + gx - 1) / gx;
    return dim3((unsigned)gx, (unsigned)(gy ? gy : 1), 1);
}

// The pre-pass kernels (k_huffman_parse_trees, k_canon_parse_lengths) walk one tile per LANE: a serial walk of some hundred
// steps per tile.  A large batch fills the chip that way (12,960 tiles = 203 waves, and every wave instruction serves 64
// tiles); a small one does not -- 1,024 tiles are 16 waves on 1,024 SIMDs, each crawling through the divergent walks of 64
// trees.  Small batches therefore give every tile a wave of its own -- an instantiation with ONE active lane, which the compiler
// turns into scalar code (with 2 or 4 lanes, or the lane count as a kernel argument, half the gain is lost): decode of the
// 1,024-tile batch 0.339 -> 0.300 ms; on the 12,960-tile batch the same costs 0.07 ms, 16 or 8 tiles per wave 0.01-0.02 ms.
#ifndef GF_PREPASS_ONE_LANE_MAX
#define GF_PREPASS_ONE_LANE_MAX 4096      // 120x150 tiles, decode ms with a lane / a wave per tile: 2,048 tiles 0.308 / 0.280, 3,000
                                          // 0.403 / 0.386, 4,096 0.511 / 0.502, 6,000 0.703 / 0.710, 8,192 0.904 / 0.920
#endif
inline unsigned gf_prepass_tiles_per_wave(size_t nTiles) { return nTiles <= GF_PREPASS_ONE_LANE_MAX ? 1u : 64u; }

// Dynamic LDS beyond the default limit must be opted into, per kernel and PER DEVICE (hipFuncSetAttribute acts on the
// current device's copy of the function).  One GfDynLdsOptIn per kernel remembers the largest size asked for on each
// device; contexts on different devices and threads of one process share it safely.
constexpr int GF_MAX_DEVICES = 64;
struct GfDynLdsOptIn {
    size_t done[GF_MAX_DEVICES] = {};
    std::mutex mu;
};
template <class K>
inline hipError_t gf_opt_in_dyn_lds(K kernel, size_t dyn, GfDynLdsOptIn &st)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= GF_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(st.mu);
    if (dyn <= st.done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (e == hipSuccess) st.done[dev] = dyn;
    return e;
}

struct GfEncodeArgs {
    const int32_t *values;     // nTiles * nRows*nCols
    uint8_t *out;              // nTiles slots of slotStride bytes
    uint32_t *lengths;
    uint8_t *predictors;       // may be null
    int32_t *status;
    size_t nTiles;
    size_t slotStride;
    int nRows, nCols;
    int codecIndex;
    int predictorMask;
    uint32_t *debug;           // optional diagnostic dump (GF_ENC_DEBUG_WORDS per tile), normally null
    int phaseLimit;            // diagnostic: stop after phase A (1) / B (2); 0 = run everything
    uint32_t *retryFlag;       // CodecHuffman: two device words (the second one counts the tiles k_huffman_pack leaves to k_huffman_pack_rare); the fast encode kernel ORs 1 into the first for every tile it leaves to
                               // the general kernel (status GF_K_RETRY inside the launch only); null: general kernel only
    uint32_t *packRecs;        // non-null (CodecHuffman only): k_huffman_encode stops after the selection and leaves per tile
                               // GF_PACK_REC_WORDS words (model, tree end bit, seed, maxN, maxLen, tree image, code table)
                               // for k_huffman_pack, which writes the packing
    int lean;                  // 1 (the one-tile-per-call path): only the kernels a tile usually needs are launched; a tile that
                               // needs another one (k_huffman_pack_rare) is reported with the internal status GF_K_LEAN_RETRY and
                               // the caller takes the batch path for it
    uint32_t *encStats;        // non-null (CodecHuffman batches): the encoder runs as k_huffman_encode (histograms only) +
                               // k_huffman_trees (one wave per tile); GF_ENC_STAT_WORDS words per tile between them
    uint8_t *plane;            // non-null (CodecHuffman batches with encStats, round 6): phase A leaves every cell's RAW ROW DIFFERENCE
                               // as one byte -- v - W, column 0: v - N, the seed cell 0 -- at plane + t * planeStride + cell, and marks
                               // the tile (word 8 of its statistics record) when every one of them is that byte exactly; the packer
                               // then derives the winner's residuals from the plane (1 byte per cell) instead of reading the tile again
    size_t planeStride;        // bytes per tile: the cells rounded up to a multiple of 16
};
constexpr int GF_ENC_STAT_WORDS = 16 + 3 * 256;   // models, seed, longest value per predictor, a "has work" mark; the three histograms
#define GF_K_LEAN_RETRY 0x7fff0002              /* (= GF_K_RETRY of the kernels: the fast kernels' own mark travels the same way) */
constexpr int GF_PACK_REC_WORDS = 8 + 88 + 512;

struct GfDecodeArgs {
    const uint8_t *blob;       // 4-byte aligned
    size_t blobBytes;
    const uint64_t *offsets;   // may be null -> t * slotStride
    size_t slotStride;
    const uint32_t *lengths;
    int32_t *values;
    int32_t *status;
    uint8_t *workspace;        // gridDim.x * workspaceStride bytes (M32 spill)
    size_t workspaceStride;
    size_t nTiles;
    int nRows, nCols;
    uint32_t ldsM32Bytes;      // capacity of the in-LDS M32 buffer
    uint32_t ldsTextBytes;     // capacity of the in-LDS copy of the packing (multiple of 16)
    uint32_t ldsStageBytes;    // k_canon_decode: bytes of value staging behind the text copy (gf_canon_decode_lds_stage)
    int phaseLimit;            // diagnostic: stop after phase 0/1/2 (value 1/2/3); 0 = run everything
    uint32_t *debug;           // diagnostic: 16 cycle stamps per tile, normally null
    int rawM32;                // 1: the container holds the M32 bytes themselves behind the 10-byte header (CodecDeflate after inflate)
    const uint32_t *trees;     // non-null: the Huffman trees were parsed by k_huffman_parse_trees (GF_TREE_REC_WORDS per tile)
    uint32_t *retryFlag;       // non-null (CodecHuffman batches with tree records): two device words; the fast kernel ORs 1 into
                               // word (ldsM32Roomy ? 1 : 0) for every tile it leaves to the general kernel (status GF_K_RETRY
                               // inside the launch only), which looks at that word.  With ldsM32Roomy the fast kernel runs twice,
                               // each run taking the tiles the pre-pass gave it (GF_TREE_ROOMY in the tile's tree record)
    uint32_t ldsM32Roomy;      // 0, or the M32 capacity of the fast kernel's second run (round 4): tiles whose M32 stream or
                               // packing outgrows ldsM32Bytes (dense in multi-byte values) get a workgroup with more LDS
                               // instead of the general kernel and its workspace in global memory
    const uint32_t *roomyList; // the tiles of the roomy run, listed by the pre-pass; retryFlag[2] = their count, retryFlag[3] = the
                               // cursor the run's workgroups draw from (both zeroed again by the general kernel, the batch's last)
    uint32_t *roomySeenHost;   // may be null: a word of page-locked host memory that receives 1 + that count from the general kernel --
                               // the host's hint for the NEXT batch (does the roomy run have work, i.e. is it worth a second stream?)
    int flagsCleared;          // 1: the tree pre-pass zeroed retryFlag (gf_launch_huffman_parse_trees), no memset in front of the kernels
    int lean;                  // 1 (the one-tile-per-call path): the fast kernel alone; what it leaves behind keeps the status
                               // GF_K_LEAN_RETRY and the caller takes the batch path for it
    uint32_t *analysis;        // non-null: CodecHuffman.analyze mode -- per tile GF_ANALYSIS_WORDS words (predictor, nM32,
                               // bits in tree, packing bytes - 10, 256-bin histogram of the M32 bytes); no values are written
    uint32_t *pairCounts;      // analyze mode, may be null: GF_PAIR_TABLES x 65536 counters, [predictor][prior << 8 | value] of
                               // neighbouring M32 bytes (CodecStats.addCountsForM32 :150-156), added to with atomics
    int noRoomyRun;            // (set by gf_launch_huffman_decode, round 6) the roomy run is not launched for this batch: the first run
                               // takes the tiles the pre-pass gave to it as well (what it cannot hold goes to the general kernel)
};
constexpr int GF_ANALYSIS_WORDS = 260;
constexpr int GF_PAIR_TABLES = 5;               // one 65536-bin table of byte pairs per predictor code (CodecStats.sB)
// per-tile record of the tree pre-pass: 8 header words (status, leaves, bit position of the first code, longest code,
// single-symbol value or -1, 3 spare), then 256 x (path bits uint64), 256 x code length, 256 x symbol
constexpr int GF_TREE_REC_WORDS = 8 + 512 + 64 + 64;
// fastBytes / roomyBytes (round 5): GfDecodeArgs::ldsM32Bytes / ldsM32Roomy of the decode launch that follows -- the pre-pass marks
// the tiles whose M32 stream or packing outgrows the first but fits the second (GF_TREE_ROOMY in word 3 of the tile's record) and
// lists them (roomyList; count in clearFlags[2]), so that the fast kernel's two runs know their tiles BEFORE either starts and
// can run side by side (GfSideStream)
hipError_t gf_launch_huffman_parse_trees(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, size_t slotStride,
                                         const uint32_t *lengths, uint32_t *trees, size_t nTiles, hipStream_t stream,
                                         uint32_t *clearFlags = nullptr,      // clearFlags: GfDecodeArgs::retryFlag, words 0 and 1 zeroed by the pre-pass
                                         uint32_t fastBytes = 0, uint32_t roomyBytes = 0, uint32_t *roomyList = nullptr);
constexpr uint32_t GF_TREE_ROOMY = 0x400u;          // word 3 of a tree record (beside GF_TREE_HAS_*): the tile belongs to the roomy run
// A second stream of the context with the two events that fork it off the caller's stream and join it again: the fast decode
// kernel's roomy run (some per cent of the tiles of rough terrain, none of smooth terrain) runs there, beside the first run
// instead of behind it.  Event record / wait only: safe inside a stream capture (the side stream joins the capture and leaves it).
struct GfSideStream {
    hipStream_t stream;
    hipEvent_t fork, join;
};

// LSOP12 containers whose entropy stage is CodecM32 bytes: type 0 (legacy Huffman of the two M32 streams) and, with
// rawM32 = 1, type 1 after the host inflated it (gvrs_decode.hip: k_lsop_unpack_m32)
struct GfLsopM32Args {
    const uint8_t *blob;       // 4-byte aligned
    size_t blobBytes;
    const uint64_t *offsets;   // may be null -> t * slotStride
    size_t slotStride;
    const uint32_t *lengths;
    int32_t *residuals;        // nTiles * resStride ints: initialisers, then interior
    size_t resStride;
    uint32_t *coefs;           // nTiles * 16 words: seed, 12 float bit patterns
    int32_t *status;           // in: only tiles marked GF_K_ERR_UNSUPPORTED are touched; out: their decode status
    uint8_t *workspace;        // gridDim.x * workspaceStride bytes (M32 spill)
    size_t workspaceStride;
    size_t nTiles;
    int nRows, nCols;
    uint32_t ldsM32Bytes;
    int rawM32;                // 1: type-1 containers hold the inflated M32 bytes of both streams behind the header
                               // 2: the inflated M32 bytes of tile t are at rawSide + t * rawSideStride (device inflate)
    const uint8_t *rawSide;
    size_t rawSideStride;
    const int32_t *sideStatus;     // mode 2, per tile (k_lsop_streams): 0 = inflated, < 0 = the first stream's failure
    const uint32_t *produced2;     // mode 2: bytes the second stream gave
    const int32_t *inflStatus2;    // mode 2: the second stream's status
};

hipError_t gf_launch_lsop_unpack_m32(const GfLsopM32Args &a, hipStream_t stream, unsigned grid);

hipError_t gf_launch_huffman_encode(const GfEncodeArgs &a, hipStream_t stream);
hipError_t gf_launch_huffman_encode_lean_t1024(const GfEncodeArgs &a, hipStream_t stream);   // 1024-thread workgroups, GfEncodeArgs::lean only
hipError_t gf_launch_huffman_decode(const GfDecodeArgs &a, hipStream_t stream, unsigned grid, const GfSideStream *side = nullptr);
hipError_t gf_launch_huffman_decode_t512(const GfDecodeArgs &a, hipStream_t stream, unsigned grid, const GfSideStream *side = nullptr);   // 512-thread workgroups
hipError_t gf_launch_huffman_decode_t1024(const GfDecodeArgs &a, hipStream_t stream, unsigned grid, const GfSideStream *side = nullptr);  // 1024-thread workgroups
// CodecCanonHuffman packings without escapes through the fast body (DEC_FAST_CANON in gvrs_decode.hip); the tiles it leaves carry
// GF_K_RETRY and a.retryFlag[0] != 0: k_canon_decode (launched with the same retryFlag) takes those
hipError_t gf_launch_huffman_decode_canon(const GfDecodeArgs &a, hipStream_t stream);
hipError_t gf_launch_huffman_decode_canon_t512(const GfDecodeArgs &a, hipStream_t stream);
hipError_t gf_launch_huffman_decode_canon_t1024(const GfDecodeArgs &a, hipStream_t stream);
size_t gf_huffman_decode_lds_per_wg(const GfDecodeArgs &a);           // LDS bytes per workgroup, 256-thread build
size_t gf_huffman_decode_lds_per_wg_t512(const GfDecodeArgs &a);      // ... 512-thread build
size_t gf_huffman_decode_lds_per_wg_t1024(const GfDecodeArgs &a);     // ... 1024-thread build
unsigned gf_huffman_decode_grid(size_t nTiles);
uint32_t gf_huffman_decode_lds_m32(int nRows, int nCols);
uint32_t gf_huffman_decode_lds_text(int nRows, int nCols);

// per-tile record of the canonical decoder's code-length pre-pass: 8 header words (status, bit position of the text
// relative to the packing, 6 spare), then the 261 code lengths (CanonHuffTreeDecoder.decodeTree) padded to 272 bytes
constexpr int GF_CANON_REC_WORDS = 8 + 68;
hipError_t gf_launch_canon_parse_lengths(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, size_t slotStride,
                                         const uint32_t *lengths, uint32_t *recs, size_t nTiles, int lsopContainer,
                                         hipStream_t stream, uint32_t *clearFlags = nullptr);   // clearFlags: GfDecodeArgs::retryFlag, zeroed

// CodecCanonHuffman (gvrs_canon_encode.hip / gvrs_canon_decode.hip); same argument blocks as the legacy codec
hipError_t gf_launch_canon_encode(const GfEncodeArgs &a, hipStream_t stream);
size_t gf_canon_stat_words();          // words per tile of GfEncodeArgs::encStats for the canonical encoder
size_t gf_canon_pack_rec_words();      // words per tile of GfEncodeArgs::packRecs for the canonical encoder
hipError_t gf_launch_canon_decode(const GfDecodeArgs &a, hipStream_t stream, unsigned grid);
uint32_t gf_canon_decode_lds_text(int nRows, int nCols);
// the 512-thread build of the canonical decoder (gvrs_canon_decode.hip compiled with -DGF_CD_THREADS=512 -DGF_CD_VARIANT)
hipError_t gf_launch_canon_decode_t512(const GfDecodeArgs &a, hipStream_t stream, unsigned grid);
uint32_t gf_canon_decode_lds_text_t512(int nRows, int nCols);
uint32_t gf_canon_decode_lds_stage_t512(int nRows, int nCols);
size_t gf_canon_decode_lds_per_wg(const GfDecodeArgs &a);
size_t gf_canon_decode_lds_per_wg_t512(const GfDecodeArgs &a);
uint32_t gf_lsop_unpack_lds_text(int nRows, int nCols);      // the same for k_lsop_unpack2 (trimmed where that gains a workgroup per CU)
uint32_t gf_canon_decode_lds_stage(int nRows, int nCols);

// status (optional): tiles whose status is not GF_K_OK take no room in the blob
hipError_t gf_launch_compact(size_t nTiles, const uint8_t *slots, size_t slotStride,
                             const uint32_t *lengths, uint64_t *offsets, uint8_t *blob,
                             size_t blobCap, hipStream_t stream, const int32_t *status = nullptr);
hipError_t gf_launch_synth_dem(uint64_t seed, int nRows, int nCols, int64_t tilesPerRow,
                               int64_t tile0, size_t nTiles, int32_t *values, hipStream_t stream, int maskPerMille = 0, int style = 0);

// CodecFloat byte planes (gvrs_float.hip); plane buffer of a tile = ceil(n/8) + 4n bytes at planeStride
hipError_t gf_launch_float_planes_encode(const uint32_t *raw, uint8_t *planes, size_t planeStride, size_t nTiles, int nRows,
                                         int nCols, hipStream_t stream);
hipError_t gf_launch_float_planes_decode(const uint8_t *planes, uint32_t *raw, size_t planeStride, size_t nTiles, int nRows,
                                         int nCols, hipStream_t stream);

// LSOP12 (gvrs_lsop.hip, gvrs_lsop_decode.hip).  residuals: per tile resStride ints [initialisers | interior];
// coefs: per tile 16 words (seed, 12 float bit patterns, 3 spare)
hipError_t gf_launch_lsop_predict(const int32_t *values, int32_t *residuals, size_t resStride, uint32_t *coefs,
                                  int32_t *status, size_t nTiles, int nRows, int nCols, hipStream_t stream);
hipError_t gf_launch_canon_pack2(const int32_t *residuals, size_t resStride, const uint32_t *coefs, const int32_t *inStatus,
                                 uint8_t *out, size_t slotStride, uint32_t *lengths, int32_t *status, size_t nTiles,
                                 uint32_t n0, uint32_t n1, int codecIndex, hipStream_t stream, int valueChecksum = 0,
                                 const uint32_t *hist16 = nullptr);
// the encoder's first kernel with the tile held in LDS as halfwords (round 5): residuals as int16 in the same per-tile regions of
// `residuals`, histogram records (gf_lsop_hist_rec_words() words per tile) for gf_launch_canon_pack2's hist16; status receives
// internal codes that only gf_launch_canon_pack2 understands
bool gf_lsop_predict16_eligible(int nRows, int nCols);
size_t gf_lsop_hist_rec_words();
hipError_t gf_launch_lsop_predict16(const int32_t *values, int32_t *residuals, size_t resStride, uint32_t *coefs, int32_t *status,
                                    uint32_t *hist, size_t nTiles, int nRows, int nCols, hipStream_t stream);
// LsHeader.computeChecksum for a batch: word 13 of every tile's coefficient record (16 words) receives the CRC-32C of its values
hipError_t gf_launch_lsop_value_crc(const int32_t *values, size_t nCells, size_t nTiles, const int32_t *inStatus, uint32_t *coefs,
                                    hipStream_t stream);
hipError_t gf_launch_lsop_reconstruct(const int32_t *residuals, size_t resStride, const uint32_t *coefs, const int32_t *inStatus,
                                      int32_t *values, int32_t *status, size_t nTiles, int nRows, int nCols,
                                      hipStream_t stream, bool planes = false);    // planes: word GF_LSOP_FMT_WORD of a tile's
                                                                                   // coefficient record says where its interior residuals are
// The interior residuals of a decoded tile as a BYTE PLANE in pipeline order (round 6).  k_lsop_reconstruct_plane walks a tile as
// k_lsop_reconstruct_pipe does, with HALF a wave: lane l (0..31) takes rows 2 + l, 2 + 32 + l, ..., three steps behind lane l - 1, a
// new row every P steps (two tiles share a wave) -- and at step s = 16 b + k a lane wants the residual of ITS cell of that step.
// k_lsop_unpack2 therefore leaves the residuals, which it has in LDS as bytes anyway, in exactly that order: byte (b * 32 + l) * 16 + k
// of the plane belongs to lane l at step 16 b + k (steps at which a lane has no interior cell are holes that nobody writes or looks
// at), so that a lane's sixteen residuals of a round are ONE 16-byte load, a tile's 512 bytes in a row -- where the int32 array
// cost four bytes per residual in 64-byte row pieces on both sides (profiles/hbm_traffic.json of round 5: 1.28 GB written, 2.52 GB
// read for 0.93 GB of values).  The plane lies in the tile's residual slot behind the initialisers (which stay int32); a tile with a
// residual outside -127..127 keeps the int32 array and the old kernels (word GF_LSOP_FMT_WORD of its coefficient record: 0 = int32
// array, 1 = plane).
constexpr uint32_t GF_LSOP_FMT_WORD = 14;
constexpr uint32_t GF_LSOP_PLANE_LANES = 32;       // lanes per tile
constexpr uint32_t GF_LSOP_PIPE_MIN_P = 112;       // lane 31 starts 93 steps into a period; lane 0 reads 2 + 16 columns ahead of it
constexpr uint32_t GF_LSOP_PLANE_FRONT = 96;       // lanes 30 / 31 "write" their columns -93.. before they start: words nobody reads
struct GfLsopPlaneGeom {
    uint32_t P;            // steps between two rows of a lane (a multiple of 16)
    uint32_t nPh;          // rows per lane
    uint32_t sEnd;         // the last step that produces a value
    uint32_t nBlocks;      // 16-step blocks: the plane is nBlocks x 32 lanes x 16 bytes
    uint32_t offWords;     // where the plane starts in the tile's residual slot (int32 words; a multiple of 4)
    uint32_t rowsWords;    // k_lsop_reconstruct_plane's LDS per tile: lane 0's two rows above (interleaved, GF_LSOP_PLANE_FRONT columns in front) ...
    uint32_t ldsBytes;     // ... for the two tiles of a wave, + the wave's value stage (33 words per lane) and the rows' store descriptors
    bool ok;               // the shape takes the plane path at all
};
__host__ __device__ inline GfLsopPlaneGeom gf_lsop_plane_geom(uint32_t nR, uint32_t nC)
{
    GfLsopPlaneGeom g{};
    if (nR < 6u || nC < 32u || nC > 4096u || nR > 1024u) return g;      // (narrow tiles: a row's two ends would share a round; tall ones:
                                                                          // k_lsop_unpack2 keeps the rows' 4 (nR - 2) initialisers in the
                                                                          // 4 KB of its token table)
    const uint32_t nInit = 4u * nR + 2u * nC - 9u, nInt = (nR - 2u) * (nC - 4u), L = GF_LSOP_PLANE_LANES;
    g.P = ((nC > GF_LSOP_PIPE_MIN_P ? nC : GF_LSOP_PIPE_MIN_P) + 15u) & ~15u;
    g.nPh = (nR - 2u + L - 1u) / L;
    const uint32_t nLast = nR - 2u - L * (g.nPh - 1u);
    g.sEnd = (g.nPh - 1u) * g.P + 3u * (nLast - 1u) + nC - 1u;
    g.nBlocks = g.sEnd / 16u + 1u;
    g.offWords = (nInit + 3u) & ~3u;
    g.rowsWords = 2u * (GF_LSOP_PLANE_FRONT + g.P + 32u);
    g.ldsBytes = (2u * g.rowsWords + 64u * 33u + 128u) * 4u;
    g.ok = (uint64_t)g.offWords * 4u + (uint64_t)g.nBlocks * (L * 16u) <= ((uint64_t)nInit + nInt) * 4u && g.ldsBytes <= 64u * 1024u;
    return g;
}
hipError_t gf_launch_lsop_unpack2(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, size_t slotStride,
                                  const uint32_t *lengths, int32_t *residuals, size_t resStride, uint32_t *coefs,
                                  int32_t *status, size_t nTiles, int nRows, int nCols, uint32_t ldsTextBytes, unsigned grid,
                                  hipStream_t stream, const uint32_t *pre = nullptr,    // pre: records of the first stream's lengths
                                  uint32_t *debug = nullptr,                            // debug: cycle stamps (diagnostic flavour)
                                  uint32_t *pre2 = nullptr,     // room for k_lsop_head's records of the second stream (as many as pre): the
                                                                // serial parts of a tile -- its first stream, the second one's code lengths --
                                                                // then run a lane per tile in front of k_lsop_unpack2
                                  bool planes = false);         // interior residuals as byte planes where a tile allows (see above)

// zlib streams inflated on the GPU, one wave per stream (gvrs_inflate.hip)
struct GfInflateStream {
    uint64_t inOffset;         // the stream (2-byte zlib header first) starts at inBase + inOffset
    uint64_t outOffset;        // its output goes to outBase + outOffset
    uint32_t inLen;            // bytes of input that belong to the stream
    uint32_t outCap;           // room for output (Inflater.inflate(byte[]) semantics: produce at most this much)
};
struct GfInflateArgs {
    const uint8_t *inBase;
    uint8_t *outBase;
    const GfInflateStream *streams;
    uint32_t *produced;        // per stream: bytes written
    int32_t *status;           // per stream: GF_K_OK or GF_K_ERR_FORMAT (what makes Inflater throw DataFormatException)
    size_t nStreams;
    uint32_t window;           // LDS window per wave: gf_inflate_window(largest outCap)
    uint32_t *consumed;        // may be null; per stream: bytes of input used (Inflater.getTotalIn()): where a following stream starts
    const uint32_t *gate;      // may be null; a device word: zero = no stream of this launch has any input, every wave leaves at once
};
uint32_t gf_inflate_window(uint32_t maxOut);
hipError_t gf_launch_inflate(const GfInflateArgs &a, hipStream_t stream);
// container walkers, one thread per tile: stream descriptors of CodecDeflate / CodecFloat packings, status merging
hipError_t gf_launch_deflate_streams(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, size_t slotStride,
                                     const uint32_t *lengths, size_t tile0, size_t nTiles, uint32_t cells, uint8_t *raw, size_t rawStride,
                                     GfInflateStream *desc, int32_t *pre, hipStream_t stream);
hipError_t gf_launch_deflate_lengths(size_t nTiles, const GfInflateStream *desc, const uint32_t *produced, const int32_t *inflStatus,
                                     int32_t *pre, uint32_t *rawLengths, uint8_t *rawBase, hipStream_t stream);
hipError_t gf_launch_merge_status(size_t nTiles, const int32_t *pre, const int32_t *decoded, int32_t *status, hipStream_t stream);
hipError_t gf_launch_float_streams(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, const uint32_t *lengths, size_t tile0,
                                   size_t nTiles, uint32_t cells, size_t planeStride, GfInflateStream *desc, int32_t *pre, hipStream_t stream);
hipError_t gf_launch_lsop_streams(const uint8_t *blob, size_t blobBytes, const uint64_t *offsets, size_t slotStride, const uint32_t *lengths,
                                  size_t nTiles, uint32_t nInit, uint32_t nInt, size_t rawStride, int pass, const uint32_t *produced,
                                  const int32_t *inflStatus, const uint32_t *consumed, GfInflateStream *desc, int32_t *side, uint32_t *gate,
                                  hipStream_t stream);
hipError_t gf_launch_float_short_planes(size_t nTiles, const int32_t *pre, const int32_t *inflStatus, const uint32_t *produced,
                                        uint8_t *planes, size_t planeStride, int nRows, int nCols, hipStream_t stream);
hipError_t gf_launch_float_status(size_t nTiles, const int32_t *pre, const int32_t *inflStatus, int32_t *status, hipStream_t stream);

// predictor -> M32 stage alone (gvrs_encode.hip): per tile up to three candidate M32 streams (CodecDeflate.java:157-199)
struct GfM32Args {
    const int32_t *values;
    uint8_t *out;              // per tile 3 sub-slots of subStride bytes (candidates in the order D, L, T; nulls: slot 0)
    size_t subStride;          // multiple of 16
    uint32_t *lengths;         // 3 per tile: M32 bytes of the candidate (0 = not a candidate)
    uint8_t *models;           // 3 per tile: predictor code of the candidate or 0
    uint32_t *seeds;           // 1 per tile
    int32_t *status;           // GF_K_OK / DECLINED / OVERFLOW (a candidate did not fit; its length is still exact) / ERR_BOUNDS
    size_t nTiles;
    int nRows, nCols;
};
hipError_t gf_launch_m32_streams(const GfM32Args &a, hipStream_t stream);
