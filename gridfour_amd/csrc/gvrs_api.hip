// gvrs_api.hip -- the C ABI of include/gvrs_hip_codec.h: context, device-resident batch
// entry points (what bench.py measures) and the host-memory entry points a JNI / FFI
// binding of ICompressionEncoder / ICompressionDecoder calls.  No CPU fallback anywhere:
// every compute call needs a HIP device.

#include <hip/hip_runtime.h>

#include <atomic>

#include <algorithm>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>
#include <zlib.h>

#include "../../include/gvrs_hip_codec.h"
#include "gvrs_kernels.h"
#include "gvrs_encode_layout.h"

namespace {

thread_local std::string g_lastError;
#ifdef GF_DIAG
// the diagnostic flavour of the library only (libgvrs_hip_diag.so, tools/): process-wide hooks for phase ablation and
// the kernels' cycle stamps.  The shipping library has no mutable global state.
int g_encPhaseLimit = 0, g_decPhaseLimit = 0;
uint32_t *g_decodeDebug = nullptr;   // 16 cycle stamps per tile
uint32_t *g_encodeDebug = nullptr;   // dump target of the next encode launches
#else
constexpr int g_encPhaseLimit = 0, g_decPhaseLimit = 0;
constexpr uint32_t *g_decodeDebug = nullptr, *g_encodeDebug = nullptr;
#endif

gf_status hipFail(hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    g_lastError = buf;
    return GF_ERR_HIP;
}

#define GF_HIP(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) return hipFail(e_, #call); \
    } while (0)

size_t roundUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

// offsets[nTiles + 1] of a caller's blob: non-decreasing, every packing shorter than 4 GiB (lengths travel as uint32)
bool offsetsValid(const uint64_t *offsets, size_t nTiles)
{
    for (size_t t = 0; t < nTiles; t++)
        if (offsets[t + 1] < offsets[t] || offsets[t + 1] - offsets[t] > 0xFFFFFFFFull) return false;
    return true;
}

// Counts the device buffers that moved (a move is rare: buffers only grow).  A recorded hipGraph holds the addresses it was
// captured with: the one-tile graphs (gf_single) remember the count they were recorded at and are dropped when it has changed --
// a batch that grew the context's tree / selection records between two replays used to leave them pointing at freed memory.
// Round 6 (advice): a context's buffers count on the CONTEXT's counter (gf_context::bufMoves) -- with one counter for the process
// another GPU's shard or another thread's batch made every context drop and re-record its graphs; this one is what is left for
// buffers that belong to no context.
static std::atomic<uint64_t> g_devBufMoves{0};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    std::atomic<uint64_t> *moves = &g_devBufMoves;
    gf_status ensure(size_t need)
    {
        if (need <= bytes) return GF_OK;
        moves->fetch_add(1, std::memory_order_relaxed);
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        need = roundUp(need + need / 8, 1 << 20);
        GF_HIP(hipMalloc(&p, need));
        bytes = need;
        return GF_OK;
    }
    void release()
    {
        if (p) {
            moves->fetch_add(1, std::memory_order_relaxed);
            (void)hipFree(p);
        }
        p = nullptr;
        bytes = 0;
    }
};

}  // namespace

struct gf_context {
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuf workspace;      // decode spill: grid * 6*cells
    DevBuf trees;          // leaf records of the tree pre-pass: GF_TREE_REC_WORDS per tile
    DevBuf flags;          // one word: tiles the fast decode kernel left to the general one (GfDecodeArgs::retryFlag)
    DevBuf packRecs;       // encoder: selection records between k_huffman_encode and k_huffman_pack
    // staging for the host-memory entry points
    DevBuf dValues, dSlots, dBlob, dLengths, dPred, dStatus, dOffsets;
    DevBuf dPlanes;        // CodecFloat plane staging
    DevBuf dResiduals, dCoefs, dStatus2;   // LSOP staging
    DevBuf dM32, dM32Len, dM32Models, dSeeds;   // CodecDeflate staging
    DevBuf dInflate, dInflOut, dInflMeta;       // GPU inflate: stream descriptors, inflated bytes, produced / status
    std::atomic<uint64_t> bufMoves{0};          // moves of THIS context's device buffers (DevBuf::moves): what its recorded graphs watch
    gf_context()
    {
        for (DevBuf *b : {&workspace, &trees, &flags, &packRecs, &dValues, &dSlots, &dBlob, &dLengths, &dPred, &dStatus, &dOffsets, &dPlanes,
                          &dResiduals, &dCoefs, &dStatus2, &dM32, &dM32Len, &dM32Models, &dSeeds, &dInflate, &dInflOut, &dInflMeta})
            b->moves = &bufMoves;
    }
    // Every entry point that takes the context holds this lock for its duration (GF_CTX_LOCK): the reference calls ONE decoder
    // instance from several threads (gvrs/RasterTileCache.java:418-421, TileDecompressionAssistant.java:68-73), and a context's
    // scratch buffers, staging slots and recorded graphs are one set.  Recursive: entry points call each other.
    std::recursive_mutex mu;
    GfSideStream side{nullptr, nullptr, nullptr};   // second stream + fork / join events: the fast decode kernel's roomy run (gvrs_kernels.h)
    uint32_t *hRoomySeen = nullptr;                 // page-locked word: GfDecodeArgs::roomySeenHost
    struct gf_host_pipe *pipe = nullptr;        // pipelined staging of the host-memory batch entry points (created on first use)
    struct gf_single *single = nullptr;         // one tile per call: page-locked buffers and replayed graphs (created on first use)
};
void gf_host_pipe_destroy(struct gf_host_pipe *p);
void gf_single_destroy(struct gf_single *s);
#define GF_CTX_LOCK(c)                                       \
    std::unique_lock<std::recursive_mutex> gfCtxLock_;       \
    if (c) gfCtxLock_ = std::unique_lock<std::recursive_mutex>((c)->mu)

struct gf_timer {
    gf_context *ctx;
    hipEvent_t start, stop;
};

// Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a container that shows 256 CPUs may
// be allowed 16 cores' worth of time; more threads than that only take turns and thrash the caches).
static unsigned hostCores()
{
    static const unsigned cached = []() -> unsigned {
        unsigned n = std::thread::hardware_concurrency();
        if (n == 0) n = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
            char q[64];
            long long period = 0;
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
                n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, atoll(q) / period));
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
            long long quota = -1, period = 0;
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(h, "%lld", &period) != 1) period = 0;
                fclose(h);
            }
            if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, quota / period));
        }
        return std::min(n, 256u);
    }();
    return cached;
}

// tiles are independent: the host-side zlib stages run on every core the process may use
template <class F>
static void parallelFor(size_t n, F f)
{
    unsigned nt = hostCores();
    if (nt > n) nt = (unsigned)n;
    if (nt == 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nt; w++)
        th.emplace_back([=]() { for (size_t i = w; i < n; i += nt) f(i); });
    for (auto &x : th) x.join();
}


extern "C" {

const char *gf_version(void) { return "gvrs-hip-codec 0.1 (gfx950)"; }

#ifdef GF_DIAG
// Not part of the public ABI (diagnostic flavour only): lets tools/ capture the encode kernel's on-chip tables
// (gvrs_encode_layout.h).  d_words must hold gf_internal_encode_debug_words() uint32 per tile.
void gf_internal_set_encode_debug(void *d_words) { g_encodeDebug = (uint32_t *)d_words; }
void gf_internal_set_decode_debug(void *d_words) { g_decodeDebug = (uint32_t *)d_words; }
void gf_internal_set_phase_limits(int enc, int dec) { g_encPhaseLimit = enc; g_decPhaseLimit = dec; }
size_t gf_internal_encode_debug_words(void) { return GF_ENC_DEBUG_WORDS; }
#endif

const char *gf_status_string(int s)
{
    switch (s) {
    case GF_OK: return "ok";
    case GF_DECLINED: return "declined (encoder returns null)";
    case GF_OVERFLOW: return "packing larger than the output slot";
    case GF_ERR_FORMAT: return "format error (IOException in the reference)";
    case GF_ERR_BOUNDS: return "out of bounds (ArrayIndexOutOfBounds in the reference)";
    case GF_ERR_CAPACITY: return "output buffer too small";
    case GF_ERR_ARG: return "bad argument";
    case GF_ERR_NO_DEVICE: return "no HIP device";
    case GF_ERR_HIP: return "HIP runtime error";
    case GF_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
    }
}

const char *gf_last_error(void) { return g_lastError.c_str(); }
// (library-internal: gvrs_multi.hip hands the text of a failing shard's thread to the thread that called gf_*_multi)
// (an internal hook of gvrs_multi.hip, not part of the ABI: hidden from the library's exports)
__attribute__((visibility("hidden"))) void gf_internal_set_last_error(const char *text) { g_lastError = text ? text : ""; }

int gf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

gf_status gf_context_create(int device, gf_context **out)
{
    if (!out) return GF_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_lastError = "no HIP device visible";
        return GF_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return GF_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return GF_ERR_NO_DEVICE;
    gf_context *c = new (std::nothrow) gf_context();
    if (!c) return GF_ERR_ARG;
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        hipFail(e, "hipStreamCreate");
        return GF_ERR_NO_DEVICE;
    }
    gf_status s = c->flags.ensure(64);
    if (s != GF_OK) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return s;
    }
    // (words 2 and 3 -- count and cursor of the roomy decode run's tile list -- start at zero; the kernels leave them so)
    if (hipMemset(c->flags.p, 0, 64) != hipSuccess) {
        (void)hipGetLastError();
        c->flags.release();
        (void)hipStreamDestroy(c->stream);
        delete c;
        return GF_ERR_HIP;
    }
    if (hipHostMalloc((void **)&c->hRoomySeen, 64, hipHostMallocDefault) == hipSuccess) *c->hRoomySeen = 0u;
    else { (void)hipGetLastError(); c->hRoomySeen = nullptr; }
    // (the side stream is an optimisation: without it the roomy run follows the first one on the caller's stream)
    if (hipStreamCreateWithFlags(&c->side.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->side.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->side.join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (c->side.fork) (void)hipEventDestroy(c->side.fork);
        if (c->side.stream) (void)hipStreamDestroy(c->side.stream);
        c->side = GfSideStream{nullptr, nullptr, nullptr};
    }
    *out = c;
    return GF_OK;
}

void gf_context_destroy(gf_context *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    gf_single_destroy(c->single);
    c->single = nullptr;
    c->workspace.release();
    c->trees.release();
    c->flags.release();
    c->packRecs.release();
    c->dValues.release();
    c->dSlots.release();
    c->dBlob.release();
    c->dLengths.release();
    c->dPred.release();
    c->dStatus.release();
    c->dOffsets.release();
    c->dPlanes.release();
    c->dResiduals.release();
    c->dCoefs.release();
    c->dStatus2.release();
    c->dM32.release();
    c->dM32Len.release();
    c->dM32Models.release();
    c->dSeeds.release();
    c->dInflate.release();
    c->dInflOut.release();
    c->dInflMeta.release();
    gf_host_pipe_destroy(c->pipe);
    if (c->hRoomySeen) (void)hipHostFree(c->hRoomySeen);
    if (c->side.stream) {
        (void)hipStreamSynchronize(c->side.stream);
        (void)hipEventDestroy(c->side.fork);
        (void)hipEventDestroy(c->side.join);
        (void)hipStreamDestroy(c->side.stream);
    }
    (void)hipStreamDestroy(c->stream);
    delete c;
}

void *gf_context_stream(gf_context *c) { return c ? (void *)c->stream : nullptr; }

gf_status gf_context_synchronize(gf_context *c)
{
    GF_CTX_LOCK(c);
    if (!c) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    GF_HIP(hipStreamSynchronize(c->stream));
    return GF_OK;
}

static size_t decodeWorkspaceStride(int nRows, int nCols)
{
    // spill layout of the decode kernel: M32 bytes (6 per cell, rounded to 32), start bitmap, rank bases
    const size_t cap = roundUp((size_t)6 * (size_t)nRows * (size_t)nCols, 32);
    return roundUp(cap + 2 * ((cap >> 5) + 2) * 4 + 64, 16);
}

// the encoders' per-tile records between their kernels (selection records, statistics) and, behind them, the legacy encoder's byte
// plane of raw row differences (GfEncodeArgs::plane): one allocation of the context
static size_t encPlaneStride(int nRows, int nCols) { return roundUp((size_t)nRows * (size_t)nCols, 16); }
static size_t encRecordBytes(int nRows, int nCols, size_t nTiles)
{
    const size_t recs = nTiles * std::max((size_t)GF_PACK_REC_WORDS + GF_ENC_STAT_WORDS, gf_canon_pack_rec_words() + gf_canon_stat_words()) * 4 + 16;
    return roundUp(recs, 256) + nTiles * encPlaneStride(nRows, nCols) + 256;
}

gf_status gf_context_reserve(gf_context *c, int nRows, int nCols, size_t nTiles)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    const unsigned grid = gf_huffman_decode_grid(nTiles);
    // (the legacy decoder's leaf records + its roomy list; the canonical decoder's length records + the slack of its fast run)
    gf_status s = c->trees.ensure(std::max(nTiles * (size_t)GF_TREE_REC_WORDS * 4 + 16 + nTiles * 4,
                                           nTiles * (size_t)GF_CANON_REC_WORDS * 4 + 16 + 4096));
    if (s != GF_OK) return s;
    if ((s = c->packRecs.ensure(encRecordBytes(nRows, nCols, nTiles))) != GF_OK) return s;
    return c->workspace.ensure((size_t)grid * decodeWorkspaceStride(nRows, nCols));
}

size_t gf_huffman_default_stride(int nRows, int nCols)
{
    return roundUp((size_t)4 * (size_t)nRows * (size_t)nCols + 1024, 16);
}

size_t gf_huffman_max_packing(int nRows, int nCols)
{
    // 80 header bits + tree (8 + 10*256 - 1) + 6 M32 bytes per cell at <= 46 bits per code
    // (depth d needs Fib(d+2) symbols; nM32 < 2^31 bounds d by 44)
    const size_t cells = (size_t)nRows * (size_t)nCols;
    const size_t bits = 80 + 8 + 2559 + cells * 6 * 46;
    return roundUp((bits + 7) / 8 + 16, 16);
}

// ------------------------------------------------------------------ device-resident

}  // extern "C"

// codec kinds behind the shared batch plumbing
enum { KIND_HUFFMAN = 0, KIND_CANON = 1, KIND_RAW_M32 = 2, KIND_DEFLATE = 3, KIND_FLOAT = 4 };
static gf_status floatDecodeDev(gf_context *c, hipStream_t st, int nRows, int nCols, size_t nTiles, const uint8_t *dBlob, size_t blobBytes,
                                const uint64_t *dOffsets, const uint32_t *dLengths, float *dValues, int32_t *dStatus);

// set around the device entry points by the one-tile-per-call path (singleEncode / singleDecode): GfEncodeArgs::lean, GfDecodeArgs::lean
static thread_local int g_lean = 0;

static gf_status encodeBatchDev(int kind, gf_context *c, void *stream, int codecIndex, int nRows, int nCols,
                                size_t nTiles, const int32_t *dValues, uint8_t *dOut, size_t slotStride,
                                uint32_t *dLengths, uint8_t *dPredictors, int32_t *dStatus, int predictorMask)
{
    if (!c || nRows < 1 || nCols < 1 || !dValues || !dOut || !dLengths || !dStatus) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    if ((size_t)nRows * (size_t)nCols >= (1ull << 28)) return GF_ERR_UNSUPPORTED;
    if (slotStride % 16 != 0 || ((uintptr_t)dOut & 15) != 0 || slotStride < 16) return GF_ERR_ARG;
    GfEncodeArgs a;
    a.values = dValues;
    a.out = dOut;
    a.lengths = dLengths;
    a.predictors = dPredictors;
    a.status = dStatus;
    a.nTiles = nTiles;
    a.slotStride = slotStride;
    a.nRows = nRows;
    a.nCols = nCols;
    a.codecIndex = codecIndex;
    a.predictorMask = predictorMask & GF_PM_ALL;
    a.debug = g_encodeDebug;
    a.phaseLimit = g_encPhaseLimit;
    a.packRecs = nullptr;
    a.retryFlag = kind == KIND_HUFFMAN ? (uint32_t *)c->flags.p + 4 : nullptr;      // (word 0 belongs to the decoder)
    {
        // (CodecHuffman: the selection records, and behind them the statistics k_huffman_encode hands to k_huffman_trees)
        const size_t need = nTiles * (kind == KIND_CANON ? gf_canon_pack_rec_words() + gf_canon_stat_words() : (size_t)GF_PACK_REC_WORDS + GF_ENC_STAT_WORDS) * 4 + 16;
        const size_t needAll = encRecordBytes(nRows, nCols, nTiles);
        if (c->packRecs.bytes < needAll) {
            GF_HIP(hipSetDevice(c->device));               // not capture-safe: gf_context_reserve sizes this too
            gf_status s = c->packRecs.ensure(needAll);
            if (s != GF_OK) return s;
        }
        a.packRecs = (uint32_t *)c->packRecs.p;
        a.lean = g_lean;
        a.encStats = a.packRecs + nTiles * (kind == KIND_CANON ? gf_canon_pack_rec_words() : (size_t)GF_PACK_REC_WORDS);
        // (round 6) the byte plane of raw row differences between phase A and the packer (GfEncodeArgs::plane), behind the records
        a.plane = nullptr;
        a.planeStride = 0;
#ifndef GF_ENC_NO_PLANE
        if ((kind == KIND_HUFFMAN || kind == KIND_CANON) && !a.lean) {
            a.planeStride = encPlaneStride(nRows, nCols);
            a.plane = (uint8_t *)c->packRecs.p + roundUp(need, 256);
        }
#endif
    }
    if (kind == KIND_CANON) GF_HIP(gf_launch_canon_encode(a, stream ? (hipStream_t)stream : c->stream));
    else if (a.lean && a.retryFlag && 6ull * (size_t)nRows * (size_t)nCols < (1ull << 23))   // one tile per call: the 1024-thread build
        GF_HIP(gf_launch_huffman_encode_lean_t1024(a, stream ? (hipStream_t)stream : c->stream));
    else GF_HIP(gf_launch_huffman_encode(a, stream ? (hipStream_t)stream : c->stream));
    return GF_OK;
}

static gf_status decodeBatchDev(int kind, gf_context *c, void *stream, int nRows, int nCols, size_t nTiles,
                                const uint8_t *dBlob, size_t blobBytes, const uint64_t *dOffsets, size_t slotStride,
                                const uint32_t *dLengths, int32_t *dValues, int32_t *dStatus, uint32_t *analysis = nullptr,
                                uint32_t *pairCounts = nullptr)
{
    if (!c || nRows < 1 || nCols < 1 || !dBlob || !dLengths || !dValues || !dStatus) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    if ((size_t)nRows * (size_t)nCols >= (1ull << 28)) return GF_ERR_UNSUPPORTED;
    if (((uintptr_t)dBlob & 3) != 0) return GF_ERR_ARG;
    const unsigned grid = gf_huffman_decode_grid(nTiles);
    const size_t wsStride = kind == KIND_CANON ? 0 : decodeWorkspaceStride(nRows, nCols);
    if (c->workspace.bytes < (size_t)grid * wsStride) {
        // not capture-safe: callers that capture graphs call gf_context_reserve first
        GF_HIP(hipSetDevice(c->device));
        gf_status s = c->workspace.ensure((size_t)grid * wsStride);
        if (s != GF_OK) return s;
    }
    GfDecodeArgs a{};
    a.blob = dBlob;
    a.blobBytes = blobBytes;
    a.offsets = dOffsets;
    a.slotStride = slotStride;
    a.lengths = dLengths;
    a.values = dValues;
    a.status = dStatus;
    a.workspace = (uint8_t *)c->workspace.p;
    a.workspaceStride = wsStride;
    a.nTiles = nTiles;
    a.nRows = nRows;
    a.nCols = nCols;
    a.phaseLimit = g_decPhaseLimit;
    a.debug = g_decodeDebug;
    a.rawM32 = kind == KIND_RAW_M32 ? 1 : 0;
    a.analysis = analysis;
    a.pairCounts = pairCounts;
    a.trees = nullptr;
    a.retryFlag = nullptr;
    a.lean = g_lean;
    if (kind == KIND_HUFFMAN) {
        // tree pre-pass: one lane per tile walks the serialised tree; the decode kernel starts from the leaf records
        const size_t need = nTiles * (size_t)GF_TREE_REC_WORDS * 4 + 16 + nTiles * 4;      // (+ the roomy run's tile list)
        if (c->trees.bytes < need) {
            GF_HIP(hipSetDevice(c->device));               // not capture-safe either: gf_context_reserve sizes this too
            gf_status s = c->trees.ensure(need);
            if (s != GF_OK) return s;
        }
        uint32_t *roomyList = (uint32_t *)c->trees.p + nTiles * (size_t)GF_TREE_REC_WORDS + 4;
        if (!analysis) a.retryFlag = (uint32_t *)c->flags.p;
        uint32_t fastBytes = 0, roomyBytes = 0;
        if (a.retryFlag && !a.lean) {
            // the fast kernel runs twice: the usual LDS budget (1.125 M32 bytes per cell) and, for the tiles that outgrow it, two bytes
            // per cell; the pre-pass sorts the tiles
            const size_t cells = (size_t)nRows * (size_t)nCols;
            fastBytes = gf_huffman_decode_lds_m32(nRows, nCols);
            const size_t roomy = std::min<size_t>(98304, (2 * cells + 1024 + 31) & ~(size_t)31);
            roomyBytes = roomy > fastBytes ? (uint32_t)roomy : 0u;
#ifdef GF_DEC_NO_ROOMY                                              // (experiment builds)
            roomyBytes = 0u;
#endif
        }
        a.ldsM32Roomy = roomyBytes;
        GF_HIP(gf_launch_huffman_parse_trees(dBlob, blobBytes, dOffsets, slotStride, dLengths, (uint32_t *)c->trees.p, nTiles,
                                             stream ? (hipStream_t)stream : c->stream, a.retryFlag, fastBytes, roomyBytes, roomyList));
        a.roomyList = roomyList;
        a.roomySeenHost = c->hRoomySeen;
        a.trees = (const uint32_t *)c->trees.p;
        a.flagsCleared = a.retryFlag ? 1 : 0;
    }
    if (kind == KIND_CANON) {
        // the same for the canonical decoder's code lengths
        // The canonical run of the fast legacy kernel first (round 5): a packing whose code has no escape, null or spare symbol is a
        // prefix-coded byte string like any other, and that kernel decodes it once (symbol pool, byte path) where k_canon_decode
        // decodes it twice.  For tile shapes its byte path takes; what it leaves (GF_K_RETRY) k_canon_decode picks up.
        const size_t cells = (size_t)nRows * (size_t)nCols;
        const uint32_t fastM32 = gf_huffman_decode_lds_m32(nRows, nCols);
        bool viaFast = !analysis && nRows >= 2 && nCols >= 4 && nCols <= 256 && cells + 8 <= fastM32;
#ifdef GF_CANON_NO_FAST_RUN                                       // (experiment builds: tools/ab.sh)
        viaFast = false;
#endif
#ifdef GF_DIAG
        // (the diagnostic build's phase limits and cycle stamps are k_canon_decode's: tools/phase_cycles_canon.py, pmc_phases_canon.sh)
        if (g_decPhaseLimit || g_decodeDebug) viaFast = false;
#endif
        if (viaFast) a.retryFlag = (uint32_t *)c->flags.p;
        const size_t need = nTiles * (size_t)GF_CANON_REC_WORDS * 4 + 16 + (viaFast ? 4096 : 0);
        if (c->trees.bytes < need) {
            GF_HIP(hipSetDevice(c->device));
            gf_status s = c->trees.ensure(need);
            if (s != GF_OK) return s;
        }
        GF_HIP(gf_launch_canon_parse_lengths(dBlob, blobBytes, dOffsets, slotStride, dLengths, (uint32_t *)c->trees.p, nTiles, 0,
                                             stream ? (hipStream_t)stream : c->stream, a.retryFlag));
        a.trees = (const uint32_t *)c->trees.p;
        if (viaFast) {
            GfDecodeArgs f = a;
            f.ldsM32Bytes = fastM32;
            f.ldsTextBytes = 0;
            f.ldsM32Roomy = 0;
            auto wgsPerCu = [](size_t lds, size_t cap) {
                const size_t step = 1280, n = (160 * 1024) / ((lds + step - 1) / step * step);
                return n < cap ? n : cap;
            };
            // (the build as for a CodecHuffman batch below)
            const size_t waves256 = 4 * wgsPerCu(gf_huffman_decode_lds_per_wg(f), 8), waves512 = 8 * wgsPerCu(gf_huffman_decode_lds_per_wg_t512(f), 4),
                         waves1024 = 16 * wgsPerCu(gf_huffman_decode_lds_per_wg_t1024(f), 2);
            int threads = 2 * waves512 >= 3 * waves256 ? 512 : 256;
            if (threads == 512 && waves1024 >= 2 * waves512) threads = 1024;
            if (a.lean) threads = 1024;                           // (one tile per call: the widest build, as below)
            hipStream_t st = stream ? (hipStream_t)stream : c->stream;
            if (threads == 1024) GF_HIP(gf_launch_huffman_decode_canon_t1024(f, st));
            else if (threads == 512) GF_HIP(gf_launch_huffman_decode_canon_t512(f, st));
            else GF_HIP(gf_launch_huffman_decode_canon(f, st));
        }
    }
    if (kind == KIND_CANON) {
        // two builds as for the legacy decoder below: 256 threads (up to five workgroups per CU) or 512 (four = 32 waves)
        a.ldsM32Bytes = 0;
        GfDecodeArgs b = a;
        a.ldsTextBytes = gf_canon_decode_lds_text(nRows, nCols);
        a.ldsStageBytes = gf_canon_decode_lds_stage(nRows, nCols);
        b.ldsTextBytes = gf_canon_decode_lds_text_t512(nRows, nCols);
        b.ldsStageBytes = gf_canon_decode_lds_stage_t512(nRows, nCols);
        auto wgsPerCu = [](size_t lds, size_t cap) {
            const size_t step = 1280, n = (160 * 1024) / ((lds + step - 1) / step * step);
            return n < cap ? n : cap;
        };
        const size_t waves256 = 4 * wgsPerCu(gf_canon_decode_lds_per_wg(a), 8), waves512 = 8 * wgsPerCu(gf_canon_decode_lds_per_wg_t512(b), 4);
        // ... where the tile is large enough to give most threads a subsequence (at least 128 bits each).  With the fused Triangle
        // inverse and the whole stream staged (late round 3) the 512-thread build wins from about 7,000 cells on: 90x120 tiles
        // 1.69 -> 1.37 ms per 16,000 tiles, 100x110 1.74 -> 1.40, 70x100 2.24 -> 2.22 per 33,000, 64x64 the same either way
        // (before those two: 70x100 2.50 with 256 threads against 2.81 with 512, and the bound was 12,000 cells)
#ifndef GF_CANON_T512_MIN_CELLS
#define GF_CANON_T512_MIN_CELLS 7000
#endif
        if (2 * waves512 >= 3 * waves256 && (size_t)nRows * (size_t)nCols >= GF_CANON_T512_MIN_CELLS)
            GF_HIP(gf_launch_canon_decode_t512(b, stream ? (hipStream_t)stream : c->stream, grid));
        else GF_HIP(gf_launch_canon_decode(a, stream ? (hipStream_t)stream : c->stream, grid));
    } else {
        a.ldsM32Bytes = gf_huffman_decode_lds_m32(nRows, nCols);
        a.ldsTextBytes = gf_huffman_decode_lds_text(nRows, nCols);
        // Occupancy is set by LDS (M32 stream + start bitmap + tables per workgroup), handed out in 1,280-byte steps, and the kernel
        // gains from every wave a CU can hold (tools/occupancy_sweep.sh).  Two builds of the same source: 256 threads (two Huffman
        // cursors per thread in lockstep, the leaner one per wave) and 512 threads (one cursor per thread, 64 VGPRs, up to four
        // workgroups = all 32 wave slots of a CU).  The 512-thread build is the faster one where it holds at least 1.5 times the waves:
        // measured 120x150 (16 against 32 waves) 1.37 -> 1.19 ms per 12,960 tiles, 100x120 1.59 -> 1.36, 200x200 (8 / 16) 3.92 -> 2.89;
        // 70x100 (24 / 32) 1.62 against 1.81 and 32x32 2.63 against 3.45: the 256-thread build stays.
        auto wgsPerCu = [](size_t lds, size_t cap) {
            const size_t step = 1280, n = (160 * 1024) / ((lds + step - 1) / step * step);
            return n < cap ? n : cap;
        };
        const size_t waves256 = 4 * wgsPerCu(gf_huffman_decode_lds_per_wg(a), 8), waves512 = 8 * wgsPerCu(gf_huffman_decode_lds_per_wg_t512(a), 4),
                     waves1024 = 16 * wgsPerCu(gf_huffman_decode_lds_per_wg_t1024(a), 2);
        int threads = 2 * waves512 >= 3 * waves256 ? 512 : 256;
        // 1024 threads where that doubles the waves again (tiles of 160x160 and more: two workgroups of 512 at most on a CU)
        if (threads == 512 && waves1024 >= 2 * waves512) threads = 1024;
#ifdef GF_DEC_LDS_PAD_ENV
        if (const char *e = getenv("GF_DEC_FORCE_THREADS")) threads = atoi(e);   // experiment builds only (tools/occupancy_sweep.sh)
#endif
        // one tile per call: the workgroup is alone on the chip and every phase is a latency chain -- the widest build (120x150:
        // 89 -> 82 us per call against the 512-thread build, 111 with 256 threads)
        if (a.lean) threads = 1024;
        const GfSideStream *side = c->side.stream ? &c->side : nullptr;
#ifdef GF_DEC_ROOMY_BEHIND                                          // (experiment builds: the roomy run behind the first, as in round 4)
        side = nullptr;
#endif
        if (threads == 1024) GF_HIP(gf_launch_huffman_decode_t1024(a, stream ? (hipStream_t)stream : c->stream, grid, side));
        else if (threads == 512) GF_HIP(gf_launch_huffman_decode_t512(a, stream ? (hipStream_t)stream : c->stream, grid, side));
        else GF_HIP(gf_launch_huffman_decode(a, stream ? (hipStream_t)stream : c->stream, grid, side));
    }
    return GF_OK;
}

// code-length pre-pass for the first stream of LSOP12 containers of the canonical type
static gf_status lsopParseLengths(gf_context *c, hipStream_t st, size_t nTiles, const uint8_t *dBlob, size_t blobBytes,
                                  const uint64_t *dOffsets, size_t slotStride, const uint32_t *dLengths)
{
    if (!c) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    const size_t need = 2 * nTiles * (size_t)GF_CANON_REC_WORDS * 4 + 16;     // (both streams' records: k_lsop_head)
    if (c->trees.bytes < need) {
        GF_HIP(hipSetDevice(c->device));                   // not capture-safe: gf_context_reserve sizes this too
        gf_status s = c->trees.ensure(need);
        if (s != GF_OK) return s;
    }
    GF_HIP(gf_launch_canon_parse_lengths(dBlob, blobBytes, dOffsets, slotStride, dLengths, (uint32_t *)c->trees.p, nTiles, 1, st));
    return GF_OK;
}

// second entropy pass of the LSOP12 decode: containers k_lsop_unpack2 left marked GF_ERR_UNSUPPORTED (legacy Huffman of
// M32; with rawM32 also the host-inflated Deflate ones)
static gf_status lsopUnpackM32(gf_context *c, hipStream_t st, int nRows, int nCols, size_t nTiles, const uint8_t *dBlob,
                               size_t blobBytes, const uint64_t *dOffsets, size_t slotStride, const uint32_t *dLengths,
                               int32_t *dResiduals, size_t resStride, uint32_t *dCoefs, int32_t *dScratchStatus, int rawM32,
                               const uint8_t *rawSide = nullptr, size_t rawSideStride = 0, const int32_t *sideStatus = nullptr,
                               const uint32_t *produced2 = nullptr, const int32_t *inflStatus2 = nullptr)
{
    if (!c) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    const unsigned grid = gf_huffman_decode_grid(nTiles);
    const size_t wsStride = decodeWorkspaceStride(nRows, nCols);
    if (c->workspace.bytes < (size_t)grid * wsStride) {
        // not capture-safe: callers that capture graphs call gf_context_reserve first
        GF_HIP(hipSetDevice(c->device));
        gf_status s = c->workspace.ensure((size_t)grid * wsStride);
        if (s != GF_OK) return s;
    }
    GfLsopM32Args a;
    a.blob = dBlob;
    a.blobBytes = blobBytes;
    a.offsets = dOffsets;
    a.slotStride = slotStride;
    a.lengths = dLengths;
    a.residuals = dResiduals;
    a.resStride = resStride;
    a.coefs = dCoefs;
    a.status = dScratchStatus;
    a.workspace = (uint8_t *)c->workspace.p;
    a.workspaceStride = wsStride;
    a.nTiles = nTiles;
    a.nRows = nRows;
    a.nCols = nCols;
    a.ldsM32Bytes = gf_huffman_decode_lds_m32(nRows, nCols);
    a.rawM32 = rawM32;
    a.rawSide = rawSide;
    a.rawSideStride = rawSideStride;
    a.sideStatus = sideStatus;
    a.produced2 = produced2;
    a.inflStatus2 = inflStatus2;
    GF_HIP(gf_launch_lsop_unpack_m32(a, st, grid));
    return GF_OK;
}


constexpr size_t LSOP_INFLATE_SCRATCH_BYTES = (size_t)384 << 20;   // LSOP12's Deflate containers (rare: see lsopUnpackM32Deflate)
constexpr size_t INFLATE_SCRATCH_BYTES = (size_t)1 << 30;     // thousands of streams per launch: a stream is one serial chain

// Second entropy pass of the LSOP12 decode with the Deflate containers inflated ON THE DEVICE (LsDecoder12.java:127-141):
// per chunk of tiles, k_lsop_streams describes the first zlib stream of every Deflate container, k_inflate runs, k_lsop_streams
// places the second stream behind what the first one consumed, k_inflate runs again, and k_lsop_unpack_m32 reads the M32 bytes
// of those tiles from the scratch (legacy Huffman containers are decoded as stored in the same launch).
static gf_status lsopUnpackM32Deflate(gf_context *c, hipStream_t st, int nRows, int nCols, size_t nTiles, const uint8_t *dBlob,
                                      size_t blobBytes, const uint64_t *dOffsets, size_t slotStride, const uint32_t *dLengths,
                                      int32_t *dResiduals, size_t resStride, uint32_t *dCoefs, int32_t *dScratchStatus)
{
    const size_t nInit = (size_t)4 * nRows + 2 * nCols - 9, nInt = (size_t)(nRows - 2) * (size_t)(nCols - 4);
    const size_t rawStride = roundUp(6 * (nInit + nInt) + 192, 16);
    // The scratch of the Deflate containers: LSOP_INFLATE_SCRATCH_BYTES, not a share of HBM per tile of the batch -- the default
    // encoder output is the canonical container and pays these gated launches for nothing (the full gigabyte of the CodecDeflate /
    // CodecFloat paths made a gigabyte of every context -- read-ahead, gf_multi shards -- that ever decoded an LSOP tile)
    const size_t chunk = std::max<size_t>(1, std::min(nTiles, LSOP_INFLATE_SCRATCH_BYTES / rawStride));
    gf_status s;
    if ((s = c->dInflOut.ensure(chunk * rawStride + 64)) != GF_OK) return s;
    if ((s = c->dInflate.ensure(chunk * sizeof(GfInflateStream))) != GF_OK) return s;
    if ((s = c->dInflMeta.ensure(chunk * 7 * 4 + 128)) != GF_OK) return s;
    uint8_t *raw = (uint8_t *)c->dInflOut.p;
    GfInflateStream *desc = (GfInflateStream *)c->dInflate.p;
    uint32_t *produced1 = (uint32_t *)c->dInflMeta.p, *consumed1 = produced1 + chunk, *produced2 = consumed1 + chunk;
    int32_t *status1 = (int32_t *)(produced2 + chunk), *status2 = status1 + chunk, *side = status2 + chunk;
    uint32_t *gate = (uint32_t *)(side + chunk);                   // number of Deflate containers in the chunk
    for (size_t t0 = 0; t0 < nTiles; t0 += chunk) {
        const size_t n = std::min(chunk, nTiles - t0);
        GF_HIP(hipMemsetAsync(gate, 0, 4, st));
        // this chunk's view of the batch
        const uint8_t *blobC = dOffsets ? dBlob : dBlob + t0 * slotStride;
        const size_t blobBytesC = dOffsets ? blobBytes : blobBytes - t0 * slotStride;
        const uint64_t *offC = dOffsets ? dOffsets + t0 : nullptr;
        for (int pass = 0; pass < 2; pass++) {
            GF_HIP(gf_launch_lsop_streams(blobC, blobBytesC, offC, slotStride, dLengths + t0, n, (uint32_t)nInit, (uint32_t)nInt, rawStride,
                                          pass, produced1, status1, consumed1, desc, side, gate, st));
            GfInflateArgs a{};
            a.inBase = blobC;
            a.outBase = raw;
            a.streams = desc;
            a.produced = pass ? produced2 : produced1;
            a.status = pass ? status2 : status1;
            a.consumed = pass ? nullptr : consumed1;
            a.gate = gate;
            a.nStreams = n;
            a.window = gf_inflate_window(0);
            GF_HIP(gf_launch_inflate(a, st));
        }
        s = lsopUnpackM32(c, st, nRows, nCols, n, blobC, blobBytesC, offC, slotStride, dLengths + t0, dResiduals + t0 * resStride, resStride,
                          dCoefs + t0 * 16, dScratchStatus + t0, 2, raw, rawStride, side, produced2, status2);
        if (s != GF_OK) return s;
    }
    return GF_OK;
}

extern "C" {

gf_status gf_huffman_encode_batch_i32_dev(gf_context *c, void *stream, int codecIndex, int nRows, int nCols,
                                          size_t nTiles, const int32_t *dValues, uint8_t *dOut, size_t slotStride,
                                          uint32_t *dLengths, uint8_t *dPredictors, int32_t *dStatus,
                                          int predictorMask)
{
    GF_CTX_LOCK(c);
    return encodeBatchDev(KIND_HUFFMAN, c, stream, codecIndex, nRows, nCols, nTiles, dValues, dOut, slotStride, dLengths,
                          dPredictors, dStatus, predictorMask);
}

gf_status gf_huffman_decode_batch_i32_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles,
                                          const uint8_t *dBlob, size_t blobBytes, const uint64_t *dOffsets,
                                          size_t slotStride, const uint32_t *dLengths, int32_t *dValues,
                                          int32_t *dStatus)
{
    GF_CTX_LOCK(c);
    return decodeBatchDev(KIND_HUFFMAN, c, stream, nRows, nCols, nTiles, dBlob, blobBytes, dOffsets, slotStride, dLengths,
                          dValues, dStatus);
}

gf_status gf_canon_encode_batch_i32_dev(gf_context *c, void *stream, int codecIndex, int nRows, int nCols,
                                        size_t nTiles, const int32_t *dValues, uint8_t *dOut, size_t slotStride,
                                        uint32_t *dLengths, uint8_t *dPredictors, int32_t *dStatus, int predictorMask)
{
    GF_CTX_LOCK(c);
    return encodeBatchDev(KIND_CANON, c, stream, codecIndex, nRows, nCols, nTiles, dValues, dOut, slotStride, dLengths,
                          dPredictors, dStatus, predictorMask);
}

gf_status gf_canon_decode_batch_i32_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles,
                                        const uint8_t *dBlob, size_t blobBytes, const uint64_t *dOffsets,
                                        size_t slotStride, const uint32_t *dLengths, int32_t *dValues, int32_t *dStatus)
{
    GF_CTX_LOCK(c);
    return decodeBatchDev(KIND_CANON, c, stream, nRows, nCols, nTiles, dBlob, blobBytes, dOffsets, slotStride, dLengths,
                          dValues, dStatus);
}

size_t gf_canon_max_packing(int nRows, int nCols)
{
    // 6 header bytes + code tables (< 750 bytes) + per value at most 4 symbols of 15 bits and 24 raw bits + end-of-text
    const size_t cells = (size_t)nRows * (size_t)nCols;
    return roundUp(6 + 768 + (cells * 84 + 15 + 7) / 8 + 16, 16);
}

gf_status gf_compact_dev(gf_context *c, void *stream, size_t nTiles, const uint8_t *dSlots, size_t slotStride,
                         const uint32_t *dLengths, uint64_t *dOffsets, uint8_t *dBlob, size_t blobCap)
{
    GF_CTX_LOCK(c);
    if (!c || !dSlots || !dLengths || !dOffsets || !dBlob) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    if (((uintptr_t)dSlots & 15) != 0 || slotStride % 16 != 0) return GF_ERR_ARG;
    GF_HIP(gf_launch_compact(nTiles, dSlots, slotStride, dLengths, dOffsets, dBlob, blobCap,
                             stream ? (hipStream_t)stream : c->stream));
    return GF_OK;
}

gf_status gf_synth_dem_dev(gf_context *c, void *stream, uint64_t seed, int nRows, int nCols, int64_t tilesPerRow,
                           int64_t tile0, size_t nTiles, int32_t *dValues)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || tilesPerRow < 1 || !dValues) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    GF_HIP(gf_launch_synth_dem(seed, nRows, nCols, tilesPerRow, tile0, nTiles, dValues,
                               stream ? (hipStream_t)stream : c->stream));
    return GF_OK;
}

gf_status gf_synth_dem_masked_dev(gf_context *c, void *stream, uint64_t seed, int nRows, int nCols, int64_t tilesPerRow,
                                  int64_t tile0, size_t nTiles, int maskPerMille, int32_t *dValues)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || tilesPerRow < 1 || !dValues || maskPerMille < 0 || maskPerMille > 1000) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    GF_HIP(gf_launch_synth_dem(seed, nRows, nCols, tilesPerRow, tile0, nTiles, dValues,
                               stream ? (hipStream_t)stream : c->stream, maskPerMille));
    return GF_OK;
}

gf_status gf_synth_dem_style_dev(gf_context *c, void *stream, uint64_t seed, int nRows, int nCols, int64_t tilesPerRow,
                                 int64_t tile0, size_t nTiles, int maskPerMille, int style, int32_t *dValues)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || tilesPerRow < 1 || !dValues || maskPerMille < 0 || maskPerMille > 1000) return GF_ERR_ARG;
    if (style != GF_DEM_STYLE_CLASSIC && style != GF_DEM_STYLE_ROUGH) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    GF_HIP(gf_launch_synth_dem(seed, nRows, nCols, tilesPerRow, tile0, nTiles, dValues,
                               stream ? (hipStream_t)stream : c->stream, maskPerMille, style));
    return GF_OK;
}

// ------------------------------------------------------------------ CodecFloat

size_t gf_float_planes_bytes(int nRows, int nCols)
{
    const size_t n = (size_t)nRows * (size_t)nCols;
    return (n + 7) / 8 + 4 * n;
}

gf_status gf_float_planes_encode_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles, const float *dValues,
                                     uint8_t *dPlanes, size_t planeStride)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || !dValues || !dPlanes || planeStride < gf_float_planes_bytes(nRows, nCols)) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    GF_HIP(gf_launch_float_planes_encode((const uint32_t *)dValues, dPlanes, planeStride, nTiles, nRows, nCols,
                                         stream ? (hipStream_t)stream : c->stream));
    return GF_OK;
}

gf_status gf_float_planes_decode_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles, const uint8_t *dPlanes,
                                     size_t planeStride, float *dValues)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || !dValues || !dPlanes || planeStride < gf_float_planes_bytes(nRows, nCols)) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    GF_HIP(gf_launch_float_planes_decode(dPlanes, (uint32_t *)dValues, planeStride, nTiles, nRows, nCols,
                                         stream ? (hipStream_t)stream : c->stream));
    return GF_OK;
}

// java.util.zip.Deflater(level): setInput, finish, deflate(.., FULL_FLUSH) == one complete zlib stream
static bool zDeflate(const uint8_t *in, size_t n, int level, std::vector<uint8_t> &out)
{
    uLongf cap = compressBound((uLong)n) + 64;
    out.resize(cap);
    if (compress2(out.data(), &cap, in, (uLong)n, level) != Z_OK) return false;
    out.resize(cap);
    return true;
}

// The same stream, given up as soon as it is longer than `limit` bytes (returns false then, as on a zlib error): the caller only
// wants it if it is no longer than that.  Feeding the whole input with Z_FINISH and draining the output in pieces gives the bytes
// of compress2 -- what comes out of deflate() does not depend on how much room each call is given.
static bool zDeflateUpTo(const uint8_t *in, size_t n, int level, size_t limit, std::vector<uint8_t> &out)
{
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit(&zs, level) != Z_OK) return false;
    const size_t bound = (size_t)compressBound((uLong)n) + 64;
    out.resize(bound);
    zs.next_in = const_cast<Bytef *>(in);
    zs.avail_in = (uInt)n;
    size_t done = 0;
    int rc = Z_OK;
    while (rc == Z_OK) {
        const size_t room = std::min<size_t>(bound - done, 2048);
        zs.next_out = out.data() + done;
        zs.avail_out = (uInt)room;
        rc = deflate(&zs, Z_FINISH);
        done += room - zs.avail_out;
        if (done > limit && rc != Z_STREAM_END) { deflateEnd(&zs); return false; }
        if (room == 0) break;
    }
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || done > limit) return false;
    out.resize(done);
    return true;
}

gf_status gf_float_encode_batch_f32(gf_context *c, int codecIndex, int nRows, int nCols, size_t nTiles, const float *values,
                                    int zlibLevel, uint8_t *blob, size_t blobCap, uint64_t *offsets)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || !values || !offsets || (!blob && blobCap)) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    const size_t n = (size_t)nRows * (size_t)nCols, nSign = (n + 7) / 8;
    const size_t stride = roundUp(gf_float_planes_bytes(nRows, nCols), 16);
    gf_status s;
    if ((s = c->dValues.ensure(nTiles * n * 4 + 16)) != GF_OK) return s;
    if ((s = c->dPlanes.ensure(nTiles * stride + 16)) != GF_OK) return s;
    GF_HIP(hipMemcpyAsync(c->dValues.p, values, nTiles * n * 4, hipMemcpyHostToDevice, c->stream));
    s = gf_float_planes_encode_dev(c, c->stream, nRows, nCols, nTiles, (const float *)c->dValues.p, (uint8_t *)c->dPlanes.p, stride);
    if (s != GF_OK) return s;
    std::vector<uint8_t> planes(nTiles * stride);
    GF_HIP(hipMemcpyAsync(planes.data(), c->dPlanes.p, nTiles * stride, hipMemcpyDeviceToHost, c->stream));
    GF_HIP(hipStreamSynchronize(c->stream));
    // framing, CodecFloat.java:371-391: codecIndex, 0, then five [int32 LE length, zlib stream]
    std::vector<std::vector<uint8_t>> packed(nTiles);
    std::vector<uint8_t> failed(nTiles, 0);
    parallelFor(nTiles, [&](size_t t) {
        const uint8_t *p = planes.data() + t * stride;
        std::vector<uint8_t> &out = packed[t];
        std::vector<uint8_t> z;
        out.push_back((uint8_t)codecIndex);
        out.push_back(0);
        size_t planeOff = 0;
        for (int k = 0; k < 5; k++) {
            const size_t pl = k == 0 ? nSign : n;
            if (!zDeflate(p + planeOff, pl, zlibLevel, z)) { failed[t] = 1; return; }
            planeOff += pl;
            const uint32_t zn = (uint32_t)z.size();
            for (int b = 0; b < 4; b++) out.push_back((uint8_t)(zn >> (8 * b)));
            out.insert(out.end(), z.begin(), z.end());
        }
    });
    uint64_t total = 0;
    for (size_t t = 0; t < nTiles; t++) {
        if (failed[t]) return GF_ERR_ARG;                     // zlib rejected the level
        offsets[t] = total;
        total += packed[t].size();
    }
    offsets[nTiles] = total;
    if (total > blobCap) return GF_ERR_CAPACITY;
    parallelFor(nTiles, [&](size_t t) { memcpy(blob + offsets[t], packed[t].data(), packed[t].size()); });
    return GF_OK;
}

static gf_status decodeBatchHost(int kind, gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                 const uint64_t *offsets, int32_t *values, int32_t *status);

// CodecFloat.decodeFloats :395-458 for a batch: the five zlib streams of every packing are inflated ON THE DEVICE
// (gvrs_inflate.hip), the planes merged there; the host only moves bytes (chunked, pinned staging).  Without a status array
// the first failing tile's status is the return value.
gf_status gf_float_decode_batch_f32(gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                    const uint64_t *offsets, float *values, int32_t *status)
{
    GF_CTX_LOCK(c);
    if (status) return decodeBatchHost(KIND_FLOAT, c, nRows, nCols, nTiles, blob, offsets, (int32_t *)values, status);
    std::vector<int32_t> st(nTiles, GF_OK);
    const gf_status s = decodeBatchHost(KIND_FLOAT, c, nRows, nCols, nTiles, blob, offsets, (int32_t *)values, st.data());
    if (s != GF_OK) return s;
    for (size_t t = 0; t < nTiles; t++)
        if (st[t] != GF_OK) return (gf_status)st[t];
    return GF_OK;
}

gf_status gf_float_decode_batch_f32_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles, const uint8_t *dBlob,
                                        size_t blobBytes, const uint64_t *dOffsets, const uint32_t *dLengths, float *dValues,
                                        int32_t *dStatus)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || !dBlob || !dOffsets || !dLengths || !dValues || !dStatus) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    return floatDecodeDev(c, stream ? (hipStream_t)stream : c->stream, nRows, nCols, nTiles, dBlob, blobBytes, dOffsets, dLengths, dValues, dStatus);
}

gf_status gf_float_encode_f32(gf_context *c, int codecIndex, int nRows, int nCols, const float *values, int zlibLevel,
                              uint8_t *out, size_t outCap, size_t *outLen)
{
    GF_CTX_LOCK(c);
    if (!outLen) return GF_ERR_ARG;
    uint64_t offsets[2] = {0, 0};
    const gf_status s = gf_float_encode_batch_f32(c, codecIndex, nRows, nCols, 1, values, zlibLevel, out, outCap, offsets);
    *outLen = (size_t)offsets[1];
    return s;
}

gf_status gf_float_decode_f32(gf_context *c, int nRows, int nCols, const uint8_t *packing, size_t len, float *values)
{
    GF_CTX_LOCK(c);
    uint64_t offsets[2] = {0, (uint64_t)len};
    int32_t st = 0;
    const gf_status s = gf_float_decode_batch_f32(c, nRows, nCols, 1, packing, offsets, values, &st);
    if (s != GF_OK) return s;
    return (gf_status)st;
}

// ------------------------------------------------------------------ device memory helpers

// zlib streams inflated on the device (gvrs_inflate.hip): stream i = d_in[in_offsets[i] .. + in_lengths[i]) -> at most out_caps[i]
// bytes at d_out + out_offsets[i].  The four descriptor arrays are HOST arrays (they are packed and uploaded here);
// d_produced / d_status are device arrays.  Enqueues only (after the small descriptor upload on the same stream).
gf_status gf_inflate_batch_dev(gf_context *c, void *stream, size_t nStreams, const uint8_t *dIn, const uint64_t *inOffsets,
                               const uint32_t *inLengths, uint8_t *dOut, const uint64_t *outOffsets, const uint32_t *outCaps,
                               uint32_t *dProduced, int32_t *dStatus)
{
    GF_CTX_LOCK(c);
    if (!c || (nStreams && (!dIn || !inOffsets || !inLengths || !dOut || !outOffsets || !outCaps || !dProduced || !dStatus)))
        return GF_ERR_ARG;
    if (nStreams == 0) return GF_OK;
    GF_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    gf_status s = c->dInflate.ensure(nStreams * sizeof(GfInflateStream) + 16);
    if (s != GF_OK) return s;
    std::vector<GfInflateStream> desc(nStreams);
    uint32_t maxCap = 0;
    for (size_t i = 0; i < nStreams; i++) {
        desc[i].inOffset = inOffsets[i];
        desc[i].outOffset = outOffsets[i];
        desc[i].inLen = inLengths[i];
        desc[i].outCap = outCaps[i];
        maxCap = std::max(maxCap, outCaps[i]);
    }
    // (pageable source: the copy is staged by the runtime before the call returns)
    GF_HIP(hipMemcpyAsync(c->dInflate.p, desc.data(), nStreams * sizeof(GfInflateStream), hipMemcpyHostToDevice, st));
    GfInflateArgs a{};
    a.inBase = dIn;
    a.outBase = dOut;
    a.streams = (const GfInflateStream *)c->dInflate.p;
    a.produced = dProduced;
    a.status = dStatus;
    a.nStreams = nStreams;
    a.window = gf_inflate_window(maxCap);
    GF_HIP(gf_launch_inflate(a, st));
    return GF_OK;
}

// page-locked host memory: the host-memory batch entry points move it over PCIe in place (no staging copy)
gf_status gf_host_alloc(size_t bytes, void **p)
{
    if (!p) return GF_ERR_ARG;
    *p = nullptr;
    GF_HIP(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocPortable));
    return GF_OK;
}

gf_status gf_host_free(void *p)
{
    if (p) GF_HIP(hipHostFree(p));
    return GF_OK;
}

gf_status gf_dev_malloc(gf_context *c, size_t bytes, void **p)
{
    GF_CTX_LOCK(c);
    if (!c || !p) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    GF_HIP(hipMalloc(p, bytes ? bytes : 16));
    return GF_OK;
}

gf_status gf_dev_free(gf_context *c, void *p)
{
    GF_CTX_LOCK(c);
    if (!c) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    GF_HIP(hipFree(p));
    return GF_OK;
}

gf_status gf_dev_memset(gf_context *c, void *p, int value, size_t bytes)
{
    GF_CTX_LOCK(c);
    if (!c) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    GF_HIP(hipMemsetAsync(p, value, bytes, c->stream));
    GF_HIP(hipStreamSynchronize(c->stream));
    return GF_OK;
}

gf_status gf_dev_upload(gf_context *c, void *d, const void *h, size_t bytes)
{
    GF_CTX_LOCK(c);
    if (!c) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    GF_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    GF_HIP(hipStreamSynchronize(c->stream));
    return GF_OK;
}

gf_status gf_dev_download(gf_context *c, void *h, const void *d, size_t bytes)
{
    GF_CTX_LOCK(c);
    if (!c) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    GF_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    GF_HIP(hipStreamSynchronize(c->stream));
    return GF_OK;
}

// ------------------------------------------------------------------ timers

gf_status gf_timer_create(gf_context *c, gf_timer **out)
{
    GF_CTX_LOCK(c);
    if (!c || !out) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    gf_timer *t = new (std::nothrow) gf_timer();
    if (!t) return GF_ERR_ARG;
    t->ctx = c;
    GF_HIP(hipEventCreate(&t->start));
    GF_HIP(hipEventCreate(&t->stop));
    *out = t;
    return GF_OK;
}

void gf_timer_destroy(gf_timer *t)
{
    if (!t) return;
    (void)hipEventDestroy(t->start);
    (void)hipEventDestroy(t->stop);
    delete t;
}

gf_status gf_timer_start(gf_timer *t, void *stream)
{
    if (!t) return GF_ERR_ARG;
    GF_HIP(hipEventRecord(t->start, stream ? (hipStream_t)stream : t->ctx->stream));
    return GF_OK;
}

gf_status gf_timer_stop(gf_timer *t, void *stream)
{
    if (!t) return GF_ERR_ARG;
    GF_HIP(hipEventRecord(t->stop, stream ? (hipStream_t)stream : t->ctx->stream));
    return GF_OK;
}

gf_status gf_timer_elapsed_ms(gf_timer *t, float *ms)
{
    if (!t || !ms) return GF_ERR_ARG;
    GF_HIP(hipEventSynchronize(t->stop));
    GF_HIP(hipEventElapsedTime(ms, t->start, t->stop));
    return GF_OK;
}

// ------------------------------------------------------------------ host-memory entry points

}  // extern "C"

// ---- Deflate-carrying containers decoded on the device: walk the packings, inflate (gvrs_inflate.hip), decode ----------
// The scratch (inflated bytes, stream descriptors, per-stream results) is bounded: batches go through it in chunks of tiles,
// one after the other in stream order.

static gf_status deflateDecodeDev(gf_context *c, hipStream_t st, int nRows, int nCols, size_t nTiles, const uint8_t *dBlob,
                                  size_t blobBytes, const uint64_t *dOffsets, size_t slotStride, const uint32_t *dLengths,
                                  int32_t *dValues, int32_t *dStatus)
{
    const size_t cells = (size_t)nRows * (size_t)nCols;
    if (cells >= (1ull << 28)) return GF_ERR_UNSUPPORTED;
    const size_t rawStride = roundUp(10 + 6 * cells, 16);                 // an M32 stream has at most six bytes per cell
    const size_t chunk = std::max<size_t>(1, std::min(nTiles, INFLATE_SCRATCH_BYTES / rawStride));
    gf_status s;
    if ((s = c->dInflOut.ensure(chunk * rawStride + 64)) != GF_OK) return s;
    if ((s = c->dInflate.ensure(chunk * sizeof(GfInflateStream) + 64)) != GF_OK) return s;
    if ((s = c->dInflMeta.ensure(chunk * 20 + 256)) != GF_OK) return s;
    uint8_t *raw = (uint8_t *)c->dInflOut.p;
    GfInflateStream *desc = (GfInflateStream *)c->dInflate.p;
    uint32_t *produced = (uint32_t *)c->dInflMeta.p, *rawLengths = produced + chunk;
    int32_t *inflStatus = (int32_t *)(rawLengths + chunk), *pre = inflStatus + chunk, *decStatus = pre + chunk;
    for (size_t t0 = 0; t0 < nTiles; t0 += chunk) {
        const size_t n = std::min(chunk, nTiles - t0);
        GF_HIP(gf_launch_deflate_streams(dBlob, blobBytes, dOffsets, slotStride, dLengths, t0, n, (uint32_t)cells, raw, rawStride, desc, pre, st));
        GfInflateArgs a{};
        a.inBase = dBlob;
        a.outBase = raw;
        a.streams = desc;
        a.produced = produced;
        a.status = inflStatus;
        a.nStreams = n;
        a.window = gf_inflate_window((uint32_t)std::min<size_t>(6 * cells, 32768));
        GF_HIP(gf_launch_inflate(a, st));
        GF_HIP(gf_launch_deflate_lengths(n, desc, produced, inflStatus, pre, rawLengths, raw, st));
        s = decodeBatchDev(KIND_RAW_M32, c, st, nRows, nCols, n, raw, chunk * rawStride + 32, nullptr, rawStride, rawLengths,
                           dValues + t0 * cells, decStatus);
        if (s != GF_OK) return s;
        GF_HIP(gf_launch_merge_status(n, pre, decStatus, dStatus + t0, st));
    }
    return GF_OK;
}

static gf_status floatDecodeDev(gf_context *c, hipStream_t st, int nRows, int nCols, size_t nTiles, const uint8_t *dBlob, size_t blobBytes,
                                const uint64_t *dOffsets, const uint32_t *dLengths, float *dValues, int32_t *dStatus)
{
    const size_t cells = (size_t)nRows * (size_t)nCols;
    if (cells >= (1ull << 28)) return GF_ERR_UNSUPPORTED;
    if (!dOffsets) return GF_ERR_ARG;
    const size_t planeStride = roundUp(gf_float_planes_bytes(nRows, nCols), 16);
    const size_t chunk = std::max<size_t>(1, std::min(nTiles, INFLATE_SCRATCH_BYTES / planeStride));
    gf_status s;
    if ((s = c->dInflOut.ensure(chunk * planeStride + 64)) != GF_OK) return s;
    if ((s = c->dInflate.ensure(chunk * 5 * sizeof(GfInflateStream) + 64)) != GF_OK) return s;
    if ((s = c->dInflMeta.ensure(chunk * (5 * 8 + 4) + 256)) != GF_OK) return s;
    uint8_t *planes = (uint8_t *)c->dInflOut.p;
    GfInflateStream *desc = (GfInflateStream *)c->dInflate.p;
    uint32_t *produced = (uint32_t *)c->dInflMeta.p;
    int32_t *inflStatus = (int32_t *)(produced + 5 * chunk), *pre = inflStatus + 5 * chunk;
    for (size_t t0 = 0; t0 < nTiles; t0 += chunk) {
        const size_t n = std::min(chunk, nTiles - t0);
        GF_HIP(hipMemsetAsync(planes, 0, n * planeStride, st));           // what a short stream does not reach reads as zero
        GF_HIP(gf_launch_float_streams(dBlob, blobBytes, dOffsets, dLengths, t0, n, (uint32_t)cells, planeStride, desc, pre, st));
        GfInflateArgs a{};
        a.inBase = dBlob;
        a.outBase = planes;
        a.streams = desc;
        a.produced = produced;
        a.status = inflStatus;
        a.nStreams = 5 * n;
        a.window = gf_inflate_window((uint32_t)std::min<size_t>(cells, 32768));
        GF_HIP(gf_launch_inflate(a, st));
        GF_HIP(gf_launch_float_short_planes(n, pre, inflStatus, produced, planes, planeStride, nRows, nCols, st));
        GF_HIP(gf_launch_float_status(n, pre, inflStatus, dStatus + t0, st));
        GF_HIP(gf_launch_float_planes_decode(planes, (uint32_t *)dValues + t0 * cells, planeStride, n, nRows, nCols, st));
    }
    return GF_OK;
}

// ---- pipelined staging of the host-memory batch entry points ---------------------------------------------------
// A batch in host memory is cut into chunks of about HOST_CHUNK_BYTES of cell values.  Each chunk travels through one of
// HOST_SLOTS slots (pinned staging buffers, device buffers, a stream of its own): the calling thread copies the caller's
// (pageable) memory into the slot's pinned buffer with a few helper threads, enqueues H2D copy + kernels + D2H copy on the
// slot's stream and moves on to the next chunk, so that the copy-in of chunk k+1, the device work of chunk k and the
// copy-out of chunk k-1 overlap.  Device and pinned memory are bounded by the chunk, not by the batch.  Memory the caller
// obtained from gf_host_alloc (or pinned itself) is used in place, without the staging copy.
constexpr int HOST_SLOTS = 3;
constexpr size_t HOST_CHUNK_BYTES = (size_t)64 << 20;

struct PinBuf {
    void *p = nullptr;
    size_t bytes = 0;
    gf_status ensure(size_t need)
    {
        if (need <= bytes) return GF_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        need = roundUp(need + need / 8, 1 << 20);
        GF_HIP(hipHostMalloc(&p, need, hipHostMallocDefault));
        bytes = need;
        return GF_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct HostSlot {
    hipStream_t stream = nullptr;
    hipEvent_t evA = nullptr, evB = nullptr;       // device work enqueued so far / final copy-out done
    hipEvent_t evK = nullptr;                      // the chunk's codec kernels are done (they use the context's per-tile records)
    DevBuf dValues, dSlots, dBlob, dLengths, dPred, dStatus, dOffsets;
    PinBuf hIn, hOut, hMeta;
    void release()
    {
        dValues.release(); dSlots.release(); dBlob.release(); dLengths.release(); dPred.release(); dStatus.release();
        dOffsets.release(); hIn.release(); hOut.release(); hMeta.release();
        if (evA) (void)hipEventDestroy(evA);
        if (evB) (void)hipEventDestroy(evB);
        if (evK) (void)hipEventDestroy(evK);
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr; evA = evB = evK = nullptr;
    }
};

struct gf_host_pipe {
    HostSlot slot[HOST_SLOTS];
};

static gf_status hostPipe(gf_context *c, gf_host_pipe **out)
{
    if (!c->pipe) {
        gf_host_pipe *p = new (std::nothrow) gf_host_pipe();
        if (!p) return GF_ERR_ARG;
        for (int i = 0; i < HOST_SLOTS; i++) {
            hipError_t e = hipStreamCreateWithFlags(&p->slot[i].stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p->slot[i].evA, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p->slot[i].evB, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p->slot[i].evK, hipEventDisableTiming);
            if (e != hipSuccess) {
                for (int j = 0; j <= i; j++) p->slot[j].release();
                delete p;
                return hipFail(e, "host pipeline set-up");
            }
        }
        c->pipe = p;
    }
    *out = c->pipe;
    return GF_OK;
}

void gf_host_pipe_destroy(gf_host_pipe *p)
{
    if (!p) return;
    for (int i = 0; i < HOST_SLOTS; i++) p->slot[i].release();
    delete p;
}

// is this host pointer page-locked (hipHostMalloc / hipHostRegister)?  Then the DMA engines read and write it directly.
static bool isPinned(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();                   // pageable memory is reported as an error; clear it
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// memcpy with a few helper threads: one thread moves 6-10 GB/s, PCIe Gen5 x16 five times that
static void parallelCopy(void *dst, const void *src, size_t bytes)
{
    const size_t per = (size_t)8 << 20;
    unsigned nt = (unsigned)std::min<size_t>(8, bytes / per);
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && nt > hw) nt = hw;
    if (nt <= 1) { memcpy(dst, src, bytes); return; }
    const size_t part = roundUp((bytes + nt - 1) / nt, 4096);
    std::vector<std::thread> th;
    for (unsigned w = 1; w < nt; w++) {
        const size_t o = (size_t)w * part;
        if (o >= bytes) break;
        th.emplace_back([=]() { memcpy((uint8_t *)dst + o, (const uint8_t *)src + o, std::min(part, bytes - o)); });
    }
    memcpy(dst, src, std::min(part, bytes));
    for (auto &x : th) x.join();
}

static size_t hostChunkTiles(size_t cells, size_t nTiles, int kind = KIND_HUFFMAN)
{
    // the inflate kernels run one wave per zlib stream and a stream is a serial chain: a chunk has to bring thousands of streams
    const size_t bytes = kind == KIND_DEFLATE || kind == KIND_FLOAT ? 4 * HOST_CHUNK_BYTES : HOST_CHUNK_BYTES;
    const size_t n = std::max<size_t>(1, bytes / (cells * 4));
    return std::min(n, std::max<size_t>(nTiles, 1));
}

static gf_status encodeBatchHost(int kind, gf_context *c, int codecIndex, int nRows, int nCols, size_t nTiles,
                                 const int32_t *values, uint8_t *blob, size_t blobCap, uint64_t *offsets,
                                 uint8_t *predictors, int32_t *status)
{
    if (!c || nRows < 1 || nCols < 1 || (!values && nTiles) || !offsets || (!blob && blobCap)) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    gf_host_pipe *P;
    gf_status s = hostPipe(c, &P);
    if (s != GF_OK) return s;
    const size_t cells = (size_t)nRows * (size_t)nCols;
    const size_t stride = gf_huffman_default_stride(nRows, nCols);
    const size_t chunk = hostChunkTiles(cells, nTiles);
    const size_t nChunks = (nTiles + chunk - 1) / chunk;
    const bool pinnedIn = nTiles && isPinned(values), pinnedOut = blob && isPinned(blob);
    // the per-tile records between the encoder kernels are per context: one launch at a time uses them, so every chunk's
    // kernels run in the context's stream order (the slot streams carry the copies and wait for / signal the kernels)
    if ((s = gf_context_reserve(c, nRows, nCols, chunk)) != GF_OK) return s;

    struct Meta { uint32_t *len; int32_t *st; uint8_t *pred; uint64_t *off; };
    auto metaOf = [&](HostSlot &S) {
        Meta m;
        uint8_t *b = (uint8_t *)S.hMeta.p;
        m.off = (uint64_t *)b;
        m.len = (uint32_t *)(b + roundUp((chunk + 1) * 8, 64));
        m.st = (int32_t *)((uint8_t *)m.len + roundUp(chunk * 4, 64));
        m.pred = (uint8_t *)m.st + roundUp(chunk * 4, 64);
        return m;
    };
    for (int i = 0; i < HOST_SLOTS && (size_t)i < nChunks; i++) {
        HostSlot &S = P->slot[i];
        if ((s = S.dValues.ensure(chunk * cells * 4 + 16)) != GF_OK) return s;
        if ((s = S.dSlots.ensure(chunk * stride + 16)) != GF_OK) return s;
        if ((s = S.dBlob.ensure(chunk * stride + 16)) != GF_OK) return s;
        if ((s = S.dLengths.ensure(chunk * 4 + 16)) != GF_OK) return s;
        if ((s = S.dPred.ensure(chunk + 16)) != GF_OK) return s;
        if ((s = S.dStatus.ensure(chunk * 4 + 16)) != GF_OK) return s;
        if ((s = S.dOffsets.ensure((chunk + 1) * 8 + 16)) != GF_OK) return s;
        if (!pinnedIn && (s = S.hIn.ensure(chunk * cells * 4)) != GF_OK) return s;
        if ((s = S.hOut.ensure(chunk * stride)) != GF_OK) return s;
        if ((s = S.hMeta.ensure(roundUp((chunk + 1) * 8, 64) + 2 * roundUp(chunk * 4, 64) + roundUp(chunk, 64) + 64)) != GF_OK) return s;
    }

    uint64_t total = 0;                            // bytes of the packings placed so far
    bool overCap = false;
    offsets[0] = 0;
    // stage B of a chunk: its kernels are done -> overflow tiles, then the compact blob comes home
    auto stageB = [&](size_t k) -> gf_status {
        HostSlot &S = P->slot[k % HOST_SLOTS];
        const size_t t0 = k * chunk, n = std::min(chunk, nTiles - t0);
        const Meta m = metaOf(S);
        GF_HIP(hipEventSynchronize(S.evA));
        const uint64_t bytes = m.off[n];
        if (bytes) GF_HIP(hipMemcpyAsync(S.hOut.p, S.dBlob.p, bytes, hipMemcpyDeviceToHost, S.stream));
        GF_HIP(hipEventRecord(S.evB, S.stream));
        return GF_OK;
    };
    // stage C: the blob of the chunk is in pinned memory -> the caller's arrays
    auto stageC = [&](size_t k) -> gf_status {
        HostSlot &S = P->slot[k % HOST_SLOTS];
        const size_t t0 = k * chunk, n = std::min(chunk, nTiles - t0);
        const Meta m = metaOf(S);
        GF_HIP(hipEventSynchronize(S.evB));
        bool anyBig = false;
        for (size_t t = 0; t < n; t++) anyBig = anyBig || m.st[t] == GF_OVERFLOW;
        if (!anyBig) {
            const uint64_t bytes = m.off[n];
            if (total + bytes <= blobCap) {
                if (bytes) parallelCopy(blob + total, S.hOut.p, bytes);
            } else {
                overCap = true;
            }
            for (size_t t = 0; t < n; t++) offsets[t0 + t + 1] = total + m.off[t + 1];
            total += bytes;
        } else {
            // tiles whose packing did not fit the default slot (longer than the raw tile): redone one by one into a
            // worst-case slot so that the bytes are still exactly the reference's
            const size_t maxp = kind == KIND_CANON ? gf_canon_max_packing(nRows, nCols) : gf_huffman_max_packing(nRows, nCols);
            DevBuf slot, meta;
            gf_status r;
            for (int i = 0; i < HOST_SLOTS; i++) GF_HIP(hipStreamSynchronize(P->slot[i].stream));   // nothing else uses the records now
            if ((r = slot.ensure(maxp)) != GF_OK) return r;
            if ((r = meta.ensure(64)) != GF_OK) { slot.release(); return r; }
            std::vector<uint8_t> big;
            for (size_t t = 0; t < n; t++) {
                uint64_t len = (m.st[t] == GF_OK) ? m.len[t] : 0;
                const uint8_t *src = (const uint8_t *)S.hOut.p + m.off[t];
                if (m.st[t] == GF_OVERFLOW) {
                    uint32_t *dLen = (uint32_t *)meta.p;
                    int32_t *dSt = (int32_t *)((uint8_t *)meta.p + 16);
                    r = encodeBatchDev(kind, c, S.stream, codecIndex, nRows, nCols, 1, (const int32_t *)S.dValues.p + t * cells,
                                       (uint8_t *)slot.p, maxp, dLen, nullptr, dSt, GF_PM_ALL);
                    uint32_t l = 0;
                    int32_t tst = 0;
                    hipError_t e = hipSuccess;
                    if (r == GF_OK) e = hipMemcpyAsync(&l, dLen, 4, hipMemcpyDeviceToHost, S.stream);
                    if (r == GF_OK && e == hipSuccess) e = hipMemcpyAsync(&tst, dSt, 4, hipMemcpyDeviceToHost, S.stream);
                    if (r == GF_OK && e == hipSuccess) e = hipStreamSynchronize(S.stream);
                    if (r == GF_OK && e == hipSuccess) {
                        big.resize(l);
                        if (l) e = hipMemcpy(big.data(), slot.p, l, hipMemcpyDeviceToHost);
                    }
                    if (r != GF_OK || e != hipSuccess) {
                        slot.release();
                        meta.release();
                        return r != GF_OK ? r : hipFail(e, "overflow tile copy");
                    }
                    m.st[t] = tst;
                    m.len[t] = l;
                    len = tst == GF_OK ? l : 0;
                    src = big.data();
                }
                if (total + len <= blobCap) {
                    if (len) memcpy(blob + total, src, len);
                } else {
                    overCap = true;
                }
                total += len;
                offsets[t0 + t + 1] = total;
            }
            slot.release();
            meta.release();
        }
        if (status) memcpy(status + t0, m.st, n * 4);
        if (predictors) memcpy(predictors + t0, m.pred, n);
        return GF_OK;
    };

    for (size_t k = 0; k < nChunks + 2; k++) {
        if (k >= 2 && k - 2 < nChunks && (s = stageC(k - 2)) != GF_OK) return s;     // frees slot (k - 2) % 3 ... used again at k + 1
        if (k < nChunks) {
            HostSlot &S = P->slot[k % HOST_SLOTS];
            const size_t t0 = k * chunk, n = std::min(chunk, nTiles - t0);
            const Meta m = metaOf(S);
            const int32_t *src = values + t0 * cells;
            if (!pinnedIn) {
                parallelCopy(S.hIn.p, src, n * cells * 4);
                src = (const int32_t *)S.hIn.p;
            }
            GF_HIP(hipMemcpyAsync(S.dValues.p, src, n * cells * 4, hipMemcpyHostToDevice, S.stream));
            // the codec kernels of successive chunks share the context's per-tile records: they run one after the other
            // (the copies around them overlap freely)
            if (k > 0) GF_HIP(hipStreamWaitEvent(S.stream, P->slot[(k - 1) % HOST_SLOTS].evK, 0));
            s = encodeBatchDev(kind, c, S.stream, codecIndex, nRows, nCols, n, (const int32_t *)S.dValues.p, (uint8_t *)S.dSlots.p,
                               stride, (uint32_t *)S.dLengths.p, (uint8_t *)S.dPred.p, (int32_t *)S.dStatus.p, GF_PM_ALL);
            if (s != GF_OK) return s;
            GF_HIP(hipEventRecord(S.evK, S.stream));
            GF_HIP(hipMemcpyAsync(m.len, S.dLengths.p, n * 4, hipMemcpyDeviceToHost, S.stream));
            GF_HIP(hipMemcpyAsync(m.st, S.dStatus.p, n * 4, hipMemcpyDeviceToHost, S.stream));
            GF_HIP(hipMemcpyAsync(m.pred, S.dPred.p, n, hipMemcpyDeviceToHost, S.stream));
            GF_HIP(gf_launch_compact(n, (const uint8_t *)S.dSlots.p, stride, (const uint32_t *)S.dLengths.p, (uint64_t *)S.dOffsets.p,
                                     (uint8_t *)S.dBlob.p, S.dBlob.bytes, S.stream, (const int32_t *)S.dStatus.p));
            GF_HIP(hipMemcpyAsync(m.off, S.dOffsets.p, (n + 1) * 8, hipMemcpyDeviceToHost, S.stream));
            GF_HIP(hipEventRecord(S.evA, S.stream));
        }
        if (k >= 1 && k - 1 < nChunks && (s = stageB(k - 1)) != GF_OK) return s;
    }
    (void)pinnedOut;
    return overCap ? GF_ERR_CAPACITY : GF_OK;
}

// The pipelined host-memory decode.  Packing t is lens[t] bytes at blob + starts[t] (starts / lens null: the usual offsets array,
// packings back to back) and its cells go to tile dstTile[t] of `values`, its status to status[dstTile[t]] (dstTile null: t).
// The scattered form (round 4) is what the default-codec-list and tile-record readers use: the packings of one codec among a
// batch's, or the elements inside framed records, are gathered straight into the pinned staging buffer of their chunk and the
// decoded tiles leave the staging buffer for their own place -- no intermediate blob, no intermediate tile array.
static gf_status decodeBatchHostG(int kind, gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                  const uint64_t *offsets, const uint64_t *starts, const uint32_t *lens, const uint32_t *dstTile,
                                  int32_t *values, int32_t *status)
{
    if (!c || nRows < 1 || nCols < 1 || !blob || (!offsets && !(starts && lens)) || (!values && nTiles)) return GF_ERR_ARG;
    if (!starts)
        for (size_t t = 0; t < nTiles; t++)
            if (offsets[t + 1] < offsets[t] || offsets[t + 1] - offsets[t] > 0xFFFFFFFFull) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    gf_host_pipe *P;
    gf_status s = hostPipe(c, &P);
    if (s != GF_OK) return s;
    const size_t cells = (size_t)nRows * (size_t)nCols;
    const size_t chunk = hostChunkTiles(cells, nTiles, kind);
    const size_t nChunks = (nTiles + chunk - 1) / chunk;
    const bool pinnedOut = nTiles && !dstTile && isPinned(values);
    if ((s = gf_context_reserve(c, nRows, nCols, chunk)) != GF_OK) return s;
    auto lenOf = [&](size_t t) -> uint64_t { return starts ? (uint64_t)lens[t] : offsets[t + 1] - offsets[t]; };
    // the largest blob slice of a chunk
    uint64_t maxSlice = 0;
    for (size_t k = 0; k < nChunks; k++) {
        const size_t t0 = k * chunk, t1 = std::min(nTiles, t0 + chunk);
        uint64_t slice = 0;
        if (starts)
            for (size_t t = t0; t < t1; t++) slice += lens[t];
        else slice = offsets[t1] - offsets[t0];
        maxSlice = std::max(maxSlice, slice);
    }
    const size_t metaBytes = roundUp((chunk + 1) * 8, 64) + 2 * roundUp(chunk * 4, 64) + 64;
    for (int i = 0; i < HOST_SLOTS && (size_t)i < nChunks; i++) {
        HostSlot &S = P->slot[i];
        if ((s = S.dValues.ensure(chunk * cells * 4 + 16)) != GF_OK) return s;
        if ((s = S.dBlob.ensure(maxSlice + 64)) != GF_OK) return s;
        if ((s = S.dLengths.ensure(chunk * 4 + 16)) != GF_OK) return s;
        if ((s = S.dStatus.ensure(chunk * 4 + 16)) != GF_OK) return s;
        if ((s = S.dOffsets.ensure((chunk + 1) * 8 + 16)) != GF_OK) return s;
        if ((s = S.hIn.ensure(maxSlice + 64)) != GF_OK) return s;
        if (!pinnedOut && (s = S.hOut.ensure(chunk * cells * 4)) != GF_OK) return s;
        if ((s = S.hMeta.ensure(metaBytes)) != GF_OK) return s;
    }
    auto finish = [&](size_t k) -> gf_status {
        HostSlot &S = P->slot[k % HOST_SLOTS];
        const size_t t0 = k * chunk, n = std::min(chunk, nTiles - t0);
        GF_HIP(hipEventSynchronize(S.evA));
        const uint8_t *mb = (const uint8_t *)S.hMeta.p;
        const int32_t *st = (const int32_t *)(mb + roundUp((chunk + 1) * 8, 64) + roundUp(chunk * 4, 64));
        if (dstTile) {
            const int32_t *src = (const int32_t *)S.hOut.p;
            parallelFor(n, [&](size_t t) {
                if (st[t] == GF_OK) memcpy(values + (size_t)dstTile[t0 + t] * cells, src + t * cells, cells * 4);
                if (status) status[dstTile[t0 + t]] = st[t];
            });
            return GF_OK;
        }
        if (!pinnedOut) parallelCopy(values + t0 * cells, S.hOut.p, n * cells * 4);
        if (status) memcpy(status + t0, st, n * 4);
        return GF_OK;
    };
    for (size_t k = 0; k < nChunks + (HOST_SLOTS - 1); k++) {
        if (k >= (size_t)(HOST_SLOTS - 1) && (s = finish(k - (HOST_SLOTS - 1))) != GF_OK) return s;
        if (k >= nChunks) continue;
        HostSlot &S = P->slot[k % HOST_SLOTS];
        const size_t t0 = k * chunk, n = std::min(chunk, nTiles - t0);
        uint8_t *mb = (uint8_t *)S.hMeta.p;
        uint64_t *rel = (uint64_t *)mb;
        uint32_t *len = (uint32_t *)(mb + roundUp((chunk + 1) * 8, 64));
        int32_t *st = (int32_t *)((uint8_t *)len + roundUp(chunk * 4, 64));
        uint64_t bytes = 0;
        for (size_t t = 0; t < n; t++) {
            rel[t] = bytes;
            len[t] = (uint32_t)lenOf(t0 + t);
            bytes += len[t];
        }
        rel[n] = bytes;
        if (bytes && !starts) memcpy(S.hIn.p, blob + offsets[t0], bytes);    // the slice starts 4-byte aligned in the staging buffer
        if (bytes && starts) {
            uint8_t *dst = (uint8_t *)S.hIn.p;
            parallelFor(n, [&](size_t t) { memcpy(dst + rel[t], blob + starts[t0 + t], len[t]); });
        }
        if (bytes) GF_HIP(hipMemcpyAsync(S.dBlob.p, S.hIn.p, bytes, hipMemcpyHostToDevice, S.stream));
        GF_HIP(hipMemcpyAsync(S.dOffsets.p, rel, (n + 1) * 8, hipMemcpyHostToDevice, S.stream));
        GF_HIP(hipMemcpyAsync(S.dLengths.p, len, n * 4, hipMemcpyHostToDevice, S.stream));
        if (k > 0) GF_HIP(hipStreamWaitEvent(S.stream, P->slot[(k - 1) % HOST_SLOTS].evK, 0));   // kernels one chunk at a time
        if (kind == KIND_DEFLATE)
            s = deflateDecodeDev(c, S.stream, nRows, nCols, n, (const uint8_t *)S.dBlob.p, bytes + 32, (const uint64_t *)S.dOffsets.p, 0,
                                 (const uint32_t *)S.dLengths.p, (int32_t *)S.dValues.p, (int32_t *)S.dStatus.p);
        else if (kind == KIND_FLOAT)
            s = floatDecodeDev(c, S.stream, nRows, nCols, n, (const uint8_t *)S.dBlob.p, bytes + 32, (const uint64_t *)S.dOffsets.p,
                               (const uint32_t *)S.dLengths.p, (float *)S.dValues.p, (int32_t *)S.dStatus.p);
        else
            s = decodeBatchDev(kind, c, S.stream, nRows, nCols, n, (const uint8_t *)S.dBlob.p, bytes + 32, (const uint64_t *)S.dOffsets.p, 0,
                               (const uint32_t *)S.dLengths.p, (int32_t *)S.dValues.p, (int32_t *)S.dStatus.p);
        if (s != GF_OK) return s;
        GF_HIP(hipEventRecord(S.evK, S.stream));
        GF_HIP(hipMemcpyAsync(pinnedOut ? (void *)(values + t0 * cells) : S.hOut.p, S.dValues.p, n * cells * 4, hipMemcpyDeviceToHost,
                              S.stream));
        GF_HIP(hipMemcpyAsync(st, S.dStatus.p, n * 4, hipMemcpyDeviceToHost, S.stream));
        GF_HIP(hipEventRecord(S.evA, S.stream));
    }
    return GF_OK;
}

static gf_status decodeBatchHost(int kind, gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                 const uint64_t *offsets, int32_t *values, int32_t *status)
{
    if (!offsets) return GF_ERR_ARG;
    return decodeBatchHostG(kind, c, nRows, nCols, nTiles, blob, offsets, nullptr, nullptr, nullptr, values, status);
}

// ------------------------------------------------------------------ one tile per call (BASELINE config 1)
// What a stock Gridfour application reaches without a patched GvrsFile: CodecMaster hands a codec ONE tile per call
// (gvrs/CodecMaster.java:150-169, RasterTileCache.java:418-421).  Through the batch machinery that was 180-190 us per tile on an
// MI355X (three stream slots, five asynchronous copies, events, a dozen API calls) against 10-20 us of kernels.  Here (round 4):
// per context a page-locked input and output buffer and, per (direction, codec, tile shape, codec index), ONE hipGraph recorded
// from the same device entry points the batches use -- host-to-device copy of the input, the kernels; the outputs (packing,
// length, status / cells, status) are written by the kernels straight into the page-locked output buffer -- replayed with one
// launch and waited for by polling the stream.  The first call of a kind takes the batch path; the second runs the lean sequence
// once outside a capture (code objects of the 1,024-thread builds, the kernels' LDS attributes: what a capture must not do) and
// records the graph.  A capture that fails is remembered (the batch path from then on); graphs are recorded again when a device
// buffer they hold has moved (gf_context::bufMoves).
struct gf_single_graph {
    int dir, kind, nRows, nCols, codecIndex;
    size_t copyBytes;
    hipGraph_t graph;
    hipGraphExec_t exec;
};
struct gf_single {
    void *hIn = nullptr, *hOut = nullptr;
    size_t hInBytes = 0, hOutBytes = 0;
    DevBuf dIn;
    std::vector<gf_single_graph> graphs;
    uint64_t moves = 0;                                          // gf_context::bufMoves when the graphs were recorded
    std::vector<std::pair<int, std::pair<int, int>>> warmed;     // (dir * 8 + kind, shape) that ran once through the batch path
    std::vector<std::pair<int, std::pair<int, int>>> refused;    // ... whose capture failed: the batch path from then on
};
static void singleDropGraphs(gf_single *sg)
{
    for (auto &g : sg->graphs) {
        (void)hipGraphExecDestroy(g.exec);
        (void)hipGraphDestroy(g.graph);
    }
    sg->graphs.clear();
}
// The graphs hold device addresses of the context's buffers (tree / selection records, flags, workspace, dIn): when any device
// buffer of the context has moved since they were recorded, they are recorded again.
static gf_status singleCheckMoves(gf_context *c, gf_single *sg)
{
    const uint64_t now = c->bufMoves.load(std::memory_order_relaxed);
    if (sg->moves == now || sg->graphs.empty()) {
        sg->moves = now;
        return GF_OK;
    }
    GF_HIP(hipStreamSynchronize(c->stream));
    singleDropGraphs(sg);
    sg->moves = now;
    return GF_OK;
}
static bool singleRefused(gf_single *sg, int key, int nRows, int nCols, bool add = false)
{
    for (auto &w : sg->refused)
        if (w.first == key && w.second.first == nRows && w.second.second == nCols) return true;
    if (add) sg->refused.push_back({key, {nRows, nCols}});
    return false;
}
void gf_single_destroy(gf_single *sg)
{
    if (!sg) return;
    singleDropGraphs(sg);
    if (sg->hIn) (void)hipHostFree(sg->hIn);
    if (sg->hOut) (void)hipHostFree(sg->hOut);
    sg->dIn.release();
    delete sg;
}
static gf_status singleEnsure(gf_context *c, size_t inBytes, size_t outBytes)
{
    if (!c->single) {
        c->single = new (std::nothrow) gf_single;
        if (c->single) c->single->dIn.moves = &c->bufMoves;
    }
    gf_single *sg = c->single;
    if (!sg) return GF_ERR_HIP;
    if (sg->hInBytes < inBytes || sg->hOutBytes < outBytes || sg->dIn.bytes < inBytes) {
        // the graphs hold the old addresses
        GF_HIP(hipStreamSynchronize(c->stream));
        singleDropGraphs(sg);
        if (sg->hInBytes < inBytes) {
            if (sg->hIn) (void)hipHostFree(sg->hIn);
            sg->hIn = nullptr;
            sg->hInBytes = 0;
            GF_HIP(hipHostMalloc(&sg->hIn, roundUp(inBytes, 4096), hipHostMallocDefault));
            sg->hInBytes = roundUp(inBytes, 4096);
        }
        if (sg->hOutBytes < outBytes) {
            if (sg->hOut) (void)hipHostFree(sg->hOut);
            sg->hOut = nullptr;
            sg->hOutBytes = 0;
            GF_HIP(hipHostMalloc(&sg->hOut, roundUp(outBytes, 4096), hipHostMallocDefault));
            sg->hOutBytes = roundUp(outBytes, 4096);
        }
        const gf_status s = sg->dIn.ensure(inBytes);
        if (s != GF_OK) return s;
    }
    return GF_OK;
}
static bool singleWarmed(gf_single *sg, int key, int nRows, int nCols)
{
    for (auto &w : sg->warmed)
        if (w.first == key && w.second.first == nRows && w.second.second == nCols) return true;
    sg->warmed.push_back({key, {nRows, nCols}});
    return false;
}
// waits for the stream without the interrupt path of hipStreamSynchronize (tens of microseconds on its own)
static gf_status singleWait(hipStream_t st)
{
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return GF_OK;
        if (e != hipErrorNotReady) {
            g_lastError = hipGetErrorString(e);
            return GF_ERR_HIP;
        }
    }
}

static gf_status encodeBatchDev(int kind, gf_context *c, void *stream, int codecIndex, int nRows, int nCols, size_t nTiles,
                                const int32_t *dValues, uint8_t *dOut, size_t slotStride, uint32_t *dLengths, uint8_t *dPredictors,
                                int32_t *dStatus, int predictorMask);

// returns GF_ERR_UNSUPPORTED where the caller should take the batch path instead (first call of a kind, a packing beyond the slot)
static gf_status singleEncode(int kind, gf_context *c, int codecIndex, int nRows, int nCols, const int32_t *values, uint8_t *out,
                              size_t outCap, size_t *outLen, int32_t *tileStatus)
{
    if (!c || nRows < 1 || nCols < 1 || !values || !outLen) return GF_ERR_ARG;
    const size_t cells = (size_t)nRows * (size_t)nCols;
    if (cells * 4 > ((size_t)8 << 20)) return GF_ERR_UNSUPPORTED;            // (large tiles: the batch path's copies are not what they wait for)
    GF_HIP(hipSetDevice(c->device));
    const size_t stride = gf_huffman_default_stride(nRows, nCols);
    gf_status s = singleEnsure(c, std::max(cells * 4, stride + 16), std::max(stride + 64, cells * 4 + 64));
    if (s != GF_OK) return s;
    gf_single *sg = c->single;
    if (singleRefused(sg, kind, nRows, nCols)) return GF_ERR_UNSUPPORTED;
    if (!singleWarmed(sg, kind, nRows, nCols)) return GF_ERR_UNSUPPORTED;
    if ((s = singleCheckMoves(c, sg)) != GF_OK) return s;
    gf_single_graph *g = nullptr;
    for (auto &x : sg->graphs)
        if (x.dir == 0 && x.kind == kind && x.nRows == nRows && x.nCols == nCols && x.codecIndex == codecIndex) g = &x;
    uint8_t *hOut = (uint8_t *)sg->hOut;
    uint32_t *hLen = (uint32_t *)(hOut + stride);
    int32_t *hSt = (int32_t *)(hOut + stride + 4);
    if (!g) {
        if ((s = gf_context_reserve(c, nRows, nCols, 1)) != GF_OK) return s;
        const uint64_t movesBefore = c->bufMoves.load(std::memory_order_relaxed);
        if (movesBefore != sg->moves) {                                       // (the reservation moved a buffer the other graphs hold)
            if ((s = singleCheckMoves(c, sg)) != GF_OK) return s;
        }
        memcpy(sg->hIn, values, cells * 4);
        auto sequence = [&]() -> gf_status {
            if (hipMemcpyAsync(sg->dIn.p, sg->hIn, cells * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) return GF_ERR_HIP;
            g_lean = 1;
            const gf_status r = encodeBatchDev(kind, c, c->stream, codecIndex, nRows, nCols, 1, (const int32_t *)sg->dIn.p, hOut, stride,
                                               hLen, nullptr, hSt, GF_PM_ALL);
            g_lean = 0;
            return r;
        };
        // once outside a capture: the lean launches use builds of the kernels (1,024 threads) that the batch path of this shape may
        // never have touched -- their code objects are loaded and their LDS attributes set here, not inside the capture
        if ((s = sequence()) != GF_OK) return s;
        GF_HIP(hipStreamSynchronize(c->stream));
        gf_single_graph ng{0, kind, nRows, nCols, codecIndex, cells * 4, nullptr, nullptr};
        GF_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        s = sequence();
        const hipError_t e2 = hipStreamEndCapture(c->stream, &ng.graph);
        const bool moved = c->bufMoves.load(std::memory_order_relaxed) != movesBefore;   // (a capture must not allocate; if it did, its addresses are void)
        if (s != GF_OK || e2 != hipSuccess || !ng.graph || moved ||
            hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0) != hipSuccess) {
            if (ng.graph) (void)hipGraphDestroy(ng.graph);
            (void)hipGetLastError();
            if (!moved) singleRefused(sg, kind, nRows, nCols, true);          // every later call of this kind: the batch path, directly
            return GF_ERR_UNSUPPORTED;
        }
        sg->graphs.push_back(ng);
        g = &sg->graphs.back();
    }
    memcpy(sg->hIn, values, cells * 4);
    GF_HIP(hipGraphLaunch(g->exec, c->stream));
    if ((s = singleWait(c->stream)) != GF_OK) return s;
    const int32_t st = *hSt;
    const size_t len = *hLen;
    if (st == GF_OVERFLOW || st == GF_K_LEAN_RETRY) return GF_ERR_UNSUPPORTED;   // longer than the slot, or a kernel this launch left out: the batch path
    *tileStatus = st;
    *outLen = st == GF_OK ? len : 0;
    if (st == GF_OK) {
        if (len > outCap) return GF_ERR_CAPACITY;
        memcpy(out, hOut, len);
    }
    return GF_OK;
}

static gf_status decodeBatchDev(int kind, gf_context *c, void *stream, int nRows, int nCols, size_t nTiles, const uint8_t *dBlob,
                                size_t blobBytes, const uint64_t *dOffsets, size_t slotStride, const uint32_t *dLengths, int32_t *dValues,
                                int32_t *dStatus, uint32_t *analysis, uint32_t *pairCounts);

static gf_status singleDecode(int kind, gf_context *c, int nRows, int nCols, const uint8_t *packing, size_t len, int32_t *values,
                              int32_t *tileStatus)
{
    if (!c || nRows < 1 || nCols < 1 || !packing || !values) return GF_ERR_ARG;
    const size_t cells = (size_t)nRows * (size_t)nCols;
    if (cells * 4 > ((size_t)8 << 20)) return GF_ERR_UNSUPPORTED;
    GF_HIP(hipSetDevice(c->device));
    const size_t stride = gf_huffman_default_stride(nRows, nCols);
    if (len + 16 > stride) return GF_ERR_UNSUPPORTED;                         // (an unusually long packing: the batch path)
    gf_status s = singleEnsure(c, std::max(cells * 4, stride + 16), std::max(stride + 64, cells * 4 + 64));
    if (s != GF_OK) return s;
    gf_single *sg = c->single;
    if (singleRefused(sg, 8 + kind, nRows, nCols)) return GF_ERR_UNSUPPORTED;
    if (!singleWarmed(sg, 8 + kind, nRows, nCols)) return GF_ERR_UNSUPPORTED;
    if ((s = singleCheckMoves(c, sg)) != GF_OK) return s;
    // the copy moves [length, 12 spare bytes, packing]: sized in powers of two so that a few graphs serve every length
    size_t copyBytes = 4096;
    while (copyBytes < len + 16 + 8) copyBytes <<= 1;                         // (+ 8: the kernels read whole words behind the last byte)
    copyBytes = std::min(copyBytes, roundUp(stride + 16, 16));
    gf_single_graph *g = nullptr;
    for (auto &x : sg->graphs)
        if (x.dir == 1 && x.kind == kind && x.nRows == nRows && x.nCols == nCols && x.copyBytes == copyBytes) g = &x;
    uint8_t *hIn = (uint8_t *)sg->hIn, *hOut = (uint8_t *)sg->hOut;
    int32_t *hSt = (int32_t *)(hOut + cells * 4);
    const uint32_t len32 = (uint32_t)len;
    auto fillInput = [&]() {
        memcpy(hIn, &len32, 4);
        memcpy(hIn + 16, packing, len);
        memset(hIn + 16 + len, 0, std::min<size_t>(8, copyBytes - 16 - len));
    };
    if (!g) {
        if ((s = gf_context_reserve(c, nRows, nCols, 1)) != GF_OK) return s;
        const uint64_t movesBefore = c->bufMoves.load(std::memory_order_relaxed);
        if (movesBefore != sg->moves) {
            if ((s = singleCheckMoves(c, sg)) != GF_OK) return s;
        }
        uint8_t *dIn = (uint8_t *)sg->dIn.p;
        fillInput();
        auto sequence = [&]() -> gf_status {
            if (hipMemcpyAsync(dIn, hIn, copyBytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return GF_ERR_HIP;
            g_lean = 1;
            const gf_status r = decodeBatchDev(kind, c, c->stream, nRows, nCols, 1, dIn + 16, copyBytes - 16, nullptr, copyBytes - 16,
                                               (const uint32_t *)dIn, (int32_t *)hOut, hSt, nullptr, nullptr);
            g_lean = 0;
            return r;
        };
        if ((s = sequence()) != GF_OK) return s;                              // (outside a capture first: see singleEncode)
        GF_HIP(hipStreamSynchronize(c->stream));
        gf_single_graph ng{1, kind, nRows, nCols, 0, copyBytes, nullptr, nullptr};
        GF_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        s = sequence();
        const hipError_t e2 = hipStreamEndCapture(c->stream, &ng.graph);
        const bool moved = c->bufMoves.load(std::memory_order_relaxed) != movesBefore;
        if (s != GF_OK || e2 != hipSuccess || !ng.graph || moved ||
            hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0) != hipSuccess) {
            if (ng.graph) (void)hipGraphDestroy(ng.graph);
            (void)hipGetLastError();
            if (!moved) singleRefused(sg, 8 + kind, nRows, nCols, true);
            return GF_ERR_UNSUPPORTED;
        }
        sg->graphs.push_back(ng);
        g = &sg->graphs.back();
    }
    fillInput();
    GF_HIP(hipGraphLaunch(g->exec, c->stream));
    if ((s = singleWait(c->stream)) != GF_OK) return s;
    if (*hSt == GF_K_LEAN_RETRY) return GF_ERR_UNSUPPORTED;                   // a tile the fast kernel leaves to the others: the batch path
    *tileStatus = *hSt;
    if (*hSt == GF_OK) memcpy(values, hOut, cells * 4);
    return GF_OK;
}

extern "C" {

gf_status gf_huffman_encode_batch_i32(gf_context *c, int codecIndex, int nRows, int nCols, size_t nTiles,
                                      const int32_t *values, uint8_t *blob, size_t blobCap, uint64_t *offsets,
                                      uint8_t *predictors, int32_t *status)
{
    GF_CTX_LOCK(c);
    return encodeBatchHost(KIND_HUFFMAN, c, codecIndex, nRows, nCols, nTiles, values, blob, blobCap, offsets, predictors, status);
}

gf_status gf_huffman_decode_batch_i32(gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                      const uint64_t *offsets, int32_t *values, int32_t *status)
{
    GF_CTX_LOCK(c);
    return decodeBatchHost(KIND_HUFFMAN, c, nRows, nCols, nTiles, blob, offsets, values, status);
}

gf_status gf_canon_encode_batch_i32(gf_context *c, int codecIndex, int nRows, int nCols, size_t nTiles,
                                    const int32_t *values, uint8_t *blob, size_t blobCap, uint64_t *offsets,
                                    uint8_t *predictors, int32_t *status)
{
    GF_CTX_LOCK(c);
    return encodeBatchHost(KIND_CANON, c, codecIndex, nRows, nCols, nTiles, values, blob, blobCap, offsets, predictors, status);
}

gf_status gf_canon_decode_batch_i32(gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                    const uint64_t *offsets, int32_t *values, int32_t *status)
{
    GF_CTX_LOCK(c);
    return decodeBatchHost(KIND_CANON, c, nRows, nCols, nTiles, blob, offsets, values, status);
}

gf_status gf_canon_encode_i32(gf_context *c, int codecIndex, int nRows, int nCols, const int32_t *values, uint8_t *out,
                              size_t outCap, size_t *outLen)
{
    GF_CTX_LOCK(c);
    if (!outLen) return GF_ERR_ARG;
    uint64_t offsets[2] = {0, 0};
    int32_t st = 0;
    gf_status s = singleEncode(KIND_CANON, c, codecIndex, nRows, nCols, values, out, outCap, outLen, &st);
    if (s == GF_OK) return (gf_status)st;
    if (s != GF_ERR_UNSUPPORTED) return s;
    s = gf_canon_encode_batch_i32(c, codecIndex, nRows, nCols, 1, values, out, outCap, offsets, nullptr, &st);
    *outLen = (size_t)offsets[1];
    if (s != GF_OK) return s;
    return (gf_status)st;
}

gf_status gf_canon_decode_i32(gf_context *c, int nRows, int nCols, const uint8_t *packing, size_t len, int32_t *values)
{
    GF_CTX_LOCK(c);
    uint64_t offsets[2] = {0, (uint64_t)len};
    int32_t st = 0;
    gf_status s = singleDecode(KIND_CANON, c, nRows, nCols, packing, len, values, &st);
    if (s == GF_OK) return (gf_status)st;
    if (s != GF_ERR_UNSUPPORTED) return s;
    s = gf_canon_decode_batch_i32(c, nRows, nCols, 1, packing, offsets, values, &st);
    if (s != GF_OK) return s;
    return (gf_status)st;
}

gf_status gf_huffman_encode_i32(gf_context *c, int codecIndex, int nRows, int nCols, const int32_t *values,
                                uint8_t *out, size_t outCap, size_t *outLen)
{
    GF_CTX_LOCK(c);
    if (!outLen) return GF_ERR_ARG;
    uint64_t offsets[2] = {0, 0};
    int32_t st = 0;
    gf_status s = singleEncode(KIND_HUFFMAN, c, codecIndex, nRows, nCols, values, out, outCap, outLen, &st);
    if (s == GF_OK) return (gf_status)st;
    if (s != GF_ERR_UNSUPPORTED) return s;
    s = gf_huffman_encode_batch_i32(c, codecIndex, nRows, nCols, 1, values, out, outCap, offsets, nullptr, &st);
    *outLen = (size_t)offsets[1];
    if (s != GF_OK) return s;
    return (gf_status)st;
}

gf_status gf_huffman_decode_i32(gf_context *c, int nRows, int nCols, const uint8_t *packing, size_t len,
                                int32_t *values)
{
    GF_CTX_LOCK(c);
    uint64_t offsets[2] = {0, (uint64_t)len};
    int32_t st = 0;
    gf_status s = singleDecode(KIND_HUFFMAN, c, nRows, nCols, packing, len, values, &st);
    if (s == GF_OK) return (gf_status)st;
    if (s != GF_ERR_UNSUPPORTED) return s;
    s = gf_huffman_decode_batch_i32(c, nRows, nCols, 1, packing, offsets, values, &st);
    if (s != GF_OK) return s;
    return (gf_status)st;
}


// ------------------------------------------------------------------ LSOP12

size_t gf_lsop12_residual_count(int nRows, int nCols)
{
    if (nRows < 6 || nCols < 6) return 0;
    return (size_t)4 * nRows + (size_t)2 * nCols - 9 + (size_t)(nRows - 2) * (size_t)(nCols - 4);
}

size_t gf_lsop12_max_packing(int nRows, int nCols)
{
    // 55 header bytes (59 with the value checksum) + two canonical-Huffman streams (tables < 750 bytes each, at most 84 bits per
    // value + end-of-text)
    const size_t n = gf_lsop12_residual_count(nRows, nCols);
    return roundUp(59 + 2 * 768 + (n * 84 + 2 * 15 + 7) / 8 + 16, 16);
}

gf_status gf_lsop12_predict_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles, const int32_t *dValues,
                                int32_t *dResiduals, size_t resStride, uint32_t *dCoefs, int32_t *dStatus)
{
    GF_CTX_LOCK(c);
    if (!c || !dValues || !dResiduals || !dCoefs || !dStatus || nRows < 1 || nCols < 1) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    if ((size_t)nRows * (size_t)nCols >= (1ull << 28)) return GF_ERR_UNSUPPORTED;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (nRows < 6 || nCols < 6) {                       // LsOptimalPredictor12.java:114-116 -> null
        if (nTiles) GF_HIP(hipMemsetD32Async((hipDeviceptr_t)dStatus, GF_DECLINED, nTiles, st));
        return GF_OK;
    }
    if (resStride < gf_lsop12_residual_count(nRows, nCols)) return GF_ERR_ARG;
    GF_HIP(gf_launch_lsop_predict(dValues, dResiduals, resStride, dCoefs, dStatus, nTiles, nRows, nCols, st));
    return GF_OK;
}

gf_status gf_lsop12_reconstruct_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles,
                                    const int32_t *dResiduals, size_t resStride, const uint32_t *dCoefs,
                                    const int32_t *dInStatus, int32_t *dValues, int32_t *dStatus)
{
    GF_CTX_LOCK(c);
    if (!c || !dValues || !dResiduals || !dCoefs || !dStatus) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    if (nRows < 6 || nCols < 6 || resStride < gf_lsop12_residual_count(nRows, nCols)) return GF_ERR_ARG;
    GF_HIP(gf_launch_lsop_reconstruct(dResiduals, resStride, dCoefs, dInStatus, dValues, dStatus, nTiles, nRows, nCols,
                                      stream ? (hipStream_t)stream : c->stream));
    return GF_OK;
}

gf_status gf_lsop12_encode_batch_i32_dev(gf_context *c, void *stream, int codecIndex, int nRows, int nCols, size_t nTiles,
                                         const int32_t *dValues, uint8_t *dOut, size_t slotStride, uint32_t *dLengths,
                                         int32_t *dStatus, int32_t *dResiduals, size_t resStride, uint32_t *dCoefs,
                                         int32_t *dScratchStatus)
{
    return gf_lsop12_encode_batch_i32_dev_ex(c, stream, codecIndex, nRows, nCols, nTiles, dValues, 0, dOut, slotStride, dLengths, dStatus,
                                             dResiduals, resStride, dCoefs, dScratchStatus);
}

}  // extern "C"

// (round 6, advice) int32Residuals: the caller reads d_residuals itself (gf_lsop12_encode_batch_i32's Deflate stage) -- a parameter of
// this internal form, no longer an undocumented bit of the public flags word
static gf_status lsopEncodeBatchDev(gf_context *c, void *stream, int codecIndex, int nRows, int nCols, size_t nTiles,
                                    const int32_t *dValues, int flags, bool int32Residuals, uint8_t *dOut, size_t slotStride,
                                    uint32_t *dLengths, int32_t *dStatus, int32_t *dResiduals, size_t resStride, uint32_t *dCoefs,
                                    int32_t *dScratchStatus);

extern "C" {

// ... with LsEncoder12's switches (flags: GF_LSOP_VALUE_CHECKSUM = setValueChecksumEnabled, lsop/LsEncoder12.java:117-119; the
// Deflate alternative needs the host's zlib and is not a device-resident operation: GF_LSOP_DEFLATE is accepted and means nothing
// here; any other bit is GF_ERR_ARG)
gf_status gf_lsop12_encode_batch_i32_dev_ex(gf_context *c, void *stream, int codecIndex, int nRows, int nCols, size_t nTiles,
                                            const int32_t *dValues, int flags, uint8_t *dOut, size_t slotStride, uint32_t *dLengths,
                                            int32_t *dStatus, int32_t *dResiduals, size_t resStride, uint32_t *dCoefs,
                                            int32_t *dScratchStatus)
{
    if (flags & ~(GF_LSOP_DEFLATE | GF_LSOP_VALUE_CHECKSUM)) return GF_ERR_ARG;
    return lsopEncodeBatchDev(c, stream, codecIndex, nRows, nCols, nTiles, dValues, flags, false, dOut, slotStride, dLengths, dStatus,
                              dResiduals, resStride, dCoefs, dScratchStatus);
}

}  // extern "C"

static gf_status lsopEncodeBatchDev(gf_context *c, void *stream, int codecIndex, int nRows, int nCols, size_t nTiles,
                                    const int32_t *dValues, int flags, bool int32Residuals, uint8_t *dOut, size_t slotStride,
                                    uint32_t *dLengths, int32_t *dStatus, int32_t *dResiduals, size_t resStride, uint32_t *dCoefs,
                                    int32_t *dScratchStatus)
{
    GF_CTX_LOCK(c);
    if (!c || !dValues || !dOut || !dLengths || !dStatus || !dResiduals || !dCoefs || !dScratchStatus) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    if (slotStride % 16 != 0 || ((uintptr_t)dOut & 15) != 0 || slotStride < 64) return GF_ERR_ARG;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (nRows < 6 || nCols < 6) {
        if (nTiles) {
            GF_HIP(hipMemsetD32Async((hipDeviceptr_t)dStatus, GF_DECLINED, nTiles, st));
            GF_HIP(hipMemsetD32Async((hipDeviceptr_t)dLengths, 0, nTiles, st));
        }
        return GF_OK;
    }
    // Terrain-sized tiles (round 5): the first kernel keeps the tile in LDS as halfwords, writes the residuals as int16 and counts
    // the histograms on the way (k_lsop_predict16; records in the context's selection-record buffer, which gf_context_reserve
    // sizes) -- unless the caller wants the int32 residuals themselves (int32Residuals: the host's Deflate stage)
    const bool fast16 = gf_lsop_predict16_eligible(nRows, nCols) && !int32Residuals && resStride >= gf_lsop12_residual_count(nRows, nCols);
    uint32_t *hist16 = nullptr;
    gf_status s;
    if (fast16) {
        const size_t need = nTiles * gf_lsop_hist_rec_words() * 4 + 16;
        if (c->packRecs.bytes < need) {                    // not capture-safe: gf_context_reserve sizes this too
            if ((s = c->packRecs.ensure(need)) != GF_OK) return s;
        }
        hist16 = (uint32_t *)c->packRecs.p;
        GF_HIP(gf_launch_lsop_predict16(dValues, dResiduals, resStride, dCoefs, dScratchStatus, hist16, nTiles, nRows, nCols, st));
    } else {
        s = gf_lsop12_predict_dev(c, stream, nRows, nCols, nTiles, dValues, dResiduals, resStride, dCoefs, dScratchStatus);
        if (s != GF_OK) return s;
    }
    const uint32_t n0 = (uint32_t)(4 * nRows + 2 * nCols - 9), n1 = (uint32_t)((nRows - 2) * (nCols - 4));
    if (4ull * ((uint64_t)n1 + 1) >= (1ull << 22)) return GF_ERR_UNSUPPORTED;       // 22-bit counts in the tree keys
    const int valueChecksum = (flags & GF_LSOP_VALUE_CHECKSUM) ? 1 : 0;
    if (valueChecksum)
        GF_HIP(gf_launch_lsop_value_crc(dValues, (size_t)nRows * (size_t)nCols, nTiles, nullptr, dCoefs, st));
    GF_HIP(gf_launch_canon_pack2(dResiduals, resStride, dCoefs, dScratchStatus, dOut, slotStride, dLengths, dStatus, nTiles,
                                 n0, n1, codecIndex, st, valueChecksum, hist16));
    return GF_OK;
}

extern "C" {

gf_status gf_lsop12_decode_batch_i32_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles,
                                         const uint8_t *dBlob, size_t blobBytes, const uint64_t *dOffsets, size_t slotStride,
                                         const uint32_t *dLengths, int32_t *dValues, int32_t *dStatus, int32_t *dResiduals,
                                         size_t resStride, uint32_t *dCoefs, int32_t *dScratchStatus)
{
    GF_CTX_LOCK(c);
    if (!c || !dBlob || !dLengths || !dValues || !dStatus || !dResiduals || !dCoefs || !dScratchStatus) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    if (((uintptr_t)dBlob & 3) != 0) return GF_ERR_ARG;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (nRows < 6 || nCols < 6) {
        if (nTiles) GF_HIP(hipMemsetD32Async((hipDeviceptr_t)dStatus, GF_ERR_BOUNDS, nTiles, st));
        return GF_OK;
    }
    if (resStride < gf_lsop12_residual_count(nRows, nCols)) return GF_ERR_ARG;
    const unsigned grid = gf_huffman_decode_grid(nTiles);
    gf_status s = lsopParseLengths(c, st, nTiles, dBlob, blobBytes, dOffsets, slotStride, dLengths);
    if (s != GF_OK) return s;
    // (round 6) interior residuals as byte planes in the reconstruction's order where a tile's values allow (gvrs_kernels.h)
#ifdef GF_LSOP_NO_PLANES                                    // (experiment builds: tools/ab_kernels.sh)
    const bool lsopPlanes = false;
#else
    const bool lsopPlanes = true;
#endif
    GF_HIP(gf_launch_lsop_unpack2(dBlob, blobBytes, dOffsets, slotStride, dLengths, dResiduals, resStride, dCoefs,
                                  dScratchStatus, nTiles, nRows, nCols, gf_lsop_unpack_lds_text(nRows, nCols), grid, st,
                                  (const uint32_t *)c->trees.p, g_decodeDebug,
                                  // (the serial walk of a lane pays where sixty-four tiles share a wave: large batches)
#ifdef GF_LSOP_NO_HEAD                                      // (experiment builds: tools/ab_kernels.sh)
                                  nullptr,
#else
                                  gf_prepass_tiles_per_wave(nTiles) == 64u ? (uint32_t *)c->trees.p + nTiles * (size_t)GF_CANON_REC_WORDS : nullptr,
#endif
                                  lsopPlanes));
    s = lsopUnpackM32Deflate(c, st, nRows, nCols, nTiles, dBlob, blobBytes, dOffsets, slotStride, dLengths, dResiduals, resStride,
                             dCoefs, dScratchStatus);
    if (s != GF_OK) return s;
    GF_HIP(gf_launch_lsop_reconstruct(dResiduals, resStride, dCoefs, dScratchStatus, dValues, dStatus, nTiles, nRows, nCols, st,
                                      lsopPlanes));
    return GF_OK;
}

}  // extern "C"

namespace {

// CodecM32.encode (compress/CodecM32.java:257-311) of a residual array: host-side glue for the Deflate container
size_t m32Pack(const int32_t *x, size_t n, std::vector<uint8_t> &out)
{
    out.resize(6 * n + 8);
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        const int len = gf_m32_len((uint32_t)x[i]);
        for (int b = 0; b < len; b++) out[k++] = (uint8_t)gf_m32_byte((uint32_t)x[i], len, b);
    }
    out.resize(k);
    return k;
}

void putLE32(uint8_t *p, uint32_t x) { p[0] = (uint8_t)x; p[1] = (uint8_t)(x >> 8); p[2] = (uint8_t)(x >> 16); p[3] = (uint8_t)(x >> 24); }
uint32_t getLE32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace

extern "C" {

// LsEncoder12.encode :122-219 for a batch in host memory.  deflateEnabled mirrors setDeflateEnabled (default true):
// the canonical-Huffman packing comes from the GPU; with Deflate enabled the host's zlib (level 6) compresses the two
// M32 streams and replaces the packing when strictly smaller (:180-216).  types[t] = container type written (2 / 1).
gf_status gf_lsop12_encode_batch_i32(gf_context *c, int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                                     int deflateEnabled, uint8_t *blob, size_t blobCap, uint64_t *offsets, uint8_t *types,
                                     int32_t *status)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || !values || !offsets || (!blob && blobCap)) return GF_ERR_ARG;
    if (deflateEnabled & ~(GF_LSOP_DEFLATE | GF_LSOP_VALUE_CHECKSUM)) return GF_ERR_ARG;      // (a bit mask since round 5: see the header)
    GF_HIP(hipSetDevice(c->device));
    if (nRows < 6 || nCols < 6) {
        for (size_t t = 0; t <= nTiles; t++) offsets[t] = 0;
        for (size_t t = 0; t < nTiles; t++) { if (status) status[t] = GF_DECLINED; if (types) types[t] = 0; }
        return GF_OK;
    }
    const size_t cells = (size_t)nRows * (size_t)nCols;
    const size_t nRes = gf_lsop12_residual_count(nRows, nCols), resStride = roundUp(nRes, 4);
    const size_t nInit = (size_t)4 * nRows + 2 * nCols - 9, nInt = nRes - nInit;
    const size_t stride = gf_lsop12_max_packing(nRows, nCols);
    gf_status s;
    if ((s = c->dValues.ensure(nTiles * cells * 4 + 16)) != GF_OK) return s;
    if ((s = c->dSlots.ensure(nTiles * stride + 16)) != GF_OK) return s;
    if ((s = c->dLengths.ensure(nTiles * 4 + 16)) != GF_OK) return s;
    if ((s = c->dStatus.ensure(nTiles * 4 + 16)) != GF_OK) return s;
    if ((s = c->dStatus2.ensure(nTiles * 4 + 16)) != GF_OK) return s;
    if ((s = c->dResiduals.ensure(nTiles * resStride * 4 + 16)) != GF_OK) return s;
    if ((s = c->dCoefs.ensure(nTiles * 64 + 16)) != GF_OK) return s;
    GF_HIP(hipMemcpyAsync(c->dValues.p, values, nTiles * cells * 4, hipMemcpyHostToDevice, c->stream));
    // deflateEnabled carries LsEncoder12's two switches as bits: GF_LSOP_DEFLATE (setDeflateEnabled) and GF_LSOP_VALUE_CHECKSUM
    // (setValueChecksumEnabled); any other bit was refused above
    const bool valueChecksum = (deflateEnabled & GF_LSOP_VALUE_CHECKSUM) != 0;
    deflateEnabled &= GF_LSOP_DEFLATE;
    const size_t hdrCanon = valueChecksum ? 59 : 55, hdrDeflate = valueChecksum ? 67 : 63;
    s = lsopEncodeBatchDev(c, c->stream, codecIndex, nRows, nCols, nTiles, (const int32_t *)c->dValues.p,
                           valueChecksum ? GF_LSOP_VALUE_CHECKSUM : 0, deflateEnabled != 0, (uint8_t *)c->dSlots.p, stride,
                           (uint32_t *)c->dLengths.p, (int32_t *)c->dStatus.p, (int32_t *)c->dResiduals.p, resStride,
                           (uint32_t *)c->dCoefs.p, (int32_t *)c->dStatus2.p);
    if (s != GF_OK) return s;
    std::vector<uint32_t> lengths(nTiles);
    std::vector<int32_t> st(nTiles);
    std::vector<uint8_t> slots(nTiles * stride);
    GF_HIP(hipMemcpyAsync(lengths.data(), c->dLengths.p, nTiles * 4, hipMemcpyDeviceToHost, c->stream));
    GF_HIP(hipMemcpyAsync(st.data(), c->dStatus.p, nTiles * 4, hipMemcpyDeviceToHost, c->stream));
    GF_HIP(hipMemcpyAsync(slots.data(), c->dSlots.p, nTiles * stride, hipMemcpyDeviceToHost, c->stream));
    std::vector<int32_t> res;
    std::vector<uint32_t> coefs;
    if (deflateEnabled) {
        res.resize(nTiles * resStride);
        coefs.resize(nTiles * 16);
        GF_HIP(hipMemcpyAsync(res.data(), c->dResiduals.p, nTiles * resStride * 4, hipMemcpyDeviceToHost, c->stream));
        GF_HIP(hipMemcpyAsync(coefs.data(), c->dCoefs.p, nTiles * 64, hipMemcpyDeviceToHost, c->stream));
    }
    GF_HIP(hipStreamSynchronize(c->stream));

    std::vector<std::vector<uint8_t>> alt(nTiles);          // Deflate container where it wins
    if (deflateEnabled) {
        parallelFor(nTiles, [&](size_t t) {
            if (st[t] != GF_OK) return;
            const size_t canonLength = lengths[t] - hdrCanon;
            const int32_t *r = res.data() + t * resStride;
            std::vector<uint8_t> mInt, mInit, zInt, zInit;
            const size_t nMX = m32Pack(r + nInit, nInt, mInt);
            if (!zDeflate(mInt.data(), nMX, 6, zInt)) return;
            if (zInt.empty() || zInt.size() >= canonLength || zInt.size() > nMX + 128) return;      // :185-187
            const size_t nMI = m32Pack(r, nInit, mInit);
            if (!zDeflate(mInit.data(), nMI, 6, zInit)) return;
            if (zInit.empty() || zInit.size() + zInt.size() >= canonLength || zInit.size() > nMI + 128) return;   // :194-196
            std::vector<uint8_t> &p = alt[t];
            p.resize(hdrDeflate + zInit.size() + zInt.size());
            p[0] = (uint8_t)codecIndex;
            p[1] = valueChecksum ? 0xC1 : 0x41;              // COMPRESSION_TYPE_DEFLATE | REVISION_FLAG (| VALUE_CHECKSUM_INCLUDED)
            p[2] = 12;
            for (int k = 0; k < 13; k++) putLE32(&p[3 + 4 * k], coefs[t * 16 + k]);
            putLE32(&p[55], (uint32_t)nMI);
            putLE32(&p[59], (uint32_t)nMX);
            if (valueChecksum) putLE32(&p[63], coefs[t * 16 + 13]);   // LsHeader.packHeader :259-261
            memcpy(&p[hdrDeflate], zInit.data(), zInit.size());
            memcpy(&p[hdrDeflate + zInit.size()], zInt.data(), zInt.size());
        });
    }
    uint64_t total = 0;
    for (size_t t = 0; t < nTiles; t++) {
        offsets[t] = total;
        if (st[t] == GF_OK) total += alt[t].empty() ? lengths[t] : alt[t].size();
        if (types) types[t] = st[t] == GF_OK ? (alt[t].empty() ? 2 : 1) : 0;
    }
    offsets[nTiles] = total;
    if (status) memcpy(status, st.data(), nTiles * 4);
    if (total > blobCap) return GF_ERR_CAPACITY;
    for (size_t t = 0; t < nTiles; t++) {
        if (st[t] != GF_OK) continue;
        if (alt[t].empty()) memcpy(blob + offsets[t], slots.data() + t * stride, lengths[t]);
        else memcpy(blob + offsets[t], alt[t].data(), alt[t].size());
    }
    return GF_OK;
}

// LsDecoder12.decode :94-160 for a batch in host memory.  Every container type is decoded on the GPU as stored: canonical
// Huffman (type 2), legacy Huffman of M32 (type 0, either header revision) and Deflate (type 1: the two zlib streams are
// inflated by k_inflate).  The host only moves bytes.
gf_status gf_lsop12_decode_batch_i32(gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                     const uint64_t *offsets, int32_t *values, int32_t *status)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || !blob || !offsets || !values) return GF_ERR_ARG;
    if (!offsetsValid(offsets, nTiles)) return GF_ERR_ARG;    // a bad array must not become an out-of-bounds read
    GF_HIP(hipSetDevice(c->device));
    if (nRows < 6 || nCols < 6) {
        for (size_t t = 0; t < nTiles && status; t++) status[t] = GF_ERR_BOUNDS;
        return GF_OK;
    }
    const size_t cells = (size_t)nRows * (size_t)nCols;
    const size_t nRes = gf_lsop12_residual_count(nRows, nCols), resStride = roundUp(nRes, 4);
    std::vector<uint32_t> lengths(nTiles);
    for (size_t t = 0; t < nTiles; t++) lengths[t] = (uint32_t)(offsets[t + 1] - offsets[t]);
    const uint8_t *gpuBlob = blob;
    const uint64_t *gpuOffsets = offsets;
    const uint64_t total = gpuOffsets[nTiles];

    gf_status s;
    if ((s = c->dBlob.ensure(total + 32)) != GF_OK) return s;
    if ((s = c->dValues.ensure(nTiles * cells * 4 + 16)) != GF_OK) return s;
    if ((s = c->dLengths.ensure(nTiles * 4 + 16)) != GF_OK) return s;
    if ((s = c->dStatus.ensure(nTiles * 4 + 16)) != GF_OK) return s;
    if ((s = c->dStatus2.ensure(nTiles * 4 + 16)) != GF_OK) return s;
    if ((s = c->dOffsets.ensure((nTiles + 1) * 8 + 16)) != GF_OK) return s;
    if ((s = c->dResiduals.ensure(nTiles * resStride * 4 + 16)) != GF_OK) return s;
    if ((s = c->dCoefs.ensure(nTiles * 64 + 16)) != GF_OK) return s;
    GF_HIP(hipMemcpyAsync(c->dBlob.p, gpuBlob, total, hipMemcpyHostToDevice, c->stream));
    GF_HIP(hipMemcpyAsync(c->dOffsets.p, gpuOffsets, (nTiles + 1) * 8, hipMemcpyHostToDevice, c->stream));
    GF_HIP(hipMemcpyAsync(c->dLengths.p, lengths.data(), nTiles * 4, hipMemcpyHostToDevice, c->stream));
    s = gf_lsop12_decode_batch_i32_dev(c, c->stream, nRows, nCols, nTiles, (const uint8_t *)c->dBlob.p, total,
                                       (const uint64_t *)c->dOffsets.p, 0, (const uint32_t *)c->dLengths.p, (int32_t *)c->dValues.p,
                                       (int32_t *)c->dStatus.p, (int32_t *)c->dResiduals.p, resStride, (uint32_t *)c->dCoefs.p,
                                       (int32_t *)c->dStatus2.p);
    if (s != GF_OK) return s;
    std::vector<int32_t> st(nTiles);
    GF_HIP(hipMemcpyAsync(values, c->dValues.p, nTiles * cells * 4, hipMemcpyDeviceToHost, c->stream));
    GF_HIP(hipMemcpyAsync(st.data(), c->dStatus.p, nTiles * 4, hipMemcpyDeviceToHost, c->stream));
    GF_HIP(hipStreamSynchronize(c->stream));
    if (status) memcpy(status, st.data(), nTiles * 4);
    return GF_OK;
}

gf_status gf_lsop12_encode_i32(gf_context *c, int codecIndex, int nRows, int nCols, const int32_t *values, int deflateEnabled,
                               uint8_t *out, size_t outCap, size_t *outLen)
{
    GF_CTX_LOCK(c);
    if (!outLen) return GF_ERR_ARG;
    uint64_t offsets[2] = {0, 0};
    int32_t st = 0;
    gf_status s = gf_lsop12_encode_batch_i32(c, codecIndex, nRows, nCols, 1, values, deflateEnabled, out, outCap, offsets,
                                             nullptr, &st);
    *outLen = (size_t)offsets[1];
    if (s != GF_OK) return s;
    return (gf_status)st;
}

gf_status gf_lsop12_decode_i32(gf_context *c, int nRows, int nCols, const uint8_t *packing, size_t len, int32_t *values)
{
    GF_CTX_LOCK(c);
    uint64_t offsets[2] = {0, (uint64_t)len};
    int32_t st = 0;
    gf_status s = gf_lsop12_decode_batch_i32(c, nRows, nCols, 1, packing, offsets, values, &st);
    if (s != GF_OK) return s;
    return (gf_status)st;
}


// ------------------------------------------------------------------ CodecDeflate (predictor + M32 on the GPU, Deflate on the host)

size_t gf_m32_default_stride(int nRows, int nCols)
{
    const size_t cells = (size_t)nRows * (size_t)nCols;
    return roundUp(cells + cells / 4 + 256, 16);
}

size_t gf_m32_max_stream(int nRows, int nCols) { return roundUp((size_t)6 * (size_t)nRows * (size_t)nCols + 32, 16); }

gf_status gf_m32_encode_batch_i32_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles, const int32_t *dValues,
                                      uint8_t *dStreams, size_t subStride, uint32_t *dLengths, uint8_t *dModels,
                                      uint32_t *dSeeds, int32_t *dStatus)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || !dValues || !dStreams || !dLengths || !dModels || !dSeeds || !dStatus) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));                        // launches and copies below go to the context's device
    if ((size_t)nRows * (size_t)nCols >= (1ull << 28)) return GF_ERR_UNSUPPORTED;
    if (subStride % 16 != 0 || subStride < 16 || ((uintptr_t)dStreams & 15) != 0) return GF_ERR_ARG;
    GfM32Args a;
    a.values = dValues;
    a.out = dStreams;
    a.subStride = subStride;
    a.lengths = dLengths;
    a.models = dModels;
    a.seeds = dSeeds;
    a.status = dStatus;
    a.nTiles = nTiles;
    a.nRows = nRows;
    a.nCols = nCols;
    GF_HIP(gf_launch_m32_streams(a, stream ? (hipStream_t)stream : c->stream));
    return GF_OK;
}

gf_status gf_m32_decode_batch_i32_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles, const uint8_t *dBlob,
                                      size_t blobBytes, const uint64_t *dOffsets, size_t slotStride, const uint32_t *dLengths,
                                      int32_t *dValues, int32_t *dStatus)
{
    GF_CTX_LOCK(c);
    return decodeBatchDev(KIND_RAW_M32, c, stream, nRows, nCols, nTiles, dBlob, blobBytes, dOffsets, slotStride, dLengths,
                          dValues, dStatus);
}

// CodecDeflate.encode :157-199 + compress :201-228 for a batch in host memory: the candidate M32 streams come from the GPU,
// java.util.zip.Deflater(6) is the host's zlib, the strictly shortest packing wins (earlier predictor on ties).
//
// The cost is zlib's: three candidate streams per tile at level 6 (tools/codec_master_rate.py: 60 MB/s of M32 bytes per core,
// 0.9 ms per 120x150 tile and core).  What can be done around it is done (round 3): the batch goes through in chunks, the GPU
// stage of chunk k + 1 (upload, k_m32_streams, download) overlapping the host threads' zlib of chunk k; a candidate is given
// up the moment its stream is longer than the shortest packing known for the tile -- the predictors' candidates among each
// other (Triangle first: it is the shortest on terrain, the ties of :195 are kept by comparing against the right side), and
// under gf_codec_master_encode_batch_i32 the packings of the list's other codecs (notLongerThan): a candidate that cannot win
// needs no bytes.  Results are byte-identical to running every stream to its end.
static gf_status deflateEncodeBatchHost(gf_context *c, int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                                        const uint32_t *notLongerThan,      // per tile or null: packings longer than this are of no use
                                        std::vector<std::vector<uint8_t>> &packs, std::vector<uint8_t> &chosen, std::vector<int32_t> &st)
{
    GF_HIP(hipSetDevice(c->device));
    const size_t cells = (size_t)nRows * (size_t)nCols;
    const size_t sub = gf_m32_default_stride(nRows, nCols), maxSub = gf_m32_max_stream(nRows, nCols);
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(nTiles, (size_t)(64u << 20) / (cells * 4)));
    gf_status s;
    if ((s = c->dValues.ensure(chunk * cells * 4 + 16)) != GF_OK) return s;
    if ((s = c->dM32.ensure(chunk * 3 * sub + 16)) != GF_OK) return s;
    if ((s = c->dM32Len.ensure(chunk * 12 + 16)) != GF_OK) return s;
    if ((s = c->dM32Models.ensure(chunk * 3 + 16)) != GF_OK) return s;
    if ((s = c->dSeeds.ensure(chunk * 4 + 16)) != GF_OK) return s;
    if ((s = c->dStatus.ensure(chunk * 4 + 16)) != GF_OK) return s;
    packs.assign(nTiles, {});
    chosen.assign(nTiles, 0);
    st.assign(nTiles, GF_OK);
    struct Stage {                                                       // what a chunk brings back from the GPU
        std::unique_ptr<uint8_t[]> streams;
        std::vector<uint8_t> models;
        std::vector<uint32_t> lens, seeds;
        std::vector<std::vector<uint8_t>> big;                           // tiles whose streams did not fit the default sub-slot
        size_t t0 = 0, n = 0;
    } stage[2];
    for (Stage &g : stage) g.streams.reset(new uint8_t[chunk * 3 * sub]);
    auto gpuStage = [&](Stage &g, size_t t0, size_t n) -> gf_status {
        g.t0 = t0;
        g.n = n;
        g.models.resize(n * 3);
        g.lens.resize(n * 3);
        g.seeds.resize(n);
        g.big.assign(n, {});
        GF_HIP(hipMemcpyAsync(c->dValues.p, values + t0 * cells, n * cells * 4, hipMemcpyHostToDevice, c->stream));
        gf_status r = gf_m32_encode_batch_i32_dev(c, c->stream, nRows, nCols, n, (const int32_t *)c->dValues.p, (uint8_t *)c->dM32.p, sub,
                                                  (uint32_t *)c->dM32Len.p, (uint8_t *)c->dM32Models.p, (uint32_t *)c->dSeeds.p,
                                                  (int32_t *)c->dStatus.p);
        if (r != GF_OK) return r;
        GF_HIP(hipMemcpyAsync(g.streams.get(), c->dM32.p, n * 3 * sub, hipMemcpyDeviceToHost, c->stream));
        GF_HIP(hipMemcpyAsync(g.lens.data(), c->dM32Len.p, n * 12, hipMemcpyDeviceToHost, c->stream));
        GF_HIP(hipMemcpyAsync(g.models.data(), c->dM32Models.p, n * 3, hipMemcpyDeviceToHost, c->stream));
        GF_HIP(hipMemcpyAsync(g.seeds.data(), c->dSeeds.p, n * 4, hipMemcpyDeviceToHost, c->stream));
        GF_HIP(hipMemcpyAsync(st.data() + t0, c->dStatus.p, n * 4, hipMemcpyDeviceToHost, c->stream));
        GF_HIP(hipStreamSynchronize(c->stream));
        // tiles with a stream longer than the default sub-slot: once more, one at a time, into worst-case slots
        for (size_t i = 0; i < n; i++) {
            if (st[t0 + i] != GF_OVERFLOW) continue;
            DevBuf slot, meta;
            if ((r = slot.ensure(3 * maxSub)) != GF_OK) return r;
            if ((r = meta.ensure(64)) != GF_OK) { slot.release(); return r; }
            uint8_t *m = (uint8_t *)meta.p;
            r = gf_m32_encode_batch_i32_dev(c, c->stream, nRows, nCols, 1, (const int32_t *)c->dValues.p + i * cells, (uint8_t *)slot.p,
                                            maxSub, (uint32_t *)m, m + 16, (uint32_t *)(m + 32), (int32_t *)(m + 48));
            g.big[i].resize(3 * maxSub);
            hipError_t e1 = hipSuccess, e2 = hipSuccess, e3 = hipSuccess;
            if (r == GF_OK) {
                e1 = hipMemcpyAsync(g.big[i].data(), slot.p, 3 * maxSub, hipMemcpyDeviceToHost, c->stream);
                e2 = hipMemcpyAsync(&st[t0 + i], m + 48, 4, hipMemcpyDeviceToHost, c->stream);
                e3 = hipStreamSynchronize(c->stream);
            }
            slot.release();
            meta.release();
            if (r != GF_OK) return r;
            if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return hipFail(e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3, "m32 overflow tile");
        }
        return GF_OK;
    };
    auto zlibStage = [&](const Stage &g) {
        parallelFor(g.n, [&](size_t i) {
            const size_t t = g.t0 + i;
            if (st[t] != GF_OK) return;
            const bool isBig = !g.big[i].empty();
            const size_t stride = isBig ? maxSub : sub;
            const uint8_t *base = isBig ? g.big[i].data() : g.streams.get() + i * 3 * sub;
            // packing length of the candidate kept so far per predictor slot (0: none); the winner is the FIRST shortest in the
            // order Differencing, Linear, Triangle (:195: a later one must be strictly shorter)
            size_t have[3] = {0, 0, 0};
            std::vector<uint8_t> z[3];
            static const int order[3] = {2, 0, 1};                           // Triangle first, then the reference's order
            for (int oi = 0; oi < 3; oi++) {
                const int p = order[oi];
                const uint32_t n = g.lens[i * 3 + p];
                const int model = g.models[i * 3 + p];
                if (model == 0 || n == 0) continue;                          // mCodeLength > 0 (:189)
                // longest packing this candidate is still of use with: against an EARLIER slot it must be strictly shorter,
                // against a LATER one no longer
                size_t cap = notLongerThan ? (size_t)notLongerThan[t] : ~(size_t)0;
                for (int q = 0; q < 3; q++)
                    if (have[q]) cap = std::min(cap, q < p ? have[q] - 1 : have[q]);
                if (cap < 11) continue;
                const size_t limit = std::min<size_t>(cap - 10, (size_t)n + 118);   // Deflater wrote into byte[nM32 + 128] from offset 10 (:204-205)
                if (!zDeflateUpTo(base + p * stride, n, 6, limit, z[p])) {
                    // given up: longer than the limit.  The reference's buffer cuts a stream at nM32 + 118 bytes; a candidate that
                    // long is kept at that length there -- run it to the end where that cut is the reason
                    if (limit == (size_t)n + 118 && zDeflate(base + p * stride, n, 6, z[p])) z[p].resize(limit);
                    else continue;
                }
                if (z[p].empty()) continue;
                have[p] = z[p].size() + 10;
            }
            int win = -1;
            for (int p = 0; p < 3; p++)
                if (have[p] && (win < 0 || have[p] < have[win])) win = p;
            if (win < 0) { st[t] = GF_DECLINED; return; }
            std::vector<uint8_t> &pk = packs[t];
            pk.resize(have[win]);
            pk[0] = (uint8_t)codecIndex;
            pk[1] = g.models[i * 3 + win];
            putLE32(&pk[2], g.seeds[i]);
            putLE32(&pk[6], g.lens[i * 3 + win]);
            memcpy(&pk[10], z[win].data(), z[win].size());
            chosen[t] = g.models[i * 3 + win];
        });
    };
    // chunk k's zlib runs on the host's threads while this thread drives the GPU stage of chunk k + 1
    // (joined by a guard: an exception on this thread -- a vector that cannot grow -- must not meet a joinable std::thread, which
    // would end the process; one on the worker thread is caught there and becomes a status)
    struct Joined {
        std::thread t;
        ~Joined() { if (t.joinable()) t.join(); }
    } worker;
    std::atomic<int> workerStatus{GF_OK};
    gf_status result = GF_OK;
    int cur = 0;
    try {
        for (size_t t0 = 0; t0 < nTiles && result == GF_OK; t0 += chunk) {
            const size_t n = std::min(chunk, nTiles - t0);
            result = gpuStage(stage[cur], t0, n);
            if (worker.t.joinable()) worker.t.join();
            if (result != GF_OK) break;
            Stage *g = &stage[cur];
            worker.t = std::thread([&, g]() {
                try {
                    zlibStage(*g);
                } catch (const std::bad_alloc &) {
                    workerStatus = GF_ERR_HIP;
                }
            });
            cur ^= 1;
        }
    } catch (const std::bad_alloc &) {
        result = GF_ERR_HIP;
    }
    if (worker.t.joinable()) worker.t.join();
    if (result == GF_OK && workerStatus != GF_OK) {
        g_lastError = "out of host memory in the Deflate stage";
        result = (gf_status)workerStatus.load();
    }
    return result;
}

gf_status gf_deflate_encode_batch_i32(gf_context *c, int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                                      uint8_t *blob, size_t blobCap, uint64_t *offsets, uint8_t *predictors, int32_t *status)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || !values || !offsets || (!blob && blobCap)) return GF_ERR_ARG;
    std::vector<std::vector<uint8_t>> packs;
    std::vector<uint8_t> chosen;
    std::vector<int32_t> st;
    gf_status s = deflateEncodeBatchHost(c, codecIndex, nRows, nCols, nTiles, values, nullptr, packs, chosen, st);
    if (s != GF_OK) return s;
    uint64_t total = 0;
    for (size_t t = 0; t < nTiles; t++) {
        offsets[t] = total;
        if (st[t] == GF_OK) total += packs[t].size();
    }
    offsets[nTiles] = total;
    if (status) memcpy(status, st.data(), nTiles * 4);
    if (predictors) memcpy(predictors, chosen.data(), nTiles);
    if (total > blobCap) return GF_ERR_CAPACITY;
    for (size_t t = 0; t < nTiles; t++)
        if (st[t] == GF_OK) memcpy(blob + offsets[t], packs[t].data(), packs[t].size());
    return GF_OK;
}

// CodecDeflate.decode :108-155: the zlib stream of every packing is inflated ON THE DEVICE (gvrs_inflate.hip), the M32 bytes
// go through the decode kernel's raw mode; the host only moves bytes (chunked, pinned staging).
gf_status gf_deflate_decode_batch_i32(gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                      const uint64_t *offsets, int32_t *values, int32_t *status)
{
    GF_CTX_LOCK(c);
    return decodeBatchHost(KIND_DEFLATE, c, nRows, nCols, nTiles, blob, offsets, values, status);
}

gf_status gf_deflate_decode_batch_i32_dev(gf_context *c, void *stream, int nRows, int nCols, size_t nTiles, const uint8_t *dBlob,
                                          size_t blobBytes, const uint64_t *dOffsets, size_t slotStride, const uint32_t *dLengths,
                                          int32_t *dValues, int32_t *dStatus)
{
    GF_CTX_LOCK(c);
    if (!c || nRows < 1 || nCols < 1 || !dBlob || !dLengths || !dValues || !dStatus) return GF_ERR_ARG;
    GF_HIP(hipSetDevice(c->device));
    return deflateDecodeDev(c, stream ? (hipStream_t)stream : c->stream, nRows, nCols, nTiles, dBlob, blobBytes, dOffsets, slotStride,
                            dLengths, dValues, dStatus);
}

gf_status gf_deflate_encode_i32(gf_context *c, int codecIndex, int nRows, int nCols, const int32_t *values, uint8_t *out,
                                size_t outCap, size_t *outLen)
{
    GF_CTX_LOCK(c);
    if (!outLen) return GF_ERR_ARG;
    uint64_t offsets[2] = {0, 0};
    int32_t st = 0;
    gf_status s = gf_deflate_encode_batch_i32(c, codecIndex, nRows, nCols, 1, values, out, outCap, offsets, nullptr, &st);
    *outLen = (size_t)offsets[1];
    if (s != GF_OK) return s;
    return (gf_status)st;
}

gf_status gf_deflate_decode_i32(gf_context *c, int nRows, int nCols, const uint8_t *packing, size_t len, int32_t *values)
{
    GF_CTX_LOCK(c);
    uint64_t offsets[2] = {0, (uint64_t)len};
    int32_t st = 0;
    gf_status s = gf_deflate_decode_batch_i32(c, nRows, nCols, 1, packing, offsets, values, &st);
    if (s != GF_OK) return s;
    return (gf_status)st;
}


// ------------------------------------------------------------------ CodecMaster: the shortest packing over a codec list

// gvrs/CodecMaster.java:150-169 for a batch in host memory.  codecs[k] = GF_CODEC_* of the k-th entry of the file's codec list
// (GF_CODEC_NONE for an entry that has no integer encoder, e.g. CodecFloat); k is the codec index written to packing[0].
// Every integer codec encodes the batch; per tile the strictly shortest non-null packing wins, list order breaks ties.
gf_status gf_codec_master_encode_batch_i32(gf_context *c, const int *codecs, int nCodecs, int nRows, int nCols, size_t nTiles,
                                           const int32_t *values, uint8_t *blob, size_t blobCap, uint64_t *offsets,
                                           uint8_t *codecUsed, int32_t *status)
{
    GF_CTX_LOCK(c);
    if (!c || !codecs || nCodecs < 1 || nCodecs > 255 || !values || !offsets || (!blob && blobCap)) return GF_ERR_ARG;
    const size_t cells = (size_t)nRows * (size_t)nCols;
    for (int k = 0; k < nCodecs; k++)
        if (codecs[k] < GF_CODEC_NONE || codecs[k] > GF_CODEC_LSOP12) return GF_ERR_ARG;
    // per codec of the list: its packings (Deflate: one vector per tile; the others: one blob + offsets) and statuses
    struct Entry {
        std::unique_ptr<uint8_t[]> blob;
        std::vector<uint64_t> off;
        std::vector<std::vector<uint8_t>> packs;
        std::vector<int32_t> st;
        bool ran = false;
    };
    std::vector<Entry> e(nCodecs);
    auto lenOf = [&](int k, size_t t) -> size_t {
        const Entry &x = e[k];
        if (!x.ran || x.st[t] != GF_OK) return 0;
        return codecs[k] == GF_CODEC_DEFLATE ? x.packs[t].size() : (size_t)(x.off[t + 1] - x.off[t]);
    };
    // The GPU codecs first, CodecDeflate last: its three zlib streams per tile are the expensive part of the list, and a stream
    // that is already longer than what another codec of the list made of the tile cannot win (:161-164: the shortest packing
    // wins, the earlier codec on ties) -- deflateEncodeBatchHost gives such candidates up early.
    for (int pass = 0; pass < 2; pass++) {
        for (int k = 0; k < nCodecs; k++) {
            const int kind = codecs[k];
            if (kind == GF_CODEC_NONE || (kind == GF_CODEC_DEFLATE) != (pass == 1)) continue;
            Entry &x = e[k];
            x.st.assign(nTiles, GF_OK);
            gf_status s = GF_OK;
            if (kind == GF_CODEC_DEFLATE) {
                // what a Deflate packing may measure at most to be of use: strictly less than the codecs before it, no more than those behind
                std::vector<uint32_t> bound(nTiles, 0xFFFFFFFFu);
                for (size_t t = 0; t < nTiles; t++)
                    for (int j = 0; j < nCodecs; j++) {
                        const size_t len = j == k ? 0 : lenOf(j, t);
                        if (len) bound[t] = (uint32_t)std::min<size_t>(bound[t], j < k ? len - 1 : len);
                    }
                std::vector<uint8_t> chosen;
                s = deflateEncodeBatchHost(c, k, nRows, nCols, nTiles, values, bound.data(), x.packs, chosen, x.st);
            } else {
                x.off.assign(nTiles + 1, 0);
                size_t cap = nTiles * (cells + 1024) + 64;                       // a byte per cell holds terrain packings; grown once if not
                for (int attempt = 0; attempt < 2; attempt++) {
                    x.blob.reset(new uint8_t[cap]);
                    switch (kind) {
                    case GF_CODEC_HUFFMAN: s = gf_huffman_encode_batch_i32(c, k, nRows, nCols, nTiles, values, x.blob.get(), cap, x.off.data(), nullptr, x.st.data()); break;
                    case GF_CODEC_CANON_HUFFMAN: s = gf_canon_encode_batch_i32(c, k, nRows, nCols, nTiles, values, x.blob.get(), cap, x.off.data(), nullptr, x.st.data()); break;
                    default: s = gf_lsop12_encode_batch_i32(c, k, nRows, nCols, nTiles, values, 1, x.blob.get(), cap, x.off.data(), nullptr, x.st.data()); break;
                    }
                    if (s != GF_ERR_CAPACITY) break;
                    cap = (size_t)x.off[nTiles] + 64;
                }
            }
            if (s != GF_OK) return s;
            x.ran = true;
        }
    }
    // per tile the first shortest packing in list order; a tile no codec packed reports the first encoder error (the Java
    // encoder threw) or "declined"
    std::vector<int32_t> bestSt(nTiles, GF_DECLINED);
    std::vector<uint8_t> used(nTiles, 0xff);
    uint64_t total = 0;
    for (size_t t = 0; t < nTiles; t++) {
        size_t bestLen = 0;
        for (int k = 0; k < nCodecs; k++) {
            if (!e[k].ran) continue;
            const int32_t stK = e[k].st[t];
            if (stK < 0) { if (used[t] == 0xff && bestSt[t] >= 0) bestSt[t] = stK; continue; }
            const size_t len = lenOf(k, t);
            if (len && (used[t] == 0xff || len < bestLen)) {                     // strictly shorter (:161-164)
                bestLen = len;
                used[t] = (uint8_t)k;
                bestSt[t] = GF_OK;
            }
        }
        offsets[t] = total;
        total += bestLen;
    }
    offsets[nTiles] = total;
    if (status) memcpy(status, bestSt.data(), nTiles * 4);
    if (codecUsed) memcpy(codecUsed, used.data(), nTiles);
    if (total > blobCap) return GF_ERR_CAPACITY;
    parallelFor(nTiles, [&](size_t t) {
        if (bestSt[t] != GF_OK) return;
        const int k = used[t];
        const uint8_t *src = codecs[k] == GF_CODEC_DEFLATE ? e[k].packs[t].data() : e[k].blob.get() + e[k].off[t];
        memcpy(blob + offsets[t], src, (size_t)(offsets[t + 1] - offsets[t]));
    });
    return GF_OK;
}

// gvrs/CodecMaster.java:195-203: dispatch on packing[0]
// CodecMaster.decode (gvrs/CodecMaster.java:195-203) for packings anywhere inside `blob`: packing t is lens[t] bytes at
// blob + starts[t]; tiles with skip[t] != 0 are left alone (raw elements, records that failed their checks).  The packings are
// sorted by the codec their first byte names and every codec's share goes through its batch decoder in the scattered form
// (decodeBatchHostG): nothing is copied on the host but the packings themselves, into the pinned staging buffers, and the
// decoded tiles from there to their place.  (Round 3 built a fresh blob per codec with vector::insert per tile, decoded into a
// temporary array and copied every tile back, single-threaded: 2.9 GB/s on top of a 40 GB/s decoder.)
static gf_status codecMasterDecodeScattered(gf_context *c, const int *codecs, int nCodecs, int nRows, int nCols, size_t nTiles,
                                            const uint8_t *blob, const uint64_t *starts, const uint32_t *lens, const uint8_t *skip,
                                            int32_t *values, int32_t *st)
{
    const size_t cells = (size_t)nRows * (size_t)nCols;
    std::vector<uint32_t> count(256, 0);
    for (size_t t = 0; t < nTiles; t++) {
        if (skip && skip[t]) continue;
        const int k = lens[t] ? (int)blob[starts[t]] : -1;
        if (k < 0 || k >= nCodecs || codecs[k] == GF_CODEC_NONE) { st[t] = GF_ERR_FORMAT; continue; }   // no such codec in the list
        count[k]++;
    }
    for (int k = 0; k < nCodecs && k < 256; k++) {
        if (!count[k]) continue;
        std::vector<uint64_t> ks(count[k]);
        std::vector<uint32_t> kl(count[k]), kd(count[k]);
        size_t i = 0;
        for (size_t t = 0; t < nTiles; t++) {
            if ((skip && skip[t]) || !lens[t] || blob[starts[t]] != (uint8_t)k) continue;
            ks[i] = starts[t];
            kl[i] = lens[t];
            kd[i] = (uint32_t)t;
            i++;
        }
        gf_status s;
        switch (codecs[k]) {
        case GF_CODEC_HUFFMAN: s = decodeBatchHostG(KIND_HUFFMAN, c, nRows, nCols, i, blob, nullptr, ks.data(), kl.data(), kd.data(), values, st); break;
        case GF_CODEC_DEFLATE: s = decodeBatchHostG(KIND_DEFLATE, c, nRows, nCols, i, blob, nullptr, ks.data(), kl.data(), kd.data(), values, st); break;
        case GF_CODEC_CANON_HUFFMAN: s = decodeBatchHostG(KIND_CANON, c, nRows, nCols, i, blob, nullptr, ks.data(), kl.data(), kd.data(), values, st); break;
        case GF_CODEC_LSOP12: {
            // (LSOP12's host path has stages of its own: its packings are gathered into one blob first, in parallel)
            std::vector<uint64_t> off(i + 1, 0);
            for (size_t j = 0; j < i; j++) off[j + 1] = off[j] + kl[j];
            std::vector<uint8_t> sub((size_t)off[i] + 16);
            parallelFor(i, [&](size_t j) { memcpy(sub.data() + off[j], blob + ks[j], kl[j]); });
            std::vector<int32_t> out(i * cells), sst(i);
            s = gf_lsop12_decode_batch_i32(c, nRows, nCols, i, sub.data(), off.data(), out.data(), sst.data());
            if (s != GF_OK) return s;
            parallelFor(i, [&](size_t j) {
                st[kd[j]] = sst[j];
                if (sst[j] == GF_OK) memcpy(values + (size_t)kd[j] * cells, out.data() + j * cells, cells * 4);
            });
            break;
        }
        default: return GF_ERR_ARG;
        }
        if (s != GF_OK) return s;
    }
    return GF_OK;
}

gf_status gf_codec_master_decode_batch_i32(gf_context *c, const int *codecs, int nCodecs, int nRows, int nCols, size_t nTiles,
                                           const uint8_t *blob, const uint64_t *offsets, int32_t *values, int32_t *status)
{
    GF_CTX_LOCK(c);
    if (!c || !codecs || nCodecs < 1 || !blob || !offsets || !values) return GF_ERR_ARG;
    if (!offsetsValid(offsets, nTiles)) return GF_ERR_ARG;    // a bad array must not become an out-of-bounds read
    std::vector<int32_t> st(nTiles, GF_ERR_FORMAT);
    std::vector<uint32_t> lens(nTiles);
    for (size_t t = 0; t < nTiles; t++) lens[t] = (uint32_t)(offsets[t + 1] - offsets[t]);
    const gf_status s = codecMasterDecodeScattered(c, codecs, nCodecs, nRows, nCols, nTiles, blob, offsets, lens.data(), nullptr, values,
                                                   st.data());
    if (s != GF_OK) return s;
    if (status) memcpy(status, st.data(), nTiles * 4);
    return GF_OK;
}


// ------------------------------------------------------------------ tile payloads (one integer element per tile)

// RasterTile.getCompressedPacking (gvrs/RasterTile.java:234-256) over TileElementInt.encode (gvrs/TileElementInt.java:196-207)
// for a batch: per tile [int32 LE n][n bytes], the bytes being the CodecMaster packing, or the raw little-endian cells when
// no codec produced one or it is not shorter than them.  What RecordManager.writeTile stores behind the tile index.
gf_status gf_tile_payload_encode_batch_i32(gf_context *c, const int *codecs, int nCodecs, int nRows, int nCols, size_t nTiles,
                                           const int32_t *values, uint8_t *blob, size_t blobCap, uint64_t *offsets,
                                           uint8_t *codecUsed)
{
    GF_CTX_LOCK(c);
    if (!c || !values || !offsets || (!blob && blobCap)) return GF_ERR_ARG;
    const size_t cells = (size_t)nRows * (size_t)nCols, rawBytes = cells * 4;
    std::vector<uint8_t> packs(nTiles * (rawBytes + 1024) + 64);
    std::vector<uint64_t> off(nTiles + 1);
    std::vector<int32_t> st(nTiles);
    std::vector<uint8_t> used(nTiles);
    gf_status s = gf_codec_master_encode_batch_i32(c, codecs, nCodecs, nRows, nCols, nTiles, values, packs.data(), packs.size(),
                                                   off.data(), used.data(), st.data());
    if (s == GF_ERR_CAPACITY) {
        packs.resize((size_t)off[nTiles] + 64);
        s = gf_codec_master_encode_batch_i32(c, codecs, nCodecs, nRows, nCols, nTiles, values, packs.data(), packs.size(), off.data(),
                                             used.data(), st.data());
    }
    if (s != GF_OK) return s;
    uint64_t total = 0;
    for (size_t t = 0; t < nTiles; t++) {
        if (st[t] < 0) return (gf_status)st[t];                          // an encoder threw: the Java call fails as a whole
        const size_t n = st[t] == GF_OK ? (size_t)(off[t + 1] - off[t]) : 0;
        const bool raw = st[t] != GF_OK || n >= rawBytes;
        offsets[t] = total;
        total += 4 + (raw ? rawBytes : n);
        if (raw) used[t] = 0xff;
    }
    offsets[nTiles] = total;
    if (codecUsed) memcpy(codecUsed, used.data(), nTiles);
    if (total > blobCap) return GF_ERR_CAPACITY;
    for (size_t t = 0; t < nTiles; t++) {
        uint8_t *p = blob + offsets[t];
        const size_t n = (size_t)(offsets[t + 1] - offsets[t]) - 4;
        putLE32(p, (uint32_t)n);
        if (used[t] == 0xff) memcpy(p + 4, values + t * cells, rawBytes);   // little-endian host == the file's byte order
        else memcpy(p + 4, packs.data() + off[t], n);
    }
    return GF_OK;
}

// TileElementInt.decode (gvrs/TileElementInt.java:209-219): an encoding of exactly 4*cells bytes is the raw cells
gf_status gf_tile_payload_decode_batch_i32(gf_context *c, const int *codecs, int nCodecs, int nRows, int nCols, size_t nTiles,
                                           const uint8_t *blob, const uint64_t *offsets, int32_t *values, int32_t *status)
{
    GF_CTX_LOCK(c);
    if (!c || !blob || !offsets || !values) return GF_ERR_ARG;
    if (!offsetsValid(offsets, nTiles)) return GF_ERR_ARG;    // a bad array must not become an out-of-bounds read
    const size_t cells = (size_t)nRows * (size_t)nCols, rawBytes = cells * 4;
    std::vector<uint64_t> starts(nTiles, 0);
    std::vector<uint32_t> lens(nTiles, 0);
    std::vector<int32_t> st(nTiles, GF_OK);
    std::vector<uint8_t> skip(nTiles, 0);
    bool anyPacked = false;
    for (size_t t = 0; t < nTiles; t++) {
        const size_t len = (size_t)(offsets[t + 1] - offsets[t]);
        skip[t] = 1;
        if (len < 4) { st[t] = GF_ERR_BOUNDS; continue; }
        const size_t n = getLE32(blob + offsets[t]);
        if (n + 4 > len) { st[t] = GF_ERR_BOUNDS; continue; }
        starts[t] = offsets[t] + 4;
        lens[t] = (uint32_t)n;
        if (n == rawBytes) { skip[t] = 2; continue; }          // the cells themselves (copied below)
        skip[t] = 0;
        anyPacked = true;
    }
    parallelFor(nTiles, [&](size_t t) {
        if (skip[t] == 2) memcpy(values + t * cells, blob + starts[t], rawBytes);
    });
    if (anyPacked) {
        if (!codecs || nCodecs < 1) return GF_ERR_ARG;
        const gf_status s = codecMasterDecodeScattered(c, codecs, nCodecs, nRows, nCols, nTiles, blob, starts.data(), lens.data(), skip.data(),
                                                       values, st.data());
        if (s != GF_OK) return s;
    }
    if (status) memcpy(status, st.data(), nTiles * 4);
    return GF_OK;
}

// ------------------------------------------------------------------ tile records (RecordManager framing)

static uint32_t crc32cTable[256];
static std::once_flag crc32cOnce;

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) as util/GridfourCRC32C.java:330-338 applies it.  The host's crc32
// instruction (SSE 4.2: eight bytes per step) where there is one, the byte-at-a-time table otherwise (round 3: the table loop
// alone, ~1 byte per cycle over every record of a batch).
#if defined(__x86_64__)
#define GF_HOST_HAS_CRC32_INSN 1
__attribute__((target("sse4.2"))) static uint32_t crc32cHw(const uint8_t *data, size_t n)
{
    uint64_t crc = 0xffffffffu;
    while (n && ((uintptr_t)data & 7)) { crc = __builtin_ia32_crc32qi((uint32_t)crc, *data++); n--; }
    for (; n >= 8; n -= 8, data += 8) {
        uint64_t w;
        memcpy(&w, data, 8);
        crc = __builtin_ia32_crc32di(crc, w);
    }
    while (n--) crc = __builtin_ia32_crc32qi((uint32_t)crc, *data++);
    return (uint32_t)crc ^ 0xffffffffu;
}
#endif

uint32_t gf_crc32c(const uint8_t *data, size_t n)
{
#ifdef GF_HOST_HAS_CRC32_INSN
    static const bool hw = __builtin_cpu_supports("sse4.2");
    if (hw) return crc32cHw(data, n);
#endif
    std::call_once(crc32cOnce, []() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t x = i;
            for (int k = 0; k < 8; k++) x = (x >> 1) ^ ((x & 1u) ? 0x82F63B78u : 0u);
            crc32cTable[i] = x;
        }
    });
    uint32_t crc = 0xffffffffu;
    for (size_t i = 0; i < n; i++) crc = crc32cTable[(crc ^ data[i]) & 0xffu] ^ (crc >> 8);
    return crc ^ 0xffffffffu;
}

static size_t elemStandardSize(int elemType, size_t cells)
{
    // TileElement.java:86-93: bytes per sample * cells, rounded up to a multiple of 4
    return elemType == GF_ELEM_SHORT ? ((cells * 2 + 3) & ~(size_t)3) : cells * 4;
}

size_t gf_tile_record_max_bytes(int elemType, int nRows, int nCols)
{
    const size_t content = 8 + elemStandardSize(elemType, (size_t)nRows * (size_t)nCols);
    return (content + 12 + 7) & ~(size_t)7;
}

// RecordManager.writeTile (gvrs/RecordManager.java:386-470) for tiles of one integer-coded element, each record as
// fileSpaceAlloc / fileSpaceInitRecord / fileSpaceFinishRecord lay it out when the file is extended (:153-204, 217-262):
//   [int32 size, multiple of 8][type 2][0 0 0][int32 tileIndex][int32 n][n bytes][zero padding][CRC-32C of all before | 0]
// The element bytes are the CodecMaster packing or, when no codec helps, the standard form (TileElementInt.java:196-207,
// TileElementShort.java:211-229: shorts go to the codecs as ints with the fill value mapped to INT4_NULL_CODE).
gf_status gf_tile_record_encode_batch(gf_context *c, const int *codecs, int nCodecs, int elemType, int fillValue, int nRows,
                                      int nCols, size_t nTiles, const int32_t *tileIndices, const void *values,
                                      int checksumEnabled, uint8_t *blob, size_t blobCap, uint64_t *offsets, uint8_t *codecUsed)
{
    GF_CTX_LOCK(c);
    if (!c || !values || !offsets || !tileIndices || (!blob && blobCap)) return GF_ERR_ARG;
    if (elemType != GF_ELEM_INT && elemType != GF_ELEM_SHORT) return GF_ERR_ARG;
    if (nRows < 1 || nCols < 1) return GF_ERR_ARG;
    const size_t cells = (size_t)nRows * (size_t)nCols, stdSize = elemStandardSize(elemType, cells);
    const int32_t *iv = (const int32_t *)values;
    std::vector<int32_t> widened;
    if (elemType == GF_ELEM_SHORT) {
        const int16_t *sv = (const int16_t *)values;
        widened.resize(nTiles * cells);
        parallelFor(nTiles, [&](size_t t) {
            for (size_t i = t * cells; i < (t + 1) * cells; i++)
                widened[i] = sv[i] == (int16_t)fillValue ? (int32_t)0x80000000 : (int32_t)sv[i];
        });
        iv = widened.data();
    }
    std::vector<uint8_t> packs;
    std::vector<uint64_t> off(nTiles + 1, 0);
    std::vector<int32_t> st(nTiles, GF_DECLINED);
    std::vector<uint8_t> used(nTiles, 0xff);
    if (nCodecs > 0 && nTiles > 0) {                                   // data compression enabled (:417)
        packs.resize(nTiles * (cells * 4 + 1024) + 64);
        gf_status s = gf_codec_master_encode_batch_i32(c, codecs, nCodecs, nRows, nCols, nTiles, iv, packs.data(), packs.size(),
                                                       off.data(), used.data(), st.data());
        if (s == GF_ERR_CAPACITY) {
            packs.resize((size_t)off[nTiles] + 64);
            s = gf_codec_master_encode_batch_i32(c, codecs, nCodecs, nRows, nCols, nTiles, iv, packs.data(), packs.size(),
                                                 off.data(), used.data(), st.data());
        }
        if (s != GF_OK) return s;
    }
    uint64_t total = 0;
    std::vector<uint32_t> elemLen(nTiles);
    for (size_t t = 0; t < nTiles; t++) {
        if (st[t] < 0) return (gf_status)st[t];                          // an encoder threw: the Java call fails as a whole
        const size_t n = st[t] == GF_OK ? (size_t)(off[t + 1] - off[t]) : 0;
        const bool raw = st[t] != GF_OK || n >= stdSize;
        if (raw) used[t] = 0xff;
        elemLen[t] = (uint32_t)(raw ? stdSize : n);
        offsets[t] = total;
        total += (8 + elemLen[t] + 12 + 7) & ~(uint64_t)7;              // multipleOf8(content + RECORD_OVERHEAD_SIZE)
    }
    offsets[nTiles] = total;
    if (codecUsed) memcpy(codecUsed, used.data(), nTiles);
    if (total > blobCap) return GF_ERR_CAPACITY;
    parallelFor(nTiles, [&](size_t t) {
        uint8_t *r = blob + offsets[t];
        const size_t size = (size_t)(offsets[t + 1] - offsets[t]);
        memset(r, 0, size);
        putLE32(r, (uint32_t)size);
        r[4] = 2;                                                      // RecordType.Tile
        putLE32(r + 8, (uint32_t)tileIndices[t]);
        putLE32(r + 12, elemLen[t]);
        if (used[t] != 0xff) memcpy(r + 16, packs.data() + off[t], elemLen[t]);
        else if (elemType == GF_ELEM_SHORT) memcpy(r + 16, (const int16_t *)values + t * cells, cells * 2);   // little-endian host
        else memcpy(r + 16, (const int32_t *)values + t * cells, cells * 4);
        if (checksumEnabled) putLE32(r + size - 4, gf_crc32c(r, size - 4));
    });
    return GF_OK;
}

// RecordManager.readTile (gvrs/RecordManager.java:472-520) + TileElementInt.decode / TileElementShort.decode for a batch of
// tile records: status[t] = GF_OK, GF_ERR_FORMAT (not a tile record, checksum mismatch when verifyChecksum != 0, a packing
// the codecs reject) or GF_ERR_BOUNDS (lengths that do not fit the record).
gf_status gf_tile_record_decode_batch(gf_context *c, const int *codecs, int nCodecs, int elemType, int nRows, int nCols,
                                      size_t nTiles, const uint8_t *blob, const uint64_t *offsets, int verifyChecksum,
                                      int32_t *tileIndices, void *values, int32_t *status)
{
    GF_CTX_LOCK(c);
    if (!c || !blob || !offsets || !values) return GF_ERR_ARG;
    if (!offsetsValid(offsets, nTiles)) return GF_ERR_ARG;    // a bad array must not become an out-of-bounds read
    if (elemType != GF_ELEM_INT && elemType != GF_ELEM_SHORT) return GF_ERR_ARG;
    if (nRows < 1 || nCols < 1) return GF_ERR_ARG;
    const size_t cells = (size_t)nRows * (size_t)nCols, stdSize = elemStandardSize(elemType, cells);
    std::vector<int32_t> st(nTiles, GF_OK);
    std::vector<uint8_t> skip(nTiles, 1);                             // 0: a packing to decode, 2: the standard form, 1: failed
    std::vector<uint64_t> starts(nTiles, 0);
    std::vector<uint32_t> lens(nTiles, 0);
    std::atomic<int> anyPacked{0};
    // the framing of every record (RecordManager.java:456-459, RasterTile.java:243-253) and its checksum, a record per turn
    parallelFor(nTiles, [&](size_t t) {
        const uint8_t *r = blob + offsets[t];
        const size_t len = (size_t)(offsets[t + 1] - offsets[t]);
        if (len < 20) { st[t] = GF_ERR_BOUNDS; return; }
        const size_t size = getLE32(r);
        if (size > len || size < 20 || (size & 7)) { st[t] = GF_ERR_BOUNDS; return; }
        if (r[4] != 2) { st[t] = GF_ERR_FORMAT; return; }
        if (tileIndices) tileIndices[t] = (int32_t)getLE32(r + 8);
        const size_t n = getLE32(r + 12);
        if (16 + n > size) { st[t] = GF_ERR_BOUNDS; return; }
        if (verifyChecksum && getLE32(r + size - 4) != gf_crc32c(r, size - 4)) { st[t] = GF_ERR_FORMAT; return; }
        starts[t] = offsets[t] + 16;
        lens[t] = (uint32_t)n;
        if (n == stdSize) skip[t] = 2;
        else { skip[t] = 0; anyPacked = 1; }
    });
    std::vector<int32_t> wide;                                        // short elements: the codecs' int32 cells before narrowing
    int32_t *decoded = (int32_t *)values;
    if (anyPacked && nCodecs < 1) {
        for (size_t t = 0; t < nTiles; t++)
            if (skip[t] == 0) { st[t] = GF_ERR_FORMAT; skip[t] = 1; }     // a packing in a file without codecs
    } else if (anyPacked) {
        if (elemType == GF_ELEM_SHORT) {
            wide.resize(nTiles * cells);
            decoded = wide.data();
        }
        const gf_status s = codecMasterDecodeScattered(c, codecs, nCodecs, nRows, nCols, nTiles, blob, starts.data(), lens.data(), skip.data(),
                                                       decoded, st.data());
        if (s != GF_OK) return s;
    }
    parallelFor(nTiles, [&](size_t t) {
        if (st[t] != GF_OK || skip[t] == 1) return;
        const uint8_t *e = blob + starts[t];
        if (elemType == GF_ELEM_SHORT) {
            int16_t *o = (int16_t *)values + t * cells;
            if (skip[t] == 2) { memcpy(o, e, cells * 2); return; }
            const int32_t *d = wide.data() + t * cells;                 // TileElementShort.java:239-246
            for (size_t i = 0; i < cells; i++) o[i] = d[i] == (int32_t)0x80000000 ? (int16_t)-32768 : (int16_t)d[i];
        } else if (skip[t] == 2) {
            memcpy((int32_t *)values + t * cells, e, cells * 4);        // (packed int tiles were decoded in place)
        }
    });
    if (status) memcpy(status, st.data(), nTiles * 4);
    return GF_OK;
}

// ------------------------------------------------------------------ CodecHuffman.analyze

// ICompressionDecoder.analyze for a batch of CodecHuffman packings (compress/CodecHuffman.java:172-199): the packings are
// Huffman-decoded on the GPU, which returns per tile the predictor, the M32 byte count, the bits of the serialised tree and
// the 256-bin histogram of the M32 bytes; the sums of CodecStats.addToCounts / addCountsForM32 (compress/CodecStats.java:
// 100-141) are then accumulated here in tile order.  stats[p], p = 0..4 by predictor code (PredictorModelType ordinal),
// stats[5] = "All Predictors"; counts ADD to what stats already holds (clearAnalysisData = zero the array).  The pair
// counts behind CodecStats.getH2 (sA / sB) come from the same pass when the caller hands in tables for them.
static gf_status analyzeBatch(gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob, const uint64_t *offsets,
                              gf_codec_stats *stats, int64_t *pairCounts, int32_t *status)
{
    if (!c || nRows < 1 || nCols < 1 || !blob || !offsets || !stats) return GF_ERR_ARG;
    if (!offsetsValid(offsets, nTiles)) return GF_ERR_ARG;    // a bad array must not become an out-of-bounds read
    GF_HIP(hipSetDevice(c->device));
    const uint64_t total = offsets[nTiles];
    gf_status s;
    if ((s = c->dBlob.ensure(total + 32)) != GF_OK) return s;
    if ((s = c->dLengths.ensure(nTiles * 4 + 16)) != GF_OK) return s;
    if ((s = c->dStatus.ensure(nTiles * 4 + 16)) != GF_OK) return s;
    if ((s = c->dOffsets.ensure((nTiles + 1) * 8 + 16)) != GF_OK) return s;
    if ((s = c->dResiduals.ensure(nTiles * GF_ANALYSIS_WORDS * 4 + 16)) != GF_OK) return s;
    if ((s = c->dValues.ensure(16)) != GF_OK) return s;
    uint32_t *dPairs = nullptr;
    const size_t pairWords = (size_t)GF_PAIR_TABLES * 65536;
    if (pairCounts) {
        if ((s = c->dCoefs.ensure(pairWords * 4)) != GF_OK) return s;
        dPairs = (uint32_t *)c->dCoefs.p;
        GF_HIP(hipMemsetAsync(dPairs, 0, pairWords * 4, c->stream));
    }
    std::vector<uint32_t> lengths(nTiles);
    for (size_t t = 0; t < nTiles; t++) {
        if (offsets[t + 1] < offsets[t]) return GF_ERR_ARG;
        lengths[t] = (uint32_t)(offsets[t + 1] - offsets[t]);
    }
    GF_HIP(hipMemcpyAsync(c->dBlob.p, blob, total, hipMemcpyHostToDevice, c->stream));
    GF_HIP(hipMemcpyAsync(c->dOffsets.p, offsets, (nTiles + 1) * 8, hipMemcpyHostToDevice, c->stream));
    GF_HIP(hipMemcpyAsync(c->dLengths.p, lengths.data(), nTiles * 4, hipMemcpyHostToDevice, c->stream));
    s = decodeBatchDev(KIND_HUFFMAN, c, c->stream, nRows, nCols, nTiles, (const uint8_t *)c->dBlob.p, total,
                       (const uint64_t *)c->dOffsets.p, 0, (const uint32_t *)c->dLengths.p, (int32_t *)c->dValues.p,
                       (int32_t *)c->dStatus.p, (uint32_t *)c->dResiduals.p, dPairs);
    if (s != GF_OK) return s;
    std::vector<uint32_t> pairs;
    if (pairCounts) {
        pairs.resize(pairWords);
        GF_HIP(hipMemcpyAsync(pairs.data(), dPairs, pairWords * 4, hipMemcpyDeviceToHost, c->stream));
    }
    std::vector<uint32_t> rec(nTiles * GF_ANALYSIS_WORDS);
    std::vector<int32_t> st(nTiles);
    GF_HIP(hipMemcpyAsync(rec.data(), c->dResiduals.p, rec.size() * 4, hipMemcpyDeviceToHost, c->stream));
    GF_HIP(hipMemcpyAsync(st.data(), c->dStatus.p, nTiles * 4, hipMemcpyDeviceToHost, c->stream));
    GF_HIP(hipStreamSynchronize(c->stream));
    const double LOG2 = std::log(2.0);
    const int64_t nValues = (int64_t)nRows * nCols;
    for (size_t t = 0; t < nTiles; t++) {
        if (status) status[t] = st[t];
        if (st[t] != GF_OK) continue;                                   // analyze throws: nothing is counted
        const uint32_t *r = rec.data() + t * GF_ANALYSIS_WORDS;
        const uint32_t nM32 = r[1];
        int64_t observed = 0;
        double e = 0;
        if (nM32 > 0) {
            const double d = (double)nM32;
            for (int i = 0; i < 256; i++) {
                if (r[4 + i] > 0) {
                    observed++;
                    const double p = r[4 + i] / d;
                    e += p * std::log(p) / LOG2;
                }
            }
        }
        gf_codec_stats *two[2] = {&stats[r[0] <= 4 ? r[0] : 0], &stats[5]};
        for (gf_codec_stats *g : two) {
            g->n_tiles++;
            g->n_bytes += r[3];
            g->n_symbols += nValues;
            g->n_bits_overhead += r[2];
            if (nM32 > 0) {
                g->n_m32_counted++;
                g->sum_length_m32 += nM32;
                g->sum_observed_m32 += observed;
                g->sum_entropy_m32 -= e;
            }
        }
    }
    if (pairCounts) {
        // sB of the predictor's CodecStats and of "All Predictors" (CodecHuffman.java:186-196 feeds both)
        for (int m = 0; m < GF_PAIR_TABLES; m++)
            for (size_t i = 0; i < 65536; i++) {
                const int64_t n = pairs[(size_t)m * 65536 + i];
                pairCounts[(size_t)m * 65536 + i] += n;
                pairCounts[(size_t)5 * 65536 + i] += n;
            }
    }
    return GF_OK;
}

gf_status gf_huffman_analyze_batch(gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob, const uint64_t *offsets,
                                   gf_codec_stats *stats, int32_t *status)
{
    GF_CTX_LOCK(c);
    return analyzeBatch(c, nRows, nCols, nTiles, blob, offsets, stats, nullptr, status);
}

gf_status gf_huffman_analyze_batch_h2(gf_context *c, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                      const uint64_t *offsets, gf_codec_stats *stats, int64_t *pairCounts, int32_t *status)
{
    GF_CTX_LOCK(c);
    if (!pairCounts) return GF_ERR_ARG;
    return analyzeBatch(c, nRows, nCols, nTiles, blob, offsets, stats, pairCounts, status);
}

// CodecStats.getH2 (CodecStats.java:157-190) from one table of pair counts: sA[v] is the column sum of sB
double gf_codec_stats_h2(const int64_t *sB)
{
    if (!sB) return 0.0;
    std::vector<int64_t> sA(256, 0);
    int64_t k = 0;
    for (int p = 0; p < 256; p++)
        for (int v = 0; v < 256; v++) sA[v] += sB[p * 256 + v];
    for (int i = 0; i < 256; i++) k += sA[i];
    if (k == 0) return 0.0;
    double h2 = 0;
    for (int i = 0; i < 256; i++) {
        if (sA[i] > 0) {
            const double pI = (double)sA[i] / (double)k;
            int64_t n = 0;
            for (int j = i * 256; j < i * 256 + 256; j++) n += sB[j];
            double sumJ = 0;
            for (int j = i * 256; j < i * 256 + 256; j++)
                if (sB[j] > 0) {
                    const double pJ = (double)sB[j] / (double)n;
                    sumJ += pJ * std::log(pJ);
                }
            h2 += pI * sumJ;
        }
    }
    return -h2;
}

}  // extern "C"
