// gvrs_readahead.hip -- the reading assistant of the tile cache as a batched, N-tile prefetch queue.
//
// Reference: gvrs/TileDecompressionAssistant.java:60-230 (one background thread that decodes the packing of ONE predicted
// tile while the application thread decodes the tile it asked for) and its caller gvrs/RasterTileCache.java:339-426
// (readTileUsingAssistant: getTilesWithWaitForIndex, then submitDecompression of tile index + 1 while fewer than two tasks
// are pending).  A GPU decodes thousands of tiles in the time the CPU codec needs for one, so the assistant here takes
// EVERYTHING that is queued when it wakes up -- up to max_batch tiles -- and decodes it as one batch through
// gf_tile_payload_decode_batch_i32 (the element bytes RecordManager.readTilePacking returns: a CodecMaster packing, or the
// raw cells when the element was stored uncompressed).  A cache that predicts a whole tile row instead of one tile gets
// the row back after one launch.
//
// Same roles as the reference's methods:
//   gf_readahead_submit      submitDecompression   (the packing is copied; the call returns at once)
//   gf_readahead_pending     getPendingTaskCount   (queued + in progress)
//   gf_readahead_cells       cells of a tile (what a taker must have room for, per tile)
//   gf_readahead_take        getTilesWithWaitForIndex: waits while wait_index is queued or in progress, then hands over
//                            finished tiles
// The assistant owns a context of its own on the cache's device (the reference's assistant owns its own CodecMaster,
// TileDecompressionAssistant.java:87-89), so the application thread may decode on its context at the same time.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/gvrs_hip_codec.h"

namespace {

struct RaTask {
    int32_t index;
    std::vector<uint8_t> packing;
};
struct RaResult {
    int32_t index;
    int32_t status;
    std::vector<int32_t> values;
};

}  // namespace

struct gf_readahead {
    gf_context *ctx = nullptr;
    std::vector<int> codecs;
    int nRows = 0, nCols = 0;
    size_t maxBatch = 1;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<RaTask> queue;
    std::vector<int32_t> inProgress;            // indices of the batch being decoded
    std::deque<RaResult> results;
    bool stop = false;
    gf_status workerError = GF_OK;              // a failing batch call (not a failing tile) is reported by the next take
    uint64_t nBatches = 0, nTiles = 0;
    std::thread worker;

    bool pendingLocked(int32_t index) const
    {
        for (int32_t i : inProgress)
            if (i == index) return true;
        for (const RaTask &t : queue)
            if (t.index == index) return true;
        return false;
    }

    void run()
    {
        const size_t cells = (size_t)nRows * (size_t)nCols;
        std::vector<RaTask> batch;
        std::vector<uint8_t> blob;
        std::vector<uint64_t> offsets;
        std::vector<int32_t> values, status;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !queue.empty(); });
                if (stop) return;
                batch.clear();
                while (!queue.empty() && batch.size() < maxBatch) {
                    batch.push_back(std::move(queue.front()));
                    queue.pop_front();
                }
                inProgress.clear();
                for (const RaTask &t : batch) inProgress.push_back(t.index);
            }
            // outside the lock: submit and take go on while the batch is decoded
            const size_t n = batch.size();
            offsets.assign(n + 1, 0);
            size_t total = 0;
            for (size_t i = 0; i < n; i++) {
                offsets[i] = total;
                total += 4 + batch[i].packing.size();
            }
            offsets[n] = total;
            blob.resize(total + 16);
            for (size_t i = 0; i < n; i++) {
                const uint32_t len = (uint32_t)batch[i].packing.size();
                uint8_t *p = blob.data() + offsets[i];
                p[0] = (uint8_t)len; p[1] = (uint8_t)(len >> 8); p[2] = (uint8_t)(len >> 16); p[3] = (uint8_t)(len >> 24);
                if (len) memcpy(p + 4, batch[i].packing.data(), len);
            }
            values.resize(n * cells);
            status.assign(n, GF_OK);
            const gf_status s = gf_tile_payload_decode_batch_i32(ctx, codecs.empty() ? nullptr : codecs.data(), (int)codecs.size(), nRows,
                                                                 nCols, n, blob.data(), offsets.data(), values.data(), status.data());
            {
                std::lock_guard<std::mutex> lk(mu);
                if (s != GF_OK) workerError = s;
                for (size_t i = 0; i < n; i++) {
                    RaResult r;
                    r.index = batch[i].index;
                    r.status = s != GF_OK ? (int32_t)s : status[i];
                    r.values.assign(values.begin() + i * cells, values.begin() + (i + 1) * cells);
                    results.push_back(std::move(r));
                }
                inProgress.clear();
                nBatches++;
                nTiles += n;
            }
            cv.notify_all();
        }
    }
};

extern "C" {

gf_status gf_readahead_create(int device, const int *codecs, int n_codecs, int n_rows, int n_cols, size_t max_batch, gf_readahead **out)
{
    if (!out) return GF_ERR_ARG;
    *out = nullptr;
    if (n_rows < 1 || n_cols < 1 || n_codecs < 0 || n_codecs > 255 || (n_codecs && !codecs) || max_batch < 1) return GF_ERR_ARG;
    gf_readahead *ra = new (std::nothrow) gf_readahead();
    if (!ra) return GF_ERR_ARG;
    const gf_status s = gf_context_create(device, &ra->ctx);
    if (s != GF_OK) {
        delete ra;
        return s;
    }
    ra->codecs.assign(codecs, codecs + n_codecs);
    ra->nRows = n_rows;
    ra->nCols = n_cols;
    ra->maxBatch = max_batch;
    ra->worker = std::thread([ra] { ra->run(); });
    *out = ra;
    return GF_OK;
}

void gf_readahead_destroy(gf_readahead *ra)
{
    if (!ra) return;
    {
        std::lock_guard<std::mutex> lk(ra->mu);
        ra->stop = true;
    }
    ra->cv.notify_all();
    if (ra->worker.joinable()) ra->worker.join();
    gf_context_destroy(ra->ctx);
    delete ra;
}

gf_status gf_readahead_submit(gf_readahead *ra, int32_t tile_index, const uint8_t *packing, size_t len)
{
    if (!ra || (!packing && len) || len > 0xFFFFFFF0ull) return GF_ERR_ARG;
    RaTask t;
    t.index = tile_index;
    t.packing.assign(packing, packing + len);
    {
        std::lock_guard<std::mutex> lk(ra->mu);
        ra->queue.push_back(std::move(t));
    }
    ra->cv.notify_all();
    return GF_OK;
}

int gf_readahead_pending(gf_readahead *ra)
{
    if (!ra) return 0;
    std::lock_guard<std::mutex> lk(ra->mu);
    return (int)(ra->queue.size() + ra->inProgress.size());
}

gf_status gf_readahead_take(gf_readahead *ra, int32_t wait_index, size_t max_tiles, int32_t *indices, int32_t *values, int32_t *status,
                            size_t *n_out)
{
    if (!ra || !n_out || (max_tiles && (!indices || !values))) return GF_ERR_ARG;
    *n_out = 0;
    const size_t cells = (size_t)ra->nRows * (size_t)ra->nCols;
    std::unique_lock<std::mutex> lk(ra->mu);
    ra->cv.wait(lk, [&] { return !ra->pendingLocked(wait_index); });
    // the tile that was waited for goes first, so that a caller with room for one tile gets that one
    size_t n = 0;
    for (int pass = 0; pass < 2 && n < max_tiles; pass++) {
        for (auto it = ra->results.begin(); it != ra->results.end() && n < max_tiles;) {
            if ((pass == 0) != (it->index == wait_index)) { ++it; continue; }
            indices[n] = it->index;
            if (status) status[n] = it->status;
            memcpy(values + n * cells, it->values.data(), cells * 4);
            n++;
            it = ra->results.erase(it);
        }
    }
    *n_out = n;
    const gf_status e = ra->workerError;
    ra->workerError = GF_OK;
    return e;
}

size_t gf_readahead_cells(gf_readahead *ra)
{
    return ra ? (size_t)ra->nRows * (size_t)ra->nCols : 0;
}

void gf_readahead_counters(gf_readahead *ra, uint64_t *n_batches, uint64_t *n_tiles)
{
    if (!ra) return;
    std::lock_guard<std::mutex> lk(ra->mu);
    if (n_batches) *n_batches = ra->nBatches;
    if (n_tiles) *n_tiles = ra->nTiles;
}

}  // extern "C"
