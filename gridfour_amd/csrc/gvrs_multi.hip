// gvrs_multi.hip -- one process, several MI355X: the tile batch of a flush or of a read-ahead sharded over the GPUs of a node.
//
// Tiles are independent (gvrs/RasterTile.java:237-241), so a batch shards as contiguous tile ranges [g T / G, (g+1) T / G)
// with no exchange between devices and no collective: what one JVM's CodecMaster (gvrs/CodecMaster.java:142-203) or
// RecordManager.writeTile (gvrs/RecordManager.java:386-490) would call to use the whole node.  A gf_multi owns one
// gf_context per listed device (a device may be listed more than once: several contexts then share it).
//   host memory    one host thread per context runs the pipelined staging of gvrs_api.hip on its tile range; the packings
//                  of the ranges are concatenated by an exclusive scan of the range totals
//   device memory  launches are asynchronous, so the calling thread enqueues every device's work on that device's stream
//                  and returns; gf_multi_synchronize waits for all of them
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gvrs_hip_codec.h"

// gf_last_error() is per thread: the text of a failing shard's thread is handed to the thread that called gf_*_multi (gvrs_api.hip)
extern "C" __attribute__((visibility("hidden"))) void gf_internal_set_last_error(const char *text);

struct gf_multi {
    std::vector<gf_context *> ctx;
    std::vector<int> device;
    // per shard: where its packings wait for the concatenation (kept between calls; plain malloc: never touched beyond what
    // a call writes)
    std::vector<uint8_t *> part;
    std::vector<size_t> partCap;
    ~gf_multi()
    {
        for (uint8_t *p : part) free(p);
    }
};

extern "C" {

gf_status gf_multi_create(const int *devices, int n, gf_multi **out)
{
    if (!out) return GF_ERR_ARG;
    *out = nullptr;
    if (!devices || n < 1 || n > 1024) return GF_ERR_ARG;
    gf_multi *m = new (std::nothrow) gf_multi();
    if (!m) return GF_ERR_ARG;
    for (int i = 0; i < n; i++) {
        gf_context *c = nullptr;
        const gf_status s = gf_context_create(devices[i], &c);
        if (s != GF_OK) {
            for (gf_context *x : m->ctx) gf_context_destroy(x);
            delete m;
            return s;
        }
        m->ctx.push_back(c);
        m->device.push_back(devices[i]);
        m->part.push_back(nullptr);
        m->partCap.push_back(0);
    }
    *out = m;
    return GF_OK;
}

void gf_multi_destroy(gf_multi *m)
{
    if (!m) return;
    for (gf_context *x : m->ctx) gf_context_destroy(x);
    delete m;
}

int gf_multi_count(const gf_multi *m) { return m ? (int)m->ctx.size() : 0; }

gf_context *gf_multi_context(gf_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }

int gf_multi_device(const gf_multi *m, int i) { return (m && i >= 0 && i < (int)m->device.size()) ? m->device[i] : -1; }

// tile range of shard i of n: [i T / n, (i+1) T / n)
void gf_multi_partition(size_t nTiles, int n, int i, size_t *t0, size_t *t1)
{
    if (n < 1 || i < 0 || i >= n) {
        if (t0) *t0 = 0;
        if (t1) *t1 = 0;
        return;
    }
    const unsigned __int128 T = nTiles;
    if (t0) *t0 = (size_t)(T * (unsigned)i / (unsigned)n);
    if (t1) *t1 = (size_t)(T * (unsigned)(i + 1) / (unsigned)n);
}

gf_status gf_multi_synchronize(gf_multi *m)
{
    if (!m) return GF_ERR_ARG;
    gf_status r = GF_OK;
    for (gf_context *c : m->ctx) {
        const gf_status s = gf_context_synchronize(c);
        if (s != GF_OK && r == GF_OK) r = s;
    }
    return r;
}

}  // extern "C"

namespace {

// the host-memory batch entry point of a codec, as a callable: (context, codec index, rows, columns, tiles, cells, blob, capacity,
// offsets, per-tile byte (predictor / container type) or null, status or null); cells are 32-bit words (int32 or float32)
typedef std::function<gf_status(gf_context *, int, int, int, size_t, const int32_t *, uint8_t *, size_t, uint64_t *, uint8_t *, int32_t *)>
    EncodeHostFn;
typedef std::function<gf_status(gf_context *, int, int, size_t, const uint8_t *, const uint64_t *, int32_t *, int32_t *)> DecodeHostFn;

// perTileGuess: bytes of staging per tile for the first attempt (0: half of the raw cells, what integer packings stay far below;
// CodecFloat packs to 3-4 bytes per cell and LSOP12 has a bound of its own -- sized too small, every shard was encoded twice)
gf_status encodeMulti(const EncodeHostFn &fn, gf_multi *m, int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                      uint8_t *blob, size_t blobCap, uint64_t *offsets, uint8_t *predictors, int32_t *status, size_t perTileGuess = 0)
{
    if (!m || nRows < 1 || nCols < 1 || (!values && nTiles) || !offsets || (!blob && blobCap)) return GF_ERR_ARG;
    const int G = (int)m->ctx.size();
    const size_t cells = (size_t)nRows * (size_t)nCols;
    const size_t stride = gf_huffman_default_stride(nRows, nCols);
    struct Part {
        size_t t0 = 0, t1 = 0;
        std::vector<uint64_t> off;
        gf_status st = GF_OK;
        std::string err;                          // gf_last_error() of the shard's thread (the text is per thread)
    };
    std::vector<Part> part(G);
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++) {
        gf_multi_partition(nTiles, G, g, &part[g].t0, &part[g].t1);
        th.emplace_back([&, g]() {
            Part &p = part[g];
            const size_t n = p.t1 - p.t0;
            p.off.assign(n + 1, 0);
            if (n == 0) return;
            // a shard's packings rarely exceed half of its raw cells; grow once if they do
            size_t cap = n * (perTileGuess ? perTileGuess : stride / 2) + 4096;
            for (int attempt = 0; attempt < 2; attempt++) {
                if (m->partCap[g] < cap) {
                    free(m->part[g]);
                    m->part[g] = (uint8_t *)malloc(cap);
                    m->partCap[g] = m->part[g] ? cap : 0;
                    if (!m->part[g]) {                                    // out of host memory: not the "grow the blob" status
                        p.st = GF_ERR_HIP;
                        p.err = "gf_multi: no host memory for a shard's staging buffer";
                        return;
                    }
                }
                p.st = fn(m->ctx[g], codecIndex, nRows, nCols, n, values + p.t0 * cells, m->part[g], m->partCap[g], p.off.data(),
                          predictors ? predictors + p.t0 : nullptr, status ? status + p.t0 : nullptr);
                if (p.st != GF_ERR_CAPACITY) break;
                cap = (size_t)p.off[n] + 64;
            }
            if (p.st != GF_OK) p.err = gf_last_error();
            if (p.st == GF_ERR_CAPACITY) {
                // still too small after the shard's own regrow: not the caller's "your blob is too small" (offsets[n] means nothing
                // to the caller here)
                p.st = GF_ERR_HIP;
                p.err = "gf_multi: a shard's staging buffer was too small twice";
            }
        });
    }
    for (auto &t : th) t.join();
    // concatenate: exclusive scan of the shard totals
    uint64_t total = 0;
    gf_status r = GF_OK;
    offsets[0] = 0;
    for (int g = 0; g < G; g++) {
        const Part &p = part[g];
        if (p.st != GF_OK && r == GF_OK) {
            r = p.st;
            gf_internal_set_last_error(p.err.c_str());            // the failing shard's text, for the caller's thread
        }
        const size_t n = p.t1 - p.t0;
        for (size_t t = 0; t < n; t++) offsets[p.t0 + t + 1] = total + p.off[t + 1];
        total += n ? p.off[n] : 0;
    }
    if (r != GF_OK) return r;
    if (total > blobCap) return GF_ERR_CAPACITY;
    th.clear();
    for (int g = 0; g < G; g++) {
        const Part &p = part[g];
        const size_t n = p.t1 - p.t0;
        if (n && p.off[n]) th.emplace_back([&, g]() { memcpy(blob + offsets[part[g].t0], m->part[g], (size_t)part[g].off[part[g].t1 - part[g].t0]); });
    }
    for (auto &t : th) t.join();
    return GF_OK;
}

gf_status decodeMulti(const DecodeHostFn &fn, gf_multi *m, int nRows, int nCols, size_t nTiles, const uint8_t *blob, const uint64_t *offsets,
                      int32_t *values, int32_t *status)
{
    if (!m || nRows < 1 || nCols < 1 || !blob || !offsets || (!values && nTiles)) return GF_ERR_ARG;
    for (size_t t = 0; t < nTiles; t++)
        if (offsets[t + 1] < offsets[t]) return GF_ERR_ARG;
    const int G = (int)m->ctx.size();
    const size_t cells = (size_t)nRows * (size_t)nCols;
    std::vector<gf_status> st(G, GF_OK);
    std::vector<std::string> err(G);
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++) {
        th.emplace_back([&, g]() {
            size_t t0, t1;
            gf_multi_partition(nTiles, G, g, &t0, &t1);
            const size_t n = t1 - t0;
            if (n == 0) return;
            // the shard's packings, offsets relative to its first byte
            std::vector<uint64_t> rel(n + 1);
            for (size_t t = 0; t <= n; t++) rel[t] = offsets[t0 + t] - offsets[t0];
            st[g] = fn(m->ctx[g], nRows, nCols, n, blob + offsets[t0], rel.data(), values + t0 * cells, status ? status + t0 : nullptr);
            if (st[g] != GF_OK) err[g] = gf_last_error();
        });
    }
    for (auto &t : th) t.join();
    for (int g = 0; g < G; g++)
        if (st[g] != GF_OK) {
            gf_internal_set_last_error(err[g].c_str());
            return st[g];
        }
    return GF_OK;
}

}  // namespace

extern "C" {

gf_status gf_huffman_encode_batch_i32_multi(gf_multi *m, int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                                            uint8_t *blob, size_t blobCap, uint64_t *offsets, uint8_t *predictors, int32_t *status)
{
    return encodeMulti(gf_huffman_encode_batch_i32, m, codecIndex, nRows, nCols, nTiles, values, blob, blobCap, offsets, predictors, status);
}

gf_status gf_huffman_decode_batch_i32_multi(gf_multi *m, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                            const uint64_t *offsets, int32_t *values, int32_t *status)
{
    return decodeMulti(gf_huffman_decode_batch_i32, m, nRows, nCols, nTiles, blob, offsets, values, status);
}

gf_status gf_canon_encode_batch_i32_multi(gf_multi *m, int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                                          uint8_t *blob, size_t blobCap, uint64_t *offsets, uint8_t *predictors, int32_t *status)
{
    return encodeMulti(gf_canon_encode_batch_i32, m, codecIndex, nRows, nCols, nTiles, values, blob, blobCap, offsets, predictors, status);
}

gf_status gf_canon_decode_batch_i32_multi(gf_multi *m, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                          const uint64_t *offsets, int32_t *values, int32_t *status)
{
    return decodeMulti(gf_canon_decode_batch_i32, m, nRows, nCols, nTiles, blob, offsets, values, status);
}

gf_status gf_deflate_encode_batch_i32_multi(gf_multi *m, int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                                            uint8_t *blob, size_t blobCap, uint64_t *offsets, uint8_t *predictors, int32_t *status)
{
    return encodeMulti(gf_deflate_encode_batch_i32, m, codecIndex, nRows, nCols, nTiles, values, blob, blobCap, offsets, predictors, status);
}

gf_status gf_deflate_decode_batch_i32_multi(gf_multi *m, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                            const uint64_t *offsets, int32_t *values, int32_t *status)
{
    return decodeMulti(gf_deflate_decode_batch_i32, m, nRows, nCols, nTiles, blob, offsets, values, status);
}

// LSOP12 (BASELINE config 5-ii sharded): types = the container each tile ended up in, as gf_lsop12_encode_batch_i32 reports it
gf_status gf_lsop12_encode_batch_i32_multi(gf_multi *m, int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                                           int deflateEnabled, uint8_t *blob, size_t blobCap, uint64_t *offsets, uint8_t *types,
                                           int32_t *status)
{
    const EncodeHostFn fn = [deflateEnabled](gf_context *c, int ci, int nr, int nc, size_t n, const int32_t *v, uint8_t *b, size_t cap,
                                             uint64_t *off, uint8_t *ty, int32_t *st) {
        return gf_lsop12_encode_batch_i32(c, ci, nr, nc, n, v, deflateEnabled, b, cap, off, ty, st);
    };
    return encodeMulti(fn, m, codecIndex, nRows, nCols, nTiles, values, blob, blobCap, offsets, types, status,
                       gf_lsop12_max_packing(nRows, nCols) / 2 + 64);
}

gf_status gf_lsop12_decode_batch_i32_multi(gf_multi *m, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                           const uint64_t *offsets, int32_t *values, int32_t *status)
{
    return decodeMulti(gf_lsop12_decode_batch_i32, m, nRows, nCols, nTiles, blob, offsets, values, status);
}

// CodecFloat (BASELINE config 5-i sharded): float32 cells; no per-tile byte, no encode status (CodecFloat never declines)
gf_status gf_float_encode_batch_f32_multi(gf_multi *m, int codecIndex, int nRows, int nCols, size_t nTiles, const float *values,
                                          int zlibLevel, uint8_t *blob, size_t blobCap, uint64_t *offsets)
{
    const EncodeHostFn fn = [zlibLevel](gf_context *c, int ci, int nr, int nc, size_t n, const int32_t *v, uint8_t *b, size_t cap,
                                        uint64_t *off, uint8_t *, int32_t *) {
        return gf_float_encode_batch_f32(c, ci, nr, nc, n, reinterpret_cast<const float *>(v), zlibLevel, b, cap, off);
    };
    // (a CodecFloat packing: five planes of a byte per cell or less, deflated; sign bits and framing on top)
    return encodeMulti(fn, m, codecIndex, nRows, nCols, nTiles, reinterpret_cast<const int32_t *>(values), blob, blobCap, offsets,
                       nullptr, nullptr, 5 * (size_t)nRows * (size_t)nCols + 4096);
}

gf_status gf_float_decode_batch_f32_multi(gf_multi *m, int nRows, int nCols, size_t nTiles, const uint8_t *blob,
                                          const uint64_t *offsets, float *values, int32_t *status)
{
    const DecodeHostFn fn = [](gf_context *c, int nr, int nc, size_t n, const uint8_t *b, const uint64_t *off, int32_t *v, int32_t *st) {
        return gf_float_decode_batch_f32(c, nr, nc, n, b, off, reinterpret_cast<float *>(v), st);
    };
    return decodeMulti(fn, m, nRows, nCols, nTiles, blob, offsets, reinterpret_cast<int32_t *>(values), status);
}

// Device-resident shards: shard g lives on context g's device.  Every array argument has gf_multi_count(m) entries; the
// calls only enqueue (one stream per device) and return -- gf_multi_synchronize waits.
gf_status gf_huffman_encode_batch_i32_multi_dev(gf_multi *m, int codecIndex, int nRows, int nCols, const size_t *nTiles,
                                                const int32_t *const *dValues, uint8_t *const *dOut, size_t slotStride,
                                                uint32_t *const *dLengths, uint8_t *const *dPredictors, int32_t *const *dStatus,
                                                int predictorMask)
{
    if (!m || !nTiles || !dValues || !dOut || !dLengths || !dStatus) return GF_ERR_ARG;
    for (size_t g = 0; g < m->ctx.size(); g++) {
        const gf_status s = gf_huffman_encode_batch_i32_dev(m->ctx[g], nullptr, codecIndex, nRows, nCols, nTiles[g], dValues[g], dOut[g],
                                                            slotStride, dLengths[g], dPredictors ? dPredictors[g] : nullptr,
                                                            dStatus[g], predictorMask);
        if (s != GF_OK) return s;
    }
    return GF_OK;
}

gf_status gf_huffman_decode_batch_i32_multi_dev(gf_multi *m, int nRows, int nCols, const size_t *nTiles, const uint8_t *const *dBlob,
                                                const size_t *blobBytes, const uint64_t *const *dOffsets, size_t slotStride,
                                                const uint32_t *const *dLengths, int32_t *const *dValues, int32_t *const *dStatus)
{
    if (!m || !nTiles || !dBlob || !blobBytes || !dLengths || !dValues || !dStatus) return GF_ERR_ARG;
    for (size_t g = 0; g < m->ctx.size(); g++) {
        const gf_status s = gf_huffman_decode_batch_i32_dev(m->ctx[g], nullptr, nRows, nCols, nTiles[g], dBlob[g], blobBytes[g],
                                                            dOffsets ? dOffsets[g] : nullptr, slotStride, dLengths[g], dValues[g],
                                                            dStatus[g]);
        if (s != GF_OK) return s;
    }
    return GF_OK;
}

gf_status gf_canon_encode_batch_i32_multi_dev(gf_multi *m, int codecIndex, int nRows, int nCols, const size_t *nTiles,
                                              const int32_t *const *dValues, uint8_t *const *dOut, size_t slotStride,
                                              uint32_t *const *dLengths, uint8_t *const *dPredictors, int32_t *const *dStatus,
                                              int predictorMask)
{
    if (!m || !nTiles || !dValues || !dOut || !dLengths || !dStatus) return GF_ERR_ARG;
    for (size_t g = 0; g < m->ctx.size(); g++) {
        const gf_status s = gf_canon_encode_batch_i32_dev(m->ctx[g], nullptr, codecIndex, nRows, nCols, nTiles[g], dValues[g], dOut[g],
                                                          slotStride, dLengths[g], dPredictors ? dPredictors[g] : nullptr, dStatus[g],
                                                          predictorMask);
        if (s != GF_OK) return s;
    }
    return GF_OK;
}

gf_status gf_canon_decode_batch_i32_multi_dev(gf_multi *m, int nRows, int nCols, const size_t *nTiles, const uint8_t *const *dBlob,
                                              const size_t *blobBytes, const uint64_t *const *dOffsets, size_t slotStride,
                                              const uint32_t *const *dLengths, int32_t *const *dValues, int32_t *const *dStatus)
{
    if (!m || !nTiles || !dBlob || !blobBytes || !dLengths || !dValues || !dStatus) return GF_ERR_ARG;
    for (size_t g = 0; g < m->ctx.size(); g++) {
        const gf_status s = gf_canon_decode_batch_i32_dev(m->ctx[g], nullptr, nRows, nCols, nTiles[g], dBlob[g], blobBytes[g],
                                                          dOffsets ? dOffsets[g] : nullptr, slotStride, dLengths[g], dValues[g],
                                                          dStatus[g]);
        if (s != GF_OK) return s;
    }
    return GF_OK;
}

}  // extern "C"
