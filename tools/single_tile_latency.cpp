// One tile per call through the C ABI (BASELINE config 1): latency of gf_huffman_encode_i32 / gf_huffman_decode_i32 and the
// canonical pair on a 120 x 150 tile in pageable host memory, without an interpreter in the way (tools/single_tile_latency.py
// measures the same through the Python mirror).
//   g++ -O2 -std=c++17 tools/single_tile_latency.cpp -Lgridfour_amd/lib -lgvrs_hip -Wl,-rpath,$PWD/gridfour_amd/lib -o tools/bin/single_tile_latency
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/gvrs_hip_codec.h"

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int nR = argc > 2 ? atoi(argv[1]) : 120, nC = argc > 2 ? atoi(argv[2]) : 150, N = 500;
    gf_context *ctx = nullptr;
    if (gf_context_create(0, &ctx) != GF_OK) { printf("{\"error\": \"no device\"}\n"); return 10; }
    const size_t cells = (size_t)nR * nC;
    void *d = nullptr;
    if (gf_dev_malloc(ctx, cells * 4, &d) != GF_OK) return 11;
    if (gf_synth_dem_dev(ctx, nullptr, 0x9E3779B97F4A7C15ull + 2, nR, nC, 144, 0, 1, (int32_t *)d) != GF_OK) return 12;
    gf_context_synchronize(ctx);
    std::vector<int32_t> tile(cells), back(cells);
    if (gf_dev_download(ctx, tile.data(), d, cells * 4) != GF_OK) return 13;
    printf("{\"tile\": \"%dx%d int32, synthetic DEM\", \"calls\": %d", nR, nC, N);
    for (int canon = 0; canon < 2; canon++) {
        std::vector<uint8_t> pk(canon ? gf_canon_max_packing(nR, nC) : gf_huffman_max_packing(nR, nC));
        size_t len = 0;
        auto enc = [&]() {
            return canon ? gf_canon_encode_i32(ctx, 0, nR, nC, tile.data(), pk.data(), pk.size(), &len)
                         : gf_huffman_encode_i32(ctx, 0, nR, nC, tile.data(), pk.data(), pk.size(), &len);
        };
        auto dec = [&]() {
            return canon ? gf_canon_decode_i32(ctx, nR, nC, pk.data(), len, back.data()) : gf_huffman_decode_i32(ctx, nR, nC, pk.data(), len, back.data());
        };
        for (int i = 0; i < 20; i++)
            if (enc() != GF_OK || dec() != GF_OK) { printf(", \"error\": \"%s\"}\n", gf_last_error()); return 14; }
        if (memcmp(back.data(), tile.data(), cells * 4) != 0) { printf(", \"error\": \"round trip differs\"}\n"); return 15; }
        const double t0 = now();
        for (int i = 0; i < N; i++) enc();
        const double t1 = now();
        for (int i = 0; i < N; i++) dec();
        const double t2 = now();
        printf(", \"%s\": {\"encode_us\": %.1f, \"decode_us\": %.1f, \"packing_bytes\": %zu}", canon ? "CodecCanonHuffman" : "CodecHuffman",
               (t1 - t0) / N * 1e6, (t2 - t1) / N * 1e6, len);
    }
    printf("}\n");
    gf_dev_free(ctx, d);
    gf_context_destroy(ctx);
    return 0;
}
