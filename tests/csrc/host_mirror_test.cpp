// Exercises the C++ host mirror (gridfour_amd/host/gvrs_hip_codec.hpp).  Without a GPU it checks
// that construction fails loudly; with a GPU it round-trips tiles and prints a digest that the
// Python test compares with the oracle.
#include <cstdio>
#include <cstdlib>

#include "../../gridfour_amd/host/gvrs_hip_codec.hpp"

int main(int argc, char **argv)
{
    try {
        gridfour::CodecHuffmanHip codec(0);
        const int nRows = 33, nCols = 65;
        std::vector<int32_t> v((size_t)nRows * nCols);
        for (size_t i = 0; i < v.size(); i++) v[i] = (int32_t)((i * 7919u) % 211u) - 100 + (int32_t)(i / 65) * 3;
        auto p = codec.encode(4, nRows, nCols, v);
        if (!p) { std::puts("null"); return 2; }
        auto back = codec.decode(nRows, nCols, *p);
        if (back != v) { std::puts("roundtrip mismatch"); return 3; }
        std::printf("ok %zu", p->size());
        for (uint8_t b : *p) std::printf(" %02x", b);
        std::printf("\n");
        std::vector<int32_t> nulls(16, GF_INT4_NULL);
        if (codec.encode(0, 4, 4, nulls)) { std::puts("expected null"); return 4; }
        bool threw = false;
        try { std::vector<uint8_t> bad(*p); bad[1] = 9; codec.decode(nRows, nCols, bad); } catch (const gridfour::IOException &) { threw = true; }
        if (!threw) { std::puts("expected IOException"); return 5; }
        (void)argc; (void)argv;
        return 0;
    } catch (const std::runtime_error &e) {
        std::printf("no-device: %s\n", e.what());
        return 10;
    }
}
