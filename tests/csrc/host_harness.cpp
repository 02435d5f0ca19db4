// Host build of the shared host/device headers of the HIP codec, so that the
// tree-construction shortcut (huff_build.h) and the stream-order helpers
// (gvrs_common.h) can be checked against the oracle on the CPU.  Test-only.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../gridfour_amd/csrc/huff_build.h"

extern "C" {

// Huffman-encodes symbols the way the HIP kernels do (sorted leaves -> merge ->
// per-leaf code/position -> tree image + text), appending at *bitPos.
int hh_huffman_encode(uint8_t *bits, size_t capBits, size_t *bitPos, const uint8_t *symbols,
                      size_t nSymbols, uint8_t *codeLen256)
{
    uint32_t hist[256] = {0};
    for (size_t i = 0; i < nSymbols; i++) hist[symbols[i]]++;
    std::vector<int> order;
    for (int s = 0; s < 256; s++) if (hist[s]) order.push_back(s);
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        return hist[a] != hist[b] ? hist[a] < hist[b] : a < b; });
    size_t pos = *bitPos;
    auto put = [&](uint64_t v, int n) {
        for (int i = 0; i < n; i++, pos++) {
            if (pos >= capBits) return;
            if ((v >> i) & 1) bits[pos >> 3] |= (uint8_t)(1u << (pos & 7));
        }
    };
    memset(codeLen256, 0, 256);
    int n = (int)order.size();
    if (n == 0) return -1;
    if (n == 1) {                       // HuffmanEncoder.java:147-157
        put(0, 8); put(1, 1); put((uint64_t)order[0], 8);
        *bitPos = pos;
        return 0;
    }
    static GfHuffTree T;
    T.n = n;
    for (int i = 0; i < n; i++) { T.cnt[i] = hist[order[i]]; T.sym[i] = (uint8_t)order[i]; }
    gf_huff_merge(T, n, true);
    uint64_t code[256]; int len[256];
    size_t treeStart = pos;
    put((uint64_t)(n - 1), 8);
    for (int i = 0; i < n; i++) {
        uint64_t c; uint32_t p;
        int l = gf_huff_leaf_code(T, i, &c, &p);
        code[T.sym[i]] = c; len[T.sym[i]] = l; codeLen256[T.sym[i]] = (uint8_t)l;
        size_t save = pos;
        pos = treeStart + 8 + p;
        put(1, 1); put(T.sym[i], 8);
        pos = save;
    }
    pos = treeStart + 8 + 10 * (size_t)n - 1;
    for (size_t i = 0; i < nSymbols; i++) put(code[symbols[i]], len[symbols[i]]);
    *bitPos = pos;
    return pos > capBits ? -3 : 0;
}

uint32_t hh_stream_cell(int model, uint32_t nR, uint32_t nC, uint32_t s) { return gf_stream_cell(model, nR, nC, s); }
int hh_m32_len(uint32_t x) { return gf_m32_len(x); }
uint32_t hh_m32_byte(uint32_t x, int n, int k) { return gf_m32_byte(x, n, k); }
}

// layout of the encode kernel's debug dump (gvrs_encode_layout.h)
#include "../../gridfour_amd/csrc/gvrs_encode_layout.h"
#include <cstddef>
extern "C" {
size_t hh_sizeof_persist() { return sizeof(EncPersist); }
size_t hh_sizeof_tree() { return sizeof(GfHuffTree); }
size_t hh_off_persist(int f)
{
    switch (f) {
    case 0: return offsetof(EncPersist, hist);
    case 1: return offsetof(EncPersist, tab);
    case 2: return offsetof(EncPersist, img);
    case 3: return offsetof(EncPersist, totalBits);
    case 4: return offsetof(EncPersist, treeEndBit);
    case 5: return offsetof(EncPersist, maxLen);
    case 6: return offsetof(EncPersist, maxN);
    case 7: return offsetof(EncPersist, nM32);
    case 8: return offsetof(EncPersist, model);
    case 9: return offsetof(EncPersist, seed);
    default: return 0;
    }
}
size_t hh_off_tree(int f)
{
    switch (f) {
    case 0: return offsetof(GfHuffTree, cnt);
    case 1: return offsetof(GfHuffTree, parent);
    case 2: return offsetof(GfHuffTree, left);
    case 3: return offsetof(GfHuffTree, nl);
    case 4: return offsetof(GfHuffTree, bq);
    case 5: return offsetof(GfHuffTree, sym);
    case 6: return offsetof(GfHuffTree, n);
    default: return 0;
    }
}
// reference tables for a histogram: sorted leaves, parents, nl (same code path as the device)
int hh_build_tree(const uint32_t *hist, GfHuffTree *T)
{
    std::vector<int> order;
    for (int s = 0; s < 256; s++) if (hist[s]) order.push_back(s);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return hist[a] != hist[b] ? hist[a] < hist[b] : a < b; });
    memset(T, 0, sizeof(*T));
    T->n = (int)order.size();
    for (int i = 0; i < T->n; i++) { T->cnt[i] = hist[order[i]]; T->sym[i] = (uint8_t)order[i]; }
    if (T->n > 1) gf_huff_merge(*T, T->n, true);
    return T->n;
}
}
