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
#define GF_ENC_LAYOUT_FULL 1          // the dump comes from the diagnostic flavour of the library
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

// ---- reference model of the data-parallel "rounds" tree construction used on the device ----
// Every round pairs up, in list order, all nodes whose count is below x0 + x1 (the sum of the two
// smallest): those merges are exactly the next merges of the sequential algorithm, and none of
// them can be affected by the branches created in the same round.  List order is a total order
// on keys (count << 9 | tie): leaves tie = 256 + sorted index, branch k tie = 254 - k (newer
// branches sort before older ones and before leaves of equal count, HuffmanEncoder.java:175-193).
extern "C" int hh_build_tree_rounds(const uint32_t *hist, GfHuffTree *T)
{
    std::vector<int> order;
    for (int s = 0; s < 256; s++) if (hist[s]) order.push_back(s);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return hist[a] != hist[b] ? hist[a] < hist[b] : a < b; });
    memset(T, 0, sizeof(*T));
    const int n = (int)order.size();
    T->n = n;
    std::vector<uint32_t> K;
    for (int i = 0; i < n; i++) {
        T->cnt[i] = hist[order[i]];
        T->sym[i] = (uint8_t)order[i];
        T->nl[i] = 1;
        K.push_back((hist[order[i]] << 9) | (uint32_t)(256 + i));
    }
    auto nodeId = [&](uint32_t key) { uint32_t tie = key & 511u; return tie >= 256 ? tie - 256 : (uint32_t)n + (254u - tie); };
    uint32_t kbase = 0;
    int rounds = 0;
    while (K.size() > 1) {
        rounds++;
        const uint32_t s0 = (K[0] >> 9) + (K[1] >> 9);
        size_t t = 0;
        while (t < K.size() && (K[t] >> 9) < s0) t++;
        const size_t P = t / 2;
        std::vector<uint32_t> next(K.begin() + 2 * P, K.end());
        for (size_t i = 0; i < P; i++) {
            const uint32_t a = nodeId(K[2 * i]), b = nodeId(K[2 * i + 1]);
            const uint32_t k = kbase + (uint32_t)i, id = (uint32_t)n + k;
            T->parent[a] = (uint16_t)id;
            T->parent[b] = (uint16_t)(id | 0x8000u);
            T->left[k] = (uint16_t)a;
            T->nl[id] = (uint16_t)(T->nl[a] + T->nl[b]);
            T->cnt[id] = (K[2 * i] >> 9) + (K[2 * i + 1] >> 9);
            next.push_back((T->cnt[id] << 9) | (254u - k));
        }
        kbase += (uint32_t)P;
        std::sort(next.begin(), next.end());
        K.swap(next);
    }
    if (n >= 1) T->parent[2 * n - 2] = 0xFFFF;
    return rounds;
}

// Huffman-encodes with the rounds construction (same framing as hh_huffman_encode)
extern "C" int hh_huffman_encode_rounds(uint8_t *bits, size_t capBits, size_t *bitPos, const uint8_t *symbols,
                                        size_t nSymbols, uint8_t *codeLen256)
{
    uint32_t hist[256] = {0};
    for (size_t i = 0; i < nSymbols; i++) hist[symbols[i]]++;
    static GfHuffTree T;
    hh_build_tree_rounds(hist, &T);
    const int n = T.n;
    size_t pos = *bitPos;
    auto put = [&](uint64_t v, int nb) {
        for (int i = 0; i < nb; i++, pos++) {
            if (pos >= capBits) return;
            if ((v >> i) & 1) bits[pos >> 3] |= (uint8_t)(1u << (pos & 7));
        }
    };
    memset(codeLen256, 0, 256);
    if (n == 0) return -1;
    if (n == 1) { put(0, 8); put(1, 1); put((uint64_t)T.sym[0], 8); *bitPos = pos; return 0; }
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
