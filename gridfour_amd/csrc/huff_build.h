// huff_build.h -- Huffman tree construction that reproduces, node for node, the
// tree compress/HuffmanEncoder.java:124-194 builds, without its linked list.
//
// Why a different algorithm gives the same tree
//   The reference keeps ONE list sorted by count: leaves ordered (count asc,
//   symbol asc) (:86-92,138), and every new branch is inserted before the first
//   node whose count is >= its own (:175-193).  Successive branch counts never
//   decrease, so the list is always the merge of
//     - the leaf queue, and
//     - the branch list: groups of equal count in creation order, and inside a
//       group the NEWEST branch first (it was inserted in front of its equals),
//   with branches ahead of leaves of equal count.  Popping "the two smallest"
//   is therefore: compare the head of the leaf queue with the top of the oldest
//   non-empty branch group (a stack), branch wins ties.  That is what
//   gf_huff_merge does with two cursors; no list scan, O(1) per pop.
//
// Node numbering: leaves 0..n-1 in sorted order, branches n..2n-2 in creation
// order; the root is node 2n-2.
//
// The functions are __host__ __device__ so that tests/ can compile this header
// with g++ and check it against the oracle's literal linked-list restatement.
#pragma once

#include "gvrs_common.h"

struct GfHuffTree {
    uint32_t cnt[511];     // node counts
    uint16_t parent[511];  // parent id, bit 15 set when the node is a RIGHT child (bit = 1)
    uint16_t left[255];    // left child of branch b (index b - n)
    uint16_t nl[511];      // leaves under the node
    uint16_t bq[256];      // branch queue slots (node ids)
    uint8_t sym[256];      // symbol of sorted leaf i
    int n;                 // number of leaves (distinct symbols)
};

// The fast encoder's tree (k_huffman_encode<true>): at most 256 leaves sorted in registers, the tree built by data-parallel
// rounds (wave_huff_rounds) -- no branch counts, no branch queue, i.e. 3.8 instead of 5.4 KB per tree; three of them fit under
// the 12 KB the histogram replicas need anyway, which is what lets eight workgroups of that kernel share a CU's LDS.
struct GfHuffTreeSlim {
    uint32_t cnt[256];     // counts of the sorted leaves
    uint16_t parent[511];
    uint16_t left[255];
    uint16_t nl[511];
    uint8_t sym[256];
    int n;
};

// GF_UNI(x): on the device the merge is executed by EVERY lane of one wave with identical
// (wave-uniform) state; readfirstlane tells the compiler so, which turns the whole loop
// into scalar code with scalar branches (no exec-mask juggling; hipcc 7.2 miscompiles the
// divergent single-lane form of this loop).  Stores are issued by the `writer` lane only.
#if defined(__HIP_DEVICE_COMPILE__)
#define GF_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define GF_UNI(x) ((uint32_t)(x))
#endif

// Sequential merge.  Leaves (cnt[0..n), sym[0..n)) must already be sorted by
// (count asc, symbol asc); nl[0..n) is set here.
//   UNI = true : wave-uniform execution (all lanes, same tree, scalarised; `n` wave-uniform,
//                `writer` = the one lane that stores)
//   UNI = false: plain per-thread execution -- on the device each LANE builds its own tree
//                (SIMT over trees), on the host this is the ordinary sequential code.
template <bool UNI>
GF_HD void gf_huff_merge_t(GfHuffTree &T, int n, bool writer)
{
#define GF_U(x) (UNI ? GF_UNI(x) : (uint32_t)(x))
    n = (int)GF_U(n);
    if (writer) {
        T.n = n;
        for (int i = 0; i < n; i++) T.nl[i] = 1;
    }
    uint32_t li = 0;                              // leaf queue head
    uint32_t gs = 0, top = 0, ge = 0, m = 0;      // branch groups: front group slots [gs,top), next group at ge, end m
    uint32_t next = (uint32_t)n;
    uint32_t leafCnt = n > 0 ? GF_U(T.cnt[0]) : 0u;      // count at the head of the leaf queue
    uint32_t topId = 0, topCnt = 0, topNl = 0;    // id, count, leaf count of the branch on top of the front group
    uint32_t lastCnt = 0;                         // count of the newest branch == count of the LAST group while
                                                  // the list is non-empty (the newest branch is popped first
                                                  // inside its group, and nothing newer can have a smaller count)
    for (int step = 0; step < n - 1; step++) {
        uint32_t pick[2], pickCnt[2], pickNl[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const bool haveB = top > gs;
            const bool haveL = li < (uint32_t)n;
            const bool takeB = haveB && (!haveL || topCnt <= leafCnt);
            if (takeB) {
                pick[k] = topId;
                pickCnt[k] = topCnt;
                pickNl[k] = topNl;
                if (ge == m) { m--; ge--; }       // front group is also the last: drop the slot
                top--;
                if (top == gs) {                  // front group exhausted: advance to the next group
                    gs = ge;
                    if (gs < m) {
                        const uint32_t c = GF_U(T.cnt[GF_U(T.bq[gs])]);
                        uint32_t e = gs + 1;
                        while (e < m && GF_U(T.cnt[GF_U(T.bq[e])]) == c) e++;
                        ge = e;
                        top = e;
                    } else {
                        gs = top = ge = m;
                    }
                }
                if (top > gs) {
                    topId = GF_U(T.bq[top - 1]);
                    topCnt = GF_U(T.cnt[topId]);
                    topNl = GF_U(T.nl[topId]);
                }
            } else {
                pick[k] = li;
                pickCnt[k] = leafCnt;
                pickNl[k] = 1u;
                li++;
                if (li < (uint32_t)n) leafCnt = GF_U(T.cnt[li]);
            }
        }
        const uint32_t a = pick[0], b = pick[1];
        const uint32_t id = next++;
        const uint32_t c = pickCnt[0] + pickCnt[1];
        // push: joins the last group when the counts are equal, else opens a new group
        const bool nonEmpty = top > gs;
        const uint32_t nlSum = pickNl[0] + pickNl[1];
        if (writer) {
            T.cnt[id] = c;
            T.parent[a] = (uint16_t)id;                 // left, bit 0  (HuffmanEncoder.java:75-83)
            T.parent[b] = (uint16_t)(id | 0x8000u);     // right, bit 1
            T.left[id - n] = (uint16_t)a;
            T.nl[id] = (uint16_t)nlSum;
            T.bq[m] = (uint16_t)id;
        }
        if (nonEmpty && lastCnt == c) {
            const bool frontIsLast = (ge == m);
            m++;
            if (frontIsLast) { ge = m; top = m; }
        } else {
            m++;
            if (!nonEmpty) { gs = m - 1; top = m; ge = m; }
        }
        lastCnt = c;
        if (top == m) {                                 // the new branch is the top of the front group
            topId = id;
            topCnt = c;
            topNl = nlSum;
        }
    }
    if (writer && n >= 1) T.parent[2 * n - 2] = 0xFFFF; // root
}
#undef GF_U

GF_HD void gf_huff_merge(GfHuffTree &T, int n, bool writer) { gf_huff_merge_t<true>(T, n, writer); }

// Code of sorted leaf i: path bits root->leaf, first step in bit 0 (the order
// HuffmanEncoder.java:198-213 appends them).  *pos = bit offset of the leaf's
// "1 + 8-bit symbol" record inside the pre-order tree serialisation
// (:221-294), counted from the root's bit.  Returns the code length.
template <class Tree>
GF_HD int gf_huff_leaf_code(const Tree &T, int i, uint64_t *code, uint32_t *pos)
{
    const int n = T.n;
    const int root = 2 * n - 2;
    uint64_t c = 0;
    uint32_t p = 0;
    int len = 0;
    int x = i;
    while (x != root && len < 256) {   // the bound only matters if the tables are corrupt
        uint32_t pr = T.parent[x];
        uint32_t isR = pr >> 15;
        pr &= 0x7fffu;
        c = (c << 1) | isR;
        // right child starts after the whole left subtree: 1 + (10*k - 1) bits for k leaves
        p += isR ? 10u * T.nl[T.left[pr - n]] : 1u;
        len++;
        x = (int)pr;
    }
    *code = c;
    *pos = p;
    return len;
}
