// gvrs_encode_layout.h -- LDS-resident state of the encode kernel, shared with the host-side
// debug tooling (tools/, tests/) so that a diagnostic dump can be decoded with offsetof.
#pragma once

#include "huff_build.h"

#define GF_ENC_THREADS 256
#define GF_ENC_WAVES (GF_ENC_THREADS / 64)
#define GF_IMG_WORDS 84                 // 80 header bits + 8 + 2559 tree bits -> 83 words

struct EncPersist {
    uint32_t hist[3][256];                      // reduced histograms
    uint64_t tab[3][256];                       // (len << 56) | code per symbol
    uint32_t img[3][GF_IMG_WORDS];                 // packing header + serialised tree
    uint64_t totalBits[3];
    uint32_t treeEndBit[3];                     // 80 + tree bits
    uint32_t maxLen[3];
    uint32_t maxN[3];
    uint32_t nM32[3];
    int32_t model[3];
    uint32_t seed;
    uint32_t flags;                             // bit0 any null, bit1 any valid
    uint32_t waveSum[GF_ENC_WAVES];
    unsigned long long sumStart;                // nulls predictor seed
    uint32_t nStart;
};


// words per tile of the optional debug dump: EncPersist followed by the three GfHuffTree
#define GF_ENC_DEBUG_WORDS ((sizeof(EncPersist) + 3 * sizeof(GfHuffTree)) / 4 + 16)
