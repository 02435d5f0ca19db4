// gvrs_encode_layout.h -- LDS-resident state of the encode kernel, shared with the host-side
// debug tooling (tools/, tests/) so that a diagnostic dump can be decoded with offsetof.
#pragma once

#include "huff_build.h"

#ifndef GF_ENC_THREADS                  // (-DGF_ENC_THREADS=1024 -DGF_ENC_VARIANT: the one-tile-per-call build of gvrs_encode.hip)
#define GF_ENC_THREADS 256
#endif
#define GF_ENC_WAVES (GF_ENC_THREADS / 64)
#define GF_IMG_WORDS 84                 // 80 header bits + 8 + 2559 tree bits -> 83 words

// GF_ENC_HIST_SEPARATE: the reduced histograms as an array of their own -- the layout of the diagnostic flavour's dump
// (tools/, tests/csrc/host_harness.cpp).  In the shipping kernels histogram p lies over the first half of tab[p]: the wave that
// builds tree p reads its histogram before anything else and writes its code table last, and the 3 KB saved are what lets
// eight workgroups of k_huffman_encode<fast> share a CU's LDS (enc_hist() in gvrs_encode.hip).
#if defined(GF_DIAG) || defined(GF_ENC_LAYOUT_FULL)
#define GF_ENC_HIST_SEPARATE 1
#endif
struct EncPersist {
#ifdef GF_ENC_HIST_SEPARATE
    uint32_t hist[3][256];                      // reduced histograms
#endif
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
    uint32_t lbBytes[3];                        // fast kernel: lower bound of a predictor's packing (header + tree + entropy of its text)
    uint32_t plain[3];                          // the predictor's stream is plain bytes: every value one M32 byte, none the null code
};


// words per tile of the optional debug dump: EncPersist followed by the three GfHuffTree
#define GF_ENC_DEBUG_WORDS ((sizeof(EncPersist) + 3 * sizeof(GfHuffTree)) / 4 + 16)
