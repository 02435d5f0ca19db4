/*
 * oracle_bits.h -- the LSB-first bit store shared by the oracle's translation units.
 * TEST INFRASTRUCTURE (see gvrs_oracle.h).
 * Bit i of the stream = bit (i&7) of byte (i>>3):
 * io/BitOutputStore.java:46-59, 205-288; io/BitInputStore.java:112-210.
 */
#ifndef GVRS_ORACLE_BITS_H
#define GVRS_ORACLE_BITS_H
#include <stddef.h>
#include <stdint.h>

typedef struct {
    uint8_t *buf;      /* must be zero-initialised beyond pos */
    size_t capBits;
    size_t pos;
    int overflow;
} bitw_t;

static inline void bw_bits(bitw_t *w, int n, uint32_t v)
{
    /* appendBits(n, v): low n bits of v, LSB first (BitOutputStore.java:224-264); n <= 32.
     * The buffer is zero beyond pos, so OR-ing whole bytes is equivalent to appending bits. */
    if (w->pos + (size_t)n > w->capBits) { w->overflow = 1; return; }
    uint64_t x = (uint64_t)(n < 32 ? (v & ((1u << n) - 1u)) : v) << (w->pos & 7);
    size_t byte = w->pos >> 3;
    int total = n + (int)(w->pos & 7);
    for (int i = 0; i < total; i += 8) {
        w->buf[byte++] |= (uint8_t)x;
        x >>= 8;
    }
    w->pos += (size_t)n;
}

static inline void bw_bit(bitw_t *w, int bit) { bw_bits(w, 1, (uint32_t)(bit & 1)); }

typedef struct {
    const uint8_t *buf;
    size_t nBits;
    size_t pos;
    int overrun;
} bitr_t;

static inline int br_bit(bitr_t *r)
{
    if (r->pos >= r->nBits) { r->overrun = 1; return 0; }
    int b = (r->buf[r->pos >> 3] >> (r->pos & 7)) & 1;
    r->pos++;
    return b;
}

static inline uint32_t br_bits(bitr_t *r, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint32_t)br_bit(r) << i;
    return v;
}


#endif
