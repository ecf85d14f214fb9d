/* mic_oracle_int.h -- helpers shared by the oracle's C files.  TEST INFRASTRUCTURE ONLY. */
#ifndef MIC_ORACLE_INT_H
#define MIC_ORACLE_INT_H
#include "mic_oracle.h"
#include <stdlib.h>
#include <string.h>

#define MAXSYM 65535u
#define MIN_TABLELOG 5      /* fseu16.go:26 */
#define MAX_TABLELOG 16     /* fseu16.go:23 */
#define DEFAULT_TABLELOG 11 /* fseu16.go:25 */

static inline int len16(uint16_t v) { /* math/bits.Len16 */
    int n = 0;
    while (v) { n++; v >>= 1; }
    return n;
}
static inline uint32_t high_bits(uint32_t v) { /* fseu16.go:170-172: Len32(v)-1, wraps for 0 */
    int n = 0;
    while (v) { n++; v >>= 1; }
    return (uint32_t)(n - 1);
}
static inline uint32_t table_step(uint32_t size) { /* fseu16.go:166-168 */
    return (size >> 1) + (size >> 3) + 3;
}


typedef struct { uint32_t new_state; uint16_t symbol; uint8_t nb_bits; } dec_sym; /* fseu16.go:48-52 */


/* bitwriter.go: LSB-first; the flush cadence never changes content, so this
 * writer is the pure concatenation. */
typedef struct { uint8_t *out; size_t len, cap; uint64_t acc; unsigned nbits; int overflow; } bitw;
static inline void bw_flush_bytes(bitw *b) {
    while (b->nbits >= 8) {
        if (b->len < b->cap) b->out[b->len++] = (uint8_t)b->acc; else b->overflow = 1;
        b->acc >>= 8; b->nbits -= 8;
    }
}
static inline void bw_add(bitw *b, uint32_t value, unsigned nb) { /* addBits32NC, bitwriter.go:50-53 */
    uint32_t m = nb >= 32 ? 0xFFFFFFFFu : (((uint32_t)1 << nb) - 1);
    b->acc |= (uint64_t)(value & m) << b->nbits;
    b->nbits += nb;
    bw_flush_bytes(b);
}
static inline void bw_close(bitw *b) { /* bitwriter.go:162-168 */
    bw_add(b, 1, 1);
    if (b->nbits > 0) {
        if (b->len < b->cap) b->out[b->len++] = (uint8_t)b->acc; else b->overflow = 1;
        b->acc = 0; b->nbits = 0;
    }
}


/* bitreader.go: reverse reader keyed by the end mark in the last byte.  Kept as
 * an absolute bit cursor; reads past the front are an error (Go: ErrUnexpectedEOF). */
typedef struct { const uint8_t *in; size_t cursor; int over; } bitr;

static inline int br_init(bitr *b, const uint8_t *in, size_t len) { /* bitreader.go:27-47 */
    if (len < 1) return MICO_ERR_CORRUPT;
    uint8_t v = in[len - 1];
    if (v == 0) return MICO_ERR_CORRUPT;
    b->in = in; b->over = 0;
    b->cursor = 8 * (len - 1) + high_bits(v);
    return MICO_OK;
}
static inline uint32_t br_get(bitr *b, unsigned n) { /* getBits32, bitreader.go:49-61 */
    if (n == 0) return 0;
    if (b->cursor < n) { b->over = 1; b->cursor = 0; return 0; }
    b->cursor -= n;
    size_t byte = b->cursor >> 3;
    unsigned sh = (unsigned)(b->cursor & 7);
    uint64_t w = 0;
    for (unsigned k = 0; k < 5; k++) {
        /* never reads past the end-mark byte: cursor+n <= 8*(len-1)+7 */
        w |= (uint64_t)b->in[byte + k] << (8 * k);
        if (8 * (k + 1) >= sh + n) break;
    }
    return (uint32_t)((w >> sh) & (((uint64_t)1 << n) - 1));
}
static inline int br_finished(const bitr *b) { return b->cursor == 0; }


int mico_rans_compress8(const uint16_t *in, size_t n, const int32_t *norm, uint32_t symbol_len,
                        uint8_t table_log, bitw *bw);
int mico_rans_decompress8(const uint8_t *bits, size_t len, const int32_t *norm, uint32_t symbol_len,
                          uint8_t table_log, uint32_t count, uint16_t *out);
#endif
