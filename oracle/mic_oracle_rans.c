/*
 * mic_oracle_rans.c -- CPU restatement of MIC's 8-lane "rANS" coder.
 * TEST INFRASTRUCTURE ONLY (see mic_oracle.h).
 *
 * Follows ransu16.go (tables, encode step) and rans8state.go (lane order,
 * framing).  Same NCount header and bit I/O as FSE; magic FF 08.
 */
#include "mic_oracle_int.h"

typedef struct { uint32_t freq, bias, threshold; uint8_t k0; } rans_enc_sym; /* ransu16.go:56-61 */

/* ransCompress8State, rans8state.go:106-219 with buildRansEncTable (ransu16.go:139-180)
 * and ransEncodeStep (ransu16.go:187-197).  Lane of symbol i is i % 8; encode order is
 * last symbol first; states start at 0; final states written lane 7 ... lane 0. */
int mico_rans_compress8(const uint16_t *in, size_t n, const int32_t *norm, uint32_t symbol_len,
                        uint8_t table_log, bitw *bw) {
    if (n <= 7) return MICO_ERR_INTERNAL;
    rans_enc_sym *tt = (rans_enc_sym *)calloc(symbol_len, sizeof(rans_enc_sym));
    if (!tt) return MICO_ERR_NOMEM;
    uint32_t cumul = 0;
    for (uint32_t sym = 0; sym < symbol_len; sym++) {
        int32_t v = norm[sym];
        if (v <= 0) continue;
        uint32_t freq = (uint32_t)v;
        uint8_t k0 = (uint8_t)(table_log - (uint8_t)high_bits(freq));
        tt[sym].freq = freq; tt[sym].bias = cumul; tt[sym].k0 = k0; tt[sym].threshold = freq << k0;
        cumul += freq;
    }
    for (uint32_t sym = 0; sym < symbol_len; sym++) {
        if (norm[sym] != -1) continue;
        tt[sym].freq = 1; tt[sym].bias = cumul; tt[sym].k0 = table_log;
        tt[sym].threshold = (uint32_t)1 << table_log;
        cumul++;
    }
    if (cumul != ((uint32_t)1 << table_log)) { free(tt); return MICO_ERR_INTERNAL; }
    uint32_t table_size = (uint32_t)1 << table_log;
    uint32_t st[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (size_t ip = n; ip > 0; ip--) {
        size_t idx = ip - 1;
        rans_enc_sym e = tt[in[idx]];
        uint32_t x = st[idx & 7];
        uint32_t xl = x + table_size;
        uint8_t k = e.k0;
        if (xl < e.threshold) k--;
        bw_add(bw, xl, k);
        st[idx & 7] = e.bias + ((xl >> k) - e.freq);
    }
    for (int k = 7; k >= 0; k--) bw_add(bw, st[k], table_log);
    bw_close(bw);
    free(tt);
    return MICO_OK;
}

/* buildRansDecTable (ransu16.go:77-135) + ransDecompress8State (rans8state.go:221-412) */
int mico_rans_decompress8(const uint8_t *bits, size_t len, const int32_t *norm, uint32_t symbol_len,
                          uint8_t table_log, uint32_t count, uint16_t *out) {
    uint32_t table_size = (uint32_t)1 << table_log;
    dec_sym *dt = (dec_sym *)calloc(table_size, sizeof(dec_sym));
    if (!dt) return MICO_ERR_NOMEM;
    uint32_t slot = 0;
    for (uint32_t sym = 0; sym < symbol_len; sym++) {
        int32_t v = norm[sym];
        if (v <= 0) continue;
        uint32_t freq = (uint32_t)v;
        for (uint32_t j = 0; j < freq; j++) {
            if (slot >= table_size) { free(dt); return MICO_ERR_CORRUPT; }
            uint32_t x_next = freq + j;
            uint8_t nb = (uint8_t)(table_log - (uint8_t)high_bits(x_next));
            uint32_t base = (x_next << nb) - table_size;
            if (base >= table_size) { free(dt); return MICO_ERR_CORRUPT; }
            dt[slot].new_state = base; dt[slot].symbol = (uint16_t)sym; dt[slot].nb_bits = nb;
            slot++;
        }
    }
    for (uint32_t sym = 0; sym < symbol_len; sym++) {
        if (norm[sym] != -1) continue;
        if (slot >= table_size) { free(dt); return MICO_ERR_CORRUPT; }
        dt[slot].new_state = 0; dt[slot].symbol = (uint16_t)sym; dt[slot].nb_bits = table_log;
        slot++;
    }
    if (slot != table_size) { free(dt); return MICO_ERR_CORRUPT; }
    bitr br;
    int rc = br_init(&br, bits, len);
    if (rc) { free(dt); return rc; }
    uint32_t st[8];
    for (int k = 0; k < 8; k++) st[k] = br_get(&br, table_log);
    if (br.over) { free(dt); return MICO_ERR_CORRUPT; }
    for (uint32_t i = 0; i < count; i++) {
        dec_sym e = dt[st[i & 7]];
        uint32_t low = br_get(&br, e.nb_bits);
        if (br.over) { free(dt); return MICO_ERR_CORRUPT; }
        st[i & 7] = e.new_state + low;
        out[i] = e.symbol;
    }
    free(dt);
    return MICO_OK;
}
