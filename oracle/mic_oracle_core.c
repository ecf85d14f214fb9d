/*
 * mic_oracle_core.c -- CPU restatement of MIC's Delta(avg)+RLE+FSE unit codec,
 * PICS strips and MIC2 containers.
 *
 * TEST INFRASTRUCTURE ONLY (see mic_oracle.h).  Plain C99, single thread,
 * written for clarity: each function follows the Go reference statement by
 * statement (same integer widths, same wrap-around) and cites it.
 * Citations are relative to the reference repository root.
 */
#include "mic_oracle_int.h"

/* ---------------------------------------------------------------- helpers */

uint64_t mico_fnv1a64(const uint8_t *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

/* ------------------------------------------------- RLE tokeniser (encode) */
/* rlecompressu16.go:8-83 */
typedef struct {
    uint16_t *out; size_t out_n, out_cap;
    uint16_t *b;   int bc;
    uint16_t mid_count;
    int same;
    int overflow;
} rle_enc;

static void rle_put(rle_enc *r, uint16_t v) {
    if (r->out_n < r->out_cap) r->out[r->out_n++] = v; else r->overflow = 1;
}

/* rlecompressu16.go:15-22 */
static int rle_init(rle_enc *r, uint16_t max_value, uint16_t *out, size_t cap) {
    int depth = len16(max_value);
    if (depth == 0) return MICO_ERR_ARGS; /* Go: negative shift panics */
    r->mid_count = (uint16_t)((1 << (depth - 1)) - 1);
    r->out = out; r->out_n = 0; r->out_cap = cap; r->overflow = 0;
    r->b = (uint16_t *)malloc(sizeof(uint16_t) * 65540);
    if (!r->b) return MICO_ERR_NOMEM;
    r->bc = 0; r->same = 0;
    rle_put(r, max_value);
    return MICO_OK;
}

/* rlecompressu16.go:24-70 */
static void rle_encode(rle_enc *r, uint16_t symbol) {
    int bc = r->bc;
    if (bc < 2) { r->b[r->bc++] = symbol; return; }
    uint16_t prev_plus_one = r->b[bc - 2], prev = r->b[bc - 1];
    if (prev_plus_one == prev && prev == symbol) {
        if (!r->same && bc > 2) {
            rle_put(r, (uint16_t)(r->mid_count + (uint16_t)(bc - 2)));
            for (int i = 0; i < bc - 2; i++) rle_put(r, r->b[i]);
            r->b[0] = r->b[bc - 2]; r->b[1] = r->b[bc - 1]; r->bc = 2;
        }
        r->same = 1;
    } else {
        if (r->same && bc > 2) {
            rle_put(r, (uint16_t)bc);
            rle_put(r, r->b[0]);
            r->bc = 0;
        }
        r->same = 0;
    }
    bc = r->bc;
    if (bc >= (int)(uint16_t)(r->mid_count - 1)) {
        if (r->same) {
            rle_put(r, (uint16_t)(bc - 2));
            rle_put(r, r->b[0]);
        } else {
            rle_put(r, (uint16_t)(r->mid_count + (uint16_t)(bc - 2)));
            for (int i = 0; i < bc - 2; i++) rle_put(r, r->b[i]);
        }
        r->b[0] = r->b[bc - 2]; r->b[1] = r->b[bc - 1]; r->bc = 2;
    }
    r->b[r->bc++] = symbol;
}

/* rlecompressu16.go:72-83 */
static void rle_flush(rle_enc *r) {
    int bc = r->bc;
    if (bc > 0) {
        if (r->same) {
            rle_put(r, (uint16_t)bc);
            rle_put(r, r->b[0]);
        } else {
            rle_put(r, (uint16_t)(r->mid_count + (uint16_t)bc));
            for (int i = 0; i < bc; i++) rle_put(r, r->b[i]);
        }
    }
}

static int rle_done(rle_enc *r, size_t *out_n) {
    free(r->b); r->b = NULL;
    if (r->overflow) return MICO_ERR_CAPACITY;
    *out_n = r->out_n;
    return MICO_OK;
}

/* rlecompressu16.go:85-93 (used by wavelet + temporal residual paths) */
int mico_rle_compress(const uint16_t *in, size_t n, uint16_t max_value,
                      uint16_t *out, size_t cap, size_t *out_n) {
    rle_enc r;
    if (len16(max_value) < 4) return MICO_ERR_ARGS; /* degenerate chunking, see delta_walk */
    int rc = rle_init(&r, max_value, out, cap);
    if (rc) return rc;
    rle_put(&r, (uint16_t)(n >> 16));
    rle_put(&r, (uint16_t)n);
    for (size_t i = 0; i < n; i++) rle_encode(&r, in[i]);
    rle_flush(&r);
    return rle_done(&r, out_n);
}

/* ------------------------------------------------ RLE pull-decoder (decode) */
/* rledecompressu16.go:10-30, :59-85 */
typedef struct {
    const uint16_t *in; size_t n, i;
    uint16_t mid_count, c, recurring;
    int err;
} rle_dec;

static int rle_dec_init(rle_dec *r, const uint16_t *in, size_t n) {
    if (n < 1) return MICO_ERR_CORRUPT;
    int depth = len16(in[0]);
    if (depth == 0) return MICO_ERR_CORRUPT;
    r->in = in; r->n = n; r->i = 1; r->c = 0; r->recurring = 0; r->err = 0;
    r->mid_count = (uint16_t)((1 << (depth - 1)) - 1);
    return MICO_OK;
}
static uint16_t rle_in(rle_dec *r) {
    if (r->i >= r->n) { r->err = 1; return 0; } /* Go: index panic */
    return r->in[r->i++];
}
/* rledecompressu16.go:59-85 */
static uint16_t rle_next2(rle_dec *r) {
    if (r->c > 0 && r->c < r->mid_count) { r->c--; return r->recurring; }
    if (r->c == 0 || r->c == r->mid_count) {
        r->c = rle_in(r);
        if (r->c <= r->mid_count) {
            r->recurring = rle_in(r);
            r->c--;
            return r->recurring;
        }
    }
    uint16_t v = rle_in(r);
    r->c--;
    return v;
}

/* rledecompressu16.go:87-97 */
int mico_rle_decompress(const uint16_t *in, size_t n, uint16_t *out, size_t cap,
                        size_t *out_n) {
    rle_dec r;
    int rc = rle_dec_init(&r, in, n);
    if (rc) return rc;
    if (n < 3) return MICO_ERR_CORRUPT;
    uint32_t outlen = ((uint32_t)in[1] << 16) + (uint32_t)in[2];
    r.i = 3;
    if (outlen > cap) return MICO_ERR_CAPACITY;
    for (uint32_t k = 0; k < outlen; k++) {
        out[k] = rle_next2(&r);
        if (r.err) return MICO_ERR_CORRUPT;
    }
    *out_n = outlen;
    return MICO_OK;
}

/* ------------------------------------------------------- delta(avg) stage */

typedef void (*sym_sink)(void *ctx, uint16_t sym);

/* gradPredict, deltagradcompressu16.go:147-167: avg(W, N) + clamp((NE - NW) >> 3, +-(|W - NW| + |N - NW|) / 2) */
static int32_t iabs32(int32_t v) { int32_t m = v >> 31; return (v ^ m) - m; }
static int32_t grad_predict(int32_t w, int32_t n, int32_t nw, int32_t ne) {
    int32_t avg = (w + n) >> 1;
    int32_t g = iabs32(w - nw) + iabs32(n - nw);
    if (g == 0) return avg;
    int32_t corr = (ne - nw) >> 3, limit = g >> 1;
    if (corr > limit) corr = limit; else if (corr < -limit) corr = -limit;
    return avg + corr;
}
/* the prediction of pixel (x, y) from pixels already known in px: pred 0 = avg(left, top) (deltarlecompressu16.go:33-46),
 * pred 1 = gradient-adaptive (deltagradrlecompressu16.go:36-53: 0 at the corner, left on row 0, top in column 0, NE = NW at the
 * right edge) */
static int32_t predict_px(const uint16_t *px, int width, int x, int y, int pred) {
    size_t index = (size_t)y * (size_t)width + (size_t)x;
    if (pred == 0) {
        int div = 0;
        int32_t prev = 0;
        if (x > 0) { prev = (int32_t)px[index - 1]; div++; }
        if (y > 0) { prev += (int32_t)px[index - (size_t)width]; div++; }
        if (div == 2) prev >>= 1;
        return prev;
    }
    if (x == 0 && y == 0) return 0;
    if (y == 0) return (int32_t)px[index - 1];
    if (x == 0) return (int32_t)px[index - (size_t)width];
    int32_t w = px[index - 1], n = px[index - (size_t)width], nw = px[index - (size_t)width - 1];
    int32_t ne = (x + 1 < width) ? (int32_t)px[index - (size_t)width + 1] : nw;
    return grad_predict(w, n, nw, ne);
}

/* deltarlecompressu16.go:24-61 / deltagradrlecompressu16.go:26-68: predictor + threshold/escape, symbol by symbol */
static int delta_walk_pred(const uint16_t *in, int width, int height,
                           uint16_t max_value, int pred, sym_sink sink, void *ctx) {
    int depth = len16(max_value);
    /* depth < 4: the Go tokeniser panics or emits empty chunks (rlecompressu16.go:57-67) */
    if (depth < 4 || width <= 0 || height <= 0) return MICO_ERR_ARGS;
    uint16_t thr = (uint16_t)((1 << (depth - 1)) - 1);
    uint16_t delim = (uint16_t)((1 << depth) - 1);
    sink(ctx, max_value);
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            size_t index = (size_t)y * (size_t)width + (size_t)x;
            int32_t prev = predict_px(in, width, x, y, pred);
            uint16_t v = in[index];
            int32_t diff = (int32_t)v - prev;
            int32_t mask = diff >> 31; /* deltacompressu16.go:122-126 abs */
            int32_t ad = (diff ^ mask) - mask;
            if ((uint16_t)ad >= thr) {
                sink(ctx, delim);
                sink(ctx, v);
            } else {
                sink(ctx, (uint16_t)((int32_t)thr + diff));
            }
        }
    }
    return MICO_OK;
}

static int delta_walk(const uint16_t *in, int width, int height, uint16_t max_value, sym_sink sink, void *ctx) {
    return delta_walk_pred(in, width, height, max_value, 0, sink, ctx);
}

static void sink_rle(void *ctx, uint16_t s) { rle_encode((rle_enc *)ctx, s); }

/* deltarlecompressu16.go:24-67 (pred 0), deltagradrlecompressu16.go:26-68 (pred 1) */
static int delta_rle_compress_pred(const uint16_t *px, int w, int h, uint16_t max_value, int pred,
                                   uint16_t *out, size_t cap, size_t *out_n) {
    int depth = len16(max_value);
    if (depth == 0) return MICO_ERR_ARGS;
    uint16_t delim = (uint16_t)((1 << depth) - 1);
    rle_enc r;
    int rc = rle_init(&r, delim, out, cap); /* Out[0] = delim, raw */
    if (rc) return rc;
    rc = delta_walk_pred(px, w, h, max_value, pred, sink_rle, &r);
    if (rc) { free(r.b); return rc; }
    rle_flush(&r);
    return rle_done(&r, out_n);
}
int mico_grad_delta_rle_compress(const uint16_t *px, int w, int h, uint16_t max_value,
                                 uint16_t *out, size_t cap, size_t *out_n) {
    return delta_rle_compress_pred(px, w, h, max_value, 1, out, cap, out_n);
}

/* deltarlecompressu16.go:24-67 */
int mico_delta_rle_compress(const uint16_t *px, int w, int h, uint16_t max_value,
                            uint16_t *out, size_t cap, size_t *out_n) {
    int depth = len16(max_value);
    if (depth == 0) return MICO_ERR_ARGS;
    uint16_t delim = (uint16_t)((1 << depth) - 1);
    rle_enc r;
    int rc = rle_init(&r, delim, out, cap); /* Out[0] = delim, raw */
    if (rc) return rc;
    rc = delta_walk(px, w, h, max_value, sink_rle, &r);
    if (rc) { free(r.b); return rc; }
    rle_flush(&r);
    return rle_done(&r, out_n);
}

typedef struct { uint16_t *out; size_t n, cap; } vec_sink;
static void sink_vec(void *ctx, uint16_t s) {
    vec_sink *v = (vec_sink *)ctx;
    if (v->n < v->cap) v->out[v->n] = s;
    v->n++;
}
int mico_delta_symbols(const uint16_t *px, int w, int h, uint16_t max_value,
                       uint16_t *out, size_t cap, size_t *out_n) {
    vec_sink v = { out, 0, cap };
    int rc = delta_walk(px, w, h, max_value, sink_vec, &v);
    if (rc) return rc;
    if (v.n > cap) return MICO_ERR_CAPACITY;
    *out_n = v.n;
    return MICO_OK;
}

/* deltarlecompressu16.go:69-128 (pred 0), deltagradrlecompressu16.go:70-133 (pred 1) */
static int delta_rle_decompress_pred(const uint16_t *in, size_t n, int width, int height, int pred,
                                     uint16_t *out) {
    rle_dec r;
    int rc = rle_dec_init(&r, in, n);
    if (rc) return rc;
    uint16_t max_value = rle_next2(&r);
    if (r.err) return MICO_ERR_CORRUPT;
    int depth = len16(max_value);
    if (depth == 0) return MICO_ERR_CORRUPT;
    uint16_t thr = (uint16_t)((1 << (depth - 1)) - 1);
    uint16_t delim = (uint16_t)((1 << depth) - 1);
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            size_t index = (size_t)y * (size_t)width + (size_t)x;
            uint16_t v = rle_next2(&r);
            if (v == delim) {
                out[index] = rle_next2(&r);
            } else {
                int32_t diff = (int32_t)v - (int32_t)thr;
                out[index] = (uint16_t)(predict_px(out, width, x, y, pred) + diff);
            }
            if (r.err) return MICO_ERR_CORRUPT;
        }
    }
    return MICO_OK;
}

int mico_delta_rle_decompress(const uint16_t *in, size_t n, int width, int height, uint16_t *out) {
    return delta_rle_decompress_pred(in, n, width, height, 0, out);
}
int mico_grad_delta_rle_decompress(const uint16_t *in, size_t n, int width, int height, uint16_t *out) {
    return delta_rle_decompress_pred(in, n, width, height, 1, out);
}

/* ===================================================================== FSE */

typedef struct { int32_t delta_find_state; uint32_t delta_nb_bits; } sym_tt;

typedef struct {
    uint32_t *count;       /* [65536]  fseu16.go:64 */
    int32_t  *norm;        /* [65536]  fseu16.go:65 */
    size_t    n;           /* s.br.remain(): number of input symbols */
    uint32_t  symbol_len;
    uint8_t   table_log;   /* actualTableLog */
    uint8_t   req_table_log; /* s.TableLog (fseu16.go:101-102); 0 = defaultTablelog */
    int       zero_bits;
    /* compression tables (fseu16.go:54-59) */
    uint16_t *table_symbol;
    uint32_t *state_table;
    sym_tt   *symbol_tt;
} fse_enc;

static void fse_enc_free(fse_enc *s) {
    free(s->count); free(s->norm); free(s->table_symbol); free(s->state_table);
    free(s->symbol_tt);
    memset(s, 0, sizeof(*s));
}

/* fsecompressu16.go:438-462 (histogram + symbolLen + max) */
static uint32_t count_simple(fse_enc *s, const uint16_t *in, size_t n) {
    for (size_t i = 0; i < n; i++) s->count[in[i]]++;
    uint32_t sym_len = 0, m = 0;
    for (uint32_t j = MAXSYM + 1; j > 0; j--) {
        uint32_t c = s->count[j - 1];
        if (c != 0) {
            if (sym_len == 0) sym_len = j;
            if (c > m) m = c;
        }
    }
    s->symbol_len = sym_len;
    return m;
}

/* fsecompressu16.go:465-518 */
static void optimal_table_log(fse_enc *s) {
    uint8_t table_log = s->req_table_log ? s->req_table_log : DEFAULT_TABLELOG; /* s.TableLog, default fseu16.go:133-135 */
    uint32_t min_bits_src = high_bits((uint32_t)(s->n - 1)) + 1;
    uint32_t min_bits_sym = high_bits(s->symbol_len - 1) + 2;
    uint8_t min_bits = (uint8_t)(min_bits_src < min_bits_sym ? min_bits_src : min_bits_sym);
    uint8_t max_bits_src = (uint8_t)((uint8_t)high_bits((uint32_t)(s->n - 1)) - 2);
    if (max_bits_src < table_log) table_log = max_bits_src;
    if (min_bits > table_log) table_log = min_bits;
    uint32_t density = (uint32_t)s->n / s->symbol_len;
    if (s->symbol_len > 512 && density > 16 && table_log < 13) table_log = 13;
    else if (density > 64 && s->symbol_len > 256 && table_log < 12) table_log = 12;
    else if (density > 32 && s->symbol_len > 128 && table_log < 12) table_log = 12;
    if (max_bits_src < table_log) table_log = max_bits_src;
    if (table_log < MIN_TABLELOG) table_log = MIN_TABLELOG;
    if (table_log > MAX_TABLELOG) table_log = MAX_TABLELOG;
    s->table_log = table_log;
}

static const uint32_t rtb_table[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };

/* fsecompressu16.go:575-667 */
static int normalize_count2(fse_enc *s) {
    const int32_t not_yet = -2;
    uint32_t distributed = 0;
    uint32_t total = (uint32_t)s->n;
    uint8_t tl = s->table_log;
    uint32_t low_threshold = total >> tl;
    uint32_t low_one = (total * 3) >> (tl + 1);
    for (uint32_t i = 0; i < s->symbol_len; i++) {
        uint32_t cnt = s->count[i];
        if (cnt == 0) { s->norm[i] = 0; continue; }
        if (cnt <= low_threshold) { s->norm[i] = -1; distributed++; total -= cnt; continue; }
        if (cnt <= low_one) { s->norm[i] = 1; distributed++; total -= cnt; continue; }
        s->norm[i] = not_yet;
    }
    /* distributed >= 2^tl: Go divides by zero (==) or wraps to_distribute and then spins
     * forever in the total==0 loop below (>) -- no behaviour to match, report it. */
    if (distributed >= ((uint32_t)1 << tl)) return MICO_ERR_INTERNAL;
    uint32_t to_distribute = ((uint32_t)1 << tl) - distributed;
    if ((total / to_distribute) > low_one) {
        low_one = (total * 3) / (to_distribute * 2);
        for (uint32_t i = 0; i < s->symbol_len; i++) {
            uint32_t cnt = s->count[i];
            if (s->norm[i] == not_yet && cnt <= low_one) {
                s->norm[i] = 1; distributed++; total -= cnt;
            }
        }
        if (distributed >= ((uint32_t)1 << tl)) return MICO_ERR_INTERNAL;
        to_distribute = ((uint32_t)1 << tl) - distributed;
    }
    if (distributed == s->symbol_len + 1) {
        uint32_t max_v = 0, max_c = 0;
        for (uint32_t i = 0; i < s->symbol_len; i++)
            if (s->count[i] > max_c) { max_v = i; max_c = s->count[i]; }
        s->norm[max_v] += (int32_t)to_distribute;
        return MICO_OK;
    }
    if (total == 0) {
        int any = 0;
        for (uint32_t i = 0; i < s->symbol_len; i++) if (s->norm[i] > 0) { any = 1; break; }
        if (!any) return MICO_ERR_INTERNAL; /* Go: endless loop */
        for (uint32_t i = 0; to_distribute > 0; i = (i + 1) % s->symbol_len)
            if (s->norm[i] > 0) { to_distribute--; s->norm[i]++; }
        return MICO_OK;
    }
    uint64_t v_step_log = 62 - (uint64_t)tl;
    uint64_t mid = ((uint64_t)1 << (v_step_log - 1)) - 1;
    uint64_t r_step = ((((uint64_t)1 << v_step_log) * (uint64_t)to_distribute) + mid) / (uint64_t)total;
    uint64_t tmp_total = mid;
    for (uint32_t i = 0; i < s->symbol_len; i++) {
        if (s->norm[i] == not_yet) {
            uint64_t end = tmp_total + (uint64_t)s->count[i] * r_step;
            uint32_t s_start = (uint32_t)(tmp_total >> v_step_log);
            uint32_t s_end = (uint32_t)(end >> v_step_log);
            uint32_t weight = s_end - s_start;
            if (weight < 1) return MICO_ERR_INTERNAL; /* "weight < 1" */
            s->norm[i] = (int32_t)weight;
            tmp_total = end;
        }
    }
    return MICO_OK;
}

/* fsecompressu16.go:524-571 */
static int normalize_count(fse_enc *s) {
    uint8_t tl = s->table_log;
    uint64_t scale = 62 - (uint64_t)tl;
    uint64_t step = ((uint64_t)1 << 62) / (uint64_t)s->n;
    uint64_t v_step = (uint64_t)1 << (scale - 20);
    int32_t still = (int32_t)1 << tl;
    uint32_t largest = 0;
    int32_t largest_p = 0;
    uint32_t low_threshold = (uint32_t)(s->n >> tl);
    for (uint32_t i = 0; i < s->symbol_len; i++) {
        uint32_t cnt = s->count[i];
        if (cnt == 0) { s->norm[i] = 0; continue; }
        if (cnt <= low_threshold) {
            s->norm[i] = -1;
            still--;
        } else {
            int32_t proba = (int32_t)(((uint64_t)cnt * step) >> scale);
            if (proba < 8) {
                uint64_t rest_to_beat = v_step * (uint64_t)rtb_table[proba];
                uint64_t v = (uint64_t)cnt * step - ((uint64_t)proba << scale);
                if (v > rest_to_beat) proba++;
            }
            if (proba > largest_p) { largest_p = proba; largest = i; }
            s->norm[i] = proba;
            still -= proba;
        }
    }
    if (-still >= (s->norm[largest] >> 1)) return normalize_count2(s);
    s->norm[largest] += still;
    return MICO_OK;
}

/* fsecompressu16.go:191-289; writes into out[*pos...] */
static int write_count(fse_enc *s, uint8_t *out, size_t cap, size_t *pos) {
    uint8_t tl = s->table_log;
    int table_size = 1 << tl;
    int previous0 = 0;
    uint32_t charnum = 0;
    size_t max_header = (((size_t)s->symbol_len * (size_t)tl) >> 3) + 3;
    uint32_t bit_stream = (uint32_t)(tl - MIN_TABLELOG);
    unsigned bit_count = 4;
    int32_t remaining = (int32_t)(table_size + 1);
    int32_t threshold = (int32_t)table_size;
    unsigned nb_bits = (unsigned)tl + 1;
    if (cap < *pos + max_header + 2) return MICO_ERR_CAPACITY;
    uint8_t *o = out + *pos;
    size_t out_p = 0;
    while (remaining > 1) {
        if (previous0) {
            uint32_t start = charnum;
            while (s->norm[charnum] == 0) {
                charnum++;
                if (charnum > MAXSYM) return MICO_ERR_INTERNAL;
            }
            while (charnum >= start + 24) {
                start += 24;
                bit_stream += (uint32_t)0xFFFF << bit_count;
                o[out_p] = (uint8_t)bit_stream;
                o[out_p + 1] = (uint8_t)(bit_stream >> 8);
                out_p += 2;
                bit_stream >>= 16;
            }
            while (charnum >= start + 3) {
                start += 3;
                bit_stream += (uint32_t)3 << bit_count;
                bit_count += 2;
            }
            bit_stream += (uint32_t)(charnum - start) << bit_count;
            bit_count += 2;
            if (bit_count > 16) {
                o[out_p] = (uint8_t)bit_stream;
                o[out_p + 1] = (uint8_t)(bit_stream >> 8);
                out_p += 2;
                bit_stream >>= 16;
                bit_count -= 16;
            }
        }
        int32_t count = s->norm[charnum];
        charnum++;
        int32_t max = (2 * threshold - 1) - remaining;
        if (count < 0) remaining += count; else remaining -= count;
        count++;
        if (count >= threshold) count += max;
        bit_stream += (uint32_t)count << bit_count;
        bit_count += nb_bits;
        if (count < max) bit_count--;
        previous0 = (count == 1);
        if (remaining < 1) return MICO_ERR_INTERNAL;
        while (remaining < threshold) { nb_bits--; threshold >>= 1; }
        if (bit_count > 16) {
            o[out_p] = (uint8_t)bit_stream;
            o[out_p + 1] = (uint8_t)(bit_stream >> 8);
            out_p += 2;
            bit_stream >>= 16;
            bit_count -= 16;
        }
    }
    o[out_p] = (uint8_t)bit_stream;
    o[out_p + 1] = (uint8_t)(bit_stream >> 8);
    out_p += (bit_count + 7) / 8;
    if (charnum > s->symbol_len) return MICO_ERR_INTERNAL;
    *pos += out_p;
    return MICO_OK;
}

/* fsecompressu16.go:329-431 */
static int build_ctable(fse_enc *s) {
    uint8_t tl = s->table_log;
    uint32_t table_size = (uint32_t)1 << tl;
    uint32_t high_threshold = table_size - 1;
    int32_t *cumul = (int32_t *)calloc(MAXSYM + 3, sizeof(int32_t));
    if (!cumul) return MICO_ERR_NOMEM;
    s->table_symbol = (uint16_t *)malloc(sizeof(uint16_t) * table_size);
    s->state_table = (uint32_t *)malloc(sizeof(uint32_t) * table_size);
    uint32_t tt_size = s->symbol_len < 256 ? 256 : s->symbol_len;
    s->symbol_tt = (sym_tt *)calloc(tt_size, sizeof(sym_tt));
    if (!s->table_symbol || !s->state_table || !s->symbol_tt) { free(cumul); return MICO_ERR_NOMEM; }
    cumul[0] = 0;
    for (uint32_t u = 0; u < s->symbol_len; u++) {
        int32_t v = s->norm[u];
        if (v == -1) {
            cumul[u + 1] = cumul[u] + 1;
            s->table_symbol[high_threshold] = (uint16_t)u;
            high_threshold--;
        } else {
            cumul[u + 1] = cumul[u] + v;
        }
    }
    if ((uint32_t)cumul[s->symbol_len] != table_size) { free(cumul); return MICO_ERR_INTERNAL; }
    cumul[s->symbol_len] = (int32_t)table_size + 1;
    /* spread */
    s->zero_bits = 0;
    {
        uint32_t step = table_step(table_size), mask = table_size - 1, position = 0;
        int32_t large_limit = (int32_t)1 << (tl - 1);
        for (uint32_t u = 0; u < s->symbol_len; u++) {
            int32_t v = s->norm[u];
            if (v > large_limit) s->zero_bits = 1;
            for (int32_t k = 0; k < v; k++) {
                s->table_symbol[position] = (uint16_t)u;
                position = (position + step) & mask;
                while (position > high_threshold) position = (position + step) & mask;
            }
        }
        if (position != 0) { free(cumul); return MICO_ERR_INTERNAL; }
    }
    for (uint32_t u = 0; u < table_size; u++) {
        uint16_t v = s->table_symbol[u];
        s->state_table[cumul[v]] = table_size + u;
        cumul[v]++;
    }
    {
        int32_t total = 0;
        uint32_t tlv = ((uint32_t)tl << 16) - ((uint32_t)1 << tl);
        for (uint32_t i = 0; i < s->symbol_len; i++) {
            int32_t v = s->norm[i];
            if (v == 0) continue;
            if (v == -1 || v == 1) {
                s->symbol_tt[i].delta_nb_bits = tlv;
                s->symbol_tt[i].delta_find_state = total - 1;
                total++;
            } else {
                uint32_t max_bits_out = (uint32_t)tl - high_bits((uint32_t)(v - 1));
                uint32_t min_state_plus = (uint32_t)v << max_bits_out;
                s->symbol_tt[i].delta_nb_bits = (max_bits_out << 16) - min_state_plus;
                s->symbol_tt[i].delta_find_state = total - v;
                total += v;
            }
        }
        if (total != (int32_t)table_size) { free(cumul); return MICO_ERR_INTERNAL; }
    }
    free(cumul);
    return MICO_OK;
}

/* cStateU16.encode, fsecompressu16.go:95-100 */
static void cstate_encode(const fse_enc *s, bitw *bw, uint32_t *state, uint16_t sym) {
    sym_tt t = s->symbol_tt[sym];
    uint32_t nb = (*state + t.delta_nb_bits) >> 16;
    int32_t dst = (int32_t)(*state >> (nb & 31)) + t.delta_find_state;
    bw_add(bw, *state, nb);
    *state = s->state_table[dst];
}

/* compress (1-state, fsecompressu16.go:109-187), compress2State (fse2state.go:122-199),
 * compress4State (fse4state.go:100-191), compress8State (fse8state.go:113-226):
 * symbol i belongs to lane i % N; symbols are encoded from the last to the first;
 * final states are written lane N-1 first ... lane 0 last; then the end mark. */
static void compress_nstate(const fse_enc *s, const uint16_t *src, size_t n, int nstates, bitw *bw) {
    uint32_t st[8];
    for (int k = 0; k < nstates; k++) st[k] = (uint32_t)1 << s->table_log;
    for (size_t ip = n; ip > 0; ip--) {
        size_t idx = ip - 1;
        cstate_encode(s, bw, &st[idx % (size_t)nstates], src[idx]);
    }
    for (int k = nstates - 1; k >= 0; k--) bw_add(bw, st[k], s->table_log);
    bw_close(bw);
}

static int fse_enc_prepare(fse_enc *s, const uint16_t *in, size_t n) {
    memset(s, 0, sizeof(*s));
    s->count = (uint32_t *)calloc(MAXSYM + 1, sizeof(uint32_t));
    s->norm = (int32_t *)calloc(MAXSYM + 1, sizeof(int32_t));
    if (!s->count || !s->norm) { fse_enc_free(s); return MICO_ERR_NOMEM; }
    s->n = n;
    (void)in;
    return MICO_OK;
}

int mico_fse_normalize(const uint16_t *in, size_t n, int32_t *norm, mico_fse_info *info) {
    fse_enc s;
    int rc = fse_enc_prepare(&s, in, n);
    if (rc) return rc;
    uint32_t max_count = count_simple(&s, in, n);
    optimal_table_log(&s);
    rc = normalize_count(&s);
    if (rc == MICO_OK) {
        memcpy(norm, s.norm, sizeof(int32_t) * (MAXSYM + 1));
        info->symbol_len = s.symbol_len; info->max_count = max_count;
        info->table_log = s.table_log;
        rc = build_ctable(&s);
        info->zero_bits = (uint8_t)s.zero_bits;
    }
    fse_enc_free(&s);
    return rc;
}


/* FSECompressU16 / TwoState / FourState / EightState and RANSCompressU16EightState
 * share one gate sequence: fsecompressu16.go:19-78, fse2state.go:22-69,
 * fse4state.go:24-69, fse8state.go:31-77, rans8state.go:31-84. */
int mico_fse_compress(const uint16_t *in, size_t n, int nstates,
                      uint8_t *out, size_t cap, size_t *out_len) {
    return mico_fse_compress_tl(in, n, nstates, 0, out, cap, out_len);
}

/* the same with ScratchU16.TableLog set by the caller (fseu16.go:101-102, :133-138: 0 = default 11, > 16 is an error) */
int mico_fse_compress_tl(const uint16_t *in, size_t n, int nstates, int table_log,
                         uint8_t *out, size_t cap, size_t *out_len) {
    int lanes = (nstates == 108) ? 8 : nstates;
    if (table_log < 0 || table_log > MAX_TABLELOG) return MICO_ERR_ARGS;   /* "tableLog (%d) > maxTableLog (%d)" */
    if (!(lanes == 1 || lanes == 2 || lanes == 4 || lanes == 8)) return MICO_ERR_ARGS;
    if (n <= (size_t)(lanes - 1) || n <= 1) return MICO_ERR_INCOMPRESSIBLE;
    if (n > ((size_t)2 << 30) - 1) return MICO_ERR_ARGS;
    fse_enc s;
    int rc = fse_enc_prepare(&s, in, n);
    if (rc) return rc;
    s.req_table_log = (uint8_t)table_log;
    uint32_t max_count = count_simple(&s, in, n);
    if ((size_t)max_count == n) { fse_enc_free(&s); return MICO_ERR_USE_RLE; }
    if (max_count == 1 || (size_t)max_count < (n >> 15)) { fse_enc_free(&s); return MICO_ERR_INCOMPRESSIBLE; }
    optimal_table_log(&s);
    rc = normalize_count(&s);
    if (rc) { fse_enc_free(&s); return rc; }
    size_t pos = (lanes == 1) ? 0 : 6;
    if (cap < pos + 8) { fse_enc_free(&s); return MICO_ERR_CAPACITY; }
    rc = write_count(&s, out, cap, &pos);
    if (rc) { fse_enc_free(&s); return rc; }
    size_t hdr_end = pos;
    bitw bw = { out + pos, 0, cap - pos, 0, 0, 0 };
    if (nstates == 108) {
        rc = mico_rans_compress8(in, n, s.norm, s.symbol_len, s.table_log, &bw);
        if (rc) { fse_enc_free(&s); return rc; }
    } else {
        rc = build_ctable(&s);
        if (rc) { fse_enc_free(&s); return rc; }
        if (n <= 2 && lanes <= 2) { fse_enc_free(&s); return MICO_ERR_INTERNAL; } /* "src too small" */
        compress_nstate(&s, in, n, lanes, &bw);
    }
    fse_enc_free(&s);
    if (bw.overflow) return MICO_ERR_CAPACITY;
    size_t body = (hdr_end - ((lanes == 1) ? 0 : 6)) + bw.len; /* len(s.Out): header + bitstream */
    if (body >= n * 2) return MICO_ERR_INCOMPRESSIBLE;
    if (lanes != 1) {
        out[0] = 0xFF;
        out[1] = (nstates == 2) ? 0x02 : (nstates == 4) ? 0x04 : (nstates == 8) ? 0x84 : 0x08;
        out[2] = (uint8_t)n; out[3] = (uint8_t)(n >> 8);
        out[4] = (uint8_t)(n >> 16); out[5] = (uint8_t)(n >> 24);
    }
    *out_len = hdr_end + bw.len;
    return MICO_OK;
}

/* ------------------------------------------------------------- FSE decode */

typedef struct {
    int32_t  *norm;        /* [65536] */
    uint32_t  symbol_len;
    uint8_t   table_log;
    int       zero_bits;
    dec_sym  *dt;
} fse_dec;

static void fse_dec_free(fse_dec *d) { free(d->norm); free(d->dt); memset(d, 0, sizeof(*d)); }

/* byteReader.Uint32 (bytereader.go:31-40); Go panics past the end -> we flag */
static uint32_t rd_u32(const uint8_t *b, size_t len, size_t off, int *err) {
    if (off + 4 > len) { *err = 1; return 0; }
    return (uint32_t)b[off] | ((uint32_t)b[off + 1] << 8) | ((uint32_t)b[off + 2] << 16) | ((uint32_t)b[off + 3] << 24);
}

/* fsedecompressu16.go:48-167.  b/len is the stream after any 6-byte prefix;
 * *consumed = bytes used by the NCount header. */
static int read_ncount(fse_dec *d, const uint8_t *b, size_t len, size_t *consumed) {
    uint32_t charnum = 0;
    int previous0 = 0;
    int err = 0;
    long iend = (long)len;
    long off = 0;
    if (iend < 4) return MICO_ERR_CORRUPT;
    uint32_t bit_stream = rd_u32(b, len, 0, &err);
    unsigned nb_bits = (bit_stream & 0xF) + MIN_TABLELOG;
    if (nb_bits > 17) return MICO_ERR_CORRUPT;
    bit_stream >>= 4;
    unsigned bit_count = 4;
    d->table_log = (uint8_t)nb_bits;
    int32_t remaining = (int32_t)((1 << nb_bits) + 1);
    int32_t threshold = (int32_t)(1 << nb_bits);
    int32_t got_total = 0;
    nb_bits++;
    while (remaining > 1) {
        if (previous0) {
            uint32_t n0 = charnum;
            while ((bit_stream & 0xFFFF) == 0xFFFF) {
                n0 += 24;
                if (off < iend - 5) {
                    off += 2;
                    bit_stream = rd_u32(b, len, (size_t)off, &err) >> bit_count;
                } else {
                    bit_stream >>= 16;
                    bit_count += 16;
                }
                if (n0 > MAXSYM + 24) return MICO_ERR_CORRUPT;
            }
            while ((bit_stream & 3) == 3) {
                n0 += 3;
                bit_stream >>= 2;
                bit_count += 2;
            }
            n0 += bit_stream & 3;
            bit_count += 2;
            if (n0 > MAXSYM) return MICO_ERR_CORRUPT;
            while (charnum < n0) { d->norm[charnum & 0xffff] = 0; charnum++; }
            if (off <= iend - 7 || off + (long)(bit_count >> 3) <= iend - 4) {
                off += (long)(bit_count >> 3);
                bit_count &= 7;
                bit_stream = rd_u32(b, len, (size_t)off, &err) >> bit_count;
            } else {
                bit_stream >>= 2;
            }
        }
        int32_t max = (2 * threshold - 1) - remaining;
        int32_t count;
        if (((int32_t)bit_stream & (threshold - 1)) < max) {
            count = (int32_t)bit_stream & (threshold - 1);
            bit_count += nb_bits - 1;
        } else {
            count = (int32_t)bit_stream & (2 * threshold - 1);
            if (count >= threshold) count -= max;
            bit_count += nb_bits;
        }
        count--;
        if (count < 0) { remaining += count; got_total -= count; }
        else { remaining -= count; got_total += count; }
        if (charnum > MAXSYM) return MICO_ERR_CORRUPT;
        d->norm[charnum & 0xffff] = count;
        charnum++;
        previous0 = (count == 0);
        while (remaining < threshold) {
            nb_bits--; threshold >>= 1;
            if (threshold == 0) break;
        }
        if (off <= iend - 7 || off + (long)(bit_count >> 3) <= iend - 4) {
            off += (long)(bit_count >> 3);
            bit_count &= 7;
        } else {
            bit_count -= (unsigned)(8 * (iend - 4 - off));
            off = iend - 4;
        }
        bit_stream = rd_u32(b, len, (size_t)off, &err) >> (bit_count & 31);
        if (err) return MICO_ERR_CORRUPT;
    }
    d->symbol_len = charnum;
    if (d->symbol_len <= 1) return MICO_ERR_CORRUPT;
    if (d->symbol_len > MAXSYM + 1) return MICO_ERR_CORRUPT;
    if (remaining != 1) return MICO_ERR_CORRUPT;
    if (bit_count > 32) return MICO_ERR_CORRUPT;
    if (got_total != (1 << d->table_log)) return MICO_ERR_CORRUPT;
    off += (long)((bit_count + 7) >> 3);
    if (off > iend) return MICO_ERR_CORRUPT;
    *consumed = (size_t)off;
    return MICO_OK;
}

/* fsedecompressu16.go:198-263 */
static int build_dtable(fse_dec *d) {
    uint32_t table_size = (uint32_t)1 << d->table_log;
    uint32_t high_threshold = table_size - 1;
    d->dt = (dec_sym *)calloc(table_size, sizeof(dec_sym));
    uint32_t *symbol_next = (uint32_t *)calloc(d->symbol_len < 256 ? 256 : d->symbol_len, sizeof(uint32_t));
    if (!d->dt || !symbol_next) { free(symbol_next); return MICO_ERR_NOMEM; }
    d->zero_bits = 0;
    int32_t large_limit = (int32_t)1 << (d->table_log - 1);
    for (uint32_t i = 0; i < d->symbol_len; i++) {
        int32_t v = d->norm[i];
        if (v == -1) {
            d->dt[high_threshold].symbol = (uint16_t)i;
            high_threshold--;
            symbol_next[i] = 1;
        } else {
            if (v >= large_limit) d->zero_bits = 1;
            symbol_next[i] = (uint32_t)v;
        }
    }
    {
        uint32_t mask = table_size - 1, step = table_step(table_size), position = 0;
        for (uint32_t ss = 0; ss < d->symbol_len; ss++) {
            int32_t v = d->norm[ss];
            for (int32_t i = 0; i < v; i++) {
                d->dt[position].symbol = (uint16_t)ss;
                position = (position + step) & mask;
                while (position > high_threshold) position = (position + step) & mask;
            }
        }
        if (position != 0) { free(symbol_next); return MICO_ERR_CORRUPT; }
    }
    for (uint32_t u = 0; u < table_size; u++) {
        uint16_t symbol = d->dt[u].symbol;
        uint32_t next_state = symbol_next[symbol];
        symbol_next[symbol] = next_state + 1;
        uint8_t n_bits = (uint8_t)(d->table_log - (uint8_t)high_bits(next_state));
        d->dt[u].nb_bits = n_bits;
        uint32_t new_state = (next_state << n_bits) - table_size;
        if (new_state >= table_size) { free(symbol_next); return MICO_ERR_CORRUPT; }
        if (new_state == u && n_bits == 0) { free(symbol_next); return MICO_ERR_CORRUPT; }
        d->dt[u].new_state = new_state;
    }
    free(symbol_next);
    return MICO_OK;
}


/* decompress (fsedecompressu16.go:267-377) */
static int decompress_1state(const fse_dec *d, const uint8_t *bits, size_t len,
                             uint16_t *out, size_t cap, size_t *out_n) {
    bitr br;
    int rc = br_init(&br, bits, len);
    if (rc) return rc;
    uint32_t state = br_get(&br, d->table_log);
    size_t n = 0;
    for (;;) {
        if (br_finished(&br) && d->dt[state].nb_bits > 0) { /* decoderU16.finished */
            if (state != 0) {
                if (n >= cap) return MICO_ERR_CAPACITY;
                out[n++] = d->dt[state].symbol;
            }
            break;
        }
        dec_sym e = d->dt[state];
        uint32_t low = br_get(&br, e.nb_bits);
        if (br.over) return MICO_ERR_CORRUPT;
        state = e.new_state + low;
        if (n >= cap) return MICO_ERR_CAPACITY;
        out[n++] = e.symbol;
    }
    *out_n = n;
    return MICO_OK;
}

/* decompress2State (fse2state.go:203-308), decompress4State (fse4state.go:195-353),
 * decompress8State (fse8state.go:230-380): states read lane 0 first; symbol i comes
 * from lane i % N; exactly `count` symbols. */
static int decompress_nstate(const fse_dec *d, const uint8_t *bits, size_t len, int nstates,
                             uint32_t count, uint16_t *out) {
    bitr br;
    int rc = br_init(&br, bits, len);
    if (rc) return rc;
    uint32_t st[8];
    for (int k = 0; k < nstates; k++) st[k] = br_get(&br, d->table_log);
    if (br.over) return MICO_ERR_CORRUPT;
    for (uint32_t i = 0; i < count; i++) {
        int k = (int)(i % (uint32_t)nstates);
        dec_sym e = d->dt[st[k]];
        uint32_t low = br_get(&br, e.nb_bits);
        if (br.over) return MICO_ERR_CORRUPT;
        st[k] = e.new_state + low;
        out[i] = e.symbol;
    }
    return MICO_OK;
}

/* FSEDecompressU16Auto, fse2state.go:102-116 */
int mico_fse_decompress_auto(const uint8_t *in, size_t len,
                             uint16_t *out, size_t cap, size_t *out_n) {
    int nstates = 1;
    if (len >= 2 && in[0] == 0xFF) {
        if (in[1] == 0x84) nstates = 8;
        else if (in[1] == 0x08) nstates = 108;
        else if (in[1] == 0x04) nstates = 4;
        else if (in[1] == 0x02) nstates = 2;
    }
    uint32_t count = 0;
    const uint8_t *b = in;
    size_t blen = len;
    if (nstates != 1) {
        if (len < 6) return MICO_ERR_CORRUPT;
        count = (uint32_t)in[2] | ((uint32_t)in[3] << 8) | ((uint32_t)in[4] << 16) | ((uint32_t)in[5] << 24);
        b = in + 6; blen = len - 6;
        if (count > cap) return MICO_ERR_CAPACITY;
    }
    fse_dec d;
    memset(&d, 0, sizeof(d));
    d.norm = (int32_t *)calloc(MAXSYM + 1, sizeof(int32_t));
    if (!d.norm) return MICO_ERR_NOMEM;
    size_t used = 0;
    int rc = read_ncount(&d, b, blen, &used);
    if (rc) { fse_dec_free(&d); return rc; }
    if (nstates == 108) {
        rc = mico_rans_decompress8(b + used, blen - used, d.norm, d.symbol_len, d.table_log, count, out);
        if (rc == MICO_OK) *out_n = count;
        fse_dec_free(&d);
        return rc;
    }
    rc = build_dtable(&d);
    if (rc) { fse_dec_free(&d); return rc; }
    if (nstates == 1) {
        rc = decompress_1state(&d, b + used, blen - used, out, cap, out_n);
    } else {
        rc = decompress_nstate(&d, b + used, blen - used, nstates, count, out);
        if (rc == MICO_OK) *out_n = count;
    }
    fse_dec_free(&d);
    return rc;
}

/* ============================================================== unit codec */

/* CompressSingleFrame / 4State / 8State, multiframecompress.go:15-93 */
static int fse_chain(const uint16_t *sym, size_t n, int nstates,
                     uint8_t *out, size_t cap, size_t *out_len) {
    static const int chain8[] = { 8, 4, 2, 1 }, chain4[] = { 4, 2, 1 }, chain2[] = { 2, 1 };
    const int *chain; int cn;
    if (nstates == 8) { chain = chain8; cn = 4; }
    else if (nstates == 4) { chain = chain4; cn = 3; }
    else if (nstates == 2) { chain = chain2; cn = 2; }
    else return MICO_ERR_ARGS;
    int rc = MICO_ERR_INTERNAL;
    for (int k = 0; k < cn; k++) {
        rc = mico_fse_compress(sym, n, chain[k], out, cap, out_len);
        if (rc == MICO_OK) return rc;
    }
    return rc; /* error of the 1-state attempt, as in the reference */
}

int mico_compress_single_frame(const uint16_t *px, int w, int h,
                               uint16_t max_value, int nstates,
                               uint8_t *out, size_t cap, size_t *out_len) {
    if (!px || w <= 0 || h <= 0) return MICO_ERR_ARGS;
    size_t scap = 4 * (size_t)w * (size_t)h + 16;
    uint16_t *sym = (uint16_t *)malloc(sizeof(uint16_t) * scap);
    if (!sym) return MICO_ERR_NOMEM;
    size_t n = 0;
    int rc = mico_delta_rle_compress(px, w, h, max_value, sym, scap, &n);
    if (rc == MICO_OK) rc = fse_chain(sym, n, nstates, out, cap, out_len);
    free(sym);
    return rc;
}

/* CompressSingleFrameGrad / DecompressSingleFrameGrad, multiframecompress.go:111-142: the gradient predictor in front of the same
 * two-state -> one-state FSE chain */
int mico_compress_single_frame_grad(const uint16_t *px, int w, int h, uint16_t max_value,
                                    uint8_t *out, size_t cap, size_t *out_len) {
    if (!px || w <= 0 || h <= 0) return MICO_ERR_ARGS;
    size_t scap = 4 * (size_t)w * (size_t)h + 16;
    uint16_t *sym = (uint16_t *)malloc(sizeof(uint16_t) * scap);
    if (!sym) return MICO_ERR_NOMEM;
    size_t n = 0;
    int rc = mico_grad_delta_rle_compress(px, w, h, max_value, sym, scap, &n);
    if (rc == MICO_OK) rc = fse_chain(sym, n, 2, out, cap, out_len);
    free(sym);
    return rc;
}
int mico_decompress_single_frame_grad(const uint8_t *in, size_t len, uint16_t *px, int w, int h) {
    if (!in || !px || w <= 0 || h <= 0) return MICO_ERR_ARGS;
    size_t scap = 4 * (size_t)w * (size_t)h + 16;
    uint16_t *sym = (uint16_t *)malloc(sizeof(uint16_t) * scap);
    if (!sym) return MICO_ERR_NOMEM;
    size_t n = 0;
    int rc = mico_fse_decompress_auto(in, len, sym, scap, &n);
    if (rc == MICO_OK) rc = mico_grad_delta_rle_decompress(sym, n, w, h, px);
    free(sym);
    return rc;
}

/* DecompressSingleFrame, multiframecompress.go:97-107 */
int mico_decompress_single_frame(const uint8_t *in, size_t len,
                                 uint16_t *px, int w, int h) {
    if (!in || !px || w <= 0 || h <= 0) return MICO_ERR_ARGS;
    size_t scap = 4 * (size_t)w * (size_t)h + 16;
    uint16_t *sym = (uint16_t *)malloc(sizeof(uint16_t) * scap);
    if (!sym) return MICO_ERR_NOMEM;
    size_t n = 0;
    int rc = mico_fse_decompress_auto(in, len, sym, scap, &n);
    if (rc == MICO_OK) rc = mico_delta_rle_decompress(sym, n, w, h, px);
    free(sym);
    return rc;
}

/* ==================================================================== PICS */

static void put_u32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* CompressParallelStrips{,4State,8State}, parallelstrips.go:55-265 */
int mico_pics_compress(const uint16_t *px, int w, int h, uint16_t max_value,
                       int num_strips, int nstates,
                       uint8_t *out, size_t cap, size_t *out_len) {
    if (!px || w <= 0 || h <= 0) return MICO_ERR_ARGS;
    if (num_strips <= 0) num_strips = 1; /* reference: GOMAXPROCS; callers pass it explicitly */
    if (num_strips > h) num_strips = h;
    if (num_strips < 1) num_strips = 1;
    int strip_h = (h + num_strips - 1) / num_strips;
    int actual = (h + strip_h - 1) / strip_h;
    size_t header = 20 + (size_t)actual * 8;
    if (cap < header) return MICO_ERR_CAPACITY;
    memcpy(out, "PICS", 4);
    put_u32(out + 4, (uint32_t)w); put_u32(out + 8, (uint32_t)h);
    put_u32(out + 12, (uint32_t)actual); put_u32(out + 16, (uint32_t)strip_h);
    size_t offset = 0;
    for (int s = 0; s < actual; s++) {
        int y0 = s * strip_h, y1 = y0 + strip_h;
        if (y1 > h) y1 = h;
        size_t blen = 0;
        int rc = mico_compress_single_frame(px + (size_t)y0 * (size_t)w, w, y1 - y0, max_value, nstates,
                                            out + header + offset, cap - header - offset, &blen);
        if (rc) return rc;
        put_u32(out + 20 + (size_t)s * 8, (uint32_t)offset);
        put_u32(out + 24 + (size_t)s * 8, (uint32_t)blen);
        offset += blen;
    }
    *out_len = header + offset;
    return MICO_OK;
}

/* DecompressParallelStrips, parallelstrips.go:270-330 */
int mico_pics_decompress(const uint8_t *in, size_t len, uint16_t *px,
                         size_t px_cap, int *w, int *h) {
    if (len < 20 || memcmp(in, "PICS", 4) != 0) return MICO_ERR_CORRUPT;
    int width = (int)get_u32(in + 4), height = (int)get_u32(in + 8);
    int num_strips = (int)get_u32(in + 12), strip_h = (int)get_u32(in + 16);
    if (num_strips < 0 || (size_t)num_strips > (len - 20) / 8) return MICO_ERR_CORRUPT;
    size_t header = 20 + (size_t)num_strips * 8;
    if (len < header) return MICO_ERR_CORRUPT;
    if (width <= 0 || height <= 0 || num_strips <= 0 || strip_h <= 0) return MICO_ERR_CORRUPT;
    *w = width; *h = height;
    if (!px) return MICO_OK;
    if ((size_t)width * (size_t)height > px_cap) return MICO_ERR_CAPACITY;
    memset(px, 0, (size_t)width * (size_t)height * 2);           /* out := make([]uint16, width*height), parallelstrips.go:288 */
    for (int s = 0; s < num_strips; s++) {
        size_t so = get_u32(in + 20 + (size_t)s * 8), sl = get_u32(in + 24 + (size_t)s * 8);
        size_t start = header + so, end = start + sl;
        if (end > len || start > end) return MICO_ERR_CORRUPT;
        long y0 = (long)s * strip_h, y1 = y0 + strip_h;
        if (y1 > height) y1 = height;
        if (y0 >= height) return MICO_ERR_CORRUPT; /* Go: slice panic */
        int rc = mico_decompress_single_frame(in + start, sl, px + (size_t)y0 * (size_t)width, width, (int)(y1 - y0));
        if (rc) return rc;
    }
    return MICO_OK;
}

/* ==================================================================== PICA */

/* adaptiveStripBoundaries, parallelstripsadaptive.go:222-289: equal-cost partition of the rows by summed |vertical delta|,
 * in float64 as the reference computes it.  starts[] has room for num_strips entries; returns the number of strips. */
int mico_pica_boundaries(const uint16_t *px, int w, int h, int num_strips, int *starts) {
    if (num_strips >= h) { for (int i = 0; i < h; i++) starts[i] = i; return h; }
    if (num_strips == 1) { starts[0] = 0; return 1; }
    double *cum = (double *)malloc(sizeof(double) * ((size_t)h + 1));
    if (!cum) return MICO_ERR_NOMEM;
    cum[0] = 0.0; cum[1] = 0.0;                                  /* rowCost[0] = 0 */
    for (int y = 1; y < h; y++) {
        uint64_t sum = 0;
        for (int x = 0; x < w; x++) {
            int32_t d = (int32_t)px[(size_t)y * w + x] - (int32_t)px[(size_t)(y - 1) * w + x];
            sum += (uint64_t)(d < 0 ? -d : d);
        }
        cum[y + 1] = cum[y] + (double)sum;
    }
    volatile double total = cum[h];
    starts[0] = 0;
    if (total == 0) {
        for (int i = 1; i < num_strips; i++) starts[i] = (int)((long long)i * h / num_strips);
    } else {
        for (int i = 1; i < num_strips; i++) {
            volatile double prod = total * (double)i;             /* Go: total * float64(i) / float64(numStrips), left to right */
            double target = prod / (double)num_strips;
            int lo = starts[i - 1] + 1, hi = h;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (cum[mid] < target) lo = mid + 1; else hi = mid; }
            if (lo >= h) lo = h - 1;
            starts[i] = lo;
        }
    }
    free(cum);
    return num_strips;
}

/* CompressParallelStripsAdaptive, parallelstripsadaptive.go:54-137: per strip both predictors, the smaller blob wins (ties: gradient) */
int mico_pica_compress(const uint16_t *px, int w, int h, uint16_t max_value, int num_strips,
                       uint8_t *out, size_t cap, size_t *out_len) {
    if (!px || w <= 0 || h <= 0) return MICO_ERR_ARGS;
    if (num_strips <= 0) num_strips = 1; /* reference: GOMAXPROCS; callers pass it explicitly */
    if (num_strips > h) num_strips = h;
    int *starts = (int *)malloc(sizeof(int) * (size_t)num_strips);
    if (!starts) return MICO_ERR_NOMEM;
    int actual = mico_pica_boundaries(px, w, h, num_strips, starts);
    if (actual < 0) { free(starts); return actual; }
    size_t header = 16 + (size_t)actual * 16;
    size_t scap = 4 * (size_t)w * (size_t)h + 262144;             /* room for the worst-case NCount header: no capacity errors of our own */
    uint8_t *ba = (uint8_t *)malloc(scap), *bg = (uint8_t *)malloc(scap);
    int rc = (ba && bg) ? MICO_OK : MICO_ERR_NOMEM;
    if (rc == MICO_OK && cap < header) rc = MICO_ERR_CAPACITY;
    size_t offset = 0;
    if (rc == MICO_OK) {
        memcpy(out, "PICA", 4);
        put_u32(out + 4, (uint32_t)w); put_u32(out + 8, (uint32_t)h); put_u32(out + 12, (uint32_t)actual);
    }
    for (int s = 0; s < actual && rc == MICO_OK; s++) {
        int y0 = starts[s], y1 = (s + 1 < actual) ? starts[s + 1] : h;
        size_t la = 0, lg = 0;
        int r1 = mico_compress_single_frame(px + (size_t)y0 * w, w, y1 - y0, max_value, 2, ba, scap, &la);
        int r2 = mico_compress_single_frame_grad(px + (size_t)y0 * w, w, y1 - y0, max_value, bg, scap, &lg);
        const uint8_t *pick; size_t plen; uint32_t flags;
        if (r2 == MICO_OK && (r1 != MICO_OK || lg <= la)) { pick = bg; plen = lg; flags = 1; }
        else { pick = ba; plen = la; flags = 0; rc = r1; }
        if (rc) break;
        if (header + offset + plen > cap) { rc = MICO_ERR_CAPACITY; break; }
        uint8_t *e = out + 16 + (size_t)s * 16;
        put_u32(e, (uint32_t)y0); put_u32(e + 4, (uint32_t)offset); put_u32(e + 8, (uint32_t)plen); put_u32(e + 12, flags);
        memcpy(out + header + offset, pick, plen);
        offset += plen;
    }
    free(ba); free(bg); free(starts);
    if (rc == MICO_OK) *out_len = header + offset;
    return rc;
}

/* DecompressParallelStripsAdaptive, parallelstripsadaptive.go:141-214 */
int mico_pica_decompress(const uint8_t *in, size_t len, uint16_t *px, size_t px_cap, int *w, int *h) {
    if (len < 16 || memcmp(in, "PICA", 4) != 0) return MICO_ERR_CORRUPT;
    int width = (int)get_u32(in + 4), height = (int)get_u32(in + 8), num_strips = (int)get_u32(in + 12);
    if (num_strips < 0 || (size_t)num_strips > (len - 16) / 16) return MICO_ERR_CORRUPT;
    size_t header = 16 + (size_t)num_strips * 16;
    if (width <= 0 || height <= 0 || num_strips <= 0) return MICO_ERR_CORRUPT;
    *w = width; *h = height;
    if (!px) return MICO_OK;
    if ((size_t)width * (size_t)height > px_cap) return MICO_ERR_CAPACITY;
    for (int s = 0; s < num_strips; s++) {
        const uint8_t *e = in + 16 + (size_t)s * 16;
        long y0 = (long)get_u32(e), y1 = (s + 1 < num_strips) ? (long)get_u32(e + 16) : height;
        size_t start = header + get_u32(e + 4), end = start + get_u32(e + 8);
        uint32_t flags = get_u32(e + 12);
        if (end > len || start > end) return MICO_ERR_CORRUPT;
        if (y0 < 0 || y1 <= y0 || y1 > height) return MICO_ERR_CORRUPT;   /* Go: slice / make panics */
        int rc = (flags & 1) ? mico_decompress_single_frame_grad(in + start, end - start, px + (size_t)y0 * width, width, (int)(y1 - y0))
                             : mico_decompress_single_frame(in + start, end - start, px + (size_t)y0 * width, width, (int)(y1 - y0));
        if (rc) return rc;
    }
    return MICO_OK;
}

/* ==================================================================== MIC2 */

static uint16_t zigzag16(int16_t v) { return (uint16_t)(((uint16_t)v << 1) ^ (uint16_t)(v >> 15)); } /* deltazigzagcompressu16.go:108-111 */
static int16_t unzigzag16(uint16_t u) { return (int16_t)((u >> 1) ^ (uint16_t)(-(int16_t)(u & 1))); } /* :113-116 */

/* compressResidualFrame, multiframecompress.go:146-163 */
static int compress_residual(const uint16_t *res, size_t n, uint16_t res_max,
                             uint8_t *out, size_t cap, size_t *out_len) {
    size_t scap = 2 * n + 16;
    uint16_t *tok = (uint16_t *)malloc(sizeof(uint16_t) * scap);
    if (!tok) return MICO_ERR_NOMEM;
    size_t tn = 0;
    int rc = mico_rle_compress(res, n, res_max, tok, scap, &tn);
    if (rc == MICO_OK) rc = fse_chain(tok, tn, 2, out, cap, out_len);
    free(tok);
    return rc;
}

/* CompressMultiFrame + WriteMIC2, multiframecompress.go:179-224, multiframe.go:49-91 */
int mico_mic2_compress(const uint16_t *frames, int w, int h, int nframes,
                       uint16_t max_value, int temporal,
                       uint8_t *out, size_t cap, size_t *out_len) {
    if (!frames || w <= 0 || h <= 0 || nframes <= 0) return MICO_ERR_ARGS;
    size_t npx = (size_t)w * (size_t)h;
    size_t header = 20 + (size_t)nframes * 8;
    if (cap < header) return MICO_ERR_CAPACITY;
    memset(out, 0, header);
    memcpy(out, "MIC2", 4);
    put_u32(out + 4, (uint32_t)w); put_u32(out + 8, (uint32_t)h); put_u32(out + 12, (uint32_t)nframes);
    out[16] = (uint8_t)(0x01 | (temporal ? 0x02 : 0));
    uint32_t offset = 0;
    uint16_t *res = NULL;
    if (temporal) { res = (uint16_t *)malloc(sizeof(uint16_t) * npx); if (!res) return MICO_ERR_NOMEM; }
    for (int i = 0; i < nframes; i++) {
        const uint16_t *f = frames + (size_t)i * npx;
        size_t blen = 0;
        int rc;
        if (temporal && i > 0) {
            const uint16_t *p = f - npx;
            uint16_t res_max = 0;
            for (size_t k = 0; k < npx; k++) { /* TemporalDeltaEncode, temporaldelta.go:11-23 */
                res[k] = zigzag16((int16_t)((int32_t)f[k] - (int32_t)p[k]));
                if (res[k] > res_max) res_max = res[k];
            }
            rc = compress_residual(res, npx, res_max, out + header + offset, cap - header - offset, &blen);
        } else {
            rc = mico_compress_single_frame(f, w, h, max_value, 2, out + header + offset, cap - header - offset, &blen);
        }
        if (rc) { free(res); return rc; }
        put_u32(out + 20 + (size_t)i * 8, offset);
        put_u32(out + 24 + (size_t)i * 8, (uint32_t)blen);
        offset += (uint32_t)blen;
    }
    free(res);
    *out_len = header + offset;
    return MICO_OK;
}

/* DecompressMultiFrame, multiframecompress.go:227-261; ReadMIC2Header multiframe.go:95-128 */
int mico_mic2_decompress(const uint8_t *in, size_t len, uint16_t *frames,
                         size_t px_cap, int *w, int *h, int *nframes) {
    if (len < 20 || memcmp(in, "MIC2", 4) != 0) return MICO_ERR_CORRUPT;
    int width = (int)get_u32(in + 4), height = (int)get_u32(in + 8), n = (int)get_u32(in + 12);
    int temporal = (in[16] & 0x02) != 0;
    if (n < 0 || (size_t)n > (len - 20) / 8) return MICO_ERR_CORRUPT;
    size_t data_off = 20 + (size_t)n * 8;
    *w = width; *h = height; *nframes = n;
    if (!frames) return MICO_OK;
    if (width <= 0 || height <= 0) return MICO_ERR_CORRUPT;
    size_t npx = (size_t)width * (size_t)height;
    if (npx * (size_t)n > px_cap) return MICO_ERR_CAPACITY;
    uint16_t *tok = NULL, *res = NULL;
    if (temporal) {
        tok = (uint16_t *)malloc(sizeof(uint16_t) * (2 * npx + 16));
        res = (uint16_t *)malloc(sizeof(uint16_t) * npx);
        if (!tok || !res) { free(tok); free(res); return MICO_ERR_NOMEM; }
    }
    int rc = MICO_OK;
    for (int i = 0; i < n && rc == MICO_OK; i++) {
        size_t start = data_off + get_u32(in + 20 + (size_t)i * 8);
        size_t blen = get_u32(in + 24 + (size_t)i * 8);
        if (start + blen > len) { rc = MICO_ERR_CORRUPT; break; }
        uint16_t *f = frames + (size_t)i * npx;
        if (temporal && i > 0) { /* decompressResidualFrame + TemporalDeltaDecode */
            size_t tn = 0, rn = 0;
            rc = mico_fse_decompress_auto(in + start, blen, tok, 2 * npx + 16, &tn);
            if (rc == MICO_OK) rc = mico_rle_decompress(tok, tn, res, npx, &rn);
            if (rc == MICO_OK && rn != npx) rc = MICO_ERR_CORRUPT;
            if (rc == MICO_OK)
                for (size_t k = 0; k < npx; k++)
                    f[k] = (uint16_t)((int32_t)(f - npx)[k] + (int32_t)unzigzag16(res[k]));
        } else {
            rc = mico_decompress_single_frame(in + start, blen, f, width, height);
        }
    }
    free(tok); free(res);
    return rc;
}
