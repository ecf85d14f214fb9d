/*
 * mic_oracle_wavelet.c -- CPU restatement of MIC's WaveletV2 pipeline.
 * TEST INFRASTRUCTURE ONLY (see mic_oracle.h).
 *
 * 5/3 integer lifting (waveletu16.go:26-122), separated/Mallat 2-D transform
 * (waveletu16.go:162-257; the SIMD variant :314-508 produces the same coefficients,
 * waveletu16_test.go:352-382), subband scan (waveletfsecompressu16.go:202-282), zigzag +
 * escape (:28-58), RLE with length prefix (rlecompressu16.go:85-93), 4-state FSE without
 * fallback, 11-byte header (:303-372, :493-534).
 */
#include "mic_oracle_int.h"

/* waveletu16.go:26-74 */
static void wt53_forward_1d(int32_t *data, size_t offset, int n, size_t stride) {
    if (n < 2) return;
    int n_half = n / 2;
    for (int i = 0; i < n_half; i++) {
        size_t odd = offset + (size_t)(2 * i + 1) * stride, left = offset + (size_t)(2 * i) * stride;
        size_t right = (2 * i + 2 < n) ? offset + (size_t)(2 * i + 2) * stride : left;
        data[odd] = data[odd] - ((data[left] + data[right]) >> 1);
    }
    int n_low = (n + 1) / 2;
    for (int i = 0; i < n_low; i++) {
        size_t even = offset + (size_t)(2 * i) * stride;
        int32_t d_right, d_left;
        if (2 * i + 1 < n) d_right = data[offset + (size_t)(2 * i + 1) * stride];
        else d_right = (i > 0) ? data[offset + (size_t)(2 * i - 1) * stride] : 0;
        d_left = (i > 0) ? data[offset + (size_t)(2 * i - 1) * stride] : d_right;
        data[even] = data[even] + ((d_left + d_right + 2) >> 2);
    }
}

/* waveletu16.go:78-122 */
static void wt53_inverse_1d(int32_t *data, size_t offset, int n, size_t stride) {
    if (n < 2) return;
    int n_half = n / 2, n_low = (n + 1) / 2;
    for (int i = 0; i < n_low; i++) {
        size_t even = offset + (size_t)(2 * i) * stride;
        int32_t d_right, d_left;
        if (2 * i + 1 < n) d_right = data[offset + (size_t)(2 * i + 1) * stride];
        else d_right = (i > 0) ? data[offset + (size_t)(2 * i - 1) * stride] : 0;
        d_left = (i > 0) ? data[offset + (size_t)(2 * i - 1) * stride] : d_right;
        data[even] = data[even] - ((d_left + d_right + 2) >> 2);
    }
    for (int i = 0; i < n_half; i++) {
        size_t odd = offset + (size_t)(2 * i + 1) * stride, left = offset + (size_t)(2 * i) * stride;
        size_t right = (2 * i + 2 < n) ? offset + (size_t)(2 * i + 2) * stride : left;
        data[odd] = data[odd] + ((data[left] + data[right]) >> 1);
    }
}

/* waveletu16.go:162-209 */
static int wt53_forward_2d(int32_t *data, int rows, int cols, int full_cols) {
    int n_col_low = (cols + 1) / 2, n_row_low = (rows + 1) / 2;
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rows > cols ? rows : cols));
    if (!tmp) return MICO_ERR_NOMEM;
    for (int y = 0; y < rows; y++) wt53_forward_1d(data, (size_t)y * full_cols, cols, 1);
    for (int y = 0; y < rows; y++) {
        int32_t *row = data + (size_t)y * full_cols;
        memcpy(tmp, row, sizeof(int32_t) * (size_t)cols);
        for (int i = 0; i < n_col_low; i++) row[i] = tmp[2 * i];
        for (int i = 0; i < cols / 2; i++) row[n_col_low + i] = tmp[2 * i + 1];
    }
    for (int x = 0; x < cols; x++) {
        wt53_forward_1d(data, (size_t)x, rows, (size_t)full_cols);
        for (int i = 0; i < rows; i++) tmp[i] = data[(size_t)i * full_cols + x];
        for (int i = 0; i < n_row_low; i++) data[(size_t)i * full_cols + x] = tmp[2 * i];
        for (int i = 0; i < rows / 2; i++) data[(size_t)(n_row_low + i) * full_cols + x] = tmp[2 * i + 1];
    }
    free(tmp);
    return MICO_OK;
}

/* waveletu16.go:213-257 */
static int wt53_inverse_2d(int32_t *data, int rows, int cols, int full_cols) {
    int n_col_low = (cols + 1) / 2, n_row_low = (rows + 1) / 2;
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rows > cols ? rows : cols));
    if (!tmp) return MICO_ERR_NOMEM;
    for (int x = 0; x < cols; x++) {
        for (int i = 0; i < n_row_low; i++) tmp[2 * i] = data[(size_t)i * full_cols + x];
        for (int i = 0; i < rows / 2; i++) tmp[2 * i + 1] = data[(size_t)(n_row_low + i) * full_cols + x];
        for (int i = 0; i < rows; i++) data[(size_t)i * full_cols + x] = tmp[i];
        wt53_inverse_1d(data, (size_t)x, rows, (size_t)full_cols);
    }
    for (int y = 0; y < rows; y++) {
        int32_t *row = data + (size_t)y * full_cols;
        memcpy(tmp, row, sizeof(int32_t) * (size_t)cols);
        for (int i = 0; i < n_col_low; i++) row[2 * i] = tmp[i];
        for (int i = 0; i < cols / 2; i++) row[2 * i + 1] = tmp[n_col_low + i];
        wt53_inverse_1d(data, (size_t)y * full_cols, cols, 1);
    }
    free(tmp);
    return MICO_OK;
}

/* multi-level loop of waveletfsecompressu16.go:319-330 */
int mico_wt53_forward(int32_t *data, int rows, int cols, int levels, int *applied) {
    int r = rows, c = cols, l;
    for (l = 0; l < levels; l++) {
        if (r < 2 || c < 2) break;
        int rc = wt53_forward_2d(data, r, c, cols);
        if (rc) return rc;
        r = (r + 1) / 2; c = (c + 1) / 2;
    }
    *applied = l;
    return MICO_OK;
}

/* waveletfsecompressu16.go:519-527 */
int mico_wt53_inverse(int32_t *data, int rows, int cols, int levels) {
    int dr[9], dc[9];
    int r = rows, c = cols;
    if (levels > 8) return MICO_ERR_ARGS;
    for (int l = 0; l < levels; l++) { dr[l] = r; dc[l] = c; r = (r + 1) / 2; c = (c + 1) / 2; }
    for (int l = levels - 1; l >= 0; l--) {
        int rc = wt53_inverse_2d(data, dr[l], dc[l], cols);
        if (rc) return rc;
    }
    return MICO_OK;
}

/* collectSubbandOrder / scatterSubbandOrder, waveletfsecompressu16.go:202-282.
 * dir = 0: out[pos] = data[...]; dir = 1: data[...] = out[pos]. */
static void subband_walk(int32_t *data, int32_t *lin, int rows, int cols, int full_cols, int levels, int dir) {
    int nr[10], nc[10];
    nr[0] = rows; nc[0] = cols;
    for (int l = 1; l <= levels; l++) { nr[l] = (nr[l - 1] + 1) / 2; nc[l] = (nc[l - 1] + 1) / 2; }
    size_t pos = 0;
#define VISIT(y, x) do { size_t _i = (size_t)(y) * full_cols + (x); if (dir) data[_i] = lin[pos]; else lin[pos] = data[_i]; pos++; } while (0)
    for (int y = 0; y < nr[levels]; y++) for (int x = 0; x < nc[levels]; x++) VISIT(y, x);
    for (int l = levels; l >= 1; l--) {
        for (int y = 0; y < nr[l]; y++) for (int x = nc[l]; x < nc[l - 1]; x++) VISIT(y, x);          /* HL */
        for (int y = nr[l]; y < nr[l - 1]; y++) for (int x = 0; x < nc[l]; x++) VISIT(y, x);          /* LH */
        for (int y = nr[l]; y < nr[l - 1]; y++) for (int x = nc[l]; x < nc[l - 1]; x++) VISIT(y, x);  /* HH */
    }
#undef VISIT
}

/* WaveletV2RLEFSECompressU16 == WaveletV2SIMDRLEFSECompressU16, waveletfsecompressu16.go:303-487 */
int mico_wavelet_v2_compress(const uint16_t *px, int rows, int cols, uint16_t max_value, int levels,
                             uint8_t *out, size_t cap, size_t *out_len) {
    if (!px || rows <= 0 || cols <= 0) return MICO_ERR_ARGS;
    if (levels < 1) levels = 1;
    if (levels > 8) levels = 8;
    size_t n = (size_t)rows * (size_t)cols;
    int32_t *data = (int32_t *)malloc(sizeof(int32_t) * n), *ord = (int32_t *)malloc(sizeof(int32_t) * n);
    uint16_t *enc = (uint16_t *)malloc(sizeof(uint16_t) * (3 * n + 8)), *tok = (uint16_t *)malloc(sizeof(uint16_t) * (6 * n + 32));
    int rc = MICO_ERR_NOMEM;
    if (data && ord && enc && tok) {
        for (size_t i = 0; i < n; i++) data[i] = (int32_t)px[i];
        rc = mico_wt53_forward(data, rows, cols, levels, &levels);
        if (rc == MICO_OK) {
            subband_walk(data, ord, rows, cols, cols, levels, 0);
            size_t m = 0;
            uint16_t zz_max = 0;
            for (size_t i = 0; i < n; i++) {                        /* waveletCoeffsToU16, :28-40 */
                int32_t v = ord[i];
                if (v >= -32767 && v <= 32767) enc[m++] = (uint16_t)((v >> 31) ^ (int32_t)((uint32_t)v << 1));
                else { uint32_t uu = (uint32_t)v; enc[m++] = 65535; enc[m++] = (uint16_t)(uu >> 16); enc[m++] = (uint16_t)uu; }
            }
            for (size_t i = 0; i < m; i++) if (enc[i] > zz_max) zz_max = enc[i];
            int depth = len16(zz_max);
            if (depth < 1) depth = 1;
            uint16_t rle_max = (uint16_t)((1u << depth) - 1);
            size_t tn = 0;
            rc = mico_rle_compress(enc, m, rle_max, tok, 6 * n + 32, &tn);
            if (rc == MICO_OK) {
                if (cap < 11) rc = MICO_ERR_CAPACITY;
                else {
                    size_t fl = 0;
                    rc = mico_fse_compress(tok, tn, 4, out + 11, cap - 11, &fl);   /* no fallback, :344 */
                    if (rc == MICO_OK) {
                        out[0] = (uint8_t)rows; out[1] = (uint8_t)(rows >> 8); out[2] = (uint8_t)(rows >> 16); out[3] = (uint8_t)((uint32_t)rows >> 24);
                        out[4] = (uint8_t)cols; out[5] = (uint8_t)(cols >> 8); out[6] = (uint8_t)(cols >> 16); out[7] = (uint8_t)((uint32_t)cols >> 24);
                        out[8] = (uint8_t)max_value; out[9] = (uint8_t)(max_value >> 8);
                        out[10] = (uint8_t)levels;
                        *out_len = 11 + fl;
                    }
                }
            }
        }
    }
    free(data); free(ord); free(enc); free(tok);
    return rc;
}

/* WaveletV2SIMDRLEFSEDecompressU16, waveletfsecompressu16.go:493-534 */
int mico_wavelet_v2_decompress(const uint8_t *in, size_t len, uint16_t *px, size_t px_cap, int *rows_out, int *cols_out) {
    if (len < 11) return MICO_ERR_CORRUPT;
    int rows = (int)((uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24));
    int cols = (int)((uint32_t)in[4] | ((uint32_t)in[5] << 8) | ((uint32_t)in[6] << 16) | ((uint32_t)in[7] << 24));
    int levels = in[10];
    *rows_out = rows; *cols_out = cols;
    if (!px) return MICO_OK;
    if (rows <= 0 || cols <= 0 || levels > 8) return MICO_ERR_CORRUPT;
    size_t n = (size_t)rows * (size_t)cols;
    if (n > px_cap) return MICO_ERR_CAPACITY;
    if (len < 13 || in[11] != 0xFF || in[12] != 0x04) return MICO_ERR_CORRUPT;        /* FSEDecompressU16FourState only */
    uint16_t *tok = (uint16_t *)malloc(sizeof(uint16_t) * (6 * n + 32)), *enc = (uint16_t *)malloc(sizeof(uint16_t) * (3 * n + 8));
    int32_t *ord = (int32_t *)malloc(sizeof(int32_t) * n), *data = (int32_t *)calloc(n, sizeof(int32_t));
    int rc = MICO_ERR_NOMEM;
    if (tok && enc && ord && data) {
        size_t tn = 0, m = 0;
        rc = mico_fse_decompress_auto(in + 11, len - 11, tok, 6 * n + 32, &tn);
        if (rc == MICO_OK) rc = mico_rle_decompress(tok, tn, enc, 3 * n + 8, &m);
        if (rc == MICO_OK) {
            size_t i = 0, k = 0;                                      /* u16ToWaveletCoeffs, :43-58 */
            while (i < m && k < n) {
                if (enc[i] != 65535) { uint32_t u = enc[i]; ord[k++] = (int32_t)((u >> 1) ^ (uint32_t)(-(int32_t)(u & 1))); i++; }
                else {
                    if (i + 2 >= m) { rc = MICO_ERR_CORRUPT; break; }
                    ord[k++] = (int32_t)(((uint32_t)enc[i + 1] << 16) | (uint32_t)enc[i + 2]); i += 3;
                }
            }
            if (rc == MICO_OK && k < n) rc = MICO_ERR_CORRUPT;          /* Go: index panic in scatter */
        }
        if (rc == MICO_OK) {
            subband_walk(data, ord, rows, cols, cols, levels, 1);
            rc = mico_wt53_inverse(data, rows, cols, levels);
            if (rc == MICO_OK) for (size_t i = 0; i < n; i++) px[i] = (uint16_t)data[i];
        }
    }
    free(tok); free(enc); free(ord); free(data);
    return rc;
}
