/*
 * mic_oracle_wsi.c -- CPU restatement of MIC's MIC3 / WSI path: 8-bit RGB and 8/16-bit greyscale.
 * TEST INFRASTRUCTURE ONLY (see mic_oracle.h).
 *
 * YCoCg-R (asm_amd64.go:88-121, ycocgr.go:19-35), 2x2 box pyramid (wsipyramid.go:10-32),
 * zero-padded tiles (wsicompress.go:529-555), plane modes (wsicompress.go:373-421, :464-500),
 * tile blob (wsicompress.go:319-364, :430-462), MIC3 container (wsiformat.go:99-285),
 * CompressWSI (wsicompress.go:27-171), DecompressWSITile (:175-217).
 * Greyscale: Downsample2xGrey (wsipyramid.go:34-55), compressGreyTileBlob / decompressGreyTileBlob
 * (wsicompress.go:366-370, :477-484), bytesToUint16Slice / uint16ToBytes (:573-603).
 */
#include "mic_oracle_int.h"

static uint16_t zz16(int16_t v) { return (uint16_t)(((uint16_t)v << 1) ^ (uint16_t)(v >> 15)); }
static int16_t unzz16(uint16_t u) { return (int16_t)((u >> 1) ^ (uint16_t)(-(int16_t)(u & 1))); }

void mico_ycocgr_forward(const uint8_t *rgb, int npx, uint16_t *y, uint16_t *co, uint16_t *cg) {
    for (int i = 0; i < npx; i++) {
        int r = rgb[i * 3], g = rgb[i * 3 + 1], b = rgb[i * 3 + 2];
        int co_v = r - b;
        int t = b + (co_v >> 1);
        int cg_v = g - t;
        int y_v = t + (cg_v >> 1);
        y[i] = (uint16_t)y_v; co[i] = zz16((int16_t)co_v); cg[i] = zz16((int16_t)cg_v);
    }
}
void mico_ycocgr_inverse(const uint16_t *y, const uint16_t *co, const uint16_t *cg, int npx, uint8_t *rgb) {
    for (int i = 0; i < npx; i++) {
        int y_v = y[i], co_v = unzz16(co[i]), cg_v = unzz16(cg[i]);
        int t = y_v - (cg_v >> 1);
        int g = cg_v + t;
        int b = t - (co_v >> 1);
        int r = co_v + b;
        rgb[i * 3] = (uint8_t)r; rgb[i * 3 + 1] = (uint8_t)g; rgb[i * 3 + 2] = (uint8_t)b;
    }
}

/* compressWSIPlane, wsicompress.go:373-421 */
static int compress_plane(const uint16_t *plane, int w, int h, uint8_t *out, size_t cap, size_t *out_len) {
    size_t n = (size_t)w * (size_t)h;
    int is_const = 1;
    uint16_t val = plane[0], max_val = plane[0];
    for (size_t i = 1; i < n; i++) { if (plane[i] != val) is_const = 0; if (plane[i] > max_val) max_val = plane[i]; }
    if (is_const) {
        if (val == 0) { if (cap < 1) return MICO_ERR_CAPACITY; out[0] = 0; *out_len = 1; return MICO_OK; }
        if (cap < 3) return MICO_ERR_CAPACITY;
        out[0] = 1; out[1] = (uint8_t)val; out[2] = (uint8_t)(val >> 8); *out_len = 3;
        return MICO_OK;
    }
    if (max_val < 255) max_val = 255;
    if (cap < 1) return MICO_ERR_CAPACITY;
    size_t cl = 0;
    int rc = mico_compress_single_frame(plane, w, h, max_val, 2, out + 1, cap - 1, &cl);
    if (rc == MICO_ERR_USE_RLE || rc == MICO_ERR_INCOMPRESSIBLE) {
        if (cap < 1 + 2 * n) return MICO_ERR_CAPACITY;
        out[0] = 3;
        for (size_t i = 0; i < n; i++) { out[1 + 2 * i] = (uint8_t)plane[i]; out[2 + 2 * i] = (uint8_t)(plane[i] >> 8); }
        *out_len = 1 + 2 * n;
        return MICO_OK;
    }
    if (rc) return rc;
    out[0] = 2;
    *out_len = 1 + cl;
    return MICO_OK;
}

/* decompressWSIPlane, wsicompress.go:464-500 */
static int decompress_plane(const uint8_t *data, size_t len, int w, int h, uint16_t *out) {
    size_t n = (size_t)w * (size_t)h;
    if (len == 0) return MICO_ERR_CORRUPT;
    switch (data[0]) {
    case 0: memset(out, 0, n * 2); return MICO_OK;
    case 1: {
        if (len < 3) return MICO_ERR_CORRUPT;
        uint16_t v = (uint16_t)(data[1] | (data[2] << 8));
        for (size_t i = 0; i < n; i++) out[i] = v;
        return MICO_OK;
    }
    case 2: return mico_decompress_single_frame(data + 1, len - 1, out, w, h);
    case 3:
        if (len < 1 + 2 * n) return MICO_ERR_CORRUPT;
        for (size_t i = 0; i < n; i++) out[i] = (uint16_t)(data[1 + 2 * i] | (data[2 + 2 * i] << 8));
        return MICO_OK;
    default: return MICO_ERR_CORRUPT;
    }
}

static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void put64(uint8_t *p, uint64_t v) { put32(p, (uint32_t)v); put32(p + 4, (uint32_t)(v >> 32)); }
static uint64_t get64(const uint8_t *p) { return (uint64_t)get32(p) | ((uint64_t)get32(p + 4) << 32); }

/* compressRGBTileBlob with the colour transform, wsicompress.go:319-364 */
int mico_wsi_compress_tile(const uint8_t *rgb, int tw, int th, uint8_t *out, size_t cap, size_t *out_len) {
    size_t n = (size_t)tw * (size_t)th;
    uint16_t *pl = (uint16_t *)malloc(sizeof(uint16_t) * 3 * n);
    if (!pl) return MICO_ERR_NOMEM;
    mico_ycocgr_forward(rgb, (int)n, pl, pl + n, pl + 2 * n);
    if (cap < 12) { free(pl); return MICO_ERR_CAPACITY; }
    size_t off = 12;
    for (int k = 0; k < 3; k++) {
        size_t bl = 0;
        int rc = compress_plane(pl + (size_t)k * n, tw, th, out + off, cap - off, &bl);
        if (rc) { free(pl); return rc; }
        put32(out + 4 * k, (uint32_t)bl);
        off += bl;
    }
    free(pl);
    *out_len = off;
    return MICO_OK;
}

/* decompressRGBTileBlob, wsicompress.go:430-462 */
int mico_wsi_decompress_tile(const uint8_t *in, size_t len, int tw, int th, uint8_t *rgb) {
    if (len < 12) return MICO_ERR_CORRUPT;
    size_t l0 = get32(in), l1 = get32(in + 4), l2 = get32(in + 8);
    if (12 + l0 + l1 + l2 > len) return MICO_ERR_CORRUPT;
    size_t n = (size_t)tw * (size_t)th;
    uint16_t *pl = (uint16_t *)malloc(sizeof(uint16_t) * 3 * n);
    if (!pl) return MICO_ERR_NOMEM;
    int rc = decompress_plane(in + 12, l0, tw, th, pl);
    if (rc == MICO_OK) rc = decompress_plane(in + 12 + l0, l1, tw, th, pl + n);
    if (rc == MICO_OK) rc = decompress_plane(in + 12 + l0 + l1, l2, tw, th, pl + 2 * n);
    if (rc == MICO_OK) mico_ycocgr_inverse(pl, pl + n, pl + 2 * n, (int)n, rgb);
    free(pl);
    return rc;
}

/* The CLI's single-frame files.  MICR (writeMICRFile, cmd/mic-compress/main.go:62-91): "MICR", width, height, CompressRGB blob
 * (rgbcompress.go:25-27 = compressRGBTileBlob on the whole image).  MIC1 (writeMicFile, main.go:26-59): "MIC1", width, height,
 * pipeline 1, payload length, CompressSingleFrame stream. */
int mico_micr_write(const uint8_t *rgb, int w, int h, uint8_t *out, size_t cap, size_t *out_len) {
    if (cap < 12) return MICO_ERR_CAPACITY;
    size_t n = 0;
    int rc = mico_wsi_compress_tile(rgb, w, h, out + 12, cap - 12, &n);
    if (rc) return rc;
    memcpy(out, "MICR", 4); put32(out + 4, (uint32_t)w); put32(out + 8, (uint32_t)h);
    *out_len = 12 + n;
    return MICO_OK;
}
int mico_micr_read(const uint8_t *in, size_t len, uint8_t *rgb, size_t cap, int *w, int *h) {
    if (len < 12 || memcmp(in, "MICR", 4) != 0) return MICO_ERR_CORRUPT;
    *w = (int)get32(in + 4); *h = (int)get32(in + 8);
    if (*w <= 0 || *h <= 0) return MICO_ERR_CORRUPT;
    if ((size_t)*w * (size_t)*h * 3 > cap) return MICO_ERR_CAPACITY;
    return mico_wsi_decompress_tile(in + 12, len - 12, *w, *h, rgb);
}
int mico_mic1_write(const uint16_t *px, int w, int h, uint16_t max_value, int nstates, uint8_t *out, size_t cap, size_t *out_len) {
    if (cap < 20) return MICO_ERR_CAPACITY;
    size_t n = 0;
    int rc = mico_compress_single_frame(px, w, h, max_value, nstates, out + 20, cap - 20, &n);
    if (rc) return rc;
    memcpy(out, "MIC1", 4); put32(out + 4, (uint32_t)w); put32(out + 8, (uint32_t)h); put32(out + 12, 1); put32(out + 16, (uint32_t)n);
    *out_len = 20 + n;
    return MICO_OK;
}
int mico_mic1_read(const uint8_t *in, size_t len, uint16_t *px, size_t cap_px, int *w, int *h) {
    if (len < 20 || memcmp(in, "MIC1", 4) != 0 || get32(in + 12) != 1) return MICO_ERR_CORRUPT;
    *w = (int)get32(in + 4); *h = (int)get32(in + 8);
    size_t n = get32(in + 16);
    if (*w <= 0 || *h <= 0 || n > len - 20) return MICO_ERR_CORRUPT;
    if ((size_t)*w * (size_t)*h > cap_px) return MICO_ERR_CAPACITY;
    return mico_decompress_single_frame(in + 20, n, px, *w, *h);
}

/* bytesToUint16Slice / uint16ToBytes, wsicompress.go:573-603: one byte per sample up to 8 bits, else little-endian pairs */
static uint16_t sample_get(const uint8_t *p, size_t i, int bps) { return bps <= 8 ? p[i] : (uint16_t)(p[2 * i] | (p[2 * i + 1] << 8)); }
static void sample_put(uint8_t *p, size_t i, int bps, uint16_t v) { if (bps <= 8) p[i] = (uint8_t)v; else { p[2 * i] = (uint8_t)v; p[2 * i + 1] = (uint8_t)(v >> 8); } }

/* compressGreyTileBlob, wsicompress.go:366-370: the tile blob is the plane blob itself (no length prefix) */
int mico_wsi_compress_grey_tile(const uint8_t *px, int tw, int th, int bps, uint8_t *out, size_t cap, size_t *out_len) {
    size_t n = (size_t)tw * (size_t)th;
    uint16_t *pl = (uint16_t *)malloc(sizeof(uint16_t) * n);
    if (!pl) return MICO_ERR_NOMEM;
    for (size_t i = 0; i < n; i++) pl[i] = sample_get(px, i, bps);
    int rc = compress_plane(pl, tw, th, out, cap, out_len);
    free(pl);
    return rc;
}

/* decompressGreyTileBlob, wsicompress.go:477-484 */
int mico_wsi_decompress_grey_tile(const uint8_t *in, size_t len, int tw, int th, int bps, uint8_t *px) {
    size_t n = (size_t)tw * (size_t)th;
    uint16_t *pl = (uint16_t *)malloc(sizeof(uint16_t) * n);
    if (!pl) return MICO_ERR_NOMEM;
    int rc = decompress_plane(in, len, tw, th, pl);
    if (rc == MICO_OK) for (size_t i = 0; i < n; i++) sample_put(px, i, bps, pl[i]);
    free(pl);
    return rc;
}

/* CompressWSI, wsicompress.go:27-171 + wsiformat.go:99-165, :244-285.  channels 3 / 8 bits takes the RGB tile blob,
 * channels 1 (8 or 16 bits) the greyscale one (compressTileBlob, :312-317). */
int mico_wsi_compress_ex(const uint8_t *px, int w, int h, int channels, int bps, int tile_w, int tile_h, int levels_req,
                         uint8_t *out, size_t cap, size_t *out_len) {
    if (!px || w <= 0 || h <= 0) return MICO_ERR_ARGS;
    if (!((channels == 3 && bps == 8) || (channels == 1 && (bps == 8 || bps == 16)))) return MICO_ERR_ARGS;
    const int grey = channels == 1;
    const size_t bpp = (size_t)channels * (bps == 16 ? 2 : 1);          /* bytesPerPixel, wsicompress.go:530-533 */
    if (tile_w == 0) tile_w = 256;
    if (tile_h == 0) tile_h = 256;
    int num_levels = levels_req;
    if (num_levels <= 0) {                                              /* autoLevelCount, wsiformat.go:273-285 */
        num_levels = 1;
        int ww = w, hh = h;
        while (ww > tile_w || hh > tile_h) { ww /= 2; hh /= 2; num_levels++; if (ww <= 1 && hh <= 1) break; }
    }
    if (num_levels > 32) return MICO_ERR_ARGS;
    int lw[32], lh[32], ltx[32], lty[32], lfirst[32];
    {                                                                   /* computeLevels, wsiformat.go:244-270 */
        int ww = w, hh = h;
        for (int i = 0; i < num_levels; i++) {
            lw[i] = ww; lh[i] = hh; ltx[i] = (ww + tile_w - 1) / tile_w; lty[i] = (hh + tile_h - 1) / tile_h;
            ww /= 2; hh /= 2; if (ww == 0) ww = 1; if (hh == 0) hh = 1;
        }
    }
    uint8_t *pyr[32]; memset(pyr, 0, sizeof pyr);
    int pw[32], ph[32];
    pyr[0] = (uint8_t *)px; pw[0] = w; ph[0] = h;
    int rc = MICO_OK;
    for (int i = 1; i < num_levels; i++) {
        int nw = pw[i - 1] / 2, nh = ph[i - 1] / 2;
        if (nw == 0 || nh == 0) { num_levels = i; break; }
        uint8_t *d = (uint8_t *)malloc((size_t)nw * nh * bpp);
        if (!d) { rc = MICO_ERR_NOMEM; num_levels = i; break; }
        const uint8_t *s = pyr[i - 1]; int sw = pw[i - 1];
        if (!grey) {                                                    /* Downsample2xRGB, wsipyramid.go:10-32 */
            for (int y = 0; y < nh; y++) for (int x = 0; x < nw; x++) for (int c = 0; c < 3; c++) {
                int v = s[((size_t)(2 * y) * sw + 2 * x) * 3 + c] + s[((size_t)(2 * y) * sw + 2 * x + 1) * 3 + c] +
                        s[((size_t)(2 * y + 1) * sw + 2 * x) * 3 + c] + s[((size_t)(2 * y + 1) * sw + 2 * x + 1) * 3 + c];
                d[((size_t)y * nw + x) * 3 + c] = (uint8_t)((v + 2) / 4);
            }
        } else {                                                        /* Downsample2xGrey on u16, wsipyramid.go:34-55 */
            for (int y = 0; y < nh; y++) for (int x = 0; x < nw; x++) {
                uint32_t v = (uint32_t)sample_get(s, (size_t)(2 * y) * sw + 2 * x, bps) + sample_get(s, (size_t)(2 * y) * sw + 2 * x + 1, bps) +
                             sample_get(s, (size_t)(2 * y + 1) * sw + 2 * x, bps) + sample_get(s, (size_t)(2 * y + 1) * sw + 2 * x + 1, bps);
                sample_put(d, (size_t)y * nw + x, bps, (uint16_t)((v + 2) / 4));
            }
        }
        pyr[i] = d; pw[i] = nw; ph[i] = nh;
        lw[i] = nw; lh[i] = nh; ltx[i] = (nw + tile_w - 1) / tile_w; lty[i] = (nh + tile_h - 1) / tile_h;
    }
    size_t total_tiles = 0;
    for (int i = 0; i < num_levels; i++) { lfirst[i] = (int)total_tiles; total_tiles += (size_t)ltx[i] * lty[i]; }
    size_t hdr = 48 + 20 * (size_t)num_levels + 16 * total_tiles;
    uint8_t *tile = (uint8_t *)malloc((size_t)tile_w * tile_h * bpp);
    if (rc == MICO_OK && (!tile || cap < hdr)) rc = tile ? MICO_ERR_CAPACITY : MICO_ERR_NOMEM;
    size_t off = 0;
    if (rc == MICO_OK) {
        memset(out, 0, hdr);
        memcpy(out, "MIC3", 4); put32(out + 4, 1); put32(out + 8, (uint32_t)w); put32(out + 12, (uint32_t)h);
        put32(out + 16, (uint32_t)tile_w); put32(out + 20, (uint32_t)tile_h);
        out[24] = (uint8_t)channels; out[25] = 0; out[26] = (uint8_t)bps;
        out[27] = (uint8_t)(0x01 | (grey ? 0 : 0x02));                  /* FlagSpatial | FlagColorTransform (RGB only, wsiformat.go:86-96) */
        out[28] = (uint8_t)num_levels; out[29] = (uint8_t)(num_levels >> 8);
        put64(out + 32, (uint64_t)total_tiles);
        for (int i = 0; i < num_levels; i++) {
            uint8_t *ld = out + 48 + 20 * (size_t)i;
            put32(ld, (uint32_t)lw[i]); put32(ld + 4, (uint32_t)lh[i]); put32(ld + 8, (uint32_t)ltx[i]); put32(ld + 12, (uint32_t)lty[i]); put32(ld + 16, (uint32_t)lfirst[i]);
        }
        size_t ti = 0;
        for (int lv = 0; lv < num_levels && rc == MICO_OK; lv++) {
            for (int ty = 0; ty < lty[lv] && rc == MICO_OK; ty++) for (int tx = 0; tx < ltx[lv] && rc == MICO_OK; tx++) {
                memset(tile, 0, (size_t)tile_w * tile_h * bpp);       /* extractTileRGB: zero padding */
                for (int yy = 0; yy < tile_h; yy++) {
                    int sy = ty * tile_h + yy; if (sy >= ph[lv]) break;
                    int sx = tx * tile_w; int cw = pw[lv] - sx; if (cw > tile_w) cw = tile_w;
                    memcpy(tile + (size_t)yy * tile_w * bpp, pyr[lv] + ((size_t)sy * pw[lv] + sx) * bpp, (size_t)cw * bpp);
                }
                size_t bl = 0;
                rc = grey ? mico_wsi_compress_grey_tile(tile, tile_w, tile_h, bps, out + hdr + off, cap - hdr - off, &bl)
                          : mico_wsi_compress_tile(tile, tile_w, tile_h, out + hdr + off, cap - hdr - off, &bl);
                if (rc) break;
                uint8_t *e = out + 48 + 20 * (size_t)num_levels + 16 * ti;
                put64(e, (uint64_t)off); put64(e + 8, (uint64_t)bl);
                off += bl; ti++;
            }
        }
    }
    for (int i = 1; i < 32; i++) free(pyr[i]);
    free(tile);
    if (rc == MICO_OK) *out_len = hdr + off;
    return rc;
}

int mico_wsi_compress(const uint8_t *rgb, int w, int h, int tile_w, int tile_h, int levels_req,
                      uint8_t *out, size_t cap, size_t *out_len) {
    return mico_wsi_compress_ex(rgb, w, h, 3, 8, tile_w, tile_h, levels_req, out, cap, out_len);
}

/* DecompressWSITile, wsicompress.go:175-217 (crop to the level's edge) */
int mico_wsi_decompress_tile_at(const uint8_t *in, size_t len, int level, int tx, int ty,
                                uint8_t *rgb, size_t cap, int *tw_out, int *th_out) {
    if (len < 48 || memcmp(in, "MIC3", 4) != 0 || get32(in + 4) != 1) return MICO_ERR_CORRUPT;
    int tile_w = (int)get32(in + 16), tile_h = (int)get32(in + 20);
    int channels = in[24] | (in[25] << 8), bps = in[26];
    int nlev = in[28] | (in[29] << 8);
    uint64_t total = get64(in + 32);
    const int grey = channels == 1 && (bps == 8 || bps == 16);
    if (!grey && (channels != 3 || bps != 8 || !(in[27] & 0x02))) return MICO_ERR_ARGS;
    const size_t bpp = (size_t)channels * (bps == 16 ? 2 : 1);
    if (len < 48 + 20 * (size_t)nlev || total > (len - 48 - 20 * (size_t)nlev) / 16) return MICO_ERR_CORRUPT;
    if (level < 0 || level >= nlev) return MICO_ERR_ARGS;
    const uint8_t *ld = in + 48 + 20 * (size_t)level;
    int lw = (int)get32(ld), lh = (int)get32(ld + 4), ltx = (int)get32(ld + 8), lty = (int)get32(ld + 12), first = (int)get32(ld + 16);
    if (tx < 0 || tx >= ltx || ty < 0 || ty >= lty) return MICO_ERR_ARGS;
    size_t gi = (size_t)first + (size_t)ty * ltx + tx;
    if (gi >= total) return MICO_ERR_CORRUPT;
    size_t data_off = 48 + 20 * (size_t)nlev + 16 * (size_t)total;
    const uint8_t *e = in + 48 + 20 * (size_t)nlev + 16 * gi;
    uint64_t bo = get64(e), bl = get64(e + 8);
    if (data_off + bo + bl > len) return MICO_ERR_CORRUPT;
    uint8_t *full = (uint8_t *)malloc((size_t)tile_w * tile_h * bpp);
    if (!full) return MICO_ERR_NOMEM;
    int rc = grey ? mico_wsi_decompress_grey_tile(in + data_off + bo, (size_t)bl, tile_w, tile_h, bps, full)     /* decompressTileBlob, :424-429 */
                  : mico_wsi_decompress_tile(in + data_off + bo, (size_t)bl, tile_w, tile_h, full);
    if (rc == MICO_OK) {
        int aw = tile_w, ah = tile_h;
        if (lw - tx * tile_w < aw) aw = lw - tx * tile_w;
        if (lh - ty * tile_h < ah) ah = lh - ty * tile_h;
        *tw_out = aw; *th_out = ah;
        if ((size_t)aw * ah * bpp > cap) rc = MICO_ERR_CAPACITY;
        else for (int y = 0; y < ah; y++) memcpy(rgb + (size_t)y * aw * bpp, full + (size_t)y * tile_w * bpp, (size_t)aw * bpp);
    }
    free(full);
    return rc;
}
