// mic_decode.hip -- decode kernels of the MIC unit codec for gfx950.
//
//   k_dec_tables   prefix / flavour detection, NCount parse, decode-table build
//                  (fse2state.go:102-116, fsedecompressu16.go:48-263, ransu16.go:77-135)
//   k_dec_tans     1/2/4/8-state tANS (and rANS-8) decode of the token stream
//                  (fsedecompressu16.go:267-377, fse2state.go:203-308, fse4state.go:195-353,
//                   fse8state.go:230-380, rans8state.go:221-412)
//   k_dec_pixels   RLE expansion + inverse Delta(avg)  (rledecompressu16.go:59-85,
//                  deltarlecompressu16.go:69-128)
//
// Launch shape: blockIdx.x = unit.
#include "mic_dev.h"
#include "mic_fse_tables.h"
#include "mic_launch.h"

// grid = units, block = 64; lane 0 parses (the NCount header is a serial bit-parse).
__global__ void __launch_bounds__(64) k_dec_tables(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (threadIdx.x != 0) return;
    u.status = MICD_OK; u.ntok = 0;
    if (u.w <= 0 || u.h <= 0 || !u.comp_in) { u.status = MICD_ERR_ARGS; return; }
    const uint8_t *b = u.comp_in;
    uint32_t len = u.comp_len;
    // FSEDecompressU16Auto, fse2state.go:102-116
    uint32_t flavour = 1;
    if (len >= 2 && b[0] == 0xFF) {
        if (b[1] == 0x84) flavour = 8;
        else if (b[1] == 0x08) flavour = 108;
        else if (b[1] == 0x04) flavour = 4;
        else if (b[1] == 0x02) flavour = 2;
    }
    uint32_t count = 0, off = 0;
    if (flavour != 1) {
        if (len < 6) { u.status = MICD_ERR_CORRUPT; return; }
        count = (uint32_t)b[2] | ((uint32_t)b[3] << 8) | ((uint32_t)b[4] << 16) | ((uint32_t)b[5] << 24);
        off = 6;
        if (count > u.tok_cap) { u.status = MICD_ERR_CORRUPT; return; }
    }
    u.flavour = flavour; u.count = count;
    uint32_t used = 0;
    int rc = mic_read_ncount(b + off, len - off, u.norm, &u.symbol_len, &u.table_log, &used);
    if (rc) { u.status = rc; return; }
    u.bits_off = off + used;
    rc = (flavour == 108) ? mic_build_rans_dtable(u) : mic_build_dtable(u);
    if (rc) { u.status = rc; return; }
}

// Reverse bit reader (bitreader.go): the highest set bit of the last byte is the end mark;
// bits are consumed from there towards the front.  `cursor` = number of unread bits.
struct BitR {
    const uint8_t *in; uint64_t cursor; bool over;
    __device__ bool init(const uint8_t *p, uint32_t len) {            // bitreader.go:27-47
        if (len < 1) return false;
        uint8_t v = p[len - 1];
        if (v == 0) return false;
        in = p; over = false;
        cursor = 8ull * (len - 1) + (uint32_t)(31 - __clz((uint32_t)v));
        return true;
    }
    __device__ uint32_t get(uint32_t n) {                             // getBits32, :49-61
        if (n == 0) return 0;
        if (cursor < n) { over = true; cursor = 0; return 0; }
        cursor -= n;
        uint64_t byte = cursor >> 3;
        uint32_t sh = (uint32_t)(cursor & 7);
        uint64_t w = 0;
        for (uint32_t k = 0; k < 4; k++) {                            // n <= 16 -> at most 3 bytes
            w |= (uint64_t)in[byte + k] << (8 * k);
            if (8 * (k + 1) >= sh + n) break;
        }
        return (uint32_t)((w >> sh) & ((1ull << n) - 1));
    }
};

// v0: one lane decodes the stream.  grid = units, block = 64.
__global__ void __launch_bounds__(64) k_dec_tans_serial(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (threadIdx.x != 0 || u.status != MICD_OK) return;
    const uint32_t tl = u.table_log;
    const uint32_t *dt = u.tt_nb;
    const uint16_t *ds = u.tab_sym;
    BitR br;
    if (u.bits_off > u.comp_len || !br.init(u.comp_in + u.bits_off, u.comp_len - u.bits_off)) { u.status = MICD_ERR_CORRUPT; return; }
    uint16_t *out = u.tok;
    if (u.flavour == 1) {
        // decompress(), fsedecompressu16.go:267-377: run until the bits are used up and the
        // current state needs more; a non-zero final state contributes one more symbol.
        uint32_t state = br.get(tl);
        uint32_t n = 0;
        for (;;) {
            uint32_t e = dt[state];
            uint32_t nb = e >> 16;
            if (br.cursor == 0 && nb > 0) {
                if (state != 0) { if (n >= u.tok_cap) { u.status = MICD_ERR_CORRUPT; return; } out[n++] = ds[state]; }
                break;
            }
            uint32_t low = br.get(nb);
            if (br.over) { u.status = MICD_ERR_CORRUPT; return; }
            if (n >= u.tok_cap) { u.status = MICD_ERR_CORRUPT; return; }
            out[n++] = ds[state];
            state = (e & 0xFFFF) + low;
        }
        u.ntok = n;
        return;
    }
    const uint32_t lanes = (u.flavour == 108) ? 8 : u.flavour;
    uint32_t st[8];
    for (uint32_t k = 0; k < lanes; k++) st[k] = br.get(tl);
    if (br.over) { u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t count = u.count;
    for (uint32_t i = 0; i < count; i++) {
        uint32_t k = i & (lanes - 1);
        uint32_t e = dt[st[k]];
        uint32_t low = br.get(e >> 16);
        if (br.over) { u.status = MICD_ERR_CORRUPT; return; }
        out[i] = ds[st[k]];
        st[k] = (e & 0xFFFF) + low;
    }
    u.ntok = count;
}

// RLE pull-iterator, rledecompressu16.go:59-85
struct RleIt {
    const uint16_t *in; uint32_t n, i; uint16_t mid, c, rec; bool err;
    __device__ uint16_t rd() { if (i >= n) { err = true; return 0; } return in[i++]; }
    __device__ uint16_t next() {
        if (c > 0 && c < mid) { c--; return rec; }
        if (c == 0 || c == mid) {
            c = rd();
            if (c <= mid) { rec = rd(); c--; return rec; }
        }
        uint16_t v = rd();
        c--;
        return v;
    }
};

// v0: one lane expands the tokens and inverts the predictor.  grid = units, block = 64.
__global__ void __launch_bounds__(64) k_dec_pixels_serial(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (threadIdx.x != 0 || u.status != MICD_OK) return;
    if (u.ntok < 1) { u.status = MICD_ERR_CORRUPT; return; }
    RleIt r; r.in = u.tok; r.n = u.ntok; r.i = 1; r.c = 0; r.rec = 0; r.err = false;
    int d0 = mic_len16(u.tok[0]);                                      // rledecompressu16.go:21-25
    if (d0 == 0) { u.status = MICD_ERR_CORRUPT; return; }
    r.mid = (uint16_t)((1u << (d0 - 1)) - 1);
    uint16_t max_value = r.next();                                     // deltarlecompressu16.go:71
    int depth = mic_len16(max_value);
    if (r.err || depth == 0) { u.status = MICD_ERR_CORRUPT; return; }
    const uint16_t thr = (uint16_t)((1u << (depth - 1)) - 1);
    const uint16_t delim = (uint16_t)((1u << depth) - 1);
    uint16_t *out = u.px_out;
    const int w = u.w, h = u.h;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            size_t idx = (size_t)y * w + x;
            uint16_t v = r.next();
            if (v == delim) {
                out[idx] = r.next();
            } else {
                int32_t diff = (int32_t)v - (int32_t)thr;
                int32_t prev = 0; int div = 0;
                if (x > 0) { prev = out[idx - 1]; div++; }
                if (y > 0) { prev += out[idx - w]; div++; }
                if (div == 2) prev >>= 1;
                out[idx] = (uint16_t)(prev + diff);
            }
            if (r.err) { u.status = MICD_ERR_CORRUPT; return; }
        }
    }
}

void mic_launch_decode(MicUnit *d_units, int n, hipStream_t stream, int variant, MicTimer *t) {
    (void)variant;
    if (t) t->mark("k_dec_tables");
    hipLaunchKernelGGL(k_dec_tables, dim3(n), dim3(64), 0, stream, d_units);
    if (t) t->mark("k_dec_tans_serial");
    hipLaunchKernelGGL(k_dec_tans_serial, dim3(n), dim3(64), 0, stream, d_units);
    if (t) t->mark("k_dec_pixels_serial");
    hipLaunchKernelGGL(k_dec_pixels_serial, dim3(n), dim3(64), 0, stream, d_units);
    if (t) t->mark("end");
}
