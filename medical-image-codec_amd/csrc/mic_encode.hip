// mic_encode.hip -- encode kernels of the MIC unit codec for gfx950.
//
//   k_enc_tokens   Delta(avg) predictor + escape + RLE tokeniser   (deltarlecompressu16.go:24-67,
//                                                                  rlecompressu16.go:24-83)
//   k_enc_hist     16-bit-alphabet histogram of the token stream   (fsecompressu16.go:438-462)
//   k_enc_tables   gates, tableLog, normalise, NCount, CTable      (fse2state.go:22-52 et al.)
//   k_enc_tans     N-state tANS encode + framing + fallback chain  (fse2state.go:122-199, ...)
//
// Launch shape: blockIdx.x = unit.  See DESIGN.md for the HBM layout.
#include "mic_dev.h"
#include "mic_fse_tables.h"
#include "mic_launch.h"

// ------------------------------------------------------------------------------------------
// RLE tokeniser, exact statement of RleCompressU16.{Encode,Flush} without the side buffer:
// pending literals are written behind a reserved header slot and the slot is patched (or the
// reservation rolled back) when the reference would have emitted the chunk.
struct RleTok {
    uint16_t *out; uint32_t cap, n;   // token stream and fill
    uint32_t hdr_pos;                 // reserved header slot of the open literal chunk
    uint32_t bc;                      // len(r.b)
    uint16_t mid, p1, p2, v;          // midCount, last two symbols, value of the same-run
    bool same, overflow;

    __device__ void put(uint32_t pos, uint16_t x) { if (pos < cap) out[pos] = x; else overflow = true; }
    __device__ void append(uint16_t x) {
        if (!same) {
            if (bc == 0) { hdr_pos = n; n++; }
            put(hdr_pos + 1 + bc, x);
            n = hdr_pos + 1 + bc + 1;
        }
        bc++;
        p2 = p1; p1 = x;
    }
    __device__ void close_literals(uint32_t keep) {  // emit all but the last `keep` pending literals
        uint32_t cnt = bc - keep;
        put(hdr_pos, (uint16_t)(mid + (uint16_t)cnt));
        n = hdr_pos + 1 + cnt;
    }
    __device__ void encode(uint16_t x) {              // rlecompressu16.go:24-70
        if (bc < 2) { append(x); return; }
        if (p2 == p1 && p1 == x) {
            if (!same) {
                if (bc > 2) close_literals(2);
                else n = hdr_pos;                     // the two pending symbols join the run
                bc = 2; v = x;                        // r.b = r.b[bc-2:]
            }
            same = true;
        } else {
            if (same && bc > 2) {
                put(n, (uint16_t)bc); put(n + 1, v); n += 2;
                bc = 0;
            }
            same = false;
        }
        if ((int)bc >= (int)(uint16_t)(mid - 1)) {    // count overflow, :57-67
            if (same) {
                put(n, (uint16_t)(bc - 2)); put(n + 1, v); n += 2;
            } else {
                uint16_t a = (hdr_pos + 1 + bc - 2 < cap) ? out[hdr_pos + 1 + bc - 2] : 0;
                uint16_t b = (hdr_pos + 1 + bc - 1 < cap) ? out[hdr_pos + 1 + bc - 1] : 0;
                close_literals(2);
                hdr_pos = n; n++;
                put(hdr_pos + 1, a); put(hdr_pos + 2, b);
                n = hdr_pos + 3;
            }
            bc = 2;
        }
        append(x);
    }
    __device__ void flush() {                          // rlecompressu16.go:72-83
        if (bc > 0) {
            if (same) { put(n, (uint16_t)bc); put(n + 1, v); n += 2; }
            else close_literals(0);
        }
    }
};

// v0: one lane walks the unit.  grid = units, block = 64.
__global__ void __launch_bounds__(64) k_enc_tokens_serial(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (threadIdx.x != 0) return;
    u.status = MICD_OK; u.ntok = 0; u.blob_len = 0; u.nstates_used = 0;
    int depth = mic_len16(u.max_value);
    if (u.w <= 0 || u.h <= 0) { u.status = MICD_ERR_ARGS; return; }
    // max_value < 8 (midCount < 7): the reference's RLE chunking degenerates (empty literal
    // chunks / slice panics in rlecompressu16.go:57-67), so there is no behaviour to match.
    if (depth < 4) { u.status = MICD_ERR_UNSUPPORTED; return; }
    const uint16_t thr = (uint16_t)((1u << (depth - 1)) - 1);
    const uint16_t delim = (uint16_t)((1u << depth) - 1);
    RleTok r;
    r.out = u.tok; r.cap = u.tok_cap; r.n = 0; r.hdr_pos = 0; r.bc = 0;
    r.mid = (uint16_t)((1u << (mic_len16(delim) - 1)) - 1);
    r.p1 = r.p2 = r.v = 0; r.same = false; r.overflow = false;
    r.put(0, delim); r.n = 1;                          // rlecompressu16.go:21
    r.encode(u.max_value);                             // deltarlecompressu16.go:29
    const uint16_t *in = u.px_in;
    const int w = u.w, h = u.h;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            size_t idx = (size_t)y * w + x;
            int32_t prev = 0; int div = 0;
            if (x > 0) { prev = in[idx - 1]; div++; }
            if (y > 0) { prev += in[idx - w]; div++; }
            if (div == 2) prev >>= 1;
            uint16_t val = in[idx];
            int32_t diff = (int32_t)val - prev;
            int32_t m = diff >> 31;
            int32_t ad = (diff ^ m) - m;
            if ((uint16_t)ad >= thr) { r.encode(delim); r.encode(val); }
            else r.encode((uint16_t)((int32_t)thr + diff));
        }
    }
    r.flush();
    if (r.overflow) { u.status = MICD_ERR_CAPACITY; return; }
    u.ntok = r.n;
}

// ------------------------------------------------------------------------------------------
// Histogram of the token stream.  grid = (blocks_per_unit, units), block = 256.
// hist must be zero on entry (the launcher memsets the workspace slab).
__global__ void __launch_bounds__(256) k_enc_hist(MicUnit *units) {
    MicUnit &u = units[blockIdx.y];
    if (u.status != MICD_OK) return;
    const uint32_t n = u.ntok;
    const uint16_t *tok = u.tok;
    uint32_t *hist = u.hist;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        atomicAdd(&hist[tok[i]], 1u);
}

// ------------------------------------------------------------------------------------------
// Gates + tables.  grid = units, block = 256 (parallel max / symbolLen reduction, then lane 0).
__global__ void __launch_bounds__(256) k_enc_tables(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK) return;
    __shared__ uint32_t s_max[256], s_len[256];
    uint32_t m = 0, sl = 0;
    for (uint32_t i = threadIdx.x; i <= MIC_MAXSYM; i += blockDim.x) {
        uint32_t c = u.hist[i];
        if (c) { if (c > m) m = c; if (i + 1 > sl) sl = i + 1; }
    }
    s_max[threadIdx.x] = m; s_len[threadIdx.x] = sl;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            if (s_max[threadIdx.x + s] > s_max[threadIdx.x]) s_max[threadIdx.x] = s_max[threadIdx.x + s];
            if (s_len[threadIdx.x + s] > s_len[threadIdx.x]) s_len[threadIdx.x] = s_len[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const uint32_t n = u.ntok;
    u.max_count = s_max[0]; u.symbol_len = s_len[0];
    // Gate order of FSECompressU16* (fse2state.go:23-42); the length gate depends on the
    // flavour and is applied per attempt in k_enc_tans.
    if (n <= 1) { u.status = MICD_ERR_INCOMPRESSIBLE; return; }
    if (u.max_count == n) { u.status = MICD_ERR_USE_RLE; return; }
    if (u.max_count == 1 || u.max_count < (n >> 15)) { u.status = MICD_ERR_INCOMPRESSIBLE; return; }
    u.table_log = mic_optimal_table_log(n, u.symbol_len);
    int rc = mic_normalize_count(u.hist, u.norm, u.symbol_len, n, u.table_log);
    if (rc) { u.status = rc; return; }
    if (u.blob_cap < 6 + 8) { u.status = MICD_ERR_CAPACITY; return; }
    rc = mic_write_ncount(u.norm, u.symbol_len, u.table_log, u.blob + 6, u.blob_cap - 6, &u.hdr_len);
    if (rc) { u.status = rc; return; }
    rc = mic_build_ctable(u);
    if (rc) { u.status = rc; return; }
}

// ------------------------------------------------------------------------------------------
// LSB-first bit writer (bitwriter.go:50-53, :162-168).  The reference's flush32 cadence never
// changes content, so the stream is the plain concatenation of (value & mask(nb), nb) chunks.
struct BitW {
    uint8_t *out; uint32_t cap, len; uint64_t acc; uint32_t nbits; bool overflow;
    __device__ void add(uint32_t value, uint32_t nb) {
        uint32_t m = nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1);
        acc |= (uint64_t)(value & m) << nbits;
        nbits += nb;
        while (nbits >= 8) {
            if (len < cap) out[len++] = (uint8_t)acc; else overflow = true;
            acc >>= 8; nbits -= 8;
        }
    }
    __device__ void close() {
        add(1, 1);
        if (nbits > 0) {
            if (len < cap) out[len++] = (uint8_t)acc; else overflow = true;
            acc = 0; nbits = 0;
        }
    }
};

// v0: one lane encodes the whole stream (all N chains interleaved exactly as the reference
// emits them).  grid = units, block = 64.
__global__ void __launch_bounds__(64) k_enc_tans_serial(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (threadIdx.x != 0 || u.status != MICD_OK) return;
    const uint32_t n = u.ntok;
    const uint16_t *src = u.tok;
    const uint32_t tl = u.table_log;
    // Fallback chain of CompressSingleFrame{,4State,8State} (multiframecompress.go:15-93).
    for (int lanes = u.nstates; lanes >= 1; lanes >>= 1) {
        int rc = MICD_OK;
        // length gates: fse8state.go:32, fse4state.go:25, fse2state.go:23, fsecompressu16.go:20
        if (n <= (uint32_t)(lanes - 1) || n <= 1) rc = MICD_ERR_INCOMPRESSIBLE;
        else if (n <= 2 && lanes <= 2) rc = MICD_ERR_INTERNAL;          // "src too small"
        uint32_t pos = (lanes == 1) ? 0 : 6;
        uint32_t out_len = 0;
        if (rc == MICD_OK) {
            uint8_t *dst = u.blob + 6 + u.hdr_len;                     // bitstream right behind the NCount
            BitW bw; bw.out = dst; bw.cap = u.blob_cap - 6 - u.hdr_len; bw.len = 0; bw.acc = 0; bw.nbits = 0; bw.overflow = false;
            uint32_t st[8];
            for (int k = 0; k < lanes; k++) st[k] = 1u << tl;
            for (uint32_t ip = n; ip > 0; ip--) {
                uint32_t idx = ip - 1;
                uint32_t k = idx & (uint32_t)(lanes - 1);
                uint16_t sym = src[idx];
                uint32_t dnb = u.tt_nb[sym]; int32_t dfind = u.tt_find[sym];
                uint32_t nb = (st[k] + dnb) >> 16;                      // fsecompressu16.go:95-100
                bw.add(st[k], nb);
                st[k] = u.state_tab[(int32_t)(st[k] >> (nb & 31)) + dfind];
            }
            for (int k = lanes - 1; k >= 0; k--) bw.add(st[k], tl);     // final states, last lane first
            bw.close();
            if (bw.overflow) rc = MICD_ERR_CAPACITY;
            else if ((uint64_t)u.hdr_len + bw.len >= (uint64_t)n * 2) rc = MICD_ERR_INCOMPRESSIBLE; // fse2state.go:58-60
            else out_len = pos + u.hdr_len + bw.len;
        }
        if (rc == MICD_OK) {
            if (lanes != 1) {
                u.blob[0] = 0xFF;
                u.blob[1] = lanes == 2 ? 0x02 : lanes == 4 ? 0x04 : 0x84;
                u.blob[2] = (uint8_t)n; u.blob[3] = (uint8_t)(n >> 8);
                u.blob[4] = (uint8_t)(n >> 16); u.blob[5] = (uint8_t)(n >> 24);
            }
            // 1-state streams have no prefix: the blob starts at u.blob + 6
            u.blob_len = out_len; u.nstates_used = lanes; u.status = MICD_OK;
            return;
        }
        u.status = rc;                                                  // error of the last attempt
        if (lanes == 1) return;
        if (rc == MICD_ERR_CAPACITY) return;
        u.status = MICD_OK;                                             // try the next flavour
    }
}

// ------------------------------------------------------------------------------------------
// Compaction: copy every unit's staging blob to its final offset.  grid = (chunks, units).
// dst_off[i] = byte offset of unit i in `dst` (exclusive scan of blob_len, done by k_scan_lens).
__global__ void __launch_bounds__(256) k_enc_pack(const MicUnit *units, const uint64_t *dst_off, uint8_t *dst) {
    const MicUnit &u = units[blockIdx.y];
    if (u.status != MICD_OK) return;
    const uint8_t *src = u.blob + (u.nstates_used == 1 ? 6 : 0);
    uint8_t *d = dst + dst_off[blockIdx.y];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < u.blob_len; i += gridDim.x * blockDim.x)
        d[i] = src[i];
}

// exclusive scan of blob lengths (single block; n units <= a few 100k)
__global__ void __launch_bounds__(1024) k_scan_lens(const MicUnit *units, int n, uint64_t *dst_off) {
    __shared__ uint64_t s_part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    uint64_t sum = 0;
    for (int i = lo; i < hi; i++) sum += (units[i].status == MICD_OK) ? units[i].blob_len : 0;
    s_part[t] = sum;
    __syncthreads();
    if (t == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; i++) { uint64_t v = s_part[i]; s_part[i] = run; run += v; }
        dst_off[n] = run;
    }
    __syncthreads();
    uint64_t run = s_part[t];
    for (int i = lo; i < hi; i++) {
        dst_off[i] = run;
        run += (units[i].status == MICD_OK) ? units[i].blob_len : 0;
    }
}

// ------------------------------------------------------------------------------------------
// launchers
void mic_launch_encode(MicUnit *d_units, int n, hipStream_t stream, int variant, MicTimer *t) {
    (void)variant;
    if (t) t->mark("k_enc_tokens_serial");
    hipLaunchKernelGGL(k_enc_tokens_serial, dim3(n), dim3(64), 0, stream, d_units);
    if (t) t->mark("k_enc_hist");
    hipLaunchKernelGGL(k_enc_hist, dim3(64, n), dim3(256), 0, stream, d_units);
    if (t) t->mark("k_enc_tables");
    hipLaunchKernelGGL(k_enc_tables, dim3(n), dim3(256), 0, stream, d_units);
    if (t) t->mark("k_enc_tans_serial");
    hipLaunchKernelGGL(k_enc_tans_serial, dim3(n), dim3(64), 0, stream, d_units);
    if (t) t->mark("end");
}
void mic_launch_pack(const MicUnit *d_units, int n, uint64_t *d_off, uint8_t *d_dst, hipStream_t stream, MicTimer *t) {
    if (t) t->mark("k_scan_lens");
    hipLaunchKernelGGL(k_scan_lens, dim3(1), dim3(1024), 0, stream, d_units, n, d_off);
    if (t) t->mark("k_enc_pack");
    hipLaunchKernelGGL(k_enc_pack, dim3(32, n), dim3(256), 0, stream, d_units, d_off, d_dst);
    if (t) t->mark("end");
}
