// mic_decode.hip -- decode kernels of the MIC unit codec for gfx950.
//
//   k_dec_tans_serial / k_dec_tans_gl   what the lane-per-state kernels of mic_decode_ls.hip leave: 1-state streams
//                  (fsedecompressu16.go:267-377) and tableLog-16 tables with 0-bit entries (table in L2), plus the decode launcher.
//   Tables: mic_tables.hip; lane-per-state tANS: mic_decode_ls.hip; RLE expansion and inverse predictor: mic_decode_px.hip.
//   (Round 1's decoders k_dec_tans_duo / k_dec_tans_lds and the serial kernel generation: tools/retired/.)
//
// Launch shape: blockIdx.x = unit.
#include "mic_dev.h"
#include "mic_fse_tables.h"
#include "mic_launch.h"


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
    if (u.ntok != 0) return;                                           // a fast variant already decoded it
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





// ==========================================================================================
// Two streams per wave (tableLog <= 13).  A SIMD issues one instruction per 4-cycle turn whatever its kind
// (measured: a stream runs at 134 cycles per symbol pair alone, 142 with one wave per SIMD, ~214 with two, and a
// launch waits for its slowest wave), so with more than one wave per SIMD the decode is bound by instructions
// issued per pair -- about 21 -- and every lane of those instructions computes the same thing.  Here lanes 0-31 run
// stream 2w and lanes 32-63 stream 2w+1 through the SAME instructions: per-half values (table base, ring base, bit
// position, states, tableLog) live in VGPRs, the two tables / rings / stages sit side by side in LDS (35.5 KiB per
// wave: four waves = eight streams per CU, one wave per SIMD), and a pair of streams costs what one did.
// The halves differ in length: the loop runs for the longer one; the shorter one keeps decoding harmless garbage
// (every table entry is valid, the ring wraps, block loads are bounds-checked) behind a snapshot of its true end
// state, and its stores and header walk are switched off.  Units of another class in a pair are left to the
// single-stream kernels.  LDS bytes: ring A 0, stage A 1056, ring B 2048, stage B 3104, table A 3584, table B 19968.
#define T2_TAB0 3584u
#define T2_WAVE 0x9000u                            // LDS bytes of one wave (its bits stay clear of the ring-index mask 0x3FC)
#define T2_WAVES 4                                 // waves per group: one per SIMD by construction (four single-wave groups
                                                   // landed two-on-a-SIMD on a third of the CUs, and the younger wave starves)

// ==========================================================================================
// The same decoder with the transition table left in HBM / L2 (u32 entries newState | nbBits << 16, as the
// tables kernel writes them): for tableLog 16 -- every 16-bit-depth frame, CT above all -- the table has
// 65536 entries and a 17-bit nextState, too much for LDS.  A chain step is then an L2 round trip instead
// of an LDS one, but a wave needs only the 1.3 KiB ring + stage, so a CU holds as many streams as it has
// wave slots and hides that latency with them.  64-symbol chunks (a chunk may take 64 x 16 bits off the
// ring).  Takes every N-state stream the LDS classes left (tableLog > 15); 1-state streams stay serial.
#define TD_RING 256                               // dwords of a stream's bit-window ring
#define TG_CH 64
template <int N>
__global__ void __launch_bounds__(64) k_dec_tans_gl(MicUnit *units) {
    __shared__ uint32_t s_gl[TD_RING + 4 + TG_CH / 2];
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK) return;
    const uint32_t flav = u.flavour;
    if (flav == 1 || ((flav == 108) ? 8u : flav) != (uint32_t)N) return;
    if (u.ntok != 0) return;                                            // already decoded by an LDS class
    const uint32_t tl = u.table_log;
    if (u.bits_off >= u.comp_len) { if (threadIdx.x == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t len = u.comp_len - u.bits_off;
    if (len >= (1u << 27)) return;                                      // bit positions are 32-bit here; the serial kernel takes it
    const uint32_t lane = threadIdx.x;
    uint32_t *ring = s_gl;
    uint32_t *stage = ring + TD_RING + 4;
    typedef const __attribute__((address_space(1))) uint16_t *gcu16;
    typedef const __attribute__((address_space(1))) uint32_t *gcu32;
    typedef __attribute__((address_space(1))) uint16_t *gu16;
    typedef __attribute__((address_space(1))) uint32_t *gu32;
    const gcu32 tabg = (gcu32)u.tt_nb;
    const gcu16 symg = (gcu16)u.tab_sym;
    const uint32_t count = u.count;
    const gu16 out = (gu16)u.tok;
    const uint8_t *bs = u.comp_in + u.bits_off;
    const uint32_t last = bs[len - 1];
    if (last == 0) { if (lane == 0) u.status = MICD_ERR_CORRUPT; return; }  // bitreader.go:36-38
    const uintptr_t addr = (uintptr_t)bs;
    const uint32_t sb = (uint32_t)(addr & 3);
    const gcu32 g = (gcu32)(addr - sb);
    const int32_t cur0 = (int32_t)(8u * (len - 1) + (uint32_t)(31 - __clz(last)) + 8u * sb);
    const int32_t top_dw = (cur0 - 1) >> 5;
    auto load_blk = [&](int32_t b) -> uint32_t {
        const int32_t idx = b * 64 + (int32_t)lane;
        return (b >= 0 && idx <= top_dw) ? __builtin_nontemporal_load(g + idx) : 0u;
    };
    auto store_blk = [&](int32_t b, uint32_t v) {
        const uint32_t slot = ((uint32_t)b & 3u) * 64u + lane;
        ring[slot] = v;
        if (slot == 0) ring[TD_RING] = v;
    };
    int32_t q = cur0 - 32;
    int32_t blk = (q >> 5) >> 6;
    store_blk(blk + 1, load_blk(blk + 1));
    store_blk(blk, load_blk(blk));
    store_blk(blk - 1, load_blk(blk - 1));
    uint32_t pf = load_blk(blk - 2);
    __syncthreads();
    auto window = [&](int32_t qq) -> uint32_t {
        const uint32_t *w = ring + (((uint32_t)qq >> 5) & (TD_RING - 1));
        return __builtin_amdgcn_alignbit(w[1], w[0], (uint32_t)qq);
    };
    auto take = [](uint32_t hi, uint32_t nb) -> uint32_t { return (uint32_t)(((uint64_t)hi << nb) >> 32); };   // top nb bits, nb = 0 .. 32
    uint32_t st[N];
#pragma unroll
    for (int p = 0; p < N; p++) { st[p] = window(q) >> (32u - tl); q -= (int32_t)tl; }
    auto group = [&](uint32_t *stage_w) {
        uint32_t e[N];
#pragma unroll
        for (int k = 0; k < N; k++) e[k] = tabg[st[k]];
#pragma unroll
        for (int p = 0; p < N; p += 2) {
            const uint32_t hi = window(q);
            stage_w[p >> 1] = st[p] | (st[p + 1] << 16);
            const uint32_t nb0 = e[p] >> 16, nb1 = e[p + 1] >> 16;           // <= 16 each
            st[p] = (e[p] & 0xFFFFu) + take(hi, nb0);
            st[p + 1] = (e[p + 1] & 0xFFFFu) + take(hi << nb0, nb1);
            q -= (int32_t)(nb0 + nb1);
        }
    };
    auto single = [&](int k, uint16_t *stage_h) {
        const uint32_t e = tabg[st[k]];
        const uint32_t hi = window(q);
        *stage_h = (uint16_t)st[k];
        const uint32_t nb = e >> 16;
        st[k] = (e & 0xFFFFu) + take(hi, nb);
        q -= (int32_t)nb;
    };
    // header walker, as in k_dec_tans_lds
    bool w_on = u.mode == 0 && u.seg != nullptr;
    bool w_err = false;
    uint32_t w_pos = 0, w_out = 0, w_nseg = 0, w_mid = 0;
    const uint32_t w_symcap = min(u.sym_cap, 2u * (uint32_t)u.w * (uint32_t)u.h + 2u), w_segcap = u.seg_cap;
    typedef uint32_t w_v2 __attribute__((ext_vector_type(2)));
    __attribute__((address_space(1))) w_v2 *const w_seg = (__attribute__((address_space(1))) w_v2 *)u.seg;
    auto walk = [&](uint32_t cend, auto get) {
        while (w_on && w_pos < cend) {
            const uint32_t h = get(w_pos);
            if (w_pos == 0) {
                const int d0 = mic_len16((uint16_t)h);
                if (d0 == 0) { w_on = false; w_err = true; break; }
                w_mid = (1u << (d0 - 1)) - 1; w_pos = 1;
                continue;
            }
            if (w_out >= w_symcap) { w_on = false; break; }
            if (h == 0 || w_nseg >= w_segcap) { w_on = false; w_err = true; break; }
            if (h <= w_mid) {
                if (w_pos + 1 >= count) { w_on = false; w_err = true; break; }
                if (lane == 0) { w_v2 r; r.x = (w_pos + 1) | 0x80000000u; r.y = w_out; w_seg[w_nseg] = r; }
                w_nseg++; w_out += h; w_pos += 2;
            } else {
                if (lane == 0) { w_v2 r; r.x = w_pos + 1; r.y = w_out; w_seg[w_nseg] = r; }
                w_nseg++; w_out += h - w_mid; w_pos += 1 + (h - w_mid);
            }
        }
    };
    constexpr uint32_t G = TG_CH / N;
    const uint32_t chunks = count / TG_CH;
    uint32_t pend_lo = 0, pend_hi = 0; bool have_pend = false;
    uint32_t obase = 0;
    auto flush_pend = [&]() {
        if (!have_pend) return;
        const uint32_t pk = pend_lo | (pend_hi << 16);
        if (lane < TG_CH / 2) __builtin_nontemporal_store(pk, (gu32)out + obase + lane);
        const uint32_t cb = obase * 2;
        walk(cb + TG_CH, [&](uint32_t pos) -> uint32_t {
            const uint32_t rel = pos - cb;
            const uint32_t pair = __builtin_amdgcn_readlane(pk, rel >> 1);
            return (rel & 1) ? (pair >> 16) : (pair & 0xFFFFu);
        });
    };
    for (uint32_t ch = 0; ch < chunks; ch++) {
#pragma unroll 4
        for (uint32_t gi = 0; gi < G; gi++) group(stage + gi * (N / 2));
        store_blk(blk - 2, pf);
        flush_pend();
        blk = (q >> 5) >> 6;
        pf = load_blk(blk - 2);
        const uint32_t s2 = stage[lane & (TG_CH / 2 - 1)];
        pend_lo = symg[s2 & 0xFFFF]; pend_hi = symg[s2 >> 16];
        have_pend = true; obase = ch * (TG_CH / 2);
    }
    flush_pend();
    {
        const uint32_t done = chunks * TG_CH;
        const uint32_t rem = count - done;
        uint32_t k = 0;
        for (; k + N <= rem; k += N) group(stage + (k / 2));
        uint16_t *st16 = (uint16_t *)stage;
#pragma unroll
        for (int j = 0; j < N - 1; j++) if (k + (uint32_t)j < rem) single(j, st16 + k + j);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        uint32_t tv0 = 0;
        if (lane < rem) { tv0 = symg[st16[lane]]; out[done + lane] = (uint16_t)tv0; }
        walk(count, [&](uint32_t pos) -> uint32_t { return __builtin_amdgcn_readlane(tv0, pos - done); });
    }
    if (lane == 0) {
        if (q + 32 - (int32_t)(8u * sb) < 0) u.status = MICD_ERR_CORRUPT;   // bitreader.go:113-120
        else {
            u.ntok = count;
            if (u.mode == 0 && u.seg != nullptr && !w_err) { u.nseg = w_nseg; u.nsym = min(w_out, w_symcap); u.walk_ok = 1; }
        }
    }
}


void mic_launch_decode(MicUnit *d_units, int n, hipStream_t stream, int variant, MicTimer *t, int *d_cls, uint32_t pred_mask, uint32_t cls_mask) {
    const bool any_grad = (variant & MIC_VARIANT_GRAD) != 0;
    if (t) t->mark("k_dec_tables_wg");
    mic_launch_dec_tables(d_units, n, stream);
    // lane-per-state kernels over compacted per-class lists (mic_decode_ls.hip): every table size has its class; what they leave
    // (a tableLog-16 table with 0-bit entries, 1-state streams) falls through to the kernels below, which skip decoded units
    mic_launch_dec_tans_ls(d_units, n, d_cls + MIC_CLS_HEAD, d_cls, stream, t, cls_mask);
    if (t) t->mark("k_dec_tans_gl<2>");
    hipLaunchKernelGGL(k_dec_tans_gl<2>, dim3(n), dim3(64), 0, stream, d_units);
    if (t) t->mark("k_dec_tans_gl<4,8>");
    hipLaunchKernelGGL(k_dec_tans_gl<4>, dim3(n), dim3(64), 0, stream, d_units);
    hipLaunchKernelGGL(k_dec_tans_gl<8>, dim3(n), dim3(64), 0, stream, d_units);
    if (t) t->mark("k_dec_tans_serial");
    hipLaunchKernelGGL(k_dec_tans_serial, dim3(n), dim3(64), 0, stream, d_units);
    mic_launch_decode_pixels(d_units, n, stream, t, any_grad, pred_mask);
    if (t) t->mark("end");
}
