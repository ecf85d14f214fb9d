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

// grid = units, block = 256.  The NCount header is a serial bit-parse and the table build has
// data-dependent loops (mic_fse_tables.h), so one lane runs them -- on LDS copies: the first
// 16 KiB of the blob (the header is at most symbolLen*tableLog/8+3 bytes), norm[], the symbol
// spread and the per-symbol counters.  The finished table goes to HBM as stores.
#define DT_STAGE_BYTES 16384
#define DT_SMALL_SYMS 8192
#define DT_SMALL_TL 13
__global__ void __launch_bounds__(256) k_dec_tables(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    __shared__ uint8_t s_in[DT_STAGE_BYTES];
    __shared__ int32_t s_norm[DT_SMALL_SYMS];
    __shared__ int32_t s_next[DT_SMALL_SYMS];
    __shared__ uint16_t s_tabsym[1 << DT_SMALL_TL];
    __shared__ uint32_t s_flag[2];
    const uint32_t len = u.comp_len;
    if (u.comp_in) for (uint32_t i = threadIdx.x; i < len && i < DT_STAGE_BYTES; i += blockDim.x) s_in[i] = u.comp_in[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        s_flag[0] = 0;
        u.status = MICD_OK; u.ntok = 0;
        do {
            if ((u.mode == 0 && (u.w <= 0 || u.h <= 0)) || !u.comp_in) { u.status = MICD_ERR_ARGS; break; }
            const uint8_t *b = (len <= DT_STAGE_BYTES) ? s_in : u.comp_in;   // short blobs parse entirely from LDS
            // FSEDecompressU16Auto, fse2state.go:102-116
            uint32_t flavour = 1;
            if (len >= 2 && s_in[0] == 0xFF) {
                if (s_in[1] == 0x84) flavour = 8;
                else if (s_in[1] == 0x08) flavour = 108;
                else if (s_in[1] == 0x04) flavour = 4;
                else if (s_in[1] == 0x02) flavour = 2;
            }
            uint32_t count = 0, off = 0;
            if (flavour != 1) {
                if (len < 6) { u.status = MICD_ERR_CORRUPT; break; }
                count = (uint32_t)s_in[2] | ((uint32_t)s_in[3] << 8) | ((uint32_t)s_in[4] << 16) | ((uint32_t)s_in[5] << 24);
                off = 6;
                if (count > u.tok_cap) { u.status = MICD_ERR_CORRUPT; break; }
            }
            u.flavour = flavour; u.count = count;
            // Parse from the staged bytes when the whole header is certain to be inside them
            // (it ends before the stage does unless the parser runs to the stage's last 8 bytes).
            uint32_t used = 0, symbol_len = 0, tl = 0;
            int rc;
            bool lds_norm = false;
            if (len > DT_STAGE_BYTES) {
                rc = mic_read_ncount(s_in + off, DT_STAGE_BYTES - off, s_norm, &symbol_len, &tl, &used, DT_SMALL_SYMS);
                if (rc == MICD_OK && used + 8 < DT_STAGE_BYTES - off && symbol_len <= DT_SMALL_SYMS) lds_norm = true;
                else rc = mic_read_ncount(u.comp_in + off, len - off, u.norm, &symbol_len, &tl, &used, 65536);
            } else {
                rc = mic_read_ncount(b + off, len - off, s_norm, &symbol_len, &tl, &used, DT_SMALL_SYMS);
                if (rc == MICD_OK) lds_norm = true;
                else if (rc == MICD_ERR_UNSUPPORTED) rc = mic_read_ncount(b + off, len - off, u.norm, &symbol_len, &tl, &used, 65536);
            }
            if (rc) { u.status = rc; break; }
            u.symbol_len = symbol_len; u.table_log = tl;
            u.bits_off = off + used;
            MicUnit v = u;
            const bool small = lds_norm && tl <= DT_SMALL_TL;
            if (lds_norm) v.norm = s_norm;
            if (small) { v.tt_find = s_next; v.tab_sym = s_tabsym; }
            rc = (flavour == 108) ? mic_build_rans_dtable(v) : mic_build_dtable(v);
            if (rc) { u.status = rc; break; }
            u.zero_bits = v.zero_bits;
            if (small) s_flag[0] = 1u << tl;
        } while (0);
    }
    __syncthreads();
    // symbol-of-state table was built in LDS: copy it out
    const uint32_t nsz = s_flag[0];
    for (uint32_t i = threadIdx.x; i < nsz; i += blockDim.x) u.tab_sym[i] = s_tabsym[i];
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
    if (threadIdx.x != 0 || u.status != MICD_OK || u.mode != 0) return;
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


// ==========================================================================================
// Fast tANS decode: one wave per unit, transition table in LDS.
//
// The N states of an N-state stream (fse2state.go:203-308, fse4state.go:195-353,
// fse8state.go:230-380, rans8state.go:221-412) share ONE reverse bitstream, so the chain
//   state_k -> table entry -> (nbBits, newState) -> bits at the running position -> state_k'
// is serial per stream and a stream cannot be split (the format has no resynchronisation
// points).  A lone wave issues one instruction every ~4-5 cycles whatever its
// kind (tools/ubench_chain.hip: dependent ds_read 70 cycles, +30..45 cycles per added branch
// or handful of scalar ops), so the loop is written for instruction COUNT:
//   * all N look-ups of a group are issued together (one LDS round trip per N symbols);
//   * the transition entry holds newState, nbBits and 32-nbBits ready-made: a state update is
//     shift + add (3 VALU ops for the second state of a pair);
//   * every lane computes the same values (no cross-lane traffic on the chain); the bit window
//     and its refill live in SGPRs, the refill is branch-free but for the buffer switch;
//   * the stream is prefetched 64 dwords per load into a lane-distributed register buffer;
//   * the loop stages STATES (16 bit), not symbols; every 128 symbols all 64 lanes translate
//     state -> symbol through the L2-resident symbol table and store 256 bytes coalesced, one
//     chunk behind the chain so the gather latency is hidden.
// LDS: chain[2^tl] u32 = newState << 16 | (32-nbBits) << 8 | nbBits ; stage[2][64] u32.
// ZB = table has 0-bit entries (zeroBits, fsedecompressu16.go:214-216).
// grid = units, block = 64, dynamic LDS = 4 << tl_hi + 512.
// TLHI names the table-size class of the launch (tableLog in (TLHI-1 .. TLHI], or <= 13): the classes
// differ only in dynamic LDS, but distinct instantiations give each its own line in a kernel trace.
template <int N, bool ZB, int TLHI>
__global__ void __launch_bounds__(64) k_dec_tans_lds(MicUnit *units) {
    constexpr uint32_t tl_lo = (TLHI == 13) ? 5u : (uint32_t)TLHI, tl_hi = (uint32_t)TLHI;
    extern __shared__ uint32_t s_mem[];
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK) return;
    const uint32_t flav = u.flavour;
    if (flav == 1 || ((flav == 108) ? 8u : flav) != (uint32_t)N) return;
    const uint32_t tl = u.table_log;
    if (tl < tl_lo || tl > tl_hi) return;
    if ((u.zero_bits != 0) != ZB) return;
    if (u.ntok != 0) return;                                            // already decoded by another variant
    const uint32_t lane = threadIdx.x;
    const uint32_t size = 1u << tl;
    uint32_t *chain = s_mem;
    uint32_t *stage = s_mem + size;                                     // 2 x 64 dwords = 2 x 128 states
    {
        const uint32_t *dt = u.tt_nb;
        for (uint32_t p = lane; p < size; p += 64) {
            const uint32_t e = dt[p];                                   // newState | nbBits << 16
            const uint32_t nb = e >> 16;
            chain[p] = ((e & 0xFFFF) << 16) | ((32u - nb) << 8) | nb;
        }
    }
    __syncthreads();
    const uint16_t *symg = u.tab_sym;                                   // state -> symbol, L2 resident
    const uint32_t count = u.count;
    uint16_t *out = u.tok;
    if (u.bits_off >= u.comp_len) { if (lane == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint8_t *bs = u.comp_in + u.bits_off;
    const uint32_t len = u.comp_len - u.bits_off;
    // readfirstlane: the byte comes back in a VGPR; everything derived from it (cursor, window,
    // refill bookkeeping) is wave-uniform and must live in SGPRs / run on the scalar unit
    const uint32_t last = __builtin_amdgcn_readfirstlane((uint32_t)bs[len - 1]);
    if (last == 0) { if (lane == 0) u.status = MICD_ERR_CORRUPT; return; }  // bitreader.go:36-38
    const uint64_t total_bits = 8ull * (len - 1) + (uint32_t)(31 - __clz(last));
    // 4-byte aligned dword grid under the stream: grid bit 0 = LSB of g[0]
    const uintptr_t addr = (uintptr_t)bs;
    const uint32_t sb = (uint32_t)(addr & 3);
    const uint32_t *g = (const uint32_t *)(addr - sb);
    const uint64_t cur = total_bits + 8ull * sb;                        // unread bits are grid bits [8*sb, cur)
    const int32_t top_dw = (int32_t)((cur - 1) >> 5);                   // dword holding the top unread bit (< 2^27)
    int32_t di = top_dw;                                                // next dword to enter the window
    // lane-distributed stream buffer: buf_a = the 64 dwords of block di>>6, buf_b = the block below
    auto load_blk = [&](int32_t b) -> uint32_t {
        const int32_t idx = b * 64 + (int32_t)lane;
        return (b >= 0 && idx <= top_dw) ? g[idx] : 0u;
    };
    uint32_t buf_a = load_blk(di >> 6), buf_b = load_blk((di >> 6) - 1);
    uint64_t W = 0; uint32_t avail = 0;
    // take T (<= 32, uniform) bits off the window and top it up to >= 32 valid bits
    auto advance = [&](uint32_t T) {
        W <<= T; avail -= T;
        const uint32_t nd = __builtin_amdgcn_readlane(buf_a, di & 63);
        const bool need = avail < 32;
        const uint64_t add = (uint64_t)nd << ((32u - avail) & 63u);
        W |= need ? add : 0ull;
        avail += need ? 32u : 0u;
        const bool sw = need && ((di & 63) == 0);
        di -= need ? 1 : 0;
        if (sw) { buf_a = buf_b; buf_b = load_blk((di >> 6) - 1); }     // every 64 refills
    };
    {   // prime the window: the top dword holds 1..32 valid bits
        const uint32_t top = (uint32_t)(cur - 32ull * (uint64_t)di);
        W = (uint64_t)__builtin_amdgcn_readlane(buf_a, di & 63) << (64 - top);
        avail = top;
        const bool sw = (di & 63) == 0;
        di--;
        if (sw) { buf_a = buf_b; buf_b = load_blk((di >> 6) - 1); }
        advance(0);
    }
    uint32_t st[N];
    // initial states: state 0 first, tl bits each (fse2state.go:210-212)
#pragma unroll
    for (int p = 0; p < N; p += 2) {
        const uint32_t hi = (uint32_t)(W >> 32);
        st[p] = __builtin_amdgcn_ubfe(hi, 32u - tl, tl);
        st[p + 1] = __builtin_amdgcn_ubfe(hi, 32u - 2u * tl, tl);
        advance(2u * tl);
    }
    // one group = N symbols, states 0..N-1 in order; bits are handed out pair by pair (<= 32 a pair)
    auto group = [&](uint32_t *stage_w, uint32_t r) {                   // r = symbols wanted from this group (N, or fewer in the tail)
        uint32_t e[N];
#pragma unroll
        for (int k = 0; k < N; k++) e[k] = chain[st[k]];
#pragma unroll
        for (int p = 0; p < N; p += 2) {
            const uint32_t hi = (uint32_t)(W >> 32);
            stage_w[p >> 1] = st[p] | (st[p + 1] << 16);
            uint32_t nb0 = e[p] & 0xFF, nb1 = e[p + 1] & 0xFF;
            if (r < (uint32_t)N) { nb0 = ((uint32_t)p < r) ? nb0 : 0u; nb1 = ((uint32_t)p + 1 < r) ? nb1 : 0u; }
            uint32_t b0, b1;
            if (ZB || r < (uint32_t)N) {
                b0 = __builtin_amdgcn_ubfe(hi, 32u - nb0, nb0);
                b1 = __builtin_amdgcn_ubfe(hi, 32u - nb0 - nb1, nb1);
            } else {                                                    // nbBits >= 1: plain shifts
                b0 = hi >> ((e[p] >> 8) & 0xFF);
                b1 = (hi << nb0) >> ((e[p + 1] >> 8) & 0xFF);
            }
            st[p] = b0 + (e[p] >> 16);
            st[p + 1] = b1 + (e[p + 1] >> 16);
            advance(__builtin_amdgcn_readfirstlane(nb0 + nb1));
        }
    };
    constexpr uint32_t G = 128 / N;                                     // groups per 128-symbol chunk
    const uint32_t chunks = count / 128;
    uint32_t pend = 0; bool have_pend = false;                          // symbols gathered for the previous chunk
    uint32_t obase = 0;                                                 // dword index of the pending chunk in out
    for (uint32_t ch = 0; ch < chunks; ch++) {
        uint32_t *sw = stage + (ch & 1) * 64;
#pragma unroll 4
        for (uint32_t gi = 0; gi < G; gi++) group(sw + gi * (N / 2), N);
        // this chunk's 128 states are staged: write out the previous chunk, gather this one
        if (have_pend) ((uint32_t *)out)[obase + lane] = pend;
        const uint32_t s2 = sw[lane];
        pend = (uint32_t)symg[s2 & 0xFFFF] | ((uint32_t)symg[s2 >> 16] << 16);
        have_pend = true; obase = ch * 64;
    }
    if (have_pend) ((uint32_t *)out)[obase + lane] = pend;
    // tail: count % 128 symbols (fse2state.go:293-305 for the last partial group)
    {
        const uint32_t done = chunks * 128;
        const uint32_t rem = count - done;
        uint32_t *sw = stage + (chunks & 1) * 64;
        uint32_t k = 0;
        for (; k + N <= rem; k += N) group(sw + (k / 2), N);
        if (k < rem) group(sw + (k / 2), rem - k);
        const uint16_t *st16 = (const uint16_t *)sw;
        for (uint32_t j = lane; j < rem; j += 64) out[done + j] = symg[st16[j]];
    }
    if (lane == 0) {
        // bits taken = total - (bits still in the window + bits in dwords not yet fetched)
        const int64_t unread = (int64_t)avail + 32ll * ((int64_t)di + 1) - 8ll * sb;
        if (unread < 0) u.status = MICD_ERR_CORRUPT;                    // bitreader.go:113-120
        else u.ntok = count;
    }
}

template <int N, bool ZB, int TLHI>
static void launch_tans_lds_class(MicUnit *d_units, int n, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)k_dec_tans_lds<N, ZB, TLHI>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_dec_tans_lds<N, ZB, TLHI>), dim3(n), dim3(64), (4u << TLHI) + 512, stream, d_units);
}
template <int N, bool ZB>
static void launch_tans_lds(MicUnit *d_units, int n, hipStream_t stream, MicTimer *t, const char *name13) {
    if (t) t->mark(name13);
    launch_tans_lds_class<N, ZB, 13>(d_units, n, stream);
    if (t) t->mark("k_dec_tans_lds<other classes>");
    launch_tans_lds_class<N, ZB, 14>(d_units, n, stream);
    launch_tans_lds_class<N, ZB, 15>(d_units, n, stream);
}

void mic_launch_decode(MicUnit *d_units, int n, hipStream_t stream, int variant, MicTimer *t) {
    if (variant == 100) {
        if (t) t->mark("k_dec_tables");
        hipLaunchKernelGGL(k_dec_tables, dim3(n), dim3(256), 0, stream, d_units);
    } else {
        if (t) t->mark("k_dec_tables_wg");
        mic_launch_dec_tables(d_units, n, stream);
    }
    if (variant != 100) {
        launch_tans_lds<2, false>(d_units, n, stream, t, "k_dec_tans_lds<2,false,13>");
        launch_tans_lds<4, false>(d_units, n, stream, t, "k_dec_tans_lds<4,false,13>");
        launch_tans_lds<8, false>(d_units, n, stream, t, "k_dec_tans_lds<8,false,13>");
        launch_tans_lds<2, true>(d_units, n, stream, t, "k_dec_tans_lds<2,true,13>");
        launch_tans_lds<4, true>(d_units, n, stream, t, "k_dec_tans_lds<4,true,13>");
        launch_tans_lds<8, true>(d_units, n, stream, t, "k_dec_tans_lds<8,true,13>");
    }
    if (t) t->mark("k_dec_tans_serial");
    hipLaunchKernelGGL(k_dec_tans_serial, dim3(n), dim3(64), 0, stream, d_units);
    if (variant == 100) {
        if (t) t->mark("k_dec_pixels_serial");
        hipLaunchKernelGGL(k_dec_pixels_serial, dim3(n), dim3(64), 0, stream, d_units);
    } else {
        if (t) t->mark("k_dec_pixels_wg");
        mic_launch_decode_pixels(d_units, n, stream);
    }
    if (t) t->mark("end");
}
