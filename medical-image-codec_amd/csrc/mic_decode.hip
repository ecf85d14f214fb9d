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
// points).  Throughput = resident streams x chain speed, so the kernel is
// built for (a) a short dependent chain, (b) few instructions (a lone wave issues one every
// ~4-5 cycles, tools/ubench_chain.hip), (c) a small LDS footprint (streams per CU):
//   * the table entry is the 16-bit `nextState` of the table construction
//     (fsedecompressu16.go:233-241: newState = nextState << nbBits - tableSize,
//     nbBits = tableLog - highBits(nextState)).  With states kept in [size, 2*size) the update
//     is   state' = {nextState : window} >> (32 - nbBits),   nbBits = clz(nextState) - (31 - tableLog)
//     = v_ffbh, v_sub, v_alignbit: three dependent VALU ops between two LDS reads.  16 KiB at
//     tableLog 13 -> 9 streams per CU;
//   * no scalar bit window: the compressed stream sits in a 256-dword LDS ring and the 32-bit
//     window of a pair is a 2-dword LDS read at the running bit position, funnel-shifted
//     (v_alignbit again).  That read is issued together with the table look-ups of the pair, so
//     a pair costs one LDS round trip; no readfirstlane, no SALU chain, no branches in a chunk;
//   * every lane computes the same values (the LDS reads are broadcasts); the 64 lanes differ
//     only when they refill the ring (one 64-dword block per 128 symbols, prefetched a chunk
//     ahead) and when they translate 128 staged states to symbols through the L2-resident
//     symbol table, 256 bytes stored coalesced, one chunk behind the chain.
// LDS: ring[256 + 1 mirror] u32 | stage[64] u32 (128 states) | chain[2^tl] u16.
// Stream, symbol table and output are addressed as global (address_space(1)) pointers: a generic
// (flat) load also counts on lgkmcnt, and the chain's LDS waits would then wait for HBM too.
// ZB = table has 0-bit entries (zeroBits, fsedecompressu16.go:214-216).
// grid = units, block = 64, dynamic LDS = 2 << tl_hi + TD_EXTRA.
// TLHI names the table-size class of the launch (tableLog in (TLHI-1 .. TLHI], or <= 13): the classes
// differ only in dynamic LDS, but distinct instantiations give each its own line in a kernel trace.
#define TD_RING 256
#define TD_EXTRA ((TD_RING + 4 + 64) * 4)
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
    if (u.bits_off >= u.comp_len) { if (threadIdx.x == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t len = u.comp_len - u.bits_off;
    if (len >= (1u << 27)) return;                                      // bit positions are 32-bit here; the serial kernel takes it
    const uint32_t lane = threadIdx.x;
    const uint32_t size = 1u << tl;
    uint32_t *ring = s_mem;
    uint32_t *stage = ring + TD_RING + 4;
    uint16_t *chain = (uint16_t *)(stage + 64);
    {
        const uint32_t *dt = u.tt_nb;
        for (uint32_t p = lane * 2; p < size; p += 128) {
            const uint2 e = *(const uint2 *)(dt + p);                   // newState | nbBits << 16
            const uint32_t n0 = ((e.x & 0xFFFF) + size) >> (e.x >> 16);
            const uint32_t n1 = ((e.y & 0xFFFF) + size) >> (e.y >> 16);
            *(uint32_t *)(chain + p) = n0 | (n1 << 16);
        }
    }
    typedef const __attribute__((address_space(1))) uint16_t *gcu16;
    typedef const __attribute__((address_space(1))) uint32_t *gcu32;
    typedef __attribute__((address_space(1))) uint16_t *gu16;
    typedef __attribute__((address_space(1))) uint32_t *gu32;
    const gcu16 symg = (gcu16)(u.tab_sym - (TLHI == 16 ? 0u : size));   // indexed by the staged state (tableLog <= 15: in [size, 2*size))
    const uint32_t count = u.count;
    const gu16 out = (gu16)u.tok;
    const uint8_t *bs = u.comp_in + u.bits_off;
    const uint32_t last = bs[len - 1];
    if (last == 0) { if (lane == 0) u.status = MICD_ERR_CORRUPT; return; }  // bitreader.go:36-38
    // 4-byte aligned dword grid under the stream: grid bit 0 = LSB of g[0]; unread bits = grid bits [8*sb, cur)
    const uintptr_t addr = (uintptr_t)bs;
    const uint32_t sb = (uint32_t)(addr & 3);
    const gcu32 g = (gcu32)(addr - sb);
    const int32_t cur0 = (int32_t)(8u * (len - 1) + (uint32_t)(31 - __clz(last)) + 8u * sb);
    const int32_t top_dw = (cur0 - 1) >> 5;                             // dword holding the top unread bit
    auto load_blk = [&](int32_t b) -> uint32_t {                        // 64 dwords of block b, zero outside the stream
        const int32_t idx = b * 64 + (int32_t)lane;
        return (b >= 0 && idx <= top_dw) ? __builtin_nontemporal_load(g + idx) : 0u;   // streamed once: keep L2 for the symbol tables
    };
    auto store_blk = [&](int32_t b, uint32_t v) {
        const uint32_t slot = ((uint32_t)b & 3u) * 64u + lane;
        ring[slot] = v;
        if (slot == 0) ring[TD_RING] = v;                               // mirror: a 2-dword read at slot 255 stays linear
    };
    // window of a pair = grid bits [q, q+32): its MSB is the next unread bit
    int32_t q = cur0 - 32;
    int32_t blk = (q >> 5) >> 6;                                        // block of the window's low dword at chunk start
    store_blk(blk + 1, load_blk(blk + 1));
    store_blk(blk, load_blk(blk));
    store_blk(blk - 1, load_blk(blk - 1));
    uint32_t pf = load_blk(blk - 2);                                    // enters the ring at the end of the chunk
    __syncthreads();
    // The ring sits at LDS address 0 (dynamic LDS of a kernel without static LDS; checked below), so the
    // byte address of its dword is a shift and a mask of q with no base to add.
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)ring != 0u) { if (lane == 0) u.status = MICD_ERR_INTERNAL; return; }
    auto window = [&](int32_t qq) -> uint32_t {
        typedef const __attribute__((address_space(3))) uint32_t *lcu32;
        const lcu32 w = (lcu32)(uintptr_t)(((uint32_t)qq >> 3) & ((TD_RING - 1) * 4u));
        return __builtin_amdgcn_alignbit(w[1], w[0], (uint32_t)qq);    // shift = qq & 31
    };
    const uint32_t C = 31u - tl;
    const uint16_t *chain_o = chain - size;                             // states carry the +size offset
    uint32_t st[N];
    // initial states: state 0 first, tl bits each (fse2state.go:210-212); kept with the +size offset
#pragma unroll
    for (int p = 0; p < N; p++) {
        st[p] = size + (window(q) >> (32u - tl));
        q -= (int32_t)tl;
    }
    // one group = N symbols, states 0..N-1 in order; a pair takes <= 30 bits off one 32-bit window
    auto group = [&](uint32_t *stage_w) {
        uint32_t e[N];
#pragma unroll
        for (int k = 0; k < N; k++) e[k] = chain_o[st[k]];
#pragma unroll
        for (int p = 0; p < N; p += 2) {
            const uint32_t hi = window(q);
            stage_w[p >> 1] = (TLHI == 16 ? (st[p] & 0xFFFFu) : st[p]) | (st[p + 1] << 16);   // tableLog 16: low half = state - size
            // m = -nbBits (nextState >= 1, so clz is defined); a funnel shift right by m mod 32 = 32 - nbBits
            // both appends the bits to nextState and moves the window on to the second state
            const uint32_t m0 = C - (uint32_t)__builtin_clz(e[p]), m1 = C - (uint32_t)__builtin_clz(e[p + 1]);
            if (ZB) {                                                   // nbBits may be 0: 64-bit shift by 32 is well defined
                const uint32_t hi1 = hi << (0u - m0);
                st[p] = (uint32_t)((((uint64_t)e[p] << 32) | hi) >> (32u + m0));
                st[p + 1] = (uint32_t)((((uint64_t)e[p + 1] << 32) | hi1) >> (32u + m1));
            } else {
                const uint32_t hi1 = __builtin_amdgcn_alignbit(hi, 0u, m0);   // hi << nbBits0
                st[p] = __builtin_amdgcn_alignbit(e[p], hi, m0);
                st[p + 1] = __builtin_amdgcn_alignbit(e[p + 1], hi1, m1);
            }
            q += (int32_t)m0 + (int32_t)m1;
        }
    };
    // one symbol with state k (tail: fse2state.go:293-305 and siblings)
    auto single = [&](int k, uint16_t *stage_h) {
        const uint32_t e = chain_o[st[k]];
        const uint32_t hi = window(q);
        *stage_h = (uint16_t)st[k];
        const uint32_t nb = (uint32_t)__builtin_clz(e) - C;
        st[k] = (uint32_t)((((uint64_t)e << 32) | hi) >> (32u - nb));
        q -= (int32_t)nb;
    };
    // ring upkeep between chunks: a chunk of 128 symbols moves q down by at most 128 * 15 bits = 60 dwords,
    // so with blocks blk+1, blk, blk-1 present at its start every window read of the chunk is served; the
    // block below was fetched during the chunk and goes into the free slot now
    // (order at a chunk end: consume the old prefetch, store the old gather, then issue the new loads,
    // so every wait on a memory counter is for a load that has had a whole chunk to come back)
    // Header walker (frames only): the RLE headers of the token stream form a linked list
    // (rledecompressu16.go:59-85); following it here, on tokens that are still in registers, costs a
    // compare per chunk where runs are long, and spares k_dec_pixels_wg a chain of dependent HBM reads.
    // Same stop and error rules as that kernel's own walk, which still runs when this one gives up.
    bool w_on = u.mode == 0 && u.seg != nullptr;
    bool w_err = false;
    uint32_t w_pos = 0, w_out = 0, w_nseg = 0, w_mid = 0;
    const uint32_t w_symcap = min(u.sym_cap, 2u * (uint32_t)u.w * (uint32_t)u.h + 2u), w_segcap = u.seg_cap;
    typedef uint32_t w_v2 __attribute__((ext_vector_type(2)));
    __attribute__((address_space(1))) w_v2 *const w_seg = (__attribute__((address_space(1))) w_v2 *)u.seg;
    auto walk = [&](uint32_t cend, auto get) {                          // headers in front of token cend
        while (w_on && w_pos < cend) {
            const uint32_t h = get(w_pos);
            if (w_pos == 0) {                                           // token 0 fixes the run/literal split
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
    constexpr uint32_t G = 128 / N;                                     // groups per 128-symbol chunk
    const uint32_t chunks = count / 128;
    uint32_t pend_lo = 0, pend_hi = 0; bool have_pend = false;          // symbols gathered for the previous chunk (joined only
                                                                        // at the store: joining earlier would wait for the gather)
    uint32_t obase = 0;                                                 // dword index of the pending chunk in out
    for (uint32_t ch = 0; ch < chunks; ch++) {
        if (TLHI == 16) {                                               // 16 bits a symbol: refresh the ring mid-chunk as well
#pragma unroll 8
            for (uint32_t gi = 0; gi < G / 2; gi++) group(stage + gi * (N / 2));
            store_blk(blk - 2, pf);
            blk = (q >> 5) >> 6;
            pf = load_blk(blk - 2);
#pragma unroll 8
            for (uint32_t gi = G / 2; gi < G; gi++) group(stage + gi * (N / 2));
        } else {
#pragma unroll 8
            for (uint32_t gi = 0; gi < G; gi++) group(stage + gi * (N / 2));
        }
        // this chunk's 128 states are staged: write out the previous chunk, gather this one
        store_blk(blk - 2, pf);
        if (have_pend) {
            const uint32_t pk = pend_lo | (pend_hi << 16);
            __builtin_nontemporal_store(pk, (gu32)out + obase + lane);
            const uint32_t cb = obase * 2;
            walk(cb + 128, [&](uint32_t pos) -> uint32_t {
                const uint32_t rel = pos - cb;
                const uint32_t pair = __builtin_amdgcn_readlane(pk, rel >> 1);
                return (rel & 1) ? (pair >> 16) : (pair & 0xFFFFu);
            });
        }
        blk = (q >> 5) >> 6;
        pf = load_blk(blk - 2);
        const uint32_t s2 = stage[lane];
        pend_lo = symg[s2 & 0xFFFF]; pend_hi = symg[s2 >> 16];
        have_pend = true; obase = ch * 64;
    }
    if (have_pend) {
        const uint32_t pk = pend_lo | (pend_hi << 16);
        ((gu32)out)[obase + lane] = pk;
        const uint32_t cb = obase * 2;
        walk(cb + 128, [&](uint32_t pos) -> uint32_t {
            const uint32_t rel = pos - cb;
            const uint32_t pair = __builtin_amdgcn_readlane(pk, rel >> 1);
            return (rel & 1) ? (pair >> 16) : (pair & 0xFFFFu);
        });
    }
    // tail: count % 128 symbols; whole groups, then the last partial group state by state
    {
        const uint32_t done = chunks * 128;
        const uint32_t rem = count - done;
        uint32_t k = 0;
        for (; k + N <= rem; k += N) group(stage + (k / 2));
        uint16_t *st16 = (uint16_t *)stage;
#pragma unroll
        for (int j = 0; j < N - 1; j++) if (k + (uint32_t)j < rem) single(j, st16 + k + j);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        uint32_t tv0 = 0, tv1 = 0;
        if (lane < rem) { tv0 = symg[st16[lane]]; out[done + lane] = (uint16_t)tv0; }
        if (lane + 64 < rem) { tv1 = symg[st16[lane + 64]]; out[done + lane + 64] = (uint16_t)tv1; }
        walk(count, [&](uint32_t pos) -> uint32_t {
            const uint32_t rel = pos - done;
            return (rel < 64) ? __builtin_amdgcn_readlane(tv0, rel) : __builtin_amdgcn_readlane(tv1, rel - 64);
        });
    }
    if (lane == 0) {
        // bits still unread = grid bits [8*sb, q+32)
        if (q + 32 - (int32_t)(8u * sb) < 0) u.status = MICD_ERR_CORRUPT;   // bitreader.go:113-120
        else {
            u.ntok = count;
            if (u.mode == 0 && u.seg != nullptr && !w_err) { u.nseg = w_nseg; u.nsym = min(w_out, w_symcap); u.walk_ok = 1; }
        }
    }
}

// ==========================================================================================
// The same decoder with the transition table left in HBM / L2 (u32 entries newState | nbBits << 16, as the
// tables kernel writes them): for tableLog 16 -- every 16-bit-depth frame, CT above all -- the table has
// 65536 entries and a 17-bit nextState, too much for LDS.  A chain step is then an L2 round trip instead
// of an LDS one, but a wave needs only the 1.3 KiB ring + stage, so a CU holds as many streams as it has
// wave slots and hides that latency with them.  64-symbol chunks (a chunk may take 64 x 16 bits off the
// ring).  Takes every N-state stream the LDS classes left (tableLog > 15); 1-state streams stay serial.
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

template <int N, bool ZB, int TLHI>
static void launch_tans_lds_class(MicUnit *d_units, int n, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)k_dec_tans_lds<N, ZB, TLHI>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_dec_tans_lds<N, ZB, TLHI>), dim3(n), dim3(64), (2u << TLHI) + TD_EXTRA, stream, d_units);
}
template <int N, bool ZB>
static void launch_tans_lds(MicUnit *d_units, int n, hipStream_t stream, MicTimer *t, const char *name13) {
    if (t) t->mark(name13);
    launch_tans_lds_class<N, ZB, 13>(d_units, n, stream);
    if (t) t->mark("k_dec_tans_lds<other classes>");
    launch_tans_lds_class<N, ZB, 14>(d_units, n, stream);
    launch_tans_lds_class<N, ZB, 15>(d_units, n, stream);
    // tableLog 16 (16-bit depths): nextState fits 16 bits exactly when the table has no 0-bit entries, and the
    // 128 KiB table then takes a CU's LDS for one stream; streams with 0-bit entries go to k_dec_tans_gl
    if (!ZB) launch_tans_lds_class<N, false, 16>(d_units, n, stream);
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
    if (variant != 100) {
        if (t) t->mark("k_dec_tans_gl<2>");
        hipLaunchKernelGGL(k_dec_tans_gl<2>, dim3(n), dim3(64), 0, stream, d_units);
        if (t) t->mark("k_dec_tans_gl<4,8>");
        hipLaunchKernelGGL(k_dec_tans_gl<4>, dim3(n), dim3(64), 0, stream, d_units);
        hipLaunchKernelGGL(k_dec_tans_gl<8>, dim3(n), dim3(64), 0, stream, d_units);
    }
    if (t) t->mark("k_dec_tans_serial");
    hipLaunchKernelGGL(k_dec_tans_serial, dim3(n), dim3(64), 0, stream, d_units);
    if (variant == 100) {
        if (t) t->mark("k_dec_pixels_serial");
        hipLaunchKernelGGL(k_dec_pixels_serial, dim3(n), dim3(64), 0, stream, d_units);
    } else {
        mic_launch_decode_pixels(d_units, n, stream, t);
    }
    if (t) t->mark("end");
}
