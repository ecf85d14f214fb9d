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
                if (u.pred) prev = mic_grad_predict_at(out, w, x, y);
                else {
                    if (x > 0) { prev = out[idx - 1]; div++; }
                    if (y > 0) { prev += out[idx - w]; div++; }
                    if (div == 2) prev >>= 1;
                }
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
    // stage layout: stage16[k * G + g] = state k of group g (one 16-bit store per state: the LDS port has slack, the
    // VALU does not, and stores 2 * G bytes apart cannot be re-packed into a v_perm + 32-bit store)
    auto group = [&](uint16_t *stage_g) {
        uint32_t e[N];
#pragma unroll
        for (int k = 0; k < N; k++) e[k] = chain_o[st[k]];
#pragma unroll
        for (int p = 0; p < N; p += 2) {
            const uint32_t hi = window(q);
            // two 16-bit stores (the LDS port has slack, the VALU does not: no pack instruction); tableLog 16: the
            // low half of a state is state - size
            stage_g[p * (128 / N)] = (uint16_t)st[p];                   // tableLog 16: the low half of a state is state - size
            stage_g[(p + 1) * (128 / N)] = (uint16_t)st[p + 1];
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
            if (N == 2 && !ZB) {
                // issue order of a pair: the four address ops, the three LDS reads the chain waits for, then the two
                // stage stores in the shadow of that wait, then the nine ops that turn the entries into the next states
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
            }
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
#ifdef MIC_STAMP
    const uint64_t ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t vzero;                                                     // a zero the compiler cannot see through: keeps the stage
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));                      // address in a VGPR (else: one v_mov from an SGPR per store)
    uint16_t *const stage_v = (uint16_t *)stage + vzero;
    uint32_t pend_lo = 0, pend_hi = 0; bool have_pend = false;          // symbols gathered for the previous chunk (joined only
                                                                        // at the store: joining earlier would wait for the gather)
    uint32_t obase = 0;                                                 // dword index of the pending chunk in out
    for (uint32_t ch = 0; ch < chunks; ch++) {
        // eight groups per loop body, their stage slots at immediate offsets from one VGPR base
        auto groups = [&](uint32_t g0, uint32_t g1) {
            for (uint32_t gi = g0; gi < g1; gi += 8) {
                uint16_t *const sg = stage_v + gi;
#pragma unroll
                for (int j = 0; j < 8; j++) group(sg + j);
            }
        };
        static_assert(G % 16 == 0, "chunk halves are whole loop bodies");
        if (TLHI == 16) {                                               // 16 bits a symbol: refresh the ring mid-chunk as well
            groups(0, G / 2);
            store_blk(blk - 2, pf);
            blk = (q >> 5) >> 6;
            pf = load_blk(blk - 2);
            groups(G / 2, G);
        } else groups(0, G);
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
        {   // lane l translates tokens 2l and 2l+1 = states (2l % N), (2l % N) + 1 of group 2l / N
            const uint16_t *sg = (const uint16_t *)stage + ((2 * lane) % N) * G + (2 * lane) / N;
            pend_lo = symg[sg[0]]; pend_hi = symg[sg[G]];
        }
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
        uint16_t *st16 = (uint16_t *)stage;
        for (; k + N <= rem; k += N) group(st16 + k / N);
#pragma unroll
        for (int j = 0; j < N - 1; j++) if (k + (uint32_t)j < rem) single(j, st16 + j * G + k / N);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        auto staged = [&](uint32_t t) -> uint32_t { return st16[(t % N) * G + t / N]; };   // token t of the tail
        uint32_t tv0 = 0, tv1 = 0;
        if (lane < rem) { tv0 = symg[staged(lane)]; out[done + lane] = (uint16_t)tv0; }
        if (lane + 64 < rem) { tv1 = symg[staged(lane + 64)]; out[done + lane + 64] = (uint16_t)tv1; }
        walk(count, [&](uint32_t pos) -> uint32_t {
            const uint32_t rel = pos - done;
            return (rel < 64) ? __builtin_amdgcn_readlane(tv0, rel) : __builtin_amdgcn_readlane(tv1, rel - 64);
        });
    }
#ifdef MIC_STAMP
    if (lane == 0) { u.dbg[8] = (uint32_t)(__builtin_amdgcn_s_memtime() - ck0); u.dbg[9] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt0); }
#endif
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
#define T2_LDS (T2_WAVES * T2_WAVE)
template <int N, bool ZB>
__global__ void __launch_bounds__(64 * T2_WAVES) k_dec_tans_duo(MicUnit *units, int n_units) {
    extern __shared__ uint32_t s_mem[];
    typedef __attribute__((address_space(3))) uint32_t *l32;
    typedef __attribute__((address_space(3))) uint16_t *l16;
    typedef const __attribute__((address_space(1))) uint16_t *gcu16;
    typedef const __attribute__((address_space(1))) uint32_t *gcu32;
    typedef __attribute__((address_space(1))) uint16_t *gu16;
    typedef uint32_t g_v2 __attribute__((ext_vector_type(2)));
    // the waves of a group share nothing: each owns T2_WAVE bytes of LDS and never meets the others at a barrier
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
#ifdef MIC_STAMP
    const uint64_t ck_entry = __builtin_amdgcn_s_memtime();
#endif
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)s_mem != 0u) return;   // layout below assumes dynamic LDS at 0
    const int ui = ((int)blockIdx.x * T2_WAVES + (int)wv) * 2 + (int)half;
    bool mine = ui < n_units;
    MicUnit &u = units[mine ? ui : 0];
    uint32_t tl = 13, count = 0, len = 4, bits_off = 0;
    if (mine) {
        const uint32_t flav = u.flavour;
        tl = u.table_log; count = u.count; bits_off = u.bits_off;
        mine = u.status == MICD_OK && flav != 1 && ((flav == 108) ? 8u : flav) == (uint32_t)N && tl >= 5 && tl <= 13 &&
               ((u.zero_bits != 0) == ZB) && u.ntok == 0;
        if (mine && bits_off >= u.comp_len) { if (hl == 0) u.status = MICD_ERR_CORRUPT; mine = false; }
        if (mine) { len = u.comp_len - bits_off; if (len >= (1u << 27)) mine = false; }   // 32-bit bit positions here; serial kernel takes it
        if (!mine) { tl = 13; count = 0; len = 4; }
    }
    if (!__any(mine)) return;
    const uint32_t size = 1u << tl;
    const uint32_t wbase = wv * T2_WAVE;
    const uint32_t ringb = wbase + (half << 11), stageb = wbase + 1056u + (half << 11), tabb = wbase + T2_TAB0 + (half << 14);
    // ---- tables: u16 nextState = (newState + size) >> nbBits, 32 lanes per stream ----
    if (mine) {
        const uint32_t *dt = u.tt_nb;
        for (uint32_t p = hl * 2; p < size; p += 64) {
            const uint2 e = *(const uint2 *)(dt + p);
            const uint32_t n0 = ((e.x & 0xFFFF) + size) >> (e.x >> 16), n1 = ((e.y & 0xFFFF) + size) >> (e.y >> 16);
            *(l32)(uintptr_t)(tabb + p * 2) = n0 | (n1 << 16);
        }
    }
    const gcu16 symg = (gcu16)(u.tab_sym - size);
    const gu16 out = (gu16)u.tok;
    const uint8_t *bs = u.comp_in + bits_off;
    uint32_t last = mine ? (uint32_t)bs[len - 1] : 1u;
    if (mine && last == 0) { if (hl == 0) u.status = MICD_ERR_CORRUPT; mine = false; count = 0; last = 1; }   // bitreader.go:36-38
    const uintptr_t addr = (uintptr_t)bs;
    const uint32_t sb = (uint32_t)(addr & 3);
    const gcu32 g = (gcu32)(addr - sb);
    const int32_t cur0 = (int32_t)(8u * (len - 1) + (uint32_t)(31 - __clz(last)) + 8u * sb);
    const int32_t top_dw = mine ? (cur0 - 1) >> 5 : -1;
    // block b of a stream = its grid dwords [64b, 64b+64): two per lane
    auto load_blk = [&](int32_t b, uint32_t &v0, uint32_t &v1) {
        const int32_t i0 = b * 64 + (int32_t)hl, i1 = i0 + 32;
        v0 = (b >= 0 && i0 <= top_dw) ? __builtin_nontemporal_load(g + i0) : 0u;
        v1 = (b >= 0 && i1 <= top_dw) ? __builtin_nontemporal_load(g + i1) : 0u;
    };
    auto store_blk = [&](int32_t b, uint32_t v0, uint32_t v1) {
        const uint32_t slot = ((uint32_t)b & 3u) * 64u + hl;
        *(l32)(uintptr_t)(ringb + slot * 4) = v0;
        *(l32)(uintptr_t)(ringb + (slot + 32) * 4) = v1;
        if (slot == 0) *(l32)(uintptr_t)(ringb + 1024) = v0;              // mirror: a 2-dword read at slot 255 stays linear
    };
    int32_t q = cur0 - 32;
    int32_t blk = (q >> 5) >> 6;
    { uint32_t a0, a1; load_blk(blk + 1, a0, a1); store_blk(blk + 1, a0, a1); load_blk(blk, a0, a1); store_blk(blk, a0, a1);
      load_blk(blk - 1, a0, a1); store_blk(blk - 1, a0, a1); }
    uint32_t pf0, pf1; load_blk(blk - 2, pf0, pf1);
    __builtin_amdgcn_s_waitcnt(0xC07F);                                   // wave-private LDS: the writes above are in before the reads below
    auto window = [&](int32_t qq) -> uint32_t {
        const uint32_t a = (((uint32_t)qq >> 3) & 0x3FCu) | ringb;
        const uint32_t w0 = *(l32)(uintptr_t)a, w1 = *(l32)(uintptr_t)(a + 4);
        return __builtin_amdgcn_alignbit(w1, w0, (uint32_t)qq);
    };
    const uint32_t C = 31u - tl;
    const uint32_t cb = tabb - 2u * size;                                   // byte address of entry s = 2*s + cb, s in [size, 2*size)
    auto entry = [&](uint32_t st) -> uint32_t { return *(l16)(uintptr_t)(st * 2 + cb); };
    uint32_t st[N];
#pragma unroll
    for (int p = 0; p < N; p++) { st[p] = size + (window(q) >> (32u - tl)); q -= (int32_t)tl; }
    constexpr uint32_t G = 128 / N;
    // stage16[k * G + g] = state k of group g
    auto group = [&](l16 sg) {                                              // sg -> stage slot (0, g); pointer, so that the
        uint32_t e[N];                                                      // slots of a loop body become immediate offsets
#pragma unroll
        for (int k = 0; k < N; k++) e[k] = entry(st[k]);
#pragma unroll
        for (int p = 0; p < N; p += 2) {
            const uint32_t hi = window(q);
            sg[p * G] = (uint16_t)st[p];
            sg[(p + 1) * G] = (uint16_t)st[p + 1];
            const uint32_t m0 = C - (uint32_t)__builtin_clz(e[p]), m1 = C - (uint32_t)__builtin_clz(e[p + 1]);
            if (ZB) {
                const uint32_t hi1 = hi << (0u - m0);
                st[p] = (uint32_t)((((uint64_t)e[p] << 32) | hi) >> (32u + m0));
                st[p + 1] = (uint32_t)((((uint64_t)e[p + 1] << 32) | hi1) >> (32u + m1));
            } else {
                const uint32_t hi1 = __builtin_amdgcn_alignbit(hi, 0u, m0);
                st[p] = __builtin_amdgcn_alignbit(e[p], hi, m0);
                st[p + 1] = __builtin_amdgcn_alignbit(e[p + 1], hi1, m1);
            }
            q += (int32_t)m0 + (int32_t)m1;
            if (N == 2 && !ZB) {
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
            }
        }
    };
    const uint32_t chunks = count / 128, rem = count - chunks * 128;
    const uint32_t maxch = max((uint32_t)__builtin_amdgcn_readlane(chunks, 0), (uint32_t)__builtin_amdgcn_readlane(chunks, 32));
    // per-half header walkers (uniform values, one set per half)
    bool w_on[2], w_err[2]; uint32_t w_pos[2], w_out[2], w_nseg[2], w_mid[2], w_symcap[2], w_segcap[2], w_cnt[2];
    __attribute__((address_space(1))) g_v2 *w_seg[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int ln = h * 32;
        const uint32_t on = (mine && u.mode == 0 && u.seg != nullptr) ? 1u : 0u;
        w_on[h] = __builtin_amdgcn_readlane(on, ln) != 0; w_err[h] = false;
        w_pos[h] = 0; w_out[h] = 0; w_nseg[h] = 0; w_mid[h] = 0;
        const uint32_t sc = min(u.sym_cap, 2u * (uint32_t)u.w * (uint32_t)u.h + 2u);
        w_symcap[h] = __builtin_amdgcn_readlane(sc, ln); w_segcap[h] = __builtin_amdgcn_readlane(u.seg_cap, ln);
        w_cnt[h] = __builtin_amdgcn_readlane(count, ln);
        const uint64_t sp = (uint64_t)(uintptr_t)u.seg;
        const uint64_t spu = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(sp >> 32), ln) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)sp, ln);
        w_seg[h] = (__attribute__((address_space(1))) g_v2 *)(uintptr_t)spu;
    }
    auto walk = [&](int h, uint32_t cend, auto get) {                       // as in k_dec_tans_lds
        while (w_on[h] && w_pos[h] < cend) {
            const uint32_t hd = get(w_pos[h]);
            if (w_pos[h] == 0) {
                const int d0 = mic_len16((uint16_t)hd);
                if (d0 == 0) { w_on[h] = false; w_err[h] = true; break; }
                w_mid[h] = (1u << (d0 - 1)) - 1; w_pos[h] = 1;
                continue;
            }
            if (w_out[h] >= w_symcap[h]) { w_on[h] = false; break; }
            if (hd == 0 || w_nseg[h] >= w_segcap[h]) { w_on[h] = false; w_err[h] = true; break; }
            if (hd <= w_mid[h]) {
                if (w_pos[h] + 1 >= w_cnt[h]) { w_on[h] = false; w_err[h] = true; break; }
                if (lane == 0) { g_v2 r; r.x = (w_pos[h] + 1) | 0x80000000u; r.y = w_out[h]; w_seg[h][w_nseg[h]] = r; }
                w_nseg[h]++; w_out[h] += hd; w_pos[h] += 2;
            } else {
                if (lane == 0) { g_v2 r; r.x = w_pos[h] + 1; r.y = w_out[h]; w_seg[h][w_nseg[h]] = r; }
                w_nseg[h]++; w_out[h] += hd - w_mid[h]; w_pos[h] += 1 + (hd - w_mid[h]);
            }
        }
    };
    // lane j of a half translates tokens 4j .. 4j+3 of a chunk: pend0 = tokens 4j, 4j+1; pend1 = 4j+2, 4j+3
    uint32_t pend[4] = { 0u, 0u, 0u, 0u }; bool have_pend = false; uint32_t pch = 0;
    auto flush_pend = [&]() {
        if (!have_pend) return;
        const uint32_t p0 = pend[0] | (pend[1] << 16), p1 = pend[2] | (pend[3] << 16);
        if (mine && pch < chunks) { g_v2 v; v.x = p0; v.y = p1; __builtin_nontemporal_store(v, (__attribute__((address_space(1))) g_v2 *)(out + pch * 128) + hl); }
        const uint32_t cbase = pch * 128;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (pch < (uint32_t)__builtin_amdgcn_readlane(chunks, h * 32))
                walk(h, cbase + 128, [&](uint32_t pos) -> uint32_t {
                    const uint32_t rel = pos - cbase;
                    const uint32_t d = (rel & 2) ? __builtin_amdgcn_readlane(p1, h * 32 + (rel >> 2)) : __builtin_amdgcn_readlane(p0, h * 32 + (rel >> 2));
                    return (rel & 1) ? (d >> 16) : (d & 0xFFFFu);
                });
        }
    };
#ifdef MIC_STAMP
    const uint64_t ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // the shorter half's true end-of-chunks state
    uint32_t sv_st[N]; int32_t sv_q = q;
#pragma unroll
    for (int k = 0; k < N; k++) sv_st[k] = st[k];
    for (uint32_t ch = 0; ch < maxch; ch++) {
        if (ch == chunks) {                                                  // (per lane) this half is done: keep its state aside
            sv_q = q;
#pragma unroll
            for (int k = 0; k < N; k++) sv_st[k] = st[k];
        }
        constexpr uint32_t UG = G;                                             // groups per loop body (a taken scalar branch costs tens of cycles)
        for (uint32_t gi = 0; gi < G; gi += UG) {
            const l16 sg = (l16)(uintptr_t)(stageb + gi * 2);
#pragma unroll
            for (uint32_t j = 0; j < UG; j++) group(sg + j);
        }
        store_blk(blk - 2, pf0, pf1);
        flush_pend();
        blk = (q >> 5) >> 6;
        load_blk(blk - 2, pf0, pf1);
        {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t tok = 4 * hl + t;
                const uint32_t sv = *(l16)(uintptr_t)(stageb + ((tok % N) * G + tok / N) * 2);
                pend[t] = mine ? (uint32_t)symg[sv] : 0u;                   // (a foreign half decodes garbage: keep it off the symbol table)
            }
        }
        have_pend = true; pch = ch;
    }
    flush_pend();
    if (chunks < maxch) {                                                    // (per lane) restore the true state of the shorter half
        q = sv_q;
#pragma unroll
        for (int k = 0; k < N; k++) st[k] = sv_st[k];
    }
    if (maxch) {   // ... and put its ring back where that state reads (the run-off moved it on); harmless for the other half
        blk = (q >> 5) >> 6;
        uint32_t a0, a1;
        load_blk(blk + 1, a0, a1); store_blk(blk + 1, a0, a1);
        load_blk(blk, a0, a1); store_blk(blk, a0, a1);
        load_blk(blk - 1, a0, a1); store_blk(blk - 1, a0, a1);
        __builtin_amdgcn_s_waitcnt(0xC07F);                                   // wave-private LDS: the writes above are in before the reads below
    }
    // ---- tails: rem < 128 tokens per half, predicated per lane ----
    {
        const uint32_t done = chunks * 128;
        for (uint32_t k = 0; k < 128; k += N) {
            if (!__any(k < rem)) break;
            const bool whole = k + N <= rem;
            uint32_t st_b[N]; const int32_t q_b = q;
#pragma unroll
            for (int i = 0; i < N; i++) st_b[i] = st[i];
            if (__any(whole)) group((l16)(uintptr_t)(stageb + (k / N) * 2));   // all N states of group k / N
            if (!whole) {                                                    // this half: the last partial group, state by state, or nothing
                q = q_b;
#pragma unroll
                for (int i = 0; i < N; i++) st[i] = st_b[i];
#pragma unroll
                for (int i = 0; i < N - 1; i++) {
                    if (k + (uint32_t)i < rem) {
                        const uint32_t e = entry(st[i]);
                        const uint32_t hi = window(q);
                        *(l16)(uintptr_t)(stageb + (i * G + k / N) * 2) = (uint16_t)st[i];
                        const uint32_t nb = (uint32_t)__builtin_clz(e) - C;
                        st[i] = (uint32_t)((((uint64_t)e << 32) | hi) >> (32u - nb));
                        q -= (int32_t)nb;
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        uint32_t tv[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t tok = 4 * hl + t;
            if (mine && tok < rem) {
                tv[t] = symg[*(l16)(uintptr_t)(stageb + ((tok % N) * G + tok / N) * 2)];
                out[done + tok] = (uint16_t)tv[t];
            }
        }
        const uint32_t p0 = tv[0] | (tv[1] << 16), p1 = tv[2] | (tv[3] << 16);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t dn = (uint32_t)__builtin_amdgcn_readlane(done, h * 32);
            walk(h, w_cnt[h], [&](uint32_t pos) -> uint32_t {
                const uint32_t rel = pos - dn;
                const uint32_t d = (rel & 2) ? __builtin_amdgcn_readlane(p1, h * 32 + (rel >> 2)) : __builtin_amdgcn_readlane(p0, h * 32 + (rel >> 2));
                return (rel & 1) ? (d >> 16) : (d & 0xFFFFu);
            });
        }
    }
#ifdef MIC_STAMP
    if (hl == 0 && mine) { u.dbg[8] = (uint32_t)(__builtin_amdgcn_s_memtime() - ck0); u.dbg[9] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt0); u.dbg[10] = maxch; u.dbg[11] = (uint32_t)(ck0 - ck_entry); }
#endif
    if (hl == 0 && mine) {
        if (q + 32 - (int32_t)(8u * sb) < 0) u.status = MICD_ERR_CORRUPT;   // bitreader.go:113-120
        else {
            u.ntok = count;
            const bool werr = half ? w_err[1] : w_err[0];
            if (u.mode == 0 && u.seg != nullptr && !werr) {
                u.nseg = half ? w_nseg[1] : w_nseg[0];
                u.nsym = min(half ? w_out[1] : w_out[0], half ? w_symcap[1] : w_symcap[0]);
                u.walk_ok = 1;
            }
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
    static MicPerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute((const void *)k_dec_tans_lds<N, ZB, TLHI>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024); });
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
    // One such stream fills a CU (7.7 ms per 256 CT frames, 30 us a frame); the L2-table kernel is ~4x slower per
    // stream but holds 32 of them per CU, so very large batches are left to it.
    if (!ZB && n <= 1280) launch_tans_lds_class<N, false, 16>(d_units, n, stream);
}

void mic_launch_decode(MicUnit *d_units, int n, hipStream_t stream, int variant, MicTimer *t, int *d_cls) {
    const bool any_grad = (variant & MIC_VARIANT_GRAD) != 0;
    variant &= ~MIC_VARIANT_GRAD;
    if (variant == 100) {
        if (t) t->mark("k_dec_tables");
        hipLaunchKernelGGL(k_dec_tables, dim3(n), dim3(256), 0, stream, d_units);
    } else {
        if (t) t->mark("k_dec_tables_wg");
        mic_launch_dec_tables(d_units, n, stream);
    }
    if (variant == 0 && d_cls) {
        // tableLog <= 13: lane-per-state kernels over compacted per-class lists (mic_decode_ls.hip); what they leave
        // (bigger tables, 1-state streams, very long streams) falls through to the classes below, which skip decoded units
        mic_launch_dec_tans_ls(d_units, n, d_cls + MIC_CLS_HEAD, d_cls, stream, t);
    } else if (variant != 100) {
        static MicPerDeviceOnce once;
        once.run([] {
            (void)hipFuncSetAttribute((const void *)k_dec_tans_duo<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
            (void)hipFuncSetAttribute((const void *)k_dec_tans_duo<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
            (void)hipFuncSetAttribute((const void *)k_dec_tans_duo<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
            (void)hipFuncSetAttribute((const void *)k_dec_tans_duo<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
            (void)hipFuncSetAttribute((const void *)k_dec_tans_duo<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
            (void)hipFuncSetAttribute((const void *)k_dec_tans_duo<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
        });
        const unsigned nw = (unsigned)((n + 2 * T2_WAVES - 1) / (2 * T2_WAVES));
        if (t) t->mark("k_dec_tans_duo<2,false>");
        hipLaunchKernelGGL((k_dec_tans_duo<2, false>), dim3(nw), dim3(64 * T2_WAVES), T2_LDS, stream, d_units, n);
        if (t) t->mark("k_dec_tans_duo<other>");
        hipLaunchKernelGGL((k_dec_tans_duo<2, true>), dim3(nw), dim3(64 * T2_WAVES), T2_LDS, stream, d_units, n);
        hipLaunchKernelGGL((k_dec_tans_duo<4, false>), dim3(nw), dim3(64 * T2_WAVES), T2_LDS, stream, d_units, n);
        hipLaunchKernelGGL((k_dec_tans_duo<4, true>), dim3(nw), dim3(64 * T2_WAVES), T2_LDS, stream, d_units, n);
        hipLaunchKernelGGL((k_dec_tans_duo<8, false>), dim3(nw), dim3(64 * T2_WAVES), T2_LDS, stream, d_units, n);
        hipLaunchKernelGGL((k_dec_tans_duo<8, true>), dim3(nw), dim3(64 * T2_WAVES), T2_LDS, stream, d_units, n);
    }
    if (variant != 100 && !(variant == 0 && d_cls)) {                       // (the lane-per-state kernels cover every table size)
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
        mic_launch_decode_pixels(d_units, n, stream, t, any_grad);
    }
    if (t) t->mark("end");
}
