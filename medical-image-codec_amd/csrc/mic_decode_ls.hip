// mic_decode_ls.hip -- tANS decode with one LANE per (stream, state): nine streams per CU.
//
// The N states of an N-state stream (fse2state.go:203-308, fse4state.go:195-353, fse8state.go:230-380,
// rans8state.go:221-412) share one reverse bitstream, so a stream is a serial chain
//   state_k -> table entry -> (nbBits, nextState) -> bits at the running position -> state_k'
// and decode throughput = resident streams / latency of one round of N symbols.  Two things bound it on a
// CU: the LDS holds the 16-bit nextState tables of at most NINE streams (tableLog 13: 16 KiB each), and a wave issues one
// instruction per ~5 cycles whatever its lanes do.  The earlier kernel (k_dec_tans_duo, mic_decode.hip) ran every lane of a
// wave through the same values -- two distinct streams per 64 lanes, N table look-ups issued one after the other -- so it
// was bound by instructions issued per symbol and got slower with more states.  Here lane g*N + k of a wave owns state k of
// the wave's stream g (three streams per wave, three waves per group, one group per CU; N = 4: lane 16 g + k, a DPP row a stream):
//   * a round = ONE table look-up instruction for all N states of all three streams, nbBits from v_ffbh, the offset of a
//     state's bits inside the round from a DPP prefix sum over the N lanes of its stream (quad_perm / row_shr: no LDS, no
//     readlane), the bits from a funnel shift of the stream's bit window, kept in a 256-dword LDS ring;
//   * N = 2 and N = 4: the round's window (32 / 64 bits) is read at the round's start position together with the table look-ups,
//     so a round costs one LDS round trip (115 cycles for two symbols, 136 for four); N = 8: every lane reads its own window at
//     its own bit position once the prefix sum is known: two round trips for eight symbols.  More states decode FASTER;
//   * the lanes of a wave that own no state clone stream 0 (same addresses: LDS broadcasts); all 64 lanes share the
//     per-chunk work, which is small: ring refill (64 dwords per stream and 128 symbols, prefetched a chunk ahead) and one
//     coalesced 256-byte store of the chunk's 128 STATES per stream.
// The states are turned into symbols by k_dec_translate (below), not here: a symbol look-up is a 2-byte gather from a 16 KiB
// table per stream -- 9 tables x 32 CUs overflow an XCD's 4 MiB L2, every wave-instruction touches 64 cache lines, and with
// the gathers inside this kernel the chain waves spent 30 % of their time queueing on the CU's miss path (3200 of 11 000
// cycles per chunk; stamps: tools/time_dec.py on an LS_STAMP build).  k_dec_translate has the table in LDS, streams the
// states once, and walks the RLE headers (rledecompressu16.go:59-85) on tiles it already holds in LDS.
// Streams come from a compacted per-class list (k_dec_classify), so a launch only touches the units of its class.
// LDS per stream: ring 1024 B (2048 at tableLog 16) | 2 mirror dwords + pad, 16 B | stage 256 B (128 u16 states) | table 2 << tableLog B.
#include <type_traits>
#include "mic_dev.h"
#include "mic_launch.h"

// Table-size classes: streams per wave x waves per group are what the LDS holds of 2 << tableLog byte tables.
//   tableLog 13: 3 x 3 = nine streams (159 120 bytes) | 14: 2 x 2 (136 256) | 15: 1 x 2 (134 736) | 16: 1 x 1 (133 392)
//   tableLog <= 12 (8 KiB tables: small units -- MIC3 planes, thin strips): 4 x 4 = sixteen streams (151 808)
// At tableLog 16 a nextState needs 17 bits when the table has 0-bit entries (zeroBits): those streams stay with k_dec_tans_gl.
#define LS_RING 0u
#define LS_CLASSES MIC_CLS_CLASSES                 // 5 table-size classes (tableLog 13, 14, 15, 16, <= 12) x (N in 2,4,8) x (zeroBits)
template <int TL> struct LsGeom {
    static constexpr int SPW = TL <= 12 ? 4 : TL == 13 ? 3 : TL == 14 ? 2 : 1;   // streams per wave
    static constexpr int WAVES = TL <= 12 ? 4 : TL == 13 ? 3 : TL == 16 ? 1 : 2;  // waves per group (one group per CU: the LDS is full)
    // The bit window's ring: blocks of 64 dwords.  A chunk of 128 symbols takes at most 4 * tableLog dwords off it (+ 2 the window
    // reads reach below the position): under a block up to tableLog 15, so three blocks stored (the position's, one above, one below)
    // and a fourth in flight do; at tableLog 16 a chunk can take a whole block, so the ring is twice as long and runs a block deeper.
    static constexpr int RING_BLOCKS = TL == 16 ? 8 : 4;
    static constexpr int DEPTH = TL == 16 ? 3 : 2;                        // the block in flight is the position's block - DEPTH
    static constexpr int RING_BITS = TL == 16 ? 9 : 8;                    // dword index bits
    static constexpr uint32_t MIRROR = 256u * RING_BLOCKS;                // two mirror dwords (slots 0, 1) + two pad dwords
    static constexpr uint32_t STAGE = MIRROR + 16u;                       // 128 u16 states of a chunk
    static constexpr uint32_t TAB = STAGE + 256u;
    static constexpr uint32_t STREAM_BYTES = TAB + (2u << TL);
    static constexpr uint32_t LDS = WAVES * SPW * STREAM_BYTES;
    static_assert(LDS <= 160 * 1024, "one group must fit a CU's LDS");
};
#define LS_TL 13                                   // the nine-streams class

typedef __attribute__((address_space(3))) uint32_t *ls_l32;
typedef __attribute__((address_space(3))) uint16_t *ls_l16;
typedef const __attribute__((address_space(1))) uint16_t *ls_gcu16;
typedef const __attribute__((address_space(1))) uint32_t *ls_gcu32;
typedef __attribute__((address_space(1))) uint16_t *ls_gu16;
typedef __attribute__((address_space(1))) uint32_t *ls_gu32;
typedef uint32_t ls_v2 __attribute__((ext_vector_type(2)));

template <int CTRL> __device__ __forceinline__ uint32_t ls_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
#define LS_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define LS_ROW_SHR(n) (0x110 + (n))
#define LS_ROW_HALF_MIRROR 0x141

__device__ __forceinline__ uint32_t ls_rl(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ uint64_t ls_rl64(uint64_t v, int lane) {
    return ((uint64_t)ls_rl((uint32_t)(v >> 32), lane) << 32) | ls_rl((uint32_t)v, lane);
}

// Per-class lists of unit indices, in unit order: list[c * n + i], count[c].  Class = 2 * log2(N / 2) + zeroBits for the
// N-state streams (rANS-8 decodes as 8-state), + 6 per table-size class (tableLog <= 13, 14, 15, 16); 1-state streams, very long
// streams and tableLog-16 tables with 0-bit entries are left to the kernels of mic_decode.hip.
// One group; the stream checks that need the blob (empty bitstream, zero end byte: bitreader.go:33-38) are made here.
__global__ void __launch_bounds__(1024) k_dec_classify(MicUnit *units, int n, int *list, int *count) {
    __shared__ uint32_t s_w[16][LS_CLASSES];
    __shared__ uint32_t s_base[LS_CLASSES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < LS_CLASSES) s_base[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + (int)tid;
        int cls = -1;
        if (i < n) {
            MicUnit &u = units[i];
            const uint32_t flav = u.flavour, tl = u.table_log;
            const uint32_t ns = (flav == 108) ? 8u : flav;
            if (u.status == MICD_OK && u.ntok == 0 && (ns == 2 || ns == 4 || ns == 8) && tl >= MIC_MIN_TABLELOG && tl <= MIC_MAX_TABLELOG &&
                !(tl == 16 && u.zero_bits)) {                               // (17-bit nextState: k_dec_tans_gl)
                if (u.bits_off >= u.comp_len) u.status = MICD_ERR_CORRUPT;
                else if (u.comp_len - u.bits_off < (1u << 27)) {            // 32-bit bit positions here; the serial kernel takes longer ones
                    if (u.comp_in[u.comp_len - 1] == 0) u.status = MICD_ERR_CORRUPT;
                    else cls = mic_dec_cls(flav, tl, u.zero_bits);
                }
            }
        }
        uint32_t rank = 0;
#pragma unroll
        for (int c = 0; c < LS_CLASSES; c++) {
            const uint64_t m = __ballot(cls == c);
            if (cls == c) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_w[wave][c] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (cls >= 0) {
            uint32_t off = s_base[cls];
            for (uint32_t w = 0; w < wave; w++) off += s_w[w][cls];
            list[(size_t)cls * (size_t)n + off + rank] = i;
        }
        __syncthreads();
        if (tid < LS_CLASSES) { uint32_t t = 0; for (int w = 0; w < 16; w++) t += s_w[w][tid]; s_base[tid] += t; }
        __syncthreads();
    }
    if (tid < LS_CLASSES) count[tid] = (int)s_base[tid];
}

template <int N, bool ZB, int TL>
__global__ void __launch_bounds__(64 * LsGeom<TL>::WAVES) k_dec_tans_ls(MicUnit *units, const int *list, const int *count_p) {
    constexpr int LS_SPW = LsGeom<TL>::SPW, LS_WAVES = LsGeom<TL>::WAVES;
    constexpr uint32_t LS_STREAM_BYTES = LsGeom<TL>::STREAM_BYTES, LS_STAGE = LsGeom<TL>::STAGE, LS_TAB = LsGeom<TL>::TAB, LS_MIRROR = LsGeom<TL>::MIRROR;
    constexpr int LS_RB = LsGeom<TL>::RING_BLOCKS, LS_DEPTH = LsGeom<TL>::DEPTH, LS_RBITS = LsGeom<TL>::RING_BITS;
    extern __shared__ uint32_t s_mem[];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)s_mem != 0u) return;   // layout assumes dynamic LDS at 0
    const int n_cls = *count_p;
    const int slot0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * LS_WAVES + (int)wv) * LS_SPW);   // (wave-uniform, provably so: descriptors below)
    if (slot0 >= n_cls) return;                                             // waves share nothing and never meet at a barrier
    // ---- roles ----------------------------------------------------------------------------------------------------
    // N = 4 gives every stream a DPP row of 16 lanes (states in lanes 0-3, the other twelve are clones that sit out the chunks):
    // row_shr then stops at the stream's first lane by itself, and the prefix sum of a round is two instructions
    constexpr int LS_LS = N == 4 ? 16 : N;                                  // lanes from one stream to the next
    const uint32_t k = (lane % LS_LS) % N;                                  // state of the chain this lane runs
    const bool real = lane % LS_LS < N && lane < LS_SPW * LS_LS;            // (not a clone inside a stream's row, nor a surplus lane)
    uint32_t g = lane / LS_LS; if (g >= LS_SPW) g = 0;                      // its stream; surplus lanes clone stream 0
    const bool have = slot0 + (int)g < n_cls;                               // the last wave of a class may hold fewer streams
    const uint32_t gs = have ? g : 0;                                       // absent streams run on stream 0's table (every access in range)
    const MicUnit &u = units[list[slot0 + (int)gs]];
    const uint32_t tl = u.table_log, size = 1u << tl;
    const uint32_t count = have ? u.count : 0u;
    const uint32_t bits_off = u.bits_off, len = u.comp_len - bits_off;
    const uint32_t sbase = (wv * LS_SPW + g) * LS_STREAM_BYTES;             // own ring / stage (also for an absent stream)
    const uint32_t tbase = (wv * LS_SPW + gs) * LS_STREAM_BYTES + LS_TAB;
    // ---- tables: u16 nextState = (newState + size) >> nbBits (fsedecompressu16.go:233-241), all 64 lanes per stream ----
#pragma unroll 1
    for (int j = 0; j < LS_SPW; j++) {
        if (slot0 + j >= n_cls) break;
        const int src = j * LS_LS;                                          // a lane that holds stream j's values
        const uint32_t sz = ls_rl(size, src);
        const uint32_t *dt = (const uint32_t *)(uintptr_t)ls_rl64((uint64_t)(uintptr_t)u.tt_nb, src);
        const uint32_t tb = (wv * LS_SPW + (uint32_t)j) * LS_STREAM_BYTES + LS_TAB;
        for (uint32_t p = lane * 4; p < sz; p += 256) {
            const uint4 e = *(const uint4 *)(dt + p);
            const uint32_t n0 = ((e.x & 0xFFFF) + sz) >> (e.x >> 16), n1 = ((e.y & 0xFFFF) + sz) >> (e.y >> 16);
            const uint32_t n2 = ((e.z & 0xFFFF) + sz) >> (e.z >> 16), n3 = ((e.w & 0xFFFF) + sz) >> (e.w >> 16);
            ls_v2 v; v.x = n0 | (n1 << 16); v.y = n2 | (n3 << 16);
            *(__attribute__((address_space(3))) ls_v2 *)(uintptr_t)(tb + p * 2) = v;
        }
    }
    // ---- bit reader (bitreader.go): grid of aligned dwords under the stream; unread bits = grid bits [8*sb, cur) ----
    const uint8_t *bs = u.comp_in + bits_off;
    const uint32_t last = bs[len - 1];                                      // non-zero: k_dec_classify
    const uintptr_t addr = (uintptr_t)bs;
    const uint32_t sb = (uint32_t)(addr & 3);
    const uint64_t gaddr = (uint64_t)(addr - sb);
    const int32_t cur0 = (int32_t)(8u * (len - 1) + (uint32_t)(31 - __clz(last)) + 8u * sb);
    const int32_t top_dw = have ? (cur0 - 1) >> 5 : -1;
    int32_t q = cur0 - 32;                                                  // window of a round = grid bits [q, q+32): its MSB is the next unread bit
    // wave-uniform per-stream values for the shared work.  The compressed stream and the output go through buffer descriptors:
    // a block of the stream that lies (partly) outside it reads as zeros and a store behind a stream's last whole chunk is
    // dropped by the range check, so the per-chunk upkeep below has no branch at all (a taken scalar branch costs a wave
    // 30-45 cycles, and the compiler-predicated form of this upkeep had fifteen of them: 1700 cycles per chunk, stamped).
    bool s_have[LS_SPW]; uint64_t s_out[LS_SPW]; int32_t s_blk[LS_SPW], s_pfb[LS_SPW]; uint32_t s_chunks[LS_SPW];
    __amdgpu_buffer_rsrc_t rs_in[LS_SPW], rs_out[LS_SPW];
#pragma unroll
    for (int j = 0; j < LS_SPW; j++) {
        const int src = j * LS_LS;
        s_have[j] = slot0 + j < n_cls;
        s_out[j] = ls_rl64((uint64_t)(uintptr_t)u.tok, src);
        s_chunks[j] = ls_rl(count, src) / 128u;
        s_blk[j] = ((int32_t)ls_rl((uint32_t)(q - (int32_t)(N * tl)), src) >> 5) >> 6;   // block of the position BEHIND the initial states (N * tableLog
                                                                            // bits at most: a chunk then never moves the position by more than one block)
        const uint32_t in_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(s_have[j] ? ((uint32_t)((int32_t)ls_rl((uint32_t)top_dw, src) + 1)) * 4u : 0u));
        const uint32_t out_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(s_have[j] ? s_chunks[j] * 256u : 0u));
        rs_in[j] = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)ls_rl64(gaddr, src), 0, (int)in_bytes, 0x00020000);
        rs_out[j] = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)s_out[j], 0, (int)out_bytes, 0x00020000);
    }
    auto load_blk = [&](int j, int32_t b) -> uint32_t {                    // dword `lane` of block b of stream j, zero outside the stream
        return __builtin_amdgcn_raw_buffer_load_b32(rs_in[j], (b * 64 + (int32_t)lane) * 4, 0, 2);   // (a negative offset is out of range too); nt: streamed once
    };
    auto store_blk = [&](int j, int32_t b, uint32_t v) {
        const uint32_t rb = (wv * LS_SPW + (uint32_t)j) * LS_STREAM_BYTES + LS_RING;
        const uint32_t slot = (uint32_t)b & (uint32_t)(LS_RB - 1);
        *(ls_l32)(uintptr_t)(rb + (slot * 64u + lane) * 4) = v;
        // mirror: a read of two or three dwords from the ring's last slots stays linear.  Lanes 0 and 1 write it when the block is
        // the ring's first, the pad dwords behind it otherwise (an address select, not a branch)
        if (lane < 2) *(ls_l32)(uintptr_t)(rb + LS_MIRROR + lane * 4 + (slot == 0u ? 0u : 8u)) = v;
    };
    uint32_t pf[LS_SPW];
#pragma unroll
    for (int j = 0; j < LS_SPW; j++) {
#pragma unroll
        for (int dd = -1; dd < LS_DEPTH; dd++) store_blk(j, s_blk[j] - dd, load_blk(j, s_blk[j] - dd));
        s_pfb[j] = s_blk[j] - LS_DEPTH;
        pf[j] = load_blk(j, s_pfb[j]);                                      // enters the ring at the end of the first chunk
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                                     // wave-private LDS: the writes above are in before the reads below
    // ---- chain state ---------------------------------------------------------------------------------------------
    const uint32_t ringb = sbase + LS_RING;
    const uint32_t stgb = sbase + LS_STAGE + 2u * k;                        // stage16[r * N + k] = state k of round r
    const uint32_t cb = tbase - 2u * size;                                  // byte address of entry s = 2 * s + cb, s in [size, 2 * size)
    const uint32_t C = 31u - tl;
    auto window = [&](int32_t qq) -> uint32_t {
        const uint32_t a = (__builtin_amdgcn_ubfe((uint32_t)qq, 5, LS_RBITS) << 2) + ringb;
        const uint32_t w0 = *(ls_l32)(uintptr_t)a, w1 = *(ls_l32)(uintptr_t)(a + 4);
        return __builtin_amdgcn_alignbit(w1, w0, (uint32_t)qq);
    };
    auto entry = [&](uint32_t s) -> uint32_t { return *(ls_l16)(uintptr_t)(s * 2 + cb); };
    // initial states: state 0 first, tableLog bits each (fse2state.go:210-212); kept with the +size offset
    uint32_t st = size + (window(q - (int32_t)(k * tl)) >> (32u - tl));
    q -= (int32_t)(N * tl);
    const uint32_t kq = k & 3u;                                             // position inside the quad (N = 8: a stream spans two quads)
    const uint32_t mk1 = (kq >= 1) ? ~0u : 0u, mk2 = (kq >= 2) ? ~0u : 0u, mk4 = (k >= 4) ? ~0u : 0u;
    // bits of a round: nb = this state's nbBits, pre = bits the earlier states of the round take, ntot = MINUS the round's bits.
    // Minus, because `q - dpp(x)` must not reach the compiler: it folds the lane permutation into a v_subrev_u32_dpp, and on this
    // toolchain (ROCm 7.2, gfx950) that instruction computes dpp(src1) - src0 instead of src1 - dpp(src0) -- measured with
    // tools/dpp_subrev_repro.hip, v_sub_u32_dpp and v_add_u32_dpp are as documented -- so the sum is negated before it is permuted
    // and then ADDED to the bit position.
    auto prefix = [&](uint32_t nb, uint32_t &pre, uint32_t &ntot) {
        if (N == 2) {
            pre = ls_dpp<LS_QP(0, 0, 2, 2)>(nb) & mk1;
            ntot = 0u - (nb + ls_dpp<LS_QP(1, 0, 3, 2)>(nb));
        } else {
            uint32_t t = nb + (ls_dpp<LS_QP(0, 0, 1, 2)>(nb) & mk1);
            t = t + (ls_dpp<LS_QP(0, 0, 0, 1)>(t) & mk2);                    // inclusive prefix inside the quad
            const uint32_t ntq = ls_dpp<LS_QP(3, 3, 3, 3)>(0u - t);         // minus the quad's sum
            if (N == 4) { pre = t - nb; ntot = ntq; }
            else {
                t = t + ((0u - ls_dpp<LS_ROW_SHR(4)>(ntq)) & mk4);
                pre = t - nb;
                ntot = ntq + ls_dpp<LS_ROW_HALF_MIRROR>(ntq);
            }
        }
    };
    // one round = N symbols of every stream of the wave; stage slot at byte offset soff from stgb
    auto round = [&](uint32_t soff) {
        const uint32_t e = entry(st);
        if (N == 2) {
            const uint32_t hi = window(q);
            *(ls_l16)(uintptr_t)(stgb + soff) = (uint16_t)st;
            const uint32_t c = (uint32_t)__builtin_clz(e);
            const uint32_t nb = c - C, m = C - c;                           // m = -nbBits: a funnel shift right by m mod 32 = 32 - nbBits
            uint32_t pre, ntot; prefix(nb, pre, ntot);
            const uint32_t hi1 = hi << pre;
            if (ZB) st = (uint32_t)((((uint64_t)e << 32) | hi1) >> (32u - nb));   // nbBits may be 0: a 64-bit shift by 32 is well defined
            else st = __builtin_amdgcn_alignbit(e, hi1, m);
            q += (int32_t)ntot;
        } else {
            *(ls_l16)(uintptr_t)(stgb + soff) = (uint16_t)st;
            const uint32_t c = (uint32_t)__builtin_clz(e);
            const uint32_t nb = c - C, m = C - c;
            uint32_t pre, ntot; prefix(nb, pre, ntot);
            const uint32_t hi = window(q - (int32_t)pre);                   // this state's own window
            if (ZB) st = (uint32_t)((((uint64_t)e << 32) | hi) >> (32u - nb));
            else st = __builtin_amdgcn_alignbit(e, hi, m);
            q += (int32_t)ntot;
        }
    };
    // N = 2: the 64 rounds of a chunk as ONE hand-scheduled instruction stream.  A lone wave issues in order, one instruction per
    // ~6 cycles whether it depends on its predecessor or not (tools/ubench_ls.hip), and a round is bound by that: what counts is
    // the number of instructions.  Sixteen per round:
    //   head  s_waitcnt (entry) | v_ffbh | v_sub: m = -nbBits | ds_write_b16: the state the round starts from, to the stage |
    //         s_waitcnt (window) | v_alignbit: the 32-bit window at q | v_and_dpp: the partner's m for the second state, 0 for the
    //         first | v_alignbit(window, window, that): a ROTATE, so the first state (shift 0) keeps the window and the second sees
    //         it past the first one's bits (what wraps into the low bits is never read: two states take <= 26 of the 32) |
    //         v_alignbit(entry, window', m): the next state | v_lshl_add: its table address | ds_read_u16
    //   tail  v_add_dpp: minus the round's bits | v_add: bit position | v_bfe + v_lshl_add: ring address | ds_read2_b32 (window)
    //         -- behind the look-up, in the shadow of its latency.
    // 115 cycles per round (tools/ubench_ls.hip: the table look-up alone, entry -> ffbh -> sub -> alignbit -> lshl_add -> entry, takes
    // 80: ~60 of LDS latency and ~4.3 per instruction, DPP, s_waitcnt and LDS issue alike; the stage store in the tail instead:
    // 122, because the window read then leaves later and the head waits for it; v_and_dpp in front of the window wait: 118).
    // The stage store, the s_waitcnt and the window v_alignbit cover the two wait states a DPP read of a freshly written VGPR
    // needs.  LDS queue at the top of a round, oldest first: entry, window.  v62 / v63 take the window's two dwords (a 64-bit asm
    // operand cannot be split into its halves, fixed registers can).
#define LS_ROUND_LOOKUP \
        "v_lshl_add_u32 %[at], %[st], 1, %[cb]\n\t" \
        "ds_read_u16 %[e], %[at]\n\t"
#define LS_ROUND_ADVANCE \
        "v_add_u32_dpp %[pre], %[m], %[m] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_add_u32 %[q], %[q], %[pre]\n\t"
#define LS_ROUND_WINDOW \
        "v_bfe_u32 %[at], %[q], 5, %[RW]\n\t" \
        "v_lshl_add_u32 %[at], %[at], 2, %[ringb]\n\t" \
        "ds_read2_b32 v[62:63], %[at] offset1:1\n\t"
#define LS_ROUND_HEAD \
        "s_waitcnt lgkmcnt(1)\n\t" \
        "v_ffbh_u32 %[c], %[e]\n\t" \
        "v_sub_u32 %[m], %[C], %[c]\n\t" \
        "ds_write_b16 %[stg], %[st] offset:ls_off\n\t" \
        ".set ls_off, ls_off+4\n\t" \
        "s_waitcnt lgkmcnt(1)\n\t" \
        "v_alignbit_b32 %[hi], v63, v62, %[q]\n\t" \
        "v_and_b32_dpp %[pre], %[m], %[mk1] quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_alignbit_b32 %[hi], %[hi], %[hi], %[pre]\n\t"
#define LS_ROUND_TAIL LS_ROUND_LOOKUP LS_ROUND_ADVANCE LS_ROUND_WINDOW
    // START / COUNT: stage byte offset of the first state and rounds of this piece (a whole chunk: 0, 64)
    // (a macro, not a lambda: clang does not capture through asm operands in a generic lambda)
#define LS_CHUNK2(START_, COUNT_) do { \
        uint32_t e, c, m, hi, pre, at; \
        if (ZB) \
            asm volatile(".set ls_off, %[S]\n\t" \
                         LS_ROUND_LOOKUP LS_ROUND_WINDOW \
                         ".rept %[R]\n\t" \
                         LS_ROUND_HEAD \
                         "v_alignbit_b32 %[hi], %[e], %[hi], %[m]\n\t" \
                         "v_cmp_eq_u32 vcc, 0, %[m]\n\t" \
                         "v_cndmask_b32 %[st], %[hi], %[e], vcc\n\t" \
                         LS_ROUND_TAIL \
                         ".endr\n\t" \
                         "s_waitcnt lgkmcnt(0)" \
                         : [st] "+v"(st), [q] "+v"(q), [e] "=&v"(e), [c] "=&v"(c), [m] "=&v"(m), [hi] "=&v"(hi), [pre] "=&v"(pre), [at] "=&v"(at) \
                         : [C] "v"(C), [mk1] "v"(mk1), [cb] "v"(cb), [ringb] "v"(ringb), [stg] "v"(stgb), [S] "n"(START_), [R] "n"(COUNT_), [RW] "n"(LS_RBITS) \
                         : "memory", "vcc", "v62", "v63"); \
        else \
            asm volatile(".set ls_off, %[S]\n\t" \
                         LS_ROUND_LOOKUP LS_ROUND_WINDOW \
                         ".rept %[R]\n\t" \
                         LS_ROUND_HEAD \
                         "v_alignbit_b32 %[st], %[e], %[hi], %[m]\n\t" \
                         LS_ROUND_TAIL \
                         ".endr\n\t" \
                         "s_waitcnt lgkmcnt(0)" \
                         : [st] "+v"(st), [q] "+v"(q), [e] "=&v"(e), [c] "=&v"(c), [m] "=&v"(m), [hi] "=&v"(hi), [pre] "=&v"(pre), [at] "=&v"(at) \
                         : [C] "v"(C), [mk1] "v"(mk1), [cb] "v"(cb), [ringb] "v"(ringb), [stg] "v"(stgb), [S] "n"(START_), [R] "n"(COUNT_), [RW] "n"(LS_RBITS) \
                         : "memory", "v62", "v63"); \
    } while (0)
    // N = 4: one LDS round trip per round as well (tools/ubench_ls.hip, k_cand4: 136 cycles per round of four symbols; with every
    // lane reading its own window behind the prefix sum -- two round trips -- it was 200).  The round's 64-bit window is read at
    // its start position with the table look-ups: three ring dwords from the dword of q - 32 on (q is moved down by 32 for the
    // chunk: the window's low half then starts at the tracked position) -> lo = bits [q-32, q), hi = bits [q, q+32) by two funnel
    // shifts; a state's bits are the top nbBits of (hi:lo) << pre, pre = the bits of the states in front of it: v_lshlrev_b64.
    // The prefix sum of m = -nbBits over the stream's four lanes runs in place on two v_add_u32_dpp, row_shr:1 and row_shr:2 without
    // bound_ctrl: a stream's lanes open a DPP row, so lane 0 (lanes 0, 1) has no source and keeps its value; the chunk runs with
    // EXEC = the states' lanes only.  The funnel shifts and a second v_sub for m fill the two wait states a DPP read of a fresh
    // VGPR needs.
    // The state a round ends with is staged in the tail, behind the window reads (the chunk's first one by the prologue).
    // LDS queue at the top of a round, oldest first: entry, window (2), stage.
#define LS_CHUNK4() do { \
        uint32_t e, c, a, m, p, at, tot; \
        const uint64_t ls_em = LS_SPW == 4 ? 0x000F000F000F000Full : LS_SPW == 3 ? 0x0000000F000F000Full : LS_SPW == 2 ? 0x00000000000F000Full : 0xFull; \
        q -= 32; \
        asm volatile(".set ls_off, 8\n\t" \
                     "s_mov_b64 exec, %[em]\n\t" \
                     "v_lshl_add_u32 %[at], %[st], 1, %[cb]\n\t" \
                     "ds_read_u16 %[e], %[at]\n\t" \
                     "v_bfe_u32 %[at], %[q], 5, %[RW]\n\t" \
                     "v_lshl_add_u32 %[at], %[at], 2, %[ringb]\n\t" \
                     "ds_read2_b32 v[60:61], %[at] offset1:1\n\t" \
                     "ds_read_b32 v62, %[at] offset:8\n\t" \
                     "ds_write_b16 %[stg], %[st]\n\t" \
                     ".rept 32\n\t" \
                     "s_waitcnt lgkmcnt(3)\n\t" \
                     "v_ffbh_u32 %[c], %[e]\n\t" \
                     "v_sub_u32 %[a], %[C], %[c]\n\t" \
                     "s_waitcnt lgkmcnt(1)\n\t" \
                     "v_alignbit_b32 v58, v61, v60, %[q]\n\t" \
                     "v_add_u32_dpp %[a], %[a], %[a] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
                     "v_alignbit_b32 v59, v62, v61, %[q]\n\t" \
                     "v_sub_u32 %[m], %[C], %[c]\n\t" \
                     "v_add_u32_dpp %[a], %[a], %[a] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
                     "v_sub_u32 %[p], %[m], %[a]\n\t" \
                     "v_lshlrev_b64 v[56:57], %[p], v[58:59]\n\t" \
                     ".if %[ZBF]\n\t" \
                     "v_alignbit_b32 %[p], %[e], v57, %[m]\n\t" \
                     "v_cmp_eq_u32 vcc, 0, %[m]\n\t" \
                     "v_cndmask_b32 %[st], %[p], %[e], vcc\n\t" \
                     ".else\n\t" \
                     "v_alignbit_b32 %[st], %[e], v57, %[m]\n\t" \
                     ".endif\n\t" \
                     "v_lshl_add_u32 %[at], %[st], 1, %[cb]\n\t" \
                     "ds_read_u16 %[e], %[at]\n\t" \
                     "v_mov_b32_dpp %[tot], %[a] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t" \
                     "v_add_u32 %[q], %[q], %[tot]\n\t" \
                     "v_bfe_u32 %[at], %[q], 5, %[RW]\n\t" \
                     "v_lshl_add_u32 %[at], %[at], 2, %[ringb]\n\t" \
                     "ds_read2_b32 v[60:61], %[at] offset1:1\n\t" \
                     "ds_read_b32 v62, %[at] offset:8\n\t" \
                     ".if ls_off < 256\n\t" \
                     "ds_write_b16 %[stg], %[st] offset:ls_off\n\t" \
                     ".endif\n\t" \
                     ".set ls_off, ls_off+8\n\t" \
                     ".endr\n\t" \
                     "s_waitcnt lgkmcnt(0)\n\t" \
                     "s_mov_b64 exec, -1" \
                     : [st] "+v"(st), [q] "+v"(q), [e] "=&v"(e), [c] "=&v"(c), [a] "=&v"(a), [m] "=&v"(m), [p] "=&v"(p), [at] "=&v"(at), [tot] "=&v"(tot) \
                     : [C] "v"(C), [cb] "v"(cb), [ringb] "v"(ringb), [stg] "v"(stgb), [RW] "n"(LS_RBITS), [ZBF] "n"(ZB ? 1 : 0), [em] "s"(ls_em) \
                     : "memory", "vcc", "v56", "v57", "v58", "v59", "v60", "v61", "v62"); \
        q += 32; \
    } while (0)
    // ---- chunks of 128 symbols per stream ---------------------------------------------------------------------------
    constexpr uint32_t R = 128 / N;                                         // rounds per chunk
    const uint32_t chunks = count / 128u, rem = count - chunks * 128u;
    uint32_t maxch = 0;
#pragma unroll
    for (int j = 0; j < LS_SPW; j++) if (s_have[j]) maxch = max(maxch, s_chunks[j]);
    maxch = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxch);           // uniform by construction; keep the chunk loop scalar
    uint32_t sv_st = st; int32_t sv_q = q;                                  // a shorter stream's true state after its last whole chunk
#ifdef LS_STAMP
    uint64_t stamp[4] = { 0, 0, 0, 0 }, t_prev = __builtin_amdgcn_s_memtime();
#define LS_T(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); stamp[i] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define LS_T(i) do { } while (0)
#endif
    for (uint32_t ch = 0; ch < maxch; ch++) {
        if (ch == chunks) { sv_st = st; sv_q = q; }                         // (per lane) this stream is done: it runs on, harmlessly, on its own table
        if (N == 2) LS_CHUNK2(0, 64);
        else if (N == 4) LS_CHUNK4();
        else {
#pragma unroll
            for (uint32_t r = 0; r < R; r++) round(r * N * 2u);
        }
        LS_T(0);
        // the chunk's 128 states per stream are staged: out they go (lane l: states 2l, 2l+1), and the ring moves on.
        // Order: every consumer of last chunk's prefetch first -- the wait on the memory counter is then for operations that
        // have had a whole chunk to complete -- then the new loads, then the stores (vmcnt counts loads and stores together, in
        // order: a ring store behind a fresh global store would wait for that store's acknowledgement).
#pragma unroll
        for (int j = 0; j < LS_SPW; j++) store_blk(j, s_pfb[j], pf[j]);
        LS_T(1);
#pragma unroll
        for (int j = 0; j < LS_SPW; j++) {
            s_blk[j] = ((int32_t)ls_rl((uint32_t)q, j * LS_LS) >> 5) >> 6;
            s_pfb[j] = s_blk[j] - LS_DEPTH;
            pf[j] = load_blk(j, s_pfb[j]);
        }
        LS_T(2);
        {
            uint32_t s2[LS_SPW];
#pragma unroll
            for (int j = 0; j < LS_SPW; j++) s2[j] = *(ls_l32)(uintptr_t)((wv * LS_SPW + (uint32_t)j) * LS_STREAM_BYTES + LS_STAGE + lane * 4);
#pragma unroll
            for (int j = 0; j < LS_SPW; j++)                                // (a finished stream decodes garbage: its descriptor ends at its last whole chunk)
                __builtin_amdgcn_raw_buffer_store_b32(s2[j], rs_out[j], (int)(ch * 256u + lane * 4u), 0, 2);
        }
        LS_T(3);
    }
#ifdef LS_STAMP
    if (lane == 0) { MicUnit &ud = units[list[slot0]]; for (int i = 0; i < 4; i++) ud.dbg[i] = (uint32_t)(stamp[i] >> 4); ud.dbg[4] = maxch; }
#endif
    if (chunks < maxch) { st = sv_st; q = sv_q; }                           // (per lane) back to the true end-of-chunks state
    if (maxch) {   // ... and the rings back to where those states read (the run-off moved them on); harmless for the longest stream
#pragma unroll
        for (int j = 0; j < LS_SPW; j++) {
            if (!s_have[j]) continue;
            const int32_t b = ((int32_t)ls_rl((uint32_t)q, j * LS_LS) >> 5) >> 6;
#pragma unroll
            for (int dd = -1; dd <= 2; dd++) store_blk(j, b - dd, load_blk(j, b - dd));   // (a tail of 127 symbols can take 64 dwords at tableLog 16)
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    // ---- tails: rem < 128 tokens per stream; a state past the stream's last token neither moves nor takes bits ----------
    {
        uint32_t maxrem = 0;
#pragma unroll
        for (int j = 0; j < LS_SPW; j++) if (s_have[j]) maxrem = max(maxrem, ls_rl(rem, j * LS_LS));
#pragma unroll 1
        for (uint32_t r = 0; r * N < maxrem; r++) {
            const bool valid = r * N + k < rem;                             // fse2state.go:293-305 and siblings: the last states in lane order
            const uint32_t e = entry(st);
            if (real) *(ls_l16)(uintptr_t)(stgb + r * N * 2u) = (uint16_t)st;
            const uint32_t nbr = (uint32_t)__builtin_clz(e) - C;
            const uint32_t nb = valid ? nbr : 0u;
            uint32_t pre, ntot; prefix(nb, pre, ntot);
            const uint32_t hi = window(q - (int32_t)pre);
            const uint32_t nx = (uint32_t)((((uint64_t)e << 32) | hi) >> (32u - nb));
            st = valid ? nx : st;
            q += (int32_t)ntot;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int j = 0; j < LS_SPW; j++) {
            if (!s_have[j]) continue;
            const uint32_t rj = ls_rl(rem, j * LS_LS), done = s_chunks[j] * 128u;
            const uint32_t s2 = *(ls_l32)(uintptr_t)((wv * LS_SPW + (uint32_t)j) * LS_STREAM_BYTES + LS_STAGE + lane * 4);
            const ls_gu16 o = (ls_gu16)(uintptr_t)s_out[j];
            if (2 * lane < rj) o[done + 2 * lane] = (uint16_t)s2;
            if (2 * lane + 1 < rj) o[done + 2 * lane + 1] = (uint16_t)(s2 >> 16);
        }
    }
    // ---- results: one lane per stream -------------------------------------------------------------------------------
    if (have && lane % LS_LS == 0 && lane < LS_SPW * LS_LS) {
        MicUnit &uo = units[list[slot0 + (int)g]];
        if (q + 32 - (int32_t)(8u * sb) < 0) uo.status = MICD_ERR_CORRUPT;  // bitreader.go:113-120: more bits taken than the stream holds
        else { uo.ntok = count; uo.walk_ok = 2; }                           // tok holds STATES: k_dec_translate turns them into symbols
    }
}

// ==========================================================================================
// States -> symbols, and the RLE header walk.  One group of 256 threads per unit that k_dec_tans_ls decoded (walk_ok == 2):
// the unit's symbol table (tableSymbol of each state, <= 16 KiB) sits in LDS, waves 1-3 stream the states through it in tiles
// of 1536 (16 bytes in, eight LDS look-ups, 16 bytes out, in place) and leave each translated tile in LDS as well, where wave 0
// follows the linked list of RLE headers through it (rledecompressu16.go:59-85: a header <= midCount is a run of one value, a
// larger one a literal chunk) one tile behind -- no dependent HBM read per header.
// The walk is a serial chain (the next header's position is this one's value), but a self-synchronising one: a walk started at
// a wrong token lands on a true header within a few steps and is the true walk from there.  Where headers are dense (run-heavy
// streams: WaveletV2 coefficients, WSI planes) every lane therefore walks its 24 tokens of the tile from their first token on
// its own, noting the positions it visits in a bit mask; the true walk then hops from lane to lane -- "is the position I
// arrive at in your mask? then your exit is mine" (two v_readlane per 24 tokens instead of a loop trip per header; a lane whose
// mask misses the arrival walks again from there) -- and the headers from the arrival bit on are the true ones.  Segment
// records then leave from all lanes at once behind two DPP prefix sums (record index, first symbol).  Where headers are sparse
// (long literal chunks: noisy frames) the plain header-to-header loop is the cheaper one; the previous tile's count picks.
// Stop and error rules are those of the walkers in mic_decode.hip: on an error the segments are dropped and the consumer
// (k_dec_pixels_wg) walks the stream itself and reports it.  Frames (mode 0) are walked, other units are translated only.
#define TR_THREADS 256                            // one walker wave + three translating waves: seven groups (walkers) per CU -- the walk is
                                                  // serial per unit, and on run-dense streams (WSI planes) it is what a unit waits for
#define TR_TILE ((TR_THREADS - 64) * 8)
#define TR_SEG (TR_TILE / 64)                     // tokens per walker lane
#define TR_DENSE 6                                // headers in a tile from which on the next one is walked by all lanes
static_assert(TR_SEG == 24 && TR_SEG * 64 == TR_TILE, "lane = position / 24 below is (position * 2731) >> 16");
#define TR_DPP(x, ctrl, rmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (rmask), 0xF, false)
__device__ __forceinline__ uint32_t tr_incl_add(uint32_t v) {             // wave-wide inclusive sum: row_shr 1/2/4/8, row_bcast 15 / 31
    v += TR_DPP(v, 0x111, 0xF); v += TR_DPP(v, 0x112, 0xF); v += TR_DPP(v, 0x114, 0xF); v += TR_DPP(v, 0x118, 0xF);
    v += TR_DPP(v, 0x142, 0xA); v += TR_DPP(v, 0x143, 0xC);
    return v;
}
template <int TRTL>   // 13: tables up to 2^13 states (two groups per CU) | 16: up to 2^16 (128 KiB of LDS: one group per CU)
__global__ void __launch_bounds__(TR_THREADS) k_dec_translate(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK || u.walk_ok != 2) return;
    if ((TRTL == 13) != (u.table_log <= 13)) return;
    __shared__ uint16_t s_sym[1 << TRTL];
    __shared__ __attribute__((aligned(16))) uint16_t s_tile[2][TR_TILE + 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t size = 1u << u.table_log, ntok = u.ntok;
    {
        const uint32_t *src = (const uint32_t *)u.tab_sym;                  // 256 KiB-aligned slab
        for (uint32_t i = tid; i < size / 2; i += TR_THREADS) ((uint32_t *)s_sym)[i] = src[i];
    }
    uint16_t *tok = u.tok;
    const bool frame = u.mode == 0 && u.seg != nullptr;
    // (a length-prefixed RLE stream, walk_mode 1 -- a whole WaveletV2 frame -- is one unit of millions of tokens: walking it here would
    //  be one wave's work per frame; it is marked walk_ok = 3 and walked in parts by k_rle_walk_parts, mic_wavelet.hip)
    const bool prefixed = u.mode == 1 && u.walk_mode == 1 && u.seg != nullptr;
    // walker (wave 0; uniform values)
    bool w_on = frame, w_err = false, w_dense = false;
    uint32_t w_pos = 0, w_out = 0, w_nseg = 0, w_mid = 0;
    const uint32_t w_cap = min(u.sym_cap, 2u * (uint32_t)u.w * (uint32_t)u.h + 2u);   // symbols the pixels can use
    const uint32_t w_segcap = u.seg_cap;
    typedef __attribute__((address_space(1))) ls_v2 *seg_p;
    const seg_p w_seg = (seg_p)u.seg;
    __syncthreads();
    typedef uint32_t tr_v4 __attribute__((ext_vector_type(4)));
    const uint32_t smask = size - 1;                                        // a state carries its +size offset: index = state - size
    for (uint32_t base = 0, it = 0; base < ntok; base += TR_TILE, it++) {
        const uint32_t tile_end = min(base + TR_TILE, ntok);
        uint16_t *tile = s_tile[it & 1];
        if (wave != 0) {
            const uint32_t l = (tid - 64) * 8, i0 = base + l;
            if (i0 + 8 <= tile_end) {
                const tr_v4 v = __builtin_nontemporal_load((const tr_v4 *)(tok + i0));
                tr_v4 o;
                o.x = (uint32_t)s_sym[v.x & smask] | ((uint32_t)s_sym[(v.x >> 16) & smask] << 16);
                o.y = (uint32_t)s_sym[v.y & smask] | ((uint32_t)s_sym[(v.y >> 16) & smask] << 16);
                o.z = (uint32_t)s_sym[v.z & smask] | ((uint32_t)s_sym[(v.z >> 16) & smask] << 16);
                o.w = (uint32_t)s_sym[v.w & smask] | ((uint32_t)s_sym[(v.w >> 16) & smask] << 16);
                *(tr_v4 *)(tok + i0) = o;
                *(tr_v4 *)(tile + l) = o;
            } else {
                for (uint32_t i = i0; i < tile_end; i++) { const uint16_t t = s_sym[tok[i] & smask]; tok[i] = t; tile[i - base] = t; }
            }
        }
        __syncthreads();                                                    // tile `it` is in LDS; the walker is done with tile it - 1
        if (wave != 0 || !w_on || w_pos >= tile_end) continue;
        if (w_pos == 0) {                                                   // token 0 fixes the run / literal split (rledecompressu16.go:21-25)
            const int d0 = mic_len16(tile[0]);
            if (d0 == 0) { w_on = false; w_err = true; continue; }
            w_mid = (1u << (d0 - 1)) - 1; w_pos = 1;
        }
        const uint32_t nseg_in = w_nseg;
        if (!w_dense) {
            // ---- header to header, from a 64-token window and v_readlane ----
            while (w_on && w_pos < tile_end) {
                const uint32_t rel = w_pos - base;                          // (>= 0: a header inside an earlier tile was taken there)
                const uint32_t wv = (rel + lane < tile_end - base) ? (uint32_t)tile[rel + lane] : 0u;
                uint32_t j = 0;
                while (j < 64 && w_on && w_pos < tile_end) {
                    const uint32_t hd = ls_rl(wv, (int)j);
                    if (w_out >= w_cap) { w_on = false; break; }
                    if (hd == 0 || w_nseg >= w_segcap) { w_on = false; w_err = true; break; }
                    uint32_t adv, len;
                    if (hd <= w_mid) {
                        if (w_pos + 1 >= ntok) { w_on = false; w_err = true; break; }
                        if (lane == 0) { ls_v2 r; r.x = (w_pos + 1) | 0x80000000u; r.y = w_out; w_seg[w_nseg] = r; }
                        len = hd; adv = 2;
                    } else {
                        if (lane == 0) { ls_v2 r; r.x = w_pos + 1; r.y = w_out; w_seg[w_nseg] = r; }
                        len = hd - w_mid; adv = 1 + len;
                    }
                    w_nseg++; w_out += len;
                    w_pos += adv; j += adv;                                 // j >= 64: the next header is outside the window
                }
            }
        } else {
            // ---- all lanes: lane l walks tokens [24 l, 24 l + 24) of the tile from the first one on; the true walk enters at rel0 ----
            const uint32_t tlen = tile_end - base, rel0 = w_pos - base;
            const uint32_t s0 = lane * TR_SEG, s1 = min(s0 + TR_SEG, tlen);
            uint32_t mask = 0, ex = 0;
            auto lane_walk = [&](uint32_t p) {                              // mask: visited positions (bit = position - s0); ex: where the walk leaves
                mask = 0;                                                   // the lane's tokens, ~0 when it met a zero header (an error if true)
                while (p < s1) {
                    const uint32_t h = tile[p];
                    if (h == 0) { p = 0xFFFFFFFFu; break; }
                    mask |= 1u << (p - s0);
                    p += (h <= w_mid) ? 2u : 1u + h - w_mid;
                }
                ex = p;
            };
            { const uint32_t p0 = max(s0, rel0); if (p0 < s1) lane_walk(p0); }
            uint32_t ent = 0xFFu;                                           // bit at which the true walk enters this lane's tokens (none: it jumps over them)
            uint32_t cur = rel0;
            while (cur < tlen) {
                const uint32_t s = (cur * 2731u) >> 16, bit = cur - s * TR_SEG;
                const uint32_t m = ls_rl(mask, (int)s);
                if (!((m >> bit) & 1u)) { if (lane == s) lane_walk(cur); }  // not where that lane's own walk went: it walks again from here
                if (lane == s) ent = bit;
                cur = ls_rl(ex, (int)s);
            }
            const uint32_t tm = (ent != 0xFFu) ? (mask & (0xFFFFFFFFu << ent)) : 0u;   // this lane's true headers
            uint32_t oc = 0;
            for (uint32_t t = tm; t; t &= t - 1) { const uint32_t h = tile[s0 + (uint32_t)__builtin_ctz(t)]; oc += (h <= w_mid) ? h : h - w_mid; }
            const uint32_t hc = (uint32_t)__popc(tm);
            const uint32_t ihc = tr_incl_add(hc), ioc = tr_incl_add(oc);
            const uint32_t tot_h = ls_rl(ihc, 63), tot_o = ls_rl(ioc, 63);
            uint32_t idx = w_nseg + ihc - hc, o = w_out + ioc - oc, nval = 0;
            bool bad = ent != 0xFFu && ex == 0xFFFFFFFFu && w_out + ioc < w_cap;   // the zero header is reached before the stream has its symbols
            for (uint32_t t = tm; t; t &= t - 1) {
                const uint32_t b = (uint32_t)__builtin_ctz(t), pos = base + s0 + b, h = tile[s0 + b];
                const bool run = h <= w_mid;
                const uint32_t len = run ? h : h - w_mid;
                if (o < w_cap) {
                    if (idx >= w_segcap || (run && pos + 1 >= ntok)) bad = true;
                    else {
                        ls_v2 r; r.x = (pos + 1) | (run ? 0x80000000u : 0u); r.y = o; w_seg[idx] = r;
                    }
                    nval++;
                }
                idx++; o += len;
            }
            if (__ballot(bad)) { w_on = false; w_err = true; continue; }
            if (w_out + tot_o >= w_cap) { w_nseg += ls_rl(tr_incl_add(nval), 63); w_on = false; }   // the stream has its symbols: headers behind that are not read
            else w_nseg += tot_h;
            w_out += tot_o;
            w_pos = base + cur;
        }
        w_dense = w_nseg - nseg_in >= TR_DENSE;
    }
    if (tid == 0) {
        if (frame && !w_err) { u.nseg = w_nseg; u.nsym = min(w_out, w_cap); u.walk_ok = 1; }
        else u.walk_ok = prefixed ? 3u : 0u;
    }
}

// States -> symbols for the units k_dec_translate has no header walk to do for (a whole WaveletV2 frame: one unit of millions of
// tokens, its headers walked in parts by k_rle_walk_parts) and whose table is the 128 KiB one: the table takes a CU's LDS, so a unit has
// the CU to itself whatever the kernel -- and then three translating waves behind a barrier per 1536 tokens leave it idle (2.0 ms for
// 256 CR frames).  Here: sixteen waves, no tile, no barrier; a wave takes every sixteenth block of 512 tokens.
#define TRW_THREADS 1024
__global__ void __launch_bounds__(TRW_THREADS) k_dec_translate_wide(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK || u.walk_ok != 2 || u.table_log <= 13) return;
    if (!(u.mode == 1 && u.walk_mode == 1 && u.seg != nullptr)) return;     // (frames and bare streams: k_dec_translate<16>, which walks)
    __shared__ uint16_t s_sym[1 << 16];
    const uint32_t tid = threadIdx.x;
    const uint32_t size = 1u << u.table_log, ntok = u.ntok, smask = size - 1;
    {
        const uint32_t *src = (const uint32_t *)u.tab_sym;
        for (uint32_t i = tid; i < size / 2; i += TRW_THREADS) ((uint32_t *)s_sym)[i] = src[i];
    }
    __syncthreads();
    uint16_t *tok = u.tok;
    typedef uint32_t tr_v4 __attribute__((ext_vector_type(4)));
    const uint32_t nvec = ntok / 8;
    for (uint32_t i = tid; i < nvec; i += TRW_THREADS) {
        const tr_v4 v = __builtin_nontemporal_load((const tr_v4 *)(tok + 8 * i));
        tr_v4 o;
        o.x = (uint32_t)s_sym[v.x & smask] | ((uint32_t)s_sym[(v.x >> 16) & smask] << 16);
        o.y = (uint32_t)s_sym[v.y & smask] | ((uint32_t)s_sym[(v.y >> 16) & smask] << 16);
        o.z = (uint32_t)s_sym[v.z & smask] | ((uint32_t)s_sym[(v.z >> 16) & smask] << 16);
        o.w = (uint32_t)s_sym[v.w & smask] | ((uint32_t)s_sym[(v.w >> 16) & smask] << 16);
        *(tr_v4 *)(tok + 8 * i) = o;
    }
    for (uint32_t i = 8 * nvec + tid; i < ntok; i += TRW_THREADS) tok[i] = s_sym[tok[i] & smask];
    __syncthreads();                                                        // (every state of the unit has been read as a state)
    if (tid == 0) u.walk_ok = 3u;                                           // translated; the headers are k_rle_walk_parts's
}

template <int N, bool ZB, int TL>
static void launch_ls_class(MicUnit *d_units, int n, const int *d_list, const int *d_count, hipStream_t stream, uint32_t cls_mask) {
    constexpr int cls = (TL <= 12 ? 4 : TL - 13) * 6 + (N == 2 ? 0 : N == 4 ? 2 : 4) + (ZB ? 1 : 0);
    if (!((cls_mask >> cls) & 1u)) return;                                  // (the session has not seen a stream of this class lately: mic_launch.h)
    constexpr int per = LsGeom<TL>::WAVES * LsGeom<TL>::SPW;
    static MicPerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute((const void *)k_dec_tans_ls<N, ZB, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, LsGeom<TL>::LDS); });
    const unsigned groups = (unsigned)((n + per - 1) / per);
    hipLaunchKernelGGL((k_dec_tans_ls<N, ZB, TL>), dim3(groups), dim3(64 * LsGeom<TL>::WAVES), LsGeom<TL>::LDS, stream, d_units,
                       d_list + (size_t)cls * (size_t)n, d_count + cls);
}
template <int TL>
static void launch_ls_tl(MicUnit *d_units, int n, const int *d_list, const int *d_count, hipStream_t stream, bool skip_first, uint32_t m) {
    if (!skip_first) launch_ls_class<2, false, TL>(d_units, n, d_list, d_count, stream, m);
    if constexpr (TL < 16) launch_ls_class<2, true, TL>(d_units, n, d_list, d_count, stream, m);
    launch_ls_class<4, false, TL>(d_units, n, d_list, d_count, stream, m);
    if constexpr (TL < 16) launch_ls_class<4, true, TL>(d_units, n, d_list, d_count, stream, m);
    launch_ls_class<8, false, TL>(d_units, n, d_list, d_count, stream, m);
    if constexpr (TL < 16) launch_ls_class<8, true, TL>(d_units, n, d_list, d_count, stream, m);
}

// d_list: LS_CLASSES * n ints, d_count: LS_CLASSES ints (session workspace)
void mic_launch_dec_tans_ls(MicUnit *d_units, int n, int *d_list, int *d_count, hipStream_t stream, MicTimer *t, uint32_t cls_mask) {
    if (t) t->mark("k_dec_classify");
    hipLaunchKernelGGL(k_dec_classify, dim3(1), dim3(1024), 0, stream, d_units, n, d_list, d_count);
    if (t) t->mark("k_dec_tans_ls<2,false,13>");
    launch_ls_class<2, false, 13>(d_units, n, d_list, d_count, stream, cls_mask);
    if (t) t->mark("k_dec_tans_ls<other,13>");
    launch_ls_tl<13>(d_units, n, d_list, d_count, stream, true, cls_mask);
    if (t) t->mark("k_dec_tans_ls<..12>");
    launch_ls_tl<12>(d_units, n, d_list, d_count, stream, false, cls_mask);
    if (t) t->mark("k_dec_tans_ls<14..16>");
    launch_ls_tl<14>(d_units, n, d_list, d_count, stream, false, cls_mask);
    launch_ls_tl<15>(d_units, n, d_list, d_count, stream, false, cls_mask);
    launch_ls_tl<16>(d_units, n, d_list, d_count, stream, false, cls_mask);
    if (t) t->mark("k_dec_translate");
    // classes 0-5: tableLog 13, 24-29: tableLog <= 12 -> the 16 KiB symbol table; 6-23: tableLog 14..16 -> the 128 KiB one
    if (cls_mask & 0x3F00003Fu) hipLaunchKernelGGL(k_dec_translate<13>, dim3((unsigned)n), dim3(TR_THREADS), 0, stream, d_units);
    if (cls_mask & 0x00FFFFC0u) {
        hipLaunchKernelGGL(k_dec_translate_wide, dim3((unsigned)n), dim3(TRW_THREADS), 0, stream, d_units);   // (marks its units walk_ok 3: the next launch skips them)
        hipLaunchKernelGGL(k_dec_translate<16>, dim3((unsigned)n), dim3(TR_THREADS), 0, stream, d_units);
    }
}
