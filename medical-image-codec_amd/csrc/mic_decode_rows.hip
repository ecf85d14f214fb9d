// mic_decode_rows.hip -- inverse Delta(avg) predictor for wide frames, ROW BY ROW: one wave per unit, the 64 lanes side by side
// inside a row.
//
//   out[y][x] = raw ? sym : ((left + top) >> 1) + sym - thr      (deltarlecompressu16.go:83-99; left only on row 0, top only in
//   column 0, 0 at the origin), 16-bit wrap-around.
//
// The recurrence runs along the row, so a row looks serial -- k_dec_predict / k_dec_predict2 (mic_decode_px.hip) therefore give
// every lane a ROW and walk a 64-row band as a diagonal wavefront, which costs them a transposition of every pixel through LDS,
// a barrier per step and one cache line per lane and access.  But k steps of the recurrence from an unknown left neighbour v have
// a closed form: with t_i the pixels above and e_i = sym_i - thr,
//   v_k = floor((v + A) / 2^k) + e_k,   A = sum over i = 1..k of 2^(i-1) * (e_(i-1) + t_i),   e_0 = 0
// (floor(floor(x / 2) + y) / 2) = floor((x + 2 y) / 4), by induction) -- as long as no intermediate value leaves 0..65535, i.e.
// as long as the 16-bit wrap-around never fires, which no stream written by an encoder does.  And for k >= 16 a pixel value v in
// 0..65535 can move floor((v + A) / 2^k) by at most one: the chunk is the step function v_k = c + (v >= theta ? 1 : 0).  Step
// functions compose into step functions, (c, theta) pairs are all a lane needs to say what its chunk does to whatever comes in
// from the left, and a DPP prefix scan over the 64 lanes hands every lane its true left neighbour.  So lane l owns the K
// consecutive pixels [l K, l K + K) of EVERY row (K = columns / 64 rounded up, 16..42: frames of 1009..2688 columns):
//   pass 1  A in three 32-bit limbs (one shift-add per pixel), then (c, theta); a lane whose chunk holds a raw pixel (stored
//           behind an escape: the recurrence restarts there) evaluates it directly -- its function is a constant;
//   scan    six DPP steps over (c, theta); lane 0 enters with left = top (column 0: (top + top) >> 1 = top);
//   pass 2  the chunk from its true left neighbour, with the check that nothing wrapped; the results stay in registers: they are
//           the next row's pixels above.  A row in which some value did wrap (damaged or adversarial streams only) is done again
//           by the lanes one after the other, with the reference's wrap-around at every pixel; so is row 0 (left only).
// Pixels come in and go out through a row buffer in LDS (128 K bytes per wave), so that every global access is a run of 1 KiB --
// lane i takes the row's 16-byte vector i, i + 64, ... -- while the arithmetic sees the lane's own 2 K bytes: K / 2 is odd for
// every chunk class (K = 18, 22 .. 42), so the lanes' dwords fall into different banks.  (First version: every lane loaded and
// stored its own 2 K bytes straight from / to memory, 16 bytes at a time at a stride of 2 K: 2.6 ms for the bench's 2304 strips,
// 1.0 ms without the stores -- sixty-four scattered 16-byte writes per instruction are sixty-four write requests.)  The next row
// is fetched while this one is computed; no barrier anywhere (a wave's LDS operations execute in order).
#include "mic_dev.h"
#include "mic_launch.h"

typedef uint32_t rw_v4 __attribute__((ext_vector_type(4)));
typedef uint32_t rw_v2 __attribute__((ext_vector_type(2)));
typedef rw_v4 RwQ __attribute__((aligned(2)));
typedef rw_v2 RwD __attribute__((aligned(2)));
typedef uint32_t RwS __attribute__((aligned(2)));
typedef __attribute__((address_space(1))) uint16_t *rw_gu16;
typedef const __attribute__((address_space(1))) uint32_t *rw_gcu32;

#define RW_NEVER 65536u
template <int CTRL, int RMASK> __device__ __forceinline__ uint32_t rw_dpp(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, RMASK, 0xF, false);
}

#ifndef RW_ABL
#define RW_ABL 0      // timing-only ablations (the output is wrong): 1 = no stores, 2 = one row's loads only, 4 = no pass 1 / scan, 8 = no pass 2
#endif
template <int K>
__global__ void __launch_bounds__(256, 3) k_dec_predict_rows(MicUnit *units, int n_units) {
    static_assert(K >= 16 && K <= 42 && (K & 3) == 2, "a chunk is 16..42 pixels (the step-function form needs 16, three 14-term limbs hold 42), an odd number of dwords");
    constexpr int KD = K / 2;                                           // dwords of a chunk
    constexpr int NVJ = (8 * K + 63) / 64;                              // rounds of 64 vectors that cover the row buffer
    extern __shared__ __attribute__((aligned(16))) uint32_t s_rw[];
    const int ui = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);       // four units per group: a group's waves land on the four SIMDs
    if (ui >= n_units) return;
    MicUnit &u = units[ui];
    if (u.status != MICD_OK || u.mode != 0 || u.pred) return;
    const int W = u.w, H = u.h;
    if (mic_rows_k(W) != K) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t *const rb = s_rw + (threadIdx.x >> 6) * (32 * K + 4);      // this wave's row buffer: the row's bytes as they lie in memory
    const uint32_t npx = (uint32_t)W * (uint32_t)H;
    const uint32_t thr = u.dec_thr;
    // (the unit's base as a wave-uniform value: every vector access is then scalar base + one 32-bit lane offset + an immediate, and
    // no per-round address lives in -- or is spilled from -- vector registers: a spilled store address made every store wait for the
    // one before it, 2.3 ms instead of 1.x)
    const uintptr_t px_u = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uintptr_t)u.px_out >> 32)) << 32) |
                           (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)u.px_out);
    const rw_gu16 px = (rw_gu16)px_u;
    typedef __attribute__((address_space(1))) uint8_t *rw_gu8;
    const uint32_t voff = lane * 16u;                                   // byte offset of this lane's vector inside a round
    const rw_gcu32 flags = (rw_gcu32)u.flags;
    const uint32_t x0 = lane * K;
    const int nv = min(max(W - (int)x0, 0), K);                        // pixels of a row in this lane's chunk
    const int nl = (W + K - 1) / K;                                     // lanes with pixels
    const uint64_t vmask = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
    const uint32_t thr2 = thr | (thr << 16);

    // tp: the row above = this lane's previous results, two pixels a dword.  e: the row's symbols minus thr, as packed 16-bit values.
    // A symbol that no encoder writes (above thr + 32767) wraps there -- and then its pixel computes as negative, which is the same
    // alarm as a wrap-around of the reference's own: the row is done again with the reference's arithmetic (mod 2^16 throughout).
    uint32_t tp[KD];
#pragma unroll
    for (int q = 0; q < KD; q++) tp[q] = 0;
    auto lo = [](uint32_t x) -> uint32_t { return x & 0xFFFFu; };
    auto hi = [](uint32_t x) -> uint32_t { return x >> 16; };
    auto slo = [](uint32_t x) -> int32_t { return (int32_t)(int16_t)(x & 0xFFFFu); };
    auto shi = [](uint32_t x) -> int32_t { return (int32_t)x >> 16; };
    auto israw = [](uint64_t raw, int j) -> bool { return j < 32 ? (((uint32_t)raw >> (j & 31)) & 1u) != 0u : (((uint32_t)(raw >> 32) >> (j & 31)) & 1u) != 0u; };
    typedef uint16_t rw_pk __attribute__((ext_vector_type(2)));
    // row y's vectors on their way to registers: vector lane + 64 j of the row, all NVJ rounds whatever the row's length -- what
    // lies behind the row's end is the next row's (cache lines the next fetch wants anyway) and is not used; the address is
    // clamped to the unit's last eight pixels, which only moves vectors that hold none of this row's pixels -- or, in the unit's
    // LAST row, the vector that holds its last few: land() fetches those again.
    const uint32_t nfull = (uint32_t)W / 8u, ntail = (uint32_t)W % 8u;  // whole vectors of a row, pixels behind them
    auto issue = [&](int y, rw_v4 (&nv4)[NVJ], uint64_t &raw) {
        const uint32_t ps = (uint32_t)y * (uint32_t)W;                  // the row's first pixel
        const rw_gu8 rowp = (rw_gu8)(px + ps);
        if (ps + 64u * 8u * NVJ <= npx) {                               // (every row but the unit's last few)
#pragma unroll
            for (int j = 0; j < NVJ; j++) nv4[j] = *(const __attribute__((address_space(1))) RwQ *)(rowp + 1024 * j + voff);
        } else {
            const uint32_t lim = (npx - 8u - ps) * 2u;
#pragma unroll
            for (int j = 0; j < NVJ; j++) nv4[j] = *(const __attribute__((address_space(1))) RwQ *)(rowp + min(1024u * j + voff, lim));
        }
        const uint32_t p = ps + x0;
        const uint32_t pf = min(p, npx - 1), w = pf >> 5;                // (a lane without pixels reads the flags of the unit's last pixel)
        const uint32_t f0 = flags[w], f1 = flags[w + 1], f2 = flags[w + 2];   // (the flag slab has spare words behind the last pixel)
        const uint32_t l32 = __builtin_amdgcn_alignbit(f1, f0, pf), h32 = __builtin_amdgcn_alignbit(f2, f1, pf);
        raw = (((uint64_t)h32 << 32) | l32) & vmask;
    };
    // ... and from there through the row buffer into the lanes' chunks, as symbols minus thr
    auto land = [&](int y, const rw_v4 (&nv4)[NVJ], uint32_t (&e)[KD]) {
#pragma unroll
        for (int j = 0; j < NVJ; j++) {
            const uint32_t i = lane + 64u * (uint32_t)j;
            if (64 * j + 64 <= 8 * K || i < 8u * K) *(uint4 *)(rb + 4 * i) = make_uint4(nv4[j].x, nv4[j].y, nv4[j].z, nv4[j].w);
        }
        if (y == H - 1 && ntail != 0 && lane < ntail)                    // (the clamp moved the unit's last vector: its pixels one by one)
            ((uint16_t *)rb)[8 * nfull + lane] = px[(uint32_t)y * (uint32_t)W + 8 * nfull + lane];
        __builtin_amdgcn_s_waitcnt(0xC07F);                             // (wave-private LDS: the writes are in before the reads)
#pragma unroll
        for (int q = 0; q < KD; q++) {
            // What lies behind the row's end in the last lanes' chunks is computed along (nothing reads the results) as the symbol
            // that changes nothing, so that it cannot look like a wrap-around.
            const rw_pk a = __builtin_bit_cast(rw_pk, rb[lane * KD + q]), t2 = __builtin_bit_cast(rw_pk, thr2);
            const uint32_t df = __builtin_bit_cast(uint32_t, (rw_pk)(a - t2));
            e[q] = (2 * q + 2 <= nv || 2 * q + 1 == nv) ? df : 0u;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                             // (the reads are done before the buffer is written again)
    };
    auto put = [&](int y) {
        const uint32_t ps = (uint32_t)y * (uint32_t)W;
#pragma unroll
        for (int q = 0; q < KD; q++) rb[lane * KD + q] = tp[q];
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int j = 0; j < NVJ; j++) {
            const uint32_t i = lane + 64u * (uint32_t)j;
            const uint4 v = *(const uint4 *)(rb + 4 * min(i, 8u * K - 1u));
            if (i < nfull) *(__attribute__((address_space(1))) RwQ *)((rw_gu8)(px + ps) + 1024 * j + voff) = rw_v4{v.x, v.y, v.z, v.w};
        }
        if (lane < ntail) px[ps + 8 * nfull + lane] = ((const uint16_t *)rb)[8 * nfull + lane];   // the row's last pixels
        __builtin_amdgcn_s_waitcnt(0xC07F);
    };
    // the lanes one after the other, with the reference's arithmetic at every pixel (deltarlecompressu16.go:90-99)
    auto slow_row = [&](const uint32_t (&e)[KD], uint64_t raw, bool row0) {
        uint32_t carry = row0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)tp[0], 0) & 0xFFFFu;   // column 0: (top + top) >> 1 = top; the origin: 0
#pragma unroll 1
        for (int l = 0; l < nl; l++) {
            uint32_t v = carry;
            const bool mine = (int)lane == l;
#pragma unroll
            for (int q = 0; q < KD; q++) {
                const uint32_t p0 = row0 ? v : ((v + lo(tp[q])) >> 1);
                const uint32_t r0 = israw(raw, 2 * q) ? ((lo(e[q]) + thr) & 0xFFFFu) : ((p0 + lo(e[q])) & 0xFFFFu);
                const uint32_t p1 = row0 ? r0 : ((r0 + hi(tp[q])) >> 1);
                const uint32_t r1 = israw(raw, 2 * q + 1) ? ((hi(e[q]) + thr) & 0xFFFFu) : ((p1 + hi(e[q])) & 0xFFFFu);
                v = r1;
                tp[q] = mine ? (r0 | (r1 << 16)) : tp[q];
            }
            carry = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
        }
    };

    uint32_t e[KD]; rw_v4 nx[NVJ]; uint64_t raw, nraw = 0;
#pragma unroll
    for (int j = 0; j < NVJ; j++) nx[j] = rw_v4{0u, 0u, 0u, 0u};
    issue(0, nx, raw);
    land(0, nx, e);
    for (int y = 0; y < H; y++) {
        if (y + 1 < H && !((RW_ABL & 2) && y > 0)) issue(y + 1, nx, nraw);
        if (y == 0) slow_row(e, raw, true);
        else {
            const bool any_raw = __any(raw != 0ull);
            uint32_t v; int32_t c = 0; uint32_t th = RW_NEVER;
            if (!(RW_ABL & 4)) {
            // ---- pass 1: what the chunk does to an unknown left neighbour: A = sum of 2^(i-1) (e_(i-1) + t_i), in limbs of 14 terms ----
            int32_t lb[3] = { 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < KD; q++) {
                const int32_t s0 = (int32_t)lo(tp[q]) + (q > 0 ? shi(e[q > 0 ? q - 1 : 0]) : 0);
                const int32_t s1 = (int32_t)hi(tp[q]) + slo(e[q]);
                lb[(2 * q) / 14] += s0 << ((2 * q) % 14);
                lb[(2 * q + 1) / 14] += s1 << ((2 * q + 1) % 14);
            }
            // (the halves are picked out of the packed registers again in pass 2 -- for free, as operand selects: kept apart from pass 1
            // on, the 4 K / 2 unpacked values are what pushes the widest instance past three waves per SIMD)
#pragma unroll
            for (int q = 0; q < KD; q++) asm volatile("" : "+v"(tp[q]), "+v"(e[q]));
            __builtin_amdgcn_sched_barrier(0);
            const int64_t A = (int64_t)lb[0] + ((int64_t)lb[1] << 14) + ((int64_t)lb[2] << 28);
            const uint64_t rem = (uint64_t)A & ((1ull << K) - 1ull);
            c = (int32_t)(A >> K) + shi(e[KD - 1]);
            const uint64_t th64 = (1ull << K) - rem;                    // v >= th64: one more
            th = th64 > 65535ull ? RW_NEVER : (uint32_t)th64;
            if (any_raw && raw != 0ull) {                               // a raw pixel restarts the recurrence: the chunk's exit is a constant
                uint32_t v = 0;
#pragma unroll
                for (int q = 0; q < KD; q++) {
                    const uint32_t r0 = israw(raw, 2 * q) ? ((lo(e[q]) + thr) & 0xFFFFu) : ((((v + lo(tp[q])) >> 1) + lo(e[q])) & 0xFFFFu);
                    const uint32_t r1 = israw(raw, 2 * q + 1) ? ((hi(e[q]) + thr) & 0xFFFFu) : ((((r0 + hi(tp[q])) >> 1) + hi(e[q])) & 0xFFFFu);
                    v = r1;
                }
                c = (int32_t)v; th = RW_NEVER;
            }
            if (lane == 0) { c += (lo(tp[0]) >= th) ? 1 : 0; th = RW_NEVER; }   // column 0 takes the pixel above as its left neighbour
            // ---- scan: (earlier, then own); a lane without a source at a step keeps its own pair ----
            auto step = [&](uint32_t cg, uint32_t tg, bool has) {
                const int32_t ch = c + (((int32_t)cg >= (int32_t)th) ? 1 : 0);      // (c may be -1: the function's value below its step)
                const uint32_t thh = (th == cg + 1u) ? tg : RW_NEVER;
                c = has ? ch : c; th = has ? thh : th;
            };
            { const uint32_t cg = rw_dpp<0x111, 0xF>((uint32_t)c, (uint32_t)c), tg = rw_dpp<0x111, 0xF>(th, th); step(cg, tg, (lane & 15u) >= 1u); }
            { const uint32_t cg = rw_dpp<0x112, 0xF>((uint32_t)c, (uint32_t)c), tg = rw_dpp<0x112, 0xF>(th, th); step(cg, tg, (lane & 15u) >= 2u); }
            { const uint32_t cg = rw_dpp<0x114, 0xF>((uint32_t)c, (uint32_t)c), tg = rw_dpp<0x114, 0xF>(th, th); step(cg, tg, (lane & 15u) >= 4u); }
            { const uint32_t cg = rw_dpp<0x118, 0xF>((uint32_t)c, (uint32_t)c), tg = rw_dpp<0x118, 0xF>(th, th); step(cg, tg, (lane & 15u) >= 8u); }
            { const uint32_t cg = rw_dpp<0x142, 0xA>((uint32_t)c, (uint32_t)c), tg = rw_dpp<0x142, 0xA>(th, th); step(cg, tg, ((lane >> 4) & 1u) != 0u); }
            { const uint32_t cg = rw_dpp<0x143, 0xC>((uint32_t)c, (uint32_t)c), tg = rw_dpp<0x143, 0xC>(th, th); step(cg, tg, lane >= 32u); }
            v = rw_dpp<0x138, 0xF>(lo(tp[0]), (uint32_t)c);              // wave_shr:1 -- the left neighbour; lane 0: the pixel above
            } else v = lo(tp[0]);
            __builtin_amdgcn_sched_barrier(0);
            // ---- pass 2: the chunk, and whether anything left 0..65535 on the way ----
            uint32_t bad = 0;
            if (RW_ABL & 8) { }
            else if (!any_raw) {
#pragma unroll
                for (int q = 0; q < KD; q++) {
                    const uint32_t r0 = ((v + lo(tp[q])) >> 1) + (uint32_t)slo(e[q]);
                    const uint32_t r1 = ((r0 + hi(tp[q])) >> 1) + (uint32_t)shi(e[q]);
                    bad |= r0 | r1; tp[q] = r0 | (r1 << 16); v = r1;
                }
            } else {
#pragma unroll
                for (int q = 0; q < KD; q++) {
                    const uint32_t f0 = ((v + lo(tp[q])) >> 1) + (uint32_t)slo(e[q]);
                    const uint32_t r0 = israw(raw, 2 * q) ? ((lo(e[q]) + thr) & 0xFFFFu) : f0;
                    const uint32_t f1 = ((r0 + hi(tp[q])) >> 1) + (uint32_t)shi(e[q]);
                    const uint32_t r1 = israw(raw, 2 * q + 1) ? ((hi(e[q]) + thr) & 0xFFFFu) : f1;
                    bad |= r0 | r1; tp[q] = r0 | (r1 << 16); v = r1;
                }
            }
            if (nv == 0) bad = 0;
            if (__any((bad >> 16) != 0u)) {                             // the wrap-around fired somewhere: the row above again, then lane by lane
                const uint32_t p = (uint32_t)(y - 1) * (uint32_t)W + x0;
#pragma unroll
                for (int q = 0; q < KD; q++) tp[q] = (uint32_t)px[min(p + 2 * q, npx - 1)] | ((uint32_t)px[min(p + 2 * q + 1, npx - 1)] << 16);
                slow_row(e, raw, false);
            }
        }
        if (!(RW_ABL & 1)) put(y);
        land(y + 1, nx, e);
        raw = nraw;
    }
}

void mic_launch_decode_rows(MicUnit *d_units, int n, hipStream_t stream, uint32_t kmask) {
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
#define RW_LAUNCH(K_) if (kmask & (1u << (((K_) - 18) / 4))) hipLaunchKernelGGL(k_dec_predict_rows<K_>, grid, block, 4 * 4 * (32 * (K_) + 4), stream, d_units, n)
    RW_LAUNCH(18); RW_LAUNCH(22); RW_LAUNCH(26); RW_LAUNCH(30); RW_LAUNCH(34); RW_LAUNCH(38); RW_LAUNCH(42);
#undef RW_LAUNCH
}
