// mic_decode_rows.hip -- inverse Delta(avg) predictor for wide frames, ROW BY ROW: one wave per unit, the 64 lanes side by side
// inside a row.
//
//   out[y][x] = raw ? sym : ((left + top) >> 1) + sym - thr      (deltarlecompressu16.go:83-99; left only on row 0, top only in
//   column 0, 0 at the origin), 16-bit wrap-around.
//
// The recurrence runs along the row, so a row looks serial -- k_dec_predict / k_dec_predict2 (mic_decode_px.hip) therefore give
// every lane a ROW and walk a 64-row band as a diagonal wavefront, which costs them a transposition of every pixel through LDS,
// a barrier per step and one cache line per lane and access.  mic_rowpred.h has the way around it: a chunk of >= 16 pixels acts on
// its unknown left neighbour as a step function, the lanes scan those, and every lane owns the same K columns of EVERY row (K =
// columns / 64 rounded up to 18, 22 .. 42: frames of 1009..2688 columns), so the row above never leaves its registers.
// Pixels come in and go out through a row buffer in LDS (128 K bytes per wave), so that every global access is a run of 1 KiB --
// lane i takes the row's 16-byte vector i, i + 64, ... -- while the arithmetic sees the lane's own 2 K bytes: K / 2 is odd for
// every chunk class (K = 18, 22 .. 42), so the lanes' dwords fall into different banks.  (First version: every lane loaded and
// stored its own 2 K bytes straight from / to memory, 16 bytes at a time at a stride of 2 K: 2.6 ms for the bench's 2304 strips,
// 1.0 ms without the stores -- sixty-four scattered 16-byte writes per instruction are sixty-four write requests.)  The next row
// is fetched while this one is computed; no barrier anywhere (a wave's LDS operations execute in order).
#include "mic_dev.h"
#include "mic_launch.h"
#include "mic_rowpred.h"

typedef uint32_t rw_v4 __attribute__((ext_vector_type(4)));
typedef uint32_t rw_v2 __attribute__((ext_vector_type(2)));
typedef rw_v4 RwQ __attribute__((aligned(2)));
typedef rw_v2 RwD __attribute__((aligned(2)));
typedef uint32_t RwS __attribute__((aligned(2)));
typedef __attribute__((address_space(1))) uint16_t *rw_gu16;
typedef const __attribute__((address_space(1))) uint32_t *rw_gcu32;

template <int K>
__global__ void __launch_bounds__(256, 3) k_dec_predict_rows(MicUnit *units, int n_units) {
    static_assert(K >= 16 && K <= 42 && (K & 3) == 2, "a chunk is 16..42 pixels (the step-function form needs 16, three 14-term limbs hold 42), an odd number of dwords");
    constexpr int KD = K / 2;                                           // dwords of a chunk
    constexpr int NVJ = (8 * K + 63) / 64;                              // rounds of 64 vectors that cover the row buffer
    extern __shared__ __attribute__((aligned(16))) uint32_t s_rw[];
    const int ui = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);       // four units per group: a group's waves land on the four SIMDs
    if (ui >= n_units) return;
    MicUnit &u = units[ui];
    if (u.status != MICD_OK || u.mode != 0 || u.pred || u.walk_ok == 4) return;   // (4: k_dec_rows_tok has made this unit's pixels)
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
    typedef RowPred<K> RP;
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
    auto land = [&](int y, const rw_v4 (&nv4)[NVJ], RowSymsReg<KD> &es) {
        uint32_t (&e)[KD] = es.v;
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
            e[q] = (2 * q + 2 <= nv) ? df : (2 * q + 1 == nv) ? (df & 0xFFFFu) : 0u;   // (the half of a dword behind the row's last pixel too)
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
    RowSymsReg<KD> e; rw_v4 nx[NVJ]; uint64_t raw, nraw = 0;
#pragma unroll
    for (int j = 0; j < NVJ; j++) nx[j] = rw_v4{0u, 0u, 0u, 0u};
    issue(0, nx, raw);
    land(0, nx, e);
    for (int y = 0; y < H; y++) {
        if (y + 1 < H && !((RW_ABL & 2) && y > 0)) issue(y + 1, nx, nraw);
        if (y == 0) RP::slow_row(tp, e, raw, true, thr, lane, nl);
        else if (!RP::fast_row(tp, e, raw, thr, lane, nv)) {            // the wrap-around fired somewhere: the row above again, then lane by lane
            const uint32_t p = (uint32_t)(y - 1) * (uint32_t)W + x0;
#pragma unroll
            for (int q = 0; q < KD; q++) tp[q] = (uint32_t)px[min(p + 2 * q, npx - 1)] | ((uint32_t)px[min(p + 2 * q + 1, npx - 1)] << 16);
            RP::slow_row(tp, e, raw, false, thr, lane, nl);
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
