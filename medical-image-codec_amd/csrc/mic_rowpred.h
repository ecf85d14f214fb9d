// mic_rowpred.h -- the inverse Delta(avg) predictor of ONE ROW with the 64 lanes of a wave side by side in the row (shared by
// k_dec_predict_rows, mic_decode_rows.hip, and k_dec_rows_tok, mic_decode_fused.hip).
//
//   out[y][x] = raw ? sym : ((left + top) >> 1) + sym - thr      (deltarlecompressu16.go:83-99; left only on row 0, top only in
//   column 0, 0 at the origin), 16-bit wrap-around.
//
// k steps of the recurrence from an unknown left neighbour v have a closed form: with t_i the pixels above and e_i = sym_i - thr,
//   v_k = floor((v + A) / 2^k) + e_k,   A = sum over i = 1..k of 2^(i-1) * (e_(i-1) + t_i),   e_0 = 0
// (floor(floor(x / 2) + y) / 2) = floor((x + 2 y) / 4), by induction) -- as long as no intermediate value leaves 0..65535, i.e.
// as long as the 16-bit wrap-around never fires, which no stream written by an encoder does.  And for k >= 16 a pixel value v in
// 0..65535 can move floor((v + A) / 2^k) by at most one: the chunk is the step function v_k = c + (v >= theta ? 1 : 0).  Step
// functions compose into step functions, (c, theta) pairs are all a lane needs to say what its chunk does to whatever comes in
// from the left, and a DPP prefix scan over the 64 lanes hands every lane its true left neighbour.  Lane l owns the K consecutive
// pixels [l K, l K + K) of EVERY row (K = 18, 22 .. 42):
//   pass 1  A in three 32-bit limbs (one shift-add per pixel), then (c, theta); a lane whose chunk holds a raw pixel (stored
//           behind an escape: the recurrence restarts there) evaluates it directly -- its function is a constant;
//   scan    six DPP steps over (c, theta); lane 0 enters with left = top (column 0: (top + top) >> 1 = top);
//   pass 2  the chunk from its true left neighbour, with the check that nothing wrapped; the results stay in registers: they are
//           the next row's pixels above.  A row in which some value did wrap (damaged or adversarial streams only) is done again
//           by the lanes one after the other, with the reference's wrap-around at every pixel (slow_row); so is row 0 (left only).
// tp: the row above = this lane's previous results, two pixels a dword.  e: the row's symbols minus thr, as packed 16-bit values.
// A symbol that no encoder writes (above thr + 32767) wraps there -- and then its pixel computes as negative, which is the same
// alarm as a wrap-around of the reference's own.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define RW_NEVER 65536u
template <int CTRL, int RMASK> __device__ __forceinline__ uint32_t rw_dpp(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, RMASK, 0xF, false);
}
#ifndef RW_ABL
#define RW_ABL 0      // timing-only ablations (the output is wrong): 1 = no stores, 2 = one row's loads only, 4 = no pass 1 / scan, 8 = no pass 2
#endif

// the row's symbols minus thr in registers (k_dec_predict_rows) ...
template <int KD_> struct RowSymsReg {
    uint32_t v[KD_];
    __device__ __forceinline__ uint32_t operator[](int q) const { return v[q]; }
    __device__ __forceinline__ void fence() {
#pragma unroll
        for (int q = 0; q < KD_; q++) asm volatile("" : "+v"(v[q]));
    }
};
// ... or in the lane's KD dwords of an LDS row (k_dec_rows_tok): read where they are used, twice a row
struct RowSymsLds {
    const uint32_t *p;
    __device__ __forceinline__ uint32_t operator[](int q) const { return p[q]; }
    __device__ __forceinline__ void fence() { __builtin_amdgcn_sched_barrier(0); }
};

template <int K> struct RowPred {
    static constexpr int KD = K / 2;
    static __device__ __forceinline__ uint32_t lo(uint32_t x) { return x & 0xFFFFu; }
    static __device__ __forceinline__ uint32_t hi(uint32_t x) { return x >> 16; }
    static __device__ __forceinline__ int32_t slo(uint32_t x) { return (int32_t)(int16_t)(x & 0xFFFFu); }
    static __device__ __forceinline__ int32_t shi(uint32_t x) { return (int32_t)x >> 16; }
    static __device__ __forceinline__ bool israw(uint64_t raw, int j) {
        return j < 32 ? (((uint32_t)raw >> (j & 31)) & 1u) != 0u : (((uint32_t)(raw >> 32) >> (j & 31)) & 1u) != 0u;
    }
    // the lanes one after the other, with the reference's arithmetic at every pixel (deltarlecompressu16.go:90-99); nl = lanes with pixels
    // e: the row's symbols minus thr -- an array, or anything with operator[] (k_dec_rows_tok keeps them in LDS: twenty-one registers less)
    template <class E>
    static __device__ __forceinline__ void slow_row(uint32_t (&tp)[KD], const E &e, uint64_t raw, bool row0, uint32_t thr, uint32_t lane, int nl) {
        uint32_t carry = row0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)tp[0], 0) & 0xFFFFu;   // column 0: (top + top) >> 1 = top; the origin: 0
#pragma unroll 1
        for (int l = 0; l < nl; l++) {
            uint32_t v = carry;
            const bool mine = (int)lane == l;
#pragma unroll
            for (int q = 0; q < KD; q++) {
                const uint32_t eq = e[q];
                const uint32_t p0 = row0 ? v : ((v + lo(tp[q])) >> 1);
                const uint32_t r0 = israw(raw, 2 * q) ? ((lo(eq) + thr) & 0xFFFFu) : ((p0 + lo(eq)) & 0xFFFFu);
                const uint32_t p1 = row0 ? r0 : ((r0 + hi(tp[q])) >> 1);
                const uint32_t r1 = israw(raw, 2 * q + 1) ? ((hi(eq) + thr) & 0xFFFFu) : ((p1 + hi(eq)) & 0xFFFFu);
                v = r1;
                tp[q] = mine ? (r0 | (r1 << 16)) : tp[q];
            }
            carry = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
        }
    }
    // A row below the first: tp (the row above) becomes this row.  false: the wrap-around fired somewhere -- tp is spoilt, the caller
    // puts the row above back into it and calls slow_row.  nv = this lane's pixels in the row (0: its results are nobody's business).
    template <class E>
    static __device__ __forceinline__ bool fast_row(uint32_t (&tp)[KD], E &e, uint64_t raw, uint32_t thr, uint32_t lane, int nv) {
        const bool any_raw = __any(raw != 0ull);
        uint32_t v; int32_t c = 0; uint32_t th = RW_NEVER;
        if (!(RW_ABL & 4)) {
            // ---- pass 1: what the chunk does to an unknown left neighbour: A = sum of 2^(i-1) (e_(i-1) + t_i), in limbs of 14 terms ----
            int32_t lb[3] = { 0, 0, 0 };
            uint32_t eprev = 0;
#pragma unroll
            for (int q = 0; q < KD; q++) {
                const uint32_t eq = e[q];
                const int32_t s0 = (int32_t)lo(tp[q]) + (q > 0 ? shi(eprev) : 0);
                const int32_t s1 = (int32_t)hi(tp[q]) + slo(eq);
                lb[(2 * q) / 14] += s0 << ((2 * q) % 14);
                lb[(2 * q + 1) / 14] += s1 << ((2 * q + 1) % 14);
                eprev = eq;
            }
            // (the halves are picked out of the packed registers again in pass 2 -- for free, as operand selects: kept apart from pass 1
            // on, the 4 K / 2 unpacked values are what pushes the widest instance past three waves per SIMD)
#pragma unroll
            for (int q = 0; q < KD; q++) asm volatile("" : "+v"(tp[q]));
            e.fence();
            __builtin_amdgcn_sched_barrier(0);
            const int64_t A = (int64_t)lb[0] + ((int64_t)lb[1] << 14) + ((int64_t)lb[2] << 28);
            const uint64_t rem = (uint64_t)A & ((1ull << K) - 1ull);
            c = (int32_t)(A >> K) + shi(eprev);
            const uint64_t th64 = (1ull << K) - rem;                    // v >= th64: one more
            th = th64 > 65535ull ? RW_NEVER : (uint32_t)th64;
            if (any_raw && raw != 0ull) {                               // a raw pixel restarts the recurrence: the chunk's exit is a constant
                uint32_t w = 0;
#pragma unroll
                for (int q = 0; q < KD; q++) {
                    const uint32_t eq = e[q];
                    const uint32_t r0 = israw(raw, 2 * q) ? ((lo(eq) + thr) & 0xFFFFu) : ((((w + lo(tp[q])) >> 1) + lo(eq)) & 0xFFFFu);
                    const uint32_t r1 = israw(raw, 2 * q + 1) ? ((hi(eq) + thr) & 0xFFFFu) : ((((r0 + hi(tp[q])) >> 1) + hi(eq)) & 0xFFFFu);
                    w = r1;
                }
                c = (int32_t)w; th = RW_NEVER;
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
                const uint32_t eq = e[q];
                const uint32_t r0 = ((v + lo(tp[q])) >> 1) + (uint32_t)slo(eq);
                const uint32_t r1 = ((r0 + hi(tp[q])) >> 1) + (uint32_t)shi(eq);
                bad |= r0 | r1; tp[q] = r0 | (r1 << 16); v = r1;
            }
        } else {
#pragma unroll
            for (int q = 0; q < KD; q++) {
                const uint32_t eq = e[q];
                const uint32_t f0 = ((v + lo(tp[q])) >> 1) + (uint32_t)slo(eq);
                const uint32_t r0 = israw(raw, 2 * q) ? ((lo(eq) + thr) & 0xFFFFu) : f0;
                const uint32_t f1 = ((r0 + hi(tp[q])) >> 1) + (uint32_t)shi(eq);
                const uint32_t r1 = israw(raw, 2 * q + 1) ? ((hi(eq) + thr) & 0xFFFFu) : f1;
                bad |= r0 | r1; tp[q] = r0 | (r1 << 16); v = r1;
            }
        }
        if (nv == 0) bad = 0;
        return !__any((bad >> 16) != 0u);
    }
};
