// mic_tables_par.h -- work-group parallel construction of the FSE tables.
//
// The reference builds its tables with serial loops (fsecompressu16.go:329-431,
// fsedecompressu16.go:198-263): lay the low-probability symbols at the top of the table, walk
// the table with step (size>>1)+(size>>3)+3 skipping the low-probability area, then number the
// slots of every symbol in ascending table position.  The same table falls out of three
// data-parallel steps:
//   1. prefix sums over the symbols: first spread visit of each symbol (positive counts only),
//      rank of each low-probability symbol, cumulative |norm|;
//   2. the visit sequence: raw index J -> position (J*step) & mask, kept when <= highThreshold;
//      a prefix sum of "kept" turns J into the visit number t, visit_pos[t] = position;
//   3. the slot number of a position = its rank among the positions of the same symbol.  A
//      symbol with few slots sorts them in registers; a symbol with many slots is ranked through a
//      position bitmap and a popcount prefix: by a wave of its own when the table fits the LDS path
//      (sixteen such symbols at a time), by the whole work-group otherwise.
// Each (position, symbol, rank) triple is handed to an emit functor: the encoder fills
// stateTable[cumul[s] + rank] = size + position, the decoder the transition of that position.
#pragma once
#include "mic_dev.h"

#define TP_THREADS 1024
#define TP_WAVES 16
#define TP_SMALLV 16          // symbols with up to this many slots are ranked by one thread

__device__ __forceinline__ uint32_t tp_wave_incl_add(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(v, d); if (lane >= (uint32_t)d) v += o; }
    return v;
}

// Work-group exclusive scan of one value per thread; returns the exclusive prefix, *total = sum.
// s_tmp: TP_WAVES words of LDS.  Contains two barriers.
__device__ __forceinline__ uint32_t tp_block_excl(uint32_t v, uint32_t *s_tmp, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = tp_wave_incl_add(v, lane);
    __syncthreads();
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TP_WAVES; w++) { const uint32_t x = s_tmp[w]; if ((uint32_t)w < wave) off += x; tot += x; }
    *total = tot;
    return off + incl - v;
}

// Scratch handed to the core.  NormT/IdxT: int16/uint16 when everything lives in LDS
// (symbol_len <= 8192, tableLog <= 13), int32/uint32 in HBM otherwise.
template <typename NormT, typename IdxT>
struct TpScratch {
    const NormT *norm;     // [symbol_len]
    IdxT *first_visit;     // [symbol_len] first spread visit of the symbol (positive counts)
    IdxT *cum_all;         // [symbol_len] cumulative |norm| in symbol order (the reference's cumul[] / total)
    uint16_t *visit_pos;   // [size] table position of spread visit t
    uint32_t *bitmap;      // [size/32 + 1] LDS
    uint32_t *wprefix;     // [size/32 + 1] LDS
    uint32_t *big_list;    // [size / TP_SMALLV + 2] LDS: symbols ranked by the whole group
    uint32_t *s_tmp;       // [TP_WAVES + 4] LDS
};

// Returns MICD_OK or MICD_ERR_INTERNAL (sum of |norm| != table size).  All threads must call.
// emit(position, symbol, rank, slots) is called exactly once per table position.
template <typename NormT, typename IdxT, typename Emit>
__device__ int tp_build(const TpScratch<NormT, IdxT> &S, uint32_t symbol_len, uint32_t tl, Emit emit) {
    const uint32_t tid = threadIdx.x;
    const uint32_t size = 1u << tl, mask = size - 1;
    const uint32_t step = mic_table_step(size);
    // ---- 1. prefix sums over the symbols ------------------------------------------------------
    uint32_t carry_pos = 0, carry_low = 0, carry_all = 0;
    for (uint32_t base = 0; base < symbol_len; base += TP_THREADS) {
        const uint32_t s = base + tid;
        const int32_t v = (s < symbol_len) ? (int32_t)S.norm[s] : 0;
        const uint32_t vp = v > 0 ? (uint32_t)v : 0u, vl = v == -1 ? 1u : 0u;
        // pack: positives in the low 20 bits, low-prob count above (both <= 65536 per tile)
        uint32_t tot;
        const uint32_t ex = tp_block_excl(vp | (vl << 20), S.s_tmp, &tot);
        if (s < symbol_len) {
            const uint32_t pos_before = carry_pos + (ex & 0xFFFFF), low_before = carry_low + (ex >> 20);
            S.first_visit[s] = (IdxT)(v == -1 ? low_before : pos_before);   // low symbols keep their rank here
            S.cum_all[s] = (IdxT)(carry_all + (ex & 0xFFFFF) + (ex >> 20));
        }
        carry_pos += tot & 0xFFFFF; carry_low += tot >> 20; carry_all += (tot & 0xFFFFF) + (tot >> 20);
    }
    if (carry_all != size) return MICD_ERR_INTERNAL;       // fsecompressu16.go:365-367 / position != 0
    const uint32_t nlow = carry_low;
    const uint32_t high_threshold = size - 1 - nlow;
    // ---- 2. visit sequence ------------------------------------------------------------------------
    if (nlow == 0) {
        for (uint32_t t = tid; t < size; t += TP_THREADS) S.visit_pos[t] = (uint16_t)((t * step) & mask);
    } else {
        uint32_t carry = 0;
        for (uint32_t base = 0; base < size; base += TP_THREADS) {
            const uint32_t J = base + tid;
            const uint32_t p = (J * step) & mask;
            const uint32_t keep = (J < size && p <= high_threshold) ? 1u : 0u;
            uint32_t tot;
            const uint32_t ex = tp_block_excl(keep, S.s_tmp, &tot);
            if (keep) S.visit_pos[carry + ex] = (uint16_t)p;
            carry += tot;
        }
    }
    if (tid == 0) S.s_tmp[TP_WAVES] = 0;                   // number of big symbols
    __syncthreads();
    // ---- 3a. symbols with few slots: rank in registers ----------------------------------------------
    for (uint32_t s = tid; s < symbol_len; s += TP_THREADS) {
        const int32_t v = (int32_t)S.norm[s];
        if (v == -1) { emit(size - 1 - (uint32_t)S.first_visit[s], s, 0u, 1u); continue; }   // fsecompressu16.go:341-346
        if (v <= 0) continue;
        const uint32_t a = (uint32_t)S.first_visit[s];
        if (v == 1) { emit((uint32_t)S.visit_pos[a], s, 0u, 1u); continue; }
        if (v > TP_SMALLV) { const uint32_t k = atomicAdd(&S.s_tmp[TP_WAVES], 1u); S.big_list[k] = s; continue; }
        uint32_t p[TP_SMALLV];
#pragma unroll
        for (int i = 0; i < TP_SMALLV; i++) p[i] = (i < v) ? (uint32_t)S.visit_pos[a + i] : 0xFFFFFFFFu;
        // rank by counting (registers only: static indices)
#pragma unroll
        for (int i = 0; i < TP_SMALLV; i++) {
            if (i < v) {
                uint32_t r = 0;
#pragma unroll
                for (int j = 0; j < TP_SMALLV; j++) r += (p[j] < p[i]) ? 1u : 0u;
                emit(p[i], s, r, (uint32_t)v);
            }
        }
    }
    __syncthreads();
    // ---- 3b. symbols with many slots: bitmap + popcount prefix ----------------------------------------------------
    const uint32_t nbig = S.s_tmp[TP_WAVES];
    const uint32_t nwords = (size + 31) / 32;
    if (sizeof(IdxT) == 2) {
        // tableLog <= 13 (everything in LDS): a WAVE per symbol, sixteen symbols at a time, no group barrier.  The bitmap of a
        // table is at most 256 words: wave w owns words [256 w, 256 w + 256) of the bitmap / prefix area (16 KiB) and 256
        // 16-bit word prefixes behind the list of big symbols (at most size / 17 of them, 2 KiB of its 16).
        const uint32_t lane = tid & 63, wave = tid >> 6;
        uint32_t *bm = S.bitmap + wave * 256;
        uint16_t *pf = (uint16_t *)(S.big_list + 1024) + wave * 256;
        for (uint32_t bi = wave; bi < nbig; bi += TP_WAVES) {
            const uint32_t s = S.big_list[bi];
            const uint32_t v = (uint32_t)S.norm[s], a = (uint32_t)S.first_visit[s];
            for (uint32_t w = lane; w < nwords; w += 64) bm[w] = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
            for (uint32_t i = lane; i < v; i += 64) { const uint32_t p = S.visit_pos[a + i]; atomicOr(&bm[p >> 5], 1u << (p & 31)); }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
            uint32_t c[4], tot = 0;                                         // lane l owns words 4 l .. 4 l + 3
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t w = lane * 4 + (uint32_t)k; c[k] = (w < nwords) ? (uint32_t)__popc(bm[w]) : 0u; tot += c[k]; }
            uint32_t run = tp_wave_incl_add(tot, lane) - tot;
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t w = lane * 4 + (uint32_t)k; if (w < nwords) pf[w] = (uint16_t)run; run += c[k]; }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
            for (uint32_t i = lane; i < v; i += 64) {
                const uint32_t p = S.visit_pos[a + i];
                const uint32_t r = (uint32_t)pf[p >> 5] + (uint32_t)__popc(bm[p >> 5] & ((1u << (p & 31)) - 1u));
                emit(p, s, r, v);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        return MICD_OK;
    }
    // larger tables (scratch in HBM): the whole group per symbol
    for (uint32_t bi = 0; bi < nbig; bi++) {
        const uint32_t s = S.big_list[bi];
        const uint32_t v = (uint32_t)S.norm[s], a = (uint32_t)S.first_visit[s];
        for (uint32_t w = tid; w < nwords; w += TP_THREADS) S.bitmap[w] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < v; i += TP_THREADS) { const uint32_t p = S.visit_pos[a + i]; atomicOr(&S.bitmap[p >> 5], 1u << (p & 31)); }
        __syncthreads();
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nwords; base += TP_THREADS) {
            const uint32_t w = base + tid;
            const uint32_t c = (w < nwords) ? (uint32_t)__popc(S.bitmap[w]) : 0u;
            uint32_t tot;
            const uint32_t ex = tp_block_excl(c, S.s_tmp, &tot);
            if (w < nwords) S.wprefix[w] = carry + ex;
            carry += tot;
        }
        __syncthreads();
        for (uint32_t i = tid; i < v; i += TP_THREADS) {
            const uint32_t p = S.visit_pos[a + i];
            const uint32_t r = S.wprefix[p >> 5] + (uint32_t)__popc(S.bitmap[p >> 5] & ((1u << (p & 31)) - 1u));
            emit(p, s, r, v);
        }
        __syncthreads();
    }
    return MICD_OK;
}
