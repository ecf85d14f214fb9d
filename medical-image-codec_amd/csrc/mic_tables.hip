// mic_tables.hip -- per-unit FSE table kernels (encode and decode side), one work-group of 1024
// threads per unit.
//
//   k_enc_tables_wg  histogram scan -> gates -> tableLog -> normalise -> NCount header -> CTable
//                    (fse2state.go:22-52; fsecompressu16.go:465-571, :191-289, :329-431)
//   k_dec_tables_wg  prefix / flavour -> NCount parse -> decode table
//                    (fse2state.go:102-116; fsedecompressu16.go:48-263; ransu16.go:77-135)
//
// What runs where: normalisation, the symbol prefix sums, the symbol spread and the slot
// numbering are data-parallel (mic_tables_par.h); the NCount header is a variable-width bit
// packing whose 32-bit accumulator wraps exactly like the reference's (fsecompressu16.go:260-262)
// and stays on one lane, reading norm[] from LDS.  "Small" units (alphabet <= 8192 symbols,
// tableLog <= 13: every 8..12-bit image) keep all scratch in 70 KiB of LDS as 16-bit values, two
// work-groups per CU; anything larger (16-bit CT: 65536 symbols, tableLog 16) uses the unit's
// HBM scratch slabs through the same code.
#include "mic_dev.h"
#include "mic_fse_tables.h"
#include "mic_tables_par.h"
#include "mic_launch.h"

#define TB_SMALL_SYMS 8192
#define TB_SMALL_TL 13

// LDS carve-up shared by both kernels (bytes)
#define TB_OFF_NORM   0                      // int16[8192]
#define TB_OFF_FIRST  (16 * 1024)            // uint16[8192]
#define TB_OFF_CUM    (32 * 1024)            // uint16[8192]
#define TB_OFF_VISIT  (48 * 1024)            // uint16[8192]           (small units only)
#define TB_OFF_BITMAP (64 * 1024)            // uint32[2048 + 1]
#define TB_OFF_WPREF  (64 * 1024 + 8256)     // uint32[2048 + 1]
#define TB_OFF_BIG    (64 * 1024 + 16512)    // uint32[4096 + 2]
#define TB_OFF_TMP    (64 * 1024 + 32960)    // uint32[128]
#define TB_LDS_BYTES  (64 * 1024 + 32960 + 512)    // ~96.4 KiB  (1 group / CU; the small path alone would fit 2)

__device__ __forceinline__ uint32_t tb_wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return v;
}

// ------------------------------------------------------------------------------------------
// writeCount (fsecompressu16.go:191-289) on the whole group, for tableLog <= 14 (a field is at most
// 15 bits there, so the reference's 32-bit accumulator never drops any; above that the serial
// writer reproduces its wrap-around).  The header is a concatenation of bit fields, LSB first:
//   [4: tableLog-5] then, per written symbol s,  [run code][count field]
// * written symbols: every non-zero count and the first zero of a zero-run; the other zeros of a
//   run (R of them) become the run code in front of the next written symbol: 16 one-bits per 24,
//   then 2 one-bits per 3, then R % 24 % 3 in 2 bits;
// * count field: remaining = 2^tl + 1 - sum of |count| before s (a prefix sum), threshold = the
//   largest power of two <= remaining (at most 2^tl), nbBits = log2(threshold) + 1,
//   max = 2*threshold - 1 - remaining, v = count + 1 (+ max when v >= threshold), nbBits - 1 bits when v < max.
// Bit offsets are a second prefix sum; fields are OR-ed into an LDS bit buffer and copied out.
// rem16 / s016: symbol_len entries each; bitbuf: zero-initialised here, bit_base + 8 * (((symbol_len * tl) >> 3) + 8) bits.
// Small alphabets: everything in LDS, the finished header is copied to `out`.  Large ones (HBM scratch): bitbuf is
// the output itself, addressed from the aligned word 2 bytes in front of it (bit_base = 16, to_out = false).
// For tableLog 16 a count field can be 17 bits; the reference's 32-bit accumulator holds 16 pending bits at
// most when a field arrives, so a field that would not fit (pending + length > 32) makes this return
// MICD_ERR_UNSUPPORTED and the caller runs the serial writer, which reproduces the wrap (fsecompressu16.go:260-262).
template <typename NormT, typename ScrT>
__device__ int tb_write_ncount_par(const NormT *norm, uint32_t symbol_len, uint32_t tl, ScrT *rem16, ScrT *s016,
                                   uint32_t *bitbuf, uint32_t bit_base, bool to_out, uint32_t *s_tmp, uint8_t *out, uint32_t cap,
                                   uint32_t *hdr_len) {
    const uint32_t tid = threadIdx.x;
    const uint32_t size = 1u << tl;
    const uint32_t max_header = ((symbol_len * tl) >> 3) + 3;
    if (cap < max_header + 2) return MICD_ERR_CAPACITY;
    const uint32_t nwords = (bit_base / 8 + max_header + 8 + 3) / 4;
    for (uint32_t i = tid; i < nwords; i += TP_THREADS) bitbuf[i] = 0;
    // pass A: remaining in front of every symbol, first zero of the run a symbol closes
    uint32_t carry_sum = 0, carry_fz = 0;
    for (uint32_t base = 0; base < symbol_len; base += TP_THREADS) {
        const uint32_t s2 = base + tid;
        const int32_t v = (s2 < symbol_len) ? (int32_t)norm[s2] : 0;
        const int32_t vp = (s2 > 0 && s2 < symbol_len) ? (int32_t)norm[s2 - 1] : 1;
        uint32_t tot;
        const uint32_t ex = tp_block_excl((uint32_t)(v < 0 ? -v : v), s_tmp, &tot);
        // exclusive max-scan of (index + 1) of first zeros
        const uint32_t fz = (s2 < symbol_len && v == 0 && vp != 0) ? s2 + 1 : 0u;
        uint32_t mi = fz;
        const uint32_t lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(mi, d); if (lane >= (uint32_t)d) mi = max(mi, o); }
        uint32_t me = __shfl_up(mi, 1); if (lane == 0) me = 0;
        __syncthreads();
        if (lane == 63) s_tmp[32 + wave] = mi;
        __syncthreads();
        uint32_t mtot = carry_fz;
        for (uint32_t w = 0; w < TP_WAVES; w++) { const uint32_t x = s_tmp[32 + w]; if (w < wave) me = max(me, x); mtot = max(mtot, x); }
        me = max(me, carry_fz);
        if (s2 < symbol_len) { rem16[s2] = (ScrT)(size + 1 - (carry_sum + ex)); s016[s2] = (ScrT)me; }
        carry_sum += tot; carry_fz = mtot;
    }
    if (carry_sum != size) return MICD_ERR_INTERNAL;          // (the serial writer ends with remaining != 1)
    __syncthreads();
    // field of symbol s: ones (run code body), then `lo` = [2-bit run remainder][count field] in lo_len bits
    auto field = [&](uint32_t s2, uint32_t &ones, uint64_t &lo, uint32_t &lo_len, uint32_t &flen_out) {
        ones = 0; lo = 0; lo_len = 0; flen_out = 0;
        const int32_t v = (int32_t)norm[s2];
        const int32_t vp = s2 > 0 ? (int32_t)norm[s2 - 1] : 1;
        if (v == 0 && vp == 0) return;                                    // inside a zero run: part of the run code
        if (vp == 0) {                                                    // closes a run of R + 1 zeros
            const uint32_t R = s2 - (uint32_t)s016[s2];
            const uint32_t r24 = R % 24;
            ones = 16 * (R / 24) + 2 * (r24 / 3);
            lo = r24 % 3; lo_len = 2;
        }
        const int32_t remaining = (int32_t)rem16[s2];
        const uint32_t hb = min(tl, (uint32_t)(31 - __clz((uint32_t)remaining)));
        const int32_t threshold = 1 << hb;
        const uint32_t nbb = hb + 1;
        const int32_t mx = (2 * threshold - 1) - remaining;
        int32_t cnt = v + 1;
        if (cnt >= threshold) cnt += mx;
        const uint32_t flen = nbb - ((cnt < mx) ? 1u : 0u);
        lo |= (uint64_t)(uint32_t)cnt << lo_len; lo_len += flen; flen_out = flen;
    };
    // pass B: offsets and packing
    uint32_t carry_bits = bit_base + 4;
    uint32_t wraps = 0;
    if (tid == 0) atomicOr(&bitbuf[bit_base >> 5], (tl - MIC_MIN_TABLELOG) << (bit_base & 31));
    for (uint32_t base = 0; base < symbol_len; base += TP_THREADS) {
        const uint32_t s2 = base + tid;
        uint32_t ones = 0, lo_len = 0, flen = 0; uint64_t lo = 0;
        if (s2 < symbol_len) field(s2, ones, lo, lo_len, flen);
        uint32_t tot;
        const uint32_t ex = tp_block_excl(ones + lo_len, s_tmp, &tot);
        uint32_t pos = carry_bits + ex;
        if (tl > 15 && flen) {                                            // pending bits when the count field arrives: 1 .. 16
            const uint32_t tb = pos - bit_base + ones + (lo_len - flen);
            if (((tb - 1) & 15u) + 1 + flen > 32) wraps = 1;
        }
        if (ones) {                                                       // bits [pos, pos + ones) set
            uint32_t left = ones;
            while (left) {
                const uint32_t sh = pos & 31, take = min(left, 32u - sh);
                const uint32_t m = (take == 32 ? 0xFFFFFFFFu : ((1u << take) - 1u)) << sh;
                atomicOr(&bitbuf[pos >> 5], m);
                pos += take; left -= take;
            }
        }
        if (lo_len) {                                                     // <= 19 bits
            const uint32_t sh = pos & 31;
            const uint64_t v = lo << sh;
            atomicOr(&bitbuf[pos >> 5], (uint32_t)v);
            if (sh + lo_len > 32) atomicOr(&bitbuf[(pos >> 5) + 1], (uint32_t)(v >> 32));
        }
        carry_bits += tot;
    }
    if (__syncthreads_or((int)wraps)) return MICD_ERR_UNSUPPORTED;
    const uint32_t nbytes = (carry_bits - bit_base + 7) >> 3;
    if (nbytes > max_header) return MICD_ERR_INTERNAL;
    if (to_out) {
        const uint8_t *bb = (const uint8_t *)bitbuf + bit_base / 8;
        for (uint32_t i = tid; i < nbytes + 8; i += TP_THREADS) out[i] = (i < nbytes) ? bb[i] : (uint8_t)0;   // + the 8 bytes k_enc_tans_wg ORs into
    }
    __threadfence_block();
    *hdr_len = nbytes;
    return MICD_OK;
}

// ------------------------------------------------------------------------------------------
template <typename NormT, typename IdxT>
__device__ void enc_tables_body(MicUnit &u, NormT *norm, IdxT *first_visit, IdxT *cum_all, uint16_t *visit_pos,
                                uint32_t *bitmap, uint32_t *wprefix, uint32_t *big_list, uint32_t *s_tmp,
                                uint32_t *s_misc, uint32_t n, uint32_t symbol_len, uint32_t tl) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t size = 1u << tl;
    const uint32_t *hist = u.hist;
    MIC_STAMP_BEGIN();
    // ---- normalise, primary method (fsecompressu16.go:524-571), one symbol per thread -----------
    {
        const uint64_t scale = 62 - (uint64_t)tl;
        const uint64_t step = (1ull << 62) / (uint64_t)n;
        const uint64_t v_step = 1ull << (scale - 20);
        const uint32_t low_threshold = n >> tl;
        uint32_t used = 0;                        // sum of proba (and 1 per low-prob symbol)
        uint32_t best_p = 0, best_s = 0;          // first symbol with the largest proba
        for (uint32_t s = tid; s < symbol_len; s += TP_THREADS) {
            const uint32_t cnt = hist[s];
            int32_t nv;
            if (cnt == 0) nv = 0;
            else if (cnt <= low_threshold) { nv = -1; used += 1; }
            else {
                int32_t proba = (int32_t)(((uint64_t)cnt * step) >> scale);
                if (proba < 8) {
                    const uint64_t rest_to_beat = v_step * (uint64_t)mic_rtb_table[proba];
                    const uint64_t v = (uint64_t)cnt * step - ((uint64_t)proba << scale);
                    if (v > rest_to_beat) proba++;
                }
                nv = proba; used += (uint32_t)proba;
                if ((uint32_t)proba > best_p) { best_p = (uint32_t)proba; best_s = s; }   // s grows: first max kept
            }
            norm[s] = (NormT)nv;
        }
        // reduce: sum(used); argmax(best_p) with the smallest symbol index on ties
        uint32_t sum = used;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d);
        // key = proba << 17 | (65536 - s)  -> larger proba wins, then smaller s   (proba <= 65536 -> use 64 bit)
        unsigned long long key = ((unsigned long long)best_p << 20) | (unsigned long long)(0xFFFFF - best_s);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const unsigned long long o = (unsigned long long)__shfl_xor((long long)key, d);
            key = key > o ? key : o;
        }
        __syncthreads();
        if (lane == 0) { s_tmp[wave] = sum; ((unsigned long long *)(s_tmp + 32))[wave] = key; }
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0; unsigned long long k = 0;
            for (int w = 0; w < TP_WAVES; w++) { tot += s_tmp[w]; const unsigned long long o = ((unsigned long long *)(s_tmp + 32))[w]; k = k > o ? k : o; }
            const int32_t still = (int32_t)size - (int32_t)tot;
            const uint32_t largest_p = (uint32_t)(k >> 20);
            const uint32_t largest = largest_p ? (0xFFFFF - (uint32_t)(k & 0xFFFFF)) : 0u;   // no proba > 0: largest stays 0
            int rc = MICD_OK; uint32_t second = 0;
            if (-still >= ((int32_t)norm[largest] >> 1)) second = 1;                           // corner case -> normalizeCount2
            else norm[largest] = (NormT)((int32_t)norm[largest] + still);
            s_misc[2] = second; s_misc[3] = (uint32_t)rc;
        }
        __syncthreads();
        if (s_misc[2]) {
            // normalizeCount2 (fsecompressu16.go:573-651).  Common for alphabets close to the table size
            // (12-bit frames at tableLog 13), so it runs on the whole group: two classification passes with
            // block reductions, then the weights  (end >> v) - (start >> v)  where `end` is a running sum of
            // count * rStep over the undecided symbols -- an exclusive 64-bit prefix sum.  The two corner
            // cases that turn on a serial loop in the reference stay serial (tid 0, HBM arrays).
            const int32_t not_yet = -2;
            uint64_t *s_t64 = (uint64_t *)(s_tmp + 32);                  // 16 wave partials + spare, 8-byte aligned
            auto reduce2 = [&](uint32_t a, uint64_t b, uint32_t &ra, uint64_t &rb) {
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) { a += (uint32_t)__shfl_xor((int)a, d); b += (uint64_t)__shfl_xor((long long)b, d); }
                __syncthreads();
                if (lane == 0) { s_tmp[wave] = a; s_t64[wave] = b; }
                __syncthreads();
                uint32_t ta = 0; uint64_t tb = 0;
                for (int w = 0; w < TP_WAVES; w++) { ta += s_tmp[w]; tb += s_t64[w]; }
                ra = ta; rb = tb;
            };
            uint32_t total = n;
            const uint32_t low_threshold2 = total >> tl;
            uint32_t low_one = (total * 3) >> (tl + 1);
            uint32_t d1 = 0; uint64_t sub1 = 0;
            for (uint32_t s2 = tid; s2 < symbol_len; s2 += TP_THREADS) {
                const uint32_t cnt = hist[s2];
                int32_t nv;
                if (cnt == 0) nv = 0;
                else if (cnt <= low_threshold2) { nv = -1; d1++; sub1 += cnt; }
                else if (cnt <= low_one) { nv = 1; d1++; sub1 += cnt; }
                else nv = not_yet;
                norm[s2] = (NormT)nv;
            }
            uint32_t distributed; uint64_t subs;
            reduce2(d1, sub1, distributed, subs);
            total -= (uint32_t)subs;
            int rc2 = MICD_OK;
            bool serial = false;
            uint32_t to_distribute = 0;
            if (distributed >= size) rc2 = MICD_ERR_INTERNAL;            // the reference divides by zero or spins (see mic_normalize_count2)
            else {
                to_distribute = size - distributed;
                if ((total / to_distribute) > low_one) {                 // uniform
                    low_one = (total * 3) / (to_distribute * 2);
                    uint32_t d2 = 0; uint64_t sub2 = 0;
                    for (uint32_t s2 = tid; s2 < symbol_len; s2 += TP_THREADS) {
                        const uint32_t cnt = hist[s2];
                        if ((int32_t)norm[s2] == not_yet && cnt <= low_one) { norm[s2] = (NormT)1; d2++; sub2 += cnt; }
                    }
                    uint32_t dd; uint64_t ss;
                    reduce2(d2, sub2, dd, ss);
                    distributed += dd; total -= (uint32_t)ss;
                    if (distributed >= size) rc2 = MICD_ERR_INTERNAL;
                    else to_distribute = size - distributed;
                }
                if (rc2 == MICD_OK && (distributed == symbol_len + 1 || total == 0)) serial = true;
            }
            if (rc2 == MICD_OK && !serial) {
                const uint64_t v_step_log = 62 - (uint64_t)tl;
                const uint64_t midv = (1ull << (v_step_log - 1)) - 1;
                const uint64_t r_step = (((1ull << v_step_log) * (uint64_t)to_distribute) + midv) / (uint64_t)total;
                uint64_t carry = midv;
                uint32_t badw = 0;
                for (uint32_t base = 0; base < symbol_len; base += TP_THREADS) {
                    const uint32_t s2 = base + tid;
                    const bool ny = s2 < symbol_len && (int32_t)norm[s2] == not_yet;
                    const uint64_t val = ny ? (uint64_t)hist[s2] * r_step : 0ull;
                    uint64_t incl = val;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const uint64_t o = (uint64_t)__shfl_up((long long)incl, d); if (lane >= (uint32_t)d) incl += o; }
                    __syncthreads();
                    if (lane == 63) s_t64[wave] = incl;
                    __syncthreads();
                    uint64_t off = 0, tot = 0;
                    for (int w = 0; w < TP_WAVES; w++) { const uint64_t x = s_t64[w]; if ((uint32_t)w < wave) off += x; tot += x; }
                    if (ny) {
                        const uint64_t start = carry + off + incl - val, end = start + val;
                        const uint32_t weight = (uint32_t)(end >> v_step_log) - (uint32_t)(start >> v_step_log);
                        if (weight < 1) badw = 1;
                        norm[s2] = (NormT)(int32_t)weight;
                    }
                    carry += tot;
                }
                if (__syncthreads_or((int)badw)) rc2 = MICD_ERR_INTERNAL;
            }
            if (rc2 == MICD_OK && serial) {
                __syncthreads();
                if (tid == 0) s_misc[3] = (uint32_t)mic_normalize_count2(u.hist, u.norm, symbol_len, n, tl);
                __threadfence_block();
                __syncthreads();
                rc2 = (int)s_misc[3];
                if (rc2 == MICD_OK && (void *)norm != (void *)u.norm)
                    for (uint32_t s2 = tid; s2 < symbol_len; s2 += TP_THREADS) norm[s2] = (NormT)u.norm[s2];
            }
            if (rc2 != MICD_OK) { if (tid == 0) u.status = rc2; return; }
        }
    }
    __threadfence_block();
    __syncthreads();
    MIC_STAMP_AT(u, 5);
    // ---- NCount header (one lane; fsecompressu16.go:191-289) ----------------------------------------
    int rc_par = MICD_ERR_UNSUPPORTED;
    if (u.blob_cap >= 6 + 16 && ((uintptr_t)u.blob & 3) == 0) {
        // scratch that tp_build only needs later: first_visit / cum_all hold the two per-symbol arrays; small alphabets
        // pack into visit_pos (LDS) and copy out, large ones pack straight into the blob (word-aligned 2 bytes ahead)
        uint32_t hdr = 0;
        if (sizeof(NormT) == 2)
            rc_par = tb_write_ncount_par(norm, symbol_len, tl, first_visit, cum_all, (uint32_t *)visit_pos, 0u, true, s_tmp,
                                         u.blob + 6, u.blob_cap - 6 - 8, &hdr);
        else
            rc_par = tb_write_ncount_par(norm, symbol_len, tl, first_visit, cum_all, (uint32_t *)(u.blob + 4), 16u, false, s_tmp,
                                         u.blob + 6, u.blob_cap - 6 - 8, &hdr);
        if (rc_par != MICD_ERR_UNSUPPORTED && tid == 0) { if (rc_par == MICD_OK) u.hdr_len = hdr; s_misc[3] = (uint32_t)rc_par; }
        __syncthreads();
    }
    if (rc_par != MICD_ERR_UNSUPPORTED) {
    } else if (tid == 0) {
        int rc = MICD_OK;
        uint32_t hdr = 0;
        if (u.blob_cap < 6 + 8) rc = MICD_ERR_CAPACITY;
        else rc = mic_write_ncount(norm, symbol_len, tl, u.blob + 6, u.blob_cap - 6, &hdr);
        if (rc == MICD_OK) {
            u.hdr_len = hdr;
            for (int k = 0; k < 8; k++) u.blob[6 + hdr + k] = 0;      // k_enc_tans_wg ORs into the first stream word
        }
        s_misc[3] = (uint32_t)rc;
    }
    __threadfence_block();
    __syncthreads();
    if ((int)s_misc[3] != MICD_OK) { if (tid == 0) u.status = (int)s_misc[3]; return; }
    MIC_STAMP_AT(u, 6);
    if (u.nstates == 108) {
        // buildRansEncTable (ransu16.go:139-180): bias = cumulative frequency, positives in symbol order, then the
        // low-probability symbols; record = freq | k0 << 20 in tt_nb, bias in tt_find.  No state table.
        uint32_t carry_pos = 0, carry_low = 0;
        for (uint32_t base = 0; base < symbol_len; base += TP_THREADS) {
            const uint32_t s2 = base + tid;
            const int32_t v = (s2 < symbol_len) ? (int32_t)norm[s2] : 0;
            uint32_t tot;
            const uint32_t ex = tp_block_excl((v > 0 ? (uint32_t)v : 0u) | ((v == -1 ? 1u : 0u) << 20), s_tmp, &tot);
            if (s2 < symbol_len) first_visit[s2] = (IdxT)(v == -1 ? carry_low + (ex >> 20) : carry_pos + (ex & 0xFFFFF));
            carry_pos += tot & 0xFFFFF; carry_low += tot >> 20;
        }
        __syncthreads();
        if (carry_pos + carry_low != size) { if (tid == 0) u.status = MICD_ERR_INTERNAL; return; }
        for (uint32_t s2 = tid; s2 < symbol_len; s2 += TP_THREADS) {
            const int32_t v = (int32_t)norm[s2];
            if (v == 0) continue;
            const uint32_t freq = v > 0 ? (uint32_t)v : 1u;
            const uint32_t k0 = tl - mic_high_bits(freq);
            u.tt_nb[s2] = freq | (k0 << 20);
            u.tt_find[s2] = (int32_t)(v > 0 ? (uint32_t)first_visit[s2] : carry_pos + (uint32_t)first_visit[s2]);
        }
        if (tid == 0) u.zero_bits = 0;
        return;
    }
    // ---- CTable: stateTable via the parallel spread, symbolTT per symbol ------------------------------
    TpScratch<NormT, IdxT> S;
    S.norm = norm; S.first_visit = first_visit; S.cum_all = cum_all; S.visit_pos = visit_pos;
    S.bitmap = bitmap; S.wprefix = wprefix; S.big_list = big_list; S.s_tmp = s_tmp;
    uint32_t *state_tab = u.state_tab;
    const int rc = tp_build(S, symbol_len, tl, [&](uint32_t p, uint32_t s, uint32_t r, uint32_t) {
        state_tab[(uint32_t)cum_all[s] + r] = size + p;                  // fsecompressu16.go:398-402
    });
    if (rc != MICD_OK) { if (tid == 0) u.status = rc; return; }
    uint32_t zb = 0;
    const uint32_t tlv = (tl << 16) - size;
    const int32_t large_limit = (int32_t)(size >> 1);
    for (uint32_t s = tid; s < symbol_len; s += TP_THREADS) {           // fsecompressu16.go:411-424
        const int32_t v = (int32_t)norm[s];
        if (v == 0) continue;
        const int32_t total = (int32_t)(uint32_t)cum_all[s];
        if (v > large_limit) zb = 1;
        if (v == -1 || v == 1) { u.tt_nb[s] = tlv; u.tt_find[s] = total - 1; }
        else {
            const uint32_t max_bits_out = tl - mic_high_bits((uint32_t)(v - 1));
            u.tt_nb[s] = (max_bits_out << 16) - ((uint32_t)v << max_bits_out);
            u.tt_find[s] = total - v;
        }
    }
    if (__syncthreads_or((int)zb)) { if (tid == 0) u.zero_bits = 1; }
    else if (tid == 0) u.zero_bits = 0;
    MIC_STAMP_AT(u, 7);
}

__global__ void __launch_bounds__(TP_THREADS) k_enc_tables_wg(MicUnit *units) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ uint32_t s_misc[8];
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *s_tmp = (uint32_t *)(s_raw + TB_OFF_TMP);
    // ---- histogram scan: symbolLen and maxCount (fsecompressu16.go:438-462) -------------------------
    uint32_t m = 0, sl = 0;
    const uint32_t hist_hi = min(u.tab_cap, (u.hist_hi >= 1 && u.hist_hi <= MIC_MAXSYM) ? u.hist_hi : MIC_MAXSYM + 1u);   // (the tokeniser's bound on what it counted)
    for (uint32_t i = tid; i < hist_hi; i += TP_THREADS) {
        const uint32_t c = u.hist[i];
        if (c) { m = max(m, c); sl = max(sl, i + 1); }
    }
    m = tb_wave_max(m); sl = tb_wave_max(sl);
    if (lane == 0) { s_tmp[wave] = m; s_tmp[16 + wave] = sl; }
    __syncthreads();
    if (tid == 0) {
        uint32_t mm = 0, ss = 0;
        for (int w = 0; w < TP_WAVES; w++) { mm = max(mm, s_tmp[w]); ss = max(ss, s_tmp[16 + w]); }
        const uint32_t n = u.ntok;
        u.max_count = mm; u.symbol_len = ss;
        int rc = MICD_OK;
        // gate order of FSECompressU16* (fse2state.go:23-42); the flavour's length gate is applied in k_enc_tans_wg
        // (a bare FSE call has no fallback chain: its own flavour's length gate comes before the histogram gates, fse8state.go:32-34)
        const uint32_t lanes0 = u.nstates == 108 ? 8u : (uint32_t)u.nstates;
        if (n <= 1 || (u.no_fallback && n <= lanes0 - 1)) rc = MICD_ERR_INCOMPRESSIBLE;
        else if (mm == n) rc = MICD_ERR_USE_RLE;
        else if (mm == 1 || mm < (n >> 15)) rc = MICD_ERR_INCOMPRESSIBLE;
        uint32_t tl = 0;
        if (rc == MICD_OK) { tl = mic_optimal_table_log(n, ss, u.req_tl); u.table_log = tl; }
        if (rc == MICD_OK && (1u << tl) > u.tab_cap) rc = MICD_INT_GROW;      // (tier 1: the table slabs hold 8192 states)
        if (rc != MICD_OK) u.status = rc;
        s_misc[0] = (uint32_t)rc; s_misc[1] = tl; s_misc[4] = ss; s_misc[5] = n;
    }
    __syncthreads();
    if ((int)s_misc[0] != MICD_OK) return;
    const uint32_t tl = s_misc[1], symbol_len = s_misc[4], n = s_misc[5];
    uint32_t *bitmap = (uint32_t *)(s_raw + TB_OFF_BITMAP), *wprefix = (uint32_t *)(s_raw + TB_OFF_WPREF), *big = (uint32_t *)(s_raw + TB_OFF_BIG);
    if (symbol_len <= TB_SMALL_SYMS && tl <= TB_SMALL_TL) {
        enc_tables_body<int16_t, uint16_t>(u, (int16_t *)(s_raw + TB_OFF_NORM), (uint16_t *)(s_raw + TB_OFF_FIRST),
                                           (uint16_t *)(s_raw + TB_OFF_CUM), (uint16_t *)(s_raw + TB_OFF_VISIT),
                                           bitmap, wprefix, big, s_tmp, s_misc, n, symbol_len, tl);
    } else {
        // HBM scratch: norm[] itself, cumul[] for the first visits, hist[] (dead after normalisation) for cum_all,
        // tab_sym[] for the visit sequence
        enc_tables_body<int32_t, uint32_t>(u, u.norm, (uint32_t *)u.cumul, (uint32_t *)u.hist, u.tab_sym,
                                           bitmap, wprefix, big, s_tmp, s_misc, n, symbol_len, tl);
    }
}

// ------------------------------------------------------------------------------------------
#define DT_STAGE 8192     // bytes of the blob staged in LDS for the header parse (aliases the visit/bitmap area)

template <typename NormT, typename IdxT>
__device__ void dec_tables_body(MicUnit &u, NormT *norm, IdxT *first_visit, IdxT *cum_all, uint16_t *visit_pos,
                                uint32_t *bitmap, uint32_t *wprefix, uint32_t *big_list, uint32_t *s_tmp,
                                uint32_t symbol_len, uint32_t tl, uint32_t flavour) {
    const uint32_t tid = threadIdx.x;
    const uint32_t size = 1u << tl;
    uint32_t *dt = u.tt_nb; uint16_t *ds = u.tab_sym;
    uint32_t bad = 0, zb = 0;
    const int32_t large_limit = (int32_t)(size >> 1);
    for (uint32_t s = tid; s < symbol_len; s += TP_THREADS) if ((int32_t)norm[s] >= large_limit) zb = 1;   // fsedecompressu16.go:214-216
    if (flavour == 108) {
        // buildRansDecTable (ransu16.go:77-135): slots filled in symbol order, positives first, then the -1 symbols
        uint32_t carry_pos = 0, carry_low = 0;
        for (uint32_t base = 0; base < symbol_len; base += TP_THREADS) {
            const uint32_t s = base + tid;
            const int32_t v = (s < symbol_len) ? (int32_t)norm[s] : 0;
            uint32_t tot;
            const uint32_t ex = tp_block_excl((v > 0 ? (uint32_t)v : 0u) | ((v == -1 ? 1u : 0u) << 20), s_tmp, &tot);
            if (s < symbol_len) { first_visit[s] = (IdxT)(v == -1 ? carry_low + (ex >> 20) : carry_pos + (ex & 0xFFFFF)); }
            carry_pos += tot & 0xFFFFF; carry_low += tot >> 20;
        }
        __syncthreads();
        if (carry_pos + carry_low != size) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
        for (uint32_t s = tid; s < symbol_len; s += TP_THREADS) {
            const int32_t v = (int32_t)norm[s];
            if (v == -1) { const uint32_t slot = carry_pos + (uint32_t)first_visit[s]; ds[slot] = (uint16_t)s; dt[slot] = 0u | (tl << 16); }
        }
        // positives: one thread per symbol walks its slots (long runs are rare and short relative to the table)
        for (uint32_t s = tid; s < symbol_len; s += TP_THREADS) {
            const int32_t v = (int32_t)norm[s];
            if (v <= 0) continue;
            const uint32_t a = (uint32_t)first_visit[s];
            for (uint32_t j = 0; j < (uint32_t)v; j++) {
                const uint32_t x_next = (uint32_t)v + j;
                const uint32_t nb = tl - mic_high_bits(x_next);
                ds[a + j] = (uint16_t)s; dt[a + j] = ((x_next << nb) - size) | (nb << 16);
            }
        }
    } else {
        TpScratch<NormT, IdxT> S;
        S.norm = norm; S.first_visit = first_visit; S.cum_all = cum_all; S.visit_pos = visit_pos;
        S.bitmap = bitmap; S.wprefix = wprefix; S.big_list = big_list; S.s_tmp = s_tmp;
        const int rc = tp_build(S, symbol_len, tl, [&](uint32_t p, uint32_t s, uint32_t r, uint32_t slots) {
            const uint32_t next = slots + r;                              // symbolNext[s]++ in table order, :244-246
            const uint32_t nb = tl - mic_high_bits(next);
            const uint32_t ns = (next << nb) - size;
            if (ns >= size || (ns == p && nb == 0)) bad = 1;              // :250-256
            ds[p] = (uint16_t)s; dt[p] = ns | (nb << 16);
        });
        if (rc != MICD_OK) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    }
    const int anybad = __syncthreads_or((int)bad);
    const int anyzb = __syncthreads_or((int)zb);
    if (tid == 0) { if (anybad) u.status = MICD_ERR_CORRUPT; u.zero_bits = anyzb ? 1u : 0u; }
}

// Stream prefix + NCount parse (FSEDecompressU16Auto, fse2state.go:102-116; readNCount, fsedecompressu16.go:48-167).
// The parse is a serial bit reader, so it gets a kernel of its own with one wave per unit: all units of a
// launch parse side by side instead of one after the other under a 1024-thread group.  The head of the
// blob is staged in LDS (the reader hops byte-wise); counts go to the unit's HBM norm[] slab.
// Two classes: BIG = false stages 8 KiB and handles alphabets up to 8192 symbols (every depth <= 13: many groups per
// CU, 32 KiB of counts to clear); what it cannot take -- a longer header, a larger alphabet: 16-bit-depth frames --
// it leaves marked for BIG = true (40 KiB stage, all 65536 counts cleared).
#define DP_DEFER 0xDEFE7u      // in u.flavour: parse left to the BIG class
template <bool BIG>
__global__ void __launch_bounds__(64) k_dec_parse(MicUnit *units) {
    constexpr uint32_t DP_STAGE = BIG ? 40u * 1024u : 8u * 1024u;
    constexpr uint32_t DP_SYMS = BIG ? 65536u : 8192u;
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DP_STAGE + 16];
    MicUnit &u = units[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint32_t len = u.comp_len;
    if (BIG && u.flavour != DP_DEFER) return;
    // the counts array starts out zero (the zero-runs of the header then cost nothing but a cursor move)
    if (u.norm) { uint4 *z = (uint4 *)u.norm; for (uint32_t i = tid; i < DP_SYMS / 4; i += 64) z[i] = make_uint4(0, 0, 0, 0); }
    if (u.comp_in) {
        const uint32_t nst = min(len, (uint32_t)DP_STAGE);
        const uint32_t head = (uint32_t)((16 - ((uintptr_t)u.comp_in & 15)) & 15);          // bytes up to 16-byte alignment
        for (uint32_t i = tid; i < min(head, nst); i += 64) s_in[i] = u.comp_in[i];
        if (nst > head) {
            const uint32_t nvec = (nst - head) / 16;
            typedef uint32_t v4 __attribute__((ext_vector_type(4)));
            typedef v4 v4u __attribute__((aligned(1)));
            for (uint32_t i = tid; i < nvec; i += 64) *(v4u *)(s_in + head + i * 16) = *(const v4 *)(u.comp_in + head + i * 16);
            for (uint32_t i = head + nvec * 16 + tid; i < nst; i += 64) s_in[i] = u.comp_in[i];
        }
    }
    __syncthreads();
    if (tid == 0) {
        u.status = MICD_OK; u.ntok = 0;
        int rc = MICD_OK;
        uint32_t flavour = 1, count = 0, off = 0, used = 0, symbol_len = 0, tl = 0;
        do {
            if ((u.mode == 0 && (u.w <= 0 || u.h <= 0)) || !u.comp_in) { rc = MICD_ERR_ARGS; break; }
            if (len >= 2 && s_in[0] == 0xFF) {
                if (s_in[1] == 0x84) flavour = 8;
                else if (s_in[1] == 0x08) flavour = 108;
                else if (s_in[1] == 0x04) flavour = 4;
                else if (s_in[1] == 0x02) flavour = 2;
            }
            // a WaveletV2 stream is decoded by FSEDecompressU16FourState and nothing else (waveletfsecompressu16.go:504): on the
            // device-resident path nobody has looked at the magic bytes yet
            if (u.walk_mode == 1 && flavour != 4) { rc = MICD_ERR_CORRUPT; break; }
            if (flavour != 1) {
                if (len < 6) { rc = MICD_ERR_CORRUPT; break; }
                count = (uint32_t)s_in[2] | ((uint32_t)s_in[3] << 8) | ((uint32_t)s_in[4] << 16) | ((uint32_t)s_in[5] << 24);
                off = 6;
                if (count > u.tok_cap) { rc = u.tier == 1 ? MICD_INT_GROW : MICD_ERR_CORRUPT; break; }   // (tier 1: a small token slab; tier 2 decides)
            }
            // (a table of more than 8192 states is the BIG class's whatever its alphabet -- the table slabs of this class's tier hold
            // 8192: a WaveletV2 stream, tableLog 16, was parsed half way here, 8192 of its ~16 k symbols, before it was handed over)
            if (!BIG && off < len && (uint32_t)(s_in[off] & 0xF) + MIC_MIN_TABLELOG > 13u) { rc = MICD_ERR_UNSUPPORTED; break; }
            // Parse from the staged bytes when that is certain to be identical: the whole blob is staged, or the
            // header ends well inside the stage; otherwise from HBM.
            // (the window reader may look 8 bytes past the bytes it is given: the stage is DP_STAGE + 16 bytes long)
            if (len <= DP_STAGE) rc = mic_read_ncount<int32_t, true>(s_in + off, len - off, u.norm, &symbol_len, &tl, &used, DP_SYMS, (const uint32_t *)s_in, off);
            else {
                rc = mic_read_ncount<int32_t, true>(s_in + off, DP_STAGE - off, u.norm, &symbol_len, &tl, &used, DP_SYMS, (const uint32_t *)s_in, off);
                if (!(rc == MICD_OK && used + 8 < DP_STAGE - off)) {
                    if (!BIG) rc = MICD_ERR_UNSUPPORTED;
                    else {
                        for (uint32_t i = 0; i < 65536; i++) u.norm[i] = 0;    // (rare: header longer than the stage) start over from HBM
                        rc = mic_read_ncount<int32_t, true>(u.comp_in + off, len - off, u.norm, &symbol_len, &tl, &used, 65536u);
                    }
                }
            }
        } while (0);
        if (!BIG && rc == MICD_ERR_UNSUPPORTED) {
            if (u.tier == 1) { u.status = MICD_INT_GROW; return; }            // (more than 8192 symbols: the count slab is tier 1's)
            u.flavour = DP_DEFER; return;
        }
        if (rc == MICD_OK && (1u << tl) > u.tab_cap) rc = MICD_INT_GROW;       // (tier 1: the decode table slabs hold 8192 states)
        if (rc == MICD_OK) { u.flavour = flavour; u.count = count; u.symbol_len = symbol_len; u.table_log = tl; u.bits_off = off + used; }
        else u.status = rc;
    }
}

__global__ void __launch_bounds__(TP_THREADS) k_dec_tables_wg(MicUnit *units) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    MicUnit &u = units[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    if (u.status != MICD_OK) return;                                      // k_dec_parse already failed this unit
    int16_t *norm16 = (int16_t *)(s_raw + TB_OFF_NORM);
    uint32_t *s_tmp = (uint32_t *)(s_raw + TB_OFF_TMP);
    const uint32_t tl = u.table_log, symbol_len = u.symbol_len, flavour = u.flavour;
    const bool small = symbol_len <= TB_SMALL_SYMS && tl <= TB_SMALL_TL;
    if (small) for (uint32_t s2 = tid; s2 < symbol_len; s2 += TP_THREADS) norm16[s2] = (int16_t)u.norm[s2];
    __syncthreads();
    uint32_t *bitmap = (uint32_t *)(s_raw + TB_OFF_BITMAP), *wprefix = (uint32_t *)(s_raw + TB_OFF_WPREF), *big = (uint32_t *)(s_raw + TB_OFF_BIG);
    if (small) {
        dec_tables_body<int16_t, uint16_t>(u, norm16, (uint16_t *)(s_raw + TB_OFF_FIRST), (uint16_t *)(s_raw + TB_OFF_CUM),
                                           (uint16_t *)(s_raw + TB_OFF_VISIT), bitmap, wprefix, big, s_tmp, symbol_len, tl, flavour);
    } else {
        // HBM scratch: cumul[] first visits, tt_find[] cum_all (not hist[]: the encode side keeps that slab zero between calls),
        // state_tab[] (u32, first half used as u16) visit sequence
        dec_tables_body<int32_t, uint32_t>(u, u.norm, (uint32_t *)u.cumul, (uint32_t *)u.tt_find, (uint16_t *)u.state_tab,
                                           bitmap, wprefix, big, s_tmp, symbol_len, tl, flavour);
    }
}

void mic_launch_enc_tables(MicUnit *d_units, int n, hipStream_t stream) {
    static MicPerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute((const void *)k_enc_tables_wg, hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS_BYTES); });
    hipLaunchKernelGGL(k_enc_tables_wg, dim3(n), dim3(TP_THREADS), TB_LDS_BYTES, stream, d_units);
}
void mic_launch_dec_tables(MicUnit *d_units, int n, hipStream_t stream) {
    static MicPerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute((const void *)k_dec_tables_wg, hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS_BYTES); });
    hipLaunchKernelGGL(k_dec_parse<false>, dim3(n), dim3(64), 0, stream, d_units);
    hipLaunchKernelGGL(k_dec_parse<true>, dim3(n), dim3(64), 0, stream, d_units);
    hipLaunchKernelGGL(k_dec_tables_wg, dim3(n), dim3(TP_THREADS), TB_LDS_BYTES, stream, d_units);
}
