// mic_fse_tables.h -- per-unit FSE table construction on the device.
//
// These routines are the integer-exact statement of the reference's table maths
// (fsecompressu16.go:465-667, :191-289, :329-431; fsedecompressu16.go:48-263).  They are
// O(symbolLen + 2^tableLog) per unit and data-dependent in their corrections
// (SURVEY.md "Hard parts": "do not improve it"), so each unit's tables are built by one
// work-group; the single-lane sections below are marked and are the ones a later round
// may spread over the wave (scan-based spread / NCount packing).
#pragma once
#include "mic_dev.h"

// fsecompressu16.go:480-518 (optimalTableLog) with minTableLog (:465-472) inlined.
// req = ScratchU16.TableLog as set by the caller (fseu16.go:101-102; 0 = defaultTablelog, :133-135)
__device__ inline uint32_t mic_optimal_table_log(uint32_t n, uint32_t symbol_len, uint32_t req = 0) {
    uint8_t table_log = req ? (uint8_t)req : (uint8_t)MIC_DEF_TABLELOG;
    uint32_t min_bits_src = mic_high_bits(n - 1) + 1;
    uint32_t min_bits_sym = mic_high_bits(symbol_len - 1) + 2;
    uint8_t min_bits = (uint8_t)(min_bits_src < min_bits_sym ? min_bits_src : min_bits_sym);
    uint8_t max_bits_src = (uint8_t)((uint8_t)mic_high_bits(n - 1) - 2);
    if (max_bits_src < table_log) table_log = max_bits_src;
    if (min_bits > table_log) table_log = min_bits;
    uint32_t density = n / symbol_len;
    if (symbol_len > 512 && density > 16 && table_log < 13) table_log = 13;
    else if (density > 64 && symbol_len > 256 && table_log < 12) table_log = 12;
    else if (density > 32 && symbol_len > 128 && table_log < 12) table_log = 12;
    if (max_bits_src < table_log) table_log = max_bits_src;
    if (table_log < MIC_MIN_TABLELOG) table_log = MIC_MIN_TABLELOG;
    if (table_log > MIC_MAX_TABLELOG) table_log = MIC_MAX_TABLELOG;
    return table_log;
}

// fsecompressu16.go:575-667 (normalizeCount2); single lane.
__device__ inline int mic_normalize_count2(const uint32_t *count, int32_t *norm, uint32_t symbol_len,
                                           uint32_t n, uint32_t tl) {
    const int32_t not_yet = -2;
    uint32_t distributed = 0, total = n;
    uint32_t low_threshold = total >> tl;
    uint32_t low_one = (total * 3) >> (tl + 1);
    for (uint32_t i = 0; i < symbol_len; i++) {
        uint32_t cnt = count[i];
        if (cnt == 0) { norm[i] = 0; continue; }
        if (cnt <= low_threshold) { norm[i] = -1; distributed++; total -= cnt; continue; }
        if (cnt <= low_one) { norm[i] = 1; distributed++; total -= cnt; continue; }
        norm[i] = not_yet;
    }
    // distributed >= 2^tl: the reference divides by zero (==) or wraps to_distribute and spins
    // forever in the total==0 loop (>) -- there is no behaviour to match; report it.
    if (distributed >= (1u << tl)) return MICD_ERR_INTERNAL;
    uint32_t to_distribute = (1u << tl) - distributed;
    if ((total / to_distribute) > low_one) {
        low_one = (total * 3) / (to_distribute * 2);
        for (uint32_t i = 0; i < symbol_len; i++) {
            uint32_t cnt = count[i];
            if (norm[i] == not_yet && cnt <= low_one) { norm[i] = 1; distributed++; total -= cnt; }
        }
        if (distributed >= (1u << tl)) return MICD_ERR_INTERNAL;
        to_distribute = (1u << tl) - distributed;
    }
    if (distributed == symbol_len + 1) {
        uint32_t max_v = 0, max_c = 0;
        for (uint32_t i = 0; i < symbol_len; i++)
            if (count[i] > max_c) { max_v = i; max_c = count[i]; }
        norm[max_v] += (int32_t)to_distribute;
        return MICD_OK;
    }
    if (total == 0) {
        bool any = false;
        for (uint32_t i = 0; i < symbol_len; i++) if (norm[i] > 0) { any = true; break; }
        if (!any) return MICD_ERR_INTERNAL;       // reference: endless loop
        for (uint32_t i = 0; to_distribute > 0; i = (i + 1) % symbol_len)
            if (norm[i] > 0) { to_distribute--; norm[i]++; }
        return MICD_OK;
    }
    uint64_t v_step_log = 62 - (uint64_t)tl;
    uint64_t mid = (1ull << (v_step_log - 1)) - 1;
    uint64_t r_step = (((1ull << v_step_log) * (uint64_t)to_distribute) + mid) / (uint64_t)total;
    uint64_t tmp_total = mid;
    for (uint32_t i = 0; i < symbol_len; i++) {
        if (norm[i] == not_yet) {
            uint64_t end = tmp_total + (uint64_t)count[i] * r_step;
            uint32_t weight = (uint32_t)(end >> v_step_log) - (uint32_t)(tmp_total >> v_step_log);
            if (weight < 1) return MICD_ERR_INTERNAL;
            norm[i] = (int32_t)weight;
            tmp_total = end;
        }
    }
    return MICD_OK;
}

__device__ __constant__ uint32_t mic_rtb_table[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };

// fsecompressu16.go:524-571 (normalizeCount); single lane.
__device__ inline int mic_normalize_count(const uint32_t *count, int32_t *norm, uint32_t symbol_len,
                                          uint32_t n, uint32_t tl) {
    uint64_t scale = 62 - (uint64_t)tl;
    uint64_t step = (1ull << 62) / (uint64_t)n;
    uint64_t v_step = 1ull << (scale - 20);
    int32_t still = (int32_t)(1u << tl);
    uint32_t largest = 0;
    int32_t largest_p = 0;
    uint32_t low_threshold = n >> tl;
    for (uint32_t i = 0; i < symbol_len; i++) {
        uint32_t cnt = count[i];
        if (cnt == 0) { norm[i] = 0; continue; }
        if (cnt <= low_threshold) { norm[i] = -1; still--; continue; }
        int32_t proba = (int32_t)(((uint64_t)cnt * step) >> scale);
        if (proba < 8) {
            uint64_t rest_to_beat = v_step * (uint64_t)mic_rtb_table[proba];
            uint64_t v = (uint64_t)cnt * step - ((uint64_t)proba << scale);
            if (v > rest_to_beat) proba++;
        }
        if (proba > largest_p) { largest_p = proba; largest = i; }
        norm[i] = proba;
        still -= proba;
    }
    if (-still >= (norm[largest] >> 1)) return mic_normalize_count2(count, norm, symbol_len, n, tl);
    norm[largest] += still;
    return MICD_OK;
}

// fsecompressu16.go:191-289 (writeCount); single lane.  Writes at out[0..), returns the
// header length through *hdr_len.  cap must cover ((symbol_len*tl)>>3)+3+2 bytes.
template <typename NormT>
__device__ inline int mic_write_ncount(const NormT *norm, uint32_t symbol_len, uint32_t tl,
                                       uint8_t *out, uint32_t cap, uint32_t *hdr_len) {
    int table_size = 1 << tl;
    bool previous0 = false;
    uint32_t charnum = 0;
    uint32_t max_header = ((symbol_len * tl) >> 3) + 3;
    if (cap < max_header + 2) return MICD_ERR_CAPACITY;
    uint32_t bit_stream = tl - MIC_MIN_TABLELOG;
    uint32_t bit_count = 4;
    int32_t remaining = table_size + 1;
    int32_t threshold = table_size;
    uint32_t nb_bits = tl + 1;
    uint32_t out_p = 0;
    while (remaining > 1) {
        if (previous0) {
            uint32_t start = charnum;
            while (norm[charnum] == 0) {
                charnum++;
                if (charnum > MIC_MAXSYM) return MICD_ERR_INTERNAL;
            }
            while (charnum >= start + 24) {
                start += 24;
                bit_stream += 0xFFFFu << bit_count;
                out[out_p] = (uint8_t)bit_stream; out[out_p + 1] = (uint8_t)(bit_stream >> 8);
                out_p += 2;
                bit_stream >>= 16;
            }
            while (charnum >= start + 3) {
                start += 3;
                bit_stream += 3u << bit_count;
                bit_count += 2;
            }
            bit_stream += (charnum - start) << bit_count;
            bit_count += 2;
            if (bit_count > 16) {
                out[out_p] = (uint8_t)bit_stream; out[out_p + 1] = (uint8_t)(bit_stream >> 8);
                out_p += 2;
                bit_stream >>= 16;
                bit_count -= 16;
            }
        }
        int32_t count = (int32_t)norm[charnum];
        charnum++;
        int32_t max = (2 * threshold - 1) - remaining;
        if (count < 0) remaining += count; else remaining -= count;
        count++;
        if (count >= threshold) count += max;
        bit_stream += (uint32_t)count << bit_count;
        bit_count += nb_bits;
        if (count < max) bit_count--;
        previous0 = (count == 1);
        if (remaining < 1) return MICD_ERR_INTERNAL;
        while (remaining < threshold) { nb_bits--; threshold >>= 1; }
        if (bit_count > 16) {
            out[out_p] = (uint8_t)bit_stream; out[out_p + 1] = (uint8_t)(bit_stream >> 8);
            out_p += 2;
            bit_stream >>= 16;
            bit_count -= 16;
        }
    }
    out[out_p] = (uint8_t)bit_stream;
    out[out_p + 1] = (uint8_t)(bit_stream >> 8);
    out_p += (bit_count + 7) / 8;
    if (charnum > symbol_len) return MICD_ERR_INTERNAL;
    *hdr_len = out_p;
    return MICD_OK;
}

// fsecompressu16.go:329-431 (buildCTable); single lane.
__device__ inline int mic_build_ctable(MicUnit &u) {
    const int32_t *norm = u.norm;
    uint32_t tl = u.table_log, symbol_len = u.symbol_len;
    uint32_t table_size = 1u << tl;
    uint32_t high_threshold = table_size - 1;
    int32_t *cumul = u.cumul;
    uint16_t *table_symbol = u.tab_sym;
    cumul[0] = 0;
    for (uint32_t s = 0; s < symbol_len; s++) {
        int32_t v = norm[s];
        if (v == -1) {
            cumul[s + 1] = cumul[s] + 1;
            table_symbol[high_threshold] = (uint16_t)s;
            high_threshold--;
        } else {
            cumul[s + 1] = cumul[s] + v;
        }
    }
    if ((uint32_t)cumul[symbol_len] != table_size) return MICD_ERR_INTERNAL;
    cumul[symbol_len] = (int32_t)table_size + 1;
    uint32_t zero_bits = 0;
    {
        uint32_t step = mic_table_step(table_size), mask = table_size - 1, position = 0;
        int32_t large_limit = (int32_t)(1u << (tl - 1));
        for (uint32_t s = 0; s < symbol_len; s++) {
            int32_t v = norm[s];
            if (v > large_limit) zero_bits = 1;
            for (int32_t k = 0; k < v; k++) {
                table_symbol[position] = (uint16_t)s;
                position = (position + step) & mask;
                while (position > high_threshold) position = (position + step) & mask;
            }
        }
        if (position != 0) return MICD_ERR_INTERNAL;
    }
    u.zero_bits = zero_bits;
    for (uint32_t p = 0; p < table_size; p++) {
        uint16_t v = table_symbol[p];
        u.state_tab[cumul[v]] = table_size + p;
        cumul[v]++;
    }
    int32_t total = 0;
    uint32_t tlv = (tl << 16) - (1u << tl);
    for (uint32_t s = 0; s < symbol_len; s++) {
        int32_t v = norm[s];
        if (v == 0) continue;
        if (v == -1 || v == 1) {
            u.tt_nb[s] = tlv;
            u.tt_find[s] = total - 1;
            total++;
        } else {
            uint32_t max_bits_out = tl - mic_high_bits((uint32_t)(v - 1));
            uint32_t min_state_plus = (uint32_t)v << max_bits_out;
            u.tt_nb[s] = (max_bits_out << 16) - min_state_plus;
            u.tt_find[s] = total - v;
            total += v;
        }
    }
    if (total != (int32_t)table_size) return MICD_ERR_INTERNAL;
    return MICD_OK;
}

// byteReader.Uint32 (bytereader.go:31-40) with an explicit bound (Go would panic).
__device__ inline uint32_t mic_rd_u32(const uint8_t *b, uint32_t len, int64_t off, int *err) {
    if (off < 0 || (uint64_t)off + 4 > len) { *err = 1; return 0; }
    return (uint32_t)b[off] | ((uint32_t)b[off + 1] << 8) | ((uint32_t)b[off + 2] << 16) | ((uint32_t)b[off + 3] << 24);
}

// fsedecompressu16.go:48-167 (readNCount); single lane.  b/len = stream after the prefix.
// norm_cap: entries available in norm[]; MICD_ERR_UNSUPPORTED when the alphabet is larger (the
// caller then parses again into the full 65536-entry array).
// PREZEROED: norm[] is all zero on entry, so zero-runs only move the symbol cursor.
// dw / boff (optional): the same bytes as aligned dwords, b = (const uint8_t *)dw + boff, readable up to b + len + 8 -- an LDS
// stage.  The bulk of the header is then read through a 64-bit register window (one LDS dword per 32 bits consumed instead of a
// byte-wise u32 per field): far from the end of the buffer the reference's reader state is a function of the bit position alone
// (off = P >> 3, bitCount = P & 7, bitStream = the bits from P on), so the window reader is the same machine; the last sixteen
// bytes -- where the reference's refill rules differ, fsedecompressu16.go:100-160 -- are left to the byte-wise loop below, which
// takes over from the canonical state.
template <typename NormT, bool PREZEROED = false>
__device__ inline int mic_read_ncount(const uint8_t *b, uint32_t len, NormT *norm,
                                      uint32_t *symbol_len_out, uint32_t *tl_out, uint32_t *consumed,
                                      uint32_t norm_cap, const uint32_t *dw = nullptr, uint32_t boff = 0) {
    uint32_t charnum = 0;
    bool previous0 = false;
    int err = 0;
    int64_t iend = (int64_t)len, off = 0;
    if (iend < 4) return MICD_ERR_CORRUPT;
    uint32_t bit_stream = mic_rd_u32(b, len, 0, &err);
    uint32_t nb_bits = (bit_stream & 0xF) + MIC_MIN_TABLELOG;
    if (nb_bits > MIC_MAX_TABLELOG) return MICD_ERR_CORRUPT; // reference allows 17 then fails to build
    bit_stream >>= 4;
    uint32_t bit_count = 4;
    uint32_t tl = nb_bits;
    int32_t remaining = (int32_t)((1u << nb_bits) + 1);
    int32_t threshold = (int32_t)(1u << nb_bits);
    int32_t got_total = 0;
    nb_bits++;
    if (dw && iend >= 24 && iend < (1 << 27)) {                            // (an LDS stage: bit positions fit 32 bits -- half the instructions of 64-bit ones)
        uint32_t P = 4;                                                   // bit position, relative to b
        const uint32_t lim = ((uint32_t)iend - 16) * 8;                   // an iteration starts (and a zero-run skip lands) at P <= lim only
        uint32_t wi = 0xFFFFFFFEu; uint64_t W = 0;                        // window: dwords wi, wi + 1 (none yet)
        auto snap = [&](uint32_t p) -> uint32_t {
            const uint32_t pa = p + 8u * boff;
            const uint32_t i = pa >> 5;
            if (i != wi) {
                if (i == wi + 1) W = (W >> 32) | ((uint64_t)dw[i + 1] << 32);
                else W = (uint64_t)dw[i] | ((uint64_t)dw[i + 1] << 32);
                wi = i;
            }
            return (uint32_t)(W >> (pa & 31u));
        };
        // The loop is written for the scalar unit it runs on (one lane's uniform work is scalarised by the compiler): a TAKEN branch
        // costs it tens of cycles, and the first form -- early returns inside nested ifs -- compiled to a dozen of them per symbol
        // (550 cycles a symbol: the parse was 0.95 ms of a 33 ms step).  Selects instead of branches, one exit, errors in `fail`.
        int fail = 0;
        // (with the LDS stage the counts are in HBM: stores through a global-address-space pointer -- a FLAT store also counts on
        // lgkmcnt, and the window refill's wait for its LDS read would wait for the store in front of it too)
        const mic_gp<NormT> gnorm = mic_g(norm);
        while (remaining > 1 && P <= lim) {
            const uint32_t P0 = P;
            uint32_t bs = snap(P);
            if (previous0) {
                uint32_t n0 = charnum;
                bool bail = false;
                while ((bs & 0xFFFF) == 0xFFFF) {
                    n0 += 24; P += 16;
                    if (P > lim) { bail = true; break; }
                    bs = snap(P);
                    if (n0 > MIC_MAXSYM + 24) { fail = MICD_ERR_CORRUPT; break; }
                }
                if (fail) break;
                if (bail) { P = P0; break; }                              // (this iteration again, byte-wise)
                while ((bs & 3) == 3) { n0 += 3; bs >>= 2; P += 2; }
                n0 += bs & 3; P += 2;
                if (n0 > MIC_MAXSYM || n0 > norm_cap) { fail = n0 > MIC_MAXSYM ? MICD_ERR_CORRUPT : MICD_ERR_UNSUPPORTED; break; }
                if (PREZEROED) charnum = max(charnum, n0);
                else while (charnum < n0) { gnorm[charnum & 0xffff] = 0; charnum++; }
                bs = snap(P);
            }
            const int32_t max = (2 * threshold - 1) - remaining;
            const int32_t lowv = (int32_t)bs & (threshold - 1), fullv = (int32_t)bs & (2 * threshold - 1);
            const bool shortf = lowv < max;                               // the field is one bit shorter
            int32_t count = shortf ? lowv : (fullv >= threshold ? fullv - max : fullv);
            P += shortf ? nb_bits - 1 : nb_bits;
            count--;
            const int32_t ac = count < 0 ? -count : count;               // (count = -1 is a symbol of probability "less than one": it takes one slot)
            remaining -= ac; got_total += ac;
            if (charnum > MIC_MAXSYM || charnum >= norm_cap) { fail = charnum > MIC_MAXSYM ? MICD_ERR_CORRUPT : MICD_ERR_UNSUPPORTED; break; }
            gnorm[charnum & 0xffff] = (NormT)count;
            charnum++;
            previous0 = (count == 0);
            if (remaining < threshold) {
                if (remaining >= 1) {                                     // threshold = the power of two at or below `remaining`, nbBits with it
                    const int32_t k = 31 - (int32_t)__builtin_clz((uint32_t)remaining);
                    nb_bits -= (uint32_t)((31 - (int32_t)__builtin_clz((uint32_t)threshold)) - k);
                    threshold = 1 << k;
                } else while (remaining < threshold) {                    // (a damaged header: the reference's loop as it stands)
                    nb_bits--; threshold >>= 1;
                    if (threshold == 0) break;
                }
            }
        }
        if (fail) return fail;
        off = (int64_t)(P >> 3); bit_count = (uint32_t)P & 7u;            // the canonical state of the byte-wise reader
        bit_stream = mic_rd_u32(b, len, off, &err) >> bit_count;
        if (err) return MICD_ERR_CORRUPT;
    }
    while (remaining > 1) {
        if (previous0) {
            uint32_t n0 = charnum;
            while ((bit_stream & 0xFFFF) == 0xFFFF) {
                n0 += 24;
                if (off < iend - 5) {
                    off += 2;
                    bit_stream = mic_rd_u32(b, len, off, &err) >> bit_count;
                } else {
                    bit_stream >>= 16;
                    bit_count += 16;
                }
                if (n0 > MIC_MAXSYM + 24) return MICD_ERR_CORRUPT;
            }
            while ((bit_stream & 3) == 3) { n0 += 3; bit_stream >>= 2; bit_count += 2; }
            n0 += bit_stream & 3;
            bit_count += 2;
            if (n0 > MIC_MAXSYM) return MICD_ERR_CORRUPT;
            if (n0 > norm_cap) return MICD_ERR_UNSUPPORTED;
            if (PREZEROED) charnum = max(charnum, n0);
            else while (charnum < n0) { norm[charnum & 0xffff] = 0; charnum++; }
            if (off <= iend - 7 || off + (int64_t)(bit_count >> 3) <= iend - 4) {
                off += (int64_t)(bit_count >> 3);
                bit_count &= 7;
                bit_stream = mic_rd_u32(b, len, off, &err) >> bit_count;
            } else {
                bit_stream >>= 2;
            }
        }
        int32_t max = (2 * threshold - 1) - remaining;
        int32_t count;
        if (((int32_t)bit_stream & (threshold - 1)) < max) {
            count = (int32_t)bit_stream & (threshold - 1);
            bit_count += nb_bits - 1;
        } else {
            count = (int32_t)bit_stream & (2 * threshold - 1);
            if (count >= threshold) count -= max;
            bit_count += nb_bits;
        }
        count--;
        if (count < 0) { remaining += count; got_total -= count; }
        else { remaining -= count; got_total += count; }
        if (charnum > MIC_MAXSYM) return MICD_ERR_CORRUPT;
        if (charnum >= norm_cap) return MICD_ERR_UNSUPPORTED;
        norm[charnum & 0xffff] = (NormT)count;
        charnum++;
        previous0 = (count == 0);
        while (remaining < threshold) {
            nb_bits--; threshold >>= 1;
            if (threshold == 0) break;
        }
        if (off <= iend - 7 || off + (int64_t)(bit_count >> 3) <= iend - 4) {
            off += (int64_t)(bit_count >> 3);
            bit_count &= 7;
        } else {
            bit_count -= (uint32_t)(8 * (iend - 4 - off));
            off = iend - 4;
        }
        bit_stream = mic_rd_u32(b, len, off, &err) >> (bit_count & 31);
        if (err) return MICD_ERR_CORRUPT;
    }
    if (charnum <= 1 || charnum > MIC_MAXSYM + 1) return MICD_ERR_CORRUPT;
    if (remaining != 1) return MICD_ERR_CORRUPT;
    if (bit_count > 32) return MICD_ERR_CORRUPT;
    if (got_total != (int32_t)(1u << tl)) return MICD_ERR_CORRUPT;
    off += (int64_t)((bit_count + 7) >> 3);
    if (off > iend) return MICD_ERR_CORRUPT;
    *symbol_len_out = charnum; *tl_out = tl; *consumed = (uint32_t)off;
    return MICD_OK;
}

// fsedecompressu16.go:198-263 (buildDtable); single lane.  Writes
//   u.tab_sym[state]  = symbol
//   u.tt_nb[state]    = newState | nbBits << 16          (newState < 2^16, nbBits <= 16)
__device__ inline int mic_build_dtable(MicUnit &u) {
    const int32_t *norm = u.norm;
    uint32_t tl = u.table_log, symbol_len = u.symbol_len;
    uint32_t table_size = 1u << tl;
    uint32_t high_threshold = table_size - 1;
    uint32_t *symbol_next = (uint32_t *)u.tt_find;
    uint32_t zero_bits = 0;
    int32_t large_limit = (int32_t)(1u << (tl - 1));
    for (uint32_t i = 0; i < symbol_len; i++) {
        int32_t v = norm[i];
        if (v == -1) {
            u.tab_sym[high_threshold] = (uint16_t)i;
            high_threshold--;
            symbol_next[i] = 1;
        } else {
            if (v >= large_limit) zero_bits = 1;
            symbol_next[i] = (uint32_t)v;
        }
    }
    u.zero_bits = zero_bits;
    {
        uint32_t mask = table_size - 1, step = mic_table_step(table_size), position = 0;
        for (uint32_t s = 0; s < symbol_len; s++) {
            int32_t v = norm[s];
            for (int32_t i = 0; i < v; i++) {
                u.tab_sym[position] = (uint16_t)s;
                position = (position + step) & mask;
                while (position > high_threshold) position = (position + step) & mask;
            }
        }
        if (position != 0) return MICD_ERR_CORRUPT;
    }
    for (uint32_t p = 0; p < table_size; p++) {
        uint16_t symbol = u.tab_sym[p];
        uint32_t next_state = symbol_next[symbol];
        symbol_next[symbol] = next_state + 1;
        uint32_t n_bits = tl - mic_high_bits(next_state);
        uint32_t new_state = (next_state << n_bits) - table_size;
        if (new_state >= table_size) return MICD_ERR_CORRUPT;
        if (new_state == p && n_bits == 0) return MICD_ERR_CORRUPT;
        u.tt_nb[p] = new_state | (n_bits << 16);
    }
    return MICD_OK;
}

// buildRansDecTable (ransu16.go:77-135): sequential fill, normals first then the -1 symbols.
__device__ inline int mic_build_rans_dtable(MicUnit &u) {
    const int32_t *norm = u.norm;
    uint32_t tl = u.table_log, symbol_len = u.symbol_len;
    uint32_t table_size = 1u << tl;
    uint32_t slot = 0, zero_bits = 0;
    int32_t large_limit = (int32_t)(1u << (tl - 1));
    for (uint32_t s = 0; s < symbol_len; s++) {
        int32_t v = norm[s];
        if (v <= 0) continue;
        if (v >= large_limit) zero_bits = 1;
        uint32_t freq = (uint32_t)v;
        for (uint32_t j = 0; j < freq; j++) {
            if (slot >= table_size) return MICD_ERR_CORRUPT;
            uint32_t x_next = freq + j;
            uint32_t nb = tl - mic_high_bits(x_next);
            uint32_t base = (x_next << nb) - table_size;
            if (base >= table_size) return MICD_ERR_CORRUPT;
            u.tab_sym[slot] = (uint16_t)s;
            u.tt_nb[slot] = base | (nb << 16);
            slot++;
        }
    }
    for (uint32_t s = 0; s < symbol_len; s++) {
        if (norm[s] != -1) continue;
        if (slot >= table_size) return MICD_ERR_CORRUPT;
        u.tab_sym[slot] = (uint16_t)s;
        u.tt_nb[slot] = 0u | (tl << 16);
        slot++;
    }
    if (slot != table_size) return MICD_ERR_CORRUPT;
    u.zero_bits = zero_bits;
    return MICD_OK;
}
