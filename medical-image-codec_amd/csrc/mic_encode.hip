// mic_encode.hip -- encode kernels of the MIC unit codec for gfx950.
//
//   k_enc_tokens   Delta(avg) predictor + escape + RLE tokeniser   (deltarlecompressu16.go:24-67,
//                                                                  rlecompressu16.go:24-83)
//   k_enc_hist     16-bit-alphabet histogram of the token stream   (fsecompressu16.go:438-462)
//   k_enc_tables   gates, tableLog, normalise, NCount, CTable      (fse2state.go:22-52 et al.)
//   k_enc_tans     N-state tANS encode + framing + fallback chain  (fse2state.go:122-199, ...)
//
// Launch shape: blockIdx.x = unit.  See DESIGN.md for the HBM layout.
#include "mic_dev.h"
#include "mic_fse_tables.h"
#include "mic_launch.h"

// ------------------------------------------------------------------------------------------
// Parallel Delta(avg) + RLE tokeniser: one work-group of 1024 threads per unit.
//
// The reference tokeniser (rlecompressu16.go:24-83) is a serial state machine, but its output
// is a pure function of the maximal runs of the delta-symbol stream x[0..M) (x[0] = maxValue):
//   * a maximal run of L >= 3 equal symbols ("same-run") is emitted as (c, v) every c symbols
//     from its (c+3)-th symbol on, then (rem, v) with rem = (L-3) % c + 3 when it ends, where
//     c = midCount - 3 (the flush at len(b) >= midCount-1 keeps two symbols buffered);
//   * the symbols between same-runs ("diff stretch") are emitted as literal chunks of c symbols,
//     header midCount + chunk length in front; a chunk boundary needs two more symbols behind
//     it (they sit in the reference's buffer when the flush fires), so the last two symbols of
//     the whole stream never open a chunk.
// Every symbol can therefore compute the tokens it is responsible for from x[i-3..i+3], its
// index inside its run / stretch (segmented max-scans) and a prefix sum of token counts.
// A tile is 2048 pixels: their symbols (1 or 2 per pixel, prefix-summed only when the tile holds
// an escape) go into an LDS window and are tokenised with a delay of 3 symbols so the look-ahead
// is always present.  A thread owns 8 consecutive window positions; the equality pattern of its
// 14-symbol neighbourhood is one bit mask and "in a same-run", "run start", "stretch start",
// "run end" are shifts and ANDs of it; run / stretch indices modulo c advance incrementally from
// one multiply-high reduction per thread and tile.
#define TK_THREADS 512              // three groups per CU (<= 80 VGPRs each): one group's barriers overlap the others' work
#define TK_WAVES 8
#define TK_PPT 8                    // pixels per thread and tile
#define TK_SPT 8                    // window positions per thread
#define TK_WIN (TK_THREADS * TK_SPT)
#define TK_HWIN 8192               // histogram window in LDS

// Wave-wide inclusive scans on the DPP network (row_shr 1/2/4/8, then row_bcast 15 and 31); lanes
// without a source read the identity 0, so the same shape serves add and max.
#define TK_DPP(x, ctrl, rmask) __builtin_amdgcn_update_dpp(0u, (x), (ctrl), (rmask), 0xF, false)
__device__ __forceinline__ uint32_t tk_wave_incl_add(uint32_t v, uint32_t) {
    v += TK_DPP(v, 0x111, 0xF); v += TK_DPP(v, 0x112, 0xF); v += TK_DPP(v, 0x114, 0xF); v += TK_DPP(v, 0x118, 0xF);
    v += TK_DPP(v, 0x142, 0xA); v += TK_DPP(v, 0x143, 0xC);
    return v;
}
__device__ __forceinline__ uint32_t tk_wave_incl_max(uint32_t v, uint32_t) {
    v = max(v, TK_DPP(v, 0x111, 0xF)); v = max(v, TK_DPP(v, 0x112, 0xF)); v = max(v, TK_DPP(v, 0x114, 0xF)); v = max(v, TK_DPP(v, 0x118, 0xF));
    v = max(v, TK_DPP(v, 0x142, 0xA)); v = max(v, TK_DPP(v, 0x143, 0xC));
    return v;
}
// TK_WAVES per-wave partials in LDS -> this wave's exclusive prefix and the group total (add / max).  Every lane
// takes partial (lane & 7); a 3-step DPP scan inside the row and two lane reads replace a loop over the partials.
__device__ __forceinline__ void tk_block16_add(const uint32_t *s, uint32_t wave, uint32_t &excl, uint32_t &total) {
    const uint32_t v = s[threadIdx.x & 7];
    uint32_t x = v;
    x += TK_DPP(x, 0x111, 0xF); x += TK_DPP(x, 0x112, 0xF); x += TK_DPP(x, 0x114, 0xF);     // inclusive over lanes 0..7 of each row
    const uint32_t w = __builtin_amdgcn_readfirstlane(wave);
    total = __builtin_amdgcn_readlane(x, 7);
    excl = __builtin_amdgcn_readlane(x, w) - __builtin_amdgcn_readlane(v, w);
}
__device__ __forceinline__ void tk_block16_max(const uint32_t *s, uint32_t wave, uint32_t &excl, uint32_t &total) {
    const uint32_t v = s[threadIdx.x & 7];
    uint32_t x = v;
    x = max(x, TK_DPP(x, 0x111, 0xF)); x = max(x, TK_DPP(x, 0x112, 0xF)); x = max(x, TK_DPP(x, 0x114, 0xF));
    const uint32_t w = __builtin_amdgcn_readfirstlane(wave);
    total = __builtin_amdgcn_readlane(x, 7);
    excl = w ? __builtin_amdgcn_readlane(x, w - 1) : 0u;
}

typedef uint32_t tk_v4 __attribute__((ext_vector_type(4)));
typedef tk_v4 TkQ __attribute__((aligned(2)));                 // 8 pixels / tokens, 2-byte aligned
// two u16 per dword on the packed VALU (v_pk_add_u16, v_pk_sub_i16, v_pk_lshrrev_b16, v_pk_min_u16)
typedef unsigned short tk_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tk_us2 tk_u(uint32_t x) { return __builtin_bit_cast(tk_us2, x); }
__device__ __forceinline__ uint32_t tk_w(tk_us2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t tk_pk_add(uint32_t a, uint32_t b) { return tk_w(tk_u(a) + tk_u(b)); }
__device__ __forceinline__ uint32_t tk_pk_sub(uint32_t a, uint32_t b) { return tk_w(tk_u(a) - tk_u(b)); }
__device__ __forceinline__ uint32_t tk_pk_shr1(uint32_t a) { return tk_w(tk_u(a) >> (tk_us2)(1)); }
__device__ __forceinline__ uint32_t tk_pk_min(uint32_t a, uint32_t b) { return tk_w(__builtin_elementwise_min(tk_u(a), tk_u(b))); }
__device__ __forceinline__ uint32_t tk_half(const uint32_t (&w)[4], int k) { return (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xFFFFu); }

// SRC 0: frame units (mode 0) -- symbols are the Delta(avg) residuals of the pixels, stream = [delim][RLE(maxValue, symbols)].
// SRC 1: RLE-of-symbols units (mode 2, wavelet / residual paths) -- RleCompressU16.Init(len,1,max).Compress(symbols)
//        (rlecompressu16.go:85-93): symbols come from u.sym[0..u.nsym), stream = [max][len>>16][len&0xFFFF][RLE(symbols)].
// PRED (frame units only) 1: the gradient-adaptive predictor of GradDeltaRleCompressU16 (deltagradrlecompressu16.go:26-68) -- the
//        same stream with mic_grad_predict(W, N, NW, NE) in place of avg(W, N); units with u.pred == 1 (PICA's second encode).
//
// A tile is 4096 pixels, eight per thread, addressed LINEARLY: pixel g's left neighbour is g - 1 and its upper one g - W wherever
// the thread's eight pixels lie -- across a row end too; the one pixel of a thread that starts a row takes its upper neighbour as
// its left one (avg(t, t) = t).  With every sample below 2^15 the residuals come two per instruction on the packed VALU.  Their
// symbols (8 per thread; 16 when every pixel is an escape) go to a 4096-symbol LDS window; a tile that holds an escape is walked
// in two halves, so the window never holds more.  All 512 threads own 8 window positions.

#ifndef TK_WPS
#define TK_WPS 6        // waves per SIMD the tokeniser is compiled for (6: three groups per CU, 80 VGPRs)
#endif
#ifndef TK_ABL
#define TK_ABL 0       // timing-only ablations (tools/abl_tok.sh): 1 = no general count loop, 2 = no general write loop
#endif
template <int SRC, int PRED = 0>
__global__ void __launch_bounds__(TK_THREADS, TK_WPS) k_enc_tokens_wg(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (!SRC && u.pred != (uint32_t)PRED) return;
    __shared__ __attribute__((aligned(16))) uint16_t xs2[2][TK_WIN + 16];   // per pass parity: [0..5] = 6 symbols before the window, [6..] = new symbols
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[16], s_run[16], s_str[16], s_tc[16];   // per-wave partials (unused tail stays 0)
    __shared__ uint32_t s_ovf, s_last[2];
    __shared__ uint32_t s_frun;                                 // run1 as a fast tile leaves it (its last thread knows it)
    // fused histogram of the token stream (fsecompressu16.go:438-462): an 8192-bin LDS window around
    // the delta threshold takes almost every token; the rest goes to HBM atomics
    __shared__ uint32_t s_hist[TK_HWIN];
    __shared__ uint32_t s_tmaxall;
    __shared__ tk_v4 s_item[TK_THREADS / 64][8];                // per wave: the threads whose positions are written one per lane (phase D)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (u.mode != (SRC ? 2u : 0u)) return;                    // bare-FSE units (mode 1) bring their own tokens
    if (SRC && u.status != MICD_OK) return;                   // the symbol producer already failed
    if (tid == 0) { u.status = MICD_OK; u.ntok = 0; u.blob_len = 0; u.nstates_used = 0; s_ovf = 0; s_last[0] = s_last[1] = 0; s_tmaxall = 0; s_frun = 0; }
    const int depth = mic_len16(u.max_value);
    if (!SRC && (u.w <= 0 || u.h <= 0)) { if (tid == 0) u.status = MICD_ERR_ARGS; return; }
    if (depth < 4) { if (tid == 0) u.status = MICD_ERR_UNSUPPORTED; return; }   // see k_enc_tokens_serial
    const uint32_t thr = (1u << (depth - 1)) - 1;
    const uint32_t delim = (1u << depth) - 1;
    const uint32_t mid = (1u << (depth - 1)) - 1;          // Len16(delim) == depth
    const uint32_t c = mid - 3;                            // >= 4
    const uint32_t cmagic = (uint32_t)(0x100000000ull / c);
    auto div_c = [&](uint32_t a, uint32_t &rem) -> uint32_t {   // a / c and a % c: the multiply-high quotient is at most 1 short
        uint32_t qq = __umulhi(a, cmagic), r = a - qq * c;
        if (r >= c) { r -= c; qq++; }
        rem = r; return qq;
    };
    auto mod_c = [&](uint32_t a) -> uint32_t { uint32_t r; (void)div_c(a, r); return r; };
    const mic_gp<const uint16_t> in = mic_g(SRC ? (const uint16_t *)u.sym : u.px_in);   // (global_load / global_store: the prefetch
    const mic_gp<uint16_t> tok = mic_g(u.tok);                                          //  below has to survive the barriers)
    typedef MIC_GLOBAL const TkQ *GTkQc; typedef MIC_GLOBAL TkQ *GTkQ;
    const uint32_t cap = u.tok_cap;
    const uint32_t W = SRC ? 1u : (uint32_t)u.w;
    const uint32_t npx = SRC ? u.nsym : W * (uint32_t)u.h;
    constexpr uint32_t TP = TK_THREADS * TK_PPT;
    const uint32_t ntiles = (npx + TP - 1) / TP;
    // carried, work-group uniform state
    uint32_t g0 = SRC ? 0u : 1u;     // symbols generated so far; frames: symbol 0 = maxValue is pre-seeded in the halo
    uint32_t outp = SRC ? 3u : 1u;   // tokens written so far; tok[0] = delimiter / max (rlecompressu16.go:21) [+ length words]
    uint32_t run1 = 0, str1 = 0;     // index+1 of the first symbol of the current run / diff stretch (0 = none)
    uint32_t par = 0;                // window parity: every pass over a window flips it
    bool prev_fast = false;          // the pass before took the fast path: run1 is in s_frun
    uint32_t fast_cool = 0;          // passes to go before a fast tile is tried again (a refused try costs a barrier)
    const uint32_t thr2 = thr | (thr << 16), lim2 = (2 * thr - 2) | ((2 * thr - 2) << 16);
    const uint32_t hlo = (!SRC && delim >= TK_HWIN && thr > TK_HWIN / 2) ? thr - TK_HWIN / 2 : 0u;   // window [hlo, hlo + TK_HWIN)
    const mic_gp<uint32_t> ghist = mic_g(u.hist);
    const uint32_t tabcap = u.tab_cap;
    for (uint32_t i = tid; i < TK_HWIN; i += TK_THREADS) s_hist[i] = 0;
    for (uint32_t i = tid; i < 2 * (TK_WIN + 16); i += TK_THREADS) (&xs2[0][0])[i] = 0;
    if (tid < 16) { s_cnt[tid] = 0; s_run[tid] = 0; s_str[tid] = 0; s_tc[tid] = 0; }
    __syncthreads();
    uint32_t tmax = 0;                                      // largest token counted outside the LDS window
    auto count_tok = [&](uint32_t v) {
        const uint32_t d = v - hlo;
        if (d < TK_HWIN) atomicAdd(&s_hist[d], 1u);
        else if (v < tabcap) { (void)__hip_atomic_fetch_add(&ghist[v], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tmax = max(tmax, v); }
        else s_ovf = 1;                                         // (tier 1: the histogram slab has 8192 bins)
    };
    if (tid == 0) {
        if (SRC) {
            if (cap > 2) {
                tok[0] = u.max_value; tok[1] = (uint16_t)(npx >> 16); tok[2] = (uint16_t)npx;   // rlecompressu16.go:21, :86-87
                count_tok(u.max_value); count_tok(npx >> 16); count_tok(npx & 0xFFFF);
            } else s_ovf = 1;
        } else {
            if (cap > 0) { tok[0] = (uint16_t)delim; count_tok(delim); }
            xs2[0][5] = u.max_value;
        }
    }
    __syncthreads();
    // Symbol units (SRC 1: one group per CU on WaveletV2 frames) fetch a tile ahead -- the group otherwise idles through an HBM round
    // trip per tile.  Frame units do not: the pixels in flight are 12 registers held across the whole tile, which at three groups
    // per CU (80 registers) cost 23 spilled registers in every tile and 1.2 ms of 4.4 per batch; the other two groups cover the trip.
    constexpr bool AHEAD = SRC != 0;
    // kind 1: the eight pixels, the eight above them and the one to the left came as vectors (all eight have an upper neighbour, or
    // none has: row 0); kind 2: the thread reads its pixels one by one (the unit's tail, the step from row 0 to row 1, W < 8)
    struct TkFetch { tk_v4 cv, tv; uint32_t lft, tl, tr, x, y, kind; };
    const uint32_t tile_dy = TP / W, tile_dx = TP - tile_dy * W;   // a tile further on: tile_dy rows and tile_dx columns
    uint32_t ny = (tid * TK_PPT) / W, nx = tid * TK_PPT - ny * W;  // position of this thread's first pixel in the next tile to fetch
    auto fetch = [&](uint32_t tile) -> TkFetch {
        TkFetch f; f.cv = tk_v4{0u, 0u, 0u, 0u}; f.tv = tk_v4{0u, 0u, 0u, 0u}; f.lft = 0; f.tl = 0; f.tr = 0; f.x = 0; f.y = 0; f.kind = 0;
        const uint32_t gb = tile * TP + tid * TK_PPT;
        if (tile >= ntiles || gb >= npx) return f;
        if (SRC) {
            if (gb + TK_PPT <= npx) { f.cv = *(GTkQc)(in + gb); f.kind = 1; } else f.kind = 2;
            return f;
        }
        f.x = nx; f.y = ny;                                     // tiles are fetched in order: the position advances by a tile
        nx += tile_dx; ny += tile_dy;
        if (nx >= W) { nx -= W; ny++; }
        f.kind = 2;
        if (gb + TK_PPT <= npx && W >= TK_PPT && (gb >= W || gb + TK_PPT <= W)) {
            f.kind = 1;
            f.cv = *(GTkQc)(in + gb);
            if (gb >= W) f.tv = *(GTkQc)(in + gb - W);
            if (gb > 0) f.lft = in[gb - 1];
            if (PRED && gb >= W) {                                  // NW of the first pixel, NE of the last
                if (gb > W) f.tl = in[gb - W - 1];
                f.tr = in[gb - W + TK_PPT];
            }
        }
        return f;
    };
    TkFetch nxt = AHEAD ? fetch(0) : TkFetch{};
    MIC_STAMP_BEGIN();
    for (uint32_t tile = 0; tile <= ntiles; tile++) {
        const bool flush = tile == ntiles;
        // ---- A: delta symbols of this tile's pixels (deltarlecompressu16.go:31-61) ----------
        uint32_t ls[2 * TK_PPT]; uint32_t cnt = 0; int esc = 0;
        bool pk_ok = false; uint32_t pk[4] = { 0u, 0u, 0u, 0u };   // the thread's eight symbols as four packed dwords (the usual case)
        const uint32_t gbase = tile * TP + tid * TK_PPT;
        TkFetch f;
        if (AHEAD) { f = nxt; nxt = fetch(tile + 1); } else f = fetch(tile);
        if (!flush && gbase < npx) {
            const uint32_t cw[4] = { f.cv.x, f.cv.y, f.cv.z, f.cv.w };
            if (SRC) {
                if (f.kind == 1) { pk_ok = true; pk[0] = cw[0]; pk[1] = cw[1]; pk[2] = cw[2]; pk[3] = cw[3]; cnt = TK_PPT; }
                else
#pragma unroll
                for (int k = 0; k < TK_PPT; k++) if (gbase + k < npx) ls[cnt++] = (uint32_t)in[gbase + k];
            } else if (f.kind == 1) {
                const uint32_t tw[4] = { f.tv.x, f.tv.y, f.tv.z, f.tv.w };
                const uint32_t x = f.x, y = f.y;
                if (!PRED && y > 0 && ((cw[0] | cw[1] | cw[2] | cw[3] | tw[0] | tw[1] | tw[2] | tw[3] | f.lft) & 0x80008000u) == 0) {
                    // prev = floor((l + t) / 2) = (l & t) + ((l ^ t) >> 1), symbol = thr + cur - prev, two pixels per instruction.  The
                    // 16-bit differences are exact below 2^15, and |diff| < thr <=> 1 <= symbol <= 2 thr - 1 (a wrapped negative
                    // lands above 2^15 + thr).  A thread with an escape takes the per-pixel code below.
                    uint32_t lw[4] = { __builtin_amdgcn_alignbit(cw[0], f.lft << 16, 16), __builtin_amdgcn_alignbit(cw[1], cw[0], 16),
                                       __builtin_amdgcn_alignbit(cw[2], cw[1], 16), __builtin_amdgcn_alignbit(cw[3], cw[2], 16) };
                    if (x == 0 || x + TK_PPT > W) {                    // pixel kz starts a row: its prediction is the pixel above it
                        const uint32_t kz = x ? W - x : 0u;
                        const uint32_t hm = (kz & 1u) ? 0xFFFF0000u : 0x0000FFFFu;
#pragma unroll
                        for (int j = 0; j < 4; j++) if ((kz >> 1) == (uint32_t)j) lw[j] = (lw[j] & ~hm) | (tw[j] & hm);
                    }
                    uint32_t bad = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t pv = tk_pk_add(lw[j] & tw[j], tk_pk_shr1(lw[j] ^ tw[j]));
                        pk[j] = tk_pk_add(tk_pk_sub(cw[j], pv), thr2);
                        const uint32_t e = tk_pk_sub(pk[j], 0x00010001u);
                        bad |= tk_pk_min(e, lim2) ^ e;
                    }
                    if (!bad) { pk_ok = true; cnt = TK_PPT; }
                }
                if (!pk_ok)
#pragma unroll
                for (int k = 0; k < TK_PPT; k++) {
                    uint32_t xk = x + k, yk = y;
                    if (xk >= W) { xk -= W; yk++; }
                    const uint32_t val = tk_half(cw, k), l = k ? tk_half(cw, k ? k - 1 : 0) : f.lft, t = tk_half(tw, k);
                    int32_t prev;
                    if (xk > 0 && yk > 0) {
                        if (PRED) {
                            const int32_t nw = (int32_t)(k ? tk_half(tw, k ? k - 1 : 0) : f.tl);
                            const int32_t ne = (xk + 1 < W) ? (int32_t)(k + 1 < TK_PPT ? tk_half(tw, k + 1 < TK_PPT ? k + 1 : 0) : f.tr) : nw;
                            prev = mic_grad_predict((int32_t)l, (int32_t)t, nw, ne);
                        } else prev = (int32_t)((l + t) >> 1);
                    }
                    else if (xk > 0) prev = (int32_t)l;
                    else prev = (int32_t)t;                              // 0 on row 0 (kind 1 there: no upper row was read)
                    const int32_t diff = (int32_t)val - prev;
                    const uint32_t ad = (uint32_t)(diff < 0 ? -diff : diff) & 0xFFFF;
                    if (ad >= thr) { ls[cnt++] = delim; ls[cnt++] = val; esc = 1; }
                    else ls[cnt++] = (uint32_t)((int32_t)thr + diff) & 0xFFFF;
                }
            } else {
#pragma unroll
                for (int k = 0; k < TK_PPT; k++) {
                    const uint32_t g = gbase + k;
                    if (g >= npx) break;
                    const uint32_t yy = g / W, xx = g - yy * W;
                    int32_t prev;
                    if (PRED) prev = mic_grad_predict_at((const uint16_t *)in, (int)W, (int)xx, (int)yy);
                    else {
                        prev = 0;
                        if (xx > 0) prev = in[g - 1];
                        if (yy > 0) prev += in[g - W];
                        if (xx > 0 && yy > 0) prev >>= 1;
                    }
                    const uint32_t val = in[g];
                    const int32_t diff = (int32_t)val - prev;
                    const uint32_t ad = (uint32_t)(diff < 0 ? -diff : diff) & 0xFFFF;
                    if (ad >= thr) { ls[cnt++] = delim; ls[cnt++] = val; esc = 1; }
                    else ls[cnt++] = (uint32_t)((int32_t)thr + diff) & 0xFFFF;
                }
            }
        }
        // The symbols go to the window as if the tile held no escape (a thread's offset is then its own); the barrier that collects the
        // escape flags orders these stores too, and a tile with an escape stores again, half by half, behind a scan.
        {
            uint16_t *xs = xs2[par];
            if (pk_ok) { uint32_t *d = (uint32_t *)(xs + 6 + tid * TK_PPT); d[0] = pk[0]; d[1] = pk[1]; d[2] = pk[2]; d[3] = pk[3]; }
            else if (!esc) {
#pragma unroll
                for (int k = 0; k < TK_PPT; k++) if ((uint32_t)k < cnt) xs[6 + tid * TK_PPT + k] = (uint16_t)ls[k];
            }
        }
        const bool esc_tile = __syncthreads_or(esc) != 0;
        if (prev_fast) { run1 = s_frun; prev_fast = false; }    // written behind the fast tile's decision barrier, ordered by this one
        if (esc_tile && pk_ok) {
#pragma unroll
            for (int k = 0; k < TK_PPT; k++) ls[k] = tk_half(pk, k);
        }
        const uint32_t npass = esc_tile ? 2u : 1u;
        for (uint32_t pass = 0; pass < npass; pass++) {
        uint16_t *xs = xs2[par], *xs_next = xs2[par ^ 1];
        uint32_t n;
        if (esc_tile) {                                         // rare: an escape doubles its pixel, offsets need a scan (per half tile)
            const bool mine = (tid >> 8) == pass;
            const uint32_t mycnt = mine ? cnt : 0u;
            const uint32_t incl = tk_wave_incl_add(mycnt, lane);
            if (lane == 63) s_cnt[wave] = incl;
            __syncthreads();
            uint32_t woff;
            tk_block16_add(s_cnt, wave, woff, n);
            const uint32_t off = woff + incl - mycnt;
#pragma unroll
            for (int k = 0; k < 2 * TK_PPT; k++) if ((uint32_t)k < mycnt) xs[6 + off + k] = (uint16_t)ls[k];
            __syncthreads();
        } else n = flush ? 0u : min(TP, npx - tile * TP);
        if (tid < 6) xs_next[tid] = xs[n + tid];                // halo of the next window (idle until then)
        const uint32_t g1 = g0 + n;
        MIC_STAMP_AT(u, 0);
        // window position p <-> symbol i = g0 - 3 + p <-> xs[p + 3];  i3 = i + 3 = g0 + p keeps the arithmetic unsigned
        const uint32_t nwin = flush ? 3u : n;
        const uint32_t p0 = tid * TK_SPT;
        // ---- B: equality pattern of the thread's neighbourhood x[i0-3 .. i0+10] -----------------
        uint32_t v[14];
        {
            const uint4 a = *(const uint4 *)(xs + p0);
            const uint2 b = *(const uint2 *)(xs + p0 + 8);
            const uint32_t c2 = *(const uint32_t *)(xs + p0 + 12);
            const uint32_t d[7] = { a.x, a.y, a.z, a.w, b.x, b.y, c2 };
#pragma unroll
            for (int m = 0; m < 14; m++) v[m] = (m & 1) ? (d[m >> 1] >> 16) : (d[m >> 1] & 0xFFFFu);
        }
        uint32_t EM = 0;                                        // bit m: x[i0-3+m] == x[i0-2+m] and both exist
#pragma unroll
        for (int m = 0; m < 13; m++) EM |= (v[m] == v[m + 1]) ? (1u << m) : 0u;
        {
            // pair m = symbols (a, a+1), a = g0 + p0 + m - 6: a >= 0, and a + 1 < g1 once the stream has ended
            const int32_t mlo = 6 - (int32_t)(g0 + p0);
            if (mlo > 0) EM &= (mlo >= 13) ? 0u : (0xFFFFFFFFu << mlo);
            if (flush) {
                const int32_t mhi = (int32_t)(g1 + 5) - (int32_t)(g0 + p0);      // pairs m < mhi exist
                EM &= (mhi <= 0) ? 0u : (mhi >= 13 ? 0x1FFFu : ((1u << mhi) - 1u));
            }
        }
        // validity of the 8 positions: p < nwin and i >= 0
        uint32_t V = 0;
        {
            const uint32_t hi = (p0 >= nwin) ? 0u : min(nwin - p0, (uint32_t)TK_SPT);
            V = (1u << hi) - 1u;
            const int32_t lo = 3 - (int32_t)(g0 + p0);                           // positions q < lo have i < 0
            if (lo > 0) V &= (lo >= 8) ? 0u : (0xFFu << lo);
        }
        const uint32_t A = EM & (EM >> 1);
        const uint32_t SS = A | (A >> 1) | (A >> 2);            // bit q+1: symbol q sits in a run of >= 3; bit 0: the symbol before
        const uint32_t SAME = (SS >> 1) & V;
        const uint32_t RS = ~(EM >> 2) & V;                     // first symbol of a maximal run
        uint32_t PREV = SS & 0xFFu;                             // bit q: isSame of symbol q-1
        if (p0 == 0) PREV = (PREV & ~1u) | s_last[par ^ 1];     // the symbol before the window was classified last pass
        {
            const int32_t q0 = 3 - (int32_t)(g0 + p0);          // position of symbol 0, if it is one of these 8
            if (q0 >= 0 && q0 < 8) PREV |= 1u << q0;            // i == 0 opens a stretch whatever came "before"
        }
        const uint32_t STS = ~SAME & PREV & V;                  // first symbol of a diff stretch
        const uint32_t LAST = ~(EM >> 3);                       // run ends here (next symbol differs or does not exist)
        const uint32_t ibase = g0 + p0 - 2;                     // (i + 1) of position 0
        const uint32_t my_run = RS ? ibase + (31 - __clz(RS)) : 0u;
        const uint32_t my_str = STS ? ibase + (31 - __clz(STS)) : 0u;
        // no same-run symbol among these 8 or right behind them (SS bits 1..9), no stretch start: eight literals of one stretch.  The
        // only thing that can happen is a chunk boundary: position bq opens a chunk (one more token: its header slot) and position
        // bq - 1 closes one (it patches the header c tokens back).
        const bool lit8 = !flush && V == 0xFFu && (SS & 0x3FEu) == 0 && STS == 0 && c >= 16;
        // ---- fast tile: every thread of a full tile is lit8 and the stretch is under way.  Token positions are closed-form then: no
        // scans, one barrier (the vote) instead of two; the only carried state that moves is the start of the last run, and the tile's
        // last thread knows it.  A refused vote costs its barrier, so it is tried again only after a while.
        bool tile_fast = false;
        if (!flush && !esc_tile && n == TP && c >= 16 && g0 >= 6 && str1 != 0 && g0 >= str1 + 3) {
            if (fast_cool) fast_cool--;
            else { tile_fast = !__syncthreads_or(lit8 ? 0 : 1); fast_cool = tile_fast ? 0u : 12u; }
        }
        uint32_t run_in = 0, str_in = str1, run_tot = run1, str_tot = str1;
        if (!tile_fast) {
            // exclusive max-scan of the thread-latest starts (positions grow with the thread index)
            const uint32_t run_incl = tk_wave_incl_max(my_run, lane), str_incl = tk_wave_incl_max(my_str, lane);
            if (lane == 63) { s_run[wave] = run_incl; s_str[wave] = str_incl; }
            run_in = __shfl_up(run_incl, 1); str_in = __shfl_up(str_incl, 1);
            if (lane == 0) { run_in = 0; str_in = 0; }
            __syncthreads();
            uint32_t ra, sa;
            tk_block16_max(s_run, wave, ra, run_tot);
            tk_block16_max(s_str, wave, sa, str_tot);
            run_in = max(max(run_in, ra), run1); str_in = max(max(str_in, sa), str1);
            run_tot = max(run_tot, run1); str_tot = max(str_tot, str1);
        }
        MIC_STAMP_AT(u, 1);
        // ---- C: tokens owned by each position ---------------------------------------------------
        // k = 1-based index in the run, j = 1-based index in the stretch; rk = (k-3) % c, sj = (j-1) % c
        uint32_t k_in = 0, j_in = 0, rk_in = 0, sj_in = 0;     // state in front of position 0 (D replays the recurrence)
        uint32_t bq = 0xFFFFu;                                  // lit8: position that opens a literal chunk (>= 8: none here)
        uint32_t tsum = 0, jq = 0;
        const bool cf = V == 0xFFu && !lit8 && !flush && c >= 16;   // counted in closed form, written one position per lane
        uint32_t itx = 0, ity = 0, itz = 0;
        const uint32_t ex2_lim = g1 + 1;                        // flush: ex2 <=> i3 < g1 + 1 ; ex1 <=> i3 < g1 + 2 ; ex3 <=> i3 < g1
        if (V) {
            // state of the symbol in front of position 0 (index i0 - 1 = ibase - 2)
            uint32_t k = (run_in != 0 && ibase >= run_in + 1) ? ibase - run_in : 0u;       // (i0-1) + 2 - run_in
            uint32_t j = (str_in != 0 && ibase >= str_in + 1) ? ibase - str_in : 0u;
            if (lit8) {
                uint32_t jr;                                     // j >= 1 here (the stretch is under way)
                jq = div_c(j, jr);
                bq = jr ? c - jr : 0u;                           // position q opens a chunk <=> (j + q) % c == 0
                jq += jr ? 1u : 0u;                              // ceil(j / c): chunks of the stretch opened in front of position 0, + 1
                tsum = TK_SPT + (bq < TK_SPT ? 1u : 0u);
            } else if (TK_ABL & 1) tsum = __popc(V);
            else {
                uint32_t rk = (k >= 3) ? mod_c(k - 3) : 0u;
                uint32_t sj = (j >= 1) ? mod_c(j - 1) : 0u;
                k_in = k; j_in = j; rk_in = rk; sj_in = sj;
                if (cf) {
                    // With c >= 16 > 8 a count wraps at most once among eight positions, and only in the run / stretch that comes in
                    // from the left (one that starts here stays below c), so the loop below has a closed form over the bit masks:
                    // a same-run symbol owns 2 tokens where its run ends (LAST) and 2 where the run's count wraps; any other symbol
                    // owns its literal, + 1 where it opens a chunk (stretch start, or the incoming stretch's index wraps).
                    const uint32_t S8 = SAME & 0xFFu, N8 = ~S8 & 0xFFu;
                    const uint32_t fr = RS ? (uint32_t)__builtin_ctz(RS) : 8u, fs = STS ? (uint32_t)__builtin_ctz(STS) : 8u;
                    const uint32_t qw = c - 1 - rk, qs = c - 1 - sj;
                    const uint32_t cfW = (k >= 3 && qw < fr) ? ((1u << qw) & S8) : 0u;
                    const uint32_t cfS = STS | ((qs < fs) ? ((1u << qs) & N8) : 0u) | ((j == 0 && fs > 0) ? (1u & N8) : 0u);
                    tsum = __popc(N8) + __popc(cfS) + 2 * (__popc(S8 & LAST) + __popc(cfW));
                    // what phase D needs of this thread, packed: SAME | RS << 8 | STS << 16 | SAME & LAST << 24 (eight positions each);
                    // positions that open a chunk | positions where the incoming run's count wraps << 8 | EM bits 3..12 << 16 | lane << 26;
                    // (count - 3) % c of the incoming run (one shorter than 3 so far: its count - 3, mod c) | (index - 1) % c of the
                    // incoming stretch << 16
                    itx = S8 | ((RS & 0xFFu) << 8) | ((STS & 0xFFu) << 16) | ((S8 & LAST) << 24);
                    ity = cfS | (cfW << 8) | (((EM >> 3) & 0x3FFu) << 16) | (lane << 26);
                    itz = (k >= 3 ? rk : c + k - 3) | (sj << 16);
                } else
#pragma unroll
                for (int q = 0; q < TK_SPT; q++) {
                    const uint32_t bit = 1u << q;
                    if (V & bit) {
                        if (RS & bit) k = 1; else k++;
                        if (k == 3) rk = 0; else if (k > 3) { rk++; if (rk == c) rk = 0; }
                        if (STS & bit) { j = 1; sj = 0; } else { j++; sj++; if (sj == c) sj = 0; }
                        uint32_t t;
                        if (SAME & bit) {
                            t = ((k > 3 && rk == 0) ? 2u : 0u) + ((LAST & bit) ? 2u : 0u);
                        } else {
                            const bool ex2 = !flush || (g0 + p0 + q) < ex2_lim;
                            t = ((j == 1) || (sj == 0 && ex2)) ? 2u : 1u;
                        }
                        tsum += t;
                    }
                }
            }
        }
#ifdef MIC_STAMP
        if (!__all(lit8 || !V) && lane == 0) atomicAdd(&u.dbg[15], 1u);   // wave-tiles that take the general path
#endif
        uint32_t pos, ttot;
        if (tile_fast) {
            // chunks opened in front of position p0 = ceil((J0 + p0) / c) - ceil(J0 / c), J0 = the stretch index of the symbol in front
            // of the tile (thread 0's j); the tile's total likewise
            uint32_t r0, rT;
            const uint32_t c0 = div_c(g0 - 2 - str1, r0) + (r0 ? 1u : 0u);
            const uint32_t cT = div_c(g0 - 2 - str1 + TP, rT) + (rT ? 1u : 0u);
            pos = outp + p0 + (jq - c0);
            ttot = TP + (cT - c0);
        } else {
            const uint32_t tincl = tk_wave_incl_add(tsum, lane);
            if (lane == 63) s_tc[wave] = tincl;
            __syncthreads();
            uint32_t toff;
            tk_block16_add(s_tc, wave, toff, ttot);
            pos = outp + toff + tincl - tsum;
        }
        MIC_STAMP_AT(u, 2);
        // ---- D: write ----------------------------------------------------------------------------
        uint64_t cf_bal = (TK_ABL & 6) ? 0ull : __ballot(cf && tsum != 0);   // (a thread inside a run owns no token)
        const bool cf_loop = __popcll(cf_bal) > 16;             // too many of them in this wave: the serial walk is cheaper
        if (lit8) {
            if (pos + TK_SPT + 1 <= cap) {
                if (bq >= TK_SPT) {
                    tk_v4 o;
                    o.x = v[3] | (v[4] << 16); o.y = v[5] | (v[6] << 16); o.z = v[7] | (v[8] << 16); o.w = v[9] | (v[10] << 16);
                    *(GTkQ)(tok + pos) = o;
                } else {                                         // header slot at pos + bq, patched by whoever closes that chunk
#pragma unroll
                    for (int q = 0; q < TK_SPT; q++) tok[pos + q + ((uint32_t)q >= bq ? 1u : 0u)] = (uint16_t)v[q + 3];
                }
                if (bq >= 1 && bq <= TK_SPT) {                   // position bq - 1 is the c-th literal of its chunk (next_starts)
                    const uint32_t lit = pos + bq - 1;
                    if (lit >= c) { tok[lit - c] = (uint16_t)(mid + c); count_tok(mid + c); } else s_ovf = 1;
                }
                uint32_t dmax = 0;                               // one window test for the 8 values
#pragma unroll
                for (int q = 0; q < TK_SPT; q++) dmax = max(dmax, v[q + 3] - hlo);
                if (dmax < TK_HWIN) {
#pragma unroll
                    for (int q = 0; q < TK_SPT; q++) atomicAdd(&s_hist[v[q + 3] - hlo], 1u);
                } else {
#pragma unroll
                    for (int q = 0; q < TK_SPT; q++) count_tok(v[q + 3]);
                }
            } else s_ovf = 1;
        } else if (V && (!cf || cf_loop) && !(TK_ABL & 2)) {
            bool ovf = false;
            uint32_t k = k_in, j = j_in, rk = rk_in, sj = sj_in;
#pragma unroll
            for (int q = 0; q < TK_SPT; q++) {
                const uint32_t bit = 1u << q;
                if (!(V & bit)) continue;
                if (RS & bit) k = 1; else k++;
                if (k == 3) rk = 0; else if (k > 3) { rk++; if (rk == c) rk = 0; }
                if (STS & bit) { j = 1; sj = 0; } else { j++; sj++; if (sj == c) sj = 0; }
                const uint32_t xv = v[q + 3];
                const uint32_t i3 = g0 + p0 + q;                                  // i + 3
                if (SAME & bit) {
                    if (k > 3 && rk == 0) {
                        if (pos + 1 < cap) { tok[pos] = (uint16_t)c; tok[pos + 1] = (uint16_t)xv; count_tok(c); count_tok(xv); } else ovf = true;
                        pos += 2;
                    }
                    if (LAST & bit) {
                        const uint32_t rem = rk + 3;                              // (k - 3) % c + 3
                        if (pos + 1 < cap) { tok[pos] = (uint16_t)rem; tok[pos + 1] = (uint16_t)xv; count_tok(rem); count_tok(xv); } else ovf = true;
                        pos += 2;
                    }
                } else {
                    const uint32_t jm = sj;
                    const bool ex1 = !flush || i3 < g1 + 2, ex2 = !flush || i3 < g1 + 1, ex3 = !flush || i3 < g1;
                    const bool starts = (j == 1) || (jm == 0 && ex2);
                    const uint32_t lit = pos + (starts ? 1u : 0u);
                    if (lit < cap) { tok[lit] = (uint16_t)xv; count_tok(xv); } else ovf = true;
                    pos += starts ? 2u : 1u;
                    // does the chunk end here?  next symbol: end of stream / start of a same-run / opens a chunk
                    bool same_next;
                    if (!ex1) same_next = false;
                    else if (v[q + 4] == xv) same_next = false;                   // a run of 2 (this symbol is not in a same-run)
                    else same_next = ex3 && v[q + 4] == v[q + 5] && v[q + 5] == v[q + 6];
                    const bool next_starts = (jm + 1 == c) && ex3;                // j % c == 0
                    if (!ex1 || same_next || next_starts) {
                        uint32_t qlen = jm + 1;
                        // boundary suppressed for the last two symbols of the stream: they extend the previous chunk
                        // (chunk start symbol i - jm has no two symbols behind it: i3 - jm + 2 >= g1 + 3)
                        if (flush && j > jm + 1 && !(i3 - jm + 2 < g1 + 3)) qlen += c;
                        if (lit >= qlen && lit - qlen < cap) { tok[lit - qlen] = (uint16_t)(mid + qlen); count_tok(mid + qlen); } else ovf = true;
                    }
                }
            }
            if (ovf) s_ovf = 1;
        }
        // Threads that are neither eight plain literals nor inside a run (the two ends of a run, a stretch start: a few per wave on
        // X-ray frames, where every row ends in a run) used to walk their eight positions one after the other while the whole wave
        // waited.  Their positions are written one per lane instead: up to eight such threads at a time leave their packed masks in
        // LDS, lane 8 e + q takes position q of thread e; a position's token offset is a popcount over the masks below it.
        if (cf_bal && !cf_loop) {
            uint64_t bal = cf_bal;
            bool pending = cf && tsum != 0;
            const uint16_t *xw = xs + 3 + (size_t)wave * 64 * TK_SPT;
            while (bal) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                const bool mine = pending && rank < 8;
                if (mine) s_item[wave][rank] = tk_v4{itx, ity, itz, pos};
                // (LDS operations of one wave are carried out in the order they are issued, and the compiler keeps a store and a load of
                // the same array in program order: no fence -- a release fence here would also wait for the pixel prefetch in flight)
                __builtin_amdgcn_wave_barrier();
                const uint32_t nitem = min(8u, (uint32_t)__popcll(bal));
                const uint32_t e = lane >> 3, q = lane & 7u;
                if (e < nitem) {
                    const tk_v4 it = s_item[wave][e];
                    const uint32_t S8 = it.x & 0xFFu, RSm = (it.x >> 8) & 0xFFu, STm = (it.x >> 16) & 0xFFu, LSm = it.x >> 24;
                    const uint32_t OPm = it.y & 0xFFu, WRm = (it.y >> 8) & 0xFFu, EMh = (it.y >> 16) & 0x3FFu, src = it.y >> 26;
                    const uint32_t N8 = ~S8 & 0xFFu, bit = 1u << q, lt = bit - 1u, le = (bit << 1) - 1u;
                    uint32_t P = it.w + __popc(N8 & lt) + __popc(OPm & lt) + 2 * (__popc(LSm & lt) + __popc(WRm & lt));
                    const uint32_t xv = xw[src * TK_SPT + q];
                    bool ovf = false;
                    if (S8 & bit) {
                        if (WRm & bit) {
                            if (P + 1 < cap) { tok[P] = (uint16_t)c; tok[P + 1] = (uint16_t)xv; count_tok(c); count_tok(xv); } else ovf = true;
                            P += 2;
                        }
                        if (LSm & bit) {
                            const uint32_t rsb = RSm & le;
                            uint32_t rkq;
                            if (rsb) rkq = q - (31u - (uint32_t)__clz(rsb)) - 2u;          // a run that began at one of these positions
                            else { rkq = (it.z & 0xFFFFu) + q + 1; if (rkq >= c) rkq -= c; }
                            const uint32_t rem = rkq + 3;                                 // (k - 3) % c + 3
                            if (P + 1 < cap) { tok[P] = (uint16_t)rem; tok[P + 1] = (uint16_t)xv; count_tok(rem); count_tok(xv); } else ovf = true;
                        }
                    } else {
                        const uint32_t lit = P + ((OPm >> q) & 1u);
                        if (lit < cap) { tok[lit] = (uint16_t)xv; count_tok(xv); } else ovf = true;
                        const uint32_t ssb = STm & le;
                        uint32_t sjq;
                        if (ssb) sjq = q - (31u - (uint32_t)__clz(ssb));
                        else { sjq = (it.z >> 16) + q + 1; if (sjq >= c) sjq -= c; }
                        // the chunk ends here when the next symbol starts a same-run (x[q+1] != x[q], x[q+1] == x[q+2] == x[q+3]) or opens a chunk
                        if (((EMh >> q) & 7u) == 6u || sjq + 1 == c) {
                            const uint32_t qlen = sjq + 1;
                            if (lit >= qlen && lit - qlen < cap) { tok[lit - qlen] = (uint16_t)(mid + qlen); count_tok(mid + qlen); } else ovf = true;
                        }
                    }
                    if (ovf) s_ovf = 1;
                }
                const uint64_t done = __ballot(mine);
                bal &= ~done;
                pending = pending && !mine;
                __builtin_amdgcn_wave_barrier();
            }
        }
        MIC_STAMP_AT(u, 3);
        // ---- E: carry --------------------------------------------------------------------------------
        // (the barriers of the next pass order these LDS words; both are double-buffered by window parity)
        outp += ttot;
        if (tile_fast) {
            if (tid == TK_THREADS - 1) s_frun = my_run;        // (RS != 0 there: eight symbols without a run of three)
            prev_fast = true;
        } else { run1 = run_tot; str1 = str_tot; }
        {
            // isSame of the last processed symbol, or the previous value when none was processed
            const bool processed = nwin > 0 && g0 + nwin - 1 >= 3;
            const uint32_t pl = processed ? nwin - 1 : 0u;
            if (processed) { if (tid == pl / TK_SPT) s_last[par] = (SAME >> (pl % TK_SPT)) & 1u; }
            else if (tid == 0) s_last[par] = s_last[par ^ 1];
        }
        g0 = g1;
        par ^= 1u;
        MIC_STAMP_AT(u, 4);
        }   // pass
    }
    __syncthreads();
    // window counts land on top of whatever the HBM atomics put there (nothing: disjoint bins)
    for (uint32_t i = tid; i < TK_HWIN; i += TK_THREADS) {
        const uint32_t vv = s_hist[i];
        if (vv) { if (hlo + i < tabcap) (void)__hip_atomic_fetch_add(&ghist[hlo + i], vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else s_ovf = 1; }
    }
    // what k_enc_tables_wg has to scan of the 65536 bins: the window and whatever was counted outside it
    tmax = tk_wave_incl_max(tmax, lane);
    if (lane == 63 && tmax) atomicMax(&s_tmaxall, tmax);
    __syncthreads();
    if (tid == 0) {
        if (s_ovf || outp > cap) u.status = u.tier == 1 ? MICD_INT_GROW : MICD_ERR_CAPACITY;
        else u.ntok = outp;
        u.hist_hi = min(min(65536u, tabcap), max(hlo + (uint32_t)TK_HWIN, s_tmaxall + 1u));
    }
}

// ------------------------------------------------------------------------------------------
// Bare FSE units (FSECompressU16 / TwoState / FourState / EightState, RANSCompressU16EightState:
// fsecompressu16.go:19, fse2state.go:22, fse4state.go:24, fse8state.go:31, rans8state.go:31):
// the caller's symbols are the token stream.  grid = (64, units), block = 256.
__global__ void __launch_bounds__(256) k_enc_symbols(MicUnit *units) {
    MicUnit &u = units[blockIdx.y];
    if (u.mode != 1) return;
    const uint32_t n = (uint32_t)u.w;
    // (a caller's symbols may be anything up to 65535: bare FSE units run on tier-2 slabs only -- every caller lays them out so)
    const bool fits = n <= u.tok_cap && u.tier != 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u.status = fits ? MICD_OK : (u.tier == 1 ? MICD_INT_GROW : MICD_ERR_CAPACITY); u.ntok = fits ? n : 0; u.blob_len = 0; u.nstates_used = 0;
    }
    if (!fits) return;
    const uint16_t *src = u.px_in; uint16_t *tok = u.tok; uint32_t *hist = u.hist;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint16_t v = src[i];
        tok[i] = v;
        atomicAdd(&hist[v], 1u);
    }
}



// ------------------------------------------------------------------------------------------
// LSB-first bit writer (bitwriter.go:50-53, :162-168).  The reference's flush32 cadence never
// changes content, so the stream is the plain concatenation of (value & mask(nb), nb) chunks.
struct BitW {
    uint8_t *out; uint32_t cap, len; uint64_t acc; uint32_t nbits; bool overflow;
    __device__ void add(uint32_t value, uint32_t nb) {
        uint32_t m = nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1);
        acc |= (uint64_t)(value & m) << nbits;
        nbits += nb;
        while (nbits >= 8) {
            if (len < cap) out[len++] = (uint8_t)acc; else overflow = true;
            acc >>= 8; nbits -= 8;
        }
    }
    __device__ void close() {
        add(1, 1);
        if (nbits > 0) {
            if (len < cap) out[len++] = (uint8_t)acc; else overflow = true;
            acc = 0; nbits = 0;
        }
    }
};

// v0: one lane encodes the whole stream (all N chains interleaved exactly as the reference
// emits them).  grid = units, block = 64.
__global__ void __launch_bounds__(64) k_enc_tans_serial(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (threadIdx.x != 0 || u.status != MICD_OK) return;
    if (u.nstates_used != 0) return;                                      // k_enc_tans_wg already wrote it
    if (u.nstates == 108) { u.status = MICD_ERR_UNSUPPORTED; return; }     // rANS encode exists only in k_enc_tans_wg
    const uint32_t n = u.ntok;
    const uint16_t *src = u.tok;
    const uint32_t tl = u.table_log;
    // Fallback chain of CompressSingleFrame{,4State,8State} (multiframecompress.go:15-93).
    for (int lanes = u.nstates; lanes >= 1; lanes >>= 1) {
        int rc = MICD_OK;
        // length gates: fse8state.go:32, fse4state.go:25, fse2state.go:23, fsecompressu16.go:20
        if (n <= (uint32_t)(lanes - 1) || n <= 1) rc = MICD_ERR_INCOMPRESSIBLE;
        else if (n <= 2 && lanes <= 2) rc = MICD_ERR_INTERNAL;          // "src too small"
        uint32_t pos = (lanes == 1) ? 0 : 6;
        uint32_t out_len = 0;
        if (rc == MICD_OK) {
            uint8_t *dst = u.blob + 6 + u.hdr_len;                     // bitstream right behind the NCount
            BitW bw; bw.out = dst; bw.cap = u.blob_cap - 6 - u.hdr_len; bw.len = 0; bw.acc = 0; bw.nbits = 0; bw.overflow = false;
            uint32_t st[8];
            for (int k = 0; k < lanes; k++) st[k] = 1u << tl;
            for (uint32_t ip = n; ip > 0; ip--) {
                uint32_t idx = ip - 1;
                uint32_t k = idx & (uint32_t)(lanes - 1);
                uint16_t sym = src[idx];
                uint32_t dnb = u.tt_nb[sym]; int32_t dfind = u.tt_find[sym];
                uint32_t nb = (st[k] + dnb) >> 16;                      // fsecompressu16.go:95-100
                bw.add(st[k], nb);
                st[k] = u.state_tab[(int32_t)(st[k] >> (nb & 31)) + dfind];
            }
            for (int k = lanes - 1; k >= 0; k--) bw.add(st[k], tl);     // final states, last lane first
            bw.close();
            if (bw.overflow) rc = u.tier == 1 ? MICD_INT_GROW : MICD_ERR_CAPACITY;
            else if ((uint64_t)u.hdr_len + bw.len >= (uint64_t)n * 2) rc = MICD_ERR_INCOMPRESSIBLE; // fse2state.go:58-60
            else out_len = pos + u.hdr_len + bw.len;
        }
        if (rc == MICD_OK) {
            if (lanes != 1) {
                u.blob[0] = 0xFF;
                u.blob[1] = lanes == 2 ? 0x02 : lanes == 4 ? 0x04 : 0x84;
                u.blob[2] = (uint8_t)n; u.blob[3] = (uint8_t)(n >> 8);
                u.blob[4] = (uint8_t)(n >> 16); u.blob[5] = (uint8_t)(n >> 24);
            }
            // 1-state streams have no prefix: the blob starts at u.blob + 6
            u.blob_len = out_len; u.nstates_used = lanes; u.status = MICD_OK;
            return;
        }
        u.status = rc;                                                  // error of the last attempt
        if (lanes == 1 || u.no_fallback) return;
        if (rc == MICD_ERR_CAPACITY || rc == MICD_INT_GROW) return;
        u.status = MICD_OK;                                             // try the next flavour
    }
}

// ------------------------------------------------------------------------------------------
// Parallel N-state tANS encode: one work-group of 1024 threads per unit.
//
// Chain k (k = index mod N) is a serial walk  state -> stateTable[(state >> nb) + dFind[sym]]
// from the last symbol to the first (fsecompressu16.go:95-100, fse2state.go:122-199).  After a
// symbol whose normalised count is v the state is one of only v table slots, so two walks over
// the same symbols started from different states merge after a few dozen symbols (measured:
// mean 40..120).  That makes the chain splittable, exactly:
//   1. every thread walks its segment of a chain from a guessed start state and records the
//      state in front of every symbol;
//   2. every thread re-walks the head of its segment from the true end state of the previous
//      segment until it meets its own recorded states; segments whose end state changed make
//      their successor repeat the step -- iterate to the fixed point (usually two rounds,
//      at worst one round per segment, which is the serial walk);
//   3. the bits of symbol i are (state & mask(nb), nb); a prefix sum of nb over the emission
//      order (last symbol first, bitwriter.go) gives every thread its bit offset, and each thread
//      packs its range into 32-bit words.  A word is stored by the thread that owns its first
//      bit; the threads that only reach into a word OR their bits in after a barrier.
// Then the final states (last lane first), the end mark, the size gate and the prefix bytes,
// and the N -> N/2 -> ... -> 1 fallback chain of multiframecompress.go:15-93.
// LDS: stateTable as u16 (state - 2^tl).  grid = units, block = 1024, dynamic LDS = 2 << tl.
#define TE_THREADS 512              // two groups per CU (128 VGPRs each): the fix-up stalls of one overlap the other's work
#define TE_WAVES 8
#ifndef TE_THREADS16
#define TE_THREADS16 1024         // threads of the tableLog-16 instance: its 128 KiB table leaves a CU one group whatever its size
#endif
#define TE_BLK 64                 // the largest block a walk reads at a time, in tokens (te_encode: BLK)
#ifndef TE_WARM_TOK
#define TE_WARM_TOK 128           // tokens of the predecessor's range a thread walks first (a multiple of 64)
#endif
#ifndef TE_NARROW_WPS
#define TE_NARROW_WPS 6          // waves per SIMD the two-state instance is compiled for (6: three groups per CU; it needs 64 registers)
#endif
#ifndef TE_BLK2
#define TE_BLK2 64                // tokens per block of the one- and two-state walks
#endif
#ifndef TE_ABL
#define TE_ABL 0         // timing-only ablations: 1 = every block read comes from the unit's first 4 KiB, 2 = the pack pass stores nothing (and is optimised away), 4 = it does its work and stores nothing
#endif
#define TE_ABL_BASE(b) ((TE_ABL & 1) ? ((b) & 2047u) : (b))
#define TE_TT_SYMS 4096           // alphabets up to this size keep their coding records in LDS (32 KiB)

// One block of B tokens as 16-byte vectors.
template <int B> struct TeBlkT { uint4 v[B / 8]; };
template <int B>
__device__ __forceinline__ TeBlkT<B> te_load(mic_gp<const uint16_t> p) {
    TeBlkT<B> b; const mic_gp<const tk_v4> q = (mic_gp<const tk_v4>)p;
#pragma unroll
    for (int i = 0; i < B / 8; i++) { const tk_v4 x = q[i]; b.v[i] = make_uint4(x.x, x.y, x.z, x.w); }
    return b;
}
template <int B>
__device__ __forceinline__ uint32_t te_get(const TeBlkT<B> &b, int j) {   // j compile-time after unrolling
    const uint4 &v = b.v[j >> 3];
    const uint32_t w = ((j >> 1) & 3) == 0 ? v.x : ((j >> 1) & 3) == 1 ? v.y : ((j >> 1) & 3) == 2 ? v.z : v.w;
    return (j & 1) ? (w >> 16) : (w & 0xFFFF);
}
template <int B>
__device__ __forceinline__ void te_set(TeBlkT<B> &b, int j, uint32_t x) {
    uint4 &v = b.v[j >> 3];
    uint32_t &w = ((j >> 1) & 3) == 0 ? v.x : ((j >> 1) & 3) == 1 ? v.y : ((j >> 1) & 3) == 2 ? v.z : v.w;
    w = (j & 1) ? ((w & 0xFFFFu) | (x << 16)) : ((w & 0xFFFF0000u) | (x & 0xFFFF));
}

// Thread t (t = 0 encodes first) owns the 32-token blocks [b_lo, b_hi); all N chains of those
// tokens are walked together (N independent LDS look-ups in flight), 64 bytes per memory access.
// TTL: the per-symbol records (symbolTT / rANS freq+bias) were copied to LDS (alphabets up to TE_TT_SYMS);
// otherwise every coding step gathers them from HBM.
#if defined(TE_INLINE)
#define TE_FN_ATTR __forceinline__
#elif defined(TE_NOINLINE)
#define TE_FN_ATTR __noinline__
#else
#define TE_FN_ATTR
#endif
#ifdef TE_BAR_VM        // diagnostic: every barrier of the walk also waits for the group's outstanding global memory operations
#define TE_SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
#else
#define TE_SYNC() __syncthreads()
#endif
template <int N, bool RANS, bool TTL, int T, bool FS>   // FS: the LDS state table holds whole states (they fit 16 bits up to tableLog 15)
__device__ TE_FN_ATTR void te_encode(MicUnit &u, uint16_t *s_stab, const uint2 *s_tt,
                          uint16_t *s_E, uint32_t *s_scan, int &rc_out, uint32_t &total_bytes_out) {   // s_E: T x N end states
    // tokens per block: the two-state walk (the usual flavour) reads a whole 128-byte line at a time (two 64-byte halves read apart were two
    // fetches: the line does not survive in a cache between them); the wider walks have no registers for that
#ifdef TE_BLK1_64   // diagnostic: the one-state walk on whole lines too (the state of the code when round 3 saw the differences)
    constexpr int BLK = (N <= 2 && !RANS && TTL) ? TE_BLK2 : 32;
#else
    constexpr int BLK = (N == 2 && !RANS && TTL) ? TE_BLK2 : 32;
#endif
    constexpr uint32_t RGRP = 128 / BLK, WARM = TE_WARM_TOK / BLK;   // blocks per fix-up record (128 tokens); blocks of the predecessor's range walked as warm-up
    typedef TeBlkT<BLK> TeBlk;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = u.ntok, tl = u.table_log, size = 1u << tl;
    const mic_gp<const uint16_t> src = mic_g((const uint16_t *)u.tok);
    const mic_gp<const uint32_t> tt_nb = mic_g((const uint32_t *)u.tt_nb); const mic_gp<const int32_t> tt_find = mic_g((const int32_t *)u.tt_find);
    // the coding record of a symbol: {deltaNbBits, deltaFindState} (rANS: {freq | k0 << 20, bias}) -- one 8-byte LDS read when the
    // alphabet's records were copied there, two gathers from HBM otherwise
    auto rec = [&](uint32_t sy) -> uint2 { return TTL ? s_tt[sy] : make_uint2(tt_nb[sy], (uint32_t)tt_find[sy]); };
    const mic_gp<uint16_t> stv = mic_g(u.sym);
    const uint32_t hdr_len = u.hdr_len;
    const mic_gp<uint8_t> bits_base = mic_g(u.blob) + 6 + hdr_len;
    const uint32_t lead = (uint32_t)((uintptr_t)bits_base & 15);          // the bit grid starts at a 16-byte boundary: the pack pass stores pairs of 64-bit units
    const mic_gp<uint32_t> words = (mic_gp<uint32_t>)(bits_base - lead);
    const uint32_t words_cap = (u.blob_cap - 6 - hdr_len - 8) / 4;
    const uint32_t nblk = (n + BLK - 1) / BLK;
    const uint32_t per = (nblk + T - 1) / T;
    const uint32_t b_hi = (tid * per < nblk) ? nblk - tid * per : 0;
    const uint32_t b_lo = (b_hi > per) ? b_hi - per : 0;
    // ---- 1. speculative walk from the guessed state 2^tl --------------------------------------
    // One coding step.  tANS: cStateU16.encode (fsecompressu16.go:95-100), state in [2^tl, 2^(tl+1)).
    // rANS: ransEncodeStep (ransu16.go:187-197) on xL = x + 2^tl with the per-symbol record
    // tt_nb = freq | k0 << 20, tt_find = bias; both emit (state & mask(nb), nb) and both forget
    // the old state down to one of `freq` values, which is what makes the walks merge.
    auto step = [&](uint32_t state, uint32_t sy, uint32_t &nb_out) -> uint32_t {
        const uint2 r = rec(sy);
        if (RANS) {
            const uint32_t e = r.x, freq = e & 0xFFFFF, k0 = e >> 20;
            const uint32_t k = k0 - ((state < (freq << k0)) ? 1u : 0u);
            nb_out = k;
            return size + r.y + ((state >> k) - freq);
        } else {
            const uint32_t nb = (state + r.x) >> 16;
            nb_out = nb;
            return (FS ? 0u : size) + s_stab[(int32_t)(state >> nb) + (int32_t)r.y];
        }
    };
    MIC_STAMP_BEGIN();
    // Per group of RGRP blocks (128 tokens) the walk leaves a record in HBM: the N states after the group
    // (u16, minus 2^tl) and the bits the group emits, one aligned vector store.  Fix-up compares and replaces
    // records; packing needs none of them (it walks again from the true start states), so tokens are read
    // twice and states never stored.  Record slot = tid * gper + group index inside the thread's range.
    constexpr uint32_t RWP = (N == 1) ? 2u : 2u * N;       // u16 per record: N states, the bit count, padding
    const uint32_t gper = (per + RGRP - 1) / RGRP;
    const mic_gp<uint32_t> rec32 = (mic_gp<uint32_t>)stv + (size_t)tid * gper * (RWP / 2);
    auto rec_store = [&](uint32_t g, const uint32_t (&stw)[N], uint32_t bits) {
        uint32_t w[RWP / 2];
#pragma unroll
        for (uint32_t i = 0; i < RWP / 2; i++) w[i] = 0;
#pragma unroll
        for (int k = 0; k < N; k++) w[k >> 1] |= ((stw[k] - size) & 0xFFFFu) << (16 * (k & 1));
        w[N >> 1] |= (bits & 0xFFFFu) << (16 * (N & 1));
#pragma unroll
        for (uint32_t i = 0; i < RWP / 2; i++) rec32[(size_t)g * (RWP / 2) + i] = w[i];
    };
    auto rec_load = [&](uint32_t g, uint32_t (&stw)[N], uint32_t &bits) {
        uint32_t w[RWP / 2];
#pragma unroll
        for (uint32_t i = 0; i < RWP / 2; i++) w[i] = rec32[(size_t)g * (RWP / 2) + i];
#pragma unroll
        for (int k = 0; k < N; k++) stw[k] = ((w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) + size;
        bits = (w[N >> 1] >> (16 * (N & 1))) & 0xFFFFu;
    };
    auto walk_block = [&](uint32_t base, uint32_t (&stw)[N]) -> uint32_t {
        const TeBlk tk = te_load<BLK>(src + TE_ABL_BASE(base));
        uint32_t bits = 0;
        if (base + BLK <= n) {                                   // every block but the stream's last: no per-token bound
#pragma unroll
            for (int j = BLK - 1; j >= 0; j--) {
                const int k = j & (N - 1);
                uint32_t nb;
                stw[k] = step(stw[k], te_get(tk, j), nb);
                bits += nb;
                if (BLK > 32 && (j & 7) == 0) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = BLK - 1; j >= 0; j--) {
                if (base + (uint32_t)j < n) {
                    const int k = j & (N - 1);
                    uint32_t nb;
                    stw[k] = step(stw[k], te_get(tk, j), nb);
                    bits += nb;
                }
            }
        }
        return bits;
    };
    uint32_t st[N];
#pragma unroll
    for (int k = 0; k < N; k++) st[k] = size;        // tANS: 1 << tl; rANS: x = 0, kept as xL = x + 2^tl
    // Warm-up: walks merge within ~100 symbols, so walking the last WARM blocks of the predecessor's range first
    // (nothing recorded) almost always lands on the true start state, and the fix-up below only has to confirm it.
    if (tid > 0 && b_hi > b_lo) {
        const uint32_t w_hi = min(b_hi + WARM, nblk);
        for (uint32_t b = w_hi; b > b_hi; b--) (void)walk_block((b - 1) * BLK, st);
    }
    uint32_t assumed[N];
#pragma unroll
    for (int k = 0; k < N; k++) assumed[k] = st[k];
    uint32_t mybits = 0;
    for (uint32_t g = 0, b = b_hi; b > b_lo; g++) {
        uint32_t bits = 0;
        for (uint32_t r = 0; r < RGRP && b > b_lo; r++, b--) bits += walk_block((b - 1) * BLK, st);
        mybits += bits;
        rec_store(g, st, bits);
    }
#pragma unroll
    for (int k = 0; k < N; k++) s_E[(tid) * N + k] = (uint16_t)(st[k] - size);
    MIC_STAMP_AT(u, 8);
    // ---- 2. fix-up rounds to the fixed point -----------------------------------------------------
    for (uint32_t round = 0; round < T; round++) {
        TE_SYNC();
        int changed = 0;
        uint32_t e_out[N], st2[N]; bool any = false;
#pragma unroll
        for (int k = 0; k < N; k++) {
            e_out[k] = (uint32_t)s_E[(tid) * N + k] + size;
            const uint32_t e_prev = (tid > 0) ? (uint32_t)s_E[(tid - 1) * N + k] + size : size;
            if (tid > 0 && e_prev != assumed[k]) { assumed[k] = e_prev; any = true; }
            st2[k] = e_prev;
        }
        // (a thread without tokens -- the tail of the group when the blocks do not divide by it -- takes no part: handing the end
        // states on through those threads was a round each, 8 to 22 rounds on an XR strip instead of one -- cheap ones, three
        // barriers and no walk, nothing the kernel's time shows; the trailer takes the states from the last thread that owns tokens,
        // below.  What the fix-up costs is its ONE round: merge lengths halve every ~128 tokens, the slowest of 512 threads walks
        // ~1000 tokens again, and its wave -- and, at the barrier, the group -- waits for it: stamps walk 265 k, fix-up 300 k,
        // pack 640 k ticks.  A longer warm-up moves that time into the walk, token for token: tools/stamp_enc.py, TE_WARM_TOK.)
        if (any && b_hi > b_lo) {
            bool merged = false;
#ifdef MIC_STAMP
            atomicAdd(&u.dbg[13], 1u);                                    // threads that re-walk, all rounds
#endif
            for (uint32_t g = 0, b = b_hi; b > b_lo && !merged; g++) {
#ifdef MIC_STAMP
                atomicMax(&u.dbg[14], g + 1);                             // longest re-walk in groups
#endif
                uint32_t bits = 0;
                for (uint32_t r = 0; r < RGRP && b > b_lo; r++, b--) bits += walk_block((b - 1) * BLK, st2);
                uint32_t old_st[N], old_bits;
                rec_load(g, old_st, old_bits);
                merged = true;
#pragma unroll
                for (int k = 0; k < N; k++) if (old_st[k] != st2[k]) merged = false;
                mybits += bits - old_bits;
                rec_store(g, st2, bits);
            }
            if (!merged) {                                   // ran to the end of the range: hand the states on
#pragma unroll
                for (int k = 0; k < N; k++) if (st2[k] != e_out[k]) { e_out[k] = st2[k]; changed = 1; }
            }
        }
        TE_SYNC();                                   // every thread has read its predecessor's states
#pragma unroll
        for (int k = 0; k < N; k++) s_E[(tid) * N + k] = (uint16_t)(e_out[k] - size);
#ifdef MIC_STAMP
        if (tid == 0) u.dbg[12] += 1;                                     // rounds
#endif
        if (!__syncthreads_or(changed)) break;
    }
    TE_SYNC();
    // empty tail threads may not have been reached by a round: propagate the final states down
    // (thread T-1 must hold the end states of every chain for the trailer)
    if (tid == 0) {
        // last thread that owns tokens
        uint32_t last = (nblk + per - 1) / per; if (last > 0) last--;
        for (int k = 0; k < N; k++) s_E[(T - 1) * N + k] = s_E[(last) * N + k];
    }
    __threadfence_block();
    TE_SYNC();
#ifdef TE_CHECK     // diagnostic: the walk's invariants, reported through u.dbg[15] (the kernel turns it into a status)
    {
        const uint32_t lastown_c = (nblk + per - 1) / per; uint32_t bad = 0;
        if (tid > 0 && tid < lastown_c) for (int k = 0; k < N; k++) if (assumed[k] != (uint32_t)s_E[(tid - 1) * N + k] + size) bad |= 1u;
        if (bad) { atomicOr(&u.dbg[15], bad); atomicMin(&u.dbg[14], tid | 0x10000u); }
    }
#endif
    MIC_STAMP_AT(u, 9);
    // ---- 3. bit offsets ---------------------------------------------------------------------------
    const uint32_t incl = tk_wave_incl_add(mybits, lane);
    if (lane == 63) s_scan[wave] = incl;
    TE_SYNC();
    uint32_t woff = 0, sym_bits = 0;
#pragma unroll
    for (int wv = 0; wv < (T / 64); wv++) { const uint32_t v = s_scan[wv]; if ((uint32_t)wv < wave) woff += v; sym_bits += v; }
    const uint64_t gstart = 8ull * lead + woff + incl - mybits;            // first grid bit of this thread
    const uint64_t total_bits = (uint64_t)sym_bits + (uint64_t)N * tl + 1;
    const uint32_t total_bytes = (uint32_t)((total_bits + 7) >> 3);
#ifdef MIC_GATE_REG
    // diagnostic build: every thread forms the verdict from its own registers (the variant that misbehaved in round 1)
    int rc_reg = MICD_OK;
    if ((8ull * lead + total_bits + 255) / 32 >= words_cap) rc_reg = MICD_ERR_CAPACITY;
    else if ((uint64_t)hdr_len + total_bytes >= (uint64_t)n * 2) rc_reg = MICD_ERR_INCOMPRESSIBLE;
#ifdef MIC_GATE_REG2
    if (rc_reg == MICD_ERR_CAPACITY) { total_bytes_out = total_bytes; rc_out = rc_reg; return; }   // (the verdict itself is used after the pack loop)
#else
    total_bytes_out = total_bytes; rc_out = rc_reg;
    if (rc_reg != MICD_OK) return;
#endif
    const int rc = rc_reg; (void)rc;
#else
    // the verdict goes through LDS so that every thread (and the caller) sees one value
    if (tid == 0) {
        int rc0 = MICD_OK;
        if ((8ull * lead + total_bits + 255) / 32 >= words_cap) rc0 = MICD_ERR_CAPACITY;
        else if ((uint64_t)hdr_len + total_bytes >= (uint64_t)n * 2) rc0 = MICD_ERR_INCOMPRESSIBLE;   // fse2state.go:58-60
        s_scan[(T / 64)] = (uint32_t)rc0; s_scan[(T / 64) + 1] = total_bytes;
    }
    TE_SYNC();
    const int rc = (int)s_scan[(T / 64)];
    total_bytes_out = s_scan[(T / 64) + 1];
    rc_out = rc;
    if (rc != MICD_OK) return;
#endif
    MIC_STAMP_AT(u, 10);
    // ---- 4. pack -------------------------------------------------------------------------------------
    // 64-bit units, stored in aligned PAIRS (16 bytes): the memory counters charge every store instruction a 32-byte write whatever
    // its width or its neighbours (1.74 GB of output as dword stores = 435 M x 32 bytes = 13.9 GB of WRITE_SIZE, measured in two
    // orders of the same stores; as 64-bit stores 8.3 GB and 6.4 -> 5.2 ms), so the pass makes a quarter as many.  A unit is stored by
    // the thread owning its first bit; the threads that only reach into it OR their bits in after a barrier.
    const mic_gp<unsigned long long> words64 = (mic_gp<unsigned long long>)words;
    const uint32_t first_q = (uint32_t)(gstart >> 6);
    const bool own_first = (gstart & 63) == 0;
    uint32_t q = first_q;
    uint64_t acc = 0; uint32_t filled = (uint32_t)(gstart & 63);
    uint64_t lead_val = 0; bool have_lead = false;
    typedef unsigned long long te_u2 __attribute__((ext_vector_type(2)));
    uint64_t abl_chk = 0;
#ifndef TE_RING
#define TE_RING 4                 // units of a store group (a power of two): 4 = 32 bytes, a memory sector
#endif
    uint64_t rg[TE_RING]; uint32_t rmask = 0;                            // the units of the current group that are waiting (the last slot: the one that completes it)
#pragma unroll
    for (int i = 0; i < TE_RING; i++) rg[i] = 0;
    auto flush4 = [&](uint32_t base, uint64_t vlast, bool have_last) {    // base: the group's first unit (a multiple of TE_RING)
        const uint32_t m = rmask | (have_last ? 1u << (TE_RING - 1) : 0u);
        rg[TE_RING - 1] = vlast;
#pragma unroll
        for (int i = 0; i < TE_RING; i += 2) {
            const uint32_t mm = (m >> i) & 3u;
            if (mm == 3u) { te_u2 pr; pr.x = rg[i]; pr.y = rg[i + 1]; *(mic_gp<te_u2>)(words64 + base + i) = pr; }
            else if (mm & 1u) words64[base + i] = rg[i];
            else if (mm & 2u) words64[base + i + 1] = rg[i + 1];
        }
    };
    auto emit = [&](uint64_t v) {                                          // unit q is complete (or the thread's last, partial one)
        if (TE_ABL & 2) { q++; return; }
        if (TE_ABL & 4) { abl_chk ^= v; q++; return; }                          // (timing only: the pass without its stores)
        if (q == first_q && !own_first) { lead_val = v; have_lead = true; }
        // Units leave in aligned groups of FOUR (32 bytes, two 16-byte stores back to back): the memory side writes 32-byte sectors,
        // and a lone 16-byte store 1.4 KiB from its lane neighbours' left a sector half written when its line was evicted -- the
        // counters showed the stream written 2.75 times.
        else if ((q & (TE_RING - 1u)) == TE_RING - 1u) { flush4(q - (TE_RING - 1u), v, true); rmask = 0; }
        else {
            const uint32_t k = q & (TE_RING - 1u);
#pragma unroll
            for (int i = 0; i < TE_RING - 1; i++) rg[i] = k == (uint32_t)i ? v : rg[i];
            rmask |= 1u << k;
        }
        q++;
    };
#ifdef TE_CHECK
    uint32_t pbits = 0;
#endif
    {
        uint32_t stp[N];
        const uint32_t lastown = (nblk + per - 1) / per;   // threads 0 .. lastown-1 own tokens; s_E[T-1] was overwritten for the trailer
#pragma unroll
        for (int k = 0; k < N; k++) stp[k] = (tid > 0 && tid < lastown) ? (uint32_t)s_E[(tid - 1) * N + k] + size : size;
        // Four tokens (<= 64 bits) are gathered branch-free before they meet the accumulator: the test "does a 64-bit unit fill up"
        // is a divergent branch that some lane takes at almost every token, so it is made once per four of them.
        for (uint32_t b = b_hi; b > b_lo; b--) {
            const uint32_t base = (b - 1) * BLK;
            const TeBlk tk = te_load<BLK>(src + base);
            const bool whole = base + BLK <= n;
#pragma unroll
            for (int j4 = BLK / 4 - 1; j4 >= 0; j4--) {
                uint64_t t4 = 0; uint32_t f4 = 0;
#pragma unroll
                for (int jj = 3; jj >= 0; jj--) {
                    const int j = j4 * 4 + jj;
                    if (whole || base + (uint32_t)j < n) {
                        const int k = j & (N - 1);
                        const uint32_t stt = stp[k];
                        uint32_t nb;
                        stp[k] = step(stt, te_get(tk, j), nb);
                        const uint32_t bv = __builtin_amdgcn_ubfe(stt, 0u, nb);   // nb <= 16
                        t4 |= (uint64_t)bv << f4;                                // f4 <= 48 here
                        f4 += nb;
                    }
                }
#ifdef TE_CHECK
                pbits += f4;
#endif
                acc |= t4 << filled;                                             // filled < 64
                const uint32_t nf = filled + f4;
                if (nf >= 64) {
                    emit(acc);
                    acc = filled ? (t4 >> (64u - filled)) : 0ull;                // what did not fit
                    filled = nf - 64;
                } else filled = nf;
            }
        }
#ifdef TE_CHECK
        {
            uint32_t bad = pbits != mybits ? 2u : 0u;
            if (tid < lastown && tid != T - 1) for (int k = 0; k < N; k++) if (stp[k] != (uint32_t)s_E[tid * N + k] + size) bad |= 4u;
            if (bad) { atomicOr(&u.dbg[15], bad); atomicMin(&u.dbg[13], tid | 0x10000u); }
        }
#endif
    }
    if (mybits > 0 && filled > 0) emit(acc);                             // the thread's last, partial unit
    if ((TE_ABL & 4) && abl_chk == 0x1234567ull) words64[first_q] = abl_chk;
    if (rmask) flush4((q - 1u) & ~(TE_RING - 1u), 0ull, false);           // (a group the next thread completes)
#ifdef TE_NO_HANDOFF    // diagnostic: the barrier round 3 had here before MIC_GROUP_HANDOFF
    __threadfence_block(); __syncthreads();
#else
    MIC_GROUP_HANDOFF();                                   // (every unit is in L2 before anything is OR-ed into it)
#endif
    if (have_lead && lead_val) (void)__hip_atomic_fetch_or(&words64[first_q], (unsigned long long)lead_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_block();
    TE_SYNC();
    if (tid == 0) {
        // final states, last lane first (fse2state.go:194-197), then the end mark.  The word the symbols' last bit lies in is shared
        // with the atomic ORs other threads have just made (executed at L2, nothing here waits for them): the trailer ORs its bits in
        // the same way instead of reading the word back; the words wholly behind that bit are zeroed first and the zeroes are
        // acknowledged before the first OR.
        uint64_t pos = 8ull * lead + sym_bits;
        const uint64_t end = pos + (uint64_t)N * tl + 1;
        for (uint64_t ww = (pos + 31) >> 5; ww <= ((end - 1) >> 5); ww++) words[ww] = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TE_OLD_TRAILER   // diagnostic: round 3's trailer, a plain read-modify-write of the shared word
        auto or32 = [&](uint32_t wi, uint32_t bits) { words[wi] |= bits; };
#else
        auto or32 = [&](uint32_t wi, uint32_t bits) { if (bits) (void)__hip_atomic_fetch_or(&words[wi], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
#endif
        for (int k = N - 1; k >= 0; k--) {
            const uint64_t v = (uint64_t)((uint32_t)s_E[(T - 1) * N + k] + size) & (((uint64_t)1 << tl) - 1);  // addBits32NC(state, tl)
            const uint32_t wi = (uint32_t)(pos >> 5), sh = (uint32_t)(pos & 31);
            or32(wi, (uint32_t)(v << sh));
            if (sh + tl > 32) or32(wi + 1, (uint32_t)(v >> (32 - sh)));
            pos += tl;
        }
        or32((uint32_t)(pos >> 5), 1u << (pos & 31));                   // bitwriter.go:162-168
    }
#ifdef MIC_GATE_REG2
    total_bytes_out = total_bytes; rc_out = rc_reg;
#endif
    MIC_STAMP_AT(u, 11);
}

// Small units get a group of ONE wave (T = 64, a 1024-symbol coding-record area: 24 KiB of LDS, six groups per CU): cut into 512 ranges
// a 35 k-token plane leaves 70 tokens per thread, walks from different states do not merge inside such a range and every correction
// ripples on at one barrier round per range (140 rounds, 80 % of the kernel on MIC3 planes); 64 ranges of 550 tokens merge, and
// six units per CU instead of two share it.
#define TE_SMALL_NTOK 98304u
#define TE_SMALL_SYMS 1024u
// class of a unit: 0 = the 512-thread instance, 1 = one wave, tableLog <= 12 and <= 512 symbols (12 KiB of LDS: thirteen groups per CU --
// 8-bit planes), 2 = one wave, the other small units (24 KiB: six per CU)
__device__ __forceinline__ int te_small_class(const MicUnit &u) {
    if (u.ntok > TE_SMALL_NTOK || u.table_log > 13 || u.symbol_len > TE_SMALL_SYMS) return 0;
    return (u.table_log <= 12 && u.symbol_len <= 512u) ? 1 : 2;
}
// Dynamic LDS: the state table (2 << TLHI bytes), TTS coding records of 8 bytes, T x e_states end states of 2 bytes.  e_states is the
// widest flavour a unit of the launch may ask for (2 when the whole batch is two-state: the 512-thread instance then takes 50 KiB
// and THREE groups share a CU -- the kernel waits on LDS round trips most of its time, SQ_WAIT_ANY 0.73 of its wave cycles --
// at 80 VGPRs; 8 otherwise).
// WIDTH (the 512-thread instance up to tableLog 13 only): 1 = units that ask for two states and whose coding records fit the LDS:
// ONE walk in the kernel, two groups per CU, no scratch; 2 = every other unit (one, four, eight states, rANS, alphabets past 4096
// symbols) and the two-state units whose first attempt failed (they start over there), compiled for one group per CU.  One instance for both had the wide walks spill 60 registers at 80 and
// charged the two-state batches a 364-byte scratch frame per lane.  0 = every unit (the other instances: registers are not tight).
template <int TLHI, int T, int TTS, int WIDTH = 0>      // table-size class of the launch: tableLog <= 13, 14, 15 or 16; threads; LDS coding records
// Waves per SIMD the instance is compiled for = what its LDS lets a CU hold: three 512-thread groups up to tableLog 13 (80 VGPRs:
// the two-state walk needs 60; the four- and eight-state walks spill there), two at 14, one at 15 and 16 (no register limit worth
// the name: the WaveletV2 four-state walk at tableLog 16 used to run on the 80 registers of the first class, 47 of them spilled).
__global__ void __launch_bounds__(T, T != 512 ? 4 : TLHI <= 13 ? (WIDTH == 2 ? 2 : TE_NARROW_WPS) : 2) k_enc_tans_wg(MicUnit *units, int e_states) {
    constexpr uint32_t tl_lo = (TLHI <= 13) ? 5u : (uint32_t)TLHI, tl_hi = (uint32_t)TLHI;
    extern __shared__ __attribute__((aligned(16))) uint16_t s_stab[];
    uint16_t *const s_E = s_stab + (1u << TLHI) + (TLHI <= 15 ? (uint32_t)TTS * 4u : 0u);
    __shared__ uint32_t s_scan[(T / 64) + 2];
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK) return;
    const bool handed = u.nstates_used == -2;                              // the two-state instance gave up on its first attempt (below)
    if (u.nstates_used != 0 && !(WIDTH == 2 && handed)) return;
#ifdef TE_COMBINED   // diagnostic build: the round-3 layout in which one-state streams differed from call to call
    if (WIDTH != 0 && ((u.nstates > 2 || u.symbol_len > TTS || handed) ? 2 : 1) != WIDTH) return;
#else
    if (WIDTH != 0 && ((u.nstates != 2 || u.symbol_len > TTS || handed) ? 2 : 1) != WIDTH) return;   // (1: two states, coding records in LDS)
#endif
    if ((u.nstates == 108 ? 8 : (int)u.nstates) > e_states) { if (threadIdx.x == 0) u.status = MICD_ERR_INTERNAL; return; }   // (the launcher sizes the LDS for e_states)
    const uint32_t tl = u.table_log;
    if (tl < tl_lo || tl > tl_hi) return;
    if (TLHI <= 13 && te_small_class(u) != (T == 64 ? (TLHI == 12 ? 1 : 2) : 0)) return;   // (small units: the one-wave instances)
    const uint32_t tid = threadIdx.x;
    const uint32_t n = u.ntok;
    const uint32_t size = 1u << tl;
    if (u.sym_cap < ((n + TE_BLK - 1) / TE_BLK) * TE_BLK) { if (tid == 0) u.status = u.tier == 1 ? MICD_INT_GROW : MICD_ERR_CAPACITY; return; }
    if (u.nstates != 108) { const mic_gp<const uint32_t> gst = mic_g((const uint32_t *)u.state_tab); for (uint32_t i = tid; i < size; i += T) s_stab[i] = (uint16_t)(gst[i] - (TLHI <= 15 ? 0u : size)); }
    uint2 *s_tt = (uint2 *)(s_stab + (1u << TLHI));                              // TTS coding records, 8 bytes each
    const bool ttl = TLHI <= 15 && u.symbol_len <= TTS;
    if (ttl) {
        const mic_gp<const uint32_t> gnb = mic_g((const uint32_t *)u.tt_nb); const mic_gp<const int32_t> gfi = mic_g((const int32_t *)u.tt_find);
        for (uint32_t i = tid; i < u.symbol_len; i += T) s_tt[i] = make_uint2(gnb[i], (uint32_t)gfi[i]);
    }
    __syncthreads();
    const uint32_t hdr_len = u.hdr_len;
    const bool rans = u.nstates == 108;                                          // rans8state.go: 8 lanes, magic FF 08
    const bool single = u.no_fallback != 0;
    for (uint32_t lanes = rans ? 8u : u.nstates; lanes >= 1; lanes >>= 1) {
        // length gates: fse8state.go:32, fse4state.go:25, fse2state.go:23, fsecompressu16.go:20, rans8state.go:32
        int rc = MICD_OK;
        if (n <= lanes - 1 || n <= 1) rc = MICD_ERR_INCOMPRESSIBLE;
        else if (n <= 2 && lanes <= 2) rc = MICD_ERR_INTERNAL;                   // "src too small"
        uint32_t total_bytes = 0;
        if (rc == MICD_OK) {
            if (TLHI <= 15 && ttl) {
                if (WIDTH != 1 && rans) te_encode<8, true, true, T, (TLHI <= 15)>(u, s_stab, s_tt, s_E, s_scan, rc, total_bytes);
                else if (WIDTH != 1 && lanes == 8) te_encode<8, false, true, T, (TLHI <= 15)>(u, s_stab, s_tt, s_E, s_scan, rc, total_bytes);
                else if (WIDTH != 1 && lanes == 4) te_encode<4, false, true, T, (TLHI <= 15)>(u, s_stab, s_tt, s_E, s_scan, rc, total_bytes);
                else if (lanes == 2) te_encode<2, false, true, T, (TLHI <= 15)>(u, s_stab, s_tt, s_E, s_scan, rc, total_bytes);
#ifndef TE_COMBINED
                else if (WIDTH == 1) rc = MICD_ERR_INTERNAL;                            // (not reached: the two-state instance hands its fall-backs over)
#endif
                else te_encode<1, false, true, T, (TLHI <= 15)>(u, s_stab, s_tt, s_E, s_scan, rc, total_bytes);
            } else {
                if (WIDTH == 1) rc = MICD_ERR_INTERNAL;                                 // (not reached: such units go to the wide instance)
                else if (rans) te_encode<8, true, false, T, (TLHI <= 15)>(u, s_stab, s_tt, s_E, s_scan, rc, total_bytes);
                else if (lanes == 8) te_encode<8, false, false, T, (TLHI <= 15)>(u, s_stab, s_tt, s_E, s_scan, rc, total_bytes);
                else if (lanes == 4) te_encode<4, false, false, T, (TLHI <= 15)>(u, s_stab, s_tt, s_E, s_scan, rc, total_bytes);
                else if (lanes == 2) te_encode<2, false, false, T, (TLHI <= 15)>(u, s_stab, s_tt, s_E, s_scan, rc, total_bytes);
                else te_encode<1, false, false, T, (TLHI <= 15)>(u, s_stab, s_tt, s_E, s_scan, rc, total_bytes);
            }
        }
        __syncthreads();
#ifdef TE_CHECK
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
            const uint32_t d = __hip_atomic_load(&u.dbg[15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d) { if (tid == 0) u.status = -40 - (int)d; return; }
        }
#endif
        if (rc == MICD_OK) {
            if (tid == 0) {
                if (lanes != 1) {
                    u.blob[0] = 0xFF;
                    u.blob[1] = rans ? 0x08 : lanes == 2 ? 0x02 : lanes == 4 ? 0x04 : 0x84;
                    u.blob[2] = (uint8_t)n; u.blob[3] = (uint8_t)(n >> 8);
                    u.blob[4] = (uint8_t)(n >> 16); u.blob[5] = (uint8_t)(n >> 24);
                }
                u.blob_len = ((lanes == 1) ? 0 : 6) + hdr_len + total_bytes;      // 1-state blob starts at blob + 6
                u.nstates_used = (int32_t)lanes;
                u.status = MICD_OK;
            }
            return;
        }
        if (tid == 0) { u.count = (uint32_t)rc; u.bits_off = total_bytes; u.flavour = lanes; }   // probe: why the attempt failed
        if (rc == MICD_ERR_CAPACITY && u.tier == 1) rc = MICD_INT_GROW;               // (the staging blob is tier 1's)
        if (lanes == 1 || rc == MICD_ERR_CAPACITY || rc == MICD_INT_GROW || single) { if (tid == 0) u.status = rc; return; }
#ifndef TE_COMBINED
        if (WIDTH == 1) { if (tid == 0) u.nstates_used = -2; return; }        // the one-state attempt is the wide instance's (it starts over)
#endif
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Compaction: copy every unit's staging blob to its final offset.  grid = (chunks, units).
// dst_off[i] = byte offset of unit i in `dst` (exclusive scan of blob_len, done by k_scan_lens).
__global__ void __launch_bounds__(256) k_enc_pack(const MicUnit *units, const uint64_t *dst_off, uint8_t *dst, uint64_t cap, int n) {
    const MicUnit &u = units[blockIdx.y];
    if (u.status != MICD_OK) return;
    if (dst_off[n] + 16 > cap) return;                                   // (the batch does not fit: the host packs it again into a larger buffer)
    const mic_gp<const uint8_t> src = mic_g((const uint8_t *)u.blob) + (u.nstates_used == 1 ? 6 : 0);
    const mic_gp<uint8_t> d = mic_g(dst) + dst_off[blockIdx.y];
    // 16 bytes per thread and step; neither side is aligned (gfx950 runs vector memory in unaligned mode)
    typedef uint32_t pk_v4 __attribute__((ext_vector_type(4)));
    typedef pk_v4 PkQ __attribute__((aligned(1)));
    const uint32_t len = u.blob_len, nvec = len / 16;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gridDim.x * blockDim.x)
        *(mic_gp<PkQ>)(d + (size_t)i * 16) = *(mic_gp<const PkQ>)(src + (size_t)i * 16);
    if (blockIdx.x == 0 && threadIdx.x < (len & 15u)) d[nvec * 16 + threadIdx.x] = src[nvec * 16 + threadIdx.x];
}

// exclusive scan of blob lengths (single block; n units <= a few 100k)
__global__ void __launch_bounds__(1024) k_scan_lens(const MicUnit *units, int n, uint64_t *dst_off) {
    __shared__ uint64_t s_part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    uint64_t sum = 0;
    for (int i = lo; i < hi; i++) sum += (units[i].status == MICD_OK) ? units[i].blob_len : 0;
    s_part[t] = sum;
    __syncthreads();
    if (t == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; i++) { uint64_t v = s_part[i]; s_part[i] = run; run += v; }
        dst_off[n] = run;
    }
    __syncthreads();
    uint64_t run = s_part[t];
    for (int i = lo; i < hi; i++) {
        dst_off[i] = run;
        run += (units[i].status == MICD_OK) ? units[i].blob_len : 0;
    }
}

// The histogram slab goes back to zero behind the chain (the session keeps it zero between calls instead of clearing 256 KiB per unit
// in front of every one: 17 GB for a 32768^2 slide's 65 535 planes): everything below the tokeniser's bound, or all of it.
__global__ void __launch_bounds__(256) k_enc_hist_clean(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    const uint32_t hi = min(u.tab_cap, (u.hist_hi >= 1 && u.hist_hi <= MIC_MAXSYM) ? u.hist_hi : MIC_MAXSYM + 1u);
    uint4 *h = (uint4 *)u.hist;                                           // (a slab is a multiple of 32 KiB)
    for (uint32_t i = threadIdx.x; i < (hi + 3) / 4; i += 256) h[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ------------------------------------------------------------------------------------------
// launchers
void mic_launch_encode(MicUnit *d_units, int n, hipStream_t stream, int variant, MicTimer *t, uint32_t enc_mask) {
    const bool any_grad = (variant & MIC_VARIANT_GRAD) != 0, frames = (variant & MIC_VARIANT_FRAMES) != 0;
    if (t) t->mark("k_enc_symbols");
    if (!frames) hipLaunchKernelGGL(k_enc_symbols, dim3(64, n), dim3(256), 0, stream, d_units);
    if (t) t->mark("k_enc_tokens_wg");
    hipLaunchKernelGGL(k_enc_tokens_wg<0>, dim3(n), dim3(TK_THREADS), 0, stream, d_units);
    if (!frames) hipLaunchKernelGGL(k_enc_tokens_wg<1>, dim3(n), dim3(TK_THREADS), 0, stream, d_units);
    if (any_grad) hipLaunchKernelGGL((k_enc_tokens_wg<0, 1>), dim3(n), dim3(TK_THREADS), 0, stream, d_units);
    if (t) t->mark("k_enc_tables_wg");
    mic_launch_enc_tables(d_units, n, stream);
    static MicPerDeviceOnce once;
    once.run([] {
        (void)hipFuncSetAttribute((const void *)k_enc_tans_wg<13, TE_THREADS, TE_TT_SYMS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        (void)hipFuncSetAttribute((const void *)k_enc_tans_wg<13, TE_THREADS, TE_TT_SYMS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        (void)hipFuncSetAttribute((const void *)k_enc_tans_wg<14, TE_THREADS, TE_TT_SYMS>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        (void)hipFuncSetAttribute((const void *)k_enc_tans_wg<15, TE_THREADS, TE_TT_SYMS>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        (void)hipFuncSetAttribute((const void *)k_enc_tans_wg<16, TE_THREADS16, TE_TT_SYMS>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    });
    const int es = (variant & MIC_VARIANT_NARROW) ? 2 : 8;                      // widest flavour in the batch -> end-state area
    const unsigned eb = TE_THREADS * (unsigned)es * 2u, eb1 = 64u * (unsigned)es * 2u;
    if (t) t->mark("k_enc_tans_wg<13>");
    // (classes the session has not seen lately are not launched; k_enc_tans_serial below takes whatever that leaves: mic_launch.h)
    if (enc_mask & MIC_ENC_CLS_NARROW2) hipLaunchKernelGGL((k_enc_tans_wg<13, TE_THREADS, TE_TT_SYMS, 1>), dim3(n), dim3(TE_THREADS), (2u << 13) + TE_TT_SYMS * 8 + TE_THREADS * 2u * 2u, stream, d_units, 2);
    if (enc_mask & MIC_ENC_CLS_WIDE) hipLaunchKernelGGL((k_enc_tans_wg<13, TE_THREADS, TE_TT_SYMS, 2>), dim3(n), dim3(TE_THREADS), (2u << 13) + TE_TT_SYMS * 8 + eb, stream, d_units, es);
    if (t) t->mark("k_enc_tans_wg<13, one wave>");
    if (enc_mask & MIC_ENC_CLS_SMALL12) hipLaunchKernelGGL((k_enc_tans_wg<12, 64, 512>), dim3(n), dim3(64), (2u << 12) + 512 * 8 + eb1, stream, d_units, es);
    if (enc_mask & MIC_ENC_CLS_SMALL13) hipLaunchKernelGGL((k_enc_tans_wg<13, 64, TE_SMALL_SYMS>), dim3(n), dim3(64), (2u << 13) + TE_SMALL_SYMS * 8 + eb1, stream, d_units, es);
    if (t) t->mark("k_enc_tans_wg<other classes>");
    if (enc_mask & MIC_ENC_CLS_TL14) hipLaunchKernelGGL((k_enc_tans_wg<14, TE_THREADS, TE_TT_SYMS>), dim3(n), dim3(TE_THREADS), (2u << 14) + TE_TT_SYMS * 8 + eb, stream, d_units, es);
    if (enc_mask & MIC_ENC_CLS_TL15) hipLaunchKernelGGL((k_enc_tans_wg<15, TE_THREADS, TE_TT_SYMS>), dim3(n), dim3(TE_THREADS), (2u << 15) + TE_TT_SYMS * 8 + eb, stream, d_units, es);
    if (enc_mask & MIC_ENC_CLS_TL16) hipLaunchKernelGGL((k_enc_tans_wg<16, TE_THREADS16, TE_TT_SYMS>), dim3(n), dim3(TE_THREADS16), (2u << 16) + TE_THREADS16 * (unsigned)es * 2u, stream, d_units, es);
    if (t) t->mark("k_enc_tans_serial");
    hipLaunchKernelGGL(k_enc_tans_serial, dim3(n), dim3(64), 0, stream, d_units);
    if (t) t->mark("k_enc_hist_clean");
    hipLaunchKernelGGL(k_enc_hist_clean, dim3(n), dim3(256), 0, stream, d_units);
    if (t) t->mark("end");
}
void mic_launch_pack(const MicUnit *d_units, int n, uint64_t *d_off, uint8_t *d_dst, uint64_t d_cap, hipStream_t stream, MicTimer *t) {
    if (t) t->mark("k_scan_lens");
    hipLaunchKernelGGL(k_scan_lens, dim3(1), dim3(1024), 0, stream, d_units, n, d_off);
    if (t) t->mark("k_enc_pack");
    hipLaunchKernelGGL(k_enc_pack, dim3(32, n), dim3(256), 0, stream, d_units, d_off, d_dst, d_cap, n);
    if (t) t->mark("end");
}
