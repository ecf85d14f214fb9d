// mic_decode_px.hip -- tokens -> pixels: RLE expansion, escape resolution and the inverse
// Delta(avg) predictor, one work-group of 1024 threads per unit.
//
// Reference: RleDecompressU16.DecodeNext2 (rledecompressu16.go:59-85) pulled once per symbol by
// DeltaRleDecompressU16.Decompress (deltarlecompressu16.go:69-128).  The reference is a serial
// pull-iterator; here the same result is produced in four phases:
//   1. header walk      wave 0 hops from RLE header to RLE header (a linked list through the
//                       token stream) and records {token index, first symbol index} segments;
//   2. expansion        all waves copy literal runs / fill same-runs into the symbol stream;
//   3. escape + index   a symbol equal to the delimiter is an escape marker unless it is itself
//                       the payload of a marker: marker[i] = isDelim[i] & !marker[i-1].  That
//                       two-state recurrence is scanned as function composition; a second scan
//                       numbers the non-marker symbols = pixels.  Each pixel slot receives its
//                       symbol (px_out, in place) and a "raw" bit (flags);
//   4. wavefront        pixel (y,x) needs (y,x-1) and (y-1,x): thread r owns row r and works on
//                       column t-r at step t, so the top neighbour was produced by thread r-1 one
//                       step earlier (LDS hand-off, one barrier per step).
#include "mic_dev.h"
#include "mic_launch.h"

#define PX_THREADS 1024
#define PX_WAVES (PX_THREADS / 64)
#define PX_T 8192                                 // symbols per tile of the expansion / numbering pass
#define PX_SPT (PX_T / PX_THREADS)                // symbols per thread
#define PX_K 8                                    // pixels per thread per wavefront step (in-work-group fallback)
#define PR_K 64                                   // pixels per lane per step in k_dec_predict: one 128-byte line per row and step
#define PR_DW (PR_K / 2)
#define PR_NARROW 1008                            // frames up to this many columns take 16-pixel groups (row buffer: 2 KiB per unit)
#define PR_MAX_W 32768                            // its row buffer is 2 bytes per column of LDS
struct __attribute__((packed, aligned(2))) PxVec { uint16_t v[PX_K]; };

__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}

// 2-state transition functions packed as f(0) | f(1) << 1; cmp(g, f) = g after f
__device__ __forceinline__ uint32_t fn_compose(uint32_t g, uint32_t f) {
    return ((g >> (f & 1)) & 1) | (((g >> ((f >> 1) & 1)) & 1) << 1);
}

// 8 waves per SIMD (= two groups per CU): keeps the kernel under 96 SGPRs and 64 VGPRs, which it nearly is anyway
__global__ void __launch_bounds__(PX_THREADS, 8) k_dec_pixels_wg(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK || u.mode != 0) return;
    if (u.walk_ok == 4) return;                                         // k_dec_rows_tok (mic_decode_fused.hip) has made this unit's pixels
    __shared__ uint32_t s_fn[PX_WAVES + 1];
    __shared__ uint32_t s_misc[8];
    __shared__ uint32_t s_segx[PX_THREADS], s_segy[PX_THREADS + 1];
    __shared__ __attribute__((aligned(16))) uint16_t s_tiles[2][PX_T + 8];   // symbols in, pixels out
    uint16_t *s_px = s_tiles[1];
    PxVec (*s_top)[PX_THREADS] = (PxVec (*)[PX_THREADS])&s_tiles[0][0];       // wide-frame fallback, after the tiles are done
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ntok = u.ntok;
    const uint16_t *tok = u.tok;
    const int W = u.w, H = u.h;
    const uint32_t npx = (uint32_t)W * (uint32_t)H;
    if (ntok < 2) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    const int d0 = mic_len16(tok[0]);                                   // rledecompressu16.go:21-25
    if (d0 == 0) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t mid = (1u << (d0 - 1)) - 1;
    const uint32_t symcap = min(u.sym_cap, 2u * npx + 2u);
    uint2 *seg = u.seg;

    // ---- phase 1: header walk (wave 0) -----------------------------------------------------
    if (u.walk_ok) {                                                    // k_dec_tans_lds walked the headers as it produced them
        if (tid == 0) { s_misc[0] = u.nseg; s_misc[1] = u.nsym; s_misc[2] = 0; s_misc[3] = 0; }
    } else if (wave == 0) {
        uint32_t pos = 1, outp = 0, nseg = 0, err = 0;
        const uint32_t segcap = u.seg_cap;
        while (pos < ntok && outp < symcap && !err) {
            const uint32_t w = (pos + lane < ntok) ? tok[pos + lane] : 0u;
            uint32_t j = 0;
            while (j < 64 && pos + j < ntok && outp < symcap) {
                uint32_t h = __builtin_amdgcn_readlane(w, (int)j);
                if (nseg >= segcap) { err = 2; break; }               // (2: the segment slab is full -- tier 1: the batch runs again in tier 2)
                // No encoder writes a count of 0, but the reference decodes one (DecodeNext2 keeps the count in a uint16: 0 passes as a run,
                // is decremented to 65535 and counts down from there as a literal chunk): one symbol = the token behind the header, then the
                // 65535 - midCount tokens behind that -- a literal chunk of 65536 - midCount symbols whose payload starts behind the header.
                // (The walkers inside the entropy kernels stop at a zero and leave the stream to this one.)
                if (h == 0) h = 65536u;
                if (h <= mid) {                                          // same-run: count, value
                    if (pos + j + 1 >= ntok) { err = 1; break; }
                    if (j == 63) break;                                  // value not in the window: reload at the header
                    if (lane == 0) seg[nseg] = make_uint2((pos + j + 1) | 0x80000000u, outp);
                    nseg++; outp += h; j += 2;
                } else {                                                 // literal run
                    if (lane == 0) seg[nseg] = make_uint2(pos + j + 1, outp);
                    nseg++; outp += h - mid; j += 1 + (h - mid);
                }
            }
            pos += j;
        }
        if (lane == 0) { s_misc[0] = nseg; s_misc[1] = min(outp, symcap); s_misc[2] = err; s_misc[3] = 0; }
    }
    __syncthreads();
    const uint32_t nseg = s_misc[0], nsym = s_misc[1];
    if (s_misc[2]) { if (tid == 0) u.status = (s_misc[2] == 2 && u.tier == 1) ? MICD_INT_GROW : MICD_ERR_CORRUPT; return; }

    // ---- phases 2+3, tile by tile: expansion, escape markers, pixel numbering ------------------------
    if (nsym < 1 || nseg < 1) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t seg0x = seg[0].x & 0x7FFFFFFFu;
    if (seg0x >= ntok) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t max_value = tok[seg0x];                          // first symbol of the stream (deltarlecompressu16.go:71)
    const int depth = mic_len16(max_value);
    if (depth == 0) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t thr = (1u << (depth - 1)) - 1;
    const uint32_t delim = (1u << depth) - 1;
    uint16_t *px = u.px_out;
    uint32_t *flags = u.flags;
    if (tid == 0) u.dec_thr = thr;
    {
        uint32_t carry_state = 0;        // marker state of the symbol in front of the tile
        uint32_t carry_px = 0;           // pixels numbered so far
        uint32_t s_lo = 0;               // first segment that reaches into the tile
        uint32_t tab_r0 = 0; bool tab_ok = false;   // segment index of the LDS table's first entry
        // this thread's 8 symbols [i0, wend) out of the LDS segment table (entries from segment r0 on): gallop from
        // table index j0, bisect, then one 16-byte load when the window lies inside one literal segment
        typedef uint32_t px_v4 __attribute__((ext_vector_type(4)));
        typedef px_v4 PxQ __attribute__((aligned(2)));
        auto fetch8 = [&](uint32_t i0, uint32_t wend, uint32_t j0, uint32_t (&w8)[4], uint32_t &bad, uint32_t &jout) {
            uint32_t j = j0, stp = 16;
            while (j + stp <= PX_THREADS && s_segy[j + stp] <= i0) { j += stp; stp <<= 1; }
            for (stp >>= 1; stp; stp >>= 1) if (j + stp <= PX_THREADS && s_segy[j + stp] <= i0) j += stp;
            uint32_t start = s_segy[j], endj = min(s_segy[j + 1], nsym), sx = s_segx[j];
            w8[0] = w8[1] = w8[2] = w8[3] = 0;
            if (wend - i0 == PX_SPT && i0 + PX_SPT <= endj) {
                const uint32_t xs = sx & 0x7FFFFFFFu;
                if (sx >> 31) {                                           // same-run: one value
                    const uint32_t v = tok[xs];
                    w8[0] = w8[1] = w8[2] = w8[3] = v | (v << 16);
                } else {
                    const uint32_t src = xs + (i0 - start);
                    if (src + PX_SPT <= ntok) {
                        const px_v4 v = *(const PxQ *)(tok + src);
                        w8[0] = v.x; w8[1] = v.y; w8[2] = v.z; w8[3] = v.w;
                    } else bad = 1;                                       // literal run past the end (Go: index panic)
                }
            } else {
#pragma unroll
                for (int k = 0; k < PX_SPT; k++) {
                    const uint32_t i = i0 + k;
                    if (i < wend) {
                        while (i >= endj) { j++; start = s_segy[j]; endj = min(s_segy[j + 1], nsym); sx = s_segx[j]; }
                        const uint32_t xs = sx & 0x7FFFFFFFu;
                        const uint32_t src = (sx >> 31) ? xs : xs + (i - start);
                        if (src < ntok) w8[k >> 1] |= (uint32_t)tok[src] << (16 * (k & 1)); else bad = 1;
                    }
                }
            }
            jout = j;
        };
        // symbols of the NEXT tile, fetched while this one is scanned (valid when the table covered that tile)
        bool pre_cov = false; uint32_t p8[4] = { 0u, 0u, 0u, 0u }; uint32_t pre_bad = 0, pre_j = 0;
        MIC_STAMP_BEGIN();
        for (uint32_t base = 0; base < nsym && carry_px < npx; base += PX_T) {
            const uint32_t tile_end = min(base + PX_T, nsym);
            // (a) every thread fetches its 8 consecutive symbols straight from the token stream.  The tile's
            // segments (start symbol, payload position | run flag) sit in an LDS table of PX_THREADS entries that
            // persists across tiles (it usually covers dozens of them) and is reloaded only when a tile reaches past
            // it; reload rounds overlap by 8 segments (a window of 8 symbols spans at most 8).
            const uint32_t i0 = base + tid * PX_SPT;
            const uint32_t wend = min(i0 + PX_SPT, tile_end);            // this thread's symbols are [i0, wend)
            const uint32_t par = (base / PX_T) & 1u;                     // s_misc[5 + par]: first segment of the next tile
            uint32_t w8[4] = { 0u, 0u, 0u, 0u };
            bool got = i0 >= tile_end;
            uint32_t my_seg = s_lo;                                      // segment this thread found (absolute index; hint for the next tile)
            if (pre_cov) {
                if (!got) {
#pragma unroll
                    for (int k = 0; k < 4; k++) w8[k] = p8[k];
                    if (pre_bad) s_misc[3] = 1;
                    if (wend == tile_end) s_misc[5 + par] = tab_r0 + pre_j;
                    my_seg = tab_r0 + pre_j; got = true;
                }
            } else {
                for (uint32_t r0 = tab_ok ? tab_r0 : s_lo;; r0 += PX_THREADS - PX_SPT) {
                    if (!(tab_ok && r0 == tab_r0)) {
                        __syncthreads();                                     // every reader of the old table is done
                        const uint32_t si = r0 + tid;
                        uint2 sg = make_uint2(0u, 0xFFFFFFFFu);
                        if (si < nseg) sg = seg[si];
                        s_segx[tid] = sg.x; s_segy[tid] = sg.y;
                        if (tid == 0) { const uint32_t sj = r0 + PX_THREADS; s_segy[PX_THREADS] = (sj < nseg) ? seg[sj].y : 0xFFFFFFFFu; }
                        __syncthreads();
                        tab_r0 = r0; tab_ok = true;
                    }
                    const uint32_t cover_end = s_segy[PX_THREADS];       // first symbol this table's segments do not cover
                    if (!got && s_segy[0] <= i0 && wend <= cover_end) {
                        uint32_t bad = 0, j;
                        fetch8(i0, wend, (s_lo > r0) ? s_lo - r0 : 0u, w8, bad, j);
                        if (bad) s_misc[3] = 1;
                        if (wend == tile_end) s_misc[5 + par] = r0 + j;  // segment of the tile's last symbol: where the next tile starts
                        my_seg = r0 + j; got = true;
                    }
                    if (!(cover_end < tile_end)) break;
                }
            }
            {   // prefetch for the next tile, if the table in LDS already covers it
                const uint32_t base2 = base + PX_T;
                const uint32_t tile_end2 = min(base2 + PX_T, nsym);
                pre_cov = base2 < nsym && s_segy[PX_THREADS] >= tile_end2;   // uniform
                if (pre_cov) {
                    const uint32_t i2 = base2 + tid * PX_SPT, wend2 = min(i2 + PX_SPT, tile_end2);
                    const uint32_t j0 = (my_seg > tab_r0) ? my_seg - tab_r0 : 0u;   // (the table may have been reloaded since)
                    pre_bad = 0; pre_j = j0;
                    if (i2 < tile_end2) fetch8(i2, wend2, j0, p8, pre_bad, pre_j);
                }
            }
            MIC_STAMP_AT(u, 0);
            // (b) marker[i] = isDelim[i] & !marker[i-1], a pixel = a non-marker symbol.
            // One scan carries, for both possible entry states, the exit state and the pixel count:
            //   t = s0 | s1 << 1 | c0 << 2 | c1 << 16
            uint32_t dmask = 0, vmask = 0;                               // bit k: symbol k is a delimiter / is inside the stream
#pragma unroll
            for (int k = 0; k < PX_SPT; k++) {
                const uint32_t x = (w8[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                const uint32_t i = i0 + k;
                const bool in = i < tile_end && i != 0;                  // symbol 0 is the max value, not a pixel
                if (in) vmask |= 1u << k;
                if (in && x == delim) dmask |= 1u << k;
            }
            // Tiles without a delimiter (almost all: escapes are rare) need no scan: every symbol is a pixel and goes
            // straight from the registers to its place.  The barrier publishes the fetch phase's verdicts either way.
            const bool plain = !__syncthreads_or((int)(dmask != 0)) && carry_state == 0;
            s_lo = s_misc[5 + par];
            if (s_misc[3]) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
            if (plain) {
                const uint32_t z = (base == 0) ? 1u : 0u;                // symbol 0 is not a pixel
                const uint32_t cnt_all = tile_end - base - z;
                const uint32_t n_out = min(cnt_all, npx - carry_px);
                const uint32_t l0 = tid * PX_SPT - z;                    // tile-local pixel index of symbol i0 (tid 0, z = 1: of symbol 1)
                if (vmask == 0xFFu && l0 + PX_SPT <= n_out) {
                    px_v4 v; v.x = w8[0]; v.y = w8[1]; v.z = w8[2]; v.w = w8[3];
                    *(PxQ *)(px + carry_px + l0) = v;
                } else {
#pragma unroll
                    for (int k = 0; k < PX_SPT; k++) {
                        const uint32_t l = tid * PX_SPT + k - z;
                        if (((vmask >> k) & 1u) && l < n_out) px[carry_px + l] = (uint16_t)((w8[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
                    }
                }
                carry_px += cnt_all;
                continue;
            }
            uint32_t tr;
            {
                uint32_t s0 = 0, s1 = 1, c0 = 0, c1 = 0;
#pragma unroll
                for (int k = 0; k < PX_SPT; k++) {
                    const uint32_t d = (dmask >> k) & 1u, v = (vmask >> k) & 1u;
                    const uint32_t m0 = d & (s0 ^ 1u), m1 = d & (s1 ^ 1u);
                    c0 += v & (m0 ^ 1u); c1 += v & (m1 ^ 1u);
                    s0 = m0; s1 = m1;
                }
                tr = s0 | (s1 << 1) | (c0 << 2) | (c1 << 16);
            }
            auto compose = [](uint32_t a, uint32_t b) -> uint32_t {     // a then b
                const uint32_t a0 = a & 1u, a1 = (a >> 1) & 1u;
                const uint32_t bc0 = (b >> 2) & 0x3FFFu, bc1 = b >> 16;
                const uint32_t n0 = (b >> a0) & 1u, n1 = (b >> a1) & 1u;
                const uint32_t c0 = ((a >> 2) & 0x3FFFu) + (a0 ? bc1 : bc0);
                const uint32_t c1 = (a >> 16) + (a1 ? bc1 : bc0);
                return n0 | (n1 << 1) | (c0 << 2) | (c1 << 16);
            };
            uint32_t incl = tr;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const uint32_t o = __shfl_up(incl, dd);
                if (lane >= (uint32_t)dd) incl = compose(o, incl);
            }
            if (lane == 63) s_fn[wave] = incl;
            uint32_t excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 2u;                                    // identity: s0 = 0, s1 = 1, no pixels
            __syncthreads();
            MIC_STAMP_AT(u, 1);
            // entry (state, pixel count) of this wave and exit of the tile, from the 16 wave totals
            uint32_t st_w = carry_state, cnt_w = 0, st_all, cnt_all;
            {
                uint32_t stt = carry_state, cc = 0;
                for (uint32_t wv = 0; wv < PX_WAVES; wv++) {
                    if (wv == wave) { st_w = stt; cnt_w = cc; }
                    const uint32_t f = s_fn[wv];
                    cc += stt ? (f >> 16) : ((f >> 2) & 0x3FFFu);
                    stt = (f >> stt) & 1u;
                }
                st_all = stt; cnt_all = cc;
            }
            uint32_t st_in = (excl >> st_w) & 1u;
            uint32_t pl = cnt_w + (st_w ? (excl >> 16) : ((excl >> 2) & 0x3FFFu));    // tile-local pixel index
#pragma unroll
            for (int k = 0; k < PX_SPT; k++) {
                const uint32_t x = (w8[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                const uint32_t d = (dmask >> k) & 1u, v = (vmask >> k) & 1u;
                const uint32_t m = d & (st_in ^ 1u);
                if (v && !m) {
                    s_px[pl] = (uint16_t)x;
                    const uint32_t pg = carry_px + pl;
                    if (st_in && pg < npx) atomicOr(&flags[pg >> 5], 1u << (pg & 31));   // stored raw behind an escape
                    pl++;
                }
                st_in = m;
            }
            __syncthreads();
            MIC_STAMP_AT(u, 2);
            // (c) the tile's pixels leave as 16-byte vectors (2-byte aligned destination)
            {
                const uint32_t n_out = min(cnt_all, npx - carry_px);
                const uint32_t o = tid * PX_SPT;
                if (o + PX_SPT <= n_out) *(PxVec *)(px + carry_px + o) = *(const PxVec *)(s_px + o);
                else for (uint32_t k = o; k < n_out; k++) px[carry_px + k] = s_px[k];
            }
            carry_px += cnt_all;
            carry_state = st_all;
            __syncthreads();
            MIC_STAMP_AT(u, 3);
        }
        if (carry_px < npx) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }   // tokens ran out (Go: panic)
    }
    __threadfence_block();
    __syncthreads();

    // ---- phase 4: inverse predictor, skewed wavefront ---------------------------------------------
    // Thread r owns row r of the band and works on the 8-pixel column group t - r at step t: its top
    // neighbours were produced by thread r-1 one step earlier (LDS hand-off, one barrier per step, 8
    // pixels per barrier), its left neighbour is its own previous pixel.  Symbols arrive and pixels
    // leave as 16-byte vectors (2-byte aligned; gfx950 runs in unaligned-access mode), fetched two
    // steps ahead so the L2 round trip is off the step's critical path.
    // Frames up to PR_MAX_W columns leave here: k_dec_predict does this phase one wave per unit.
    if (W <= PR_MAX_W) return;
    if (u.pred) { if (tid == 0) u.status = MICD_ERR_UNSUPPORTED; return; }   // the gradient predictor has no in-group wavefront
    const int ngrp = (W + PX_K - 1) / PX_K;
    for (int rb = 0; rb < H; rb += PX_THREADS) {
        const int y = rb + (int)tid;
        const bool row_ok = y < H;
        const int rows = min(H - rb, PX_THREADS);
        const int steps = ngrp + rows - 1;
        const size_t rowp = (size_t)y * (size_t)W;
        uint32_t left = 0;
        auto fetch = [&](int g, PxVec &v, uint32_t &rawbits) {
            v = PxVec{}; rawbits = 0;
            if (!row_ok || g < 0 || g >= ngrp) return;
            const size_t p = rowp + (size_t)g * PX_K;
            if (g * PX_K + PX_K <= W) v = *(const PxVec *)(px + p);
            else for (int k = 0; k < PX_K; k++) if (g * PX_K + k < W) v.v[k] = px[p + k];
            // raw bits of pixels p .. p+7: two dwords, funnel-shifted
            const uint32_t w0 = flags[p >> 5], w1 = flags[(p >> 5) + 1];
            rawbits = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (p & 31)) & 0xFFu;
        };
        PxVec v0, v1; uint32_t r0, r1;
        fetch(0 - (int)tid, v0, r0);
        fetch(1 - (int)tid, v1, r1);
        for (int t = 0; t < steps; t++) {
            const int g = t - (int)tid;
            const bool act = row_ok && g >= 0 && g < ngrp;
            const PxVec vin = v0; const uint32_t raw = r0;
            v0 = v1; r0 = r1;
            fetch(g + 2, v1, r1);                                      // two steps ahead
            PxVec res = PxVec{};
            if (act) {
                const int c0 = g * PX_K;
                PxVec top = PxVec{};
                if (y > 0) {
                    if (tid > 0) top = s_top[(t + 1) & 1][tid - 1];
                    else {                                             // first row of a later band: previous band's last row
                        const size_t q = (size_t)(y - 1) * (size_t)W + (size_t)c0;
                        if (c0 + PX_K <= W) top = *(const PxVec *)(px + q);
                        else for (int k = 0; k < PX_K; k++) if (c0 + k < W) top.v[k] = px[q + k];
                    }
                }
#pragma unroll
                for (int k = 0; k < PX_K; k++) {
                    const int col = c0 + k;
                    int32_t pred;
                    if (col > 0 && y > 0) pred = (int32_t)((left + (uint32_t)top.v[k]) >> 1);
                    else if (col > 0) pred = (int32_t)left;
                    else if (y > 0) pred = (int32_t)top.v[k];
                    else pred = 0;
                    const uint32_t val = vin.v[k];
                    const uint32_t r = ((raw >> k) & 1) ? val : (uint32_t)(uint16_t)(pred + ((int32_t)val - (int32_t)thr));   // deltarlecompressu16.go:96-98
                    res.v[k] = (uint16_t)r;
                    if (col < W) left = r;
                }
                const size_t p = rowp + (size_t)c0;
                if (c0 + PX_K <= W) *(PxVec *)(px + p) = res;
                else for (int k = 0; k < PX_K; k++) if (c0 + k < W) px[p + k] = res.v[k];
            }
            s_top[t & 1][tid] = res;
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
    }
}


// ==========================================================================================
// Phase 4 as its own kernel: inverse Delta(avg) predictor, one wave per unit.
//
//   out[y][x] = raw ? sym : ((left + top) >> 1) + sym - thr        (deltarlecompressu16.go:83-99;
//   left only on row 0, top only in column 0, 0 at the origin), 16-bit wrap-around.
//
// Lane r owns rows r, r+64, r+128, ... and walks each in groups of 32 pixels (64 bytes); at step t
// it is at virtual index t - r of its sequence (band b = index / P, group g = index % P, P =
// max(groups per row, 64)).  Its top neighbours were produced by lane r-1 one step earlier and
// arrive by DPP wave_shr:1; lane 0's come from lane 63 of the band before, through a row buffer in
// LDS that lane 63 fills as it goes (P >= 64 makes that hand-off causal).  Row 0 of the unit is a
// segmented prefix sum, computed up front into the same row buffer, so lane 0 simply copies it.
// Per step a lane reads 64 contiguous bytes of symbols (2-byte aligned vector loads, fetched three
// steps ahead) and writes 64 bytes of pixels: every cache line is consumed whole within two steps.
// Escapes are rare: a step takes the select-free path unless some lane of the wave holds a raw pixel.
typedef uint32_t pr_v4 __attribute__((ext_vector_type(4)));
typedef uint32_t pr_v2 __attribute__((ext_vector_type(2)));
typedef pr_v4 PrQ __attribute__((aligned(2)));                // 2-byte aligned vector accesses (gfx950: unaligned mode)
typedef pr_v2 PrD __attribute__((aligned(2)));
typedef uint32_t PrS __attribute__((aligned(2)));
typedef __attribute__((address_space(1))) uint16_t *pr_gu16;

// cnt (1..K, wave-uniform) pixels of a group, packed two per dword
template <int K>
__device__ __forceinline__ void pr_store_cnt(pr_gu16 dst, const uint32_t (&d)[K / 2], int cnt) {
#pragma unroll
    for (int q = 0; q < K / 8; q++) {
        const int j = 4 * q;
        pr_gu16 o = dst + 8 * q;
        if (cnt >= 8 * q + 8) {
            pr_v4 v; v.x = d[j]; v.y = d[j + 1]; v.z = d[j + 2]; v.w = d[j + 3];
            *(__attribute__((address_space(1))) PrQ *)o = v;
        } else if (cnt > 8 * q) {
            const int r = cnt - 8 * q;
            if (r & 4) {
                pr_v2 v; v.x = d[j]; v.y = d[j + 1];
                *(__attribute__((address_space(1))) PrD *)o = v;
                if (r & 2) { *(__attribute__((address_space(1))) PrS *)(o + 4) = d[j + 2]; if (r & 1) o[6] = (uint16_t)d[j + 3]; }
                else if (r & 1) o[4] = (uint16_t)d[j + 2];
            } else if (r & 2) {
                *(__attribute__((address_space(1))) PrS *)o = d[j];
                if (r & 1) o[2] = (uint16_t)d[j + 1];
            } else {
                o[0] = (uint16_t)d[j];
            }
        }
    }
}

#pragma push_macro("PR_K")
#pragma push_macro("PR_DW")
#undef PR_K
#undef PR_DW
template <int K>
struct PrSlot {
    uint32_t d[K / 2];    // K symbols
    uint32_t w0, w1, w2;  // flag words around the group (funnel-shifted at use: consuming them at fetch time would
                          // make the fetch wait for its own loads)
    int32_t g, y;         // group and row; g < 0 or act == 0: nothing to do
    uint32_t p;           // pixel index of the group's first pixel
    uint32_t act;
};

// WIDE names the row-buffer class of the launch (a distinct instantiation has its own line in a kernel trace).
// The ordinary class runs four units per group, one wave each with its own row buffer: a group's waves land on the four
// SIMDs by construction, whereas single-wave groups pile up unevenly and the younger wave of a crowded SIMD starves.
// K = pixels per lane and step.  A lane owns a row and trails the row above by one group, so a band of 64 rows keeps a lane busy
// (columns / K) steps out of every max(columns / K, 64): 64-pixel groups (one 128-byte line per row and step) suit wide strips, frames
// of a few hundred columns (MIC3 planes, CT slices) take 16-pixel groups -- at 256 columns 25 % of the steps are work instead of 6 %.
template <int WIDE, int K>
__global__ void __launch_bounds__(WIDE ? 64 : 256) k_dec_predict(MicUnit *units, int n_units, int w_lo, int w_hi, uint32_t rb_dwords) {
    constexpr int PR_K = K, PR_DW = K / 2;                       // (shadow the file-scope defaults)
    constexpr int WPG = WIDE ? 1 : 4;                            // waves (= units) per group
    const int ui = (int)blockIdx.x * WPG + (int)(threadIdx.x >> 6);
    if (ui >= n_units) return;
    MicUnit &u = units[ui];
    if (u.status != MICD_OK || u.mode != 0 || u.pred) return;
    const int W = u.w, H = u.h;
    if (W <= w_lo || W > w_hi) return;
    extern __shared__ uint32_t s_rowbuf_all[];                   // per wave: ngrp x PR_DW dwords
    uint32_t *const s_rowbuf = s_rowbuf_all + (WIDE ? 0u : (threadIdx.x >> 6) * rb_dwords);   // 2 bytes per column and wave (launch class)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t npx = (uint32_t)W * (uint32_t)H;
    const uint32_t thr = u.dec_thr;
    const pr_gu16 px = (pr_gu16)u.px_out;
    const __attribute__((address_space(1))) uint32_t *flags = (const __attribute__((address_space(1))) uint32_t *)u.flags;
    const int ngrp = (W + PR_K - 1) / PR_K;
    const int tail = W - (ngrp - 1) * PR_K;                      // pixels in the last group of a row, 1..32
    const int P = max(ngrp, 64);
    const int nb = (H + 63) >> 6;

    // ---- row 0: out[x] = raw ? sym : out[x-1] + sym - thr, out[-1] = 0; a scan of (reset, sum) pairs ----
    {
        uint16_t *row16 = (uint16_t *)s_rowbuf;
        for (int x = (int)lane; x < ngrp * PR_K; x += 64) row16[x] = (x < W) ? px[x] : (uint16_t)0;
        __builtin_amdgcn_s_waitcnt(0xC07F);                          // wave-private LDS: writes above land before the reads below
        const int cw = (W + 63) >> 6;
        const int x0 = min(W, (int)lane * cw), x1 = min(W, x0 + cw);
        uint32_t rs = 0, sum = 0;
        for (int x = x0; x < x1; x++) {
            const uint32_t raw = (flags[x >> 5] >> (x & 31)) & 1u, v = row16[x];
            if (raw) { rs = 1; sum = v; } else sum = (sum + v - thr) & 0xFFFFu;
        }
        // inclusive scan of the composition (a then b) = b.reset ? b : (a.reset, a.sum + b.sum)
        uint32_t is = sum, ir = rs;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t os = __shfl_up(is, dd), orr = __shfl_up(ir, dd);
            if (lane >= (uint32_t)dd && !ir) { is = (is + os) & 0xFFFFu; ir = orr; }
        }
        uint32_t cur = __shfl_up(is, 1);
        if (lane == 0) cur = 0;
        for (int x = x0; x < x1; x++) {
            const uint32_t raw = (flags[x >> 5] >> (x & 31)) & 1u, v = row16[x];
            cur = raw ? v : ((cur + v - thr) & 0xFFFFu);
            row16[x] = (uint16_t)cur;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                          // wave-private LDS: writes above land before the reads below
    }

    // ---- the pipeline ----
    int32_t cg = -(int32_t)lane, cb = 0;                         // cursor of the NEXT group to fetch
    auto fetch = [&](PrSlot<K> &s) {
        const int32_t y = (int32_t)lane + 64 * cb;
        s.g = cg; s.y = y;
        s.act = (cg >= 0 && cg < ngrp && cb < nb && y < H) ? 1u : 0u;
        s.p = (uint32_t)y * (uint32_t)W + (uint32_t)cg * PR_K;
        s.w0 = 0; s.w1 = 0; s.w2 = 0;
#pragma unroll
        for (int i = 0; i < PR_DW; i++) s.d[i] = 0;
        if (s.act) {
            const pr_gu16 src = px + s.p;
            if (s.p + PR_K <= npx) {
#pragma unroll
                for (int q = 0; q < PR_K / 8; q++) {
                    const pr_v4 v = *(const __attribute__((address_space(1))) PrQ *)(src + 8 * q);
                    s.d[4 * q] = v.x; s.d[4 * q + 1] = v.y; s.d[4 * q + 2] = v.z; s.d[4 * q + 3] = v.w;
                }
            } else {                                             // last row's tail: stay inside the buffer
                for (int k = 0; k < PR_K; k++) if (s.p + (uint32_t)k < npx) {
                    const uint32_t v = src[k];
#pragma unroll
                    for (int i = 0; i < PR_DW; i++) if (i == (k >> 1)) s.d[i] |= v << (16 * (k & 1));
                }
            }
            s.w0 = flags[s.p >> 5]; s.w1 = flags[(s.p >> 5) + 1]; s.w2 = flags[(s.p >> 5) + 2];
        }
        if (++cg == P) { cg = 0; cb++; }
    };
    uint32_t last[PR_DW];                                        // this lane's previous result = next lane's top
#pragma unroll
    for (int i = 0; i < PR_DW; i++) last[i] = 0;
    uint32_t left = 0;
    auto step = [&](const PrSlot<K> &s) {
        // top neighbours: lane r-1's previous result; lane 0 takes the row buffer entry of its group
        uint32_t top[PR_DW];
        {
            const int32_t gc = min(max(s.g, 0), ngrp - 1);
            const uint4 *rb4 = (const uint4 *)(s_rowbuf + gc * PR_DW);
            uint32_t lv[PR_DW];
#pragma unroll
            for (int i = 0; i < PR_DW; i++) lv[i] = 0;
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < PR_DW / 4; q++) { const uint4 v = rb4[q]; lv[4 * q] = v.x; lv[4 * q + 1] = v.y; lv[4 * q + 2] = v.z; lv[4 * q + 3] = v.w; }
            }
#pragma unroll
            for (int i = 0; i < PR_DW; i++) top[i] = __builtin_amdgcn_update_dpp(lv[i], last[i], 0x138, 0xF, 0xF, false);   // wave_shr:1
        }
        uint32_t res[PR_DW];
        if (s.g == 0) left = top[0] & 0xFFFFu;                   // column 0: predictor = top ((top+top)>>1)
        uint32_t raw_lo = __builtin_amdgcn_alignbit(s.w1, s.w0, s.p);   // bit k: pixel p + k is stored raw
        uint32_t raw_hi = __builtin_amdgcn_alignbit(s.w2, s.w1, s.p);   // pixels p + 32 ..
        if (s.y == 0) { raw_lo = 0xFFFFFFFFu; raw_hi = 0xFFFFFFFFu; }   // row 0 comes ready-made from the row buffer
        if (!s.act) { raw_lo = 0; raw_hi = 0; }
        const bool any_raw = __any((raw_lo | raw_hi) != 0);
        if (!any_raw) {
#pragma unroll
            for (int i = 0; i < PR_DW; i++) {
                const uint32_t t0 = top[i] & 0xFFFFu, t1 = top[i] >> 16;
                const uint32_t c0 = (s.d[i] & 0xFFFFu) - thr, c1 = (s.d[i] >> 16) - thr;
                const uint32_t r0 = (((left + t0) >> 1) + c0) & 0xFFFFu;
                const uint32_t r1 = (((r0 + t1) >> 1) + c1) & 0xFFFFu;
                res[i] = r0 | (r1 << 16);
                left = r1;
            }
        } else {
            const bool row0 = s.y == 0;
#pragma unroll
            for (int i = 0; i < PR_DW; i++) {
                const uint32_t raw = (i < 16) ? raw_lo >> (2 * (i & 15)) : raw_hi >> (2 * (i & 15));   // bits 0, 1: this dword's pixels
                const uint32_t src = row0 ? top[i] : s.d[i];
                const uint32_t t0 = top[i] & 0xFFFFu, t1 = top[i] >> 16;
                const uint32_t v0 = src & 0xFFFFu, v1 = src >> 16;
                const uint32_t r0 = (raw & 1u) ? v0 : ((((left + t0) >> 1) + v0 - thr) & 0xFFFFu);
                const uint32_t r1 = (raw & 2u) ? v1 : ((((r0 + t1) >> 1) + v1 - thr) & 0xFFFFu);
                res[i] = r0 | (r1 << 16);
                left = r1;
            }
        }
        if (s.act) {
            const pr_gu16 dst = px + s.p;
            if (s.g < ngrp - 1 || tail == PR_K) pr_store_cnt<K>(dst, res, PR_K);
            else pr_store_cnt<K>(dst, res, tail);
            if (lane == 63) {
                uint4 *rb4 = (uint4 *)(s_rowbuf + s.g * PR_DW);
#pragma unroll
                for (int q = 0; q < PR_DW / 4; q++) rb4[q] = make_uint4(res[4 * q], res[4 * q + 1], res[4 * q + 2], res[4 * q + 3]);
            }
        }
#pragma unroll
        for (int i = 0; i < PR_DW; i++) last[i] = res[i];
    };
    PrSlot<K> sa, sb, sc;
    fetch(sa); fetch(sb); fetch(sc);
    const int steps = nb * P + 63;
    for (int t = 0; t < steps; t += 3) {
        step(sa); fetch(sa);
        step(sb); fetch(sb);
        step(sc); fetch(sc);
    }
}

#pragma pop_macro("PR_DW")
#pragma pop_macro("PR_K")

// ------------------------------------------------------------------------------------------
// The same wavefront for wide frames, with the memory traffic taken off the computing wave: TWO waves per unit.
//
// k_dec_predict above gives every lane a row, so each of its vector loads and stores touches 64 different cache lines (one per
// lane), a line is consumed over eight instructions, and every conditional store makes the compiler wait for ALL outstanding
// memory (s_waitcnt vmcnt(0): the counter is in order, loads and stores together) -- its prefetch never runs ahead.  Here
//   * wave M (memory) moves TILES: the 64 groups of a step -- lane r's group t - r, 32 pixels = 64 bytes each, a diagonal of the
//     band -- come in with four 16-byte loads in which four consecutive lanes cover one group (a quad of lanes = one 64-byte
//     request), are written to LDS one group per 80-byte line (16 bytes of padding: bank-conflict-free both ways), and the
//     finished tile of the step before leaves the same way.  Its waits for memory stall nobody else; it also turns the flag words
//     of a group into the 32 raw bits the computing lane needs;
//   * wave C (compute) reads its group from LDS (four 16-byte reads), runs the recurrence, writes the result in place.  It issues
//     no global memory operation inside the loop, so it never waits for one.
// One barrier per step keeps the two in lockstep: C works on tile t while M stores tile t - 1 and fetches tile t + 2.
#define P2_K 32
#define P2_DW 16
#define P2_LS 20                                   // dwords per LDS line: 16 of data + 4 of padding
#define P2_TILE (64 * P2_LS)
__global__ void __launch_bounds__(128) k_dec_predict2(MicUnit *units, int w_lo, int w_hi, uint32_t rb_dwords) {
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK || u.mode != 0 || u.pred) return;
    const int W = u.w, H = u.h;
    if (W <= w_lo || W > w_hi) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_p2[];   // row buffer | two tiles | two x 64 raw-bit words
    uint32_t *const s_rowbuf = s_p2;
    uint32_t *const s_tile = s_p2 + rb_dwords;
    uint32_t *const s_raw = s_tile + 2 * P2_TILE;
    const uint32_t lane = threadIdx.x & 63;
    const bool wave_c = threadIdx.x < 64;
    const uint32_t npx = (uint32_t)W * (uint32_t)H;
    const uint32_t thr = u.dec_thr;
    const pr_gu16 px = (pr_gu16)u.px_out;
    const __attribute__((address_space(1))) uint32_t *flags = (const __attribute__((address_space(1))) uint32_t *)u.flags;
    const int ngrp = (W + P2_K - 1) / P2_K;
    const int P = max(ngrp, 64);
    const int nb = (H + 63) >> 6;
    const int steps = nb * P + 63;

    if (wave_c) {
        // ---- row 0: out[x] = raw ? sym : out[x-1] + sym - thr, out[-1] = 0; a scan of (reset, sum) pairs (as in k_dec_predict) ----
        uint16_t *row16 = (uint16_t *)s_rowbuf;
        for (int x = (int)lane; x < ngrp * P2_K; x += 64) row16[x] = (x < W) ? px[x] : (uint16_t)0;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const int cw = (W + 63) >> 6;
        const int x0 = min(W, (int)lane * cw), x1 = min(W, x0 + cw);
        uint32_t rs = 0, sum = 0;
        for (int x = x0; x < x1; x++) {
            const uint32_t raw = (flags[x >> 5] >> (x & 31)) & 1u, v = row16[x];
            if (raw) { rs = 1; sum = v; } else sum = (sum + v - thr) & 0xFFFFu;
        }
        uint32_t is = sum, ir = rs;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t os = __shfl_up(is, dd), orr = __shfl_up(ir, dd);
            if (lane >= (uint32_t)dd && !ir) { is = (is + os) & 0xFFFFu; ir = orr; }
        }
        uint32_t cur = __shfl_up(is, 1);
        if (lane == 0) cur = 0;
        for (int x = x0; x < x1; x++) {
            const uint32_t raw = (flags[x >> 5] >> (x & 31)) & 1u, v = row16[x];
            cur = raw ? v : ((cur + v - thr) & 0xFFFFu);
            row16[x] = (uint16_t)cur;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }

    // ---- wave M: tile mover -------------------------------------------------------------------------------------------------
    // lane i handles, for k = 0..3, the 16-byte chunk (i & 3) of line 16 k + (i >> 2), and the raw bits of line i.  A cursor walks
    // a line's sequence of groups incrementally (group, band, first pixel of the chunk; a line whose rows have run out parks at a
    // group number that stays negative).  Loads run two tiles ahead of the stores; a chunk goes out of and into the SAME LDS slot
    // of the same lane (tile t - 1's result out, tile t + 1's pixels in), so the exchange needs no second buffer and no barrier.
    if (!wave_c) {
        const uint32_t ch = lane & 3;
        const uint32_t cmax = ((uint32_t)W - ch * 8 + (P2_K - 1)) / P2_K;      // groups of a row in which this lane's chunk has pixels
        const uint32_t crem = (uint32_t)W - ch * 8 - (cmax - 1) * P2_K;         // pixels of the chunk in the last of them (>= 1)
        struct Cur { int32_t cg, cb; uint32_t p; };
        constexpr int32_t PARKED = -(1 << 30);
        auto cur_init = [&](Cur &c, int line, int tile, uint32_t coff) {        // cursor of `line` at tile `tile` <= line (tile - line <= 0)
            c.cg = (line < H) ? tile - line : PARKED; c.cb = 0;
            c.p = (uint32_t)line * (uint32_t)W + coff + (uint32_t)((tile - line) * P2_K);
        };
        auto cur_next = [&](Cur &c, int line, uint32_t coff) {
            c.cg++; c.p += P2_K;
            if (c.cg == P) {
                c.cb++;
                const int y = line + 64 * c.cb;
                c.cg = (c.cb < nb && y < H) ? 0 : PARKED;
                c.p = (uint32_t)y * (uint32_t)W + coff;
            }
        };
        Cur lc[4], sc[4], fc;                                                   // load cursors (two tiles ahead), store cursors, flag cursor
#pragma unroll
        for (int k = 0; k < 4; k++) { cur_init(lc[k], 16 * k + (int)(lane >> 2), 0, ch * 8); cur_init(sc[k], 16 * k + (int)(lane >> 2), -1, ch * 8); }
        cur_init(fc, (int)lane, 0, 0);
        auto load_tile = [&](pr_v4 (&r)[4], uint32_t &rawbits) {              // the tile under the load cursors; advances them
#pragma unroll
            for (int k = 0; k < 4; k++) {
                r[k] = pr_v4{0u, 0u, 0u, 0u};
                if ((uint32_t)lc[k].cg < cmax) {
                    const uint32_t p = lc[k].p;
#if defined(P2_ABL) && (P2_ABL & 2)
                    r[k] = pr_v4{p, p, p, p};
#else
                    if (p + 8 <= npx) r[k] = *(const __attribute__((address_space(1))) PrQ *)(px + p);
                    else {                                                   // the unit's last pixels: stay inside the buffer
                        uint32_t e[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) e[j] = (p + (uint32_t)j < npx) ? (uint32_t)px[p + j] : 0u;
                        r[k] = pr_v4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
                    }
#endif
                }
                cur_next(lc[k], 16 * k + (int)(lane >> 2), ch * 8);
            }
            rawbits = 0;
            if ((uint32_t)fc.cg < (uint32_t)ngrp) {
                const uint32_t fp = fc.p;
                rawbits = __builtin_amdgcn_alignbit(flags[(fp >> 5) + 1], flags[fp >> 5], fp);   // bit k: pixel p + k is stored raw
                if (fc.cb == 0 && lane == 0) rawbits = 0xFFFFFFFFu;           // row 0 comes ready-made from the row buffer
            }
            cur_next(fc, (int)lane, 0);
        };
        // tile `tile`'s pixels into its LDS slots; what the slots held -- the results of tile `tile` - 2, under the store cursors -- to memory
        auto swap_tile = [&](int tile, const pr_v4 (&r)[4], uint32_t rawbits) {
            uint32_t *tb = s_tile + (tile & 1) * P2_TILE;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int line = 16 * k + (int)(lane >> 2);
                uint4 *slot = (uint4 *)(tb + line * P2_LS + ch * 4);
                const uint4 v = *slot;
                *slot = make_uint4(r[k].x, r[k].y, r[k].z, r[k].w);
                if ((uint32_t)sc[k].cg < cmax) {
                    const pr_gu16 dst = px + sc[k].p;
                    const uint32_t cnt = ((uint32_t)sc[k].cg == cmax - 1) ? min(8u, crem) : 8u;
#if !(defined(P2_ABL) && (P2_ABL & 4))
                    if (cnt == 8) { pr_v4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; *(__attribute__((address_space(1))) PrQ *)dst = w; }   // (a non-temporal store here: 3.3 -> 4.0 ms)
                    else { const uint32_t d[4] = { v.x, v.y, v.z, v.w }; pr_store_cnt<8>(dst, d, (int)cnt); }
#endif
                }
                cur_next(sc[k], line, ch * 8);
            }
            s_raw[(tile & 1) * 64 + lane] = rawbits;
        };
        pr_v4 ra[4], rb[4]; uint32_t fa, fb;
        const pr_v4 zero4[4] = { pr_v4{0u, 0u, 0u, 0u}, pr_v4{0u, 0u, 0u, 0u}, pr_v4{0u, 0u, 0u, 0u}, pr_v4{0u, 0u, 0u, 0u} };
        load_tile(ra, fa);                                                      // tile 0
        {   // tile 0 goes in without anything coming out (the store cursors stand at tile -1: step 0 takes that)
            uint32_t *tb = s_tile;
#pragma unroll
            for (int k = 0; k < 4; k++) *(uint4 *)(tb + (16 * k + (int)(lane >> 2)) * P2_LS + ch * 4) = make_uint4(ra[k].x, ra[k].y, ra[k].z, ra[k].w);
            s_raw[lane] = fa;
        }
        load_tile(ra, fa);                                                      // tile 1
        load_tile(rb, fb);                                                      // tile 2
        __syncthreads();                                                        // tile 0 and row 0 are in place
        for (int t = 0; t < steps; t += 2) {
            // step t (even): tile t + 1 sits in ra, tile t + 2 in rb; tile t + 3 goes to ra.  Out goes tile t - 1.
            swap_tile(t + 1, ra, fa);
            load_tile(ra, fa);
            __syncthreads();
            if (t + 1 >= steps) break;
            swap_tile(t + 2, rb, fb);
            load_tile(rb, fb);
            __syncthreads();
        }
        swap_tile(steps + 1, zero4, 0u);                                        // the last tile's results (same slots as tile steps - 1)
        return;
    }

    // ---- wave C: the recurrence -------------------------------------------------------------------------------------------------
    __syncthreads();
    int32_t cg = -(int32_t)lane, cb = 0;                         // this lane's group and band at the current step
    uint32_t last[P2_DW];                                        // this lane's previous result = next lane's top
#pragma unroll
    for (int i = 0; i < P2_DW; i++) last[i] = 0;
    uint32_t left = 0;
    for (int t = 0; t < steps; t++) {
        uint32_t *tb = s_tile + (t & 1) * P2_TILE + lane * P2_LS;
        const int32_t y = (int32_t)lane + 64 * cb;
        const bool act = cg >= 0 && cg < ngrp && cb < nb && y < H;
        uint32_t d[P2_DW];
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint4 v = *(const uint4 *)(tb + 4 * q); d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w; }
        uint32_t raw = act ? s_raw[(t & 1) * 64 + lane] : 0u;
        // top neighbours: lane r-1's previous result; lane 0 takes the row buffer entry of its group
        uint32_t top[P2_DW];
        {
            const int32_t gc = min(max(cg, 0), ngrp - 1);
            const uint4 *rb4 = (const uint4 *)(s_rowbuf + gc * P2_DW);
            uint32_t lv[P2_DW];
#pragma unroll
            for (int i = 0; i < P2_DW; i++) lv[i] = 0;
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) { const uint4 v = rb4[q]; lv[4 * q] = v.x; lv[4 * q + 1] = v.y; lv[4 * q + 2] = v.z; lv[4 * q + 3] = v.w; }
            }
#pragma unroll
            for (int i = 0; i < P2_DW; i++) top[i] = __builtin_amdgcn_update_dpp(lv[i], last[i], 0x138, 0xF, 0xF, false);   // wave_shr:1
        }
        uint32_t (&res)[P2_DW] = last;                           // (every lane has read its neighbour's `last` by now)
        if (cg == 0) left = top[0] & 0xFFFFu;                    // column 0: predictor = top ((top+top)>>1)
#if defined(P2_ABL) && (P2_ABL & 1)
        if (true) {
#pragma unroll
            for (int i = 0; i < P2_DW; i++) res[i] = d[i] + top[i];
        } else
#endif
        if (!__any(raw != 0)) {
#pragma unroll
            for (int i = 0; i < P2_DW; i++) {
                const uint32_t t0 = top[i] & 0xFFFFu, t1 = top[i] >> 16;
                const uint32_t c0 = (d[i] & 0xFFFFu) - thr, c1 = (d[i] >> 16) - thr;
                const uint32_t r0 = (((left + t0) >> 1) + c0) & 0xFFFFu;
                const uint32_t r1 = (((r0 + t1) >> 1) + c1) & 0xFFFFu;
                res[i] = r0 | (r1 << 16);
                left = r1;
            }
        } else {
            const bool row0 = y == 0;
#pragma unroll
            for (int i = 0; i < P2_DW; i++) {
                const uint32_t rw = raw >> (2 * i);                    // bits 0, 1: this dword's pixels
                const uint32_t src = row0 ? top[i] : d[i];
                const uint32_t t0 = top[i] & 0xFFFFu, t1 = top[i] >> 16;
                const uint32_t v0 = src & 0xFFFFu, v1 = src >> 16;
                const uint32_t r0 = (rw & 1u) ? v0 : ((((left + t0) >> 1) + v0 - thr) & 0xFFFFu);
                const uint32_t r1 = (rw & 2u) ? v1 : ((((r0 + t1) >> 1) + v1 - thr) & 0xFFFFu);
                res[i] = r0 | (r1 << 16);
                left = r1;
            }
        }
        if (act) {
#pragma unroll
            for (int q = 0; q < 4; q++) *(uint4 *)(tb + 4 * q) = make_uint4(res[4 * q], res[4 * q + 1], res[4 * q + 2], res[4 * q + 3]);
            if (lane == 63) {
                uint4 *rb4 = (uint4 *)(s_rowbuf + cg * P2_DW);
#pragma unroll
                for (int q = 0; q < 4; q++) rb4[q] = make_uint4(res[4 * q], res[4 * q + 1], res[4 * q + 2], res[4 * q + 3]);
            }
        }
        if (++cg == P) { cg = 0; cb++; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Inverse gradient-adaptive predictor (GradDeltaRleDecompressU16.Decompress, deltagradrlecompressu16.go:70-133) for units with
// pred = 1 (PICA strips flagged picaFlagGradPredictor).  Input as for k_dec_predict: px_out holds each pixel's delta symbol (or the
// raw value where its flag bit is set).  One wave per unit, lane r owns row 64 b + r of band b and lags the row above by one
// four-pixel group: at step t it rebuilds columns 4 (t - r) .. + 3 from W = its own previous pixel, N / NW = the group lane r - 1
// produced one step earlier (DPP wave_shr), and NE of the group's last pixel = the first pixel lane r - 1 produces in THIS step,
// handed over in the middle of the step.  Lane 0 takes the row above from an LDS row buffer that lane 63 of the previous band
// filled (wave-private, no barriers).  Symbols are fetched one step ahead as 8-byte vectors.
#define PG_Q 4
__global__ void __launch_bounds__(64) k_dec_predict_grad(MicUnit *units, int w_lo, int w_hi) {
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK || u.mode != 0 || !u.pred) return;
    const int W = u.w, H = u.h;
    if (W <= w_lo || W > w_hi) return;                           // row-buffer class of this launch (wider than PR_MAX_W: k_dec_pixels_wg has flagged it)
    extern __shared__ uint16_t s_grow[];                         // the last row of the previous band (+ PG_Q slack)
    const uint32_t lane = threadIdx.x;
    const int32_t thr = (int32_t)u.dec_thr;
    const pr_gu16 px = (pr_gu16)u.px_out;
    const __attribute__((address_space(1))) uint32_t *flags = (const __attribute__((address_space(1))) uint32_t *)u.flags;
    const int nq = (W + PG_Q - 1) / PG_Q;                        // groups per row
    const uint32_t npx = (uint32_t)W * (uint32_t)H;
    for (int x = (int)lane; x < nq * PG_Q + PG_Q; x += 64) s_grow[x] = 0;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    struct Grp { uint32_t v[PG_Q]; uint32_t raw; };              // symbols of a group and the raw bits of its pixels
    for (int y0 = 0; y0 < H; y0 += 64) {
        const int y = y0 + (int)lane;
        const bool row_ok = y < H;
        const uint32_t rowp = (uint32_t)y * (uint32_t)W;
        auto fetch = [&](int q) -> Grp {
            Grp g; g.raw = 0;
#pragma unroll
            for (int k = 0; k < PG_Q; k++) g.v[k] = 0;
            if (!row_ok || q < 0 || q >= nq) return g;
            const uint32_t p = rowp + (uint32_t)q * PG_Q;
            if (p + PG_Q <= npx) {
                const pr_v2 d = *(const __attribute__((address_space(1))) PrD *)(px + p);
                g.v[0] = d.x & 0xFFFFu; g.v[1] = d.x >> 16; g.v[2] = d.y & 0xFFFFu; g.v[3] = d.y >> 16;
            } else {
#pragma unroll
                for (int k = 0; k < PG_Q; k++) if (p + (uint32_t)k < npx) g.v[k] = px[p + (uint32_t)k];
            }
            const uint32_t w0 = flags[p >> 5], w1 = flags[(p >> 5) + 1];   // (the flag slab has a spare word behind the last pixel)
            g.raw = __builtin_amdgcn_alignbit(w1, w0, p) & 0xFu;
            return g;
        };
        uint32_t left = 0;                                       // W of the next pixel
        uint32_t nw = 0;                                         // ... and at the column before them
        uint32_t mine[PG_Q] = { 0u, 0u, 0u, 0u };                // this lane's previous result
        Grp nx = fetch(-(int)lane);
        const int steps = nq + 63;
        for (int t = 0; t < steps; t++) {
            const int q = t - (int)lane;
            const Grp g = nx;
            nx = fetch(q + 1);
            const bool act = row_ok && q >= 0 && q < nq;
            // N of the four columns: what lane r - 1 produced in the previous step; lane 0: the row buffer
            uint32_t up[PG_Q];
            {
                const int qc = min(max(q, 0), nq - 1);
                uint32_t rb0 = 0, rb1 = 0;
                if (lane == 0) { const uint2 r = *(const uint2 *)(s_grow + qc * PG_Q); rb0 = r.x; rb1 = r.y; }
                const uint32_t m0 = mine[0] | (mine[1] << 16), m1 = mine[2] | (mine[3] << 16);
                const uint32_t t0 = __builtin_amdgcn_update_dpp(rb0, m0, 0x138, 0xF, 0xF, false);   // wave_shr:1
                const uint32_t t1 = __builtin_amdgcn_update_dpp(rb1, m1, 0x138, 0xF, 0xF, false);
                up[0] = t0 & 0xFFFFu; up[1] = t0 >> 16; up[2] = t1 & 0xFFFFu; up[3] = t1 >> 16;
            }
            const uint32_t x0 = (uint32_t)max(q, 0) * PG_Q;
            uint32_t res[PG_Q];
            auto pixel = [&](int k, uint32_t ne) {
                const uint32_t x = x0 + (uint32_t)k;
                int32_t pred;
                if (y == 0) pred = x ? (int32_t)left : 0;                            // first row: left only, corner 0 (:81-91)
                else if (x == 0) pred = (int32_t)up[0];                              // first column: top only (:97-104)
                else pred = mic_grad_predict((int32_t)left, (int32_t)up[k], (int32_t)(k ? up[k - 1] : nw), (int32_t)ne);
                const uint32_t r = ((g.raw >> k) & 1u) ? g.v[k] : (uint32_t)(pred + (int32_t)g.v[k] - thr) & 0xFFFFu;
                res[k] = r; left = r;
            };
            // columns 0..2: NE is the next column of the same group above; at the right edge NE = NW (:113-116)
#pragma unroll
            for (int k = 0; k < PG_Q - 1; k++) {
                const uint32_t x = x0 + (uint32_t)k;
                const uint32_t nwk = k ? up[k - 1] : nw;
                pixel(k, (x + 1 < (uint32_t)W) ? up[k + 1] : nwk);
            }
            // column 3: NE is the first pixel lane r - 1 has just produced for ITS group (one group further right)
            {
                uint32_t rb = 0;
                if (lane == 0) rb = s_grow[min(q + 1, nq) * PG_Q];
                const uint32_t ne_in = __builtin_amdgcn_update_dpp(rb, res[0], 0x138, 0xF, 0xF, false);
                const uint32_t x = x0 + 3u;
                pixel(3, (x + 1 < (uint32_t)W) ? ne_in : up[2]);
            }
            nw = up[PG_Q - 1];
            if (act) {
                const uint32_t p = rowp + (uint32_t)q * PG_Q;
                const int cnt = min(PG_Q, W - q * PG_Q);
                const uint32_t d0 = res[0] | (res[1] << 16), d1 = res[2] | (res[3] << 16);
                if (cnt == PG_Q) { pr_v2 d; d.x = d0; d.y = d1; *(__attribute__((address_space(1))) PrD *)(px + p) = d; }
                else for (int k = 0; k < cnt; k++) px[p + (uint32_t)k] = (uint16_t)res[k];
                if (lane == 63) *(uint2 *)(s_grow + q * PG_Q) = make_uint2(d0, d1);   // the next band's row above (read 63 steps ahead of this write)
            }
#pragma unroll
            for (int k = 0; k < PG_Q; k++) mine[k] = act ? res[k] : mine[k];
            if (!act) { left = 0; if (q < 0) nw = 0; }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                        // wave-private LDS: lane 63's row is in before the next band reads it
    }
}

void mic_launch_decode_pixels(MicUnit *d_units, int n, hipStream_t stream, MicTimer *t, bool any_grad, uint32_t pred_mask) {
#ifndef MIC_NO_FUSED       // (A / B builds: the two-kernel path for every unit)
    if (pred_mask & 0x7Fu) {
        if (t) t->mark("k_dec_rows_tok");
        mic_launch_decode_fused(d_units, n, stream, pred_mask & 0x7Fu);
    }
#endif
    if (t) t->mark("k_dec_pixels_wg");
    hipLaunchKernelGGL(k_dec_pixels_wg, dim3(n), dim3(PX_THREADS), 0, stream, d_units);
    // row-buffer classes so that ordinary widths keep many waves per CU: a row buffer is 2 bytes per column and unit, four units per
    // group; up to 4032 columns that is 8 KiB per unit and twenty units per CU (a latency-bound kernel: one wave per unit)
    if (t) t->mark("k_dec_predict<0>");
    if (pred_mask & MIC_PRED_NARROW)
        hipLaunchKernelGGL((k_dec_predict<0, 16>), dim3((n + 3) / 4), dim3(256), 4 * 1024 * 2, stream, d_units, n, 0, PR_NARROW, 512u);   // narrow frames: 16-pixel groups (8-pixel groups: slower)
    // frames of 1009 .. 2688 columns: row by row, the lanes side by side in a row (k_dec_predict_rows, mic_decode_rows.hip)
    static_assert(PR_NARROW == MIC_ROWS_LO, "the narrow class ends where the row-by-row class begins");
    if (t) t->mark("k_dec_predict_rows");
#ifdef MIC_PREDICT2_ALL      // (A / B builds: the two-wave wavefront kernel of round 3 for those widths too)
    hipLaunchKernelGGL(k_dec_predict2, dim3(n), dim3(128), (1344u + 2 * P2_TILE + 128) * 4, stream, d_units, PR_NARROW, MIC_ROWS_HI, 1344u);
#else
    mic_launch_decode_rows(d_units, n, stream, pred_mask & 0x7Fu);
#endif
    // wider frames: two waves per unit (k_dec_predict2: a mover wave and a computing wave over a 64-row band, 16 KiB of row buffer)
    if (t) t->mark("k_dec_predict2");
    if (pred_mask & MIC_PRED_WAVE2)
        hipLaunchKernelGGL(k_dec_predict2, dim3(n), dim3(128), (4096u + 2 * P2_TILE + 128) * 4, stream, d_units, MIC_ROWS_HI, 8192 - PR_K, 4096u);
    if (t) t->mark("k_dec_predict<wide>");
    if (pred_mask & MIC_PRED_WIDE)
        hipLaunchKernelGGL((k_dec_predict<1, 64>), dim3(n), dim3(64), (PR_MAX_W + PR_K) * 2, stream, d_units, n, 8192 - PR_K, PR_MAX_W, 0u);
    if (any_grad) {
        if (t) t->mark("k_dec_predict_grad");
        static MicPerDeviceOnce once;
        once.run([] { (void)hipFuncSetAttribute((const void *)k_dec_predict_grad, hipFuncAttributeMaxDynamicSharedMemorySize, (PR_MAX_W + 2 * PG_Q) * 2); });
        hipLaunchKernelGGL(k_dec_predict_grad, dim3(n), dim3(64), 8192 * 2, stream, d_units, 0, 8192 - 2 * PG_Q);
        hipLaunchKernelGGL(k_dec_predict_grad, dim3(n), dim3(64), (PR_MAX_W + 2 * PG_Q) * 2, stream, d_units, 8192 - 2 * PG_Q, PR_MAX_W);
    }
}
