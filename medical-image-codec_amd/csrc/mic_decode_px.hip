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
#define PX_K 8                                    // pixels per thread per wavefront step
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

__global__ void __launch_bounds__(PX_THREADS) k_dec_pixels_wg(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK || u.mode != 0) return;
    __shared__ uint32_t s_scan[PX_WAVES + 1];
    __shared__ uint32_t s_fn[PX_WAVES + 1];
    __shared__ uint32_t s_misc[8];
    __shared__ PxVec s_top[2][PX_THREADS];
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
    uint16_t *sym = u.sym;

#ifdef MIC_STAMP
    const uint64_t st0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- phase 1: header walk (wave 0) -----------------------------------------------------
    if (wave == 0) {
        uint32_t pos = 1, outp = 0, nseg = 0, err = 0;
        const uint32_t segcap = u.seg_cap;
        while (pos < ntok && outp < symcap && !err) {
            const uint32_t w = (pos + lane < ntok) ? tok[pos + lane] : 0u;
            uint32_t j = 0;
            while (j < 64 && pos + j < ntok && outp < symcap) {
                const uint32_t h = __builtin_amdgcn_readlane(w, (int)j);
                if (h == 0 || nseg >= segcap) { err = 1; break; }        // count 0 is never written by an encoder
                if (h <= mid) {                                          // same-run: count, value
                    if (pos + j + 1 >= ntok) { err = 1; break; }
                    if (j == 63) break;                                  // value not in the window: reload at the header
                    if (lane == 0) seg[nseg] = make_uint2(pos + j, outp);
                    nseg++; outp += h; j += 2;
                } else {                                                 // literal run
                    if (lane == 0) seg[nseg] = make_uint2(pos + j, outp);
                    nseg++; outp += h - mid; j += 1 + (h - mid);
                }
            }
            pos += j;
        }
        if (lane == 0) { s_misc[0] = nseg; s_misc[1] = min(outp, symcap); s_misc[2] = err; s_misc[3] = 0; }
    }
    __syncthreads();
    const uint32_t nseg = s_misc[0], nsym = s_misc[1];
    if (s_misc[2]) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }

#ifdef MIC_STAMP
    const uint64_t st1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- phase 2: expansion ------------------------------------------------------------------
    {
        uint32_t bad = 0;
        for (uint32_t si = wave; si < nseg; si += PX_WAVES) {
            const uint2 r = seg[si];
            const uint32_t h = tok[r.x];
            if (h <= mid) {
                const uint16_t v = tok[r.x + 1];
                for (uint32_t k = lane; k < h && r.y + k < symcap; k += 64) sym[r.y + k] = v;
            } else {
                const uint32_t cnt = h - mid;
                for (uint32_t k = lane; k < cnt && r.y + k < symcap; k += 64) {
                    if (r.x + 1 + k < ntok) sym[r.y + k] = tok[r.x + 1 + k]; else bad = 1;
                }
            }
        }
        if (bad) s_misc[3] = 1;                                          // literal run past the end (Go: index panic)
    }
    __threadfence_block();
    __syncthreads();
    if (s_misc[3]) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    if (nsym < 1) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t max_value = sym[0];                                  // deltarlecompressu16.go:71
    const int depth = mic_len16(max_value);
    if (depth == 0) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t thr = (1u << (depth - 1)) - 1;
    const uint32_t delim = (1u << depth) - 1;
    uint16_t *px = u.px_out;
    uint32_t *flags = u.flags;

#ifdef MIC_STAMP
    const uint64_t st2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- phase 3: escape markers and pixel numbering -------------------------------------------
    {
        uint32_t carry_marker = 0;       // marker state of the symbol before the tile
        uint32_t carry_px = 0;           // pixels numbered so far
        for (uint32_t base = 1; base < nsym && carry_px < npx; base += PX_THREADS) {
            const uint32_t i = base + tid;
            const bool in = i < nsym;
            const uint32_t x = in ? sym[i] : 0u;
            const bool d = in && x == delim;
            // f: state -> marker[i];  delim: swap (1,0) = 0b01 ; other: zero (0,0) = 0b00
            uint32_t f = d ? 1u : 0u;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                uint32_t o = __shfl_up(f, dd);
                if (lane >= (uint32_t)dd) f = fn_compose(f, o);
            }
            if (lane == 63) s_fn[wave] = f;
            __syncthreads();
            uint32_t st_in = carry_marker;
            for (uint32_t wv = 0; wv < wave; wv++) st_in = (s_fn[wv] >> st_in) & 1;
            const uint32_t marker = (f >> st_in) & 1;
            const uint32_t prev_marker = __shfl_up(marker, 1);
            // marker of the previous symbol: previous lane, or the state entering this wave
            const uint32_t raw = (lane == 0) ? st_in : prev_marker;
            // tile-wide exclusive count of pixels (non-marker symbols)
            const uint32_t is_px = (in && !marker) ? 1u : 0u;
            const uint32_t incl = wave_incl_add(is_px, lane);
            if (lane == 63) s_scan[wave] = incl;
            uint32_t last_marker_tile = 0;
            __syncthreads();
            uint32_t woff = 0, total = 0;
            for (uint32_t wv = 0; wv < PX_WAVES; wv++) { uint32_t v = s_scan[wv]; if (wv < wave) woff += v; total += v; }
            {   // marker state leaving the tile = compose all waves
                uint32_t stt = carry_marker;
                for (uint32_t wv = 0; wv < PX_WAVES; wv++) stt = (s_fn[wv] >> stt) & 1;
                last_marker_tile = stt;
            }
            const uint32_t p = carry_px + woff + incl - is_px;
            if (is_px && p < npx) {
                px[p] = (uint16_t)x;
                if (raw) atomicOr(&flags[p >> 5], 1u << (p & 31));
            }
            carry_px += total;
            carry_marker = last_marker_tile;
            __syncthreads();
        }
        if (carry_px < npx) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }   // tokens ran out (Go: panic)
    }
    __threadfence_block();
    __syncthreads();

#ifdef MIC_STAMP
    const uint64_t st3 = __builtin_amdgcn_s_memtime();
#endif
    // ---- phase 4: inverse predictor, skewed wavefront ---------------------------------------------
    // Thread r owns row r of the band and works on the 8-pixel column group t - r at step t: its top
    // neighbours were produced by thread r-1 one step earlier (LDS hand-off, one barrier per step, 8
    // pixels per barrier), its left neighbour is its own previous pixel.  Symbols arrive and pixels
    // leave as 16-byte vectors (2-byte aligned; gfx950 runs in unaligned-access mode), fetched two
    // steps ahead so the L2 round trip is off the step's critical path.
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
#ifdef MIC_STAMP
    if (tid == 0) {
        const uint64_t st4 = __builtin_amdgcn_s_memtime();
        u.max_count = (uint32_t)(st1 - st0); u.hdr_len = (uint32_t)(st2 - st1); u.zero_bits = (uint32_t)(st3 - st2); u.flavour = (uint32_t)(st4 - st3);
        u.nseg = nseg; u.nsym = nsym;
    }
#endif
}

void mic_launch_decode_pixels(MicUnit *d_units, int n, hipStream_t stream) {
    hipLaunchKernelGGL(k_dec_pixels_wg, dim3(n), dim3(PX_THREADS), 0, stream, d_units);
}
