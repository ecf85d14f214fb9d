// mic_wavelet.hip -- WaveletV2 on the GPU (waveletu16.go, waveletfsecompressu16.go:303-534).
//
// WaveletV2{,SIMD}RLEFSECompressU16 = up to 8 levels of the Le Gall 5/3 integer lifting in Mallat
// layout -> subband scan -> zigzag (+ 3-word escape) -> RLE with length prefix -> 4-state FSE (no
// fallback) behind an 11-byte header.  The reference lifts in place, predict pass then update pass,
// rows then columns, AVX2 over 8-column blocks (wavelet_simd_amd64.s).  Here every output sample is
// written straight from the <= 5 input samples it depends on,
//     d[i] = x[2i+1] - ((x[2i] + x[2i+2]) >> 1)        s[i] = x[2i] + ((d[i-1] + d[i] + 2) >> 2)
// with the reference's boundary rules, so there is no serial dependence at all; a LEVEL is one kernel (k_wv_fwd2d / k_wv_inv2d:
// rows and columns over a tile in LDS, de-interleaving on the fly, the smooth plane handed from level to level in a scratch plane).
// The subband scan is a closed-form index map, so collect + zigzag + escape + max is one pass with a
// prefix sum for the rare escapes.  RLE / FSE reuse the unit-codec kernels (mode 2 / mode 1 units).
#include "mic_session.h"

namespace {

#define WV_THREADS 1024
#define WV_WAVES 16

struct WvDims { int rows, cols, levels; int nr[9], nc[9]; };

__host__ __device__ inline WvDims wv_dims(int rows, int cols, int levels) {
    WvDims d; d.rows = rows; d.cols = cols; d.levels = levels;
    d.nr[0] = rows; d.nc[0] = cols;
    for (int l = 1; l <= 8; l++) { d.nr[l] = (d.nr[l - 1] + 1) / 2; d.nc[l] = (d.nc[l - 1] + 1) / 2; }
    return d;
}

// ---- lifting, one output sample at a time (waveletu16.go:26-122) ---------------------------------
template <typename Get>
__device__ __forceinline__ int32_t wv_d(Get x, int n, int i) {           // detail i (0 <= i < n/2)
    const int32_t l = x(2 * i), r = (2 * i + 2 < n) ? x(2 * i + 2) : l;  // symmetric extension on the right
    return x(2 * i + 1) - ((l + r) >> 1);
}
template <typename Get>
__device__ __forceinline__ int32_t wv_s(Get x, int n, int i) {           // smooth i (0 <= i < (n+1)/2)
    int32_t d_right, d_left;
    if (2 * i + 1 < n) d_right = wv_d(x, n, i);
    else d_right = (i > 0) ? wv_d(x, n, i - 1) : 0;
    d_left = (i > 0) ? wv_d(x, n, i - 1) : d_right;
    return x(2 * i) + ((d_left + d_right + 2) >> 2);
}
// inverse: c(k) reads the Mallat-ordered line (low half first); returns sample k of the restored line
template <typename Get>
__device__ __forceinline__ int32_t wv_even(Get c, int n, int i) {        // x[2i]
    const int n_low = (n + 1) / 2;
    auto d = [&](int k) { return c(n_low + k); };
    int32_t d_right, d_left;
    if (2 * i + 1 < n) d_right = d(i);
    else d_right = (i > 0) ? d(i - 1) : 0;
    d_left = (i > 0) ? d(i - 1) : d_right;
    return c(i) - ((d_left + d_right + 2) >> 2);
}
template <typename Get>
__device__ __forceinline__ int32_t wv_sample(Get c, int n, int k) {
    if (n < 2) return c(k);
    const int n_low = (n + 1) / 2;
    if ((k & 1) == 0) return wv_even(c, n, k >> 1);
    const int i = k >> 1;
    const int32_t l = wv_even(c, n, i), r = (2 * i + 2 < n) ? wv_even(c, n, i + 1) : l;
    return c(n_low + i) + ((l + r) >> 1);
}

// inverse: columns first, then rows   (waveletu16.go:213-257); a thread restores samples 2i and 2i+1 of its line
template <typename Get>
__device__ __forceinline__ void wv_pair(Get cf, int n, int i, int32_t &ev, int32_t &od) {
    if (n < 2) { ev = cf(0); od = 0; return; }
    const int n_low = (n + 1) / 2;
    ev = wv_even(cf, n, i);
    od = 0;
    if (2 * i + 1 < n) { const int32_t rr = (2 * i + 2 < n) ? wv_even(cf, n, i + 1) : ev; od = cf(n_low + i) + ((ev + rr) >> 1); }
}
// ---- one level of the transform in ONE pass, no LDS ------------------------------------------------------------------------------
// A lane owns one column PAIR (samples 2 gi, 2 gi + 1 of every row) and walks down a strip of rows.  The rows pass of a row needs
// the neighbour lanes' samples only -- x[2 gi + 2] from the lane to the right, d[gi - 1] from the lane to the left: two DPP wave
// shifts -- and the columns pass is a window of three rows of the rows pass's output that slides down the lane's registers.  A wave
// covers 64 pairs and writes 62 (one pair of context on either side), a strip is WS_ROWS output rows (+ one pair of rows of
// context at its top); every access is a run of 256 bytes (128 for 16-bit pixels) per wave, the addresses advance by a stride a row.
// (Rounds 2-3: rows and columns over a 64 x 16 tile in LDS -- one pass over memory as well, 9 N bytes a transform, but ~100
// instructions of tile index arithmetic per sample: 4.9 + 6.4 ms for 256 CR frames both ways where the memory traffic is ~4.)
// The detail subbands go to their Mallat places in `a` (row stride `stride`); the smooth subband -- the next level's input -- goes to
// a compact scratch plane (`ll`, row stride = its width), except the last level's, which goes to the top-left corner of `a`:
// nothing is read and written in the same buffer by one launch.
#define WS_ROWS 64
#define WS_LANES 62                 // pairs a wave writes
template <int CTRL> __device__ __forceinline__ int32_t wv_shift(int32_t v) {    // 0x130: from the lane to the right, 0x138: from the lane to the left
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
typedef uint32_t wv_u32u __attribute__((aligned(2)));
typedef unsigned long long wv_u64u __attribute__((aligned(4)));

template <bool FROM_U16>
__global__ void __launch_bounds__(256) k_wv_fwd2d(const void *__restrict__ src_, int sstride, size_t sfs, int32_t *__restrict__ a, int stride, size_t fs,
                                                  int32_t *__restrict__ ll, int llstride, size_t llfs, int r, int c, int nf) {
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int nlc = (c + 1) / 2, nlr = (r + 1) / 2;
    const int gi = ((int)blockIdx.x * 4 + wave) * WS_LANES - 1 + lane;          // this lane's pair (lanes 0 and 63: context only)
    const int gp0 = (int)blockIdx.y * WS_ROWS, gp1 = min(gp0 + WS_ROWS, nlr);
    if (((int)blockIdx.x * 4 + wave) * WS_LANES >= nlc) return;
    const bool has0 = gi >= 0 && 2 * gi < c, has1 = gi >= 0 && 2 * gi + 1 < c, hasr = 2 * gi + 2 < c, left = gi > 0;
    const bool out_lo = lane >= 1 && lane <= WS_LANES && has0, out_hi = lane >= 1 && lane <= WS_LANES && has1;
    for (int f = (int)blockIdx.z; f < nf; f += (int)gridDim.z) {
        // input row y: this lane's two samples ...
        auto fetch = [&](int y, int32_t &x0, int32_t &x1) {
            x0 = 0; x1 = 0;
            const size_t o = (size_t)f * sfs + (size_t)y * sstride + (size_t)(2 * gi);
            if (FROM_U16) {
                const uint16_t *p = (const uint16_t *)src_ + o;
                if (has1) { const uint32_t w = *(const wv_u32u *)p; x0 = (int32_t)(w & 0xFFFFu); x1 = (int32_t)(w >> 16); }
                else if (has0) x0 = (int32_t)p[0];
            } else {
                const int32_t *p = (const int32_t *)src_ + o;
                if (has1) { const unsigned long long w = *(const wv_u64u *)p; x0 = (int32_t)(uint32_t)w; x1 = (int32_t)(uint32_t)(w >> 32); }
                else if (has0) x0 = p[0];
            }
        };
        // ... and its rows pass (waveletu16.go:26-74, :170-182): smooth and detail of the pair
        auto lift = [&](int32_t x0, int32_t x1, int32_t &lo, int32_t &hi) {
            const int32_t xn = wv_shift<0x130>(x0), xr = hasr ? xn : x0;       // symmetric extension on the right
            const int32_t d = x1 - ((x0 + xr) >> 1);
            const int32_t dp = wv_shift<0x138>(d);
            const int32_t dr = has1 ? d : (left ? dp : 0), dl = left ? dp : dr;
            lo = x0 + ((dl + dr + 2) >> 2); hi = d;
        };
        auto row = [&](int y, int32_t &lo, int32_t &hi) { int32_t x0, x1; fetch(y, x0, x1); lift(x0, x1, lo, hi); };
        // the columns pass (:183-208) down the strip: rows 2 gp, 2 gp + 1 in (l0, h0), (l1, h1); row 2 gp + 2 in (ln, hn)
        int32_t l0, h0, l1 = 0, h1 = 0, ln = 0, hn = 0, dlo = 0, dhi = 0;       // dlo, dhi: the details of pair-row gp - 1
        if (gp0 > 0) {
            int32_t pl0, ph0, pl1, ph1;
            row(2 * gp0 - 2, pl0, ph0); row(2 * gp0 - 1, pl1, ph1); row(2 * gp0, l0, h0);
            dlo = pl1 - ((pl0 + l0) >> 1); dhi = ph1 - ((ph0 + h0) >> 1);
        } else row(0, l0, h0);
        if (2 * gp0 + 1 < r) row(2 * gp0 + 1, l1, h1);
        for (int gp = gp0; gp < gp1; gp++) {
            const bool odd = 2 * gp + 1 < r, more = 2 * gp + 2 < r, more1 = 2 * gp + 3 < r;
            int32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;                              // (both rows of the next pair are fetched before either is used)
            if (more) fetch(2 * gp + 2, a0, a1);
            if (more1) fetch(2 * gp + 3, b0, b1);
            if (more) lift(a0, a1, ln, hn);
            const int32_t rl = more ? ln : l0, rh = more ? hn : h0;
            const int32_t d_l = l1 - ((l0 + rl) >> 1), d_h = h1 - ((h0 + rh) >> 1);
            const int32_t drl = odd ? d_l : (gp > 0 ? dlo : 0), drh = odd ? d_h : (gp > 0 ? dhi : 0);
            const int32_t dll = gp > 0 ? dlo : drl, dlh = gp > 0 ? dhi : drh;
            const int32_t s_l = l0 + ((dll + drl + 2) >> 2), s_h = h0 + ((dlh + drh + 2) >> 2);
            if (out_lo) ll[(size_t)f * llfs + (size_t)gp * llstride + gi] = s_l;
            if (out_hi) a[(size_t)f * fs + (size_t)gp * stride + nlc + gi] = s_h;
            if (odd) {
                if (out_lo) a[(size_t)f * fs + (size_t)(nlr + gp) * stride + gi] = d_l;
                if (out_hi) a[(size_t)f * fs + (size_t)(nlr + gp) * stride + nlc + gi] = d_h;
            }
            dlo = d_l; dhi = d_h; l0 = ln; h0 = hn;
            if (more1) lift(b0, b1, l1, h1);
        }
    }
}
// The inverse of one level (columns, then rows: waveletu16.go:213-257): the smooth subband from `ll` (the level below wrote it; the
// coarsest level's lies in `a`), the detail subbands from `a`, the restored region to `dst` (a compact plane, or the 16-bit pixels).
// A lane owns smooth column gi and detail column gi of the Mallat layout -- after the rows pass, samples 2 gi and 2 gi + 1.
template <bool TO_U16>
__global__ void __launch_bounds__(256) k_wv_inv2d(const int32_t *__restrict__ a, int stride, size_t fs, const int32_t *__restrict__ ll, int llstride, size_t llfs,
                                                  void *__restrict__ dst_, int dstride, size_t dfs, int r, int c, int nf) {
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int nlc = (c + 1) / 2, nlr = (r + 1) / 2;
    const int gi = ((int)blockIdx.x * 4 + wave) * WS_LANES - 1 + lane;
    const int gp0 = (int)blockIdx.y * WS_ROWS, gp1 = min(gp0 + WS_ROWS, nlr);
    if (((int)blockIdx.x * 4 + wave) * WS_LANES >= nlc) return;
    const bool has0 = gi >= 0 && 2 * gi < c, has1 = gi >= 0 && 2 * gi + 1 < c, hasr = 2 * gi + 2 < c, left = gi > 0;
    const bool out0 = lane >= 1 && lane <= WS_LANES && has0;
    for (int f = (int)blockIdx.z; f < nf; f += (int)gridDim.z) {
        const int32_t *af = a + (size_t)f * fs, *lf = ll + (size_t)f * llfs;
        // Mallat row j of this lane's two columns: smooth rows j < nlr, detail rows nlr + j
        auto smooth = [&](int j, int32_t &vs, int32_t &vd) {
            vs = has0 ? lf[(size_t)j * llstride + gi] : 0;
            vd = has1 ? af[(size_t)j * stride + nlc + gi] : 0;
        };
        auto detail = [&](int j, int32_t &vs, int32_t &vd) {
            vs = has0 ? af[(size_t)(nlr + j) * stride + gi] : 0;
            vd = has1 ? af[(size_t)(nlr + j) * stride + nlc + gi] : 0;
        };
        // even sample j of a column from its smooth value and the details on either side (wv_even's rules)
        auto even = [&](int j, int32_t cs, int32_t dm, int32_t d0) -> int32_t {
            const int32_t dr = (2 * j + 1 < r) ? d0 : (j > 0 ? dm : 0), dl = j > 0 ? dm : dr;
            return cs - ((dl + dr + 2) >> 2);
        };
        // one restored row of the columns pass through the rows pass, to `dst`
        auto put = [&](int y, int32_t vs, int32_t vd) {
            const int32_t dp = wv_shift<0x138>(vd);
            const int32_t dr = has1 ? vd : (left ? dp : 0), dl = left ? dp : dr;
            const int32_t ev = vs - ((dl + dr + 2) >> 2);
            const int32_t en = wv_shift<0x130>(ev);
            const int32_t od = vd + ((ev + (hasr ? en : ev)) >> 1);
            if (!out0) return;
            const size_t o = (size_t)f * dfs + (size_t)y * dstride + (size_t)(2 * gi);
            if (TO_U16) {
                uint16_t *d = (uint16_t *)dst_ + o;
                if (has1) *(wv_u32u *)d = ((uint32_t)ev & 0xFFFFu) | ((uint32_t)od << 16); else d[0] = (uint16_t)ev;
            } else {
                int32_t *d = (int32_t *)dst_ + o;
                if (has1) *(wv_u64u *)d = (uint32_t)ev | ((unsigned long long)(uint32_t)od << 32); else d[0] = ev;
            }
        };
        // the columns pass down the strip: (es, ed) = even sample gp of the two columns, (ds, dd) = their details gp
        int32_t ss, sd, ds = 0, dd = 0, pms = 0, pmd = 0;
        if (gp0 > 0) detail(gp0 - 1, pms, pmd);
        smooth(gp0, ss, sd);
        if (2 * gp0 + 1 < r) detail(gp0, ds, dd);
        int32_t es = even(gp0, ss, pms, ds), ed = even(gp0, sd, pmd, dd);
        for (int gp = gp0; gp < gp1; gp++) {
            const bool odd = 2 * gp + 1 < r, more = 2 * gp + 2 < r;
            int32_t ns = 0, nd = 0, nds = 0, ndd = 0, es1 = es, ed1 = ed;
            if (more) {
                smooth(gp + 1, ns, nd);
                if (2 * gp + 3 < r) detail(gp + 1, nds, ndd);
                es1 = even(gp + 1, ns, ds, nds); ed1 = even(gp + 1, nd, dd, ndd);
            }
            put(2 * gp, es, ed);
            if (odd) put(2 * gp + 1, ds + ((es + es1) >> 1), dd + ((ed + ed1) >> 1));
            es = es1; ed = ed1; ds = nds; dd = ndd;
        }
    }
}
__global__ void __launch_bounds__(256) k_wv_load(const uint16_t *px, int32_t *a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (int32_t)px[i];
}
__global__ void __launch_bounds__(256) k_wv_store(const int32_t *a, uint16_t *px, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) px[i] = (uint16_t)a[i];
}

// collectSubbandOrder / scatterSubbandOrder as an index map (waveletfsecompressu16.go:202-282):
// linear position p -> element offset y*cols + x.  LL of the coarsest level, then HL, LH, HH from the
// coarsest level to the finest.
__device__ __forceinline__ size_t wv_pos_to_index(const WvDims &d, size_t p) {
    const int L = d.levels;
    size_t sz = (size_t)d.nr[L] * d.nc[L];
    if (p < sz) return (p / d.nc[L]) * d.cols + (p % d.nc[L]);
    p -= sz;
    for (int l = L; l >= 1; l--) {
        const int hw = d.nc[l - 1] - d.nc[l], lh = d.nr[l - 1] - d.nr[l];
        sz = (size_t)d.nr[l] * hw;                                                        // HL: rows [0,nr[l]), cols [nc[l],nc[l-1])
        if (p < sz) return (p / hw) * d.cols + d.nc[l] + (p % hw);
        p -= sz;
        sz = (size_t)lh * d.nc[l];                                                        // LH: rows [nr[l],nr[l-1]), cols [0,nc[l])
        if (p < sz) return (d.nr[l] + p / d.nc[l]) * d.cols + (p % d.nc[l]);
        p -= sz;
        sz = (size_t)lh * hw;                                                             // HH
        if (p < sz) return (d.nr[l] + p / hw) * d.cols + d.nc[l] + (p % hw);
        p -= sz;
    }
    return 0;
}

__device__ __forceinline__ uint32_t wv_wave_incl(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { uint32_t o = __shfl_up(v, dd); if (lane >= (uint32_t)dd) v += o; }
    return v;
}

// subband scan + waveletCoeffsToU16 (:28-40) + zzMax (:335-349).  One work-group per image; writes the
// symbol stream into u.sym, its length into u.nsym and the RLE maxValue into u.max_value.
__global__ void __launch_bounds__(WV_THREADS) k_wv_symbols(MicUnit *units, const int32_t *a, WvDims d) {
    MicUnit &u = units[blockIdx.x];
    if (!u.wv_slow) return;                                               // k_wv_symbols_par did this frame
    a += (size_t)blockIdx.x * (size_t)d.rows * (size_t)d.cols;             // one group per frame of the batch
    __shared__ uint32_t s_scan[WV_WAVES], s_max[WV_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t n = (size_t)d.rows * d.cols;
    uint16_t *sym = u.sym; const uint32_t cap = u.sym_cap;
    uint32_t carry = 0, zmax = 0; bool ovf = false;
    for (size_t base = 0; base < n; base += WV_THREADS) {
        const size_t p = base + tid;
        int32_t v = 0; uint32_t cnt = 0;
        if (p < n) { v = a[wv_pos_to_index(d, p)]; cnt = (v >= -32767 && v <= 32767) ? 1u : 3u; }
        const uint32_t incl = wv_wave_incl(cnt, lane);
        __syncthreads();
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < WV_WAVES; w++) { const uint32_t x = s_scan[w]; if ((uint32_t)w < wave) woff += x; tot += x; }
        const uint32_t o = carry + woff + incl - cnt;
        if (cnt == 1) {
            const uint32_t z = (uint32_t)((v >> 31) ^ (int32_t)((uint32_t)v << 1)) & 0xFFFF;   // zigzagEncode16, :538-541
            if (o < cap) sym[o] = (uint16_t)z; else ovf = true;
            zmax = max(zmax, z);
        } else if (cnt == 3) {
            if (o + 2 < cap) { sym[o] = 65535; sym[o + 1] = (uint16_t)((uint32_t)v >> 16); sym[o + 2] = (uint16_t)(uint32_t)v; } else ovf = true;
            zmax = 65535;
        }
        carry += tot;
    }
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) zmax = max(zmax, (uint32_t)__shfl_xor((int)zmax, dd));
    __syncthreads();
    if (lane == 0) s_max[wave] = zmax;
    const int any_ovf = __syncthreads_or(ovf ? 1 : 0);
    if (tid == 0) {
        uint32_t m = 0;
        for (int w = 0; w < WV_WAVES; w++) m = max(m, s_max[w]);
        int depth = m ? 32 - __clz(m) : 0;
        if (depth < 1) depth = 1;                                                         // :345-348
        u.max_value = (uint16_t)((1u << depth) - 1);
        u.nsym = carry;
        u.status = any_ovf ? MICD_ERR_CAPACITY : MICD_OK;
    }
}

// tokens -> symbols: RleDecompressU16.Init + Decompress (rledecompressu16.go:21-30, :87-97); header walk by
// wave 0, expansion by all waves (as in mic_decode_px.hip), length taken from the two prefix words.
// One work-group per unit; mode_filter >= 0 restricts the launch to units of that mode (MIC2 temporal residuals).
__global__ void __launch_bounds__(WV_THREADS) k_wv_expand(MicUnit *units, int mode_filter, int only_slow) {
    MicUnit &u = units[blockIdx.x];
    if (mode_filter >= 0 && u.mode != (uint32_t)mode_filter) return;
    if (only_slow && !u.wv_slow) return;                                  // k_wv_scatter did this frame
    if (u.status != MICD_OK) return;
    __shared__ uint32_t s_misc[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ntok = u.ntok; const uint16_t *tok = u.tok;
    if (ntok < 3) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    const int d0 = mic_len16(tok[0]);
    if (d0 == 0) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t mid = (1u << (d0 - 1)) - 1;
    const uint32_t outlen = ((uint32_t)tok[1] << 16) + tok[2];
    if (outlen > u.sym_cap) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    uint2 *seg = u.seg; uint16_t *sym = u.sym;
    if (wave == 0) {
        uint32_t pos = 3, outp = 0, nseg = 0, err = 0;
        const uint32_t segcap = u.seg_cap;
        while (pos < ntok && outp < outlen && !err) {
            const uint32_t w = (pos + lane < ntok) ? tok[pos + lane] : 0u;
            uint32_t j = 0;
            while (j < 64 && pos + j < ntok && outp < outlen) {
                uint32_t h = __builtin_amdgcn_readlane(w, (int)j);
                if (nseg >= segcap) { err = 1; break; }
                if (h == 0) h = 65536u;                                  // the reference's reading of a zero count: a literal chunk of 65536 - midCount (mic_decode_px.hip)
                if (h <= mid) {
                    if (pos + j + 1 >= ntok) { err = 1; break; }
                    if (j == 63) break;
                    if (lane == 0) seg[nseg] = make_uint2(pos + j, outp);
                    nseg++; outp += h; j += 2;
                } else {
                    if (lane == 0) seg[nseg] = make_uint2(pos + j, outp);
                    nseg++; outp += h - mid; j += 1 + (h - mid);
                }
            }
            pos += j;
        }
        if (outp < outlen) err = 1;                                    // tokens ran out (Go: index panic)
        if (lane == 0) { s_misc[0] = nseg; s_misc[1] = err; s_misc[2] = 0; }
    }
    __syncthreads();
    if (s_misc[1]) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    const uint32_t nseg = s_misc[0];
    uint32_t bad = 0;
    for (uint32_t si = wave; si < nseg; si += WV_WAVES) {
        const uint2 r = seg[si];
        uint32_t h = tok[r.x];
        if (h == 0) h = 65536u;                                          // (a zero count: see the walker above)
        if (h <= mid) {
            const uint16_t v = tok[r.x + 1];
            for (uint32_t k = lane; k < h && r.y + k < outlen; k += 64) sym[r.y + k] = v;
        } else {
            const uint32_t cnt = h - mid;
            for (uint32_t k = lane; k < cnt && r.y + k < outlen; k += 64) { if (r.x + 1 + k < ntok) sym[r.y + k] = tok[r.x + 1 + k]; else bad = 1; }
        }
    }
    if (__syncthreads_or((int)bad)) { if (tid == 0) u.status = MICD_ERR_CORRUPT; return; }
    if (tid == 0) u.nsym = outlen;
}

// u16ToWaveletCoeffs (:43-58) + scatterSubbandOrder (:244-282): symbol stream -> Mallat image.
// A symbol is an escape marker iff it is 65535 and not one of the two payload words of a previous
// marker; with three-word escapes that is a 3-state recurrence, scanned as function composition.
__device__ __forceinline__ uint32_t wv_fn_compose(uint32_t g, uint32_t f) {      // functions on {0,1,2}, 2 bits per value
    const uint32_t f0 = f & 3, f1 = (f >> 2) & 3, f2 = (f >> 4) & 3;
    return ((g >> (2 * f0)) & 3) | (((g >> (2 * f1)) & 3) << 2) | (((g >> (2 * f2)) & 3) << 4);
}
__global__ void __launch_bounds__(WV_THREADS) k_wv_coeffs(MicUnit *units, int32_t *a, WvDims d) {
    MicUnit &u = units[blockIdx.x];
    a += (size_t)blockIdx.x * (size_t)d.rows * (size_t)d.cols;
    if (u.status != MICD_OK || !u.wv_slow) return;
    __shared__ uint32_t s_scan[WV_WAVES], s_fn[WV_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t m = u.nsym; const uint16_t *sym = u.sym;
    const size_t n = (size_t)d.rows * d.cols;
    // state = payload words still to skip (0: a symbol starts here).  marker: 0 -> 2; payload: k -> k-1.
    // non-escape symbol: f = (0, 0, 1) i.e. state 0 -> 0, 1 -> 0, 2 -> 1 ; escape value 65535: f = (2, 0, 1)
    uint32_t carry_state = 0, carry_n = 0; bool bad = false;
    for (uint32_t base = 0; base < m && carry_n < n; base += WV_THREADS) {
        const uint32_t i = base + tid;
        const bool in = i < m;
        const uint32_t x = in ? sym[i] : 0u;
        uint32_t f = (x == 65535u && in) ? (2u | (0u << 2) | (1u << 4)) : (0u | (0u << 2) | (1u << 4));
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = __shfl_up(f, dd); if (lane >= (uint32_t)dd) f = wv_fn_compose(f, o); }
        __syncthreads();
        if (lane == 63) s_fn[wave] = f;
        __syncthreads();
        uint32_t st_in = carry_state;
        for (uint32_t w = 0; w < wave; w++) st_in = (s_fn[w] >> (2 * st_in)) & 3;
        const uint32_t st_after = (f >> (2 * st_in)) & 3;                          // state after symbol i
        uint32_t st_before = __shfl_up(st_after, 1); if (lane == 0) st_before = st_in;
        const bool starts = in && st_before == 0;                                  // a coefficient begins at i
        const uint32_t incl = wv_wave_incl(starts ? 1u : 0u, lane);
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0, st_end = carry_state;
#pragma unroll
        for (int w = 0; w < WV_WAVES; w++) { const uint32_t v = s_scan[w]; if ((uint32_t)w < wave) woff += v; tot += v; st_end = (s_fn[w] >> (2 * st_end)) & 3; }
        if (starts) {
            const size_t k = (size_t)carry_n + woff + incl - 1;
            if (k < n) {
                int32_t v;
                if (x != 65535u) v = (int32_t)((x >> 1) ^ (uint32_t)(-(int32_t)(x & 1)));        // zigzagDecode16, :543-546
                else if (i + 2 < m) v = (int32_t)(((uint32_t)sym[i + 1] << 16) | (uint32_t)sym[i + 2]);
                else { v = 0; bad = true; }
                a[wv_pos_to_index(d, k)] = v;
            }
        }
        carry_n += tot; carry_state = st_end;
    }
    const int anybad = __syncthreads_or(bad ? 1 : 0);
    if (tid == 0 && (anybad || carry_n < n)) u.status = MICD_ERR_CORRUPT;          // fewer coefficients than pixels (Go: panic)
}

// ---- the usual case in parallel: frames whose coefficients all fit 16 bits (no 3-word escapes), many groups per frame -------------
// Scan position p -> element offset, with the count of positions from p to the end of its subband row (they are consecutive
// in memory): 32-bit arithmetic (a frame has at most 2^27 samples).
__device__ __forceinline__ void wv_pos_row(const WvDims &d, uint32_t p, uint32_t &idx, uint32_t &left) {
    const int L = d.levels;
    const uint32_t cols = (uint32_t)d.cols;
    {
        const uint32_t w = (uint32_t)d.nc[L], sz = (uint32_t)d.nr[L] * w;
        if (p < sz) { const uint32_t y = p / w, x = p - y * w; idx = y * cols + x; left = w - x; return; }
        p -= sz;
    }
    for (int l = L; l >= 1; l--) {
        const uint32_t nc = (uint32_t)d.nc[l], nr = (uint32_t)d.nr[l], hw = (uint32_t)d.nc[l - 1] - nc, lh = (uint32_t)d.nr[l - 1] - nr;
        uint32_t sz = nr * hw;
        if (p < sz) { const uint32_t y = p / hw, x = p - y * hw; idx = y * cols + nc + x; left = hw - x; return; }
        p -= sz;
        sz = lh * nc;
        if (p < sz) { const uint32_t y = p / nc, x = p - y * nc; idx = (nr + y) * cols + x; left = nc - x; return; }
        p -= sz;
        sz = lh * hw;
        if (p < sz) { const uint32_t y = p / hw, x = p - y * hw; idx = (nr + y) * cols + nc + x; left = hw - x; return; }
        p -= sz;
    }
    idx = 0; left = 1;
}
#define WS_T 8192                                  // scan positions per group (a power of two: >> 13 below)
typedef uint32_t wv_v4 __attribute__((ext_vector_type(4)));
typedef wv_v4 WvQ2 __attribute__((aligned(2)));
typedef wv_v4 WvQ4 __attribute__((aligned(4)));
// subband scan + zigzag, WS_T positions per group (grid: x = position tiles, y = frames).  A coefficient outside 16 bits sends the
// frame to k_wv_symbols (wv_slow); the frame's largest symbol is gathered with an atomic max.
__global__ void __launch_bounds__(1024) k_wv_symbols_par(MicUnit *units, const int32_t *a, WvDims d, int nf) {
    const uint32_t n = (uint32_t)d.rows * (uint32_t)d.cols, tid = threadIdx.x, lane = tid & 63;
    const uint32_t i0 = blockIdx.x * WS_T + tid * 8;
    for (int f = (int)blockIdx.y; f < nf; f += (int)gridDim.y) {
        MicUnit &u = units[f];
        const int32_t *af = a + (size_t)f * n;
        uint32_t w8[4] = { 0u, 0u, 0u, 0u }, zmax = 0, idx = 0, left = 0; bool wide = false;
        int32_t v8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (i0 < n) wv_pos_row(d, i0, idx, left);
        if (i0 + 8 <= n && left >= 8) {                                    // the usual case: eight positions inside one subband row
            const wv_v4 lo = *(const WvQ4 *)(af + idx), hi = *(const WvQ4 *)(af + idx + 4);
            v8[0] = (int32_t)lo.x; v8[1] = (int32_t)lo.y; v8[2] = (int32_t)lo.z; v8[3] = (int32_t)lo.w;
            v8[4] = (int32_t)hi.x; v8[5] = (int32_t)hi.y; v8[6] = (int32_t)hi.z; v8[7] = (int32_t)hi.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t p = i0 + (uint32_t)k;
                if (p < n) {
                    if (left == 0) wv_pos_row(d, p, idx, left);
                    v8[k] = af[idx]; idx++; left--;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int32_t v = v8[k];
            if (v < -32767 || v > 32767) wide = true;
            const uint32_t z = (uint32_t)((v >> 31) ^ (int32_t)((uint32_t)v << 1)) & 0xFFFF;   // zigzagEncode16, :538-541
            zmax = max(zmax, z);
            w8[k >> 1] |= z << (16 * (k & 1));
        }
        uint16_t *sym = u.sym;
        if (i0 + 8 <= n) { wv_v4 v; v.x = w8[0]; v.y = w8[1]; v.z = w8[2]; v.w = w8[3]; *(wv_v4 *)(sym + i0) = v; }   // (256-byte aligned slab, i0 a multiple of 8)
        else for (int k = 0; k < 8; k++) if (i0 + (uint32_t)k < n) sym[i0 + k] = (uint16_t)(w8[k >> 1] >> (16 * (k & 1)));
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) zmax = max(zmax, (uint32_t)__shfl_xor((int)zmax, dd));
        // (one address per frame: an atomic per wave was 59 k atomics in a row on it.  The maximum only grows, so a wave whose own is
        // not above what the frame has already -- nearly every wave after the first few -- has nothing to say; a stale read only costs
        // an atomic that changes nothing)
        if (lane == 0 && zmax > __hip_atomic_load(&u.wv_zmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&u.wv_zmax, zmax);
        if (wide) u.wv_slow = 1;
    }
}
// zzMax (:335-349) of the frames k_wv_symbols_par finished
__global__ void __launch_bounds__(256) k_wv_symbols_fin(MicUnit *units, WvDims d, int nf) {
    const int f = (int)(blockIdx.x * 256 + threadIdx.x);
    if (f >= nf) return;
    MicUnit &u = units[f];
    if (u.wv_slow) return;
    const uint32_t m = u.wv_zmax;
    int depth = m ? 32 - __clz(m) : 0;
    if (depth < 1) depth = 1;
    u.max_value = (uint16_t)((1u << depth) - 1);
    u.nsym = (uint32_t)d.rows * (uint32_t)d.cols;
    u.status = (u.nsym <= u.sym_cap) ? MICD_OK : MICD_ERR_CAPACITY;
}

// ---- the RLE header walk of a whole frame, in parts -----------------------------------------------------------------------------
// A WaveletV2 frame is ONE token stream of millions of tokens whose headers form a linked list (rledecompressu16.go:59-85: the
// next header's position is this one's value), and a walk costs a wave some hundreds of cycles per header whatever else the GPU
// does.  But the list is self-synchronising: a walk started at a token that is no header lands on a true one within a few steps
// and is the true walk from there.  So the stream is cut into WP_PARTS parts, a wave per part walks from the part's first token
// (part 0: from the first header; a walk that takes a zero token for a header has proved itself wrong -- no encoder writes a zero
// count, and zero symbols are common in literal chunks -- and starts over behind it) and leaves its records {payload token | run
// flag, first symbol RELATIVE to the part} in a region of its own; k_rle_walk_fix then hops from part to part -- the true walk
// enters a part at the exit of the part before; the true headers the part's own walk has not got are taken from a 64-token window
// (up to WP_EXTRA of them, else the frame goes to k_wv_expand) until it stands on a position the part has a record of -- and
// k_rle_walk_compact moves the true records to the front of `seg` with their absolute symbol positions, filling `flags` with
// the segment that holds every WS_T-th symbol.  A zero count that IS on the true walk is the reference's literal chunk of
// 65536 - midCount (k_wv_expand reads it so): the fix kernel hands such frames over.  Stop and error rules are k_wv_expand's.
#define WP_PARTS 64
#define WP_MINLEN 4096u
#define WP_EXTRA 1024u                             // true headers a part may hold in front of the point where its own walk joins the true one
// per part, in the unit's cumul[] slab (free after the tables)
struct WpPart { uint32_t start, exit, nrec, out_total, err, first, base_seg, out_delta, nextra, extra_base, pad0, pad1; };
__device__ __forceinline__ uint32_t wp_part_len(uint32_t ntok) { return max(WP_MINLEN, (((ntok + WP_PARTS - 1) / WP_PARTS) + 63u) & ~63u); }
__device__ __forceinline__ uint32_t wp_stride(uint32_t L) { return L / 2 + 1 + WP_EXTRA + 1; }     // records of a part: its own walk's, then the extras (+ an end mark)
__device__ __forceinline__ bool wp_fits(const MicUnit &u, uint32_t L) {        // temporary records in the upper half of seg, final ones in the lower
    const uint32_t nparts = (u.ntok + L - 1) / L;
    return u.seg != nullptr && (uint64_t)nparts * wp_stride(L) <= u.seg_cap / 2 && u.ntok / 2 + 2 <= u.seg_cap / 2;
}
__global__ void __launch_bounds__(64) k_rle_walk_parts(MicUnit *units) {
    MicUnit &u = units[blockIdx.y];
    if (u.status != MICD_OK || u.walk_mode != 1 || u.walk_ok != 3) return;
    const uint32_t ntok = u.ntok, p = blockIdx.x, lane = threadIdx.x;
    if (ntok < 3) return;
    const uint16_t *tok = u.tok;
    const uint32_t L = wp_part_len(ntok), lo = p * L;
    if (lo >= ntok || !wp_fits(u, L)) return;
    const uint32_t hi = min(lo + L, ntok);
    const int d0 = mic_len16(tok[0]);
    if (d0 == 0) return;
    const uint32_t mid = (1u << (d0 - 1)) - 1;
    uint2 *rec = u.seg + u.seg_cap / 2 + (size_t)p * wp_stride(L);
    uint32_t pos = p ? lo : 3u, out = 0, nrec = 0, err = 0;
    const uint32_t start = pos;
    while (pos < hi && !err) {
        const uint32_t w = (pos + lane < ntok) ? tok[pos + lane] : 0u;
        uint32_t j = 0;
        while (j < 64 && pos + j < hi) {
            const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)j);
            if (h == 0) {
                // an encoder never writes a zero count.  The first part's walk is the true one: an error; any other part's walk started at
                // some token and has just proved itself wrong (zero symbols are common in literal chunks): it starts over behind the zero
                if (p == 0) { err = 1; break; }
                nrec = 0; out = 0; j += 1;
                continue;
            }
            if (h <= mid) {
                if (pos + j + 1 >= ntok) { if (p == 0) err = 1; else { nrec = 0; out = 0; j = 64; pos = hi; } break; }
                if (lane == 0) rec[nrec] = make_uint2((pos + j + 1) | 0x80000000u, out);
                nrec++; out += h; j += 2;
            } else {
                if (lane == 0) rec[nrec] = make_uint2(pos + j + 1, out);
                nrec++; out += h - mid; j += 1 + (h - mid);
            }
        }
        pos += j;
    }
    if (lane == 0) {
        WpPart &s = ((WpPart *)u.cumul)[p];
        s.start = start; s.exit = err ? 0xFFFFFFFFu : pos; s.nrec = nrec; s.out_total = out; s.err = err; s.first = 0xFFFFFFFFu; s.nextra = 0;
    }
}
// One wave per unit: the true walk from part to part.  In a part it first takes the true headers the part's own walk has not got
// (that walk started at a token that need not be a header; it joins the true one after a few of them) -- read from a 64-token
// window, noted as "extras" with their absolute symbol positions -- until it stands on a position the part has a record of.
__global__ void __launch_bounds__(64) k_rle_walk_fix(MicUnit *units) {
    MicUnit &u = units[blockIdx.x];
    if (u.status != MICD_OK || u.walk_mode != 1 || u.walk_ok != 3) return;
    const uint32_t ntok = u.ntok, lane = threadIdx.x;
    const uint32_t L = wp_part_len(max(ntok, 1u));
    const uint16_t *tok = u.tok;
    bool ok = ntok >= 3 && wp_fits(u, L) && mic_len16(tok[0]) != 0;
    const uint32_t cap = ok ? ((uint32_t)tok[1] << 16) + tok[2] : 0u;
    const uint32_t mid = ok ? (1u << (mic_len16(tok[0]) - 1)) - 1 : 0u;
    if (cap > u.sym_cap) ok = false;
    WpPart *S = (WpPart *)u.cumul;
    uint32_t e = 3, N = 0, O = 0;                                           // the next true header, records and symbols in front of it
    uint32_t wbase = 0xFFFFFFFFu, wtok = 0;                                 // token window [wbase, wbase + 64)
    while (ok && e < ntok && O < cap) {
        const uint32_t q = e / L, hi = min((q + 1) * L, ntok);
        const WpPart s = S[q];
        uint2 *rec = u.seg + u.seg_cap / 2 + (size_t)q * wp_stride(L);
        uint2 *extra = rec + (L / 2 + 1);
        uint32_t blk = 0, nx = 0, first = s.nrec;
        const uint32_t N0 = N;
        uint2 r = (lane < s.nrec) ? rec[lane] : make_uint2(0xFFFFFFFFu, 0u);
        bool synced = false;
        uint32_t r0 = 0;
        while (e < hi && O < cap) {
            // first record of the part at or behind e
            uint64_t m;
            while (!(m = __ballot((r.x & 0x7FFFFFFFu) >= e + 1)) && (blk + 1) * 64 < s.nrec) {
                blk++;
                r = (blk * 64 + lane < s.nrec) ? rec[blk * 64 + lane] : make_uint2(0xFFFFFFFFu, 0u);
            }
            if (m) {
                const int i = (int)__builtin_ctzll(m);
                const uint32_t px = (uint32_t)__builtin_amdgcn_readlane((int)r.x, i) & 0x7FFFFFFFu;
                if (px == e + 1 && blk * 64 + (uint32_t)i < s.nrec) {       // the part's own walk stands here too: the rest of it is true
                    first = blk * 64 + (uint32_t)i;
                    r0 = (uint32_t)__builtin_amdgcn_readlane((int)r.y, i);
                    synced = true;
                    break;
                }
            }
            // a true header the part has no record of
            if (e - wbase >= 64u) { wbase = e; wtok = (e + lane < ntok) ? tok[e + lane] : 0u; }
            const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)wtok, (int)(e - wbase));
            if (h == 0 || nx >= WP_EXTRA) { ok = false; break; }                // (a zero count: corrupt; too many: the one-group kernels take the frame)
            const bool run = h <= mid;
            if (run && e + 1 >= ntok) { ok = false; break; }
            const uint32_t len = run ? h : h - mid;
            if (lane == 0) extra[nx] = make_uint2((e + 1) | (run ? 0x80000000u : 0u), O);
            nx++; O += len; e += run ? 2u : 1u + len;
        }
        if (!ok) break;
        if (lane == 0) {
            extra[nx] = make_uint2(0u, O);                                  // end mark: the symbol position behind the last extra
            S[q].first = first; S[q].nextra = nx; S[q].extra_base = N0; S[q].base_seg = N0 + nx; S[q].out_delta = O - r0;
        }
        N = N0 + nx;
        if (!synced) continue;                                              // the true walk left the part on its own (or the stream has its symbols)
        N += s.nrec - first; O += s.out_total - r0;
        if (s.err) { if (O < cap) ok = false; break; }                      // the zero header is reached before the stream has its symbols
        e = s.exit;
    }
    if (ok && O < cap) ok = false;                                          // tokens ran out (Go: index panic)
    if (lane == 0) {
        if (ok) { u.nseg = N; u.nsym = cap; u.walk_ok = 1; }
        else u.walk_ok = 0;
    }
}
__global__ void __launch_bounds__(256) k_rle_walk_compact(MicUnit *units) {
    MicUnit &u = units[blockIdx.y];
    if (u.status != MICD_OK || u.walk_mode != 1 || u.walk_ok != 1) return;
    const uint32_t L = wp_part_len(u.ntok), q = blockIdx.x;
    if (q * L >= u.ntok) return;
    const WpPart s = ((const WpPart *)u.cumul)[q];
    if (s.first == 0xFFFFFFFFu) return;                                     // the true walk jumps over this part (or ends before it)
    const uint2 *rec = u.seg + u.seg_cap / 2 + (size_t)q * wp_stride(L);
    const uint2 *extra = rec + (L / 2 + 1);
    uint32_t *tidx = u.flags;
    const uint32_t nsym = u.nsym;
    auto put = [&](uint32_t idx, uint32_t x, uint32_t o, uint32_t len) {
        u.seg[idx] = make_uint2(x, o);
        if (((o + len - 1) >> 13) != ((o - 1) >> 13))
            for (uint32_t kk = (o + (WS_T - 1)) >> 13; (kk << 13) < o + len && (kk << 13) < nsym; kk++) tidx[kk] = idx;
    };
    for (uint32_t r = threadIdx.x; r < s.nextra; r += 256) {                // (absolute symbol positions; the end mark carries the position behind the last)
        const uint2 v = extra[r];
        put(s.extra_base + r, v.x, v.y, extra[r + 1].y - v.y);
    }
    for (uint32_t r = s.first + threadIdx.x; r < s.nrec; r += 256) {
        const uint2 v = rec[r];
        const uint32_t nxt = (r + 1 < s.nrec) ? rec[r + 1].y : s.out_total;
        put(s.base_seg + (r - s.first), v.x, v.y + s.out_delta, nxt - v.y);
    }
}

// RLE expansion + zigzag decode + subband scatter in one pass, output-driven: a group owns WS_T consecutive scan positions, a
// thread eight; the RLE segments k_dec_translate's walker left ({payload token | run flag, first symbol}, and in `flags` the
// segment that holds every WS_T-th symbol) say where a position's symbol lies in the token stream (the fetch is
// k_dec_pixels_wg's: segment table in LDS, gallop + bisection, one 16-byte load inside a literal chunk).  A frame with an escape
// word (65535) in its symbols, fewer symbols than samples, or a stream the walker refused goes to k_wv_expand + k_wv_coeffs.
__global__ void __launch_bounds__(1024) k_wv_scatter(MicUnit *units, int32_t *a, WvDims d) {
    MicUnit &u = units[blockIdx.y];
    if (u.status != MICD_OK) return;
    const uint32_t n = (uint32_t)d.rows * (uint32_t)d.cols, tid = threadIdx.x;
    if (u.walk_ok != 1 || u.nsym < n) { if (tid == 0) u.wv_slow = 1; return; }
    __shared__ uint32_t s_segx[1024], s_segy[1024 + 1];
    a += (size_t)blockIdx.y * n;
    const uint32_t nseg = u.nseg, ntok = u.ntok;
    const uint16_t *tok = u.tok;
    const uint2 *seg = u.seg;
    const uint32_t o0 = blockIdx.x * WS_T, tile_end = min(o0 + WS_T, n);
    const uint32_t i0 = o0 + tid * 8, wend = min(i0 + 8, tile_end);
    uint32_t w8[4] = { 0u, 0u, 0u, 0u }, bad = 0;
    bool got = i0 >= tile_end;
    for (uint32_t r0 = u.flags[blockIdx.x];; r0 += 1024 - 8) {            // table rounds overlap by 8 segments (8 symbols span at most 8)
        __syncthreads();
        const uint32_t si = r0 + tid;
        uint2 sg = make_uint2(0u, 0xFFFFFFFFu);
        if (si < nseg) sg = seg[si];
        s_segx[tid] = sg.x; s_segy[tid] = sg.y;
        if (tid == 0) { const uint32_t sj = r0 + 1024; s_segy[1024] = (sj < nseg) ? seg[sj].y : 0xFFFFFFFFu; }
        __syncthreads();
        const uint32_t cover_end = s_segy[1024];
        if (!got && s_segy[0] <= i0 && wend <= cover_end) {
            uint32_t j = 0, stp = 16;
            while (j + stp <= 1024 && s_segy[j + stp] <= i0) { j += stp; stp <<= 1; }
            for (stp >>= 1; stp; stp >>= 1) if (j + stp <= 1024 && s_segy[j + stp] <= i0) j += stp;
            uint32_t start = s_segy[j], endj = s_segy[j + 1], sx = s_segx[j];
            if (wend - i0 == 8 && i0 + 8 <= endj) {
                const uint32_t xs = sx & 0x7FFFFFFFu;
                if (sx >> 31) { const uint32_t v = (xs < ntok) ? tok[xs] : 0u; w8[0] = w8[1] = w8[2] = w8[3] = v | (v << 16); }
                else {
                    const uint32_t src = xs + (i0 - start);
                    if (src + 8 <= ntok) { const wv_v4 v = *(const WvQ2 *)(tok + src); w8[0] = v.x; w8[1] = v.y; w8[2] = v.z; w8[3] = v.w; }
                    else bad = 1;                                           // literal chunk past the end (Go: index panic)
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t i = i0 + (uint32_t)k;
                    if (i < wend) {
                        while (i >= endj) { j++; start = s_segy[j]; endj = s_segy[j + 1]; sx = s_segx[j]; }
                        const uint32_t xs = sx & 0x7FFFFFFFu;
                        const uint32_t src = (sx >> 31) ? xs : xs + (i - start);
                        if (src < ntok) w8[k >> 1] |= (uint32_t)tok[src] << (16 * (k & 1)); else bad = 1;
                    }
                }
            }
            got = true;
        }
        if (!(cover_end < tile_end)) break;
    }
    if (!got) bad = 1;
    uint32_t idx = 0, left = 0;
    int32_t v8[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t x = (w8[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        if (x == 65535u && i0 + (uint32_t)k < wend) bad = 1;               // an escape marker: positions and symbols part ways
        v8[k] = (int32_t)((x >> 1) ^ (uint32_t)(-(int32_t)(x & 1)));       // zigzagDecode16, :543-546
    }
    if (bad) { u.wv_slow = 1; return; }
    if (i0 >= wend) return;
    wv_pos_row(d, i0, idx, left);
    if (wend - i0 == 8 && left >= 8) {
        wv_v4 lo, hi; lo.x = (uint32_t)v8[0]; lo.y = (uint32_t)v8[1]; lo.z = (uint32_t)v8[2]; lo.w = (uint32_t)v8[3];
        hi.x = (uint32_t)v8[4]; hi.y = (uint32_t)v8[5]; hi.z = (uint32_t)v8[6]; hi.w = (uint32_t)v8[7];
        *(WvQ4 *)(a + idx) = lo; *(WvQ4 *)(a + idx + 4) = hi;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (i0 + (uint32_t)k < wend) {
                if (left == 0) wv_pos_row(d, i0 + (uint32_t)k, idx, left);
                a[idx] = v8[k]; idx++; left--;
            }
        }
    }
}

int grid_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 4096); }

}  // namespace

void mic_launch_rle_expand(MicUnit *d_units, int n, hipStream_t stream, int mode_filter) {
    hipLaunchKernelGGL(k_wv_expand, dim3((unsigned)n), dim3(WV_THREADS), 0, stream, d_units, mode_filter, 0);
}

namespace {

// nf frames of rows x cols, contiguous on the device in s->io_px: forward transform, symbols, RLE + 4-state FSE in one batch.
// blobs[i] receives frame i's FSE stream (without the 11-byte header), st[i] its status.
// d_src: the frames on the device.  blobs == nullptr: the streams stay on the device (session_encode_finish's packed buffer:
// *d_blobs_out / offs_out), nothing is copied to the host.
int wv_compress_frames(mic_hip_session *s, const uint16_t *d_src, int nf, int rows, int cols, int applied,
                       std::vector<std::vector<uint8_t>> *blobs, std::vector<int32_t> &st,
                       const uint8_t **d_blobs_out = nullptr, uint64_t *offs_out = nullptr) {
    const size_t n = (size_t)rows * (size_t)cols;
    int rc;
    if ((rc = s->ensure(nf, 2 * n + 16))) return rc;                     // room for 3-word escapes
    DevBuf &a = s->wv_a, &b = s->wv_b;                                   // coefficient planes: kept by the session (no hipMalloc per call)
    // b: the smooth planes between levels, two per frame (levels alternate), each (rows + 1) / 2 x (cols + 1) / 2 at most
    const size_t ll_half = (size_t)((rows + 1) / 2) * (size_t)((cols + 1) / 2), ll_fs = 2 * ll_half;
    if ((rc = a.reserve(n * 4 * (size_t)nf + 64)) || (rc = b.reserve(ll_fs * 4 * (size_t)nf + 64))) return rc;
    auto done = [&](int code) { return code; };
    int32_t *A = (int32_t *)a.p, *B = (int32_t *)b.p;
    s->timer.reset(s->stream);
    s->timer.mark("k_wv_fwd2d");
    if (applied == 0) hipLaunchKernelGGL(k_wv_load, dim3(grid_for(n * (size_t)nf)), dim3(256), 0, s->stream, d_src, A, n * (size_t)nf);
    { int r = rows, c = cols;
      for (int l = 0; l < applied; l++) {                                  // a level per launch: level l's smooth plane in B, parity l
          const int rn = (r + 1) / 2, cn = (c + 1) / 2;
          const dim3 g((unsigned)((cn + 4 * WS_LANES - 1) / (4 * WS_LANES)), (unsigned)((rn + WS_ROWS - 1) / WS_ROWS), (unsigned)std::min(nf, 65535));
          const bool last = l == applied - 1;
          int32_t *lld = last ? A : B + ((l & 1) ? ll_half : 0);
          const int lls = last ? cols : cn; const size_t llf = last ? n : ll_fs;
          if (l == 0) hipLaunchKernelGGL(k_wv_fwd2d<true>, g, dim3(256), 0, s->stream, (const void *)d_src, cols, n, A, cols, n, lld, lls, llf, r, c, nf);
          else hipLaunchKernelGGL(k_wv_fwd2d<false>, g, dim3(256), 0, s->stream, (const void *)(B + (((l - 1) & 1) ? ll_half : 0)), c, ll_fs, A, cols, n, lld, lls, llf, r, c, nf);
          r = rn; c = cn;
      } }
    { const int arc = s->h_units.assign((size_t)nf, MicUnit{}); if (arc) return arc; }
    for (int i = 0; i < nf; i++) {
        MicUnit &u = s->h_units[(size_t)i];
        u.w = 1; u.h = 1; u.nstates = 4; u.mode = 2; u.no_fallback = 1; // FSECompressU16FourState, no fallback (:344)
        s->fill_workspace(u, i);
    }
    if (s->h_units.upload(s->units.p, (size_t)nf, s->stream) != MIC_OK) return done(MIC_ERR_DEVICE);
    if ((rc = s->prepare_hist(nf))) return done(rc);
    s->timer.mark("k_wv_symbols");
    {
        const WvDims d = wv_dims(rows, cols, applied);
        hipLaunchKernelGGL(k_wv_symbols_par, dim3((unsigned)((n + WS_T - 1) / WS_T), (unsigned)std::min(nf, 65535)), dim3(1024), 0, s->stream,
                           (MicUnit *)s->units.p, (const int32_t *)A, d, nf);
        hipLaunchKernelGGL(k_wv_symbols_fin, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s->stream, (MicUnit *)s->units.p, d, nf);
        hipLaunchKernelGGL(k_wv_symbols, dim3((unsigned)nf), dim3(WV_THREADS), 0, s->stream, (MicUnit *)s->units.p, (const int32_t *)A, d);   // frames with wide coefficients
    }
    mic_launch_encode((MicUnit *)s->units.p, nf, s->stream, s->variant, &s->timer);
    if (hipGetLastError() != hipSuccess) { s->hist_unknown(); return done(MIC_ERR_DEVICE); }
    s->begin_chain(nf);
    std::vector<uint64_t> offs((size_t)nf + 1); std::vector<int32_t> ns((size_t)nf); const uint8_t *d_blobs = nullptr;
    st.assign((size_t)nf, 0);
    if ((rc = session_encode_finish(s, &d_blobs, offs.data(), st.data(), ns.data()))) return done(rc);
    if (!blobs) {
        if (d_blobs_out) *d_blobs_out = d_blobs;
        if (offs_out) memcpy(offs_out, offs.data(), sizeof(uint64_t) * ((size_t)nf + 1));
        return done(MIC_OK);
    }
    std::vector<uint8_t> host((size_t)offs[(size_t)nf] + 16);
    if (offs[(size_t)nf] && hipMemcpy(host.data(), d_blobs, (size_t)offs[(size_t)nf], hipMemcpyDeviceToHost) != hipSuccess) return done(MIC_ERR_DEVICE);
    blobs->assign((size_t)nf, std::vector<uint8_t>());
    for (int i = 0; i < nf; i++) if (st[(size_t)i] == MIC_OK) (*blobs)[(size_t)i].assign(host.begin() + (long)offs[(size_t)i], host.begin() + (long)offs[(size_t)i + 1]);
    return done(MIC_OK);
}

// nf FSE streams (already in s->io_comp at offs[i] .. offs[i + 1]) of frames of one shape -> pixels in s->io_px
int wv_decompress_frames(mic_hip_session *s, const uint8_t *d_comp, uint16_t *d_dst, int nf, const uint64_t *offs, int rows, int cols, int levels,
                         std::vector<int32_t> &st) {
    const size_t n = (size_t)rows * (size_t)cols;
    int rc;
    // everything the launches below assume is checked before the first of them: grid limits, stream ranges
    if (nf <= 0 || nf > 65535 || n >= ((size_t)65535 * WS_T)) return MIC_ERR_UNSUPPORTED;
    for (int i = 0; i < nf; i++)
        if (offs[(size_t)i + 1] < offs[(size_t)i] || offs[(size_t)i + 1] - offs[(size_t)i] > 0xFFFFFFF0ull) return MIC_ERR_ARGS;
    if ((rc = s->ensure(nf, 2 * n + 16))) return rc;
    DevBuf &a = s->wv_a, &b = s->wv_b;
    const size_t ll_half = (size_t)((rows + 1) / 2) * (size_t)((cols + 1) / 2), ll_fs = 2 * ll_half;   // (the smooth planes between levels, as in wv_compress_frames)
    if ((rc = a.reserve(n * 4 * (size_t)nf + 64)) || (rc = b.reserve(ll_fs * 4 * (size_t)nf + 64))) return rc;
    auto done = [&](int code) { return code; };
    { const int arc = s->h_units.assign((size_t)nf, MicUnit{}); if (arc) return arc; }
    for (int i = 0; i < nf; i++) {
        MicUnit &u = s->h_units[(size_t)i];
        u.comp_in = d_comp + offs[(size_t)i]; u.comp_len = (uint32_t)(offs[(size_t)i + 1] - offs[(size_t)i]); u.w = 1; u.h = 1; u.mode = 1; u.walk_mode = 1;
        s->fill_workspace(u, i);
    }
    if (s->h_units.upload(s->units.p, (size_t)nf, s->stream) != MIC_OK) return done(MIC_ERR_DEVICE);
    int32_t *A = (int32_t *)a.p, *B = (int32_t *)b.p;
    s->timer.reset(s->stream);
    mic_launch_decode((MicUnit *)s->units.p, nf, s->stream, s->variant, &s->timer, (int *)s->cls.p);
    const WvDims d = wv_dims(rows, cols, levels);
    if (s->timer.used) { s->timer.used--; s->timer.names.pop_back(); }   // (drop the chain's "end" mark: the wavelet kernels follow)
    s->timer.mark("k_rle_walk_parts+fix+compact");
    hipLaunchKernelGGL(k_rle_walk_parts, dim3(WP_PARTS, (unsigned)nf), dim3(64), 0, s->stream, (MicUnit *)s->units.p);
    hipLaunchKernelGGL(k_rle_walk_fix, dim3((unsigned)nf), dim3(64), 0, s->stream, (MicUnit *)s->units.p);
    hipLaunchKernelGGL(k_rle_walk_compact, dim3(WP_PARTS, (unsigned)nf), dim3(256), 0, s->stream, (MicUnit *)s->units.p);
    s->timer.mark("k_wv_scatter");
    hipLaunchKernelGGL(k_wv_scatter, dim3((unsigned)((n + WS_T - 1) / WS_T), (unsigned)nf), dim3(1024), 0, s->stream, (MicUnit *)s->units.p, A, d);
    s->timer.mark("k_wv_expand+coeffs (escape frames)");
    hipLaunchKernelGGL(k_wv_expand, dim3((unsigned)nf), dim3(WV_THREADS), 0, s->stream, (MicUnit *)s->units.p, -1, 1);
    hipLaunchKernelGGL(k_wv_coeffs, dim3((unsigned)nf), dim3(WV_THREADS), 0, s->stream, (MicUnit *)s->units.p, A, d);
    s->timer.mark("k_wv_inv2d");
    if (levels == 0) hipLaunchKernelGGL(k_wv_store, dim3(grid_for(n * (size_t)nf)), dim3(256), 0, s->stream, (const int32_t *)A, d_dst, n * (size_t)nf);
    for (int l = levels - 1; l >= 0; l--) {                                                 // coarse -> fine, :519-527
        const int r = d.nr[l], cc = d.nc[l];
        const dim3 g((unsigned)(((cc + 1) / 2 + 4 * WS_LANES - 1) / (4 * WS_LANES)), (unsigned)(((r + 1) / 2 + WS_ROWS - 1) / WS_ROWS), (unsigned)std::min(nf, 65535));
        const bool coarsest = l == levels - 1;
        const int32_t *lls = coarsest ? A : B + (((l + 1) & 1) ? ll_half : 0);
        const int llst = coarsest ? cols : d.nc[l + 1]; const size_t llf = coarsest ? n : ll_fs;
        if (l == 0) hipLaunchKernelGGL(k_wv_inv2d<true>, g, dim3(256), 0, s->stream, (const int32_t *)A, cols, n, lls, llst, llf, (void *)d_dst, cols, n, r, cc, nf);
        else hipLaunchKernelGGL(k_wv_inv2d<false>, g, dim3(256), 0, s->stream, (const int32_t *)A, cols, n, lls, llst, llf, (void *)(B + ((l & 1) ? ll_half : 0)), cc, ll_fs, r, cc, nf);
    }
    s->timer.mark("end");
    if (hipGetLastError() != hipSuccess) return done(MIC_ERR_DEVICE);
    s->begin_chain(nf);
    st.assign((size_t)nf, 0);
    if ((rc = session_decode_finish(s, st.data()))) return done(rc);
    return done(MIC_OK);
}

void wv_put_header(uint8_t *out, int rows, int cols, uint16_t max_value, int applied) {     // :346-350
    out[0] = (uint8_t)rows; out[1] = (uint8_t)(rows >> 8); out[2] = (uint8_t)(rows >> 16); out[3] = (uint8_t)((uint32_t)rows >> 24);
    out[4] = (uint8_t)cols; out[5] = (uint8_t)(cols >> 8); out[6] = (uint8_t)(cols >> 16); out[7] = (uint8_t)((uint32_t)cols >> 24);
    out[8] = (uint8_t)max_value; out[9] = (uint8_t)(max_value >> 8);
    out[10] = (uint8_t)applied;
}

size_t wv_frames_per_batch(size_t n) { return std::max<size_t>(1, kWorkspaceBudget / (unit_ws_bytes(2 * n + 16) + 8 * n)); }

}  // namespace

extern "C" {

// WaveletV2RLEFSECompressU16 over nframes frames of one shape in one launch chain (the reference codes one image per call; a
// WaveletV2 file is ONE serial FSE stream, so the GPU only pays off when many frames are coded side by side).
// frames: nframes x rows*cols u16, contiguous.  Frame i's file goes to out + i * out_stride, its length to out_lens[i], its
// status to status[i] (a frame that fails does not stop the others).
int mic_hip_wavelet_v2_compress_batch(const uint16_t *frames, int nframes, int rows, int cols, uint16_t max_value, int levels,
                                      uint8_t *out, size_t out_stride, size_t *out_lens, int32_t *status) try {
    if (!frames || !out || !out_lens || !status || nframes <= 0 || rows <= 0 || cols <= 0) return MIC_ERR_ARGS;
    const size_t n = (size_t)rows * (size_t)cols;
    if (n > ((size_t)1 << 27)) return MIC_ERR_UNSUPPORTED;
    if (out_stride < 11) return MIC_ERR_CAPACITY;
    if (levels < 1) levels = 1;
    if (levels > 8) levels = 8;
    int applied = 0;
    { int r = rows, c = cols; for (; applied < levels; applied++) { if (r < 2 || c < 2) break; r = (r + 1) / 2; c = (c + 1) / 2; } }   // :321-330
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    mic_hip_session *s = cur_default();
    const size_t per = wv_frames_per_batch(n);
    for (size_t f0 = 0; f0 < (size_t)nframes; f0 += per) {
        const int nf = (int)std::min(per, (size_t)nframes - f0);
        if ((rc = s->ensure(nf, 2 * n + 16))) return rc;
        if ((rc = s->io_px.reserve(n * 2 * (size_t)nf + 64))) return rc;
        HIP_TRY(hipMemcpyAsync(s->io_px.p, frames + f0 * n, n * 2 * (size_t)nf, hipMemcpyHostToDevice, s->stream));
        std::vector<std::vector<uint8_t>> blobs; std::vector<int32_t> st;
        if ((rc = wv_compress_frames(s, (const uint16_t *)s->io_px.p, nf, rows, cols, applied, &blobs, st))) return rc;
        for (int i = 0; i < nf; i++) {
            uint8_t *o = out + (f0 + (size_t)i) * out_stride;
            status[f0 + (size_t)i] = st[(size_t)i]; out_lens[f0 + (size_t)i] = 0;
            if (st[(size_t)i] != MIC_OK) continue;
            if (11 + blobs[(size_t)i].size() > out_stride) { status[f0 + (size_t)i] = MIC_ERR_CAPACITY; continue; }
            wv_put_header(o, rows, cols, max_value, applied);
            memcpy(o + 11, blobs[(size_t)i].data(), blobs[(size_t)i].size());
            out_lens[f0 + (size_t)i] = 11 + blobs[(size_t)i].size();
        }
    }
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// WaveletV2RLEFSECompressU16 / WaveletV2SIMDRLEFSECompressU16 (waveletfsecompressu16.go:303, :374)
int mic_hip_wavelet_v2_compress(const uint16_t *pixels, int rows, int cols, uint16_t max_value, int levels,
                                uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!pixels || !out || !out_len || rows <= 0 || cols <= 0) return MIC_ERR_ARGS;
    if (out_cap < 11) return MIC_ERR_CAPACITY;
    size_t len = 0; int32_t st = 0;
    const int rc = mic_hip_wavelet_v2_compress_batch(pixels, 1, rows, cols, max_value, levels, out, out_cap, &len, &st);
    if (rc) return rc;
    if (st != MIC_OK) return st;
    *out_len = len;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_wavelet_v2_info(const uint8_t *c, size_t len, int *rows, int *cols, int *max_value, int *levels) try {
    if (!c) return MIC_ERR_ARGS;
    if (len < 11) return MIC_ERR_CORRUPT;                                                   // :494-496
    if (rows) *rows = (int)((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24));
    if (cols) *cols = (int)((uint32_t)c[4] | ((uint32_t)c[5] << 8) | ((uint32_t)c[6] << 16) | ((uint32_t)c[7] << 24));
    if (max_value) *max_value = c[8] | (c[9] << 8);
    if (levels) *levels = c[10];
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// WaveletV2RLEFSEDecompressU16 over nframes files of ONE shape (same rows, cols, levels) in one launch chain; pixels_out receives
// nframes x rows*cols u16, status[i] frame i's status.  Files of another shape than the first: MIC_ERR_ARGS for that frame.
int mic_hip_wavelet_v2_decompress_batch(const uint8_t *const *files, const size_t *lens, int nframes, uint16_t *pixels_out, size_t out_cap_px,
                                        int32_t *status) try {
    if (!files || !lens || !pixels_out || !status || nframes <= 0) return MIC_ERR_ARGS;
    int rows, cols, maxv, levels;
    int rc = mic_hip_wavelet_v2_info(files[0], lens[0], &rows, &cols, &maxv, &levels);
    if (rc) return rc;
    if (rows <= 0 || cols <= 0 || levels > 8) return MIC_ERR_CORRUPT;
    const size_t n = (size_t)rows * (size_t)cols;
    if (n > ((size_t)1 << 27)) return MIC_ERR_UNSUPPORTED;
    if (n * (size_t)nframes > out_cap_px) return MIC_ERR_CAPACITY;
    DefaultLease lease;
    if ((rc = lease.acquire())) return rc;
    mic_hip_session *s = cur_default();
    const size_t per = wv_frames_per_batch(n);
    for (size_t f0 = 0; f0 < (size_t)nframes; f0 += per) {
        const int nf = (int)std::min(per, (size_t)nframes - f0);
        std::vector<int> slot((size_t)nf, -1); std::vector<uint64_t> offs(1, 0); int good = 0;   // streams packed back to back
        for (int i = 0; i < nf; i++) {
            const uint8_t *c = files[f0 + (size_t)i]; const size_t len = lens[f0 + (size_t)i];
            int r2, c2, m2, l2;
            status[f0 + (size_t)i] = MIC_OK;
            if (!c) { status[f0 + (size_t)i] = MIC_ERR_ARGS; continue; }
            if (mic_hip_wavelet_v2_info(c, len, &r2, &c2, &m2, &l2) != MIC_OK) { status[f0 + (size_t)i] = MIC_ERR_CORRUPT; continue; }
            if (r2 != rows || c2 != cols || l2 != levels) { status[f0 + (size_t)i] = MIC_ERR_ARGS; continue; }
            if (len < 13 || c[11] != 0xFF || c[12] != 0x04 || len - 11 > 0xFFFFFFF0ull) { status[f0 + (size_t)i] = MIC_ERR_CORRUPT; continue; }   // FSEDecompressU16FourState only, :503
            slot[(size_t)i] = good++; offs.push_back(offs.back() + (len - 11));
        }
        if (!good) continue;
        if ((rc = s->ensure(good, 2 * n + 16))) return rc;
        if ((rc = s->io_comp.reserve((size_t)offs.back() + 64)) || (rc = s->io_px.reserve(n * 2 * (size_t)good + 64))) return rc;
        for (int i = 0; i < nf; i++) if (slot[(size_t)i] >= 0)
            HIP_TRY(hipMemcpyAsync((uint8_t *)s->io_comp.p + offs[(size_t)slot[(size_t)i]], files[f0 + (size_t)i] + 11, lens[f0 + (size_t)i] - 11,
                                   hipMemcpyHostToDevice, s->stream));
        std::vector<int32_t> st;
        if ((rc = wv_decompress_frames(s, (const uint8_t *)s->io_comp.p, (uint16_t *)s->io_px.p, good, offs.data(), rows, cols, levels, st))) return rc;
        for (int i = 0; i < nf; i++) if (slot[(size_t)i] >= 0) {
            const size_t k = (size_t)slot[(size_t)i];
            status[f0 + (size_t)i] = st[k];
            if (st[k] == MIC_OK)
                HIP_TRY(hipMemcpyAsync(pixels_out + (f0 + (size_t)i) * n, (uint16_t *)s->io_px.p + k * n, n * 2, hipMemcpyDeviceToHost, s->stream));
        }
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// WaveletV2RLEFSEDecompressU16 / WaveletV2SIMDRLEFSEDecompressU16 (:380-425, :493-534)
int mic_hip_wavelet_v2_decompress(const uint8_t *c, size_t len, uint16_t *pixels_out, size_t out_cap_px) try {
    if (!c || !pixels_out) return MIC_ERR_ARGS;
    int32_t st = 0;
    const int rc = mic_hip_wavelet_v2_decompress_batch(&c, &len, 1, pixels_out, out_cap_px, &st);
    return rc ? rc : st;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// ---- device-resident forms (what bench.py times for BASELINE config 3): frames, streams and pixels stay in HBM ----------------
// nframes frames of rows x cols u16, contiguous at d_frames -> their WaveletV2 streams WITHOUT the 11-byte file header (rows, cols,
// maxValue, levels: the caller has them), packed back to back on the device: *d_streams, h_offsets[nframes + 1], h_status[nframes];
// *levels_applied = the level count the header would carry (waveletfsecompressu16.go:321-330).
int mic_hip_session_wavelet_v2_encode(mic_hip_session *s, const uint16_t *d_frames, int nframes, int rows, int cols, int levels,
                                      const uint8_t **d_streams, uint64_t *h_offsets, int32_t *h_status, int *levels_applied) try {
    if (!s || !d_frames || !d_streams || !h_offsets || !h_status || nframes <= 0 || rows <= 0 || cols <= 0) return MIC_ERR_ARGS;
    const size_t n = (size_t)rows * (size_t)cols;
    if (n > ((size_t)1 << 27)) return MIC_ERR_UNSUPPORTED;
    if (levels < 1) levels = 1;
    if (levels > 8) levels = 8;
    int applied = 0;
    { int r = rows, c = cols; for (; applied < levels; applied++) { if (r < 2 || c < 2) break; r = (r + 1) / 2; c = (c + 1) / 2; } }
    if (levels_applied) *levels_applied = applied;
    int rc = s->activate();
    if (rc) return rc;
    std::vector<int32_t> st;
    if ((rc = wv_compress_frames(s, d_frames, nframes, rows, cols, applied, nullptr, st, d_streams, h_offsets))) return rc;
    for (int i = 0; i < nframes; i++) h_status[i] = st[(size_t)i];
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
// The inverse: nframes header-less streams at d_streams + h_offsets[i] (all of one shape / level count) -> pixels at d_pixels_out.
int mic_hip_session_wavelet_v2_decode(mic_hip_session *s, const uint8_t *d_streams, const uint64_t *h_offsets, int nframes,
                                      int rows, int cols, int levels, uint16_t *d_pixels_out, int32_t *h_status) try {
    if (!s || !d_streams || !h_offsets || !d_pixels_out || !h_status || nframes <= 0 || rows <= 0 || cols <= 0 || levels < 0 || levels > 8) return MIC_ERR_ARGS;
    if ((size_t)rows * (size_t)cols > ((size_t)1 << 27)) return MIC_ERR_UNSUPPORTED;
    int rc = s->activate();
    if (rc) return rc;
    std::vector<int32_t> st;
    if ((rc = wv_decompress_frames(s, d_streams, d_pixels_out, nframes, h_offsets, rows, cols, levels, st))) return rc;
    for (int i = 0; i < nframes; i++) h_status[i] = st[(size_t)i];
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

}  // extern "C"
