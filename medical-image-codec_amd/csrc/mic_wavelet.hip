// mic_wavelet.hip -- WaveletV2 on the GPU (waveletu16.go, waveletfsecompressu16.go:303-534).
//
// WaveletV2{,SIMD}RLEFSECompressU16 = up to 8 levels of the Le Gall 5/3 integer lifting in Mallat
// layout -> subband scan -> zigzag (+ 3-word escape) -> RLE with length prefix -> 4-state FSE (no
// fallback) behind an 11-byte header.  The reference lifts in place, predict pass then update pass,
// rows then columns, AVX2 over 8-column blocks (wavelet_simd_amd64.s).  Here every output sample is
// written straight from the <= 5 input samples it depends on,
//     d[i] = x[2i+1] - ((x[2i] + x[2i+2]) >> 1)        s[i] = x[2i] + ((d[i-1] + d[i] + 2) >> 2)
// with the reference's boundary rules, so a pass is one out-of-place, fully parallel kernel (row pass
// A -> B, column pass B -> A, both de-interleaving on the fly) and there is no serial dependence at all.
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

__global__ void __launch_bounds__(256) k_wv_load(const uint16_t *px, int32_t *a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (int32_t)px[i];
}
__global__ void __launch_bounds__(256) k_wv_store(const int32_t *a, uint16_t *px, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) px[i] = (uint16_t)a[i];
}
// rows of the r x c region: src -> dst, de-interleaved [low | high]   (waveletu16.go:170-182)
__global__ void __launch_bounds__(256) k_wv_fwd_rows(const int32_t *src, int32_t *dst, int r, int c, int stride) {
    const size_t n = (size_t)r * c; const int n_low = (c + 1) / 2;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(t / c), k = (int)(t % c);
        const int32_t *row = src + (size_t)y * stride;
        auto x = [&](int j) { return row[j]; };
        dst[(size_t)y * stride + k] = (k < n_low) ? wv_s(x, c, k) : wv_d(x, c, k - n_low);
    }
}
// columns of the r x c region   (waveletu16.go:183-208)
__global__ void __launch_bounds__(256) k_wv_fwd_cols(const int32_t *src, int32_t *dst, int r, int c, int stride) {
    const size_t n = (size_t)r * c; const int n_low = (r + 1) / 2;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(t / c), xcol = (int)(t % c);
        auto x = [&](int j) { return src[(size_t)j * stride + xcol]; };
        dst[(size_t)k * stride + xcol] = (k < n_low) ? wv_s(x, r, k) : wv_d(x, r, k - n_low);
    }
}
// inverse: columns first, then rows   (waveletu16.go:213-257)
__global__ void __launch_bounds__(256) k_wv_inv_cols(const int32_t *src, int32_t *dst, int r, int c, int stride) {
    const size_t n = (size_t)r * c;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(t / c), xcol = (int)(t % c);
        auto cf = [&](int j) { return src[(size_t)j * stride + xcol]; };
        dst[(size_t)k * stride + xcol] = wv_sample(cf, r, k);
    }
}
__global__ void __launch_bounds__(256) k_wv_inv_rows(const int32_t *src, int32_t *dst, int r, int c, int stride) {
    const size_t n = (size_t)r * c;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(t / c), k = (int)(t % c);
        const int32_t *row = src + (size_t)y * stride;
        auto cf = [&](int j) { return row[j]; };
        dst[(size_t)y * stride + k] = wv_sample(cf, c, k);
    }
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
    MicUnit &u = units[0];
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
__global__ void __launch_bounds__(WV_THREADS) k_wv_expand(MicUnit *units, int mode_filter) {
    MicUnit &u = units[blockIdx.x];
    if (mode_filter >= 0 && u.mode != (uint32_t)mode_filter) return;
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
                const uint32_t h = __builtin_amdgcn_readlane(w, (int)j);
                if (h == 0 || nseg >= segcap) { err = 1; break; }
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
        const uint32_t h = tok[r.x];
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
    MicUnit &u = units[0];
    if (u.status != MICD_OK) return;
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

int grid_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 4096); }

}  // namespace

void mic_launch_rle_expand(MicUnit *d_units, int n, hipStream_t stream, int mode_filter) {
    hipLaunchKernelGGL(k_wv_expand, dim3((unsigned)n), dim3(WV_THREADS), 0, stream, d_units, mode_filter);
}

extern "C" {

// WaveletV2RLEFSECompressU16 / WaveletV2SIMDRLEFSECompressU16 (waveletfsecompressu16.go:303, :374)
int mic_hip_wavelet_v2_compress(const uint16_t *pixels, int rows, int cols, uint16_t max_value, int levels,
                                uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!pixels || !out || !out_len || rows <= 0 || cols <= 0) return MIC_ERR_ARGS;
    const size_t n = (size_t)rows * (size_t)cols;
    if (n > ((size_t)1 << 27)) return MIC_ERR_UNSUPPORTED;
    if (out_cap < 11) return MIC_ERR_CAPACITY;
    if (levels < 1) levels = 1;
    if (levels > 8) levels = 8;
    int applied = 0;
    { int r = rows, c = cols; for (; applied < levels; applied++) { if (r < 2 || c < 2) break; r = (r + 1) / 2; c = (c + 1) / 2; } }   // :321-330
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_device();
    if (rc) return rc;
    mic_hip_session *s = &g_default;
    if ((rc = s->ensure(1, 2 * n + 16))) return rc;                      // room for 3-word escapes
    if ((rc = s->io_px.reserve(n * 2 + 64))) return rc;
    DevBuf a, b;
    if ((rc = a.reserve(n * 4 + 64)) || (rc = b.reserve(n * 4 + 64))) { a.release(); b.release(); return rc; }
    auto done = [&](int code) { a.release(); b.release(); return code; };
    if (hipMemcpyAsync(s->io_px.p, pixels, n * 2, hipMemcpyHostToDevice, s->stream) != hipSuccess) return done(MIC_ERR_DEVICE);
    int32_t *A = (int32_t *)a.p, *B = (int32_t *)b.p;
    hipLaunchKernelGGL(k_wv_load, dim3(grid_for(n)), dim3(256), 0, s->stream, (const uint16_t *)s->io_px.p, A, n);
    { int r = rows, c = cols;
      for (int l = 0; l < applied; l++) {
          hipLaunchKernelGGL(k_wv_fwd_rows, dim3(grid_for((size_t)r * c)), dim3(256), 0, s->stream, (const int32_t *)A, B, r, c, cols);
          hipLaunchKernelGGL(k_wv_fwd_cols, dim3(grid_for((size_t)r * c)), dim3(256), 0, s->stream, (const int32_t *)B, A, r, c, cols);
          r = (r + 1) / 2; c = (c + 1) / 2;
      } }
    s->h_units.assign(1, MicUnit{});
    MicUnit &u = s->h_units[0];
    u.w = 1; u.h = 1; u.nstates = 4; u.mode = 2; u.no_fallback = 1;     // FSECompressU16FourState, no fallback (:344)
    s->fill_workspace(u, 0);
    if (hipMemcpyAsync(s->units.p, s->h_units.data(), sizeof(MicUnit), hipMemcpyHostToDevice, s->stream) != hipSuccess) return done(MIC_ERR_DEVICE);
    if (hipMemsetAsync(s->hist.p, 0, kSym * 4, s->stream) != hipSuccess) return done(MIC_ERR_DEVICE);
    hipLaunchKernelGGL(k_wv_symbols, dim3(1), dim3(WV_THREADS), 0, s->stream, (MicUnit *)s->units.p, (const int32_t *)A, wv_dims(rows, cols, applied));
    mic_launch_encode((MicUnit *)s->units.p, 1, s->stream, s->variant, nullptr);
    if (hipGetLastError() != hipSuccess) return done(MIC_ERR_DEVICE);
    s->n_last = 1;
    uint64_t offs[2]; int32_t st = 0, ns = 0; const uint8_t *d_blobs = nullptr;
    if ((rc = session_encode_finish(s, &d_blobs, offs, &st, &ns))) return done(rc);
    if (st != MIC_OK) return done(st);
    const size_t len = (size_t)offs[1];
    if (11 + len > out_cap) return done(MIC_ERR_CAPACITY);
    if (hipMemcpy(out + 11, d_blobs, len, hipMemcpyDeviceToHost) != hipSuccess) return done(MIC_ERR_DEVICE);
    out[0] = (uint8_t)rows; out[1] = (uint8_t)(rows >> 8); out[2] = (uint8_t)(rows >> 16); out[3] = (uint8_t)((uint32_t)rows >> 24);   // :346-350
    out[4] = (uint8_t)cols; out[5] = (uint8_t)(cols >> 8); out[6] = (uint8_t)(cols >> 16); out[7] = (uint8_t)((uint32_t)cols >> 24);
    out[8] = (uint8_t)max_value; out[9] = (uint8_t)(max_value >> 8);
    out[10] = (uint8_t)applied;
    *out_len = 11 + len;
    return done(MIC_OK);
}

int mic_hip_wavelet_v2_info(const uint8_t *c, size_t len, int *rows, int *cols, int *max_value, int *levels) {
    if (!c) return MIC_ERR_ARGS;
    if (len < 11) return MIC_ERR_CORRUPT;                                                   // :494-496
    if (rows) *rows = (int)((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24));
    if (cols) *cols = (int)((uint32_t)c[4] | ((uint32_t)c[5] << 8) | ((uint32_t)c[6] << 16) | ((uint32_t)c[7] << 24));
    if (max_value) *max_value = c[8] | (c[9] << 8);
    if (levels) *levels = c[10];
    return MIC_OK;
}

// WaveletV2RLEFSEDecompressU16 / WaveletV2SIMDRLEFSEDecompressU16 (:380-425, :493-534)
int mic_hip_wavelet_v2_decompress(const uint8_t *c, size_t len, uint16_t *pixels_out, size_t out_cap_px) {
    if (!c || !pixels_out) return MIC_ERR_ARGS;
    int rows, cols, maxv, levels;
    int rc = mic_hip_wavelet_v2_info(c, len, &rows, &cols, &maxv, &levels);
    if (rc) return rc;
    if (rows <= 0 || cols <= 0 || levels > 8) return MIC_ERR_CORRUPT;
    const size_t n = (size_t)rows * (size_t)cols;
    if (n > ((size_t)1 << 27)) return MIC_ERR_UNSUPPORTED;
    if (n > out_cap_px) return MIC_ERR_CAPACITY;
    if (len < 13 || c[11] != 0xFF || c[12] != 0x04) return MIC_ERR_CORRUPT;                 // FSEDecompressU16FourState only, :503
    std::lock_guard<std::mutex> lk(g_mu);
    if ((rc = ensure_device())) return rc;
    mic_hip_session *s = &g_default;
    if ((rc = s->ensure(1, 2 * n + 16))) return rc;
    if ((rc = s->io_comp.reserve(len + 64)) || (rc = s->io_px.reserve(n * 2 + 64))) return rc;
    DevBuf a, b;
    if ((rc = a.reserve(n * 4 + 64)) || (rc = b.reserve(n * 4 + 64))) { a.release(); b.release(); return rc; }
    auto done = [&](int code) { a.release(); b.release(); return code; };
    if (hipMemcpyAsync(s->io_comp.p, c + 11, len - 11, hipMemcpyHostToDevice, s->stream) != hipSuccess) return done(MIC_ERR_DEVICE);
    s->h_units.assign(1, MicUnit{});
    MicUnit &u = s->h_units[0];
    u.comp_in = (const uint8_t *)s->io_comp.p; u.comp_len = (uint32_t)(len - 11); u.w = 1; u.h = 1; u.mode = 1;
    s->fill_workspace(u, 0);
    if (hipMemcpyAsync(s->units.p, s->h_units.data(), sizeof(MicUnit), hipMemcpyHostToDevice, s->stream) != hipSuccess) return done(MIC_ERR_DEVICE);
    int32_t *A = (int32_t *)a.p, *B = (int32_t *)b.p;
    if (hipMemsetAsync(A, 0, n * 4, s->stream) != hipSuccess) return done(MIC_ERR_DEVICE);
    mic_launch_decode((MicUnit *)s->units.p, 1, s->stream, s->variant, nullptr);
    const WvDims d = wv_dims(rows, cols, levels);
    hipLaunchKernelGGL(k_wv_expand, dim3(1), dim3(WV_THREADS), 0, s->stream, (MicUnit *)s->units.p, -1);
    hipLaunchKernelGGL(k_wv_coeffs, dim3(1), dim3(WV_THREADS), 0, s->stream, (MicUnit *)s->units.p, A, d);
    for (int l = levels - 1; l >= 0; l--) {                                                 // coarse -> fine, :519-527
        const int r = d.nr[l], cc = d.nc[l];
        hipLaunchKernelGGL(k_wv_inv_cols, dim3(grid_for((size_t)r * cc)), dim3(256), 0, s->stream, (const int32_t *)A, B, r, cc, cols);
        hipLaunchKernelGGL(k_wv_inv_rows, dim3(grid_for((size_t)r * cc)), dim3(256), 0, s->stream, (const int32_t *)B, A, r, cc, cols);
    }
    hipLaunchKernelGGL(k_wv_store, dim3(grid_for(n)), dim3(256), 0, s->stream, (const int32_t *)A, (uint16_t *)s->io_px.p, n);
    if (hipGetLastError() != hipSuccess) return done(MIC_ERR_DEVICE);
    s->n_last = 1;
    int32_t st = 0;
    if ((rc = session_decode_finish(s, &st))) return done(rc);
    if (st != MIC_OK) return done(st);
    if (hipMemcpy(pixels_out, s->io_px.p, n * 2, hipMemcpyDeviceToHost) != hipSuccess) return done(MIC_ERR_DEVICE);
    return done(MIC_OK);
}

}  // extern "C"
