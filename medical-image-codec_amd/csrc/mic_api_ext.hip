// mic_api_ext.hip -- MIC3 / WSI on the GPU (wsicompress.go, wsiformat.go, wsipyramid.go, ycocgr.go).
// 8-bit RGB slides (three YCoCg-R planes per tile) and 8/16-bit greyscale slides (one plane per tile).
//
// CompressWSI = pyramid (2x2 box) -> zero-padded tiles -> YCoCg-R -> three planes per tile ->
// per plane: constant-zero / constant / CompressSingleFrame / raw fallback -> tile blobs -> MIC3.
// On the device: the pyramid, the tile extraction fused with the colour transform and the
// per-plane min/max (which decides the plane mode and the maxValue handed to the unit codec),
// and the unit codec itself over every non-constant plane of the slide in one batch.  The host
// writes the container around the plane blobs.  Decode mirrors it: one batch over the planes of
// the requested tiles, then inverse transform + crop on the device.
#include "mic_session.h"

static constexpr size_t kMaxGridX = 0x7FFFFFFF;
static constexpr size_t kMaxGridY = 65535;   // HIP grid limit in y and z: launches that put tiles there take at most this many per sub-batch

namespace {

void put_u32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
void put_u64(uint8_t *p, uint64_t v) { put_u32(p, (uint32_t)v); put_u32(p + 4, (uint32_t)(v >> 32)); }
uint64_t get_u64(const uint8_t *p) { return (uint64_t)get_u32(p) | ((uint64_t)get_u32(p + 4) << 32); }

// Downsample2xRGB (wsipyramid.go:10-32): (v00+v10+v01+v11+2)/4 per channel, odd edge dropped.
__global__ void __launch_bounds__(256) k_wsi_downsample(const uint8_t *src, int sw, uint8_t *dst, int dw, int dh) {
    const size_t n = (size_t)dw * dh * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3); const size_t px = i / 3;
        const int x = (int)(px % dw), y = (int)(px / dw);
        const size_t a = ((size_t)(2 * y) * sw + 2 * x) * 3 + c, b = a + 3, d = a + (size_t)sw * 3, e = d + 3;
        dst[i] = (uint8_t)(((int)src[a] + src[b] + src[d] + src[e] + 2) / 4);
    }
}

// extractTileRGB (wsicompress.go:529-555) fused with YCoCgRForward (asm_amd64.go:88-104) and the
// constant / max scan of compressWSIPlane (wsicompress.go:375-385).  grid = (chunks, tiles).
// planes: [tile][3][tw*th] u16 ; stats: [tile][3] {min, max} as u32 pairs (pre-set to 0xFFFFFFFF / 0).
__global__ void __launch_bounds__(256) k_wsi_tile_planes(const uint8_t *img, int iw, int ih, int tw, int th, int tiles_x,
                                                       int tile_base, uint16_t *planes, uint32_t *stats) {
    const int tile = blockIdx.y;                                  // slab-local index into planes / stats
    const int tx = (tile + tile_base) % tiles_x, ty = (tile + tile_base) / tiles_x;
    const size_t npx = (size_t)tw * th;
    uint16_t *py = planes + (size_t)tile * 3 * npx, *pco = py + npx, *pcg = pco + npx;
    uint32_t mn[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, mx[3] = { 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % tw), y = (int)(i / tw);
        const int sx = tx * tw + x, sy = ty * th + y;
        int r = 0, g = 0, b = 0;
        if (sx < iw && sy < ih) { const uint8_t *p = img + ((size_t)sy * iw + sx) * 3; r = p[0]; g = p[1]; b = p[2]; }
        const int co = r - b;
        const int t = b + (co >> 1);
        const int cg = g - t;
        const int yv = t + (cg >> 1);
        const uint32_t v0 = (uint16_t)yv;
        const uint32_t v1 = (uint16_t)(((int16_t)co << 1) ^ ((int16_t)co >> 15));     // ZigZag, deltazigzagcompressu16.go:108-111
        const uint32_t v2 = (uint16_t)(((int16_t)cg << 1) ^ ((int16_t)cg >> 15));
        py[i] = (uint16_t)v0; pco[i] = (uint16_t)v1; pcg[i] = (uint16_t)v2;
        mn[0] = min(mn[0], v0); mx[0] = max(mx[0], v0);
        mn[1] = min(mn[1], v1); mx[1] = max(mx[1], v1);
        mn[2] = min(mn[2], v2); mx[2] = max(mx[2], v2);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { mn[k] = min(mn[k], (uint32_t)__shfl_xor((int)mn[k], d)); mx[k] = max(mx[k], (uint32_t)__shfl_xor((int)mx[k], d)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&stats[((size_t)tile * 3 + k) * 2], mn[k]); atomicMax(&stats[((size_t)tile * 3 + k) * 2 + 1], mx[k]); }
    }
}

// YCoCgRInverse (asm_amd64.go:106-121) + cropTile (wsicompress.go:557-570): planes of tile `t` -> dst
// image region.  grid = (chunks, tiles).  place[t] = {dst x0, dst y0, crop w, crop h}.
__global__ void __launch_bounds__(256) k_wsi_planes_to_rgb(const uint16_t *planes, int tw, int th, const int4 *place,
                                                         uint8_t *dst, int dst_w) {
    const int tile = blockIdx.y;
    const int4 pl = place[tile];
    const size_t npx = (size_t)tw * th;
    const uint16_t *py = planes + (size_t)tile * 3 * npx, *pco = py + npx, *pcg = pco + npx;
    const size_t n = (size_t)pl.z * pl.w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % pl.z), y = (int)(i / pl.z);
        const size_t si = (size_t)y * tw + x;
        const int yv = py[si];
        const uint32_t uco = pco[si], ucg = pcg[si];
        const int co = (int)(int16_t)((uco >> 1) ^ (uint16_t)(-(int)(uco & 1)));       // UnZigZag, :113-116
        const int cg = (int)(int16_t)((ucg >> 1) ^ (uint16_t)(-(int)(ucg & 1)));
        const int t = yv - (cg >> 1);
        const int g = cg + t;
        const int b = t - (co >> 1);
        const int r = co + b;
        uint8_t *o = dst + ((size_t)(pl.y + y) * dst_w + (pl.x + x)) * 3;
        o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
    }
}

// Greyscale slides (channels = 1, 8 or 16 bits per sample; T = the sample type, little-endian like bytesToUint16Slice,
// wsicompress.go:573-603).  Downsample2xGrey (wsipyramid.go:34-55) works on the samples widened to u16.
template <typename T>
__global__ void __launch_bounds__(256) k_wsi_downsample_grey(const T *src, int sw, T *dst, int dw, int dh) {
    const size_t n = (size_t)dw * dh;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % dw), y = (int)(i / dw);
        const size_t a = (size_t)(2 * y) * sw + 2 * x, d = a + (size_t)sw;
        dst[i] = (T)(((uint32_t)src[a] + src[a + 1] + src[d] + src[d + 1] + 2) / 4);
    }
}

// extractTileRGB for one channel + bytesToUint16Slice + the constant / max scan.  planes: [tile][tw*th] u16 ; stats: [tile] {min, max}
template <typename T>
__global__ void __launch_bounds__(256) k_wsi_tile_plane_grey(const T *img, int iw, int ih, int tw, int th, int tiles_x,
                                                           int tile_base, uint16_t *planes, uint32_t *stats) {
    const int tile = blockIdx.y;
    const int tx = (tile + tile_base) % tiles_x, ty = (tile + tile_base) / tiles_x;
    const size_t npx = (size_t)tw * th;
    uint16_t *pl = planes + (size_t)tile * npx;
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % tw), y = (int)(i / tw);
        const int sx = tx * tw + x, sy = ty * th + y;
        const uint32_t v = (sx < iw && sy < ih) ? (uint32_t)img[(size_t)sy * iw + sx] : 0u;
        pl[i] = (uint16_t)v;
        mn = min(mn, v); mx = max(mx, v);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, d)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, d)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&stats[(size_t)tile * 2], mn); atomicMax(&stats[(size_t)tile * 2 + 1], mx); }
}

// uint16ToBytes (wsicompress.go:589-603) + cropTile: the plane of tile `t` -> dst image region
template <typename T>
__global__ void __launch_bounds__(256) k_wsi_plane_to_grey(const uint16_t *planes, int tw, int th, const int4 *place, T *dst, int dst_w) {
    const int tile = blockIdx.y;
    const int4 pl = place[tile];
    const uint16_t *src = planes + (size_t)tile * tw * th;
    const size_t n = (size_t)pl.z * pl.w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % pl.z), y = (int)(i / pl.z);
        dst[(size_t)(pl.y + y) * dst_w + (pl.x + x)] = (T)src[(size_t)y * tw + x];
    }
}

__global__ void __launch_bounds__(256) k_fill_u16(uint16_t *p, size_t n, uint16_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// constant planes of a slab in one launch: grid = (chunks, planes); fill[k] = {plane index, value}
__global__ void __launch_bounds__(256) k_fill_planes(uint16_t *planes, size_t npx, const uint2 *fill) {
    const uint2 f = fill[blockIdx.y];
    uint16_t *p = planes + (size_t)f.x * npx;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint16_t)f.y;
}

struct Level { int w, h, tx, ty, first; };

// autoLevelCount + computeLevels (wsiformat.go:244-285) with the truncation of wsicompress.go:47-77
std::vector<Level> plan_levels(int w, int h, int tw, int th, int req) {
    int n = req;
    if (n <= 0) { n = 1; int ww = w, hh = h; while (ww > tw || hh > th) { ww /= 2; hh /= 2; n++; if (ww <= 1 && hh <= 1) break; } }
    std::vector<Level> lv;
    int ww = w, hh = h;
    for (int i = 0; i < n; i++) {
        if (i > 0) { const int nw = lv[i - 1].w / 2, nh = lv[i - 1].h / 2; if (nw == 0 || nh == 0) break; ww = nw; hh = nh; }
        lv.push_back(Level{ ww, hh, (ww + tw - 1) / tw, (hh + th - 1) / th, 0 });
    }
    int idx = 0;
    for (auto &l : lv) { l.first = idx; idx += l.tx * l.ty; }
    return lv;
}

struct Mic3 {
    int w, h, tw, th, channels, bps, flags, nlev; uint64_t total; size_t data_off;
    std::vector<Level> lv;
    bool rgb() const { return channels == 3 && bps == 8; }                    // compressTileBlob / decompressTileBlob, wsicompress.go:312-317, :424-429
    bool grey() const { return channels == 1 && (bps == 8 || bps == 16); }
    bool supported() const { return (rgb() && (flags & 0x02)) || grey(); }
    int planes() const { return rgb() ? 3 : 1; }
    size_t bpp() const { return (size_t)channels * (bps == 16 ? 2 : 1); }  // bytesPerPixel, wsicompress.go:530-533
};
int parse_mic3(const uint8_t *c, size_t len, Mic3 &m) {                       // ReadMIC3Header, wsiformat.go:169-227
    if (len < 48 || memcmp(c, "MIC3", 4) != 0) return MIC_ERR_CORRUPT;
    if (get_u32(c + 4) != 1) return MIC_ERR_CORRUPT;
    m.w = (int)get_u32(c + 8); m.h = (int)get_u32(c + 12); m.tw = (int)get_u32(c + 16); m.th = (int)get_u32(c + 20);
    m.channels = c[24] | (c[25] << 8); m.bps = c[26]; m.flags = c[27]; m.nlev = c[28] | (c[29] << 8); m.total = get_u64(c + 32);
    if (len < 48 + 20 * (size_t)m.nlev) return MIC_ERR_CORRUPT;
    if (m.total > (len - 48 - 20 * (size_t)m.nlev) / 16) return MIC_ERR_CORRUPT;
    m.lv.clear();
    for (int i = 0; i < m.nlev; i++) {
        const uint8_t *p = c + 48 + 20 * (size_t)i;
        m.lv.push_back(Level{ (int)get_u32(p), (int)get_u32(p + 4), (int)get_u32(p + 8), (int)get_u32(p + 12), (int)get_u32(p + 16) });
    }
    m.data_off = 48 + 20 * (size_t)m.nlev + 16 * (size_t)m.total;
    if (m.tw <= 0 || m.th <= 0 || (size_t)m.tw * m.th > ((size_t)1 << 26)) return MIC_ERR_CORRUPT;
    // the level table must be what computeLevels writes (wsiformat.go:244-271): positive dimensions, tile counts that are the
    // ceilings of dimension / tile size, tile ranges inside the tile table.  The decoders walk tx * ty tiles of a level, so a
    // descriptor that lies about them would drive the host loops (and int arithmetic) wherever the file says.
    for (const Level &l : m.lv) {
        if (l.w <= 0 || l.h <= 0 || l.first < 0) return MIC_ERR_CORRUPT;
        if ((int64_t)l.tx != ((int64_t)l.w + m.tw - 1) / m.tw || (int64_t)l.ty != ((int64_t)l.h + m.th - 1) / m.th) return MIC_ERR_CORRUPT;
        if ((uint64_t)l.first + (uint64_t)l.tx * (uint64_t)l.ty > m.total) return MIC_ERR_CORRUPT;
    }
    return MIC_OK;
}

// decode the given tiles (global indices) of one level into dst (an image of dst_w x dst_h pixels of the slide's format);
// place[k] = where tile k goes and how much of it is kept
struct TileBlob { const uint8_t *p; size_t len; };
int decode_blobs(const Mic3 &m, const std::vector<TileBlob> &tiles, const std::vector<int4> &place,
                 uint8_t *rgb_out, int dst_w, int dst_h) {
    if (!m.supported()) return MIC_ERR_UNSUPPORTED;
    mic_hip_session *s = cur_default();
    const size_t npx = (size_t)m.tw * m.th;
    const size_t ntile = tiles.size();
    const size_t P = (size_t)m.planes(), bpp = m.bpp();
    // per-tile chunking keeps the unit workspace bounded
    const size_t per = std::min<size_t>(kMaxGridY / P, batch_units_for(npx, P));   // (tiles are a launch's grid y)
    struct Bufs { DevBuf planes, d_place, d_out; ~Bufs() { planes.release(); d_place.release(); d_out.release(); } } bufs;   // freed on every return path
    DevBuf &planes = bufs.planes, &d_place = bufs.d_place, &d_out = bufs.d_out;
    int rc;
    if ((rc = d_out.reserve((size_t)dst_w * dst_h * bpp + 64))) return rc;
    if (tiles.empty()) HIP_TRY(hipMemsetAsync(d_out.p, 0, (size_t)dst_w * dst_h * bpp, s->stream));   // nothing will write it
    for (size_t t0 = 0; t0 < ntile && rc == MIC_OK; t0 += per) {
        const size_t nt = std::min(per, ntile - t0);
        if ((rc = planes.reserve(nt * P * npx * 2 + 64))) break;
        if ((rc = d_place.reserve(nt * sizeof(int4) + 64))) break;
        if ((rc = s->ensure(1, npx))) break;
        std::vector<mic_hip_unit> units; std::vector<uint64_t> offs(1, 0); std::vector<uint8_t> comp;
        struct Fill { size_t plane; int mode; uint16_t val; const uint8_t *raw; };
        std::vector<Fill> fills;
        for (size_t k = 0; k < nt && rc == MIC_OK; k++) {
            const uint8_t *blob = tiles[t0 + k].p; const size_t bl = tiles[t0 + k].len;
            size_t pl_off[3] = { 0, 0, 0 }, pl_len[3] = { (size_t)bl, 0, 0 };              // greyscale: the blob is the plane, :477-484
            if (P == 3) {
                if (bl < 12) { rc = MIC_ERR_CORRUPT; break; }
                const size_t l0 = get_u32(blob), l1 = get_u32(blob + 4), l2 = get_u32(blob + 8);
                if (12 + l0 + l1 + l2 > bl) { rc = MIC_ERR_CORRUPT; break; }               // wsicompress.go:440-442
                pl_off[0] = 12; pl_off[1] = 12 + l0; pl_off[2] = 12 + l0 + l1; pl_len[0] = l0; pl_len[1] = l1; pl_len[2] = l2;
            }
            for (size_t p = 0; p < P; p++) {                                               // decompressWSIPlane, :487-527
                const uint8_t *d = blob + pl_off[p]; const size_t dl = pl_len[p];
                const size_t plane = k * P + p;
                if (dl == 0) { rc = MIC_ERR_CORRUPT; break; }
                if (d[0] == 0) fills.push_back(Fill{ plane, 0, 0, nullptr });
                else if (d[0] == 1) { if (dl < 3) { rc = MIC_ERR_CORRUPT; break; } fills.push_back(Fill{ plane, 1, (uint16_t)(d[1] | (d[2] << 8)), nullptr }); }
                else if (d[0] == 2) {
                    units.push_back(mic_hip_unit{ plane * npx, m.tw, m.th, 0, 0 });
                    comp.insert(comp.end(), d + 1, d + dl);
                    offs.push_back(comp.size());
                } else if (d[0] == 3) { if (dl < 1 + 2 * npx) { rc = MIC_ERR_CORRUPT; break; } fills.push_back(Fill{ plane, 3, 0, d + 1 }); }
                else { rc = MIC_ERR_CORRUPT; break; }
            }
        }
        if (rc) break;
        uint16_t *dp = (uint16_t *)planes.p;
        for (const Fill &f : fills) {
            if (f.mode == 3) HIP_TRY(hipMemcpyAsync(dp + f.plane * npx, f.raw, npx * 2, hipMemcpyHostToDevice, s->stream));
            else hipLaunchKernelGGL(k_fill_u16, dim3(64), dim3(256), 0, s->stream, dp + f.plane * npx, npx, f.val);
        }
        if (!units.empty()) {
            if ((rc = s->io_comp.reserve(comp.size() + 64))) break;
            HIP_TRY(hipMemcpyAsync(s->io_comp.p, comp.data(), comp.size(), hipMemcpyHostToDevice, s->stream));
            if ((rc = session_decode_enqueue(s, (const uint8_t *)s->io_comp.p, offs.data(), units.data(), (int)units.size(), dp))) break;
            std::vector<int32_t> st(units.size());
            if ((rc = session_decode_finish(s, st.data()))) break;
            for (int32_t v : st) if (v != MIC_OK) { rc = v; break; }
            if (rc) break;
        }
        HIP_TRY(hipMemcpyAsync(d_place.p, place.data() + t0, nt * sizeof(int4), hipMemcpyHostToDevice, s->stream));
        if (P == 3)
            hipLaunchKernelGGL(k_wsi_planes_to_rgb, dim3(16, (unsigned)nt), dim3(256), 0, s->stream, dp, m.tw, m.th, (const int4 *)d_place.p,
                               (uint8_t *)d_out.p, dst_w);
        else if (m.bps == 16)
            hipLaunchKernelGGL(k_wsi_plane_to_grey<uint16_t>, dim3(16, (unsigned)nt), dim3(256), 0, s->stream, dp, m.tw, m.th,
                               (const int4 *)d_place.p, (uint16_t *)d_out.p, dst_w);
        else
            hipLaunchKernelGGL(k_wsi_plane_to_grey<uint8_t>, dim3(16, (unsigned)nt), dim3(256), 0, s->stream, dp, m.tw, m.th,
                               (const int4 *)d_place.p, (uint8_t *)d_out.p, dst_w);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    if (rc == MIC_OK) {
        hipError_t e = hipMemcpy(rgb_out, d_out.p, (size_t)dst_w * dst_h * bpp, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = MIC_ERR_DEVICE;
    }
    return rc;
}

// the given tiles (global indices) of a MIC3 container: ExtractTileBlob (wsiformat.go:230-241), then decode_blobs
int decode_tiles(const uint8_t *c, size_t len, const Mic3 &m, const std::vector<size_t> &tiles, const std::vector<int4> &place,
                 uint8_t *rgb_out, int dst_w, int dst_h) {
    std::vector<TileBlob> blobs;
    for (size_t gi : tiles) {
        if (gi >= m.total) return MIC_ERR_CORRUPT;
        const uint8_t *e = c + 48 + 20 * (size_t)m.nlev + 16 * gi;
        const uint64_t bo = get_u64(e), bl = get_u64(e + 8);
        if (bo > len || bl > len || m.data_off + bo + bl > len) return MIC_ERR_CORRUPT;
        blobs.push_back(TileBlob{ c + m.data_off + bo, (size_t)bl });
    }
    return decode_blobs(m, blobs, place, rgb_out, dst_w, dst_h);
}

// every tile of one pyramid level (image d_img on the device): extraction + transform + plane statistics, the unit codec over all
// non-constant planes in slabs that keep planes + unit workspace bounded, then the tile blobs (compressTileBlob, wsicompress.go:312-370)
int compress_level_tiles(mic_hip_session *s, const void *d_img, const Level &L, int tile_w, int tile_h, const Mic3 &fmt,
                         std::vector<uint8_t> *blobs_out) {
    const size_t P = (size_t)fmt.planes();
    const size_t npx = (size_t)tile_w * tile_h;
    const size_t per = std::min<size_t>(kMaxGridY / P, std::max<size_t>(1, std::min<size_t>(batch_units_for(npx, P), ((size_t)8 << 30) / (P * npx * 2))));
    DevBuf planes, stats;
    int rc = MIC_OK;
    const size_t ntl = (size_t)L.tx * L.ty;
    for (size_t t0 = 0; t0 < ntl && rc == MIC_OK; t0 += per) {
        const size_t nt = std::min(per, ntl - t0);
        if ((rc = planes.reserve(nt * P * npx * 2 + 64))) break;
        if ((rc = stats.reserve(nt * P * 8 + 64))) break;
        std::vector<uint32_t> st(nt * P * 2);
        for (size_t k = 0; k < nt * P; k++) { st[2 * k] = 0xFFFFFFFFu; st[2 * k + 1] = 0; }
        if (hipMemcpyAsync(stats.p, st.data(), st.size() * 4, hipMemcpyHostToDevice, s->stream) != hipSuccess) { rc = MIC_ERR_DEVICE; break; }
        if (P == 3)
            hipLaunchKernelGGL(k_wsi_tile_planes, dim3(8, (unsigned)nt), dim3(256), 0, s->stream, (const uint8_t *)d_img, L.w, L.h,
                               tile_w, tile_h, L.tx, (int)t0, (uint16_t *)planes.p, (uint32_t *)stats.p);
        else if (fmt.bps == 16)
            hipLaunchKernelGGL(k_wsi_tile_plane_grey<uint16_t>, dim3(8, (unsigned)nt), dim3(256), 0, s->stream, (const uint16_t *)d_img,
                               L.w, L.h, tile_w, tile_h, L.tx, (int)t0, (uint16_t *)planes.p, (uint32_t *)stats.p);
        else
            hipLaunchKernelGGL(k_wsi_tile_plane_grey<uint8_t>, dim3(8, (unsigned)nt), dim3(256), 0, s->stream, (const uint8_t *)d_img,
                               L.w, L.h, tile_w, tile_h, L.tx, (int)t0, (uint16_t *)planes.p, (uint32_t *)stats.p);
        if (hipGetLastError() != hipSuccess) { rc = MIC_ERR_DEVICE; break; }
        if (hipMemcpyAsync(st.data(), stats.p, st.size() * 4, hipMemcpyDeviceToHost, s->stream) != hipSuccess) { rc = MIC_ERR_DEVICE; break; }
        if (hipStreamSynchronize(s->stream) != hipSuccess) { rc = MIC_ERR_DEVICE; break; }
        // plane modes (compressWSIPlane, wsicompress.go:373-421)
        std::vector<mic_hip_unit> units; std::vector<size_t> unit_plane;
        for (size_t p = 0; p < nt * P; p++) {
            const uint32_t mn = st[2 * p], mx = st[2 * p + 1];
            if (mn == mx) continue;                                                    // constant plane
            units.push_back(mic_hip_unit{ p * npx, tile_w, tile_h, (uint16_t)std::max<uint32_t>(mx, 255u), 2 });   // :398-402
            unit_plane.push_back(p);
        }
        std::vector<uint64_t> offs(units.size() + 1, 0); std::vector<int32_t> ust(units.size()), uns(units.size());
        std::vector<uint8_t> packed;
        if (!units.empty()) {
            if ((rc = session_encode_enqueue(s, (const uint16_t *)planes.p, units.data(), (int)units.size()))) break;
            const uint8_t *d_blobs = nullptr;
            if ((rc = session_encode_finish(s, &d_blobs, offs.data(), ust.data(), uns.data()))) break;
            packed.resize((size_t)offs.back() + 16);
            if (offs.back() && hipMemcpy(packed.data(), d_blobs, (size_t)offs.back(), hipMemcpyDeviceToHost) != hipSuccess) { rc = MIC_ERR_DEVICE; break; }
        }
        std::vector<long> unit_of(nt * P, -1);
        for (size_t k = 0; k < units.size(); k++) unit_of[unit_plane[k]] = (long)k;
        std::vector<uint16_t> rawbuf;
        for (size_t k = 0; k < nt && rc == MIC_OK; k++) {
            std::vector<uint8_t> &tb = blobs_out[t0 + k];
            tb.assign(P == 3 ? 12 : 0, 0);                                             // RGB: three plane lengths, :341-363; grey: bare plane, :366-370
            for (size_t p = 0; p < P; p++) {
                const size_t pi = k * P + p;
                const size_t before = tb.size();
                const uint32_t mn = st[2 * pi], mx = st[2 * pi + 1];
                if (mn == mx) {
                    if (mn == 0) tb.push_back(0);                                      // planeConstantZero
                    else { tb.push_back(1); tb.push_back((uint8_t)mn); tb.push_back((uint8_t)(mn >> 8)); }
                } else {
                    const long ui = unit_of[pi];
                    const int32_t ustat = ust[(size_t)ui];
                    if (ustat == MIC_OK) {
                        tb.push_back(2);
                        tb.insert(tb.end(), packed.begin() + (long)offs[(size_t)ui], packed.begin() + (long)offs[(size_t)ui + 1]);
                    } else if (ustat == MIC_ERR_USE_RLE || ustat == MIC_ERR_INCOMPRESSIBLE) {      // raw fallback, :403-414
                        rawbuf.resize(npx);
                        if (hipMemcpy(rawbuf.data(), (uint16_t *)planes.p + pi * npx, npx * 2, hipMemcpyDeviceToHost) != hipSuccess) { rc = MIC_ERR_DEVICE; break; }
                        tb.push_back(3);
                        const uint8_t *rb = (const uint8_t *)rawbuf.data();
                        tb.insert(tb.end(), rb, rb + npx * 2);
                    } else { rc = ustat; break; }
                }
                if (P == 3) put_u32(tb.data() + 4 * p, (uint32_t)(tb.size() - before));
            }
        }
    }
    planes.release(); stats.release();
    return rc;
}

}  // namespace

extern "C" {

// CompressWSI (wsicompress.go:27-171): 8-bit RGB (channels 3) or 8/16-bit greyscale (channels 1, little-endian samples)
int mic_hip_wsi_compress_ex(const uint8_t *rgb, int width, int height, int channels, int bits_per_sample, int tile_w, int tile_h,
                            int levels, uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!rgb || !out || !out_len || width <= 0 || height <= 0 || tile_w < 0 || tile_h < 0) return MIC_ERR_ARGS;
    Mic3 fmt; fmt.channels = channels; fmt.bps = bits_per_sample; fmt.flags = 0x01 | (channels == 3 ? 0x02 : 0);   // defaults(), wsiformat.go:86-96
    if (!fmt.supported()) return MIC_ERR_UNSUPPORTED;
    const size_t P = (size_t)fmt.planes(), bpp = fmt.bpp();
    if (tile_w == 0) tile_w = 256;                                                          // WSIOptions.defaults, wsiformat.go:86-96
    if (tile_h == 0) tile_h = 256;
    if ((size_t)tile_w * tile_h > ((size_t)1 << 26) || levels > 32) return MIC_ERR_UNSUPPORTED;
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    mic_hip_session *s = cur_default();
    if ((rc = s->ensure(1, (size_t)tile_w * tile_h))) return rc;
    const std::vector<Level> lv = plan_levels(width, height, tile_w, tile_h, levels);
    const int nlev = (int)lv.size();
    size_t total_tiles = 0;
    for (const Level &l : lv) total_tiles += (size_t)l.tx * l.ty;
    const size_t hdr = 48 + 20 * (size_t)nlev + 16 * total_tiles;
    if (out_cap < hdr) return MIC_ERR_CAPACITY;
    // pyramid on the device
    std::vector<DevBuf> img((size_t)nlev);
    auto cleanup = [&]() { for (auto &b : img) b.release(); };
    if ((rc = img[0].reserve((size_t)width * height * bpp + 64))) { cleanup(); return rc; }
    if (hipMemcpyAsync(img[0].p, rgb, (size_t)width * height * bpp, hipMemcpyHostToDevice, s->stream) != hipSuccess) { cleanup(); return MIC_ERR_DEVICE; }
    for (int i = 1; i < nlev; i++) {
        if ((rc = img[(size_t)i].reserve((size_t)lv[i].w * lv[i].h * bpp + 64))) { cleanup(); return rc; }
        if (P == 3)
            hipLaunchKernelGGL(k_wsi_downsample, dim3(1024), dim3(256), 0, s->stream, (const uint8_t *)img[(size_t)i - 1].p, lv[i - 1].w,
                               (uint8_t *)img[(size_t)i].p, lv[i].w, lv[i].h);
        else if (bits_per_sample == 16)
            hipLaunchKernelGGL(k_wsi_downsample_grey<uint16_t>, dim3(1024), dim3(256), 0, s->stream, (const uint16_t *)img[(size_t)i - 1].p,
                               lv[i - 1].w, (uint16_t *)img[(size_t)i].p, lv[i].w, lv[i].h);
        else
            hipLaunchKernelGGL(k_wsi_downsample_grey<uint8_t>, dim3(1024), dim3(256), 0, s->stream, (const uint8_t *)img[(size_t)i - 1].p,
                               lv[i - 1].w, (uint8_t *)img[(size_t)i].p, lv[i].w, lv[i].h);
    }
    std::vector<std::vector<uint8_t>> tile_blobs(total_tiles);
    for (int li = 0; li < nlev && rc == MIC_OK; li++)
        rc = compress_level_tiles(s, img[(size_t)li].p, lv[(size_t)li], tile_w, tile_h, fmt, tile_blobs.data() + lv[(size_t)li].first);
    cleanup();
    if (rc) return rc;
    size_t total = 0;
    for (const auto &tb : tile_blobs) total += tb.size();
    if (out_cap < hdr + total) return MIC_ERR_CAPACITY;
    memset(out, 0, hdr);                                                                    // WriteMIC3, wsiformat.go:99-165
    memcpy(out, "MIC3", 4); put_u32(out + 4, 1); put_u32(out + 8, (uint32_t)width); put_u32(out + 12, (uint32_t)height);
    put_u32(out + 16, (uint32_t)tile_w); put_u32(out + 20, (uint32_t)tile_h);
    out[24] = (uint8_t)channels; out[25] = 0; out[26] = (uint8_t)bits_per_sample; out[27] = (uint8_t)fmt.flags;
    out[28] = (uint8_t)nlev; out[29] = (uint8_t)(nlev >> 8);
    put_u64(out + 32, (uint64_t)total_tiles);
    for (int i = 0; i < nlev; i++) {
        uint8_t *ld = out + 48 + 20 * (size_t)i;
        put_u32(ld, (uint32_t)lv[(size_t)i].w); put_u32(ld + 4, (uint32_t)lv[(size_t)i].h); put_u32(ld + 8, (uint32_t)lv[(size_t)i].tx);
        put_u32(ld + 12, (uint32_t)lv[(size_t)i].ty); put_u32(ld + 16, (uint32_t)lv[(size_t)i].first);
    }
    size_t off = 0;
    for (size_t t = 0; t < total_tiles; t++) {
        uint8_t *e = out + 48 + 20 * (size_t)nlev + 16 * t;
        put_u64(e, (uint64_t)off); put_u64(e + 8, (uint64_t)tile_blobs[t].size());
        memcpy(out + hdr + off, tile_blobs[t].data(), tile_blobs[t].size());
        off += tile_blobs[t].size();
    }
    *out_len = hdr + total;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// CompressRGB (rgbcompress.go:25-27) = compressRGBTileBlob on the whole image: one "tile" of width x height
int mic_hip_rgb_compress(const uint8_t *rgb, int width, int height, uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!rgb || !out || !out_len || width <= 0 || height <= 0) return MIC_ERR_ARGS;
    if ((size_t)width * height > ((size_t)1 << 26)) return MIC_ERR_UNSUPPORTED;
    Mic3 fmt; fmt.channels = 3; fmt.bps = 8; fmt.flags = 0x03;
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    mic_hip_session *s = cur_default();
    if ((rc = s->ensure(1, (size_t)width * height))) return rc;
    DevBuf img;
    if ((rc = img.reserve((size_t)width * height * 3 + 64))) return rc;
    if (hipMemcpyAsync(img.p, rgb, (size_t)width * height * 3, hipMemcpyHostToDevice, s->stream) != hipSuccess) { img.release(); return MIC_ERR_DEVICE; }
    std::vector<uint8_t> blob;
    rc = compress_level_tiles(s, img.p, Level{ width, height, 1, 1, 0 }, width, height, fmt, &blob);
    img.release();
    if (rc) return rc;
    if (blob.size() > out_cap) return MIC_ERR_CAPACITY;
    memcpy(out, blob.data(), blob.size());
    *out_len = blob.size();
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// DecompressRGB (rgbcompress.go:31-33)
int mic_hip_rgb_decompress(const uint8_t *c, size_t len, int width, int height, uint8_t *rgb_out, size_t out_cap) try {
    if (!c || !rgb_out || width <= 0 || height <= 0) return MIC_ERR_ARGS;
    if ((size_t)width * height > ((size_t)1 << 26)) return MIC_ERR_UNSUPPORTED;
    if ((size_t)width * height * 3 > out_cap) return MIC_ERR_CAPACITY;
    Mic3 m; m.w = width; m.h = height; m.tw = width; m.th = height; m.channels = 3; m.bps = 8; m.flags = 0x03; m.nlev = 1; m.total = 1; m.data_off = 0;
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    return decode_blobs(m, std::vector<TileBlob>(1, TileBlob{ c, len }), std::vector<int4>(1, make_int4(0, 0, width, height)), rgb_out, width, height);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// MICR file = "MICR", width, height (u32 LE), CompressRGB blob (writeMICRFile, cmd/mic-compress/main.go:62-91)
int mic_hip_micr_compress(const uint8_t *rgb, int width, int height, uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!out || !out_len) return MIC_ERR_ARGS;
    if (out_cap < 12) return MIC_ERR_CAPACITY;
    size_t n = 0;
    const int rc = mic_hip_rgb_compress(rgb, width, height, out + 12, out_cap - 12, &n);
    if (rc) return rc;
    memcpy(out, "MICR", 4); put_u32(out + 4, (uint32_t)width); put_u32(out + 8, (uint32_t)height);
    *out_len = 12 + n;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_micr_info(const uint8_t *c, size_t len, int *width, int *height) try {
    if (!c) return MIC_ERR_ARGS;
    if (len < 12 || memcmp(c, "MICR", 4) != 0) return MIC_ERR_CORRUPT;
    const uint32_t w = get_u32(c + 4), h = get_u32(c + 8);
    if (w == 0 || h == 0 || w > (1u << 26) || h > (1u << 26)) return MIC_ERR_CORRUPT;
    if (width) *width = (int)w; if (height) *height = (int)h;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_micr_decompress(const uint8_t *c, size_t len, uint8_t *rgb_out, size_t out_cap) try {
    int w = 0, h = 0;
    const int rc = mic_hip_micr_info(c, len, &w, &h);
    if (rc) return rc;
    return mic_hip_rgb_decompress(c + 12, len - 12, w, h, rgb_out, out_cap);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// MIC1 file = "MIC1", width, height, pipeline 1, payload length (u32 LE each), CompressSingleFrame stream
// (writeMicFile, cmd/mic-compress/main.go:26-59; the stream's own magic tells the state count)
int mic_hip_mic1_compress(const uint16_t *pixels, int width, int height, uint16_t max_value, int n_states,
                          uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!out || !out_len) return MIC_ERR_ARGS;
    if (out_cap < 20) return MIC_ERR_CAPACITY;
    size_t n = 0;
    const int rc = mic_hip_compress_frame(pixels, width, height, max_value, n_states, out + 20, out_cap - 20, &n);
    if (rc) return rc;
    if (n > 0xFFFFFFFFu) return MIC_ERR_UNSUPPORTED;
    memcpy(out, "MIC1", 4); put_u32(out + 4, (uint32_t)width); put_u32(out + 8, (uint32_t)height); put_u32(out + 12, 1); put_u32(out + 16, (uint32_t)n);
    *out_len = 20 + n;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_mic1_info(const uint8_t *c, size_t len, int *width, int *height) try {
    if (!c) return MIC_ERR_ARGS;
    if (len < 20 || memcmp(c, "MIC1", 4) != 0) return MIC_ERR_CORRUPT;
    const uint32_t w = get_u32(c + 4), h = get_u32(c + 8);
    if (w == 0 || h == 0 || w > (1u << 26) || h > (1u << 26) || get_u32(c + 12) != 1 || (size_t)get_u32(c + 16) > len - 20) return MIC_ERR_CORRUPT;
    if (width) *width = (int)w; if (height) *height = (int)h;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_mic1_decompress(const uint8_t *c, size_t len, uint16_t *pixels_out, size_t out_cap_px) try {
    int w = 0, h = 0;
    const int rc = mic_hip_mic1_info(c, len, &w, &h);
    if (rc) return rc;
    if ((size_t)w * h > out_cap_px) return MIC_ERR_CAPACITY;
    return mic_hip_decompress_frame(c + 20, get_u32(c + 16), pixels_out, w, h);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_wsi_compress(const uint8_t *rgb, int width, int height, int tile_w, int tile_h, int levels,
                         uint8_t *out, size_t out_cap, size_t *out_len) try {
    return mic_hip_wsi_compress_ex(rgb, width, height, 3, 8, tile_w, tile_h, levels, out, out_cap, out_len);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// WSIHeader.Channels / BitsPerSample / ColorTransform (wsiformat.go:169-227)
int mic_hip_wsi_format(const uint8_t *c, size_t len, int *channels, int *bits_per_sample, int *color_transform) try {
    if (!c) return MIC_ERR_ARGS;
    Mic3 m; int rc = parse_mic3(c, len, m);
    if (rc) return rc;
    if (channels) *channels = m.channels; if (bits_per_sample) *bits_per_sample = m.bps; if (color_transform) *color_transform = (m.flags & 0x02) ? 1 : 0;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// ReadWSIHeader (wsicompress.go:299-306)
int mic_hip_wsi_info(const uint8_t *c, size_t len, int *width, int *height, int *tile_w, int *tile_h, int *levels, uint64_t *total_tiles) try {
    if (!c) return MIC_ERR_ARGS;
    Mic3 m; int rc = parse_mic3(c, len, m);
    if (rc) return rc;
    if (width) *width = m.w; if (height) *height = m.h; if (tile_w) *tile_w = m.tw; if (tile_h) *tile_h = m.th;
    if (levels) *levels = m.nlev; if (total_tiles) *total_tiles = m.total;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_wsi_level_info(const uint8_t *c, size_t len, int level, int *width, int *height, int *tiles_x, int *tiles_y) try {
    if (!c) return MIC_ERR_ARGS;
    Mic3 m; int rc = parse_mic3(c, len, m);
    if (rc) return rc;
    if (level < 0 || level >= m.nlev) return MIC_ERR_ARGS;
    const Level &L = m.lv[(size_t)level];
    if (width) *width = L.w; if (height) *height = L.h; if (tiles_x) *tiles_x = L.tx; if (tiles_y) *tiles_y = L.ty;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// DecompressWSITile (wsicompress.go:175-217): one tile, cropped at the level's edge
int mic_hip_wsi_decompress_tile(const uint8_t *c, size_t len, int level, int tile_x, int tile_y,
                                uint8_t *rgb_out, size_t out_cap, int *out_w, int *out_h) try {
    if (!c || !rgb_out) return MIC_ERR_ARGS;
    Mic3 m; int rc = parse_mic3(c, len, m);
    if (rc) return rc;
    if (level < 0 || level >= m.nlev) return MIC_ERR_ARGS;
    const Level &L = m.lv[(size_t)level];
    if (tile_x < 0 || tile_x >= L.tx || tile_y < 0 || tile_y >= L.ty) return MIC_ERR_ARGS;
    const int aw = std::min(m.tw, L.w - tile_x * m.tw), ah = std::min(m.th, L.h - tile_y * m.th);
    if (aw <= 0 || ah <= 0) return MIC_ERR_CORRUPT;
    if (!m.supported()) return MIC_ERR_UNSUPPORTED;
    if ((size_t)aw * ah * m.bpp() > out_cap) return MIC_ERR_CAPACITY;
    if (out_w) *out_w = aw; if (out_h) *out_h = ah;
    DefaultLease lease;
    if ((rc = lease.acquire())) return rc;
    std::vector<size_t> tiles(1, (size_t)L.first + (size_t)tile_y * L.tx + tile_x);
    std::vector<int4> place(1, make_int4(0, 0, aw, ah));
    return decode_tiles(c, len, m, tiles, place, rgb_out, aw, ah);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// Whole pyramid level in one batch: every tile of the level, stitched (viewer / bench path)
int mic_hip_wsi_decompress_level(const uint8_t *c, size_t len, int level, uint8_t *rgb_out, size_t out_cap) try {
    if (!c || !rgb_out) return MIC_ERR_ARGS;
    Mic3 m; int rc = parse_mic3(c, len, m);
    if (rc) return rc;
    if (level < 0 || level >= m.nlev) return MIC_ERR_ARGS;
    const Level &L = m.lv[(size_t)level];
    if (!m.supported()) return MIC_ERR_UNSUPPORTED;
    if (L.w <= 0 || L.h <= 0 || (size_t)L.w * L.h * m.bpp() > out_cap) return (L.w <= 0 || L.h <= 0) ? MIC_ERR_CORRUPT : MIC_ERR_CAPACITY;
    if ((size_t)L.tx * m.tw < (size_t)L.w || (size_t)L.ty * m.th < (size_t)L.h) return MIC_ERR_CORRUPT;
    std::vector<size_t> tiles; std::vector<int4> place;
    for (int ty = 0; ty < L.ty; ty++) for (int tx = 0; tx < L.tx; tx++) {
        const int aw = std::min(m.tw, L.w - tx * m.tw), ah = std::min(m.th, L.h - ty * m.th);
        if (aw <= 0 || ah <= 0) continue;
        tiles.push_back((size_t)L.first + (size_t)ty * L.tx + tx);
        place.push_back(make_int4(tx * m.tw, ty * m.th, aw, ah));
    }
    DefaultLease lease;
    if ((rc = lease.acquire())) return rc;
    return decode_tiles(c, len, m, tiles, place, rgb_out, L.w, L.h);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// DecompressWSIRegion (wsicompress.go:219-297): the tiles that overlap the rectangle are decoded in one batch into
// their tile-aligned bounding box, the rectangle is cut out of it.  w / h are clamped to the level like the reference does.
int mic_hip_wsi_decompress_region(const uint8_t *c, size_t len, int level, int x, int y, int w, int h,
                                  uint8_t *rgb_out, size_t out_cap, int *out_w, int *out_h) try {
    if (!c || !rgb_out) return MIC_ERR_ARGS;
    Mic3 m; int rc = parse_mic3(c, len, m);
    if (rc) return rc;
    if (level < 0 || level >= m.nlev || x < 0 || y < 0) return MIC_ERR_ARGS;
    const Level &L = m.lv[(size_t)level];
    if (L.w <= 0 || L.h <= 0 || m.tw <= 0 || m.th <= 0) return MIC_ERR_CORRUPT;
    if ((size_t)L.tx * m.tw < (size_t)L.w || (size_t)L.ty * m.th < (size_t)L.h) return MIC_ERR_CORRUPT;
    if ((int64_t)x + w > L.w) w = L.w - x;                                                  // :232-237
    if ((int64_t)y + h > L.h) h = L.h - y;
    if (w <= 0 || h <= 0) return MIC_ERR_ARGS;                                              // "MIC3: empty region"
    if (!m.supported()) return MIC_ERR_UNSUPPORTED;
    const size_t bpp = m.bpp();
    if ((size_t)w * h * bpp > out_cap) return MIC_ERR_CAPACITY;
    const int tx0 = x / m.tw, ty0 = y / m.th, tx1 = (x + w - 1) / m.tw, ty1 = (y + h - 1) / m.th;
    const int bx = tx0 * m.tw, by = ty0 * m.th;
    const int bw = std::min((tx1 + 1) * m.tw, L.w) - bx, bh = std::min((ty1 + 1) * m.th, L.h) - by;
    std::vector<size_t> tiles; std::vector<int4> place;
    for (int ty = ty0; ty <= ty1; ty++) for (int tx = tx0; tx <= tx1; tx++) {
        const int aw = std::min(m.tw, L.w - tx * m.tw), ah = std::min(m.th, L.h - ty * m.th);
        if (aw <= 0 || ah <= 0) continue;
        tiles.push_back((size_t)L.first + (size_t)ty * L.tx + tx);
        place.push_back(make_int4(tx * m.tw - bx, ty * m.th - by, aw, ah));
    }
    std::vector<uint8_t> box((size_t)bw * bh * bpp);
    {
        DefaultLease lease;
        if ((rc = lease.acquire())) return rc;
        if ((rc = decode_tiles(c, len, m, tiles, place, box.data(), bw, bh))) return rc;
    }
    for (int r = 0; r < h; r++)
        memcpy(rgb_out + (size_t)r * w * bpp, box.data() + ((size_t)(y - by + r) * bw + (size_t)(x - bx)) * bpp, (size_t)w * bpp);
    if (out_w) *out_w = w;
    if (out_h) *out_h = h;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)


}  // extern "C"

// ==========================================================================================
// MIC3 on a device-resident slide (what bench.py times for BASELINE config 5).  mic_hip_session_wsi_encode runs the pyramid, the
// tile extraction + YCoCg-R + plane statistics and the unit codec exactly as mic_hip_wsi_compress_ex does, but the coded planes
// never leave the device: they are appended to a store the session owns (device bytes + one host record per plane).
// mic_hip_session_wsi_write turns the store into the MIC3 file (WriteMIC3, wsiformat.go:99-165: the only step that needs the
// bytes on the host); mic_hip_session_wsi_decode_level decodes every tile of a level from the store into a device image.
struct WsiPlane { uint8_t mode; uint16_t value; uint64_t off; uint32_t len; };          // mode 0 / 1: no bytes; 2: stream; 3: raw pixels
struct mic_hip_wsi_store {
    Mic3 fmt; std::vector<Level> lv; size_t total_tiles = 0;
    std::vector<WsiPlane> planes;                                                       // total_tiles * fmt.planes(), tile-major
    DevBuf bytes; size_t used = 0;
    int append(hipStream_t st, const void *d_src, size_t n, uint64_t *off) {            // grows by copying (rare: starts at the raw size / 2)
        if (used + n > bytes.cap) {
            DevBuf nb;
            int rc = nb.reserve(std::max(bytes.cap * 2, used + n));
            if (rc) return rc;
            if (used && hipMemcpyAsync(nb.p, bytes.p, used, hipMemcpyDeviceToDevice, st) != hipSuccess) { nb.release(); return MIC_ERR_DEVICE; }
            if (hipStreamSynchronize(st) != hipSuccess) { nb.release(); return MIC_ERR_DEVICE; }
            bytes.release(); bytes = nb;
        }
        if (n && hipMemcpyAsync((char *)bytes.p + used, d_src, n, hipMemcpyDeviceToDevice, st) != hipSuccess) return MIC_ERR_DEVICE;
        *off = used; used += n;
        return MIC_OK;
    }
};

void mic_wsi_store_free(mic_hip_wsi_store *w) { if (w) { w->bytes.release(); delete w; } }

namespace {

// one pyramid level into the store: as compress_level_tiles up to the unit codec, then device-to-device appends
int store_level_tiles(mic_hip_session *s, mic_hip_wsi_store &W, const void *d_img, const Level &L) {
    const Mic3 &fmt = W.fmt;
    const size_t P = (size_t)fmt.planes(), npx = (size_t)fmt.tw * fmt.th;
    const size_t per = std::min<size_t>(kMaxGridY / P, std::max<size_t>(1, std::min<size_t>(batch_units_for(npx, P), ((size_t)16 << 30) / (P * npx * 2))));
    DevBuf &planes = s->wsi_planes, &stats = s->wsi_stats;
    const size_t ntl = (size_t)L.tx * L.ty;
    int rc;
    for (size_t t0 = 0; t0 < ntl; t0 += per) {
        const size_t nt = std::min(per, ntl - t0);
        if ((rc = planes.reserve(nt * P * npx * 2 + 64)) || (rc = stats.reserve(nt * P * 8 + 64))) return rc;
        std::vector<uint32_t> st(nt * P * 2);
        for (size_t k = 0; k < nt * P; k++) { st[2 * k] = 0xFFFFFFFFu; st[2 * k + 1] = 0; }
        HIP_TRY(hipMemcpyAsync(stats.p, st.data(), st.size() * 4, hipMemcpyHostToDevice, s->stream));
        s->timer.reset(s->stream); s->timer.mark("k_wsi_tile_planes");
        if (P == 3)
            hipLaunchKernelGGL(k_wsi_tile_planes, dim3(8, (unsigned)nt), dim3(256), 0, s->stream, (const uint8_t *)d_img, L.w, L.h,
                               fmt.tw, fmt.th, L.tx, (int)t0, (uint16_t *)planes.p, (uint32_t *)stats.p);
        else if (fmt.bps == 16)
            hipLaunchKernelGGL(k_wsi_tile_plane_grey<uint16_t>, dim3(8, (unsigned)nt), dim3(256), 0, s->stream, (const uint16_t *)d_img,
                               L.w, L.h, fmt.tw, fmt.th, L.tx, (int)t0, (uint16_t *)planes.p, (uint32_t *)stats.p);
        else
            hipLaunchKernelGGL(k_wsi_tile_plane_grey<uint8_t>, dim3(8, (unsigned)nt), dim3(256), 0, s->stream, (const uint8_t *)d_img,
                               L.w, L.h, fmt.tw, fmt.th, L.tx, (int)t0, (uint16_t *)planes.p, (uint32_t *)stats.p);
        s->timer.mark("end");
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(st.data(), stats.p, st.size() * 4, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        std::vector<mic_hip_unit> units; std::vector<size_t> unit_plane;
        for (size_t p = 0; p < nt * P; p++) {
            const uint32_t mn = st[2 * p], mx = st[2 * p + 1];
            if (mn == mx) continue;
            units.push_back(mic_hip_unit{ p * npx, fmt.tw, fmt.th, (uint16_t)std::max<uint32_t>(mx, 255u), 2 });   // wsicompress.go:398-402
            unit_plane.push_back(p);
        }
        std::vector<uint64_t> offs(units.size() + 1, 0); std::vector<int32_t> ust(units.size()), uns(units.size());
        const uint8_t *d_blobs = nullptr; uint64_t base = 0;
        if (!units.empty()) {
            if ((rc = session_encode_enqueue(s, (const uint16_t *)planes.p, units.data(), (int)units.size()))) return rc;
            if ((rc = session_encode_finish(s, &d_blobs, offs.data(), ust.data(), uns.data()))) return rc;
            if ((rc = W.append(s->stream, d_blobs, (size_t)offs.back(), &base))) return rc;          // every stream of the slab in one copy
        }
        std::vector<long> unit_of(nt * P, -1);
        for (size_t k = 0; k < units.size(); k++) unit_of[unit_plane[k]] = (long)k;
        for (size_t p = 0; p < nt * P; p++) {
            WsiPlane &wp = W.planes[((size_t)L.first + t0) * P + p];
            const uint32_t mn = st[2 * p], mx = st[2 * p + 1];
            if (mn == mx) { wp = WsiPlane{ (uint8_t)(mn == 0 ? 0 : 1), (uint16_t)mn, 0, 0 }; continue; }
            const long ui = unit_of[p];
            const int32_t ustat = ust[(size_t)ui];
            if (ustat == MIC_OK) wp = WsiPlane{ 2, 0, base + offs[(size_t)ui], (uint32_t)(offs[(size_t)ui + 1] - offs[(size_t)ui]) };
            else if (ustat == MIC_ERR_USE_RLE || ustat == MIC_ERR_INCOMPRESSIBLE) {                     // raw fallback, :403-414
                uint64_t o = 0;
                if ((rc = W.append(s->stream, (uint16_t *)planes.p + p * npx, npx * 2, &o))) return rc;
                wp = WsiPlane{ 3, 0, o, (uint32_t)(npx * 2) };
            } else return ustat;
        }
        HIP_TRY(hipStreamSynchronize(s->stream));                                                    // the slab's planes are reused by the next one
    }
    return MIC_OK;
}

}  // namespace

extern "C" {

int mic_hip_session_wsi_encode(mic_hip_session *s, const uint8_t *d_pixels, int width, int height, int channels, int bits_per_sample,
                               int tile_w, int tile_h, int levels, uint64_t *total_tiles, uint64_t *compressed_bytes) try {
    if (!s || !d_pixels || width <= 0 || height <= 0 || tile_w < 0 || tile_h < 0) return MIC_ERR_ARGS;
    Mic3 fmt; fmt.channels = channels; fmt.bps = bits_per_sample; fmt.flags = 0x01 | (channels == 3 ? 0x02 : 0);
    if (!fmt.supported()) return MIC_ERR_UNSUPPORTED;
    if (tile_w == 0) tile_w = 256;
    if (tile_h == 0) tile_h = 256;
    if ((size_t)tile_w * tile_h > ((size_t)1 << 26) || levels > 32) return MIC_ERR_UNSUPPORTED;
    int rc = s->activate();
    if (rc) return rc;
    if ((rc = s->ensure(1, (size_t)tile_w * tile_h))) return rc;
    if (!s->wsi) s->wsi = new mic_hip_wsi_store();
    mic_hip_wsi_store &W = *s->wsi;
    fmt.w = width; fmt.h = height; fmt.tw = tile_w; fmt.th = tile_h;
    W.fmt = fmt; W.lv = plan_levels(width, height, tile_w, tile_h, levels);
    W.fmt.nlev = (int)W.lv.size();
    W.total_tiles = 0;
    for (const Level &l : W.lv) W.total_tiles += (size_t)l.tx * l.ty;
    W.fmt.total = W.total_tiles;
    W.planes.assign(W.total_tiles * (size_t)fmt.planes(), WsiPlane{ 0, 0, 0, 0 });
    W.used = 0;
    const size_t bpp = fmt.bpp();
    if ((rc = W.bytes.reserve((size_t)width * height * bpp / 2 + (1 << 20)))) return rc;
    // pyramid on the device (Downsample2xRGB / Downsample2xGrey, wsipyramid.go:10-55); level 0 is the caller's buffer
    std::vector<DevBuf> &img = s->wsi_pyr;
    if (img.size() < W.lv.size()) img.resize(W.lv.size());
    const void *prev = d_pixels;
    for (size_t i = 1; i < W.lv.size(); i++) if ((rc = img[i].reserve((size_t)W.lv[i].w * W.lv[i].h * bpp + 64))) return rc;
    s->timer.reset(s->stream); s->timer.mark("k_wsi_downsample");
    for (size_t i = 1; i < W.lv.size(); i++) {
        if (channels == 3)
            hipLaunchKernelGGL(k_wsi_downsample, dim3(1024), dim3(256), 0, s->stream, (const uint8_t *)prev, W.lv[i - 1].w, (uint8_t *)img[i].p, W.lv[i].w, W.lv[i].h);
        else if (bits_per_sample == 16)
            hipLaunchKernelGGL(k_wsi_downsample_grey<uint16_t>, dim3(1024), dim3(256), 0, s->stream, (const uint16_t *)prev, W.lv[i - 1].w, (uint16_t *)img[i].p, W.lv[i].w, W.lv[i].h);
        else
            hipLaunchKernelGGL(k_wsi_downsample_grey<uint8_t>, dim3(1024), dim3(256), 0, s->stream, (const uint8_t *)prev, W.lv[i - 1].w, (uint8_t *)img[i].p, W.lv[i].w, W.lv[i].h);
        prev = img[i].p;
    }
    s->timer.mark("end");
    HIP_TRY(hipGetLastError());
    for (size_t i = 0; i < W.lv.size(); i++)
        if ((rc = store_level_tiles(s, W, i == 0 ? (const void *)d_pixels : (const void *)img[i].p, W.lv[i]))) return rc;
    if (total_tiles) *total_tiles = W.total_tiles;
    if (compressed_bytes) {                                                     // size of the file mic_hip_session_wsi_write would produce
        uint64_t n = 48 + 20 * (uint64_t)W.lv.size() + 16 * (uint64_t)W.total_tiles;
        const size_t P = (size_t)fmt.planes();
        for (size_t t = 0; t < W.total_tiles; t++) {
            if (P == 3) n += 12;
            for (size_t p = 0; p < P; p++) { const WsiPlane &wp = W.planes[t * P + p]; n += wp.mode == 0 ? 1 : wp.mode == 1 ? 3 : 1 + (uint64_t)wp.len; }
        }
        *compressed_bytes = n;
    }
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// WriteMIC3 (wsiformat.go:99-165) around the store: header, level table, tile table, tile blobs ([Ylen][Colen][Cglen] + planes,
// wsicompress.go:341-363; grey: the bare plane).  One device-to-host copy of the store's bytes.
// The container's payload -- every tile blob, in container order -- is put together on the DEVICE: the plane records are host
// data (mode, constant, offset and length of each coded plane), so the host lays the tiles out (a prefix sum over 16 bytes per
// plane) and one kernel copies every plane's bytes from the store to its place, with the mode byte in front and, for RGB, the
// tile's three plane lengths (compressTileBlob, wsicompress.go:334-364).  WriteMIC3 then is one transfer of the payload straight
// into the caller's buffer behind the header and the tile index; a multi-GPU writer gathers the payload device to device.
struct WsiRec { uint64_t src, dst; uint32_t len; uint32_t mode_value; };          // mode_value = mode | value << 8
__global__ void __launch_bounds__(256) k_wsi_assemble(const WsiRec *recs, int P, const uint8_t *store, uint8_t *payload) {
    const size_t t = blockIdx.x;
    typedef uint32_t wv4 __attribute__((ext_vector_type(4)));
    typedef wv4 WQ __attribute__((aligned(1)));
    for (int p = 0; p < P; p++) {
        const WsiRec r = recs[t * (size_t)P + (size_t)p];
        const uint32_t mode = r.mode_value & 0xFFu, value = r.mode_value >> 8;
        uint8_t *d = payload + r.dst;
        const uint32_t plen = mode == 0 ? 1u : mode == 1 ? 3u : 1u + r.len;
        if (threadIdx.x == 0) {
            d[0] = (uint8_t)mode;
            if (mode == 1) { d[1] = (uint8_t)value; d[2] = (uint8_t)(value >> 8); }
            if (P == 3) {                                                               // [Y_len][Co_len][Cg_len], u32 LE, in front of the tile's planes
                uint8_t *h = payload + recs[t * 3].dst - 12 + 4 * p;
                h[0] = (uint8_t)plen; h[1] = (uint8_t)(plen >> 8); h[2] = (uint8_t)(plen >> 16); h[3] = (uint8_t)(plen >> 24);
            }
        }
        if (mode >= 2) {
            const uint8_t *sp = store + r.src; uint8_t *dp = d + 1;
            const uint32_t nv = r.len / 16;
            for (uint32_t i = threadIdx.x; i < nv; i += 256) *(WQ *)(dp + (size_t)i * 16) = *(const WQ *)(sp + (size_t)i * 16);
            if (threadIdx.x < (r.len & 15u)) dp[(size_t)nv * 16 + threadIdx.x] = sp[(size_t)nv * 16 + threadIdx.x];
        }
    }
}

// lays the payload out and builds it in s->wsi_payload; tlen[t] = bytes of tile t, *total = their sum
static int wsi_assemble(mic_hip_session *s, std::vector<uint64_t> &tlen, uint64_t *total_out) {
    mic_hip_wsi_store &W = *s->wsi;
    const size_t P = (size_t)W.fmt.planes();
    std::vector<WsiRec> recs(W.total_tiles * P);
    tlen.assign(W.total_tiles, 0);
    uint64_t off = 0;
    for (size_t t = 0; t < W.total_tiles; t++) {
        const uint64_t t0 = off;
        if (P == 3) off += 12;
        for (size_t p = 0; p < P; p++) {
            const WsiPlane &wp = W.planes[t * P + p];
            const uint64_t n = wp.mode == 0 ? 1 : wp.mode == 1 ? 3 : 1 + (uint64_t)wp.len;
            recs[t * P + p] = WsiRec{ wp.off, off, wp.mode >= 2 ? wp.len : 0u, (uint32_t)wp.mode | ((uint32_t)wp.value << 8) };
            off += n;
        }
        tlen[t] = off - t0;
    }
    *total_out = off;
    int rc;
    if ((rc = s->wsi_payload.reserve((size_t)off + 64))) return rc;
    if ((rc = s->wsi_recs.reserve(recs.size() * sizeof(WsiRec) + 64))) return rc;
    if (!recs.empty()) {
        HIP_TRY(hipMemcpyAsync(s->wsi_recs.p, recs.data(), recs.size() * sizeof(WsiRec), hipMemcpyHostToDevice, s->stream));
        hipLaunchKernelGGL(k_wsi_assemble, dim3((unsigned)W.total_tiles), dim3(256), 0, s->stream, (const WsiRec *)s->wsi_recs.p, (int)P,
                           (const uint8_t *)W.bytes.p, (uint8_t *)s->wsi_payload.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s->stream));                                       // (recs is a stack-scope vector)
    }
    return MIC_OK;
}

// The store as the container's payload, on the device: *d_payload (valid until the session's next wsi call), its size, and the
// byte length of every tile in container order (tile_lens[cap >= total tiles], host).  What a multi-GPU writer gathers.
int mic_hip_session_wsi_payload(mic_hip_session *s, const uint8_t **d_payload, uint64_t *payload_bytes, uint64_t *tile_lens, size_t cap) try {
    if (!s || !d_payload || !payload_bytes || !tile_lens || !s->wsi) return MIC_ERR_ARGS;
    int rc = s->activate();
    if (rc) return rc;
    if (cap < s->wsi->total_tiles) return MIC_ERR_CAPACITY;
    if ((size_t)kMaxGridX < s->wsi->total_tiles) return MIC_ERR_UNSUPPORTED;
    std::vector<uint64_t> tlen; uint64_t total = 0;
    if ((rc = wsi_assemble(s, tlen, &total))) return rc;
    for (size_t t = 0; t < tlen.size(); t++) tile_lens[t] = tlen[t];
    *d_payload = (const uint8_t *)s->wsi_payload.p; *payload_bytes = total;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_session_wsi_write(mic_hip_session *s, uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!s || !out || !out_len || !s->wsi) return MIC_ERR_ARGS;
    int rc = s->activate();
    if (rc) return rc;
    mic_hip_wsi_store &W = *s->wsi;
    if ((size_t)kMaxGridX < W.total_tiles) return MIC_ERR_UNSUPPORTED;
    const size_t nlev = W.lv.size();
    const size_t hdr = 48 + 20 * nlev + 16 * W.total_tiles;
    std::vector<uint64_t> tlen; uint64_t total = 0;
    if ((rc = wsi_assemble(s, tlen, &total))) return rc;
    if (out_cap < hdr + total) return MIC_ERR_CAPACITY;
    if ((rc = micapi::host_copy(s->device, s->wsi_payload.p, out + hdr, (size_t)total, false))) return rc;   // (the header is written meanwhile? no: after -- it is 0.4 MB)
    memset(out, 0, hdr);
    memcpy(out, "MIC3", 4); put_u32(out + 4, 1); put_u32(out + 8, (uint32_t)W.fmt.w); put_u32(out + 12, (uint32_t)W.fmt.h);
    put_u32(out + 16, (uint32_t)W.fmt.tw); put_u32(out + 20, (uint32_t)W.fmt.th);
    out[24] = (uint8_t)W.fmt.channels; out[25] = 0; out[26] = (uint8_t)W.fmt.bps; out[27] = (uint8_t)W.fmt.flags;
    out[28] = (uint8_t)nlev; out[29] = (uint8_t)(nlev >> 8);
    put_u64(out + 32, (uint64_t)W.total_tiles);
    for (size_t i = 0; i < nlev; i++) {
        uint8_t *ld = out + 48 + 20 * i;
        put_u32(ld, (uint32_t)W.lv[i].w); put_u32(ld + 4, (uint32_t)W.lv[i].h); put_u32(ld + 8, (uint32_t)W.lv[i].tx);
        put_u32(ld + 12, (uint32_t)W.lv[i].ty); put_u32(ld + 16, (uint32_t)W.lv[i].first);
    }
    uint64_t off = 0;
    for (size_t t = 0; t < W.total_tiles; t++) {
        uint8_t *e = out + 48 + 20 * nlev + 16 * t;
        put_u64(e, off); put_u64(e + 8, tlen[t]);
        off += tlen[t];
    }
    *out_len = hdr + (size_t)total;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// every tile of one level, from the store, into a device image of the level's size (bytes per pixel as the slide's)
int mic_hip_session_wsi_decode_level(mic_hip_session *s, int level, uint8_t *d_pixels_out, size_t out_cap) try {
    if (!s || !d_pixels_out || !s->wsi) return MIC_ERR_ARGS;
    int rc = s->activate();
    if (rc) return rc;
    mic_hip_wsi_store &W = *s->wsi;
    if (level < 0 || level >= (int)W.lv.size()) return MIC_ERR_ARGS;
    const Mic3 &m = W.fmt;
    const Level &L = W.lv[(size_t)level];
    const size_t P = (size_t)m.planes(), bpp = m.bpp(), npx = (size_t)m.tw * m.th;
    if ((size_t)L.w * L.h * bpp > out_cap) return MIC_ERR_CAPACITY;
    const size_t ntl = (size_t)L.tx * L.ty;
    const size_t per = std::min<size_t>(kMaxGridY / P, std::max<size_t>(1, std::min<size_t>(batch_units_for(npx, P), ((size_t)16 << 30) / (P * npx * 2))));
    DevBuf &planes = s->wsi_planes, &aux = s->wsi_stats;
    for (size_t t0 = 0; t0 < ntl; t0 += per) {
        const size_t nt = std::min(per, ntl - t0);
        if ((rc = planes.reserve(nt * P * npx * 2 + 64))) return rc;
        std::vector<mic_hip_unit> units; std::vector<uint64_t> begins, ends; std::vector<uint2> fills; std::vector<int4> place(nt);
        uint16_t *dp = (uint16_t *)planes.p;
        for (size_t k = 0; k < nt; k++) {
            const size_t t = t0 + k; const int tx = (int)(t % (size_t)L.tx), ty = (int)(t / (size_t)L.tx);
            place[k] = make_int4(tx * m.tw, ty * m.th, std::min(m.tw, L.w - tx * m.tw), std::min(m.th, L.h - ty * m.th));
            for (size_t p = 0; p < P; p++) {
                const WsiPlane &wp = W.planes[((size_t)L.first + t) * P + p];
                const size_t plane = k * P + p;
                if (wp.mode <= 1) fills.push_back(make_uint2((uint32_t)plane, wp.mode ? wp.value : 0u));
                else if (wp.mode == 2) { units.push_back(mic_hip_unit{ plane * npx, m.tw, m.th, 0, 0 }); begins.push_back(wp.off); ends.push_back(wp.off + wp.len); }
                else HIP_TRY(hipMemcpyAsync(dp + plane * npx, (const char *)W.bytes.p + wp.off, npx * 2, hipMemcpyDeviceToDevice, s->stream));
            }
        }
        const size_t aux_bytes = fills.size() * sizeof(uint2) + nt * sizeof(int4) + 64;
        if ((rc = aux.reserve(aux_bytes))) return rc;
        int4 *d_place = (int4 *)aux.p; uint2 *d_fill = (uint2 *)((char *)aux.p + nt * sizeof(int4));
        HIP_TRY(hipMemcpyAsync(d_place, place.data(), nt * sizeof(int4), hipMemcpyHostToDevice, s->stream));
        if (!fills.empty()) {
            HIP_TRY(hipMemcpyAsync(d_fill, fills.data(), fills.size() * sizeof(uint2), hipMemcpyHostToDevice, s->stream));
            s->timer.reset(s->stream); s->timer.mark("k_fill_planes");
            for (size_t f0 = 0; f0 < fills.size(); f0 += 65535)
                hipLaunchKernelGGL(k_fill_planes, dim3(4, (unsigned)std::min<size_t>(65535, fills.size() - f0)), dim3(256), 0, s->stream, dp, npx, (const uint2 *)d_fill + f0);
        }
        if (!units.empty()) {
            if ((rc = session_decode_enqueue_spans(s, (const uint8_t *)W.bytes.p, begins.data(), ends.data(), units.data(), (int)units.size(), dp))) return rc;
            std::vector<int32_t> st(units.size());
            if ((rc = session_decode_finish(s, st.data()))) return rc;
            for (int32_t v : st) if (v != MIC_OK) return v;
        }
        s->timer.reset(s->stream); s->timer.mark("k_wsi_planes_to_pixels");
        if (P == 3)
            hipLaunchKernelGGL(k_wsi_planes_to_rgb, dim3(16, (unsigned)nt), dim3(256), 0, s->stream, dp, m.tw, m.th, (const int4 *)d_place, d_pixels_out, L.w);
        else if (m.bps == 16)
            hipLaunchKernelGGL(k_wsi_plane_to_grey<uint16_t>, dim3(16, (unsigned)nt), dim3(256), 0, s->stream, dp, m.tw, m.th, (const int4 *)d_place, (uint16_t *)d_pixels_out, L.w);
        else
            hipLaunchKernelGGL(k_wsi_plane_to_grey<uint8_t>, dim3(16, (unsigned)nt), dim3(256), 0, s->stream, dp, m.tw, m.th, (const int4 *)d_place, d_pixels_out, L.w);
        s->timer.mark("end");
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_session_wsi_levels(mic_hip_session *s, int *levels, int *widths, int *heights, int cap) try {
    if (!s || !s->wsi || !levels) return MIC_ERR_ARGS;
    *levels = (int)s->wsi->lv.size();
    for (int i = 0; i < *levels && i < cap; i++) { if (widths) widths[i] = s->wsi->lv[(size_t)i].w; if (heights) heights[i] = s->wsi->lv[(size_t)i].h; }
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

}  // extern "C"
