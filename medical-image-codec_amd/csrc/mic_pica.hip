// mic_pica.hip -- PICA: content-adaptive strips with a per-strip predictor choice (parallelstripsadaptive.go).
//
// CompressParallelStripsAdaptive = adaptiveStripBoundaries (equal-cost partition of the rows by summed |vertical delta|) ->
// every strip coded twice, CompressSingleFrame (avg predictor) and CompressSingleFrameGrad (gradient-adaptive predictor), the
// smaller blob kept (ties: gradient) -> "PICA" header + 16-byte entries {y0, offset, length, flags} + blobs.
// On the device: the image is uploaded once; one kernel sums the row costs; the 2 x strips units (two predictors over the same
// pixels) go through the unit codec in one batch; the host does the float64 partition (it must round exactly as the reference's
// float64 does) and writes the container.  Decode: one batch over the strips, each with the predictor its flags word names.
#include "mic_session.h"

namespace {

void put_u32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// rowCost[y] = sum over x of |p[y][x] - p[y-1][x]|, y >= 1 (parallelstripsadaptive.go:236-247).  One group per row.
__global__ void __launch_bounds__(256) k_pica_rowcost(const uint16_t *px, int w, int h, unsigned long long *cost) {
    const int y = (int)blockIdx.x + 1;
    if (y >= h) return;
    const uint16_t *a = px + (size_t)y * w, *b = a - w;
    unsigned long long sum = 0;
    for (int x = (int)threadIdx.x; x < w; x += 256) sum += __sad((int)a[x], (int)b[x], 0u);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    __shared__ unsigned long long s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) cost[y] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// adaptiveStripBoundaries, parallelstripsadaptive.go:222-289 -- float64 throughout, evaluated in the reference's order
std::vector<int> pica_boundaries(const std::vector<unsigned long long> &cost, int height, int num_strips) {
    std::vector<int> starts;
    if (num_strips >= height) { for (int i = 0; i < height; i++) starts.push_back(i); return starts; }
    if (num_strips == 1) return std::vector<int>(1, 0);
    std::vector<double> cum((size_t)height + 1, 0.0);
    for (int y = 0; y < height; y++) cum[(size_t)y + 1] = cum[(size_t)y] + (y ? (double)cost[(size_t)y] : 0.0);
    volatile double total = cum[(size_t)height];
    starts.assign((size_t)num_strips, 0);
    if (total == 0) {
        for (int i = 1; i < num_strips; i++) starts[(size_t)i] = (int)((long long)i * height / num_strips);
        return starts;
    }
    for (int i = 1; i < num_strips; i++) {
        volatile double prod = total * (double)i;
        const double target = prod / (double)num_strips;
        int lo = starts[(size_t)i - 1] + 1, hi = height;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cum[(size_t)mid] < target) lo = mid + 1; else hi = mid; }
        if (lo >= height) lo = height - 1;
        starts[(size_t)i] = lo;
    }
    return starts;
}

}  // namespace

extern "C" {

// CompressSingleFrameGrad (multiframecompress.go:111-129): gradient-adaptive predictor, two-state FSE, one-state fallback
int mic_hip_compress_frame_grad(const uint16_t *pixels, int width, int height, uint16_t max_value,
                                uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!pixels || !out || !out_len || width <= 0 || height <= 0) return MIC_ERR_ARGS;
    const size_t npx = (size_t)width * (size_t)height;
    if (npx > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    mic_hip_session *s = cur_default();
    if ((rc = s->ensure(1, npx))) return rc;
    if ((rc = s->io_px.reserve(npx * 2 + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(s->io_px.p, pixels, npx * 2, hipMemcpyHostToDevice, s->stream));
    const mic_hip_unit unit{ 0, width, height, max_value, (uint16_t)(2 | MIC_HIP_PRED_GRAD) };
    if ((rc = session_encode_enqueue(s, (const uint16_t *)s->io_px.p, &unit, 1))) return rc;
    uint64_t offs[2]; int32_t st, ns; const uint8_t *d_blobs = nullptr;
    if ((rc = session_encode_finish(s, &d_blobs, offs, &st, &ns))) return rc;
    if (st != MIC_OK) return st;
    const size_t len = (size_t)(offs[1] - offs[0]);
    if (len > out_cap) return MIC_ERR_CAPACITY;
    HIP_TRY(hipMemcpy(out, d_blobs + offs[0], len, hipMemcpyDeviceToHost));
    *out_len = len;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// DecompressSingleFrameGrad (multiframecompress.go:132-142)
int mic_hip_decompress_frame_grad(const uint8_t *c, size_t len, uint16_t *pixels_out, int width, int height) try {
    if (!c || !pixels_out || width <= 0 || height <= 0) return MIC_ERR_ARGS;
    if (len == 0) return MIC_ERR_CORRUPT;
    const size_t npx = (size_t)width * (size_t)height;
    if (npx > ((size_t)1 << 28) || len > 0xFFFFFFF0ull) return MIC_ERR_UNSUPPORTED;
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    mic_hip_session *s = cur_default();
    if ((rc = s->ensure(1, npx))) return rc;
    if ((rc = s->io_px.reserve(npx * 2 + 64))) return rc;
    if ((rc = s->io_comp.reserve(len + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(s->io_comp.p, c, len, hipMemcpyHostToDevice, s->stream));
    const mic_hip_unit unit{ 0, width, height, 0, (uint16_t)(2 | MIC_HIP_PRED_GRAD) };
    const uint64_t offs[2] = { 0, len };
    if ((rc = session_decode_enqueue(s, (const uint8_t *)s->io_comp.p, offs, &unit, 1, (uint16_t *)s->io_px.p))) return rc;
    int32_t st;
    if ((rc = session_decode_finish(s, &st))) return rc;
    if (st != MIC_OK) return st;
    HIP_TRY(hipMemcpy(pixels_out, s->io_px.p, npx * 2, hipMemcpyDeviceToHost));
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// CompressParallelStripsAdaptive (parallelstripsadaptive.go:54-137)
int mic_hip_pica_compress(const uint16_t *pixels, int width, int height, uint16_t max_value, int num_strips,
                          uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!pixels || !out || !out_len || width <= 0 || height <= 0 || num_strips <= 0) return MIC_ERR_ARGS;
    if ((size_t)width * (size_t)height > ((size_t)1 << 31)) return MIC_ERR_UNSUPPORTED;
    if (num_strips > height) num_strips = height;                                          // :61-66
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    mic_hip_session *s = cur_default();
    if ((rc = s->ensure(1, 1))) return rc;                                                  // (the stream)
    const size_t npx = (size_t)width * (size_t)height;
    if ((rc = s->io_px.reserve(npx * 2 + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(s->io_px.p, pixels, npx * 2, hipMemcpyHostToDevice, s->stream));
    std::vector<unsigned long long> cost((size_t)height, 0ull);
    if (num_strips > 1 && num_strips < height) {
        DevBuf d_cost;
        if ((rc = d_cost.reserve((size_t)height * 8))) return rc;
        hipLaunchKernelGGL(k_pica_rowcost, dim3((unsigned)std::max(1, height - 1)), dim3(256), 0, s->stream, (const uint16_t *)s->io_px.p, width, height,
                           (unsigned long long *)d_cost.p);
        hipError_t e = hipMemcpyAsync(cost.data(), d_cost.p, (size_t)height * 8, hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        d_cost.release();
        if (e != hipSuccess) return MIC_ERR_DEVICE;
        cost[0] = 0;
    }
    const std::vector<int> starts = pica_boundaries(cost, height, num_strips);
    const int actual = (int)starts.size();
    const size_t header = 16 + (size_t)actual * 16;
    if (out_cap < header) return MIC_ERR_CAPACITY;
    // units 2 s (avg) and 2 s + 1 (gradient) over the same pixels, in sub-batches that keep the workspace bounded
    std::vector<std::vector<uint8_t>> blob((size_t)actual);
    std::vector<uint32_t> flags((size_t)actual, 0);
    int s0 = 0;
    while (s0 < actual) {
        size_t max_px = 0; int s1 = s0;
        while (s1 < actual) {
            const int y0 = starts[(size_t)s1], y1 = (s1 + 1 < actual) ? starts[(size_t)s1 + 1] : height;
            const size_t mp = std::max(max_px, (size_t)(y1 - y0) * (size_t)width);
            if (mp > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
            if (s1 > s0 && unit_ws_bytes(mp) * 2 * (size_t)(s1 - s0 + 1) > kWorkspaceBudget) break;
            max_px = mp; s1++;
        }
        const int ns = s1 - s0;
        // a strip of no rows (the partition clamps late boundaries to the last row, :282-284) has nothing to code: the unit codec
        // rejects it like the oracle does, in strip order with the other strips' errors
        std::vector<mic_hip_unit> units; std::vector<int> slot((size_t)ns, -1);
        for (int k = 0; k < ns; k++) {
            const int y0 = starts[(size_t)(s0 + k)], y1 = (s0 + k + 1 < actual) ? starts[(size_t)(s0 + k) + 1] : height;
            if (y1 <= y0) continue;
            slot[(size_t)k] = (int)units.size();
            units.push_back(mic_hip_unit{ (uint64_t)y0 * (uint64_t)width, width, y1 - y0, max_value, 2 });                                          // :92
            units.push_back(mic_hip_unit{ (uint64_t)y0 * (uint64_t)width, width, y1 - y0, max_value, (uint16_t)(2 | MIC_HIP_PRED_GRAD) });         // :94
        }
        const int nu = (int)units.size();
        std::vector<uint64_t> offs((size_t)nu + 1, 0); std::vector<int32_t> st((size_t)nu), nst((size_t)nu);
        const uint8_t *d_blobs = nullptr;
        if (nu) {
            if ((rc = session_encode_enqueue(s, (const uint16_t *)s->io_px.p, units.data(), nu))) return rc;
            if ((rc = session_encode_finish(s, &d_blobs, offs.data(), st.data(), nst.data()))) return rc;
        }
        for (int k = 0; k < ns; k++) {
            if (slot[(size_t)k] < 0) return MIC_ERR_ARGS;
            const size_t a = (size_t)slot[(size_t)k];
            const int32_t e1 = st[a], e2 = st[a + 1];
            const size_t la = (size_t)(offs[a + 1] - offs[a]), lg = (size_t)(offs[a + 2] - offs[a + 1]);
            int pick;
            if (e2 == MIC_OK && (e1 != MIC_OK || lg <= la)) pick = 1;                          // the smaller, or the one that succeeded (:97-105)
            else { pick = 0; if (e1 != MIC_OK) return e1; }                                    // "pica: strip %d: ..." (:110-114)
            const size_t len = pick ? lg : la;
            blob[(size_t)(s0 + k)].resize(len);
            flags[(size_t)(s0 + k)] = pick ? 1u : 0u;
            if (len) HIP_TRY(hipMemcpy(blob[(size_t)(s0 + k)].data(), d_blobs + offs[a + (size_t)pick], len, hipMemcpyDeviceToHost));
        }
        s0 = s1;
    }
    size_t total = 0;
    for (const auto &b : blob) total += b.size();
    if (total > 0xFFFFFFFFull) return MIC_ERR_UNSUPPORTED;
    if (out_cap < header + total) return MIC_ERR_CAPACITY;
    memcpy(out, "PICA", 4);
    put_u32(out + 4, (uint32_t)width); put_u32(out + 8, (uint32_t)height); put_u32(out + 12, (uint32_t)actual);
    size_t off = 0;
    for (int k = 0; k < actual; k++) {
        uint8_t *e = out + 16 + (size_t)k * 16;
        put_u32(e, (uint32_t)starts[(size_t)k]); put_u32(e + 4, (uint32_t)off); put_u32(e + 8, (uint32_t)blob[(size_t)k].size()); put_u32(e + 12, flags[(size_t)k]);
        memcpy(out + header + off, blob[(size_t)k].data(), blob[(size_t)k].size());
        off += blob[(size_t)k].size();
    }
    *out_len = header + total;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_pica_info(const uint8_t *c, size_t len, int *width, int *height, int *num_strips) try {
    if (!c) return MIC_ERR_ARGS;
    if (len < 16 || memcmp(c, "PICA", 4) != 0) return MIC_ERR_CORRUPT;                      // :142-144
    const int w = (int)get_u32(c + 4), h = (int)get_u32(c + 8), n = (int)get_u32(c + 12);
    if (n < 0 || (size_t)n > (len - 16) / 16) return MIC_ERR_CORRUPT;                       // truncated header, :150-153
    if (w <= 0 || h <= 0 || n <= 0) return MIC_ERR_CORRUPT;                                 // :154-156
    if (width) *width = w; if (height) *height = h; if (num_strips) *num_strips = n;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// DecompressParallelStripsAdaptive (parallelstripsadaptive.go:141-214)
int mic_hip_pica_decompress(const uint8_t *c, size_t len, uint16_t *pixels_out, int width, int height) try {
    if (!c || !pixels_out) return MIC_ERR_ARGS;
    int w, h, n;
    int rc = mic_hip_pica_info(c, len, &w, &h, &n);
    if (rc) return rc;
    if (w != width || h != height) return MIC_ERR_ARGS;
    const size_t header = 16 + (size_t)n * 16;
    struct Strip { long y0, y1; size_t start, end; uint32_t flags; };
    std::vector<Strip> st((size_t)n);
    for (int k = 0; k < n; k++) {
        const uint8_t *e = c + 16 + (size_t)k * 16;
        Strip &t = st[(size_t)k];
        t.y0 = (long)get_u32(e); t.y1 = (k + 1 < n) ? (long)get_u32(e + 16) : h;
        t.start = header + get_u32(e + 4); t.end = t.start + get_u32(e + 8); t.flags = get_u32(e + 12);
        if (t.end > len || t.start > t.end) return MIC_ERR_CORRUPT;                         // :186-190
        if (t.y0 < 0 || t.y1 <= t.y0 || t.y1 > h) return MIC_ERR_CORRUPT;                   // Go: make / slice panics
    }
    DefaultLease lease;
    if ((rc = lease.acquire())) return rc;
    mic_hip_session *s = cur_default();
    const size_t npx = (size_t)w * (size_t)h;
    if ((rc = s->ensure(1, 1))) return rc;
    if ((rc = s->io_px.reserve(npx * 2 + 64))) return rc;
    HIP_TRY(hipMemsetAsync(s->io_px.p, 0, npx * 2, s->stream));                              // rows no strip covers stay 0 (make([]uint16), :175)
    int k0 = 0;
    while (k0 < n) {
        size_t max_px = 0, comp = 0; int k1 = k0;
        while (k1 < n) {
            const size_t px = (size_t)(st[(size_t)k1].y1 - st[(size_t)k1].y0) * (size_t)w;
            const size_t mp = std::max(max_px, px);
            if (mp > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
            if (k1 > k0 && unit_ws_bytes(mp) * (size_t)(k1 - k0 + 1) > kWorkspaceBudget) break;
            max_px = mp; comp += st[(size_t)k1].end - st[(size_t)k1].start; k1++;
        }
        const int ns = k1 - k0;
        if ((rc = s->io_comp.reserve(comp + 64))) return rc;
        std::vector<mic_hip_unit> units((size_t)ns); std::vector<uint64_t> offs((size_t)ns + 1, 0);
        for (int k = 0; k < ns; k++) {
            const Strip &t = st[(size_t)(k0 + k)];
            if (t.end == t.start) return MIC_ERR_CORRUPT;
            HIP_TRY(hipMemcpyAsync((uint8_t *)s->io_comp.p + offs[(size_t)k], c + t.start, t.end - t.start, hipMemcpyHostToDevice, s->stream));
            offs[(size_t)k + 1] = offs[(size_t)k] + (t.end - t.start);
            units[(size_t)k] = mic_hip_unit{ (uint64_t)t.y0 * (uint64_t)w, w, (int)(t.y1 - t.y0), 0,
                                            (uint16_t)(2 | ((t.flags & 1u) ? MIC_HIP_PRED_GRAD : 0)) };   // picaFlagGradPredictor, :198-202
        }
        if ((rc = session_decode_enqueue(s, (const uint8_t *)s->io_comp.p, offs.data(), units.data(), ns, (uint16_t *)s->io_px.p))) return rc;
        std::vector<int32_t> stt((size_t)ns);
        if ((rc = session_decode_finish(s, stt.data()))) return rc;
        for (int32_t v : stt) if (v != MIC_OK) return v;
        k0 = k1;
    }
    HIP_TRY(hipMemcpy(pixels_out, s->io_px.p, npx * 2, hipMemcpyDeviceToHost));
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

}  // extern "C"
