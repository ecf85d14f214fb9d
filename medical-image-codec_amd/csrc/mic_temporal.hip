// mic_temporal.hip -- MIC2 with the temporal pipeline (flags 0x01 | 0x02).
//
// Reference: CompressMultiFrame(..., temporal=true) (multiframecompress.go:179-224): frame 0 is a
// spatial frame; frame i > 0 is TemporalDeltaEncode(frame i, frame i-1) = ZigZag(cur - prev)
// (temporaldelta.go:11-23) coded by compressResidualFrame = RleCompressU16 + FSE 2-state with the
// 1-state fallback (multiframecompress.go:146-163).  DecompressMultiFrame (:227-261) inverts it frame by
// frame.  Both directions are parallel over frames here: the encoder differences ORIGINAL frames, and on the
// decode side  frame_i = frame_0 + sum_{j<=i} UnZigZag(res_j)  (mod 2^16) is a running sum along the frame
// axis, so every residual stream is entropy-decoded at once and one kernel accumulates per pixel.
#include <cstring>
#include <vector>
#include "../../include/mic_hip.h"
#include "mic_session.h"
#include "mic_launch.h"

using namespace micapi;

namespace {

__device__ __forceinline__ uint32_t zigzag16(int32_t v) { const uint32_t x = (uint32_t)v & 0xFFFFu; return ((x << 1) ^ ((x & 0x8000u) ? 0xFFFFu : 0u)) & 0xFFFFu; }   // deltazigzagcompressu16.go:108-111
__device__ __forceinline__ uint32_t unzigzag16(uint32_t u) { return ((u >> 1) ^ ((u & 1u) ? 0xFFFFu : 0u)) & 0xFFFFu; }                                            // :113-116

// residual symbols of the units r0 .. n-1 into their symbol slabs, their maxima into dec_thr (free on the encode side).  first = the frame
// of unit 0 on the device; the frame in front of it (unit 0's reference when r0 = 0) lies directly before it.
__global__ void __launch_bounds__(256) k_tmp_residual(MicUnit *units, const uint16_t *first, uint32_t npx, int r0) {
    const int i = (int)blockIdx.y + r0;
    MicUnit &u = units[i];
    const uint16_t *cur = first + (size_t)i * npx, *prev = cur - npx;
    uint32_t m = 0;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < npx; k += gridDim.x * blockDim.x) {
        const uint32_t r = zigzag16((int32_t)cur[k] - (int32_t)prev[k]);
        u.sym[k] = (uint16_t)r;
        m = max(m, r);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&u.dec_thr, m);
}
__global__ void k_tmp_set_max(MicUnit *units, int n, int r0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + r0;
    if (i < n) units[i].max_value = (uint16_t)units[i].dec_thr;
}

// frame_i = frame_{i-1} + UnZigZag(res_i), all frames of one pixel by one thread.  Unit i's frame lives in slot lead + i of `frames`;
// the running sum starts from slot lead + r0 - 1 (the sub-batch's own spatial frame, or the frame carried from the sub-batch before).
__global__ void __launch_bounds__(256) k_tmp_accumulate(MicUnit *units, int n, uint16_t *frames, uint32_t npx, int r0, int lead) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < npx; k += gridDim.x * blockDim.x) {
        uint32_t acc = frames[(size_t)(lead + r0 - 1) * npx + k];
        for (int i = r0; i < n; i++) {
            acc = (acc + unzigzag16(units[i].sym[k])) & 0xFFFFu;        // temporaldelta.go:27-37
            frames[(size_t)(lead + i) * npx + k] = (uint16_t)acc;
        }
    }
}
// residual streams must expand to exactly one frame (multiframecompress.go:170-172)
__global__ void k_tmp_check(MicUnit *units, int n, uint32_t npx, int r0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + r0;
    if (i < n && units[i].status == MICD_OK && units[i].nsym != npx) units[i].status = MICD_ERR_CORRUPT;
}

inline void put_u32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
inline uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace

void mic_launch_rle_expand(MicUnit *d_units, int n, hipStream_t stream, int mode_filter);   // mic_wavelet.hip

namespace micapi {

// Frames go through in sub-batches under the workspace ceiling.  Encode: a residual needs the ORIGINAL frame before it, so a sub-batch
// uploads one frame more than it codes (its predecessor's last) and is otherwise independent of the others.  Decode: the running sum
// needs the frame before the sub-batch's first residual: it is carried on the device in the slot in front of the sub-batch's frames.
int mic2_temporal_compress(const uint16_t *frames, int width, int height, int nframes, uint16_t max_value,
                           uint8_t *out, size_t out_cap, size_t *out_len) {
    const size_t npx = (size_t)width * (size_t)height;
    if (npx > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
    const size_t header = 20 + (size_t)nframes * 8;
    if (out_cap < header) return MIC_ERR_CAPACITY;
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    mic_hip_session *s = cur_default();
    const size_t per = std::max<size_t>(1, std::min<size_t>(kWorkspaceBudget / (unit_ws_bytes(npx) + 2 * npx), 65535));
    std::vector<uint32_t> lens((size_t)nframes);
    uint64_t total = 0;
    memset(out, 0, header);
    for (size_t f0 = 0; f0 < (size_t)nframes; f0 += per) {
        const int nb = (int)std::min(per, (size_t)nframes - f0);
        const int lead = f0 ? 1 : 0;                                      // the frame in front of the sub-batch (its residuals' reference)
        if ((rc = s->io_px.reserve(npx * 2 * (size_t)(nb + lead)))) return rc;
        if ((rc = s->ensure(nb, npx))) return rc;
        HIP_TRY(hipMemcpyAsync(s->io_px.p, frames + (f0 - (size_t)lead) * npx, npx * 2 * (size_t)(nb + lead), hipMemcpyHostToDevice, s->stream));
        const uint16_t *d_first = (const uint16_t *)s->io_px.p + (size_t)lead * npx;   // frame f0 on the device
        { const int arc = s->h_units.assign((size_t)nb, MicUnit{}); if (arc) return arc; }
        for (int i = 0; i < nb; i++) {
            MicUnit &u = s->h_units[(size_t)i];
            u.w = width; u.h = height; u.nstates = 2;
            s->fill_workspace(u, i);
            u.tok_cap = (uint32_t)tok_cap_for(npx);
            if (f0 == 0 && i == 0) { u.mode = 0; u.px_in = d_first; u.max_value = max_value; }
            else { u.mode = 2; u.nsym = (uint32_t)npx; u.max_value = 0; }
        }
        { const int urc = s->h_units.upload(s->units.p, (size_t)nb, s->stream); if (urc) return urc; }
        if ((rc = s->prepare_hist(nb))) return rc;
        const int r0 = f0 ? 0 : 1;                                        // first residual unit of the sub-batch
        if (nb > r0) {
            const unsigned bx = (unsigned)std::min<size_t>((npx + 255) / 256, 1024);
            hipLaunchKernelGGL(k_tmp_residual, dim3(bx, (unsigned)(nb - r0)), dim3(256), 0, s->stream,
                               (MicUnit *)s->units.p, d_first, (uint32_t)npx, r0);
            hipLaunchKernelGGL(k_tmp_set_max, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s->stream, (MicUnit *)s->units.p, nb, r0);
        }
        s->timer.reset(s->stream);
        mic_launch_encode((MicUnit *)s->units.p, nb, s->stream, s->variant, nullptr);
        if (hipGetLastError() != hipSuccess) { s->hist_unknown(); return MIC_ERR_DEVICE; }
        s->begin_chain(nb);
        std::vector<uint64_t> offs((size_t)nb + 1);
        std::vector<int32_t> st((size_t)nb), ns((size_t)nb);
        const uint8_t *d_blobs = nullptr;
        if ((rc = session_encode_finish(s, &d_blobs, offs.data(), st.data(), ns.data()))) return rc;
        for (int i = 0; i < nb; i++) if (st[(size_t)i] != MIC_OK) return st[(size_t)i];
        const uint64_t bytes = offs[(size_t)nb];
        if (total + bytes > 0xFFFFFFFFull) return MIC_ERR_UNSUPPORTED;    // u32 offsets, multiframe.go:75-80
        if (out_cap < header + total + bytes) return MIC_ERR_CAPACITY;
        for (int i = 0; i < nb; i++) {
            put_u32(out + 20 + (f0 + (size_t)i) * 8, (uint32_t)(total + offs[(size_t)i]));
            put_u32(out + 24 + (f0 + (size_t)i) * 8, (uint32_t)(offs[(size_t)i + 1] - offs[(size_t)i]));
        }
        if (bytes) HIP_TRY(hipMemcpy(out + header + total, d_blobs, (size_t)bytes, hipMemcpyDeviceToHost));
        total += bytes;
    }
    memcpy(out, "MIC2", 4);
    put_u32(out + 4, (uint32_t)width); put_u32(out + 8, (uint32_t)height); put_u32(out + 12, (uint32_t)nframes);
    out[16] = 0x01 | 0x02;                                              // PipelineSpatial | PipelineTemporal, multiframe.go:28-29
    *out_len = header + (size_t)total;
    return MIC_OK;
}

int mic2_temporal_decompress(const uint8_t *c, size_t len, int w, int h, int n_total, int n, uint16_t *frames_out) {
    const size_t npx = (size_t)w * (size_t)h;
    if (npx > ((size_t)1 << 28) || len > 0xFFFFFFF0ull) return MIC_ERR_UNSUPPORTED;
    const size_t data_off = 20 + (size_t)n_total * 8;                      // decodes frames 0 .. n-1 of n_total
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    mic_hip_session *s = cur_default();
    const size_t per = std::max<size_t>(1, std::min<size_t>(kWorkspaceBudget / (unit_ws_bytes(npx) + 2 * npx), 65535));
    for (size_t f0 = 0; f0 < (size_t)n; f0 += per) {
        const int nb = (int)std::min(per, (size_t)n - f0);
        const int lead = f0 ? 1 : 0;                                      // slot 0 holds the frame in front of the sub-batch
        if ((rc = s->io_px.reserve(npx * 2 * (size_t)(nb + 1)))) return rc;   // (grown before the carry below could be lost: sized for the first pass too)
        if ((rc = s->ensure(nb, npx))) return rc;
        size_t c0 = (size_t)-1, c1 = 0;                                   // byte range of the sub-batch's streams
        { const int arc = s->h_units.assign((size_t)nb, MicUnit{}); if (arc) return arc; }
        for (int i = 0; i < nb; i++) {
            const size_t fi = f0 + (size_t)i;
            const size_t start = data_off + get_u32(c + 20 + fi * 8), bl = get_u32(c + 24 + fi * 8);
            if (start + bl > len) return MIC_ERR_CORRUPT;                 // multiframe.go:137-139
            if (bl == 0) return MIC_ERR_CORRUPT;
            c0 = std::min(c0, start); c1 = std::max(c1, start + bl);
        }
        if ((rc = s->io_comp.reserve(c1 - c0 + 64))) return rc;
        for (int i = 0; i < nb; i++) {
            const size_t fi = f0 + (size_t)i;
            const size_t start = data_off + get_u32(c + 20 + fi * 8), bl = get_u32(c + 24 + fi * 8);
            MicUnit &u = s->h_units[(size_t)i];
            u.comp_in = (const uint8_t *)s->io_comp.p + (start - c0); u.comp_len = (uint32_t)bl;
            u.w = w; u.h = h;
            s->fill_workspace(u, i);
            u.tok_cap = (uint32_t)tok_cap_for(npx);
            if (fi == 0) { u.mode = 0; u.px_out = (uint16_t *)s->io_px.p; }
            else u.mode = 3;                                             // FSE + RLE-of-symbols into u.sym
        }
        HIP_TRY(hipMemcpyAsync(s->io_comp.p, c + c0, c1 - c0, hipMemcpyHostToDevice, s->stream));
        { const int urc = s->h_units.upload(s->units.p, (size_t)nb, s->stream); if (urc) return urc; }
        HIP_TRY(hipMemsetAsync(s->flags.p, 0, s->flag_stride * (size_t)nb, s->stream));
        s->timer.reset(s->stream);
        mic_launch_decode((MicUnit *)s->units.p, nb, s->stream, s->variant, nullptr, (int *)s->cls.p);
        const int r0 = f0 ? 0 : 1;                                        // first residual unit
        if (nb > r0) {
            mic_launch_rle_expand((MicUnit *)s->units.p, nb, s->stream, 3);
            hipLaunchKernelGGL(k_tmp_check, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s->stream, (MicUnit *)s->units.p, nb, (uint32_t)npx, r0);
        }
        HIP_TRY(hipGetLastError());
        s->begin_chain(nb);
        std::vector<int32_t> st((size_t)nb);
        if ((rc = session_decode_finish(s, st.data()))) return rc;
        for (int i = 0; i < nb; i++) if (st[(size_t)i] != MIC_OK) return st[(size_t)i];
        // frames of the sub-batch at slots lead .. lead + nb - 1; slot 0 = frame f0 - 1 (carried) when lead
        uint16_t *d_frames = (uint16_t *)s->io_px.p;
        if (nb > r0) {
            const unsigned bx = (unsigned)std::min<size_t>((npx + 255) / 256, 4096);
            hipLaunchKernelGGL(k_tmp_accumulate, dim3(bx), dim3(256), 0, s->stream, (MicUnit *)s->units.p, nb, d_frames, (uint32_t)npx, r0, lead);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(frames_out + f0 * npx, d_frames + (size_t)lead * npx, npx * 2 * (size_t)nb, hipMemcpyDeviceToHost, s->stream));
        if (f0 + (size_t)nb < (size_t)n && lead + nb - 1 != 0)            // carry the last frame to slot 0 for the next sub-batch
            HIP_TRY(hipMemcpyAsync(d_frames, d_frames + (size_t)(lead + nb - 1) * npx, npx * 2, hipMemcpyDeviceToDevice, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return MIC_OK;
}

}  // namespace micapi
