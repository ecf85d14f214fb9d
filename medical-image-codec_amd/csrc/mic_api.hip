// mic_api.hip -- host side of libmic_hip.so: the C ABI of include/mic_hip.h, the HIP batch
// launcher that replaces the reference's goroutine / pthread fan-out
// (parallelstrips.go:77-93, mic_parallel.c:146-181, wsicompress.go:126-145) and the
// PICS / MIC2 container assembly (parallelstrips.go:101-123, multiframe.go:49-91).
//
// There is no CPU codec in this file: every pixel and every compressed byte is produced by
// the kernels in mic_encode.hip / mic_decode.hip.  The host only moves buffers, fills unit
// descriptors and writes container headers.
#include "mic_session.h"

namespace micapi {

std::mutex g_mu;
int g_device = 0;
bool g_device_ok = false;
std::string g_device_name;
mic_hip_session g_default;   // guarded by g_mu

// gfx950 or nothing: the code objects are built for that target only.  One answer per device, remembered.
int check_device(int device) {
    static std::mutex mu; static std::vector<int8_t> known;              // 0 unknown, 1 ok, -1 not usable
    std::lock_guard<std::mutex> lk(mu);
    int n = 0;
    if (device < 0 || hipGetDeviceCount(&n) != hipSuccess || device >= n) return MIC_ERR_DEVICE;
    if (known.size() < (size_t)n) known.resize((size_t)n, 0);
    if (known[(size_t)device] == 0) {
        hipDeviceProp_t p;
        known[(size_t)device] = (hipGetDeviceProperties(&p, device) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ? 1 : -1;
    }
    return known[(size_t)device] == 1 ? MIC_OK : MIC_ERR_DEVICE;
}

int ensure_device() {
    if (g_device_ok) { return hipSetDevice(g_device) == hipSuccess ? MIC_OK : MIC_ERR_DEVICE; }
    int rc = check_device(g_device);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(g_device));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, g_device));
    char buf[256];
    snprintf(buf, sizeof buf, "%s %d CUs %.0f GiB (%s)", p.gcnArchName, p.multiProcessorCount,
             (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), p.name);
    g_device_name = buf;
    g_device_ok = true;
    g_default.device = g_device;
    return MIC_OK;
}

// ---- encode -------------------------------------------------------------------------------
int session_encode_enqueue(mic_hip_session *s, const uint16_t *d_pixels, const mic_hip_unit *units, int n) {
    if (n <= 0) return MIC_ERR_ARGS;
    size_t max_px = 0;
    for (int i = 0; i < n; i++) {
        if (units[i].width <= 0 || units[i].height <= 0) return MIC_ERR_ARGS;
        max_px = std::max(max_px, (size_t)units[i].width * (size_t)units[i].height);
    }
    if (max_px > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
    int rc = s->ensure(n, max_px);
    if (rc) return rc;
    s->h_units.assign((size_t)n, MicUnit{});
    bool any_grad = false;
    for (int i = 0; i < n; i++) {
        MicUnit &u = s->h_units[(size_t)i];
        u.px_in = d_pixels + units[i].px_offset;
        u.w = units[i].width; u.h = units[i].height;
        u.max_value = units[i].max_value; u.nstates = units[i].nstates & 0xFF;
        u.pred = (units[i].nstates & MIC_HIP_PRED_GRAD) ? 1u : 0u; any_grad |= u.pred != 0;
        s->fill_workspace(u, i);
        u.tok_cap = (uint32_t)tok_cap_for((size_t)u.w * (size_t)u.h);
        if (!(u.nstates == 2 || u.nstates == 4 || u.nstates == 8) || (units[i].nstates & ~(0xFF | MIC_HIP_PRED_GRAD))) return MIC_ERR_ARGS;
    }
    HIP_TRY(hipMemcpyAsync(s->units.p, s->h_units.data(), sizeof(MicUnit) * (size_t)n, hipMemcpyHostToDevice, s->stream));
    if ((rc = s->prepare_hist(n))) return rc;
    s->timer.reset(s->stream);
    mic_launch_encode((MicUnit *)s->units.p, n, s->stream, s->variant | (any_grad ? MIC_VARIANT_GRAD : 0), &s->timer);
    if (hipGetLastError() != hipSuccess) { s->hist_unknown(); return MIC_ERR_DEVICE; }
    s->n_last = n;
    return MIC_OK;
}

int session_encode_finish(mic_hip_session *s, const uint8_t **d_blobs, uint64_t *h_offsets, int32_t *h_status, int32_t *h_nstates) {
    int n = s->n_last;
    if (n <= 0) return MIC_ERR_ARGS;
    HIP_TRY(hipMemcpyAsync(s->h_units.data(), s->units.p, sizeof(MicUnit) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    uint64_t total = 0;
    for (int i = 0; i < n; i++) {
        const MicUnit &u = s->h_units[(size_t)i];
        h_offsets[i] = total;
        if (u.status == MICD_OK) total += u.blob_len;
        h_status[i] = u.status;
        if (h_nstates) h_nstates[i] = u.nstates_used;
    }
    h_offsets[n] = total;
    int rc = s->packed.reserve((size_t)total + 16);
    if (rc) return rc;
    mic_launch_pack((const MicUnit *)s->units.p, n, (uint64_t *)s->offsets.p, (uint8_t *)s->packed.p, s->stream, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (d_blobs) *d_blobs = (const uint8_t *)s->packed.p;
    return MIC_OK;
}

// ---- decode -------------------------------------------------------------------------------
int session_decode_enqueue(mic_hip_session *s, const uint8_t *d_blobs, const uint64_t *h_offsets,
                           const mic_hip_unit *units, int n, uint16_t *d_pixels_out) {
    if (n <= 0) return MIC_ERR_ARGS;
    return session_decode_enqueue_spans(s, d_blobs, h_offsets, h_offsets + 1, units, n, d_pixels_out);
}

int session_decode_enqueue_spans(mic_hip_session *s, const uint8_t *d_blobs, const uint64_t *begins, const uint64_t *ends,
                                 const mic_hip_unit *units, int n, uint16_t *d_pixels_out) {
    if (n <= 0) return MIC_ERR_ARGS;
    size_t max_px = 0;
    for (int i = 0; i < n; i++) {
        if (units[i].width <= 0 || units[i].height <= 0) return MIC_ERR_ARGS;
        max_px = std::max(max_px, (size_t)units[i].width * (size_t)units[i].height);
    }
    if (max_px > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
    int rc = s->ensure(n, max_px);
    if (rc) return rc;
    s->h_units.assign((size_t)n, MicUnit{});
    bool any_grad = false;
    for (int i = 0; i < n; i++) {
        MicUnit &u = s->h_units[(size_t)i];
        uint64_t len = ends[i] - begins[i];
        if (ends[i] < begins[i] || len > 0xFFFFFFF0ull) return MIC_ERR_ARGS;
        u.comp_in = d_blobs + begins[i]; u.comp_len = (uint32_t)len;
        u.px_out = d_pixels_out + units[i].px_offset;
        u.w = units[i].width; u.h = units[i].height;
        u.pred = (units[i].nstates & MIC_HIP_PRED_GRAD) ? 1u : 0u; any_grad |= u.pred != 0;
        s->fill_workspace(u, i);
        u.tok_cap = (uint32_t)tok_cap_for((size_t)u.w * (size_t)u.h);
    }
    HIP_TRY(hipMemcpyAsync(s->units.p, s->h_units.data(), sizeof(MicUnit) * (size_t)n, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemsetAsync(s->flags.p, 0, s->flag_stride * (size_t)n, s->stream));
    s->timer.reset(s->stream);
    mic_launch_decode((MicUnit *)s->units.p, n, s->stream, s->variant | (any_grad ? MIC_VARIANT_GRAD : 0), &s->timer, (int *)s->cls.p);
    HIP_TRY(hipGetLastError());
    s->n_last = n;
    return MIC_OK;
}

int session_decode_finish(mic_hip_session *s, int32_t *h_status) {
    int n = s->n_last;
    if (n <= 0) return MIC_ERR_ARGS;
    HIP_TRY(hipMemcpyAsync(s->h_units.data(), s->units.p, sizeof(MicUnit) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (int i = 0; i < n; i++) h_status[i] = s->h_units[(size_t)i].status;
    return MIC_OK;
}

// ---- host-pointer batches on the default session ---------------------------------------------
// Units are processed in sub-batches that keep the workspace under a fixed budget.

size_t unit_ws_bytes(size_t px) { return tok_cap_for(px) * 4 + blob_cap_for(px) + (2 * px + 8) * 8 + px / 8 + kSym * 4 * 6 + kSym * 2 + 8192; }

int compress_batch_locked(mic_hip_enc_job *jobs, int njobs) {
    mic_hip_session *s = &g_default;
    int i0 = 0;
    while (i0 < njobs) {
        size_t max_px = 0, tot_px = 0; int i1 = i0;
        while (i1 < njobs) {
            mic_hip_enc_job &j = jobs[i1];
            if (!j.pixels || !j.out || j.width <= 0 || j.height <= 0 || (size_t)j.width * (size_t)j.height > ((size_t)1 << 28) ||
                !(j.nstates == 2 || j.nstates == 4 || j.nstates == 8)) {
                if (i1 == i0) { j.status = MIC_ERR_ARGS; j.out_len = 0; j.nstates_used = 0; i0++; i1++; continue; }
                break;
            }
            size_t px = (size_t)j.width * (size_t)j.height;
            size_t mp = std::max(max_px, px);
            if (i1 > i0 && unit_ws_bytes(mp) * (size_t)(i1 - i0 + 1) > kWorkspaceBudget) break;
            max_px = mp; tot_px += px; i1++;
        }
        int n = i1 - i0;
        if (n <= 0) continue;
        int rc = s->io_px.reserve(tot_px * 2);
        if (rc) return rc;
        std::vector<mic_hip_unit> units((size_t)n);
        size_t off = 0;
        for (int k = 0; k < n; k++) {
            mic_hip_enc_job &j = jobs[i0 + k];
            size_t px = (size_t)j.width * (size_t)j.height;
            HIP_TRY(hipMemcpyAsync((uint16_t *)s->io_px.p + off, j.pixels, px * 2, hipMemcpyHostToDevice, s->stream ? s->stream : 0));
            units[(size_t)k] = mic_hip_unit{ off, j.width, j.height, j.max_value, j.nstates };
            off += px;
        }
        if (!s->stream) HIP_TRY(hipDeviceSynchronize());
        rc = session_encode_enqueue(s, (const uint16_t *)s->io_px.p, units.data(), n);
        if (rc) return rc;
        std::vector<uint64_t> offs((size_t)n + 1); std::vector<int32_t> st((size_t)n), ns((size_t)n);
        const uint8_t *d_blobs = nullptr;
        rc = session_encode_finish(s, &d_blobs, offs.data(), st.data(), ns.data());
        if (rc) return rc;
        std::vector<uint8_t> host((size_t)offs[(size_t)n] + 16);
        if (offs[(size_t)n]) HIP_TRY(hipMemcpy(host.data(), d_blobs, (size_t)offs[(size_t)n], hipMemcpyDeviceToHost));
        for (int k = 0; k < n; k++) {
            mic_hip_enc_job &j = jobs[i0 + k];
            j.status = st[(size_t)k]; j.nstates_used = ns[(size_t)k]; j.out_len = 0;
            if (j.status != MIC_OK) continue;
            size_t len = (size_t)(offs[(size_t)k + 1] - offs[(size_t)k]);
            if (len > j.out_cap) { j.status = MIC_ERR_CAPACITY; continue; }
            memcpy(j.out, host.data() + offs[(size_t)k], len);
            j.out_len = len;
        }
        i0 = i1;
    }
    return MIC_OK;
}

int decompress_batch_locked(mic_hip_dec_job *jobs, int njobs) {
    mic_hip_session *s = &g_default;
    int i0 = 0;
    while (i0 < njobs) {
        size_t max_px = 0, tot_px = 0, tot_comp = 0; int i1 = i0;
        while (i1 < njobs) {
            mic_hip_dec_job &j = jobs[i1];
            if (!j.compressed || !j.pixels_out || j.width <= 0 || j.height <= 0 || j.compressed_len == 0 ||
                j.compressed_len > 0xFFFFFFF0ull || (size_t)j.width * (size_t)j.height > ((size_t)1 << 28)) {
                if (i1 == i0) { j.status = (j.compressed && j.compressed_len == 0) ? MIC_ERR_CORRUPT : MIC_ERR_ARGS; i0++; i1++; continue; }
                break;
            }
            size_t px = (size_t)j.width * (size_t)j.height;
            size_t mp = std::max(max_px, px);
            if (i1 > i0 && unit_ws_bytes(mp) * (size_t)(i1 - i0 + 1) > kWorkspaceBudget) break;
            max_px = mp; tot_px += px; tot_comp += align_up(j.compressed_len, 16); i1++;
        }
        int n = i1 - i0;
        if (n <= 0) continue;
        int rc = s->io_px.reserve(tot_px * 2);
        if (rc) return rc;
        if ((rc = s->io_comp.reserve(tot_comp + 64))) return rc;
        if ((rc = s->ensure(n, max_px))) return rc;
        std::vector<mic_hip_unit> units((size_t)n);
        std::vector<uint64_t> offs((size_t)n + 1);
        // blobs are placed 16-byte aligned; decode takes explicit [begin,end) per unit
        std::vector<uint64_t> begins((size_t)n), ends((size_t)n);
        size_t poff = 0, coff = 0;
        for (int k = 0; k < n; k++) {
            mic_hip_dec_job &j = jobs[i0 + k];
            HIP_TRY(hipMemcpyAsync((uint8_t *)s->io_comp.p + coff, j.compressed, j.compressed_len, hipMemcpyHostToDevice, s->stream));
            begins[(size_t)k] = coff; ends[(size_t)k] = coff + j.compressed_len;
            coff += align_up(j.compressed_len, 16);
            units[(size_t)k] = mic_hip_unit{ poff, j.width, j.height, 0, 0 };
            poff += (size_t)j.width * (size_t)j.height;
        }
        // session_decode_enqueue wants contiguous offsets; fill the descriptors directly instead
        s->h_units.assign((size_t)n, MicUnit{});
        for (int k = 0; k < n; k++) {
            MicUnit &u = s->h_units[(size_t)k];
            u.comp_in = (const uint8_t *)s->io_comp.p + begins[(size_t)k];
            u.comp_len = (uint32_t)(ends[(size_t)k] - begins[(size_t)k]);
            u.px_out = (uint16_t *)s->io_px.p + units[(size_t)k].px_offset;
            u.w = units[(size_t)k].width; u.h = units[(size_t)k].height;
            s->fill_workspace(u, k);
            u.tok_cap = (uint32_t)tok_cap_for((size_t)u.w * (size_t)u.h);
        }
        HIP_TRY(hipMemcpyAsync(s->units.p, s->h_units.data(), sizeof(MicUnit) * (size_t)n, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemsetAsync(s->flags.p, 0, s->flag_stride * (size_t)n, s->stream));
        mic_launch_decode((MicUnit *)s->units.p, n, s->stream, s->variant, nullptr, (int *)s->cls.p);
        HIP_TRY(hipGetLastError());
        s->n_last = n;
        std::vector<int32_t> st((size_t)n);
        rc = session_decode_finish(s, st.data());
        if (rc) return rc;
        for (int k = 0; k < n; k++) {
            mic_hip_dec_job &j = jobs[i0 + k];
            j.status = st[(size_t)k];
            if (j.status != MIC_OK) continue;
            HIP_TRY(hipMemcpyAsync(j.pixels_out, (uint16_t *)s->io_px.p + units[(size_t)k].px_offset,
                                   (size_t)j.width * (size_t)j.height * 2, hipMemcpyDeviceToHost, s->stream));
        }
        HIP_TRY(hipStreamSynchronize(s->stream));
        i0 = i1;
    }
    return MIC_OK;
}

void put_u32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace micapi

// ================================================================================ C ABI

extern "C" {

const char *mic_hip_version(void) { return "mic-hip 0.2 (gfx950)"; }

// device -> device copy on the calling thread's current device, complete on return (a session's result buffers are reused by its
// next call: a caller that keeps them copies them out)
int mic_hip_device_copy(void *d_dst, const void *d_src, size_t bytes) {
    if ((!d_dst || !d_src) && bytes) return MIC_ERR_ARGS;
    if (bytes) HIP_TRY(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice));
    return MIC_OK;
}

int mic_hip_set_device(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (device < 0) return MIC_ERR_ARGS;
    if (g_device_ok && device != g_device) { (void)g_default.activate(); g_default.release(); g_default = mic_hip_session(); g_device_ok = false; }
    g_device = device;
    return ensure_device();
}

const char *mic_hip_device_name(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (ensure_device() != MIC_OK) return "";
    return g_device_name.c_str();
}

int mic_hip_compress_batch(mic_hip_enc_job *jobs, int njobs) {
    if (!jobs || njobs < 0) return MIC_ERR_ARGS;
    if (njobs == 0) return MIC_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_device();
    if (rc) return rc;
    if (!g_default.stream) HIP_TRY(hipStreamCreate(&g_default.stream));
    return compress_batch_locked(jobs, njobs);
}

int mic_hip_decompress_batch(mic_hip_dec_job *jobs, int njobs) {
    if (!jobs || njobs < 0) return MIC_ERR_ARGS;
    if (njobs == 0) return MIC_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_device();
    if (rc) return rc;
    if (!g_default.stream) HIP_TRY(hipStreamCreate(&g_default.stream));
    return decompress_batch_locked(jobs, njobs);
}

int mic_hip_compress_frame(const uint16_t *pixels, int width, int height, uint16_t max_value, int nstates,
                           uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!pixels || !out || !out_len || width <= 0 || height <= 0) return MIC_ERR_ARGS;
    mic_hip_enc_job j{};
    j.pixels = pixels; j.width = width; j.height = height; j.max_value = max_value; j.nstates = (uint16_t)nstates;
    j.out = out; j.out_cap = out_cap;
    if (!(nstates == 2 || nstates == 4 || nstates == 8)) return MIC_ERR_ARGS;
    int rc = mic_hip_compress_batch(&j, 1);
    if (rc) return rc;
    if (j.status == MIC_OK) *out_len = j.out_len;
    return j.status;
}

int mic_hip_decompress_frame(const uint8_t *compressed, size_t compressed_len, uint16_t *pixels_out, int width, int height) {
    if (!compressed || !pixels_out || width <= 0 || height <= 0) return MIC_ERR_ARGS;
    mic_hip_dec_job j{};
    j.compressed = compressed; j.compressed_len = compressed_len; j.pixels_out = pixels_out; j.width = width; j.height = height;
    int rc = mic_hip_decompress_batch(&j, 1);
    if (rc) return rc;
    return j.status;
}

// ---- bare FSE stage (fsecompressu16.go:19, fse2state.go:22/102, fse4state.go:24, fse8state.go:31, rans8state.go:31)
int mic_hip_fse_compress_u16(const uint16_t *symbols, size_t n, int flavour, uint8_t *out, size_t out_cap, size_t *out_len) {
    return mic_hip_fse_compress_u16_ex(symbols, n, flavour, 0, out, out_cap, out_len);
}

int mic_hip_fse_compress_u16_ex(const uint16_t *symbols, size_t n, int flavour, int table_log, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!symbols || !out || !out_len) return MIC_ERR_ARGS;
    if (!(flavour == 1 || flavour == 2 || flavour == 4 || flavour == 8 || flavour == 108)) return MIC_ERR_ARGS;
    if (table_log < 0 || table_log > MIC_MAX_TABLELOG) return MIC_ERR_ARGS;   // prepare(): "tableLog (%d) > maxTableLog (%d)", fseu16.go:136-138
    if (n <= 1) return MIC_ERR_INCOMPRESSIBLE;                          // first gate of every FSECompressU16* variant
    if (n > ((size_t)1 << 30)) return MIC_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_device();
    if (rc) return rc;
    mic_hip_session *s = &g_default;
    const size_t px = (n + 3) / 4 + 16;
    if ((rc = s->ensure(1, px))) return rc;
    if ((rc = s->io_px.reserve(n * 2 + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(s->io_px.p, symbols, n * 2, hipMemcpyHostToDevice, s->stream));
    s->h_units.assign(1, MicUnit{});
    MicUnit &u = s->h_units[0];
    u.px_in = (const uint16_t *)s->io_px.p; u.w = (int32_t)n; u.h = 1; u.max_value = 0;
    u.nstates = (uint16_t)flavour; u.mode = 1; u.no_fallback = 1; u.req_tl = (uint32_t)table_log;
    s->fill_workspace(u, 0);
    HIP_TRY(hipMemcpyAsync(s->units.p, s->h_units.data(), sizeof(MicUnit), hipMemcpyHostToDevice, s->stream));
    if ((rc = s->prepare_hist(1))) return rc;
    mic_launch_encode((MicUnit *)s->units.p, 1, s->stream, s->variant, nullptr);
    if (hipGetLastError() != hipSuccess) { s->hist_unknown(); return MIC_ERR_DEVICE; }
    s->n_last = 1;
    uint64_t offs[2]; int32_t st = 0, ns = 0; const uint8_t *d_blobs = nullptr;
    if ((rc = session_encode_finish(s, &d_blobs, offs, &st, &ns))) return rc;
    if (st != MIC_OK) return st;
    const size_t len = (size_t)offs[1];
    if (len > out_cap) return MIC_ERR_CAPACITY;
    HIP_TRY(hipMemcpy(out, d_blobs, len, hipMemcpyDeviceToHost));
    *out_len = len;
    return MIC_OK;
}

int mic_hip_fse_decompress_u16_auto(const uint8_t *in, size_t in_len, uint16_t *out, size_t out_cap, size_t *out_n) {
    return mic_hip_fse_decompress_u16_ex(in, in_len, 0, out, out_cap, out_n);
}

// ScratchU16.DecompressLimit (fseu16.go:87-91): the reference compares len(OutU16) with the limit every time its 65536-symbol ring
// wraps (fse2state.go:249/283, fse4state.go:246/..., fse8state.go, fsedecompressu16.go:318/353) and, for 1-state streams, once more
// at the end (fsedecompressu16.go:372): an N-state stream of `count` symbols fails iff floor(count / 65536) * 65536 >= limit,
// a 1-state stream iff its symbol count >= limit.  0 = the default, 2 GiB - 1.
int mic_hip_fse_decompress_u16_ex(const uint8_t *in, size_t in_len, int64_t decompress_limit, uint16_t *out, size_t out_cap, size_t *out_n) {
    if (decompress_limit < 0) return MIC_ERR_ARGS;
    const uint64_t limit = decompress_limit ? (uint64_t)decompress_limit : ((2ull << 30) - 1);
    if (in && in_len >= 6 && in[0] == 0xFF && (in[1] == 0x02 || in[1] == 0x04 || in[1] == 0x84 || in[1] == 0x08)) {
        const uint64_t count = (uint64_t)in[2] | ((uint64_t)in[3] << 8) | ((uint64_t)in[4] << 16) | ((uint64_t)in[5] << 24);
        if (count >= 65536 && (count / 65536) * 65536 >= limit) return MIC_ERR_CAPACITY;   // "output size (%d) > DecompressLimit (%d)"
    }
    if (!in || !out || !out_n || in_len == 0) return in && in_len == 0 ? MIC_ERR_CORRUPT : MIC_ERR_ARGS;
    if (in_len > 0xFFFFFFF0ull || out_cap > ((size_t)1 << 30)) return MIC_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_device();
    if (rc) return rc;
    mic_hip_session *s = &g_default;
    const size_t px = (out_cap + 3) / 4 + 16;
    if ((rc = s->ensure(1, px))) return rc;
    if ((rc = s->io_comp.reserve(in_len + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(s->io_comp.p, in, in_len, hipMemcpyHostToDevice, s->stream));
    s->h_units.assign(1, MicUnit{});
    MicUnit &u = s->h_units[0];
    u.comp_in = (const uint8_t *)s->io_comp.p; u.comp_len = (uint32_t)in_len; u.w = 1; u.h = 1; u.mode = 1;
    s->fill_workspace(u, 0);
    u.tok_cap = (uint32_t)std::min<size_t>(out_cap, u.tok_cap);
    HIP_TRY(hipMemcpyAsync(s->units.p, s->h_units.data(), sizeof(MicUnit), hipMemcpyHostToDevice, s->stream));
    mic_launch_decode((MicUnit *)s->units.p, 1, s->stream, s->variant, nullptr, (int *)s->cls.p);
    HIP_TRY(hipGetLastError());
    s->n_last = 1;
    int32_t st = 0;
    if ((rc = session_decode_finish(s, &st))) return rc;
    if (st != MIC_OK) return st;
    const size_t n = s->h_units[0].ntok;
    if (s->h_units[0].flavour == 1 && (uint64_t)n >= limit) return MIC_ERR_CAPACITY;        // fsedecompressu16.go:372
    if (n > out_cap) return MIC_ERR_CAPACITY;
    if (n) HIP_TRY(hipMemcpy(out, s->h_units[0].tok, n * 2, hipMemcpyDeviceToHost));
    *out_n = n;
    return MIC_OK;
}

// ---- PICS (parallelstrips.go) ----------------------------------------------------------------
// Units cut out of ONE host pixel buffer (the strips of an image, the frames of a stack): the buffer is uploaded once, the units are
// coded in sub-batches that keep the workspace bounded, and the streams come back into one exactly-sized host vector
// (no per-unit worst-case staging).  res[i] = {status, offset into store, length}.
struct UnitResult { int32_t status; size_t off, len; };
static int encode_units_of_buffer(const uint16_t *pixels, size_t total_px, const std::vector<mic_hip_unit> &units,
                                  std::vector<UnitResult> &res, std::vector<uint8_t> &store) {
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_device();
    if (rc) return rc;
    mic_hip_session *s = &g_default;
    if ((rc = s->ensure(1, 1))) return rc;                                  // (the stream)
    if ((rc = s->io_px.reserve(total_px * 2 + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(s->io_px.p, pixels, total_px * 2, hipMemcpyHostToDevice, s->stream));
    const int n = (int)units.size();
    res.assign((size_t)n, UnitResult{ MIC_OK, 0, 0 });
    store.clear();
    int i0 = 0;
    while (i0 < n) {
        size_t max_px = 0; int i1 = i0;
        while (i1 < n) {
            const size_t px = (size_t)units[(size_t)i1].width * (size_t)units[(size_t)i1].height;
            if (px > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
            const size_t mp = std::max(max_px, px);
            if (i1 > i0 && unit_ws_bytes(mp) * (size_t)(i1 - i0 + 1) > kWorkspaceBudget) break;
            max_px = mp; i1++;
        }
        const int nb = i1 - i0;
        if ((rc = session_encode_enqueue(s, (const uint16_t *)s->io_px.p, units.data() + i0, nb))) return rc;
        std::vector<uint64_t> offs((size_t)nb + 1); std::vector<int32_t> st((size_t)nb), ns((size_t)nb);
        const uint8_t *d_blobs = nullptr;
        if ((rc = session_encode_finish(s, &d_blobs, offs.data(), st.data(), ns.data()))) return rc;
        const size_t base = store.size();
        store.resize(base + (size_t)offs[(size_t)nb]);
        if (offs[(size_t)nb]) HIP_TRY(hipMemcpy(store.data() + base, d_blobs, (size_t)offs[(size_t)nb], hipMemcpyDeviceToHost));
        for (int k = 0; k < nb; k++)
            res[(size_t)(i0 + k)] = UnitResult{ st[(size_t)k], base + (size_t)offs[(size_t)k], (size_t)(offs[(size_t)k + 1] - offs[(size_t)k]) };
        i0 = i1;
    }
    return MIC_OK;
}

// The mirror image for decode: the streams of the units lie inside ONE host buffer (a container file) and their pixels form one
// contiguous image / stack.  The file goes up once, the units are decoded in sub-batches, the pixels come down once.
struct UnitSpan { size_t start, len; uint64_t px_offset; int32_t width, height; uint16_t flags; };
static int decode_units_of_file(const uint8_t *file, size_t file_len, const std::vector<UnitSpan> &spans, size_t total_px,
                                uint16_t *pixels_out, std::vector<int32_t> &status) {
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_device();
    if (rc) return rc;
    mic_hip_session *s = &g_default;
    if ((rc = s->ensure(1, 1))) return rc;
    if ((rc = s->io_comp.reserve(file_len + 64)) || (rc = s->io_px.reserve(total_px * 2 + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(s->io_comp.p, file, file_len, hipMemcpyHostToDevice, s->stream));
    const int n = (int)spans.size();
    status.assign((size_t)n, MIC_OK);
    {   // pixels no unit writes come back as zeros, like the reference's make([]uint16, w*h) (parallelstrips.go:288): a PICS header
        // whose strips do not cover the image is accepted there, and the staging buffer holds an earlier call's pixels
        size_t covered = 0;
        for (const UnitSpan &u : spans) if (u.width > 0 && u.height > 0) covered += (size_t)u.width * (size_t)u.height;
        if (covered < total_px) HIP_TRY(hipMemsetAsync(s->io_px.p, 0, total_px * 2, s->stream));
    }
    int i0 = 0;
    while (i0 < n) {
        size_t max_px = 0; int i1 = i0;
        while (i1 < n) {
            const UnitSpan &u = spans[(size_t)i1];
            if (u.width <= 0 || u.height <= 0 || u.len == 0 || u.len > 0xFFFFFFF0ull || u.start + u.len > file_len) break;   // reported below
            const size_t px = (size_t)u.width * (size_t)u.height;
            if (px > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
            const size_t mp = std::max(max_px, px);
            if (i1 > i0 && unit_ws_bytes(mp) * (size_t)(i1 - i0 + 1) > kWorkspaceBudget) break;
            max_px = mp; i1++;
        }
        if (i1 == i0) {                                                     // a span that cannot be decoded at all
            const UnitSpan &u = spans[(size_t)i0];
            status[(size_t)i0] = (u.len == 0 || u.start + u.len > file_len) ? MIC_ERR_CORRUPT : MIC_ERR_ARGS;
            i0++; continue;
        }
        const int nb = i1 - i0;
        if ((rc = s->ensure(nb, max_px))) return rc;
        s->h_units.assign((size_t)nb, MicUnit{});
        bool any_grad = false;
        for (int k = 0; k < nb; k++) {
            const UnitSpan &sp = spans[(size_t)(i0 + k)];
            MicUnit &u = s->h_units[(size_t)k];
            u.comp_in = (const uint8_t *)s->io_comp.p + sp.start; u.comp_len = (uint32_t)sp.len;
            u.px_out = (uint16_t *)s->io_px.p + sp.px_offset;
            u.w = sp.width; u.h = sp.height;
            u.pred = (sp.flags & MIC_HIP_PRED_GRAD) ? 1u : 0u; any_grad |= u.pred != 0;
            s->fill_workspace(u, k);
            u.tok_cap = (uint32_t)tok_cap_for((size_t)u.w * (size_t)u.h);
        }
        HIP_TRY(hipMemcpyAsync(s->units.p, s->h_units.data(), sizeof(MicUnit) * (size_t)nb, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemsetAsync(s->flags.p, 0, s->flag_stride * (size_t)nb, s->stream));
        mic_launch_decode((MicUnit *)s->units.p, nb, s->stream, s->variant | (any_grad ? MIC_VARIANT_GRAD : 0), nullptr, (int *)s->cls.p);
        HIP_TRY(hipGetLastError());
        s->n_last = nb;
        std::vector<int32_t> st((size_t)nb);
        if ((rc = session_decode_finish(s, st.data()))) return rc;
        for (int k = 0; k < nb; k++) status[(size_t)(i0 + k)] = st[(size_t)k];
        i0 = i1;
    }
    for (int32_t v : status) if (v != MIC_OK) return MIC_OK;               // the caller reports the first failing unit; no pixels owed
    HIP_TRY(hipMemcpy(pixels_out, s->io_px.p, total_px * 2, hipMemcpyDeviceToHost));
    return MIC_OK;
}

int mic_hip_pics_compress(const uint16_t *pixels, int width, int height, uint16_t max_value, int num_strips, int nstates,
                          uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!pixels || !out || !out_len || width <= 0 || height <= 0 || num_strips <= 0) return MIC_ERR_ARGS;
    if (!(nstates == 2 || nstates == 4 || nstates == 8)) return MIC_ERR_ARGS;
    if (num_strips > height) num_strips = height;                    // parallelstrips.go:62-67
    int strip_h = (height + num_strips - 1) / num_strips;            // :70
    int actual = (height + strip_h - 1) / strip_h;                   // :72
    size_t header = 20 + (size_t)actual * 8;
    if (out_cap < header) return MIC_ERR_CAPACITY;
    std::vector<mic_hip_unit> units((size_t)actual);
    for (int s = 0; s < actual; s++) {
        const int y0 = s * strip_h, y1 = std::min(height, y0 + strip_h);
        units[(size_t)s] = mic_hip_unit{ (uint64_t)y0 * (uint64_t)width, width, y1 - y0, max_value, (uint16_t)nstates };   // global maxValue for every strip, :88
    }
    std::vector<UnitResult> res; std::vector<uint8_t> store;
    int rc = encode_units_of_buffer(pixels, (size_t)width * (size_t)height, units, res, store);
    if (rc) return rc;
    size_t total = 0;
    for (int s = 0; s < actual; s++) {
        if (res[(size_t)s].status != MIC_OK) return res[(size_t)s].status;   // first failing strip, :95-99
        total += res[(size_t)s].len;
    }
    if (total > 0xFFFFFFFFull) return MIC_ERR_UNSUPPORTED;
    if (out_cap < header + total) return MIC_ERR_CAPACITY;
    memcpy(out, "PICS", 4);
    put_u32(out + 4, (uint32_t)width); put_u32(out + 8, (uint32_t)height);
    put_u32(out + 12, (uint32_t)actual); put_u32(out + 16, (uint32_t)strip_h);
    size_t off = 0;
    for (int s = 0; s < actual; s++) {
        put_u32(out + 20 + (size_t)s * 8, (uint32_t)off);
        put_u32(out + 24 + (size_t)s * 8, (uint32_t)res[(size_t)s].len);
        memcpy(out + header + off, store.data() + res[(size_t)s].off, res[(size_t)s].len);
        off += res[(size_t)s].len;
    }
    *out_len = header + total;
    return MIC_OK;
}

int mic_hip_pics_info(const uint8_t *c, size_t len, int *width, int *height, int *num_strips, int *strip_height) {
    if (!c) return MIC_ERR_ARGS;
    if (len < 20 || memcmp(c, "PICS", 4) != 0) return MIC_ERR_CORRUPT;   // parallelstrips.go:271-273
    int w = (int)get_u32(c + 4), h = (int)get_u32(c + 8), n = (int)get_u32(c + 12), sh = (int)get_u32(c + 16);
    if (n < 0 || (size_t)n > (len - 20) / 8) return MIC_ERR_CORRUPT;     // truncated header, :281-283
    if (w <= 0 || h <= 0 || n <= 0 || sh <= 0) return MIC_ERR_CORRUPT;   // :284-286
    if (width) *width = w; if (height) *height = h; if (num_strips) *num_strips = n; if (strip_height) *strip_height = sh;
    return MIC_OK;
}

int mic_hip_pics_decompress(const uint8_t *c, size_t len, uint16_t *pixels_out, int width, int height) {
    if (!c || !pixels_out) return MIC_ERR_ARGS;
    int w, h, n, sh;
    int rc = mic_hip_pics_info(c, len, &w, &h, &n, &sh);
    if (rc) return rc;
    if (w != width || h != height) return MIC_ERR_ARGS;
    size_t header = 20 + (size_t)n * 8;
    std::vector<UnitSpan> spans((size_t)n);
    for (int s = 0; s < n; s++) {
        size_t so = get_u32(c + 20 + (size_t)s * 8), sl = get_u32(c + 24 + (size_t)s * 8);
        size_t start = header + so, end = start + sl;
        if (end > len || start > end) return MIC_ERR_CORRUPT;            // :300-304
        long y0 = (long)s * sh, y1 = std::min<long>(h, y0 + sh);
        if (y0 >= h) return MIC_ERR_CORRUPT;
        spans[(size_t)s] = UnitSpan{ start, sl, (uint64_t)y0 * (uint64_t)w, w, (int32_t)(y1 - y0), 0 };
    }
    std::vector<int32_t> st;
    rc = decode_units_of_file(c, len, spans, (size_t)w * (size_t)h, pixels_out, st);
    if (rc) return rc;
    for (int s = 0; s < n; s++) if (st[(size_t)s] != MIC_OK) return st[(size_t)s];
    return MIC_OK;
}

// ---- MIC2 independent mode (multiframe.go, multiframecompress.go:179-261) -----------------------
int mic_hip_mic2_compress(const uint16_t *frames, int width, int height, int nframes, uint16_t max_value,
                          uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!frames || !out || !out_len || width <= 0 || height <= 0 || nframes <= 0) return MIC_ERR_ARGS;
    size_t npx = (size_t)width * (size_t)height;
    size_t header = 20 + (size_t)nframes * 8;
    if (out_cap < header) return MIC_ERR_CAPACITY;
    std::vector<mic_hip_unit> units((size_t)nframes);
    for (int i = 0; i < nframes; i++) units[(size_t)i] = mic_hip_unit{ (uint64_t)npx * (uint64_t)i, width, height, max_value, 2 };
    std::vector<UnitResult> res; std::vector<uint8_t> store;
    int rc = encode_units_of_buffer(frames, npx * (size_t)nframes, units, res, store);
    if (rc) return rc;
    size_t total = 0;
    for (int i = 0; i < nframes; i++) { if (res[(size_t)i].status != MIC_OK) return res[(size_t)i].status; total += res[(size_t)i].len; }
    if (total > 0xFFFFFFFFull) return MIC_ERR_UNSUPPORTED;              // u32 offsets, multiframe.go:75-80
    if (out_cap < header + total) return MIC_ERR_CAPACITY;
    memset(out, 0, header);
    memcpy(out, "MIC2", 4);
    put_u32(out + 4, (uint32_t)width); put_u32(out + 8, (uint32_t)height); put_u32(out + 12, (uint32_t)nframes);
    out[16] = 0x01;                                                     // PipelineSpatial, multiframe.go:28
    size_t off = 0;
    for (int i = 0; i < nframes; i++) {
        put_u32(out + 20 + (size_t)i * 8, (uint32_t)off);
        put_u32(out + 24 + (size_t)i * 8, (uint32_t)res[(size_t)i].len);
        memcpy(out + header + off, store.data() + res[(size_t)i].off, res[(size_t)i].len);
        off += res[(size_t)i].len;
    }
    *out_len = header + total;
    return MIC_OK;
}

int mic_hip_mic2_compress_temporal(const uint16_t *frames, int width, int height, int nframes, uint16_t max_value,
                                   uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!frames || !out || !out_len || width <= 0 || height <= 0 || nframes <= 0) return MIC_ERR_ARGS;
    return mic2_temporal_compress(frames, width, height, nframes, max_value, out, out_cap, out_len);
}

int mic_hip_mic2_info(const uint8_t *c, size_t len, int *width, int *height, int *nframes, int *temporal) {
    if (!c) return MIC_ERR_ARGS;
    if (len < 20 || memcmp(c, "MIC2", 4) != 0) return MIC_ERR_CORRUPT;   // multiframe.go:96-103
    int w = (int)get_u32(c + 4), h = (int)get_u32(c + 8), n = (int)get_u32(c + 12);
    if (n < 0 || (size_t)n > (len - 20) / 8) return MIC_ERR_CORRUPT;     // :112-116
    if (width) *width = w; if (height) *height = h; if (nframes) *nframes = n; if (temporal) *temporal = (c[16] & 0x02) != 0;
    return MIC_OK;
}

int mic_hip_mic2_decompress(const uint8_t *c, size_t len, uint16_t *frames_out, size_t frames_cap_px) {
    if (!c || !frames_out) return MIC_ERR_ARGS;
    int w, h, n, temporal;
    int rc = mic_hip_mic2_info(c, len, &w, &h, &n, &temporal);
    if (rc) return rc;
    if (w <= 0 || h <= 0 || n <= 0) return MIC_ERR_CORRUPT;
    size_t npx = (size_t)w * (size_t)h;
    if (npx * (size_t)n > frames_cap_px) return MIC_ERR_CAPACITY;
    if (temporal) return mic2_temporal_decompress(c, len, w, h, n, n, frames_out);   // mic_temporal.hip
    size_t data_off = 20 + (size_t)n * 8;
    std::vector<UnitSpan> spans((size_t)n);
    for (int i = 0; i < n; i++) {
        size_t start = data_off + get_u32(c + 20 + (size_t)i * 8), bl = get_u32(c + 24 + (size_t)i * 8);
        if (start + bl > len) return MIC_ERR_CORRUPT;                     // multiframe.go:137-139
        spans[(size_t)i] = UnitSpan{ start, bl, (uint64_t)npx * (uint64_t)i, w, h, 0 };
    }
    std::vector<int32_t> st;
    rc = decode_units_of_file(c, len, spans, npx * (size_t)n, frames_out, st);
    if (rc) return rc;
    for (int i = 0; i < n; i++) if (st[(size_t)i] != MIC_OK) return st[(size_t)i];
    return MIC_OK;
}

// DecompressFrame (multiframecompress.go:266-315): one frame of a MIC2 file.  Independent mode decodes just that
// frame; temporal mode needs frames 0..idx (all their residual streams are decoded at once, mic_temporal.hip).
int mic_hip_mic2_decompress_frame(const uint8_t *c, size_t len, int frame_idx, uint16_t *pixels_out, size_t pixels_cap) {
    if (!c || !pixels_out) return MIC_ERR_ARGS;
    int w, h, n, temporal;
    int rc = mic_hip_mic2_info(c, len, &w, &h, &n, &temporal);
    if (rc) return rc;
    if (frame_idx < 0 || frame_idx >= n) return MIC_ERR_ARGS;           // "frame index out of range"
    if (w <= 0 || h <= 0) return MIC_ERR_CORRUPT;
    const size_t npx = (size_t)w * (size_t)h;
    if (npx > pixels_cap) return MIC_ERR_CAPACITY;
    const size_t data_off = 20 + (size_t)n * 8;
    if (!temporal) {
        const size_t start = data_off + get_u32(c + 20 + (size_t)frame_idx * 8), bl = get_u32(c + 24 + (size_t)frame_idx * 8);
        if (start + bl > len) return MIC_ERR_CORRUPT;                     // ExtractFrame, multiframe.go:131-142
        return mic_hip_decompress_frame(c + start, bl, pixels_out, w, h);
    }
    std::vector<uint16_t> tmp(npx * (size_t)(frame_idx + 1));
    rc = mic2_temporal_decompress(c, len, w, h, n, frame_idx + 1, tmp.data());
    if (rc) return rc;
    memcpy(pixels_out, tmp.data() + npx * (size_t)frame_idx, npx * 2);
    return MIC_OK;
}

// ---- sessions ------------------------------------------------------------------------------------
int mic_hip_session_create_on(int device, mic_hip_session **out, int max_units, size_t max_px_per_unit) {
    if (!out || max_units <= 0 || max_px_per_unit == 0) return MIC_ERR_ARGS;
    int rc = check_device(device);
    if (rc) return rc;
    mic_hip_session *s = new mic_hip_session();
    s->device = device;
    if ((rc = s->activate())) { delete s; return rc; }
    rc = s->ensure(max_units, max_px_per_unit);
    if (rc) { s->release(); delete s; return rc; }
    const char *v = getenv("MIC_HIP_VARIANT");
    if (v) s->variant = atoi(v);
    *out = s;
    return MIC_OK;
}
int mic_hip_session_create(mic_hip_session **out, int max_units, size_t max_px_per_unit) {
    int dev;
    { std::lock_guard<std::mutex> lk(g_mu); dev = g_device; }
    return mic_hip_session_create_on(dev, out, max_units, max_px_per_unit);
}
int mic_hip_session_device(mic_hip_session *s) { return s ? s->device : -1; }
void mic_hip_session_destroy(mic_hip_session *s) { if (s) { (void)s->activate(); s->release(); delete s; } }
void *mic_hip_session_stream(mic_hip_session *s) { return s ? (void *)s->stream : nullptr; }

int mic_hip_session_encode_enqueue(mic_hip_session *s, const uint16_t *d_pixels, const mic_hip_unit *units, int n) {
    if (!s || !d_pixels || !units) return MIC_ERR_ARGS;
    { const int arc = s->activate(); if (arc) return arc; }
    return session_encode_enqueue(s, d_pixels, units, n);
}
int mic_hip_session_encode_finish(mic_hip_session *s, const uint8_t **d_blobs, uint64_t *h_offsets, int32_t *h_status, int32_t *h_nstates) {
    if (!s || !h_offsets || !h_status) return MIC_ERR_ARGS;
    { const int arc = s->activate(); if (arc) return arc; }
    return session_encode_finish(s, d_blobs, h_offsets, h_status, h_nstates);
}
int mic_hip_session_encode(mic_hip_session *s, const uint16_t *d_pixels, const mic_hip_unit *units, int n,
                           const uint8_t **d_blobs, uint64_t *h_offsets, int32_t *h_status, int32_t *h_nstates) {
    int rc = mic_hip_session_encode_enqueue(s, d_pixels, units, n);
    if (rc) return rc;
    return mic_hip_session_encode_finish(s, d_blobs, h_offsets, h_status, h_nstates);
}
int mic_hip_session_decode_enqueue(mic_hip_session *s, const uint8_t *d_blobs, const uint64_t *h_offsets,
                                   const mic_hip_unit *units, int n, uint16_t *d_pixels_out) {
    if (!s || !d_blobs || !h_offsets || !units || !d_pixels_out) return MIC_ERR_ARGS;
    { const int arc = s->activate(); if (arc) return arc; }
    return session_decode_enqueue(s, d_blobs, h_offsets, units, n, d_pixels_out);
}
int mic_hip_session_decode_finish(mic_hip_session *s, int32_t *h_status) {
    if (!s || !h_status) return MIC_ERR_ARGS;
    { const int arc = s->activate(); if (arc) return arc; }
    return session_decode_finish(s, h_status);
}
int mic_hip_session_decode(mic_hip_session *s, const uint8_t *d_blobs, const uint64_t *h_offsets, const mic_hip_unit *units, int n,
                           uint16_t *d_pixels_out, int32_t *h_status) {
    int rc = mic_hip_session_decode_enqueue(s, d_blobs, h_offsets, units, n, d_pixels_out);
    if (rc) return rc;
    return mic_hip_session_decode_finish(s, h_status);
}
// debug probe (not part of the public header): raw result fields of unit i after a *_finish
int mic_hip_debug_unit(mic_hip_session *s, int i, uint32_t *out8) {
    if (!s || i < 0 || i >= s->n_last) return MIC_ERR_ARGS;
    const MicUnit &u = s->h_units[(size_t)i];
    out8[0] = u.ntok; out8[1] = u.blob_len; out8[2] = u.table_log; out8[3] = u.symbol_len;
    out8[4] = u.max_count; out8[5] = u.hdr_len; out8[6] = u.zero_bits; out8[7] = u.flavour;
    out8[8] = u.count; out8[9] = u.bits_off; out8[10] = (uint32_t)u.nstates_used; out8[11] = (uint32_t)u.status; out8[12] = u.nseg; out8[13] = u.nsym; out8[14] = u.seg_cap;
    for (int k = 0; k < 16; k++) out8[16 + k] = u.dbg[k];
    return MIC_OK;
}
// debug probe (not in the public header): bytes of unit i's histogram slab after a *_finish (LS_DEBUG builds dump there)
int mic_hip_debug_fetch_hist(mic_hip_session *s, int i, void *dst, size_t bytes) {
    if (!s || i < 0 || i >= s->n_last || bytes > kSym * 4) return MIC_ERR_ARGS;
    HIP_TRY(hipMemcpy(dst, s->h_units[(size_t)i].hist, bytes, hipMemcpyDeviceToHost));
    return MIC_OK;
}
// debug probe (not in the public header): the first n u16 of unit i's token slab after a *_finish
int mic_hip_debug_fetch_tok(mic_hip_session *s, int i, void *dst, size_t n) {
    if (!s || i < 0 || i >= s->n_last || n > s->h_units[(size_t)i].tok_cap) return MIC_ERR_ARGS;
    HIP_TRY(hipMemcpy(dst, s->h_units[(size_t)i].tok, n * 2, hipMemcpyDeviceToHost));
    return MIC_OK;
}
int mic_hip_session_set_timing(mic_hip_session *s, int enabled) {
    if (!s) return MIC_ERR_ARGS;
    s->timer.enabled = enabled != 0;
    s->timer.accumulate = enabled == 2;                                   // 2: sum over every launch chain until the next set_timing
    s->timer.clear();
    return MIC_OK;
}
int mic_hip_session_last_timings(mic_hip_session *s, const char **names, float *ms, int cap) {
    if (!s) return 0;
    s->t_names.clear(); s->t_ms.clear();
    if (s->timer.used >= 2) {
        if (hipEventSynchronize(s->timer.pool[s->timer.used - 1]) != hipSuccess) return 0;
        for (size_t i = 0; i + 1 < s->timer.used; i++) {
            if (strcmp(s->timer.names[i], "end") == 0) continue;           // the gap between two launch chains
            float v = 0.f;
            if (hipEventElapsedTime(&v, s->timer.pool[i], s->timer.pool[i + 1]) != hipSuccess) v = -1.f;
            size_t k = 0;
            while (k < s->t_names.size() && s->t_names[k] != s->timer.names[i]) k++;
            if (k == s->t_names.size()) { s->t_names.push_back(s->timer.names[i]); s->t_ms.push_back(v); }
            else s->t_ms[k] += v;
        }
    }
    int n = (int)std::min<size_t>(s->t_names.size(), (size_t)std::max(cap, 0));
    for (int i = 0; i < n; i++) { names[i] = s->t_names[(size_t)i].c_str(); ms[i] = s->t_ms[(size_t)i]; }
    return n;
}

}  // extern "C"
