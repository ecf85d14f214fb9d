// mic_session.h -- host-side session state shared by mic_api.hip and mic_api_ext.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <atomic>
#include <vector>

#include "../../include/mic_hip.h"
#include "mic_dev.h"
#include "mic_launch.h"

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            if (getenv("MIC_HIP_DEBUG"))                                                  \
                fprintf(stderr, "mic_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? MIC_ERR_NOMEM : MIC_ERR_DEVICE;            \
        }                                                                                 \
    } while (0)

namespace micapi {

extern std::mutex g_mu;     // guards the device choice and the pool of default sessions (mic_api.hip)
extern int g_device;
int ensure_device();
int check_device(int device);      // MIC_OK when `device` exists and is gfx950 (cached per device)

// A session's stream.  MIC_HIP_SESSION_PRIO_CYCLE=1 (experiments): successive sessions get successive priority levels, i.e.
// hardware queues of their own -- the runtime keeps a queue per level, and streams of one level may share one.
inline hipError_t mic_stream_create(hipStream_t *st) {
    static const bool cycle = getenv("MIC_HIP_SESSION_PRIO_CYCLE") != nullptr;
    if (!cycle) return hipStreamCreate(st);
    static std::atomic<int> next{0};
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);                       // (lo is the numerically larger, least urgent one)
    const int span = lo - hi + 1, k = next.fetch_add(1) % (span > 0 ? span : 1);
    return hipStreamCreateWithPriority(st, hipStreamDefault, lo - k);
}

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    uint64_t gen = 0;                       // bumped by every (re)allocation: the contents are undefined from then on, whatever the address
    int reserve(size_t bytes) {
        if (bytes <= cap) return MIC_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 4096;
        gen++;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return MIC_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; gen++; }
};

// The host copy of a launch's unit descriptors, in pinned memory: the upload in front of every launch chain and the download of
// the results behind it are DMA transfers that really are asynchronous (a pageable vector cost ~0.1 ms of staging each way per call).
// Because the upload is asynchronous, the descriptors may not be rewritten while it is in flight: assign() waits for it.
struct PinnedUnits {
    MicUnit *p = nullptr; size_t cap = 0, n = 0;
    hipEvent_t ev = nullptr; bool inflight = false;
    int assign(size_t count, const MicUnit &v) {
        if (inflight) { (void)hipEventSynchronize(ev); inflight = false; }
        if (count > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            const size_t want = count + count / 4 + 16;
            HIP_TRY(hipHostMalloc((void **)&p, want * sizeof(MicUnit), hipHostMallocDefault));
            cap = want;
        }
        for (size_t i = 0; i < count; i++) p[i] = v;
        n = count;
        return MIC_OK;
    }
    int upload(void *d_dst, size_t count, hipStream_t stream) {
        HIP_TRY(hipMemcpyAsync(d_dst, p, sizeof(MicUnit) * count, hipMemcpyHostToDevice, stream));
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, stream));
        inflight = true;
        return MIC_OK;
    }
    MicUnit *data() { return p; }
    MicUnit &operator[](size_t i) { return p[i]; }
    const MicUnit &operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    void release() { if (inflight) (void)hipEventSynchronize(ev); inflight = false; if (p) (void)hipHostFree(p); p = nullptr; cap = n = 0; if (ev) (void)hipEventDestroy(ev); ev = nullptr; }
};

// a small pinned array the device writes results into (offsets of the packed streams)
struct PinnedU64 {
    uint64_t *p = nullptr; size_t cap = 0;
    int reserve(size_t count) {
        if (count <= cap) return MIC_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = count + count / 4 + 64;
        HIP_TRY(hipHostMalloc((void **)&p, want * sizeof(uint64_t), hipHostMallocDefault));
        cap = want;
        return MIC_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

constexpr size_t kSym = 65536;

// Two tiers of per-unit slabs.  Tier 2 is the worst case: two tokens per pixel (every pixel an escape), a 65536-symbol alphabet,
// a segment per token pair -- 42 bytes per pixel + 1.66 MB of tables.  Tier 1 is what well-formed data needs: one token per
// pixel and an eighth, tables for 8192 symbols / tableLog 13, a segment per eight pixels -- 7 bytes per pixel + 0.2 MB.  The unit
// codec runs in tier 1 first; a kernel that would cross a tier-1 capacity marks its unit MICD_INT_GROW and the batch is run
// again in tier 2 (session_*_finish), so results never depend on the tier.
inline size_t tok_cap_for(size_t px) { return 4 * px + 16; }                                  // tier 2 (and the legit bound of a stream's token count)
inline size_t tok_cap_tier(size_t px, int tier) { return tier == 1 ? px + px / 8 + 4096 : tok_cap_for(px); }
inline size_t blob_cap_tok(size_t tokc) { return 8 + 131080 + 2 * tokc + 16; }
inline size_t blob_cap_for(size_t px) { return blob_cap_tok(tok_cap_for(px)); }
inline size_t seg_cap_tier(size_t px, int tier) { return tier == 1 ? px / 8 + 1024 : 2 * px + 8; }
inline size_t tab_syms_tier(int tier) { return tier == 1 ? 8192 : 65536; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace micapi
using namespace micapi;

struct mic_hip_wsi_store;                       // MIC3 store of a device-resident slide (mic_api_ext.hip)
void mic_wsi_store_free(mic_hip_wsi_store *w);

struct mic_hip_session {
    int device = 0;                         // the HIP device this session's stream and workspace live on (mic_hip_session_create_on)
    int max_units = 0; size_t max_px = 0;   // shape of the current workspace layout (see ensure)
    int tier = 2;                           // tier of the current layout
    bool force_big = false;                 // a batch of this session needed tier 2: later ones start there ...
    int calm_batches = 0;                   // ... until kTierCalm batches in a row would have fitted tier 1 (tier_review, mic_api.hip):
    bool shrink_pending = false;            //     set there; the next tier-1 layout then starts from released slabs
    static constexpr int kTierCalm = 8;     //     the session goes back to the small slabs and gives the large ones' memory back
    // what a tier-1 launch chain needs to be run again in tier 2 (session_*_finish)
    struct Retry { int kind = 0; const void *d_in = nullptr; void *d_out = nullptr; std::vector<mic_hip_unit> units; std::vector<uint64_t> begins, ends; } retry;
    hipStream_t stream = nullptr;
    // Entry points may be called from any OS thread (cgo: any goroutine's thread), and a process may hold sessions on several
    // devices: every public call makes the session's device the calling thread's current one first.
    int activate() { return hipSetDevice(device) == hipSuccess ? MIC_OK : MIC_ERR_DEVICE; }
    DevBuf units, cls, tok, hist, norm, tt_nb, tt_find, state_tab, tab_sym, cumul, blob, packed, offsets, seg, sym, flags;
    DevBuf io_px, io_comp;                 // staging for the host-pointer entry points
    DevBuf io_px2, io_comp2, packed2;      // their second halves: sub-batch k + 1 comes up while k is coded and k - 1 goes down (mic_host_io.hip)
    DevBuf wv_a, wv_b;                     // WaveletV2 coefficient planes (int32, two per frame of the batch)
    mic_hip_wsi_store *wsi = nullptr;      // mic_hip_session_wsi_*: coded planes of a slide, on the device
    DevBuf wsi_planes, wsi_stats, wsi_payload, wsi_recs; std::vector<DevBuf> wsi_pyr;
    PinnedUnits h_units;
    std::vector<uint64_t> h_off;
    // what an enqueue has already put behind its chain (session_*_finish then only synchronises): the read-back of the descriptors;
    // for encode also scan + pack into `packed` (capacity pack_cap at that time) and the read-back of the offsets into pin_off
    bool readback_queued = false, pack_queued = false;
    size_t pack_cap = 0, pack_hint = 0;           // pack_hint: bytes the session's last batch packed to
    PinnedU64 pin_off;
    int n_last = 0;
    // a launch chain over n units has been enqueued; nothing is queued behind it yet (the enqueue that queues its read-back says so after this)
    void begin_chain(int n) { n_last = n; readback_queued = false; pack_queued = false; learn_decode = learn_encode = false; }
    bool learn_decode = false, learn_encode = false;   // the chain in flight was launched under the session's class masks: finish updates them
    int variant = 0;                        // launch flags (MIC_VARIANT_GRAD is OR-ed in per call)
    // Kernel classes this session's last batches used (mic_launch.h: launch masks): age[c] = batches since class c was last seen;
    // a class is launched while its age is below kClsKeep.  Nothing seen yet: everything is launched.
    static constexpr uint8_t kClsKeep = 8;
    struct ClsMemory {
        uint8_t age[32]; bool any = false;
        ClsMemory() { for (uint8_t &a : age) a = 255; }
        uint32_t mask() const { if (!any) return ~0u; uint32_t m = 0; for (int c = 0; c < 32; c++) if (age[c] < kClsKeep) m |= 1u << c; return m; }
        void learn(uint32_t seen) { any = true; for (int c = 0; c < 32; c++) age[c] = ((seen >> c) & 1u) ? 0 : (uint8_t)std::min<int>(age[c] + 1, 255); }
    } dec_classes, enc_classes;
    // The per-unit 65536-bin histograms are ZERO between calls: the encode chain leaves them so (k_enc_hist_clean re-zeroes what a
    // unit's tokeniser counted), and nothing else writes them.  hist_zero_units = leading unit slabs known to be zero (0 after a
    // reallocation, after a failed launch).  The invariant is tied to the ALLOCATION (DevBuf::gen),
    // not to the address: hipFree + a larger hipMalloc may hand the same address back with undefined contents.
    uint64_t hist_zero_gen = ~0ull; size_t hist_zero_units = 0;
    int prepare_hist(int n) {
        if (hist.gen != hist_zero_gen) { hist_zero_gen = hist.gen; hist_zero_units = 0; }
        if ((size_t)n > hist_zero_units) {
            HIP_TRY(hipMemsetAsync((char *)hist.p + tab_syms * 4 * hist_zero_units, 0, tab_syms * 4 * ((size_t)n - hist_zero_units), stream));
            hist_zero_units = (size_t)n;
        }
        return MIC_OK;
    }
    void hist_unknown() { hist_zero_units = 0; }
    MicTimer timer;
    std::vector<std::string> t_names; std::vector<float> t_ms;
    size_t tok_stride = 0, blob_stride = 0, seg_stride = 0, sym_stride = 0, flag_stride = 0;

    size_t tab_syms = kSym;                  // symbols / states the table slabs of the current layout hold per unit
    int ensure(int n, size_t px, int want_tier = 2) {
        if (!stream) HIP_TRY(mic_stream_create(&stream));
        // The workspace takes the shape of the current call (n units of up to px pixels, in the tier asked for); buffers only ever
        // grow.  Sizing for max(n) x max(px) over a session's history would ask for the bounding box of unrelated calls (one
        // 4-megapixel wavelet frame, then 3000 WSI planes of 256 x 256).
        if (n == max_units && px == max_px && want_tier == tier) return MIC_OK;
        // From here on the layout is in flux: a reservation that fails half way (DevBuf::reserve frees before it allocates) must not
        // leave the old shape key standing over new strides and freed slabs -- the next call of the old shape would take the early
        // return above and hand the kernels null or short slabs.  The key is cleared first and set again only when every slab stands.
        // (a session also goes from a tier-2 layout to a tier-1 one when its caller alternates between paths -- a WaveletV2 or pyramid
        // call lays out in tier 2, the unit codec in tier 1: that is no reason to give memory back, it would be bought again at once)
        const bool back_to_small = shrink_pending && want_tier == 1;
        if (back_to_small) shrink_pending = false;
        max_units = 0; max_px = 0; tier = 0;
        if (back_to_small) {                                            // the worst-case slabs go back to the device: reserve() only ever grows
            DevBuf *slabs[] = { &tok, &hist, &norm, &tt_nb, &tt_find, &state_tab, &tab_sym, &cumul, &blob, &seg, &sym, &flags };
            for (DevBuf *b : slabs) b->release();
            hist_unknown();
        }
        int nn = n; size_t pp = px;
        const size_t tokc = tok_cap_tier(pp, want_tier), ts = tab_syms_tier(want_tier);
        const size_t tok_s = align_up(tokc * 2, 256);
        const size_t blob_s = align_up(blob_cap_tok(tokc), 256);
        const size_t seg_s = align_up(seg_cap_tier(pp, want_tier) * 8, 256);
        const size_t sym_s = align_up((tokc + 64) * 2, 256);       // + a block: the tANS encoder rounds its per-token states up to 32
        const size_t flag_s = align_up(pp / 8 + 16, 256);          // + the predictor's 3-word read at the last pixel
        const int rc = [&]() -> int {
            int r;
            if ((r = units.reserve(sizeof(MicUnit) * (size_t)nn))) return r;
            if ((r = cls.reserve(4 * MIC_CLS_INTS(nn)))) return r;
            if ((r = tok.reserve(tok_s * (size_t)nn))) return r;
            if ((r = hist.reserve(ts * 4 * (size_t)nn))) return r;
            if ((r = norm.reserve(ts * 4 * (size_t)nn))) return r;
            if ((r = tt_nb.reserve(ts * 4 * (size_t)nn))) return r;
            if ((r = tt_find.reserve(ts * 4 * (size_t)nn))) return r;
            if ((r = state_tab.reserve(ts * 4 * (size_t)nn))) return r;
            if ((r = tab_sym.reserve(ts * 2 * (size_t)nn))) return r;
            if ((r = cumul.reserve((ts + 64) * 4 * (size_t)nn))) return r;
            if ((r = blob.reserve(blob_s * (size_t)nn))) return r;
            if ((r = offsets.reserve(8 * ((size_t)nn + 1)))) return r;
            if ((r = seg.reserve(seg_s * (size_t)nn))) return r;
            if ((r = sym.reserve(sym_s * (size_t)nn))) return r;
            return flags.reserve(flag_s * (size_t)nn);
        }();
        if (rc) { hist_unknown(); return rc; }             // (whatever the histogram slab holds now, nothing of it is known to be zero)
        if (ts != tab_syms) hist_unknown();                // (the histogram slabs are laid out anew: nothing is known to be zero)
        tok_stride = tok_s; blob_stride = blob_s; seg_stride = seg_s; sym_stride = sym_s; flag_stride = flag_s;
        max_units = nn; max_px = pp; tier = want_tier; tab_syms = ts;
        return MIC_OK;
    }
    void fill_workspace(MicUnit &u, int i) {
        const size_t ts = tab_syms;
        u.tier = (uint32_t)tier; u.tab_cap = (uint32_t)ts;
        u.tok = (uint16_t *)((char *)tok.p + tok_stride * (size_t)i);
        u.tok_cap = (uint32_t)std::min<size_t>(tok_cap_tier(max_px, tier), 0xFFFFFFF0u);
        u.hist = (uint32_t *)hist.p + ts * (size_t)i;
        u.norm = (int32_t *)norm.p + ts * (size_t)i;
        u.tt_nb = (uint32_t *)tt_nb.p + ts * (size_t)i;
        u.tt_find = (int32_t *)tt_find.p + ts * (size_t)i;
        u.state_tab = (uint32_t *)state_tab.p + ts * (size_t)i;
        u.tab_sym = (uint16_t *)tab_sym.p + ts * (size_t)i;
        u.cumul = (int32_t *)cumul.p + (ts + 64) * (size_t)i;
        u.blob = (uint8_t *)blob.p + blob_stride * (size_t)i;
        u.blob_cap = (uint32_t)std::min<size_t>(blob_cap_tok(tok_cap_tier(max_px, tier)), 0xFFFFFFF0u);
        u.seg = (uint2 *)((char *)seg.p + seg_stride * (size_t)i);
        u.seg_cap = (uint32_t)std::min<size_t>(seg_cap_tier(max_px, tier), 0xFFFFFFF0u);
        u.sym = (uint16_t *)((char *)sym.p + sym_stride * (size_t)i);
        u.sym_cap = (uint32_t)std::min<size_t>(tok_cap_tier(max_px, tier) + 64, 0xFFFFFFF0u);
        u.flags = (uint32_t *)((char *)flags.p + flag_stride * (size_t)i);
    }
    size_t reserved_bytes() const {
        const DevBuf *all[] = { &units, &cls, &tok, &hist, &norm, &tt_nb, &tt_find, &state_tab, &tab_sym, &cumul, &blob, &packed, &offsets, &seg, &sym, &flags, &io_px, &io_comp, &io_px2, &io_comp2, &packed2, &wv_a, &wv_b, &wsi_planes, &wsi_stats, &wsi_payload, &wsi_recs };
        size_t t = 0;
        for (const DevBuf *b : all) t += b->cap;
        return t;
    }
    void release() {
        DevBuf *all[] = { &units, &cls, &tok, &hist, &norm, &tt_nb, &tt_find, &state_tab, &tab_sym, &cumul, &blob, &packed, &offsets, &seg, &sym, &flags, &io_px, &io_comp, &io_px2, &io_comp2, &packed2, &wv_a, &wv_b, &wsi_planes, &wsi_stats, &wsi_payload, &wsi_recs };
        for (DevBuf *b : all) b->release();
        for (DevBuf &b : wsi_pyr) b.release();
        wsi_pyr.clear();
        if (wsi) { mic_wsi_store_free(wsi); wsi = nullptr; }
        h_units.release(); pin_off.release();
        readback_queued = pack_queued = false;
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        timer.destroy();
    }
};


namespace micapi {
// A session of the default pool, held by the calling thread for the duration of a host-pointer entry point (mic_api.hip).
struct DefaultLease {
    mic_hip_session *s = nullptr; bool held = false;
    int acquire(int device = -1);   // MIC_OK, or why there is no device; blocks while every pooled session of the device is out.
                                    // device: one of mic_hip_set_devices' list (-1: its first -- the default)
    ~DefaultLease();
    DefaultLease() = default;
    DefaultLease(const DefaultLease &) = delete;
    DefaultLease &operator=(const DefaultLease &) = delete;
};
mic_hip_session *cur_default();     // the session the calling thread holds
std::vector<int> default_devices(); // the devices of the host-pointer entry points (mic_hip_set_devices), the default first
int session_encode_enqueue(mic_hip_session *s, const uint16_t *d_pixels, const mic_hip_unit *units, int n);
int session_encode_finish(mic_hip_session *s, const uint8_t **d_blobs, uint64_t *h_offsets, int32_t *h_status, int32_t *h_nstates);
int session_decode_enqueue(mic_hip_session *s, const uint8_t *d_blobs, const uint64_t *h_offsets,
                           const mic_hip_unit *units, int n, uint16_t *d_pixels_out);
// the same with an explicit byte range [begins[i], ends[i]) of d_base per unit (streams that do not lie back to back)
int session_decode_enqueue_spans(mic_hip_session *s, const uint8_t *d_base, const uint64_t *begins, const uint64_t *ends,
                                 const mic_hip_unit *units, int n, uint16_t *d_pixels_out);
int session_decode_finish(mic_hip_session *s, int32_t *h_status);
size_t unit_ws_bytes(size_t px);
size_t unit_ws_bytes_tier(size_t px, int tier);
size_t batch_units_for(size_t px, size_t mult, size_t extra = 0);   // units per sub-batch of the tiered unit codec (mic_api.hip)
// MIC2 temporal pipeline (mic_temporal.hip)
int mic2_temporal_compress(const uint16_t *frames, int width, int height, int nframes, uint16_t max_value,
                           uint8_t *out, size_t out_cap, size_t *out_len);
int mic2_temporal_decompress(const uint8_t *c, size_t len, int w, int h, int n_total, int n, uint16_t *frames_out);
size_t workspace_budget();        // per-call workspace ceiling (mic_api.hip)
// one blocking host <-> device copy through the transfer engine of mic_host_io.hip (pinned host memory: DMA in place; ordinary
// memory: staged through pinned slots by the worker threads).  The device side must be ready / is complete on return.
int host_copy(int device, void *dev, void *host, size_t bytes, bool to_device);
#define kWorkspaceBudget (micapi::workspace_budget())
}  // namespace micapi
