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

// The entry points that take host pointers run on a small POOL of default sessions (ojph/mic_parallel.h:47-48 promises re-entrancy;
// one session behind one mutex would serialise every goroutine of a Go host): a call leases a free session for its duration,
// creates one while the pool is below its size (MIC_HIP_POOL, default 3), waits otherwise.  g_mu guards the device choice and
// the pool's lists; a leased session is used without any lock.
// Several devices (mic_hip_set_devices): a pool per device; the batch entry points cut their jobs into one contiguous shard per
// listed device and run the shards side by side, each on a session of its device's pool (mic_host_io.hip).  g_device is the first
// of the list: what every other host-pointer entry point runs on.
std::mutex g_mu;
std::condition_variable g_pool_cv;
int g_device = 0;
std::vector<int> g_devices(1, 0);
bool g_device_ok = false;
std::string g_device_name;
struct DevPool { std::vector<mic_hip_session *> free_; int made = 0; };
std::map<int, DevPool> g_pools;                        // by device
thread_local mic_hip_session *tl_default = nullptr;   // the session the calling thread holds (leases nest: containers call the unit codec)
thread_local int tl_depth = 0;

static int pool_max() {
    static const int v = [] { const char *e = getenv("MIC_HIP_POOL"); const int n = e ? atoi(e) : 3; return std::min(std::max(n, 1), 16); }();
    return v;
}

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

static int ensure_device_locked() {                                     // g_mu held
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
    return MIC_OK;
}
int ensure_device() { std::lock_guard<std::mutex> lk(g_mu); return ensure_device_locked(); }

mic_hip_session *cur_default() { return tl_default; }

int DefaultLease::acquire(int device) {
    if (tl_default) {                                                   // nested call on this thread: the session it already holds
        s = tl_default; tl_depth++; held = true;
        return s->activate();
    }
    {
        std::unique_lock<std::mutex> lk(g_mu);
        int rc = ensure_device_locked();
        if (rc) return rc;
        if (device < 0) device = g_device;
        else if (std::find(g_devices.begin(), g_devices.end(), device) == g_devices.end()) return MIC_ERR_ARGS;
        DevPool &pool = g_pools[device];
        while (pool.free_.empty() && pool.made >= pool_max()) g_pool_cv.wait(lk);
        if (!pool.free_.empty()) { s = pool.free_.back(); pool.free_.pop_back(); }
        else { s = new mic_hip_session(); s->device = device; pool.made++; }
    }
    tl_default = s; tl_depth = 1; held = true;
    int rc = s->activate();
    if (rc) return rc;
    if (!s->stream) HIP_TRY(mic_stream_create(&s->stream));
    return MIC_OK;
}
DefaultLease::~DefaultLease() {
    if (!held) return;
    if (--tl_depth > 0) return;
    tl_default = nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    g_pools[s->device].free_.push_back(s);
    g_pool_cv.notify_all();
}
std::vector<int> default_devices() { std::lock_guard<std::mutex> lk(g_mu); return g_devices; }

// Per-call workspace ceiling (the container entry points cut their unit lists into sub-batches that stay under it): what the session
// already holds, or -- never more than a quarter of the device -- half of what is free right now, so that several sessions on one GPU
// (the pool, explicit sessions, the caller's own tensors) do not each grow towards the same ceiling.  MIC_HIP_WS_BUDGET_MB overrides it
// (tests walk the sub-batch loops with small inputs).
// Sessions of this thread's device that are out on lease right now (this thread's included; at least 1): each of them reads the same
// free-memory figure at about the same time, so each may plan with its share of it only.
static size_t leased_here() {
    std::lock_guard<std::mutex> lk(g_mu);
    const int dev = tl_default ? tl_default->device : g_device;
    auto it = g_pools.find(dev);
    if (it == g_pools.end()) return 1;
    return (size_t)std::max(1, it->second.made - (int)it->second.free_.size());
}
size_t workspace_budget() {
    static const long env_mb = [] { const char *e = getenv("MIC_HIP_WS_BUDGET_MB"); return e ? atol(e) : 0L; }();
    if (env_mb > 0) return (size_t)env_mb << 20;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { tot = (size_t)96 << 30; fr = tot / 2; }
    const size_t held = tl_default ? tl_default->reserved_bytes() : 0;
    return std::max<size_t>(std::max<size_t>(held, std::min<size_t>(tot / 4, fr / 2 / leased_here())), (size_t)1 << 30);
}

// ---- encode -------------------------------------------------------------------------------
// A launch chain in the given tier.  Tier 1 (the small slabs) is where a batch starts unless the session has needed tier 2 before or
// a unit's depth says so (14 bits and more: an alphabet past the 8192-bin tables); session_*_finish runs the batch again in tier 2
// when a unit reports MICD_INT_GROW.
static int encode_enqueue_tier(mic_hip_session *s, const uint16_t *d_pixels, const mic_hip_unit *units, int n, int tier) {
    size_t max_px = 0;
    for (int i = 0; i < n; i++) max_px = std::max(max_px, (size_t)units[i].width * (size_t)units[i].height);
    int rc = s->ensure(n, max_px, tier);
    if (rc) return rc;
    { const int arc = s->h_units.assign((size_t)n, MicUnit{}); if (arc) return arc; }
    bool any_grad = false, narrow = true;
    for (int i = 0; i < n; i++) {
        MicUnit &u = s->h_units[(size_t)i];
        narrow &= (units[i].nstates & 0xFF) <= 2;
        u.px_in = d_pixels + units[i].px_offset;
        u.w = units[i].width; u.h = units[i].height;
        u.max_value = units[i].max_value; u.nstates = units[i].nstates & 0xFF;
        u.pred = (units[i].nstates & MIC_HIP_PRED_GRAD) ? 1u : 0u; any_grad |= u.pred != 0;
        s->fill_workspace(u, i);
        u.tok_cap = (uint32_t)tok_cap_tier((size_t)u.w * (size_t)u.h, tier);
    }
    { const int urc = s->h_units.upload(s->units.p, (size_t)n, s->stream); if (urc) return urc; }
    if ((rc = s->prepare_hist(n))) return rc;
    s->timer.reset(s->stream);
    // The encoder's classes: what the session's last batches used, plus what the HOST can tell -- the two big instances always (a
    // two-state unit whose attempt fails is handed to the wide one; a miss there is the serial encoder, 0.8 s for an XR batch:
    // tools/mask_miss.py), the one-wave instances when a unit is small enough for them, the tableLog 14-16 ones when a unit is deep
    // enough to have such an alphabet.  What is left to memory alone costs ~5 us a launch when it is wrong the other way.
    uint32_t enc_hint = MIC_ENC_CLS_NARROW2 | MIC_ENC_CLS_WIDE;
    for (int i = 0; i < n; i++) {
        if ((size_t)units[i].width * (size_t)units[i].height <= 131072u) enc_hint |= MIC_ENC_CLS_SMALL12 | MIC_ENC_CLS_SMALL13;
        if (units[i].max_value >= 2048u) enc_hint |= MIC_ENC_CLS_TL14 | MIC_ENC_CLS_TL15 | MIC_ENC_CLS_TL16;
    }
    mic_launch_encode((MicUnit *)s->units.p, n, s->stream, s->variant | MIC_VARIANT_FRAMES | (any_grad ? MIC_VARIANT_GRAD : 0) | (narrow ? MIC_VARIANT_NARROW : 0), &s->timer,
                      s->enc_classes.mask() | enc_hint);
    if (hipGetLastError() != hipSuccess) { s->hist_unknown(); return MIC_ERR_DEVICE; }
    s->begin_chain(n);
    s->learn_encode = true;                                              // (behind begin_chain, which clears it: the masks were never learned)
    // Compaction and the read-back of the results ride behind the chain, so that session_encode_finish is ONE synchronisation (round 3:
    // descriptors down, a synchronisation, sizes summed on the host, scan + pack launched, a second synchronisation -- 86 us of idle
    // device between the chain and the pack of every call).  The packed buffer is sized from what the session's last batch needed;
    // k_enc_pack leaves a batch that does not fit alone and finish packs it again into a buffer of the right size.
    {
        size_t raw = 0;
        for (int i = 0; i < n; i++) raw += (size_t)units[i].width * (size_t)units[i].height * 2;
        const size_t want = std::max(s->pack_hint + s->pack_hint / 8, raw / 3) + ((size_t)64 << 10);
        if ((rc = s->packed.reserve(want))) return rc;
        if ((rc = s->pin_off.reserve((size_t)n + 1))) return rc;
        mic_launch_pack((const MicUnit *)s->units.p, n, (uint64_t *)s->offsets.p, (uint8_t *)s->packed.p, (uint64_t)s->packed.cap, s->stream, &s->timer);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(s->h_units.data(), s->units.p, sizeof(MicUnit) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(s->pin_off.p, s->offsets.p, 8 * ((size_t)n + 1), hipMemcpyDeviceToHost, s->stream));
        s->readback_queued = true; s->pack_queued = true; s->pack_cap = s->packed.cap;
    }
    return MIC_OK;
}

int session_encode_enqueue(mic_hip_session *s, const uint16_t *d_pixels, const mic_hip_unit *units, int n) {
    if (n <= 0) return MIC_ERR_ARGS;
    if (n > 65535) return MIC_ERR_UNSUPPORTED;                           // units are a launch's grid y in several kernels: callers sub-batch
    size_t max_px = 0;
    bool deep = false;
    for (int i = 0; i < n; i++) {
        if (units[i].width <= 0 || units[i].height <= 0) return MIC_ERR_ARGS;
        max_px = std::max(max_px, (size_t)units[i].width * (size_t)units[i].height);
        const int ns = units[i].nstates & 0xFF;
        if (!(ns == 2 || ns == 4 || ns == 8) || (units[i].nstates & ~(0xFF | MIC_HIP_PRED_GRAD))) return MIC_ERR_ARGS;
        deep |= units[i].max_value >= (1u << 13);                        // depth 14+: tokens reach past 8192
    }
    if (max_px > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
    const int tier = (s->force_big || deep) ? 2 : 1;
    s->retry.kind = 0;
    if (tier == 1) { s->retry.kind = 1; s->retry.d_in = d_pixels; s->retry.units.assign(units, units + n); }
    return encode_enqueue_tier(s, d_pixels, units, n, tier);
}

// reads the units back; a tier-1 chain that reported MICD_INT_GROW somewhere is run again in tier 2 first
static int finish_units(mic_hip_session *s, int n) {
    for (int pass = 0; pass < 2; pass++) {
        if (!s->readback_queued) HIP_TRY(hipMemcpyAsync(s->h_units.data(), s->units.p, sizeof(MicUnit) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
        s->readback_queued = false;
        HIP_TRY(hipStreamSynchronize(s->stream));
        bool grow = false;
        for (int i = 0; i < n; i++) grow |= s->h_units[(size_t)i].status == MICD_INT_GROW;
        if (!grow) return MIC_OK;
        if (pass == 1 || s->retry.kind == 0) break;                      // (tier 2 never reports it; a caller that laid the units out itself has no retry)
        s->force_big = true;
        const mic_hip_session::Retry r = s->retry;                       // (the enqueue below rewrites it)
        int rc;
        if (r.kind == 1) rc = encode_enqueue_tier(s, (const uint16_t *)r.d_in, r.units.data(), n, 2);
        else rc = session_decode_enqueue_spans(s, (const uint8_t *)r.d_in, r.begins.data(), r.ends.data(), r.units.data(), n, (uint16_t *)r.d_out);
        if (rc) return rc;
    }
    for (int i = 0; i < n; i++) if (s->h_units[(size_t)i].status == MICD_INT_GROW) s->h_units[(size_t)i].status = MICD_ERR_INTERNAL;
    return MIC_OK;
}

// The tier is sticky, not permanent: a session that one escape-heavy (or 16-bit) batch sent to the worst-case slabs -- 42 bytes a pixel
// -- goes back to the ordinary ones when kTierCalm batches in a row would have fitted them (every unit's tokens, segments, alphabet
// and table inside the tier-1 capacities), and ensure() then returns the large slabs' memory.  Called with the descriptors read back.
static void tier_review(mic_hip_session *s, int n) {
    if (!s->force_big || s->tier != 2) return;
    bool fits = true;
    const size_t ts1 = tab_syms_tier(1);
    for (int i = 0; i < n && fits; i++) {
        const MicUnit &u = s->h_units[(size_t)i];
        if (u.mode != 0) { fits = false; break; }                       // (units laid out by the wavelet / temporal paths: theirs to decide)
        const size_t px = (size_t)std::max(u.w, 0) * (size_t)std::max(u.h, 0);
        const size_t ntok = std::max<size_t>(u.ntok, u.count);
        fits = u.status == MICD_OK && ntok + 64 <= tok_cap_tier(px, 1) && (size_t)u.nseg + 8 <= seg_cap_tier(px, 1) &&
               u.symbol_len <= ts1 && u.table_log <= 13 && u.max_value < (1u << 13);
    }
    if (!fits) { s->calm_batches = 0; return; }
    if (++s->calm_batches >= mic_hip_session::kTierCalm) { s->force_big = false; s->calm_batches = 0; s->shrink_pending = true; }
}

// what the batch just finished tells the next one (mic_launch.h: launch masks)
static void learn_classes(mic_hip_session *s, int n) {
    tier_review(s, n);
    if (s->learn_decode) {
        uint32_t seen = 0;
        for (int i = 0; i < n; i++) {
            const MicUnit &u = s->h_units[(size_t)i];
            const int c = mic_dec_cls(u.flavour, u.table_log, u.zero_bits);
            if (c >= 0 && u.comp_len - u.bits_off < (1u << 27)) seen |= 1u << c;
        }
        s->dec_classes.learn(seen);
    }
    if (s->learn_encode) {
        uint32_t seen = 0;
        for (int i = 0; i < n; i++) {
            const MicUnit &u = s->h_units[(size_t)i];
            const uint32_t tl = u.table_log;
            if (tl < MIC_MIN_TABLELOG || tl > MIC_MAX_TABLELOG) continue;                 // (the chain stopped in front of the tANS stage)
            if (tl == 14) seen |= MIC_ENC_CLS_TL14;
            else if (tl == 15) seen |= MIC_ENC_CLS_TL15;
            else if (tl == 16) seen |= MIC_ENC_CLS_TL16;
            else if (u.ntok <= 98304u && u.symbol_len <= 1024u) seen |= (tl <= 12 && u.symbol_len <= 512u) ? MIC_ENC_CLS_SMALL12 : MIC_ENC_CLS_SMALL13;   // te_small_class, mic_encode.hip
            else if (u.nstates == 2 && u.symbol_len <= 4096u) seen |= MIC_ENC_CLS_NARROW2 | (u.nstates_used != 2 ? MIC_ENC_CLS_WIDE : 0u);
            else seen |= MIC_ENC_CLS_WIDE;
        }
        s->enc_classes.learn(seen);
    }
    s->learn_decode = s->learn_encode = false;
}

int session_encode_finish(mic_hip_session *s, const uint8_t **d_blobs, uint64_t *h_offsets, int32_t *h_status, int32_t *h_nstates) {
    int n = s->n_last;
    if (n <= 0) return MIC_ERR_ARGS;
    { const int frc = finish_units(s, n); if (frc) return frc; }
    learn_classes(s, n);
    uint64_t total = 0;
    for (int i = 0; i < n; i++) {
        const MicUnit &u = s->h_units[(size_t)i];
        h_offsets[i] = total;
        if (u.status == MICD_OK) total += u.blob_len;
        h_status[i] = u.status;
        if (h_nstates) h_nstates[i] = u.nstates_used;
    }
    h_offsets[n] = total;
    const bool packed_already = s->pack_queued && (size_t)total + 16 <= s->pack_cap && s->pin_off.cap > (size_t)n && s->pin_off.p[n] == total;
    s->pack_queued = false;
    s->pack_hint = (size_t)total;
    if (!packed_already) {                         // a caller that launched its own chain, or a batch larger than the session's last
        int rc = s->packed.reserve((size_t)total + (size_t)total / 8 + ((size_t)64 << 10));   // (what the next enqueue of such a batch will ask for: no second reallocation)
        if (rc) return rc;
        mic_launch_pack((const MicUnit *)s->units.p, n, (uint64_t *)s->offsets.p, (uint8_t *)s->packed.p, (uint64_t)s->packed.cap, s->stream, nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
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
    if (n > 65535) return MIC_ERR_UNSUPPORTED;
    size_t max_px = 0;
    for (int i = 0; i < n; i++) {
        if (units[i].width <= 0 || units[i].height <= 0) return MIC_ERR_ARGS;
        max_px = std::max(max_px, (size_t)units[i].width * (size_t)units[i].height);
    }
    if (max_px > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
    for (int i = 0; i < n; i++) if (ends[i] < begins[i] || ends[i] - begins[i] > 0xFFFFFFF0ull) return MIC_ERR_ARGS;
    const int tier = s->force_big ? 2 : 1;                               // (a stream's alphabet is not known before its header is parsed)
    if (tier == 1) {                                                     // what a second run in tier 2 needs (the arrays may be the caller's temporaries)
        mic_hip_session::Retry r;
        r.kind = 2; r.d_in = d_blobs; r.d_out = d_pixels_out; r.units.assign(units, units + n);
        r.begins.assign(begins, begins + n); r.ends.assign(ends, ends + n);
        s->retry = std::move(r);
        begins = s->retry.begins.data(); ends = s->retry.ends.data(); units = s->retry.units.data();
    } else if (s->retry.kind != 0 && begins != s->retry.begins.data()) s->retry.kind = 0;
    int rc = s->ensure(n, max_px, tier);
    if (rc) return rc;
    { const int arc = s->h_units.assign((size_t)n, MicUnit{}); if (arc) return arc; }
    bool any_grad = false;
    uint32_t rows_kmask = 0;                                             // predictor classes (by width) this batch holds
    for (int i = 0; i < n; i++) {
        MicUnit &u = s->h_units[(size_t)i];
        rows_kmask |= mic_pred_bit(units[i].width);
        const uint64_t len = ends[i] - begins[i];
        u.comp_in = d_blobs + begins[i]; u.comp_len = (uint32_t)len;
        u.px_out = d_pixels_out + units[i].px_offset;
        u.w = units[i].width; u.h = units[i].height;
        u.pred = (units[i].nstates & MIC_HIP_PRED_GRAD) ? 1u : 0u; any_grad |= u.pred != 0;
        s->fill_workspace(u, i);
        u.tok_cap = (uint32_t)tok_cap_tier((size_t)u.w * (size_t)u.h, tier);
        u.sym_cap = (uint32_t)std::min<size_t>(tok_cap_for((size_t)u.w * (size_t)u.h) + 64, 0xFFFFFFF0u);   // (a bound on the symbols a frame can use, not a slab size: the symbol slab is idle on this side)
    }
    { const int urc = s->h_units.upload(s->units.p, (size_t)n, s->stream); if (urc) return urc; }
    HIP_TRY(hipMemsetAsync(s->flags.p, 0, s->flag_stride * (size_t)n, s->stream));
    s->timer.reset(s->stream);
    mic_launch_decode((MicUnit *)s->units.p, n, s->stream, s->variant | (any_grad ? MIC_VARIANT_GRAD : 0), &s->timer, (int *)s->cls.p, rows_kmask,
                      s->dec_classes.mask());
    HIP_TRY(hipGetLastError());
    s->begin_chain(n);
    s->learn_decode = true;
    HIP_TRY(hipMemcpyAsync(s->h_units.data(), s->units.p, sizeof(MicUnit) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
    s->readback_queued = true;
    return MIC_OK;
}

int session_decode_finish(mic_hip_session *s, int32_t *h_status) {
    int n = s->n_last;
    if (n <= 0) return MIC_ERR_ARGS;
    { const int frc = finish_units(s, n); if (frc) return frc; }
    learn_classes(s, n);
    for (int i = 0; i < n; i++) h_status[i] = s->h_units[(size_t)i].status;
    return MIC_OK;
}

// (the host-pointer batch and container entry points live in mic_host_io.hip)

size_t unit_ws_bytes_tier(size_t px, int tier) {
    const size_t tokc = tok_cap_tier(px, tier), ts = tab_syms_tier(tier);
    return tokc * 4 + blob_cap_tok(tokc) + seg_cap_tier(px, tier) * 8 + px / 8 + ts * 4 * 6 + ts * 2 + 8192;
}
size_t unit_ws_bytes(size_t px) { return unit_ws_bytes_tier(px, 2); }      // (the paths that lay their units out themselves: tier 2)
// Units of px pixels (x `mult` slabs each) a sub-batch of the tiered unit codec may hold: tier-1 slabs under the workspace ceiling --
// and tier-2 slabs, should the batch have to run again, inside nine tenths of what the device can give this session.
// `extra`: bytes per unit the caller holds beside the slabs (the host path's staging halves).
size_t batch_units_for(size_t px, size_t mult, size_t extra) {
    const size_t n1 = workspace_budget() / (mult * (unit_ws_bytes_tier(px, 1) + extra));
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); fr = (size_t)32 << 30; }
    const size_t held = tl_default ? tl_default->reserved_bytes() : 0;
    const size_t n2 = (size_t)((double)(fr / leased_here() + held) * 0.9) / (mult * (unit_ws_bytes_tier(px, 2) + extra));
    return std::max<size_t>(1, std::min(n1, n2));
}

void put_u32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace micapi

// ================================================================================ C ABI

extern "C" {

const char *mic_hip_version(void) { return "mic-hip 0.2 (gfx950)"; }

// device -> device copy on the calling thread's current device, complete on return (a session's result buffers are reused by its
// next call: a caller that keeps them copies them out)
int mic_hip_device_copy(void *d_dst, const void *d_src, size_t bytes) try {
    if ((!d_dst || !d_src) && bytes) return MIC_ERR_ARGS;
    if (bytes) HIP_TRY(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice));
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_set_device(int device) try { return mic_hip_set_devices(&device, 1); } catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_set_devices(const int *devices, int n) try {
    if (!devices || n <= 0 || n > 64) return MIC_ERR_ARGS;
    for (int i = 0; i < n; i++) {
        if (devices[i] < 0) return MIC_ERR_ARGS;
        const int rc = check_device(devices[i]);
        if (rc) return rc;
    }
    std::unique_lock<std::mutex> lk(g_mu);
    // the pools move with the list: wait for every leased session to come back, then drop those of devices that left it
    auto all_home = [&] { for (auto &kv : g_pools) if ((int)kv.second.free_.size() != kv.second.made) return false; return true; };
    while (!all_home()) g_pool_cv.wait(lk);
    for (auto it = g_pools.begin(); it != g_pools.end();) {
        if (std::find(devices, devices + n, it->first) == devices + n) {
            for (mic_hip_session *p : it->second.free_) { (void)p->activate(); p->release(); delete p; }
            it = g_pools.erase(it);
        } else ++it;
    }
    g_devices.assign(devices, devices + n);
    if (g_device != devices[0]) g_device_ok = false;
    g_device = devices[0];
    return ensure_device_locked();
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_get_devices(int *devices, int cap) try {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < (int)g_devices.size() && i < cap; i++) if (devices) devices[i] = g_devices[(size_t)i];
    return (int)g_devices.size();
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

const char *mic_hip_device_name(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (ensure_device_locked() != MIC_OK) return "";
    return g_device_name.c_str();
}

int mic_hip_compress_frame(const uint16_t *pixels, int width, int height, uint16_t max_value, int nstates,
                           uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!pixels || !out || !out_len || width <= 0 || height <= 0) return MIC_ERR_ARGS;
    mic_hip_enc_job j{};
    j.pixels = pixels; j.width = width; j.height = height; j.max_value = max_value; j.nstates = (uint16_t)nstates;
    j.out = out; j.out_cap = out_cap;
    if (!(nstates == 2 || nstates == 4 || nstates == 8)) return MIC_ERR_ARGS;
    int rc = mic_hip_compress_batch(&j, 1);
    if (rc) return rc;
    if (j.status == MIC_OK) *out_len = j.out_len;
    return j.status;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_decompress_frame(const uint8_t *compressed, size_t compressed_len, uint16_t *pixels_out, int width, int height) try {
    if (!compressed || !pixels_out || width <= 0 || height <= 0) return MIC_ERR_ARGS;
    mic_hip_dec_job j{};
    j.compressed = compressed; j.compressed_len = compressed_len; j.pixels_out = pixels_out; j.width = width; j.height = height;
    int rc = mic_hip_decompress_batch(&j, 1);
    if (rc) return rc;
    return j.status;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// ---- bare FSE stage (fsecompressu16.go:19, fse2state.go:22/102, fse4state.go:24, fse8state.go:31, rans8state.go:31)
int mic_hip_fse_compress_u16(const uint16_t *symbols, size_t n, int flavour, uint8_t *out, size_t out_cap, size_t *out_len) try {
    return mic_hip_fse_compress_u16_ex(symbols, n, flavour, 0, out, out_cap, out_len);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_fse_compress_u16_ex(const uint16_t *symbols, size_t n, int flavour, int table_log, uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!symbols || !out || !out_len) return MIC_ERR_ARGS;
    if (!(flavour == 1 || flavour == 2 || flavour == 4 || flavour == 8 || flavour == 108)) return MIC_ERR_ARGS;
    if (table_log < 0 || table_log > MIC_MAX_TABLELOG) return MIC_ERR_ARGS;   // prepare(): "tableLog (%d) > maxTableLog (%d)", fseu16.go:136-138
    if (n <= 1) return MIC_ERR_INCOMPRESSIBLE;                          // first gate of every FSECompressU16* variant
    if (n > ((size_t)1 << 30)) return MIC_ERR_UNSUPPORTED;
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    mic_hip_session *s = cur_default();
    const size_t px = (n + 3) / 4 + 16;
    if ((rc = s->ensure(1, px))) return rc;
    if ((rc = s->io_px.reserve(n * 2 + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(s->io_px.p, symbols, n * 2, hipMemcpyHostToDevice, s->stream));
    { const int arc = s->h_units.assign(1, MicUnit{}); if (arc) return arc; }
    MicUnit &u = s->h_units[0];
    u.px_in = (const uint16_t *)s->io_px.p; u.w = (int32_t)n; u.h = 1; u.max_value = 0;
    u.nstates = (uint16_t)flavour; u.mode = 1; u.no_fallback = 1; u.req_tl = (uint32_t)table_log;
    s->fill_workspace(u, 0);
    { const int urc = s->h_units.upload(s->units.p, 1, s->stream); if (urc) return urc; }
    if ((rc = s->prepare_hist(1))) return rc;
    mic_launch_encode((MicUnit *)s->units.p, 1, s->stream, s->variant, nullptr);
    if (hipGetLastError() != hipSuccess) { s->hist_unknown(); return MIC_ERR_DEVICE; }
    s->begin_chain(1);
    uint64_t offs[2]; int32_t st = 0, ns = 0; const uint8_t *d_blobs = nullptr;
    if ((rc = session_encode_finish(s, &d_blobs, offs, &st, &ns))) return rc;
    if (st != MIC_OK) return st;
    const size_t len = (size_t)offs[1];
    if (len > out_cap) return MIC_ERR_CAPACITY;
    HIP_TRY(hipMemcpy(out, d_blobs, len, hipMemcpyDeviceToHost));
    *out_len = len;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_fse_decompress_u16_auto(const uint8_t *in, size_t in_len, uint16_t *out, size_t out_cap, size_t *out_n) try {
    return mic_hip_fse_decompress_u16_ex(in, in_len, 0, out, out_cap, out_n);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// ScratchU16.DecompressLimit (fseu16.go:87-91): the reference compares len(OutU16) with the limit every time its 65536-symbol ring
// wraps (fse2state.go:249/283, fse4state.go:246/..., fse8state.go, fsedecompressu16.go:318/353) and, for 1-state streams, once more
// at the end (fsedecompressu16.go:372): an N-state stream of `count` symbols fails iff floor(count / 65536) * 65536 >= limit,
// a 1-state stream iff its symbol count >= limit.  0 = the default, 2 GiB - 1.
int mic_hip_fse_decompress_u16_ex(const uint8_t *in, size_t in_len, int64_t decompress_limit, uint16_t *out, size_t out_cap, size_t *out_n) try {
    if (decompress_limit < 0) return MIC_ERR_ARGS;
    const uint64_t limit = decompress_limit ? (uint64_t)decompress_limit : ((2ull << 30) - 1);
    if (in && in_len >= 6 && in[0] == 0xFF && (in[1] == 0x02 || in[1] == 0x04 || in[1] == 0x84 || in[1] == 0x08)) {
        const uint64_t count = (uint64_t)in[2] | ((uint64_t)in[3] << 8) | ((uint64_t)in[4] << 16) | ((uint64_t)in[5] << 24);
        if (count >= 65536 && (count / 65536) * 65536 >= limit) return MIC_ERR_CAPACITY;   // "output size (%d) > DecompressLimit (%d)"
    }
    if (!in || !out || !out_n || in_len == 0) return in && in_len == 0 ? MIC_ERR_CORRUPT : MIC_ERR_ARGS;
    if (in_len > 0xFFFFFFF0ull || out_cap > ((size_t)1 << 30)) return MIC_ERR_UNSUPPORTED;
    DefaultLease lease;
    int rc = lease.acquire();
    if (rc) return rc;
    mic_hip_session *s = cur_default();
    const size_t px = (out_cap + 3) / 4 + 16;
    if ((rc = s->ensure(1, px))) return rc;
    if ((rc = s->io_comp.reserve(in_len + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(s->io_comp.p, in, in_len, hipMemcpyHostToDevice, s->stream));
    { const int arc = s->h_units.assign(1, MicUnit{}); if (arc) return arc; }
    MicUnit &u = s->h_units[0];
    u.comp_in = (const uint8_t *)s->io_comp.p; u.comp_len = (uint32_t)in_len; u.w = 1; u.h = 1; u.mode = 1;
    s->fill_workspace(u, 0);
    u.tok_cap = (uint32_t)std::min<size_t>(out_cap, u.tok_cap);
    { const int urc = s->h_units.upload(s->units.p, 1, s->stream); if (urc) return urc; }
    mic_launch_decode((MicUnit *)s->units.p, 1, s->stream, s->variant, nullptr, (int *)s->cls.p);
    HIP_TRY(hipGetLastError());
    s->begin_chain(1);
    int32_t st = 0;
    if ((rc = session_decode_finish(s, &st))) return rc;
    if (st != MIC_OK) return st;
    const size_t n = s->h_units[0].ntok;
    if (s->h_units[0].flavour == 1 && (uint64_t)n >= limit) return MIC_ERR_CAPACITY;        // fsedecompressu16.go:372
    if (n > out_cap) return MIC_ERR_CAPACITY;
    if (n) HIP_TRY(hipMemcpy(out, s->h_units[0].tok, n * 2, hipMemcpyDeviceToHost));
    *out_n = n;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// ---- PICS / MIC2 headers (the container codecs themselves: mic_host_io.hip) ------------------------
int mic_hip_pics_info(const uint8_t *c, size_t len, int *width, int *height, int *num_strips, int *strip_height) try {
    if (!c) return MIC_ERR_ARGS;
    if (len < 20 || memcmp(c, "PICS", 4) != 0) return MIC_ERR_CORRUPT;   // parallelstrips.go:271-273
    int w = (int)get_u32(c + 4), h = (int)get_u32(c + 8), n = (int)get_u32(c + 12), sh = (int)get_u32(c + 16);
    if (n < 0 || (size_t)n > (len - 20) / 8) return MIC_ERR_CORRUPT;     // truncated header, :281-283
    if (w <= 0 || h <= 0 || n <= 0 || sh <= 0) return MIC_ERR_CORRUPT;   // :284-286
    if (width) *width = w; if (height) *height = h; if (num_strips) *num_strips = n; if (strip_height) *strip_height = sh;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_mic2_compress_temporal(const uint16_t *frames, int width, int height, int nframes, uint16_t max_value,
                                   uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!frames || !out || !out_len || width <= 0 || height <= 0 || nframes <= 0) return MIC_ERR_ARGS;
    return mic2_temporal_compress(frames, width, height, nframes, max_value, out, out_cap, out_len);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_mic2_info(const uint8_t *c, size_t len, int *width, int *height, int *nframes, int *temporal) try {
    if (!c) return MIC_ERR_ARGS;
    if (len < 20 || memcmp(c, "MIC2", 4) != 0) return MIC_ERR_CORRUPT;   // multiframe.go:96-103
    int w = (int)get_u32(c + 4), h = (int)get_u32(c + 8), n = (int)get_u32(c + 12);
    if (n < 0 || (size_t)n > (len - 20) / 8) return MIC_ERR_CORRUPT;     // :112-116
    if (width) *width = w; if (height) *height = h; if (nframes) *nframes = n; if (temporal) *temporal = (c[16] & 0x02) != 0;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// DecompressFrame (multiframecompress.go:266-315): one frame of a MIC2 file.  Independent mode decodes just that
// frame; temporal mode needs frames 0..idx (all their residual streams are decoded at once, mic_temporal.hip).
int mic_hip_mic2_decompress_frame(const uint8_t *c, size_t len, int frame_idx, uint16_t *pixels_out, size_t pixels_cap) try {
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
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// ---- sessions ------------------------------------------------------------------------------------
int mic_hip_session_create_on(int device, mic_hip_session **out, int max_units, size_t max_px_per_unit) try {
    if (!out || max_units <= 0 || max_px_per_unit == 0) return MIC_ERR_ARGS;
    int rc = check_device(device);
    if (rc) return rc;
    mic_hip_session *s = new mic_hip_session();
    s->device = device;
    if ((rc = s->activate())) { delete s; return rc; }
    rc = s->ensure(max_units, max_px_per_unit, 1);                      // tier-1 slabs; a batch that needs the big ones grows them
    if (rc) { s->release(); delete s; return rc; }
    *out = s;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_session_create(mic_hip_session **out, int max_units, size_t max_px_per_unit) try {
    int dev;
    { std::lock_guard<std::mutex> lk(g_mu); dev = g_device; }
    return mic_hip_session_create_on(dev, out, max_units, max_px_per_unit);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_session_device(mic_hip_session *s) try { return s ? s->device : -1; } catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
// device memory the session holds right now (workspace slabs, staging, stores), and whether a batch has needed the tier-2 slabs
size_t mic_hip_session_workspace_bytes(mic_hip_session *s, int *tier2) {
    if (!s) return 0;
    if (tier2) *tier2 = s->force_big ? 1 : 0;
    return s->reserved_bytes();
}
void mic_hip_session_destroy(mic_hip_session *s) { if (s) { (void)s->activate(); s->release(); delete s; } }
void *mic_hip_session_stream(mic_hip_session *s) { return s ? (void *)s->stream : nullptr; }

int mic_hip_session_encode_enqueue(mic_hip_session *s, const uint16_t *d_pixels, const mic_hip_unit *units, int n) try {
    if (!s || !d_pixels || !units) return MIC_ERR_ARGS;
    { const int arc = s->activate(); if (arc) return arc; }
    return session_encode_enqueue(s, d_pixels, units, n);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_session_encode_finish(mic_hip_session *s, const uint8_t **d_blobs, uint64_t *h_offsets, int32_t *h_status, int32_t *h_nstates) try {
    if (!s || !h_offsets || !h_status) return MIC_ERR_ARGS;
    { const int arc = s->activate(); if (arc) return arc; }
    return session_encode_finish(s, d_blobs, h_offsets, h_status, h_nstates);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_session_encode(mic_hip_session *s, const uint16_t *d_pixels, const mic_hip_unit *units, int n,
                           const uint8_t **d_blobs, uint64_t *h_offsets, int32_t *h_status, int32_t *h_nstates) try {
    int rc = mic_hip_session_encode_enqueue(s, d_pixels, units, n);
    if (rc) return rc;
    return mic_hip_session_encode_finish(s, d_blobs, h_offsets, h_status, h_nstates);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_session_decode_enqueue(mic_hip_session *s, const uint8_t *d_blobs, const uint64_t *h_offsets,
                                   const mic_hip_unit *units, int n, uint16_t *d_pixels_out) try {
    if (!s || !d_blobs || !h_offsets || !units || !d_pixels_out) return MIC_ERR_ARGS;
    { const int arc = s->activate(); if (arc) return arc; }
    return session_decode_enqueue(s, d_blobs, h_offsets, units, n, d_pixels_out);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_session_decode_finish(mic_hip_session *s, int32_t *h_status) try {
    if (!s || !h_status) return MIC_ERR_ARGS;
    { const int arc = s->activate(); if (arc) return arc; }
    return session_decode_finish(s, h_status);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_session_decode(mic_hip_session *s, const uint8_t *d_blobs, const uint64_t *h_offsets, const mic_hip_unit *units, int n,
                           uint16_t *d_pixels_out, int32_t *h_status) try {
    int rc = mic_hip_session_decode_enqueue(s, d_blobs, h_offsets, units, n, d_pixels_out);
    if (rc) return rc;
    return mic_hip_session_decode_finish(s, h_status);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
// debug probe (not part of the public header): raw result fields of unit i after a *_finish
int mic_hip_debug_unit(mic_hip_session *s, int i, uint32_t *out8) try {
    if (!s || i < 0 || i >= s->n_last) return MIC_ERR_ARGS;
    const MicUnit &u = s->h_units[(size_t)i];
    out8[0] = u.ntok; out8[1] = u.blob_len; out8[2] = u.table_log; out8[3] = u.symbol_len;
    out8[4] = u.max_count; out8[5] = u.hdr_len; out8[6] = u.zero_bits; out8[7] = u.flavour;
    out8[8] = u.count; out8[9] = u.bits_off; out8[10] = (uint32_t)u.nstates_used; out8[11] = (uint32_t)u.status; out8[12] = u.nseg; out8[13] = u.nsym; out8[14] = u.seg_cap;
    for (int k = 0; k < 16; k++) out8[16 + k] = u.dbg[k];
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
// debug probe (not in the public header): bytes of unit i's histogram slab after a *_finish (LS_DEBUG builds dump there)
int mic_hip_debug_fetch_hist(mic_hip_session *s, int i, void *dst, size_t bytes) try {
    if (!s || i < 0 || i >= s->n_last || bytes > kSym * 4) return MIC_ERR_ARGS;
    HIP_TRY(hipMemcpy(dst, s->h_units[(size_t)i].hist, bytes, hipMemcpyDeviceToHost));
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
// debug probe (not in the public header): the first n u16 of unit i's token slab after a *_finish
int mic_hip_debug_fetch_tok(mic_hip_session *s, int i, void *dst, size_t n) try {
    if (!s || i < 0 || i >= s->n_last || n > s->h_units[(size_t)i].tok_cap) return MIC_ERR_ARGS;
    HIP_TRY(hipMemcpy(dst, s->h_units[(size_t)i].tok, n * 2, hipMemcpyDeviceToHost));
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_session_set_timing(mic_hip_session *s, int enabled) try {
    if (!s) return MIC_ERR_ARGS;
    s->timer.enabled = enabled != 0;
    s->timer.accumulate = enabled == 2;                                   // 2: sum over every launch chain until the next set_timing
    s->timer.clear();
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_session_last_timings(mic_hip_session *s, const char **names, float *ms, int cap) try {
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
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

}  // extern "C"
