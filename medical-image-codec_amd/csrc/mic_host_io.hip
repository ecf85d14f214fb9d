// mic_host_io.hip -- the entry points a cgo caller uses: host buffers in, host buffers out.
//
// The reference hands Go slices to its C codec (ojph/mic_c.go:26-39, :169-185) and fans strips / frames out over goroutines
// (parallelstrips.go:77-93, multiframecompress.go:201-203).  Here a call
//   * leases a session of the default pool (mic_api.hip: concurrent callers run on different sessions),
//   * cuts its units into sub-batches and runs them as a three-stage pipeline -- sub-batch k + 1 comes up over PCIe while k is
//     coded and k - 1 goes down -- with two halves of every staging buffer,
//   * moves bytes through the transfer engine below: a caller's buffer is ordinary pageable memory (a Go slice), which the
//     runtime can only DMA through its own small bounce buffer; a handful of worker threads copy 4 MiB chunks into pinned slots
//     and DMA them on their own streams, so the link stays busy.  Memory from mic_hip_host_alloc (or any registered memory) is
//     pinned already and is DMA-ed in place.
// There is no CPU codec here: the host copies bytes and writes container headers.
#include "mic_session.h"
#include <chrono>

#include <atomic>
#include <deque>
#include <thread>

namespace micapi {
void put_u32(uint8_t *p, uint32_t v);
uint32_t get_u32(const uint8_t *p);
}

namespace {

// ============================================================================ transfer engine
constexpr size_t kChunk = (size_t)4 << 20;        // bytes per pinned slot
constexpr size_t kDirectChunk = (size_t)64 << 20; // pinned user memory: pieces this large, spread over the workers' streams
constexpr size_t kInline = (size_t)128 << 10;     // pageable transfers up to this size: a plain hipMemcpy on the calling thread

struct IoReq {                                    // a set of transfers the caller waits for together
    std::atomic<int> pending{0};
    std::atomic<int> error{0};
    std::mutex mu; std::condition_variable cv;
    void add(int n) { std::lock_guard<std::mutex> lk(mu); pending.fetch_add(n); }
    void done(bool ok) {                              // (the count goes down under the lock: the waiter may free the request right after)
        std::lock_guard<std::mutex> lk(mu);
        if (!ok) error.store(1);
        if (pending.fetch_sub(1) == 1) cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return pending.load() == 0; });
        return error.load() ? MIC_ERR_DEVICE : MIC_OK;
    }
};

struct IoChunk { IoReq *req; int device; void *dev; void *host; size_t bytes; bool to_device; bool direct; };

bool host_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

class IoPool {
public:
    static IoPool &get() { static IoPool p; return p; }
    void push(const IoChunk &c) {
        { std::lock_guard<std::mutex> lk(mu_); q_[c.to_device ? 0 : 1].push_back(c); }
        cv_.notify_all();
    }
    int threads() const { return (int)th_.size(); }
private:
    struct Slot { void *pin = nullptr; hipEvent_t ev = nullptr; bool busy = false; IoReq *req = nullptr; void *host_dst = nullptr; size_t bytes = 0; };
    struct PerDev { hipStream_t st = nullptr; Slot slot[2]; int next = 0; bool ok = false; };
    std::vector<std::thread> th_;
    std::deque<IoChunk> q_[2];                     // uploads, downloads: half the workers look at one first, half at the other, so a
                                                   // download queued behind a long upload starts at once and the link runs both ways
    std::mutex mu_; std::condition_variable cv_;
    bool stop_ = false;

    IoPool() {
        const char *e = getenv("MIC_HIP_IO_THREADS");
        int n = e ? atoi(e) : 0;
        if (n <= 0) n = (int)std::min<unsigned>(8u, std::max<unsigned>(2u, std::thread::hardware_concurrency() / 2));
        n = std::min(n, 32);
        for (int i = 0; i < n; i++) th_.emplace_back([this, i] { run(i & 1); });
    }
    ~IoPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) if (t.joinable()) t.join();
    }
    static void retire(Slot &s) {
        if (!s.busy) return;
        const bool ok = hipEventSynchronize(s.ev) == hipSuccess;
        if (ok && s.host_dst) memcpy(s.host_dst, s.pin, s.bytes);
        s.busy = false;
        s.req->done(ok);
    }
    // a slot whose transfer has completed is retired at once -- its request is signalled (and a pageable download copied out) before
    // the worker takes more work: waiting for the slot's next use held a request's last chunks back for two more submissions, and
    // with callers on several devices for as long as the other device's request kept the worker fed
    static void retire_finished(Slot &s) {
        if (s.busy && hipEventQuery(s.ev) == hipSuccess) retire(s);
        else (void)hipGetLastError();                  // (hipErrorNotReady is not an error to keep)
    }
    void run(int first) {
        std::vector<PerDev> devs;
        for (;;) {
            IoChunk c;
            bool have = false;
            for (auto &d : devs) { retire_finished(d.slot[0]); retire_finished(d.slot[1]); }
            {
                std::unique_lock<std::mutex> lk(mu_);
                bool inflight = false;
                for (auto &d : devs) inflight |= d.slot[0].busy || d.slot[1].busy;
                auto none = [&] { return q_[0].empty() && q_[1].empty(); };
                if (none() && !inflight && !stop_) cv_.wait(lk, [&] { return stop_ || !none(); });
                std::deque<IoChunk> &q = !q_[first].empty() ? q_[first] : q_[first ^ 1];
                if (!q.empty()) { c = q.front(); q.pop_front(); have = true; }
                else if (stop_ && !inflight) return;
            }
            if (!have) {                                   // nothing new: finish what is in flight
                for (auto &d : devs) { retire(d.slot[0]); retire(d.slot[1]); }
                continue;
            }
            if ((int)devs.size() <= c.device) devs.resize((size_t)c.device + 1);
            PerDev &d = devs[(size_t)c.device];
            bool ok = hipSetDevice(c.device) == hipSuccess;
            if (ok && !d.ok) {
                // a stream of the HIGHEST priority: the runtime keeps a hardware queue per priority level, and a transfer on an
                // ordinary stream can land on the queue the session's kernels sit in and wait behind them (seen: the upload of
                // sub-batch k + 1 and the download of k - 1 only made progress between two decodes)
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
                ok = hipStreamCreateWithPriority(&d.st, hipStreamNonBlocking, hi) == hipSuccess;
                for (int i = 0; ok && i < 2; i++)
                    ok = hipHostMalloc(&d.slot[i].pin, kChunk, hipHostMallocDefault) == hipSuccess &&
                         hipEventCreateWithFlags(&d.slot[i].ev, hipEventDisableTiming) == hipSuccess;
                d.ok = ok;
            }
            if (!ok) { c.req->done(false); continue; }
            Slot &s = d.slot[d.next]; d.next ^= 1;
            retire(s);                                     // (its DMA had the other slot's copy to complete in)
            hipError_t e;
            if (c.direct) {
                e = hipMemcpyAsync(c.to_device ? c.dev : c.host, c.to_device ? c.host : c.dev, c.bytes,
                                   c.to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, d.st);
                s.host_dst = nullptr;
            } else if (c.to_device) {
                memcpy(s.pin, c.host, c.bytes);
                e = hipMemcpyAsync(c.dev, s.pin, c.bytes, hipMemcpyHostToDevice, d.st);
                s.host_dst = nullptr;
            } else {
                e = hipMemcpyAsync(s.pin, c.dev, c.bytes, hipMemcpyDeviceToHost, d.st);
                s.host_dst = c.host;
            }
            if (e == hipSuccess) e = hipEventRecord(s.ev, d.st);
            if (e != hipSuccess) { c.req->done(false); continue; }
            s.busy = true; s.req = c.req; s.bytes = c.bytes;
        }
    }
};

// One host <-> device transfer, joined to `req`.  The device side must be ready when this is called (uploads: the target is
// not in use; downloads: the producing stream has been synchronised) and is complete after req.wait().
int io_submit(IoReq &req, int device, void *dev, const void *host, size_t bytes, bool to_device) {
    if (bytes == 0) return MIC_OK;
    void *h = const_cast<void *>(host);
    const bool pinned = host_is_pinned(host);
    if (!pinned && bytes <= kInline) {
        HIP_TRY(to_device ? hipMemcpy(dev, h, bytes, hipMemcpyHostToDevice) : hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost));
        return MIC_OK;
    }
    IoPool &pool = IoPool::get();
    const size_t step = pinned ? kDirectChunk : kChunk;
    const int n = (int)((bytes + step - 1) / step);
    req.add(n);
    for (int i = 0; i < n; i++) {
        const size_t off = (size_t)i * step, len = std::min(step, bytes - off);
        pool.push(IoChunk{ &req, device, (char *)dev + off, (char *)h + off, len, to_device, pinned });
    }
    return MIC_OK;
}

// ============================================================================ pipelined unit codec over host buffers
// A GROUP is what one destination buffer receives (a job, a PICS image, a MIC2 stack); its units are consecutive, their pixels
// lie back to back in the group's host buffer, their streams go back to back behind `hdr` bytes of the group's output.
struct EncGroup {
    const uint16_t *h_px; uint8_t *out; size_t out_cap, hdr;
    int first, n;
    size_t written = 0; int32_t status = MIC_OK;
    int failed = -1;                                  // the group's first unit that failed, counted from `first` ("strip %d", parallelstrips.go:97)
};
struct EncUnit {
    uint64_t px_off; int32_t w, h; uint16_t maxv, nstates; int group;
    int32_t status = MIC_OK, nstates_used = 0; size_t len = 0;
};
struct DecGroup {
    const uint8_t *h_comp; uint16_t *h_px;
    int first, n;
    int32_t status = MIC_OK;
    int failed = -1;
};
struct DecUnit {
    size_t comp_off, comp_len; uint64_t px_off; int32_t w, h; uint16_t flags; int group;
    int32_t status = MIC_OK;
};

// units [i0, i1) of the next sub-batch: under the workspace ceiling, and -- when the call is large enough to be worth a pipeline --
// about `target` units (the kernels want a couple of thousand units per launch: the tANS decode chain takes as long for ten
// streams as for 2304)
template <class U>
int next_cut(const std::vector<U> &units, int i0, size_t target) {
    const int n = (int)units.size();
    size_t max_px = 0, cap = 0; int i1 = i0;
    while (i1 < n) {
        const size_t px = (size_t)units[(size_t)i1].w * (size_t)units[(size_t)i1].h;
        const size_t mp = std::max(max_px, px);
        if (mp != max_px || cap == 0) cap = batch_units_for(mp, 1, 12 * mp);   // (+ the staging: two halves each of the pixels, the streams, the packed buffer)
        if (i1 > i0 && ((size_t)(i1 - i0 + 1) > cap || (size_t)(i1 - i0) >= target || i1 - i0 >= 65535)) break;
        max_px = mp; i1++;
    }
    return i1;
}
// Sub-batches of a call: the transfers of one overlap the kernels of its neighbours.  A decode costs the tANS chain's ~17 ms whatever
// its size (one serial chain per stream), so decode cuts into parts of >= 1152 units; encode kernels scale with their units (7 ms
// for 2304), so encode cuts into parts of >= 768 -- three at most either way: measured on 2304 strips in pinned buffers, encode /
// decode with 2 parts 75 / 92 ms (decode's default), 3 parts 78 / 94 (encode's default: 75 with decode at 2), 4 parts 88 / 106.
inline size_t pipeline_target(size_t n_units, bool encode) {
    const size_t min_units = encode ? 768 : 1152, max_parts = 3;
    static const char *ov = getenv("MIC_HIP_PIPELINE_PARTS");              // (experiments: force the number of parts)
    size_t parts = ov ? (size_t)std::max(1, atoi(ov)) : std::min(max_parts, n_units / min_units);
    if (parts < 2) return n_units ? n_units : 1;
    return (n_units + parts - 1) / parts;
}

int encode_groups(mic_hip_session *s, std::vector<EncGroup> &G, std::vector<EncUnit> &U) {
    const int n = (int)U.size();
    if (n == 0) return MIC_OK;
    const size_t target = pipeline_target((size_t)n, true);
    struct Sub { int i0, i1; IoReq up, down; std::vector<mic_hip_unit> units; size_t px = 0; };
    std::vector<std::unique_ptr<Sub>> subs;
    for (int i0 = 0; i0 < n;) { auto sb = std::make_unique<Sub>(); sb->i0 = i0; sb->i1 = next_cut(U, i0, target); i0 = sb->i1; subs.push_back(std::move(sb)); }
    DevBuf *in[2] = { &s->io_px, &s->io_px2 };
    int rc = MIC_OK;
    auto upload = [&](Sub &sb, int half) -> int {
        sb.units.resize((size_t)(sb.i1 - sb.i0));
        size_t off = 0;
        for (int i = sb.i0; i < sb.i1; i++) {
            const EncUnit &u = U[(size_t)i];
            sb.units[(size_t)(i - sb.i0)] = mic_hip_unit{ off, u.w, u.h, u.maxv, u.nstates };
            off += (size_t)u.w * (size_t)u.h;
        }
        sb.px = off;
        int r = in[half]->reserve(off * 2 + 64);
        if (r) return r;
        // one transfer per run of units that are neighbours in one host buffer
        size_t doff = 0;
        for (int i = sb.i0; i < sb.i1;) {
            const EncGroup &g = G[(size_t)U[(size_t)i].group];
            int j = i; size_t run = 0;
            while (j < sb.i1 && U[(size_t)j].group == U[(size_t)i].group && U[(size_t)j].px_off == U[(size_t)i].px_off + run) {
                run += (size_t)U[(size_t)j].w * (size_t)U[(size_t)j].h; j++;
            }
            if ((r = io_submit(sb.up, s->device, (uint16_t *)in[half]->p + doff, g.h_px + U[(size_t)i].px_off, run * 2, true))) return r;
            doff += run; i = j;
        }
        return MIC_OK;
    };
    if ((rc = upload(*subs[0], 0))) { (void)subs[0]->up.wait(); return rc; }
    for (size_t k = 0; k < subs.size() && rc == MIC_OK; k++) {
        Sub &sb = *subs[k];
        const int half = (int)(k & 1);
        rc = sb.up.wait();
        if (rc == MIC_OK && k + 1 < subs.size()) rc = upload(*subs[k + 1], half ^ 1);
        const int nb = sb.i1 - sb.i0;
        std::vector<uint64_t> offs((size_t)nb + 1); std::vector<int32_t> st((size_t)nb), ns((size_t)nb);
        const uint8_t *d_blobs = nullptr;
        if (rc == MIC_OK) rc = session_encode_enqueue(s, (const uint16_t *)in[half]->p, sb.units.data(), nb);
        if (rc == MIC_OK) rc = session_encode_finish(s, &d_blobs, offs.data(), st.data(), ns.data());
        if (k >= 1) { const int r2 = subs[k - 1]->down.wait(); if (rc == MIC_OK) rc = r2; }   // (frees the packed buffer the next finish writes)
        if (rc != MIC_OK) break;
        std::swap(s->packed, s->packed2);                  // d_blobs stays where it is while the next sub-batch packs into the other half
        for (int i = sb.i0; i < sb.i1;) {                  // per group: its streams of this sub-batch, one transfer
            EncGroup &g = G[(size_t)U[(size_t)i].group];
            int j = i;
            while (j < sb.i1 && U[(size_t)j].group == U[(size_t)i].group) j++;
            size_t bytes = 0;
            for (int q = i; q < j; q++) {
                EncUnit &u = U[(size_t)q];
                u.status = st[(size_t)(q - sb.i0)]; u.nstates_used = ns[(size_t)(q - sb.i0)];
                u.len = (size_t)(offs[(size_t)(q - sb.i0) + 1] - offs[(size_t)(q - sb.i0)]);
                if (u.status != MIC_OK && g.status == MIC_OK) { g.status = u.status; g.failed = q - g.first; }   // the first failing unit names the error
                bytes += u.len;
            }
            if (g.status == MIC_OK) {
                if (g.hdr + g.written + bytes > g.out_cap) g.status = MIC_ERR_CAPACITY;
                else {
                    rc = io_submit(sb.down, s->device, const_cast<uint8_t *>(d_blobs) + offs[(size_t)(i - sb.i0)], g.out + g.hdr + g.written, bytes, false);
                    g.written += bytes;
                    if (rc) break;
                }
            }
            i = j;
        }
    }
    for (auto &sb : subs) { const int r2 = sb->up.wait(); const int r3 = sb->down.wait(); if (rc == MIC_OK) rc = r2 ? r2 : r3; }
    return rc;
}

int decode_groups(mic_hip_session *s, std::vector<DecGroup> &G, std::vector<DecUnit> &U) {
    const int n = (int)U.size();
    if (n == 0) return MIC_OK;
    const size_t target = pipeline_target((size_t)n, false);
    struct Sub { int i0, i1; IoReq up, down; std::vector<mic_hip_unit> units; std::vector<uint64_t> begins, ends; size_t px = 0; };
    std::vector<std::unique_ptr<Sub>> subs;
    for (int i0 = 0; i0 < n;) { auto sb = std::make_unique<Sub>(); sb->i0 = i0; sb->i1 = next_cut(U, i0, target); i0 = sb->i1; subs.push_back(std::move(sb)); }
    DevBuf *in[2] = { &s->io_comp, &s->io_comp2 }, *outb[2] = { &s->io_px, &s->io_px2 };
    int rc = MIC_OK;
    auto upload = [&](Sub &sb, int half) -> int {
        const int nb = sb.i1 - sb.i0;
        sb.units.resize((size_t)nb); sb.begins.resize((size_t)nb); sb.ends.resize((size_t)nb);
        // device layout: every run of streams that are neighbours in one host buffer keeps its shape, runs start 16-byte aligned
        size_t coff = 0, poff = 0;
        struct Run { size_t dev, host_off, len; const uint8_t *base; };
        std::vector<Run> runs;
        for (int i = sb.i0; i < sb.i1;) {
            const int gi = U[(size_t)i].group;
            const size_t start = U[(size_t)i].comp_off;
            size_t end = start;
            int j = i;
            while (j < sb.i1 && U[(size_t)j].group == gi && U[(size_t)j].comp_off >= end && U[(size_t)j].comp_off - end <= 64) {
                sb.begins[(size_t)(j - sb.i0)] = coff + (U[(size_t)j].comp_off - start);
                sb.ends[(size_t)(j - sb.i0)] = sb.begins[(size_t)(j - sb.i0)] + U[(size_t)j].comp_len;
                end = U[(size_t)j].comp_off + U[(size_t)j].comp_len; j++;
            }
            runs.push_back(Run{ coff, start, end - start, G[(size_t)gi].h_comp });
            coff += align_up(end - start, 16);
            i = j;
        }
        for (int i = sb.i0; i < sb.i1; i++) {
            const DecUnit &u = U[(size_t)i];
            sb.units[(size_t)(i - sb.i0)] = mic_hip_unit{ poff, u.w, u.h, 0, u.flags };
            poff += (size_t)u.w * (size_t)u.h;
        }
        sb.px = poff;
        int r = in[half]->reserve(coff + 64);
        if (r) return r;
        for (const Run &q : runs)
            if ((r = io_submit(sb.up, s->device, (uint8_t *)in[half]->p + q.dev, q.base + q.host_off, q.len, true))) return r;
        return MIC_OK;
    };
    // MIC_HIP_TRACE=1: the stages of the pipeline with their wall times on stderr (tools/: where a call's time goes)
    static const bool trace = getenv("MIC_HIP_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto mark = [&](const char *what, size_t k) {
        if (trace) fprintf(stderr, "[mic_hip decode] %7.2f ms  %s %zu\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what, k);
    };
    if ((rc = upload(*subs[0], 0))) { (void)subs[0]->up.wait(); return rc; }
    for (size_t k = 0; k < subs.size() && rc == MIC_OK; k++) {
        Sub &sb = *subs[k];
        const int half = (int)(k & 1);
        rc = sb.up.wait();
        mark("streams on the device", k);
        // this sub-batch decodes into the pixel half that sub-batch k - 2 is (was) being downloaded from
        if (k >= 2) { const int r2 = subs[k - 2]->down.wait(); if (rc == MIC_OK) rc = r2; mark("pixels of sub-batch on the host", k - 2); }
        if (rc == MIC_OK && k + 1 < subs.size()) rc = upload(*subs[k + 1], half ^ 1);
        if (rc == MIC_OK) rc = outb[half]->reserve(sb.px * 2 + 64);
        const int nb = sb.i1 - sb.i0;
        std::vector<int32_t> st((size_t)nb);
        if (rc == MIC_OK) rc = session_decode_enqueue_spans(s, (const uint8_t *)in[half]->p, sb.begins.data(), sb.ends.data(), sb.units.data(), nb, (uint16_t *)outb[half]->p);
        if (rc == MIC_OK) rc = session_decode_finish(s, st.data());
        mark("decoded", k);
        if (rc != MIC_OK) break;
        for (int i = sb.i0; i < sb.i1;) {
            DecGroup &g = G[(size_t)U[(size_t)i].group];
            int j = i;
            while (j < sb.i1 && U[(size_t)j].group == U[(size_t)i].group) j++;
            for (int q = i; q < j; q++) {
                U[(size_t)q].status = st[(size_t)(q - sb.i0)];
                if (U[(size_t)q].status != MIC_OK && g.status == MIC_OK) { g.status = U[(size_t)q].status; g.failed = q - g.first; }
            }
            if (g.status == MIC_OK) {                      // runs of units whose pixels are neighbours in the host buffer
                for (int q = i; q < j;) {
                    int e = q; size_t run = 0;
                    while (e < j && U[(size_t)e].px_off == U[(size_t)q].px_off + run) { run += (size_t)U[(size_t)e].w * (size_t)U[(size_t)e].h; e++; }
                    rc = io_submit(sb.down, s->device, (uint16_t *)outb[half]->p + sb.units[(size_t)(q - sb.i0)].px_offset, g.h_px + U[(size_t)q].px_off, run * 2, false);
                    if (rc) break;
                    q = e;
                }
                if (rc) break;
            }
            i = j;
        }
    }
    for (auto &sb : subs) { const int r2 = sb->up.wait(); const int r3 = sb->down.wait(); if (rc == MIC_OK) rc = r2 ? r2 : r3; }
    mark("all pixels on the host", subs.size());
    return rc;
}

// ============================================================================ several devices
// mic_hip_set_devices lists the GPUs the batch entry points may use.  A call's groups (jobs, images) are cut into one CONTIGUOUS
// shard per listed device, balanced by pixels -- the static assignment of the reference's fan-outs (parallelstrips.go:77-93,
// multiframecompress.go:186-209, wsicompress.go:126-145) -- and the shards run side by side, each on a thread of its own with a
// session of its device's pool, its own sub-batch pipeline and the transfer engine's per-device streams and pinned slots.  Results
// go straight into the caller's buffers: the caller's memory is what a rank-0 view would be, there is no gather.  (A device may be
// listed twice: two shards, two sessions of one GPU.)
// first[k] .. first[k + 1]: the items of shard k; boundary k is where the running weight first reaches k / shards of the total
void shard_plan(const uint64_t *w, int n, int shards, int *first) {
    uint64_t total = 0;
    for (int i = 0; i < n; i++) total += w[i] ? w[i] : 1;
    first[0] = 0;
    uint64_t run = 0; int i = 0;
    for (int k = 1; k < shards; k++) {
        const uint64_t goal = (uint64_t)(((unsigned __int128)total * (unsigned)k) / (unsigned)shards);
        while (i < n && run < goal) { run += w[i] ? w[i] : 1; i++; }
        first[k] = i;
    }
    first[shards] = n;
}

template <class Grp, class Unt, class Run>
int run_shards(std::vector<Grp> &G, std::vector<Unt> &U, Run run_one) {
    const std::vector<int> devs = default_devices();
    const int ng = (int)G.size();
    int shards = (int)std::min<size_t>(devs.size(), (size_t)ng);
    if (cur_default()) shards = 1;                    // (a nested call runs on the session its thread already holds)
    if (shards <= 1) {
        DefaultLease lease;
        const int rc = lease.acquire();
        return rc ? rc : run_one(lease.s, G, U);
    }
    std::vector<uint64_t> w((size_t)ng);
    for (int g = 0; g < ng; g++) {
        uint64_t px = 0;
        for (int q = G[(size_t)g].first; q < G[(size_t)g].first + G[(size_t)g].n; q++) px += (uint64_t)U[(size_t)q].w * (uint64_t)U[(size_t)q].h;
        w[(size_t)g] = px;
    }
    std::vector<int> first((size_t)shards + 1);
    shard_plan(w.data(), ng, shards, first.data());
    struct Shard { std::vector<Grp> G; std::vector<Unt> U; int rc = MIC_OK; };
    std::vector<Shard> S((size_t)shards);
    for (int k = 0; k < shards; k++) {
        const int g0 = first[(size_t)k], g1 = first[(size_t)k + 1];
        if (g0 == g1) continue;
        const int u0 = G[(size_t)g0].first, u1 = G[(size_t)g1 - 1].first + G[(size_t)g1 - 1].n;
        S[(size_t)k].G.assign(G.begin() + g0, G.begin() + g1);
        S[(size_t)k].U.assign(U.begin() + u0, U.begin() + u1);
        for (Grp &g : S[(size_t)k].G) g.first -= u0;
        for (Unt &u : S[(size_t)k].U) u.group -= g0;
    }
    auto work = [&](int k) {
        if (S[(size_t)k].G.empty()) return;
        DefaultLease lease;
        int rc = lease.acquire(devs[(size_t)k]);
        if (rc == MIC_OK) rc = run_one(lease.s, S[(size_t)k].G, S[(size_t)k].U);
        S[(size_t)k].rc = rc;
    };
    std::vector<std::thread> th;
    int started = 1;                                  // shards 1 .. started - 1 have a thread; what could not get one runs here, one after the other
    try {
        th.reserve((size_t)shards);
        for (; started < shards; started++) th.emplace_back(work, started);
    } catch (...) { }                                 // (no exception crosses the C ABI: a thread the system refuses costs overlap, not the call)
    work(0);
    for (int k = started; k < shards; k++) work(k);
    for (auto &t : th) t.join();
    int rc = MIC_OK;
    for (int k = 0; k < shards; k++) {
        if (S[(size_t)k].rc != MIC_OK && rc == MIC_OK) rc = S[(size_t)k].rc;
        const int g0 = first[(size_t)k];
        if (S[(size_t)k].G.empty()) continue;
        const int u0 = G[(size_t)g0].first;
        for (size_t i = 0; i < S[(size_t)k].G.size(); i++) { Grp g = S[(size_t)k].G[i]; g.first += u0; G[(size_t)g0 + i] = g; }
        for (size_t i = 0; i < S[(size_t)k].U.size(); i++) { Unt u = S[(size_t)k].U[i]; u.group += g0; U[(size_t)u0 + i] = u; }
    }
    return rc;
}
int encode_sharded(std::vector<EncGroup> &G, std::vector<EncUnit> &U) {
    return run_shards(G, U, [](mic_hip_session *s, std::vector<EncGroup> &g, std::vector<EncUnit> &u) { return encode_groups(s, g, u); });
}
int decode_sharded(std::vector<DecGroup> &G, std::vector<DecUnit> &U) {
    return run_shards(G, U, [](mic_hip_session *s, std::vector<DecGroup> &g, std::vector<DecUnit> &u) { return decode_groups(s, g, u); });
}

// the strips of a PICS image (parallelstrips.go:59-72)
inline void pics_geometry(int height, int num_strips, int &strip_h, int &actual) {
    if (num_strips > height) num_strips = height;                    // :62-67
    strip_h = (height + num_strips - 1) / num_strips;                // :70
    actual = (height + strip_h - 1) / strip_h;                       // :72
}

}  // namespace

namespace micapi {
int host_copy(int device, void *dev, void *host, size_t bytes, bool to_device) {
    IoReq req;
    const int rc = io_submit(req, device, dev, host, bytes, to_device);
    const int r2 = req.wait();
    return rc ? rc : r2;
}
}  // namespace micapi

// ================================================================================ C ABI
extern "C" {

// The cut of `n` weighted items into `shards` contiguous shards that the batch entry points use across the devices of
// mic_hip_set_devices (first[0 .. shards]: item i belongs to shard k iff first[k] <= i < first[k + 1]); no device needed.
int mic_hip_shard_plan(const uint64_t *weights, int n, int shards, int *first) try {
    if (!weights || !first || n < 0 || shards <= 0) return MIC_ERR_ARGS;
    shard_plan(weights, n, shards, first);
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// Pinned host memory for a caller's frame and stream buffers: the entry points below DMA such buffers in place instead of
// staging them through the transfer engine's slots.
void *mic_hip_host_alloc(size_t bytes) {
    if (ensure_device() != MIC_OK) return nullptr;                       // (makes the default device current; no session is held or made)
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void mic_hip_host_free(void *p) { if (p) (void)hipHostFree(p); }

int mic_hip_compress_batch(mic_hip_enc_job *jobs, int njobs) try {
    if (!jobs || njobs < 0) return MIC_ERR_ARGS;
    if (njobs == 0) return MIC_OK;
    int rc = ensure_device();
    if (rc) return rc;
    std::vector<EncGroup> G; std::vector<EncUnit> U; std::vector<int> job_of;
    for (int i = 0; i < njobs; i++) {
        mic_hip_enc_job &j = jobs[i];
        j.out_len = 0; j.nstates_used = 0;
        if (!j.pixels || !j.out || j.width <= 0 || j.height <= 0 || (size_t)j.width * (size_t)j.height > ((size_t)1 << 28) ||
            !(j.nstates == 2 || j.nstates == 4 || j.nstates == 8)) { j.status = MIC_ERR_ARGS; continue; }
        G.push_back(EncGroup{ j.pixels, j.out, j.out_cap, 0, (int)U.size(), 1 });
        U.push_back(EncUnit{ 0, j.width, j.height, j.max_value, j.nstates, (int)G.size() - 1 });
        job_of.push_back(i);
    }
    if ((rc = encode_sharded(G, U))) return rc;
    for (size_t k = 0; k < G.size(); k++) {
        mic_hip_enc_job &j = jobs[job_of[k]];
        j.status = G[k].status; j.nstates_used = U[k].nstates_used;
        j.out_len = G[k].status == MIC_OK ? G[k].written : 0;
    }
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_decompress_batch(mic_hip_dec_job *jobs, int njobs) try {
    if (!jobs || njobs < 0) return MIC_ERR_ARGS;
    if (njobs == 0) return MIC_OK;
    int rc = ensure_device();
    if (rc) return rc;
    std::vector<DecGroup> G; std::vector<DecUnit> U; std::vector<int> job_of;
    for (int i = 0; i < njobs; i++) {
        mic_hip_dec_job &j = jobs[i];
        if (!j.compressed || !j.pixels_out || j.width <= 0 || j.height <= 0 || j.compressed_len == 0 ||
            j.compressed_len > 0xFFFFFFF0ull || (size_t)j.width * (size_t)j.height > ((size_t)1 << 28)) {
            j.status = (j.compressed && j.compressed_len == 0) ? MIC_ERR_CORRUPT : MIC_ERR_ARGS; continue;
        }
        G.push_back(DecGroup{ j.compressed, j.pixels_out, (int)U.size(), 1 });
        U.push_back(DecUnit{ 0, j.compressed_len, 0, j.width, j.height, 0, (int)G.size() - 1 });
        job_of.push_back(i);
    }
    if ((rc = decode_sharded(G, U))) return rc;
    for (size_t k = 0; k < G.size(); k++) jobs[job_of[k]].status = G[k].status;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// ---- PICS (parallelstrips.go) ----------------------------------------------------------------
int mic_hip_pics_compress_batch(mic_hip_pics_enc_job *jobs, int njobs) try {
    if (!jobs || njobs < 0) return MIC_ERR_ARGS;
    if (njobs == 0) return MIC_OK;
    int rc = ensure_device();
    if (rc) return rc;
    std::vector<EncGroup> G; std::vector<EncUnit> U; std::vector<int> job_of;
    struct Geo { int strip_h, actual; };
    std::vector<Geo> geo;
    for (int i = 0; i < njobs; i++) {
        mic_hip_pics_enc_job &j = jobs[i];
        j.out_len = 0; j.failed_strip = -1;
        if (!j.pixels || !j.out || j.width <= 0 || j.height <= 0 || j.num_strips <= 0 || !(j.nstates == 2 || j.nstates == 4 || j.nstates == 8)) { j.status = MIC_ERR_ARGS; continue; }
        int strip_h, actual; pics_geometry(j.height, j.num_strips, strip_h, actual);
        const size_t header = 20 + (size_t)actual * 8;
        if ((size_t)j.width * (size_t)strip_h > ((size_t)1 << 28)) { j.status = MIC_ERR_UNSUPPORTED; continue; }
        if (j.out_cap < header) { j.status = MIC_ERR_CAPACITY; continue; }
        G.push_back(EncGroup{ j.pixels, j.out, j.out_cap, header, (int)U.size(), actual });
        for (int s = 0; s < actual; s++) {
            const int y0 = s * strip_h, y1 = std::min(j.height, y0 + strip_h);
            U.push_back(EncUnit{ (uint64_t)y0 * (uint64_t)j.width, j.width, y1 - y0, j.max_value, j.nstates, (int)G.size() - 1 });   // global maxValue for every strip, :88
        }
        job_of.push_back(i); geo.push_back(Geo{ strip_h, actual });
    }
    if ((rc = encode_sharded(G, U))) return rc;
    for (size_t k = 0; k < G.size(); k++) {
        mic_hip_pics_enc_job &j = jobs[job_of[k]];
        const EncGroup &g = G[k];
        j.status = g.status;                                                   // first failing strip, :95-99
        j.failed_strip = g.failed;                                             // "parallelstrips: strip %d: %w", :97
        if (j.status != MIC_OK) continue;
        if (g.written > 0xFFFFFFFFull) { j.status = MIC_ERR_UNSUPPORTED; continue; }
        uint8_t *out = j.out;
        memcpy(out, "PICS", 4);
        put_u32(out + 4, (uint32_t)j.width); put_u32(out + 8, (uint32_t)j.height);
        put_u32(out + 12, (uint32_t)geo[k].actual); put_u32(out + 16, (uint32_t)geo[k].strip_h);
        size_t off = 0;
        for (int s = 0; s < g.n; s++) {
            const size_t len = U[(size_t)(g.first + s)].len;
            put_u32(out + 20 + (size_t)s * 8, (uint32_t)off);
            put_u32(out + 24 + (size_t)s * 8, (uint32_t)len);
            off += len;
        }
        j.out_len = g.hdr + g.written;
    }
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_pics_compress(const uint16_t *pixels, int width, int height, uint16_t max_value, int num_strips, int nstates,
                          uint8_t *out, size_t out_cap, size_t *out_len) try {
    return mic_hip_pics_compress_ex(pixels, width, height, max_value, num_strips, nstates, out, out_cap, out_len, nullptr);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_pics_compress_ex(const uint16_t *pixels, int width, int height, uint16_t max_value, int num_strips, int nstates,
                             uint8_t *out, size_t out_cap, size_t *out_len, int *failed_strip) try {
    if (failed_strip) *failed_strip = -1;
    if (!pixels || !out || !out_len || width <= 0 || height <= 0 || num_strips <= 0) return MIC_ERR_ARGS;
    if (!(nstates == 2 || nstates == 4 || nstates == 8)) return MIC_ERR_ARGS;
    mic_hip_pics_enc_job j{};
    j.pixels = pixels; j.width = width; j.height = height; j.max_value = max_value; j.nstates = (uint16_t)nstates; j.num_strips = num_strips;
    j.out = out; j.out_cap = out_cap;
    const int rc = mic_hip_pics_compress_batch(&j, 1);
    if (rc) return rc;
    if (j.status == MIC_OK) *out_len = j.out_len;
    else if (failed_strip) *failed_strip = j.failed_strip;
    return j.status;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_pics_decompress_batch(mic_hip_pics_dec_job *jobs, int njobs) try {
    if (!jobs || njobs < 0) return MIC_ERR_ARGS;
    if (njobs == 0) return MIC_OK;
    int rc = ensure_device();
    if (rc) return rc;
    std::vector<DecGroup> G; std::vector<DecUnit> U; std::vector<int> job_of;
    for (int i = 0; i < njobs; i++) {
        mic_hip_pics_dec_job &j = jobs[i];
        j.failed_strip = -1;
        if (!j.compressed || !j.pixels_out) { j.status = MIC_ERR_ARGS; continue; }
        int w, h, n, sh;
        if ((j.status = mic_hip_pics_info(j.compressed, j.compressed_len, &w, &h, &n, &sh))) continue;
        if (w != j.width || h != j.height) { j.status = MIC_ERR_ARGS; continue; }
        const uint8_t *c = j.compressed; const size_t len = j.compressed_len;
        const size_t header = 20 + (size_t)n * 8;
        const size_t u0 = U.size();
        int32_t bad = MIC_OK; size_t covered = 0;
        for (int s = 0; s < n && bad == MIC_OK; s++) {
            const size_t so = get_u32(c + 20 + (size_t)s * 8), sl = get_u32(c + 24 + (size_t)s * 8);
            const size_t start = header + so, end = start + sl;
            if (end > len || start > end) { bad = MIC_ERR_CORRUPT; break; }          // :300-304
            const long y0 = (long)s * sh, y1 = std::min<long>(h, y0 + sh);
            if (y0 >= h) { bad = MIC_ERR_CORRUPT; break; }
            if (sl == 0) { bad = MIC_ERR_CORRUPT; break; }
            if ((size_t)w * (size_t)(y1 - y0) > ((size_t)1 << 28)) { bad = MIC_ERR_UNSUPPORTED; break; }
            U.push_back(DecUnit{ start, sl, (uint64_t)y0 * (uint64_t)w, w, (int32_t)(y1 - y0), 0, (int)G.size() });
            covered += (size_t)w * (size_t)(y1 - y0);
        }
        if (bad != MIC_OK) { U.resize(u0); j.status = bad; continue; }
        // pixels no strip writes come back as zeros, like the reference's make([]uint16, w*h) (parallelstrips.go:288): a header
        // whose strips do not cover the image is accepted there
        if (covered < (size_t)w * (size_t)h) memset(j.pixels_out, 0, (size_t)w * (size_t)h * 2);
        G.push_back(DecGroup{ c, j.pixels_out, (int)u0, n });
        job_of.push_back(i);
    }
    if ((rc = decode_sharded(G, U))) return rc;
    for (size_t k = 0; k < G.size(); k++) { jobs[job_of[k]].status = G[k].status; jobs[job_of[k]].failed_strip = G[k].failed; }   // "strip %d: %w", parallelstrips.go:316
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_pics_decompress(const uint8_t *c, size_t len, uint16_t *pixels_out, int width, int height) try {
    return mic_hip_pics_decompress_ex(c, len, pixels_out, width, height, nullptr);
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)
int mic_hip_pics_decompress_ex(const uint8_t *c, size_t len, uint16_t *pixels_out, int width, int height, int *failed_strip) try {
    if (failed_strip) *failed_strip = -1;
    if (!c || !pixels_out) return MIC_ERR_ARGS;
    mic_hip_pics_dec_job j{};
    j.compressed = c; j.compressed_len = len; j.pixels_out = pixels_out; j.width = width; j.height = height;
    const int rc = mic_hip_pics_decompress_batch(&j, 1);
    if (rc == MIC_OK && j.status != MIC_OK && failed_strip) *failed_strip = j.failed_strip;
    return rc ? rc : j.status;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

// ---- MIC2 independent mode (multiframe.go, multiframecompress.go:179-261) -----------------------
int mic_hip_mic2_compress(const uint16_t *frames, int width, int height, int nframes, uint16_t max_value,
                          uint8_t *out, size_t out_cap, size_t *out_len) try {
    if (!frames || !out || !out_len || width <= 0 || height <= 0 || nframes <= 0) return MIC_ERR_ARGS;
    const size_t npx = (size_t)width * (size_t)height;
    if (npx > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
    const size_t header = 20 + (size_t)nframes * 8;
    if (out_cap < header) return MIC_ERR_CAPACITY;
    int rc = ensure_device();
    if (rc) return rc;
    // One group per device: a shard's streams land at a provisional place of the caller's buffer -- where they would start if every
    // frame before them came out at its bound -- and are moved down behind the shard before once all lengths are known (the frames of
    // the first shard lie where they belong at once).  That needs the buffer the capacity contract promises; a smaller one: one shard.
    const size_t fb = MIC_HIP_FRAME_BOUND(npx);
    int shards = cur_default() ? 1 : (int)std::min<size_t>(default_devices().size(), (size_t)nframes / 8);
    if (shards < 1 || out_cap < header + (size_t)nframes * fb) shards = 1;
    std::vector<EncGroup> G; std::vector<EncUnit> U;
    for (int k = 0; k < shards; k++) {
        const int f0 = (int)((int64_t)nframes * k / shards), f1 = (int)((int64_t)nframes * (k + 1) / shards);
        const size_t start = shards == 1 ? header : header + (size_t)f0 * fb;
        G.push_back(EncGroup{ frames, out, shards == 1 ? out_cap : start + (size_t)(f1 - f0) * fb, start, f0, f1 - f0 });
        for (int i = f0; i < f1; i++) U.push_back(EncUnit{ (uint64_t)npx * (uint64_t)i, width, height, max_value, 2, k });
    }
    if ((rc = encode_sharded(G, U))) return rc;
    size_t written = 0;
    for (int k = 0; k < shards; k++) {
        if (G[(size_t)k].status != MIC_OK) return G[(size_t)k].status;
        if (k && G[(size_t)k].written) memmove(out + header + written, out + G[(size_t)k].hdr, G[(size_t)k].written);
        written += G[(size_t)k].written;
    }
    G[0].written = written;
    if (G[0].written > 0xFFFFFFFFull) return MIC_ERR_UNSUPPORTED;       // u32 offsets, multiframe.go:75-80
    memset(out, 0, 20);
    memcpy(out, "MIC2", 4);
    put_u32(out + 4, (uint32_t)width); put_u32(out + 8, (uint32_t)height); put_u32(out + 12, (uint32_t)nframes);
    out[16] = 0x01;                                                     // PipelineSpatial, multiframe.go:28
    size_t off = 0;
    for (int i = 0; i < nframes; i++) {
        put_u32(out + 20 + (size_t)i * 8, (uint32_t)off);
        put_u32(out + 24 + (size_t)i * 8, (uint32_t)U[(size_t)i].len);
        off += U[(size_t)i].len;
    }
    *out_len = header + G[0].written;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

int mic_hip_mic2_decompress(const uint8_t *c, size_t len, uint16_t *frames_out, size_t frames_cap_px) try {
    if (!c || !frames_out) return MIC_ERR_ARGS;
    int w, h, n, temporal;
    int rc = mic_hip_mic2_info(c, len, &w, &h, &n, &temporal);
    if (rc) return rc;
    if (w <= 0 || h <= 0 || n <= 0) return MIC_ERR_CORRUPT;
    const size_t npx = (size_t)w * (size_t)h;
    if (npx * (size_t)n > frames_cap_px) return MIC_ERR_CAPACITY;
    if (npx > ((size_t)1 << 28)) return MIC_ERR_UNSUPPORTED;
    if (temporal) return mic2_temporal_decompress(c, len, w, h, n, n, frames_out);   // mic_temporal.hip
    const size_t data_off = 20 + (size_t)n * 8;
    if ((rc = ensure_device())) return rc;
    const int shards = cur_default() ? 1 : std::max(1, (int)std::min<size_t>(default_devices().size(), (size_t)n / 8));   // (frames decode into fixed places: any cut will do)
    std::vector<DecGroup> G; std::vector<DecUnit> U;
    for (int k = 0; k < shards; k++) {
        const int f0 = (int)((int64_t)n * k / shards), f1 = (int)((int64_t)n * (k + 1) / shards);
        G.push_back(DecGroup{ c, frames_out, f0, f1 - f0 });
        for (int i = f0; i < f1; i++) {
            const size_t start = data_off + get_u32(c + 20 + (size_t)i * 8), bl = get_u32(c + 24 + (size_t)i * 8);
            if (start + bl > len) return MIC_ERR_CORRUPT;                 // multiframe.go:137-139
            if (bl == 0) return MIC_ERR_CORRUPT;
            U.push_back(DecUnit{ start, bl, (uint64_t)npx * (uint64_t)i, w, h, 0, k });
        }
    }
    if ((rc = decode_sharded(G, U))) return rc;
    for (const DecGroup &g : G) if (g.status != MIC_OK) return g.status;
    return MIC_OK;
} catch (const std::bad_alloc &) { return MIC_ERR_NOMEM; } catch (...) { return MIC_ERR_INTERNAL; }   // (no C++ exception crosses the C ABI)

}  // extern "C"
