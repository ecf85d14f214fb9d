// mic_launch.h -- launcher interface between mic_api.hip and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "mic_dev.h"

// Kernel attributes (opt-in dynamic LDS) are per device, and launches come from any thread: run(setup) calls `setup` once per device
// (the calling thread's current one) and returns only after it has completed there, whichever thread ran it -- the device's bit is
// published AFTER the attribute calls, under the mutex, so no thread can launch a > 64 KiB-LDS kernel on a device whose opt-in is
// still on its way.  The fast path is one acquire load.
struct MicPerDeviceOnce {
    std::atomic<uint64_t> mask{0};
    std::mutex mu;
    template <class F> void run(F &&setup) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) { setup(); return; }
        const uint64_t bit = 1ull << (d & 63);
        if (mask.load(std::memory_order_acquire) & bit) return;
        std::lock_guard<std::mutex> lk(mu);
        if (mask.load(std::memory_order_relaxed) & bit) return;
        setup();
        mask.fetch_or(bit, std::memory_order_release);
    }
};

// Optional per-kernel timing: when enabled the launchers drop a HIP event on the launch
// stream in front of every kernel (and one at the end); mic_hip_session_last_timings turns
// consecutive events into per-kernel device milliseconds.
struct MicTimer {
    bool enabled = false;
    bool accumulate = false;           // keep the marks of earlier enqueues (a call that runs several launch chains, e.g. slabs)
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> pool;
    std::vector<const char *> names;   // names[i] labels the span ev[i] .. ev[i+1]
    size_t used = 0;
    void reset(hipStream_t s) { stream = s; if (!accumulate) { used = 0; names.clear(); } }
    void clear() { used = 0; names.clear(); }
    void mark(const char *name) {
        if (!enabled) return;
        if (used == pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; pool.push_back(e); }
        (void)hipEventRecord(pool[used++], stream);
        names.push_back(name);
    }
    void destroy() { for (auto e : pool) (void)hipEventDestroy(e); pool.clear(); used = 0; names.clear(); }
};

// variant: launch flags -- MIC_VARIANT_GRAD when some unit has pred = 1 (their tokeniser / predictor instantiations are only launched then)
#define MIC_VARIANT_GRAD 0x1000
#define MIC_VARIANT_NARROW 0x2000       // encode: no unit of the batch asks for more than two states (sizes k_enc_tans_wg's end-state area)
#define MIC_VARIANT_FRAMES 0x4000       // encode: every unit is a frame (mode 0, avg or gradient predictor): the symbol-unit kernels are not launched
// Launch masks.  Most kernels of a chain come in classes (table size, flavour, frame width) and a homogeneous batch uses one or two
// of them; a launch of a class without units is an empty grid that still costs ~5 us of the device's time (~45 of them per encode +
// decode of a PICS batch, 0.2 ms).  What the HOST knows (widths, modes) it says outright; what only the streams know (flavour,
// tableLog) a session remembers from its last batches -- the catch-all kernels (k_dec_tans_gl / _serial, k_enc_tans_serial) are
// always launched and take whatever a stale mask leaves, so a wrong guess costs time, never correctness.  All ones: launch everything.
// What a wrong guess costs (tools/mask_miss.py, 2304 XR strips, a session that had only seen two-state batches meets a four-state
// one): decode 69 ms once instead of 15.5 (k_dec_tans_gl takes the batch; the next call has learned the class).  On the encode side
// the same miss was 0.77 s through the serial encoder, so there the host's own knowledge is OR-ed in (mic_api.hip: the two big
// instances always, the small and deep ones by unit size and depth) and memory only decides what the host cannot know.
#define MIC_ENC_CLS_NARROW2 0x01u       // k_enc_tans_wg<13, 512, .., 1>: two states, records in LDS
#define MIC_ENC_CLS_WIDE    0x02u       // k_enc_tans_wg<13, 512, .., 2>
#define MIC_ENC_CLS_SMALL12 0x04u       // one-wave instances
#define MIC_ENC_CLS_SMALL13 0x08u
#define MIC_ENC_CLS_TL14    0x10u
#define MIC_ENC_CLS_TL15    0x20u
#define MIC_ENC_CLS_TL16    0x40u
void mic_launch_encode(MicUnit *d_units, int n, hipStream_t stream, int variant, MicTimer *t, uint32_t enc_mask = ~0u);
// d_cls: per-session scratch of MIC_CLS_INTS(n) ints for the per-class unit lists of the lane-per-state tANS decoder (mic_decode_ls.hip)
#define MIC_CLS_HEAD 32
#define MIC_CLS_CLASSES 30
#define MIC_CLS_INTS(n) (MIC_CLS_HEAD + MIC_CLS_CLASSES * (size_t)(n))
// pred_mask: predictor classes by frame width (mic_pred_bit); cls_mask: bit c = lane-per-state class c of mic_decode_ls.hip (MIC_CLS_CLASSES of them)
void mic_launch_decode(MicUnit *d_units, int n, hipStream_t stream, int variant, MicTimer *t, int *d_cls, uint32_t pred_mask = ~0u, uint32_t cls_mask = ~0u);
void mic_launch_dec_tans_ls(MicUnit *d_units, int n, int *d_list, int *d_count, hipStream_t stream, MicTimer *t, uint32_t cls_mask = ~0u);
// d_cap: bytes of d_dst -- a batch whose streams do not fit is left alone (dst_off[n], the total, says so)
void mic_launch_pack(const MicUnit *d_units, int n, uint64_t *d_off, uint8_t *d_dst, uint64_t d_cap, hipStream_t stream, MicTimer *t);
// pred_mask: the predictor kernels the batch's widths call for (mic_pred_bit) -- bits 0..6 the chunk classes of k_dec_predict_rows
// (bit (K - 18) / 4), MIC_PRED_* the others; ~0u when the caller does not know its widths
#define MIC_PRED_NARROW 0x100u          // k_dec_predict<0, 16>: up to MIC_ROWS_LO columns
#define MIC_PRED_WAVE2  0x200u          // k_dec_predict2: MIC_ROWS_HI < columns <= 8128
#define MIC_PRED_WIDE   0x400u          // k_dec_predict<1, 64>: wider
void mic_launch_decode_pixels(MicUnit *d_units, int n, hipStream_t stream, MicTimer *t, bool any_grad = false, uint32_t pred_mask = ~0u);
void mic_launch_decode_rows(MicUnit *d_units, int n, hipStream_t stream, uint32_t kmask);
// tokens -> pixels in one kernel for the same widths (mic_decode_fused.hip); units it takes are marked walk_ok = 4 and skipped by the kernels behind it
void mic_launch_decode_fused(MicUnit *d_units, int n, hipStream_t stream, uint32_t kmask);
// Frames of MIC_ROWS_LO < columns <= MIC_ROWS_HI take the row-by-row predictor: pixels per lane and row (0: another kernel's frame)
#define MIC_ROWS_LO 1008
#define MIC_ROWS_HI 2688
__host__ __device__ inline int mic_rows_k(int w) {
    if (w <= MIC_ROWS_LO || w > MIC_ROWS_HI) return 0;
    const int k = (w + 63) >> 6;                    // (16 .. 42; the classes are 18, 22 .. 42: an odd number of dwords per lane, mic_decode_rows.hip)
    return 18 + 4 * ((max(k, 18) - 18 + 3) / 4);
}
__host__ __device__ inline uint32_t mic_rows_kbit(int w) { const int k = mic_rows_k(w); return k ? 1u << ((k - 18) / 4) : 0u; }
__host__ __device__ inline uint32_t mic_pred_bit(int w) {
    return w <= MIC_ROWS_LO ? MIC_PRED_NARROW : w <= MIC_ROWS_HI ? mic_rows_kbit(w) : w <= 8192 - 64 ? MIC_PRED_WAVE2 : MIC_PRED_WIDE;
}
// lane-per-state decode class of a stream (mic_decode_ls.hip: k_dec_classify), -1: left to k_dec_tans_gl / k_dec_tans_serial
__host__ __device__ inline int mic_dec_cls(uint32_t flavour, uint32_t tl, uint32_t zero_bits) {
    const uint32_t ns = flavour == 108 ? 8u : flavour;
    if (!(ns == 2 || ns == 4 || ns == 8) || tl < MIC_MIN_TABLELOG || tl > MIC_MAX_TABLELOG || (tl == 16 && zero_bits)) return -1;
    return (tl <= 12 ? 4 : (int)tl - 13) * 6 + (ns == 2 ? 0 : ns == 4 ? 2 : 4) + (zero_bits ? 1 : 0);
}
void mic_launch_enc_tables(MicUnit *d_units, int n, hipStream_t stream);
void mic_launch_dec_tables(MicUnit *d_units, int n, hipStream_t stream);
