// mic_dev.h -- device-side data layout shared by the encode / decode kernels and the launcher.
//
// One "unit" is one independently coded stream of the reference: a PICS strip
// (parallelstrips.go:77-93), a MIC2 frame (multiframecompress.go:186-209) or a MIC3 plane
// (wsicompress.go:373-421).  The launcher fills an array of MicUnit in HBM; every kernel
// is launched over that array (blockIdx.x or blockIdx.y = unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MIC_MAXSYM        65535u
#define MIC_MIN_TABLELOG  5      // fseu16.go:26
#define MIC_MAX_TABLELOG  16     // fseu16.go:23
#define MIC_DEF_TABLELOG  11     // fseu16.go:25

// status codes: keep in sync with include/mic_hip.h
#define MICD_OK                  0
#define MICD_ERR_ARGS           -1
#define MICD_ERR_USE_RLE        -3
#define MICD_ERR_CAPACITY       -5
#define MICD_ERR_CORRUPT        -6
#define MICD_ERR_INTERNAL       -8
#define MICD_ERR_UNSUPPORTED    -9
#define MICD_ERR_INCOMPRESSIBLE -10
#define MICD_INT_GROW           -21    // internal: a tier-1 slab would overflow; the host runs the batch again in tier 2 (never surfaces)

struct MicUnit {
    // ---- inputs ------------------------------------------------------------------
    const uint16_t *px_in;    // encode: source pixels (w*h u16)
    uint16_t       *px_out;   // decode: destination pixels
    const uint8_t  *comp_in;  // decode: compressed blob
    uint32_t        comp_len;
    int32_t         w, h;
    uint16_t        max_value;
    uint16_t        nstates;  // encode: requested flavour 1/2/4/8, 108 = rANS-8
    uint32_t        mode;     // 0 = frame (Delta+RLE around the FSE stage), 1 = bare FSE: px_in / px_out hold u16 symbols, w = count
    uint32_t        no_fallback; // 1 = FSECompressU16* semantics (no N -> ... -> 1 chain)
    uint32_t        req_tl;   // ScratchU16.TableLog of a bare FSE call (fseu16.go:101-102); 0 = the default 11
    uint32_t        pred;     // mode 0: 0 = avg(left, top) predictor, 1 = gradient-adaptive (deltagradrlecompressu16.go)
    uint32_t        tier;     // 1: the slabs below are the small ones -- crossing one of their capacities is MICD_INT_GROW, not an error
    uint32_t        tab_cap;  // entries of hist / norm / tt_* / state_tab / tab_sym (8192 or 65536)
    // ---- per-unit workspace (HBM) ------------------------------------------------
    uint16_t *tok;            // RLE token stream (encode: produced, decode: FSE output)
    uint32_t  tok_cap;
    uint32_t *hist;           // [65536] symbol histogram (fseu16.go:64)
    int32_t  *norm;           // [65536] normalised counts (fseu16.go:65)
    uint32_t *tt_nb;          // [65536] enc: symbolTT.deltaNbBits ; dec: dtab (newState | nbBits<<16)
    int32_t  *tt_find;        // [65536] enc: symbolTT.deltaFindState ; dec: symbolNext scratch
    uint32_t *state_tab;      // [65536] enc: stateTable
    uint16_t *tab_sym;        // [65536] enc: tableSymbol ; dec: symbol of each state
    int32_t  *cumul;          // [65538]
    uint8_t  *blob;           // encode: staging output (NCount + bitstream, 6-byte prefix first)
    uint32_t  blob_cap;
    uint2    *seg;            // decode: RLE segments {token index of the payload | same-run << 31, first symbol index}
    uint32_t  seg_cap;
    uint16_t *sym;            // decode: expanded delta-symbol stream; encode: per-token tANS states
    uint32_t  sym_cap;
    uint32_t *flags;          // decode: 1 bit per pixel, set = pixel stored raw behind an escape
    uint32_t  nseg;
    uint32_t  nsym;
    uint32_t  walk_ok;        // decode: seg/nseg/nsym already produced by the tANS kernel's header walker
    uint32_t  dec_thr;        // decode: delta threshold (1 << (depth-1)) - 1 of the stream's own max value
    uint32_t  walk_mode;      // decode, mode 1 units: 1 = the symbols are a length-prefixed RLE stream (WaveletV2): k_dec_translate marks it
                              // walk_ok = 3 and k_rle_walk_* (mic_wavelet.hip) walk its headers in parts: seg / nseg / nsym, walk_ok = 1,
                              // and in `flags` the segment that holds every 8192nd symbol
    uint32_t  wv_slow;        // WaveletV2: this frame takes the one-group kernels (escape words in its stream, or a stream the walker refused)
    uint32_t  wv_zmax;        // WaveletV2 encode: largest zigzag symbol of the frame
    uint32_t  hist_hi;        // encode: every token the tokeniser counted is below this (0: unknown -- all 65536 bins are scanned)
    // ---- results -----------------------------------------------------------------
    uint32_t ntok;            // number of u16 in tok
    uint32_t blob_len;
    int32_t  status;
    int32_t  nstates_used;
    uint32_t symbol_len;
    uint32_t max_count;
    uint32_t table_log;
    uint32_t zero_bits;
    uint32_t hdr_len;         // NCount header bytes
    uint32_t bits_off;        // decode: offset of the bitstream inside comp_in
    uint32_t count;           // decode: symbol count from the 6-byte prefix
    uint32_t flavour;         // decode: 1/2/4/8, 108 = rANS-8
    uint32_t dbg[16];         // MIC_STAMP builds: shader-clock ticks per kernel phase (tools/stamp_*.py)
};

// A pointer read out of a MicUnit is "generic" to the compiler, and generic accesses are flat_load / flat_store: those count on
// lgkmcnt as well as on vmcnt, so every s_waitcnt lgkmcnt(0) -- there is one in front of every work-group barrier -- waits for the
// HBM loads and stores in flight (a prefetch issued before a barrier is no prefetch any more).  mic_g() states what every
// workspace pointer is: global memory (global_load / global_store: vmcnt only).
#define MIC_GLOBAL __attribute__((address_space(1)))
template <class T> using mic_gp = MIC_GLOBAL T *;
template <class T> __device__ __forceinline__ mic_gp<T> mic_g(T *p) { return (mic_gp<T>)p; }

// Phase stamps for diagnostic builds (EXTRA_FLAGS=-DMIC_STAMP); they compile to nothing otherwise.
#ifdef MIC_STAMP
#define MIC_STAMP_BEGIN() uint64_t _mic_t0 = __builtin_amdgcn_s_memtime()
#define MIC_STAMP_AT(u, k) do { const uint64_t _t = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) (u).dbg[k] += (uint32_t)(_t - _mic_t0); _mic_t0 = _t; } while (0)
#else
#define MIC_STAMP_BEGIN() do { } while (0)
#define MIC_STAMP_AT(u, k) do { } while (0)

#endif

// A barrier across which threads of one work-group hand GLOBAL memory to each other where the order matters to something outside the
// group's own L1 -- the tANS encoder's plain stores of 64-bit units, then the atomic ORs other threads make into them (executed at L2).
// __syncthreads() / __threadfence_block() emit no wait for outstanding stores at work-group scope on gfx950 (the memory model leaves
// that order to the CU's L1, which is enough for loads, not for an L2 atomic that could overtake a store still in flight): the writers
// wait for their stores to be acknowledged, the readers drop their L1 lines.
#define MIC_GROUP_HANDOFF() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); \
                                 __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); } while (0)

__device__ __forceinline__ int mic_len16(uint32_t v) { return v ? 32 - __clz(v) : 0; }
// gradPredict (deltagradcompressu16.go:147-167): avg(W, N) + clamp((NE - NW) >> 3, +-(|W - NW| + |N - NW|) / 2); no gradient, no correction
__device__ __forceinline__ int32_t mic_grad_predict(int32_t w, int32_t n, int32_t nw, int32_t ne) {
    const int32_t lim = (int32_t)((__sad((unsigned)w, (unsigned)nw, 0u) + __sad((unsigned)n, (unsigned)nw, 0u)) >> 1);
    const int32_t corr = (ne - nw) >> 3;
    return ((w + n) >> 1) + min(max(corr, -lim), lim);
}
// the gradient-adaptive prediction of pixel (x, y) from a plain pixel array (deltagradrlecompressu16.go:36-53)
__device__ __forceinline__ int32_t mic_grad_predict_at(const uint16_t *px, int w, int x, int y) {
    const size_t idx = (size_t)y * (size_t)w + (size_t)x;
    if (y == 0) return x ? (int32_t)px[idx - 1] : 0;
    if (x == 0) return (int32_t)px[idx - (size_t)w];
    const int32_t nw = px[idx - (size_t)w - 1];
    return mic_grad_predict(px[idx - 1], px[idx - (size_t)w], nw, (x + 1 < w) ? (int32_t)px[idx - (size_t)w + 1] : nw);
}
// fseu16.go:170-172 -- Len32(v)-1, wraps to 0xFFFFFFFF for 0
__device__ __forceinline__ uint32_t mic_high_bits(uint32_t v) { return (uint32_t)(31 - __clz(v)) ; }
__device__ __forceinline__ uint32_t mic_table_step(uint32_t size) { return (size >> 1) + (size >> 3) + 3; }
