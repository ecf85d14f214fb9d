// mic_decode_fused.hip -- tokens -> pixels in ONE kernel for wide frames: RLE expansion, escape resolution and the inverse
// Delta(avg) predictor row by row, one wave per unit (k_dec_rows_tok).
//
// Reference: RleDecompressU16.DecodeNext2 (rledecompressu16.go:59-85) pulled once per symbol by
// DeltaRleDecompressU16.Decompress (deltarlecompressu16.go:69-128).  The two-kernel form (k_dec_pixels_wg: tokens -> one delta
// symbol per pixel + a raw bit, in HBM; k_dec_predict_rows: symbols -> pixels) writes every pixel twice and reads it once in
// between: 6 N bytes of traffic that only carry an intermediate.  The row-by-row predictor (mic_rowpred.h) consumes a frame in row
// order, which is the order the token stream produces it in, so a wave can make each row's symbols itself:
//   * the header walk has left the unit's RLE segments {payload position | run flag, first symbol} in HBM (k_dec_translate); a row
//     of W pixels is W consecutive symbols (more when it holds escapes): a handful of pieces of those segments -- a literal chunk is
//     up to midCount symbols long, a run as long as it likes.  A window of 64 segment records lives in two registers; a row's
//     pieces (up to eight) are described one per lane, every 16-byte vector of the LDS row finds the piece it lies in and is loaded
//     from the token stream at whatever 2-byte offset that is (the destination side is what is aligned; a run's value comes in
//     the same kind of load and is spread in registers), the vectors a piece ends in are put together element by element, eight
//     lanes apiece.  The loads of a row are issued one row ahead.  Rows of more pieces are copied piece by piece (assemble).
//   * a symbol equal to the delimiter is an escape marker unless it is the payload of one (marker[i] = isDelim[i] & !marker[i-1]);
//     a row without a delimiter -- almost every row -- is its symbols as they stand.  A row with one takes the marker scan of
//     k_dec_pixels_wg at wave scale (per-lane transition functions, composed across the lanes, pixels scattered into a second LDS
//     row with their raw bits) over up to 512 symbols more than the row has pixels.
//   * then the predictor, and the row leaves through the same LDS buffer in 1 KiB runs (mic_decode_rows.hip).
// Whatever is out of the ordinary -- a literal piece that points past the stream, symbols that run out, a hole or a backward step in
// the segment list, fewer than eight tokens -- is NOT handled here: the unit is left as it was (walk_ok stays 1) and the
// two-kernel path behind this one decodes it, with the reference's error behaviour.  A unit done here is marked walk_ok = 4.
#include "mic_dev.h"
#include "mic_launch.h"
#include "mic_rowpred.h"

typedef uint32_t rf_v4 __attribute__((ext_vector_type(4)));
typedef rf_v4 RfQ __attribute__((aligned(2)));
typedef __attribute__((address_space(1))) uint16_t *rf_gu16;
typedef const __attribute__((address_space(1))) uint16_t *rf_gcu16;
typedef __attribute__((address_space(1))) uint8_t *rf_gu8;
typedef const __attribute__((address_space(1))) uint2 *rf_gseg;

#define RF_SLACK 512                       // symbols beyond a row's pixels the escape path looks at (an escape costs one)

template <int K>
__global__ void __launch_bounds__(256, 3) k_dec_rows_tok(MicUnit *units, int n_units) {
    typedef RowPred<K> RP;
    constexpr int KD = K / 2;
    constexpr int CH = K + 8, CHD = CH / 2;                              // symbols per lane in the marker scan: 64 CH = 64 K + RF_SLACK
    static_assert(64 * CH == 64 * K + RF_SLACK && (CHD & 1) == 1, "the scan covers a row and its slack, an odd number of dwords per lane");
    constexpr int NVJ = (8 * K + 63) / 64;                              // rounds of 64 vectors that cover a row of 64 K pixels
    constexpr int SB32 = 32 * CH + 4;                                   // dwords of the symbol / staging buffer (64 CH symbols)
    constexpr int PB32 = 32 * K + 4;                                    // ... of the escape rows' pixel row
    constexpr int RB32 = 2 * K + 4;                                     // ... of its raw bits
    extern __shared__ __attribute__((aligned(16))) uint32_t s_rf[];
    // (everything that is one value per wave is made so for the compiler too -- readfirstlane -- and lives in scalar registers: the
    // vector registers are what this kernel runs out of, and a spilled loop invariant is reloaded behind a wait for ALL memory
    // operations in flight, the row before's stores included)
    const int ui = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (ui >= n_units) return;
    MicUnit &u = units[ui];
    if (u.status != MICD_OK || u.mode != 0 || u.pred || u.walk_ok != 1) return;
    const int W = __builtin_amdgcn_readfirstlane(u.w), H = __builtin_amdgcn_readfirstlane(u.h);
    if (mic_rows_k(W) != K) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t *const sb = s_rf + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (SB32 + PB32 + RB32);
    uint32_t *const pb = sb + SB32, *const rawb = pb + PB32;
    uint16_t *const sb16 = (uint16_t *)sb;
    const uint32_t npx = (uint32_t)W * (uint32_t)H;
    const uint32_t ntok = (uint32_t)__builtin_amdgcn_readfirstlane((int)u.ntok), nseg = (uint32_t)__builtin_amdgcn_readfirstlane((int)u.nseg);
    const uint32_t nsym = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(u.nsym, min(u.sym_cap, 2u * npx + 2u)));
    if (ntok < 8 || nseg < 1 || nsym < 1 || nseg > 16u * (uint32_t)H + 64u) return;
    const uintptr_t px_u = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uintptr_t)u.px_out >> 32)) << 32) |
                           (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)u.px_out);
    const uintptr_t tok_u = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uintptr_t)u.tok >> 32)) << 32) |
                            (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)u.tok);
    const rf_gu16 px = (rf_gu16)px_u;
    const rf_gcu16 tok = (rf_gcu16)tok_u;
    const uintptr_t seg_u = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uintptr_t)u.seg >> 32)) << 32) |
                            (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)u.seg);
    const rf_gseg seg = (rf_gseg)seg_u;
    // the stream's own parameters (deltarlecompressu16.go:25-27, :71): the first symbol is the max value
    const uint32_t seg0x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(seg[0].x & 0x7FFFFFFFu));
    if (seg0x >= ntok) return;
    const int depth = mic_len16((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tok[seg0x]));
    if (depth == 0) return;
    const uint32_t thr = (1u << (depth - 1)) - 1u, delim = (1u << depth) - 1u;
    const uint32_t thr2 = thr | (thr << 16);
    const uint32_t x0 = lane * K;
    const int nv = min(max(W - (int)x0, 0), K);
    const int nl = (W + K - 1) / K;
    const uint32_t voff = lane * 16u;
    const uint32_t nfull = (uint32_t)W / 8u, ntail = (uint32_t)W % 8u;
    typedef uint16_t rf_pk __attribute__((ext_vector_type(2)));

    // ---- the segment window: records wbase + lane of the walk, in registers ----
    uint32_t wbase = 0, segx = 0, segy = 0;
    auto win_load = [&](uint32_t base) {
        wbase = base;
        const uint32_t i = base + lane;
        uint2 r = make_uint2(0u, nsym);
        if (i < nseg) { r.x = seg[i].x; r.y = seg[i].y; }
        segx = r.x; segy = min(r.y, nsym);
    };
    win_load(0);
    auto seg_y = [&](uint32_t i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)segy, (int)(i - wbase)); };   // wbase <= i < wbase + 64
    auto seg_x = [&](uint32_t i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)segx, (int)(i - wbase)); };

    // ---- assembly: symbols [spos + from, spos + to) of the stream into sb16[from .. to) ----
    uint32_t sg = 0;                                                     // a segment at or before the one that holds symbol spos
    auto assemble = [&](uint32_t spos, uint32_t from, uint32_t to) -> bool {
        uint32_t done = from;
#pragma unroll 1
        while (done < to) {
            // the segment that holds symbol spos + done
            const uint32_t s = spos + done;
#pragma unroll 1
            for (;;) {
                if (sg < wbase || sg + 1 >= wbase + 63) win_load(sg);    // (the window always holds sg and sg + 1)
                if (sg + 1 < nseg && seg_y(sg + 1) <= s) sg++; else break;
            }
            const uint32_t ys = seg_y(sg), ye = (sg + 1 < nseg) ? seg_y(sg + 1) : nsym, xs = seg_x(sg);
            if (s < ys || s >= ye) return false;                          // (a hole in the walk: not here)
            const uint32_t hi_ = min(to, ye - spos), len = hi_ - done;    // the piece: sb16[done .. hi_)
            const uint32_t x = xs & 0x7FFFFFFFu;
            const uint32_t a = min(hi_, (done + 7u) & ~7u);               // first 16-byte boundary of the destination inside the piece
            const uint32_t nvec = (hi_ - a) / 8u, tail0 = a + 8u * nvec;  // whole vectors, then the piece's last elements
            if (xs >> 31) {                                              // a run: one value
                if (x >= ntok) return false;
                const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tok[x]), v2 = v | (v << 16);
                if (lane < a - done) sb16[done + lane] = (uint16_t)v;
                for (uint32_t d = lane; d < nvec; d += 64) *(uint4 *)(sb + (a + 8u * d) / 2u) = make_uint4(v2, v2, v2, v2);
                if (lane < hi_ - tail0) sb16[tail0 + lane] = (uint16_t)v;
            } else {                                                     // a literal chunk: tokens x + (s - ys) ...
                const uint32_t t0 = x + (s - ys);
                if (t0 + len > ntok) return false;                       // (a literal run past the end: Go index panic -- the other path reports it)
                const rf_gcu16 src = tok + t0;
                if (lane < a - done) sb16[done + lane] = src[lane];
                const rf_gu8 vsrc = (rf_gu8)(src + (a - done));
                rf_v4 v[NVJ];
#pragma unroll
                for (int r = 0; r < NVJ; r++) if (lane + 64u * r < nvec) v[r] = *(const __attribute__((address_space(1))) RfQ *)(vsrc + 1024 * r + voff);
#pragma unroll
                for (int r = 0; r < NVJ; r++) if (lane + 64u * r < nvec) *(uint4 *)(sb + a / 2u + 4u * (lane + 64u * r)) = make_uint4(v[r].x, v[r].y, v[r].z, v[r].w);
                for (uint32_t d = lane + 64u * NVJ; d < nvec; d += 64) {  // (a piece longer than a row: the escape path's slack)
                    const rf_v4 w = *(const __attribute__((address_space(1))) RfQ *)(vsrc + 16u * d);
                    *(uint4 *)(sb + a / 2u + 4u * d) = make_uint4(w.x, w.y, w.z, w.w);
                }
                if (lane < hi_ - tail0) sb16[tail0 + lane] = src[tail0 - done + lane];
            }
            done = hi_;
        }
        return true;
    };

    // The same as a GATHER, for rows of up to eight pieces (all but run-dense rows): the pieces are looked at first (no memory
    // operation), every lane then finds the piece each of its destination vectors lies in and issues its loads in one go, and there
    // is ONE wait for the row (the piece-by-piece form above waits several times per piece, every wait behind the row before's
    // stores as well: 35 us a row).  The vectors that straddle a piece boundary -- and the row's last, partial one -- are put together
    // element by element by eight lanes each.  1: done, 0: not here (leave the unit), -1: more than eight pieces, take the other form.
    // The loads of a row are ISSUED one row ahead -- behind the row before's symbols, in front of its predictor and its stores -- and
    // LANDED (registers -> LDS) when that row has left: the wait for them is then a wait for loads that have had a whole row's time,
    // and not one behind the stores just issued (the memory counter is in order).
    rf_v4 nx[NVJ]; uint32_t nx_kind = 0, nx_bval = 0, nx_ep = 0xFFFFFFFFu;   // 5 bits per round: kind (1 literal vector, 2 run), the run value's place in its vector; the boundary element this lane carries
#pragma unroll
    for (int r = 0; r < NVJ; r++) nx[r] = rf_v4{0u, 0u, 0u, 0u};
    auto issue = [&](uint32_t spos, uint32_t to) -> int {
#pragma unroll 1
        for (;;) {
            if (sg < wbase || sg + 1 >= wbase + 63) win_load(sg);
            if (sg + 1 < nseg && seg_y(sg + 1) <= spos) sg++; else break;
        }
        if (sg + 10 >= wbase + 63) win_load(sg);                         // (room for the row's pieces)
        // The row's pieces, one per LANE (lane p: segment sg + p; at most eight are used, a ninth sends the row to the piece-by-piece
        // form): the window's records are shuffled into place, every lane checks and describes its piece, and the vectors then find
        // theirs by counting piece ends -- no walk from piece to piece (the first form: ~110 scalar-unit instructions a piece, 40 %
        // of the kernel).  A piece is needed iff it begins in front of the row's end; the y of a record behind the last is nsym.
        const uint32_t end = spos + to;
        const uint32_t wl = min(sg - wbase + lane, 62u);
        const uint32_t ys = (uint32_t)__shfl((int)segy, (int)wl), yn = (uint32_t)__shfl((int)segy, (int)wl + 1), xs = (uint32_t)__shfl((int)segx, (int)wl);
        const uint32_t si = sg + lane;                                   // this lane's segment
        const uint32_t ye = (si + 1 < nseg) ? yn : nsym, x = xs & 0x7FFFFFFFu, run = xs >> 31;
        const bool needed = lane < 9u && si < nseg && (lane == 0u || ys < end);
        const uint64_t need_m = __ballot(needed);
        if (need_m & 0x100ull) return -1;                                // more than eight pieces
        if ((need_m & (need_m + 1ull)) != 0ull) return 0;                // (not a prefix of the lanes: the records are out of order -- not here)
        const uint32_t P = (uint32_t)__popcll(need_m);
        const uint32_t lo_abs = max(ys, spos), hi_abs = min(ye, end);
        bool bad = ye <= lo_abs;                                         // the piece holds none of the row (a hole, an empty record)
        if (lane == 0u) bad = bad || spos < ys;
        bad = bad || (run ? x >= ntok : x + (hi_abs - ys) > ntok);        // (a literal run past the end: the other path reports it)
        if (__any(needed && bad)) return 0;
        // (a piece's end is the next one's start -- or nsym behind the last record: the needed pieces reach the row's end by construction)
        // (a run's value comes in the same kind of load as a literal vector -- the eight tokens that hold it, the last eight of the
        // unit at most: two kinds of load into one register made the compiler wait for the first before it issued the second, six
        // memory round trips a row)
        const uint32_t xb = min(x, ntok - 8u), xsh = x - xb;
        const uint32_t plo = lo_abs - spos, phi = needed ? hi_abs - spos : 0xFFFu;       // row positions (< 4096)
        const uint32_t pa = run ? xb : x + spos - ys;                     // literal: the token of row position q is pa + q
        const uint32_t pw = plo | (phi << 12) | ((run ? 2u + (xsh << 2) : 1u) << 24);
        uint32_t pend[8];                                                // the pieces' ends, wave-uniform (0xFFF behind the last)
#pragma unroll
        for (int q = 0; q < 8; q++) pend[q] = (uint32_t)__builtin_amdgcn_readlane((int)phi, q);
        auto piece_of = [&](uint32_t pos) -> uint32_t {                   // pieces that end at or in front of `pos`
            uint32_t c = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) c += (pend[q] <= pos) ? 1u : 0u;
            return c;
        };
        uint32_t kind[NVJ], off[NVJ];
#pragma unroll
        for (int r = 0; r < NVJ; r++) {
            const uint32_t vlo = 8u * (lane + 64u * (uint32_t)r), pi = min(piece_of(vlo), 63u);
            const uint32_t w = (uint32_t)__shfl((int)pw, (int)pi), av = (uint32_t)__shfl((int)pa, (int)pi);
            const bool inside = pi < P && vlo >= (w & 0xFFFu) && vlo + 8u <= ((w >> 12) & 0xFFFu);
            const uint32_t kd = w >> 24;
            kind[r] = inside ? kd : 0u;
            off[r] = (kd & 2u) ? av : av + vlo;
        }
        // the vector in which piece `bslot` ends, element by element (the row's last, partial vector is the last piece's)
        const uint32_t bslot = lane >> 3, bk = lane & 7u;
        const uint32_t bend = ((uint32_t)__shfl((int)pw, (int)bslot) >> 12) & 0xFFFu;
        uint32_t ep = 0xFFFFFFFFu, bsrc = 0; bool bhave = false;
        if (bslot < P && (bend & 7u) != 0u) ep = (bend & ~7u) + bk;
        {
            const uint32_t pe = min(piece_of(min(ep, 0xFFEu)), 63u);
            const uint32_t w = (uint32_t)__shfl((int)pw, (int)pe), av = (uint32_t)__shfl((int)pa, (int)pe);
            bhave = ep < to && pe < P;
            bsrc = (w & (2u << 24)) ? av + (w >> 26) : av + ep;           // a run: its value's own token
        }
        nx_kind = 0;
#pragma unroll
        for (int r = 0; r < NVJ; r++) {
            if (kind[r] != 0u) nx[r] = *(const __attribute__((address_space(1))) RfQ *)(tok + off[r]);
            nx_kind |= kind[r] << (5 * r);
        }
        nx_ep = bhave ? ep : 0xFFFFFFFFu;
        if (bhave) nx_bval = tok[bsrc];
        sg += P - 1;                                                     // (the row's last piece: where the next row starts looking)
        return 1;
    };
    auto land = [&]() {
#pragma unroll
        for (int r = 0; r < NVJ; r++) {
            const uint32_t d = lane + 64u * (uint32_t)r, kf = (nx_kind >> (5 * r)) & 31u, kd = kf & 3u, sh = kf >> 2;
            rf_v4 v = nx[r];
            if (kd == 2u) {                                              // a run: token `sh` of the eight, in every position
                const uint32_t dw = (sh & 4u) ? ((sh & 2u) ? v.w : v.z) : ((sh & 2u) ? v.y : v.x);
                const uint32_t h = (sh & 1u) ? (dw >> 16) : (dw & 0xFFFFu), w = h | (h << 16);
                v = rf_v4{w, w, w, w};
            }
            if (kd != 0u) *(uint4 *)(sb + 4u * d) = make_uint4(v.x, v.y, v.z, v.w);
        }
        if (nx_ep != 0xFFFFFFFFu) sb16[nx_ep] = (uint16_t)nx_bval;
    };

    uint32_t tp[KD];
#pragma unroll
    for (int q = 0; q < KD; q++) tp[q] = 0;
    auto put = [&](int y) {
        const uint32_t ps = (uint32_t)y * (uint32_t)W;
#pragma unroll
        for (int q = 0; q < KD; q++) sb[lane * KD + q] = tp[q];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        // (the row's vectors come out of LDS three at a time, ONE wait for the three, then their stores: left to itself the compiler
        // sinks every read into its store's branch -- seven LDS round trips in a row, one behind the other)
#pragma unroll
        for (int j0 = 0; j0 < NVJ; j0 += 3) {
            uint4 v[3];
#pragma unroll
            for (int t = 0; t < 3; t++) if (j0 + t < NVJ) v[t] = *(const uint4 *)(sb + 4 * min(lane + 64u * (uint32_t)(j0 + t), 8u * K - 1u));
            const uint32_t tv = (j0 + 3 >= NVJ && lane < ntail) ? (uint32_t)sb16[8 * nfull + lane] : 0u;
            __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int t = 0; t < 3; t++) if (j0 + t < NVJ) asm volatile("" : "+v"(v[t].x), "+v"(v[t].y), "+v"(v[t].z), "+v"(v[t].w));
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int j = j0 + t;
                if (j < NVJ && lane + 64u * (uint32_t)j < nfull)
                    *(__attribute__((address_space(1))) RfQ *)((rf_gu8)(px + ps) + 1024 * j + voff) = rf_v4{v[t].x, v[t].y, v[t].z, v[t].w};
            }
            if (j0 + 3 >= NVJ && lane < ntail) px[ps + 8 * nfull + lane] = (uint16_t)tv;
        }
    };

#ifdef RF_STATS     // diagnostic build: rows by path and shader-clock ticks by phase, in u.dbg (tools/time_fused.py)
    uint32_t st_rows[4] = { 0, 0, 0, 0 }; uint64_t st_t[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, st_prev = __builtin_amdgcn_s_memtime();
#define RF_T(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); st_t[i] += t_ - st_prev; st_prev = t_; } while (0)
#define RF_ROW(i) (st_rows[i]++)
#else
#define RF_T(i) do { } while (0)
#define RF_ROW(i) do { } while (0)
#endif
    uint32_t spos = 1;                                                   // symbol 0 is the max value, not a pixel
    if (spos + (uint32_t)W > nsym) return;                               // tokens ran out (Go: panic): the other path's error
    int ar = issue(spos, (uint32_t)W);                                   // row 0
    if (ar == 0) return;
    for (int y = 0; y < H; y++) {
        if (ar > 0) land();
        else { RF_ROW(1); if (!assemble(spos, 0u, (uint32_t)W)) return; } // (more than eight pieces: piece by piece)
        // What lies behind the row's end in the row buffer (the next row's first symbols, or the results staged for the last store) is
        // computed along by the last lanes: there it must be the symbol that changes nothing, or a stale result would look like a
        // wrap-around to the predictor.  Fewer than 256 positions -- written here, after the row (LDS operations of a wave are in
        // order), instead of a select per dword of every lane's chunk.
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t i = (uint32_t)W + lane + 64u * (uint32_t)t;
            if (i < 64u * K) sb16[i] = (uint16_t)thr;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        RF_T(0);
        // A delimiter is the largest value a symbol takes (anything above it can only be the payload behind one, or a damaged stream's):
        // the row has one iff the maximum over its symbols reaches it -- one packed max per dword.
        uint32_t e[KD]; uint64_t raw = 0;
        typedef uint16_t rf_pk2 __attribute__((ext_vector_type(2)));
        rf_pk2 mx = { 0, 0 };
#pragma unroll
        for (int q = 0; q < KD; q++) {
            e[q] = sb[lane * KD + q];
        }
#pragma unroll
        for (int q = 0; q < KD; q++) mx = __builtin_elementwise_max(mx, __builtin_bit_cast(rf_pk2, e[q]));
        const uint32_t hasd = ((uint32_t)mx.x >= delim || (uint32_t)mx.y >= delim) ? 1u : 0u;
        uint32_t used = (uint32_t)W;
        const bool esc_row = __any(hasd != 0u);
        RF_T(4);
        if (esc_row) {
            RF_ROW(2);
            // ---- escapes in this row: markers, pixel numbering, raw bits (k_dec_pixels_wg's scan, one wave) ----
            const uint32_t have = min((uint32_t)W + RF_SLACK, nsym - spos);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            const uint32_t sg_keep = sg;                                 // (the next row starts inside what the slack covers)
            if (have > (uint32_t)W && !assemble(spos, (uint32_t)W, have)) return;
            sg = sg_keep;
            for (uint32_t i = lane; i < (uint32_t)RB32; i += 64) rawb[i] = 0;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            // (rolled loops over the LDS row, a symbol at a time: escape rows are rare, and unrolled over registers the two passes are
            // what the kernel's register count would be set by)
            const uint32_t i0 = lane * CH;
            uint32_t s0 = 0, s1 = 1, c0 = 0, c1 = 0;                     // exit state / pixels for entry state 0 and 1
#pragma unroll 1
            for (int k = 0; k < CH; k++) {
                const uint32_t x = sb16[i0 + k];
                const uint32_t in = (i0 + k < have) ? 1u : 0u, d = (in && x == delim) ? 1u : 0u;
                const uint32_t m0 = d & (s0 ^ 1u), m1 = d & (s1 ^ 1u);
                c0 += in & (m0 ^ 1u); c1 += in & (m1 ^ 1u);
                s0 = in ? m0 : s0; s1 = in ? m1 : s1;
            }
            uint32_t tr = s0 | (s1 << 1) | (c0 << 2) | (c1 << 17);
            auto compose = [](uint32_t a, uint32_t b) -> uint32_t {       // a then b
                const uint32_t a0 = a & 1u, a1 = (a >> 1) & 1u;
                const uint32_t bc0 = (b >> 2) & 0x7FFFu, bc1 = b >> 17;
                const uint32_t n0 = (b >> a0) & 1u, n1 = (b >> a1) & 1u;
                const uint32_t q0 = ((a >> 2) & 0x7FFFu) + (a0 ? bc1 : bc0);
                const uint32_t q1 = (a >> 17) + (a1 ? bc1 : bc0);
                return n0 | (n1 << 1) | (q0 << 2) | (q1 << 17);
            };
            uint32_t incl = tr;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const uint32_t o = __shfl_up(incl, dd);
                if (lane >= (uint32_t)dd) incl = compose(o, incl);
            }
            uint32_t excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 2u;                                    // identity: s0 = 0, s1 = 1, no pixels
            uint32_t st = excl & 1u, pl = (excl >> 2) & 0x7FFFu;         // (a row starts behind a pixel: entry state 0)
            const uint32_t total = ((uint32_t)__builtin_amdgcn_readlane((int)incl, 63) >> 2) & 0x7FFFu;
            if (total < (uint32_t)W) return;                             // more escapes than the slack holds: the other path
            uint16_t *const pb16 = (uint16_t *)pb;
            uint32_t last_at = 0xFFFFFFFFu;
#pragma unroll 1
            for (int k = 0; k < CH; k++) {
                const uint32_t x = sb16[i0 + k];
                const bool in = i0 + k < have;
                const uint32_t m = (in && x == delim) ? (st ^ 1u) : 0u;
                if (in && !m) {
                    if (pl < (uint32_t)W) {
                        pb16[pl] = (uint16_t)x;
                        if (st) atomicOr(&rawb[pl >> 5], 1u << (pl & 31));      // stored raw behind an escape
                        if (pl == (uint32_t)W - 1u) last_at = i0 + k;
                    }
                    pl++;
                }
                if (in) st = m;
            }
            const uint64_t who = __ballot(last_at != 0xFFFFFFFFu);
            used = (uint32_t)__builtin_amdgcn_readlane((int)last_at, (int)__builtin_ctzll(who)) + 1u;   // symbols this row took
            for (uint32_t i = (uint32_t)W + lane; i < 64u * K; i += 64) pb16[i] = (uint16_t)thr;   // (behind the row's end: the symbol that changes nothing)
            __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int q = 0; q < KD; q++) e[q] = pb[lane * KD + q];
            const uint32_t pf = min(x0, (uint32_t)W - 1u), w = pf >> 5;
            const uint32_t f0 = rawb[w], f1 = rawb[w + 1], f2 = rawb[w + 2];
            const uint32_t l32 = __builtin_amdgcn_alignbit(f1, f0, pf), h32 = __builtin_amdgcn_alignbit(f2, f1, pf);
            raw = (((uint64_t)h32 << 32) | l32) & (nv >= 64 ? ~0ull : ((1ull << nv) - 1ull));
        }
        // symbols minus thr, packed, back into LDS (the lane's own dwords of the pixel row): the predictor reads them from there, twice --
        // held in registers next to the row above AND the next row's loads they are what spills
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int q = 0; q < KD; q++) {
            const rf_pk a = __builtin_bit_cast(rf_pk, e[q]), t2 = __builtin_bit_cast(rf_pk, thr2);
            pb[lane * KD + q] = __builtin_bit_cast(uint32_t, (rf_pk)(a - t2));   // (invalid positions were set to thr above: zero here)
        }
        RowSymsLds es{ pb + lane * KD };
        __builtin_amdgcn_s_waitcnt(0xC07F);                             // (the buffers are read: the row's results may go in)
        RF_T(5);
        // the next row's loads go out now: its first symbol is known, and nothing of the predictor or of put() touches their registers
        if (y + 1 < H) {
            if (spos + used + (uint32_t)W > nsym) return;
            ar = issue(spos + used, (uint32_t)W);
            if (ar == 0) return;
        }
        RF_T(1);
        // (two instances of the predictor: the usual row has no raw pixel, and its instance no register for the raw bits -- the few
        // registers the widest class is short of; a spill's reload waits for every memory operation in flight, the row before's
        // stores included)
        bool fine;
        if (y == 0) { RP::slow_row(tp, es, raw, true, thr, lane, nl); fine = true; }
        else fine = RP::fast_row(tp, es, raw, thr, lane, nv);
        if (!fine) {                                                    // the wrap-around fired somewhere: the row above again, then lane by lane
            const uint32_t p = (uint32_t)(y - 1) * (uint32_t)W + x0;
#pragma unroll
            for (int q = 0; q < KD; q++) tp[q] = (uint32_t)px[min(p + 2 * q, npx - 1)] | ((uint32_t)px[min(p + 2 * q + 1, npx - 1)] << 16);
            RP::slow_row(tp, es, raw, false, thr, lane, nl);
            RF_ROW(3);
        }
        RF_T(2);
        put(y);
        RF_T(3);
        RF_ROW(0);
        spos += used;
    }
#ifdef RF_STATS
    if (lane == 0) { for (int i = 0; i < 4; i++) { u.dbg[i] = st_rows[i]; u.dbg[4 + i] = (uint32_t)(st_t[i] >> 4); } u.dbg[1] = (uint32_t)(st_t[4] >> 4); u.dbg[2] = (uint32_t)(st_t[5] >> 4); }
#endif
    if (lane == 0) { u.dec_thr = thr; u.walk_ok = 4; }                   // pixels done: the two-kernel path skips this unit
}

void mic_launch_decode_fused(MicUnit *d_units, int n, hipStream_t stream, uint32_t kmask) {
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
#define RF_LAUNCH(K_) if (kmask & (1u << (((K_) - 18) / 4))) \
        hipLaunchKernelGGL(k_dec_rows_tok<K_>, grid, block, 4 * 4 * ((32 * ((K_) + 8) + 4) + (32 * (K_) + 4) + (2 * (K_) + 4)), stream, d_units, n)
    RF_LAUNCH(18); RF_LAUNCH(22); RF_LAUNCH(26); RF_LAUNCH(30); RF_LAUNCH(34); RF_LAUNCH(38); RF_LAUNCH(42);
#undef RF_LAUNCH
}
