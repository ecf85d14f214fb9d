// ubench_pair.hip -- how one wave alone on its SIMD overlaps two tANS state chains (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_pair ubench_pair.hip ; run on the GPU box.
// Every variant walks a 8192-entry u16 table in LDS: e = T[s]; m = C - clz(e); s' = alignbit(e, h, m); address = 2 s' + base.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define WAVE_LDS 0x9000

template <int V>
__global__ void __launch_bounds__(256) k_pair(const uint16_t *tab, uint32_t *out, int iters, unsigned long long *cyc) {
    extern __shared__ uint32_t s_mem[];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t base = wv * WAVE_LDS + 3584u;                 // table of 8192 u16 (states 8192..16383 -> entry s - 8192)
    for (uint32_t i = lane; i < 4096; i += 64) s_mem[(base >> 2) + i] = ((const uint32_t *)tab)[i];
    __syncthreads();
    const uint32_t cb = base - 2u * 8192u;
    const uint32_t C = 31u - 13u;
    uint32_t sA = 8192u + (out[0] & 8191u), sB = 8192u + (out[1] & 8191u);
    uint32_t h = 0x9E3779B9u + lane * 0u, stage = wv * WAVE_LDS + 1056u, ring = wv * WAVE_LDS + 16u;
    uint32_t eA, eB, t, t2, aA, aB, x0 = 1, x1 = 2, x2 = 3, x3 = 4, wcv = 0;
    int n = iters;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (V == 1) {          // one chain
        asm volatile(
            "v_lshl_add_u32 %[aA], %[sA], 1, %[cb]\n\tds_read_u16 %[eA], %[aA]\n"
            "1:\n\ts_waitcnt lgkmcnt(0)\n\t"
            "v_ffbh_u32 %[t], %[eA]\n\tv_sub_u32 %[t], %[C], %[t]\n\tv_alignbit_b32 %[sA], %[eA], %[h], %[t]\n\t"
            "v_lshl_add_u32 %[aA], %[sA], 1, %[cb]\n\tds_read_u16 %[eA], %[aA]\n\t"
            "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)"
            : [sA] "+v"(sA), [eA] "=&v"(eA), [t] "=&v"(t), [aA] "=&v"(aA), [n] "+s"(n) : [cb] "v"(cb), [C] "v"(C), [h] "v"(h) : "scc");
    } else if (V == 2) {   // two chains, staggered: each waits only for its own entry
        asm volatile(
            "v_lshl_add_u32 %[aA], %[sA], 1, %[cb]\n\tds_read_u16 %[eA], %[aA]\n\t"
            "v_lshl_add_u32 %[aB], %[sB], 1, %[cb]\n\tds_read_u16 %[eB], %[aB]\n"
            "1:\n\ts_waitcnt lgkmcnt(1)\n\t"
            "v_ffbh_u32 %[t], %[eA]\n\tv_sub_u32 %[t], %[C], %[t]\n\tv_alignbit_b32 %[sA], %[eA], %[h], %[t]\n\t"
            "v_lshl_add_u32 %[aA], %[sA], 1, %[cb]\n\tds_read_u16 %[eA], %[aA]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_ffbh_u32 %[t2], %[eB]\n\tv_sub_u32 %[t2], %[C], %[t2]\n\tv_alignbit_b32 %[sB], %[eB], %[h], %[t2]\n\t"
            "v_lshl_add_u32 %[aB], %[sB], 1, %[cb]\n\tds_read_u16 %[eB], %[aB]\n\t"
            "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)"
            : [sA] "+v"(sA), [sB] "+v"(sB), [eA] "=&v"(eA), [eB] "=&v"(eB), [t] "=&v"(t), [t2] "=&v"(t2), [aA] "=&v"(aA), [aB] "=&v"(aB), [n] "+s"(n)
            : [cb] "v"(cb), [C] "v"(C), [h] "v"(h) : "scc");
    } else if (V == 3) {   // two chains, both entries requested together (the shape of the LDS-window loop)
        asm volatile(
            "1:\n\t"
            "v_lshl_add_u32 %[aA], %[sA], 1, %[cb]\n\tv_lshl_add_u32 %[aB], %[sB], 1, %[cb]\n\t"
            "ds_read_u16 %[eA], %[aA]\n\tds_read_u16 %[eB], %[aB]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_ffbh_u32 %[t], %[eA]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_ffbh_u32 %[t2], %[eB]\n\tv_sub_u32 %[t], %[C], %[t]\n\tv_sub_u32 %[t2], %[C], %[t2]\n\t"
            "v_alignbit_b32 %[sA], %[eA], %[h], %[t]\n\tv_alignbit_b32 %[sB], %[eB], %[h], %[t2]\n\t"
            "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b"
            : [sA] "+v"(sA), [sB] "+v"(sB), [eA] "=&v"(eA), [eB] "=&v"(eB), [t] "=&v"(t), [t2] "=&v"(t2), [aA] "=&v"(aA), [aB] "=&v"(aB), [n] "+s"(n)
            : [cb] "v"(cb), [C] "v"(C), [h] "v"(h) : "scc");
    } else if (V == 4 || V == 5 || V == 6) {   // staggered + the rest of a pair: 7 VALU of window upkeep, 2 stage writes, 1 window read
        // V == 5: the same without the two stage writes ; V == 6: without the window read
        asm volatile(
            "v_lshl_add_u32 %[aA], %[sA], 1, %[cb]\n\tds_read_u16 %[eA], %[aA]\n\t"
            "v_lshl_add_u32 %[aB], %[sB], 1, %[cb]\n\tds_read_u16 %[eB], %[aB]\n\t"
            "ds_read_b32 %[wc], %[ring]\n"
            "1:\n\t"
            ".if %[V] == 6\n\ts_waitcnt lgkmcnt(3)\n\t.elseif %[V] == 5\n\ts_waitcnt lgkmcnt(2)\n\t.else\n\ts_waitcnt lgkmcnt(4)\n\t.endif\n\t"
            "v_ffbh_u32 %[t], %[eA]\n\tv_sub_u32 %[t], %[C], %[t]\n\tv_alignbit_b32 %[sA], %[eA], %[h], %[t]\n\t"
            "v_lshl_add_u32 %[aA], %[sA], 1, %[cb]\n\tv_alignbit_b32 %[x0], %[h], 0, %[t]\n\tds_read_u16 %[eA], %[aA]\n\t"
            ".if %[V] == 6\n\ts_waitcnt lgkmcnt(3)\n\t.elseif %[V] == 5\n\ts_waitcnt lgkmcnt(2)\n\t.else\n\ts_waitcnt lgkmcnt(4)\n\t.endif\n\t"
            "v_ffbh_u32 %[t2], %[eB]\n\tv_sub_u32 %[t2], %[C], %[t2]\n\tv_alignbit_b32 %[sB], %[eB], %[x0], %[t2]\n\t"
            "v_lshl_add_u32 %[aB], %[sB], 1, %[cb]\n\tds_read_u16 %[eB], %[aB]\n\t"
            ".if %[V] != 6\n\ts_waitcnt lgkmcnt(2)\n\t.endif\n\t"
            "v_add3_u32 %[x1], %[t], %[x1], %[t2]\n\tv_bfe_u32 %[x2], %[x1], 5, 8\n\tv_cmp_eq_u32 vcc, %[x2], %[x3]\n\t"
            "v_cndmask_b32 %[x3], %[x2], %[x3], vcc\n\tv_cndmask_b32 %[x0], %[wc], %[x0], vcc\n\t"
            "v_lshl_add_u32 %[t], %[x2], 2, %[ring]\n\tv_alignbit_b32 %[h], %[x3], %[x0], %[x1]\n\t"
            ".if %[V] != 5\n\tds_write_b16 %[stage], %[sA]\n\tds_write_b16 %[stage], %[sB] offset:128\n\t.endif\n\t"
            ".if %[V] != 6\n\tds_read_b32 %[wc], %[ring]\n\t.endif\n\t"
            "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)"
            : [sA] "+v"(sA), [sB] "+v"(sB), [eA] "=&v"(eA), [eB] "=&v"(eB), [t] "=&v"(t), [t2] "=&v"(t2), [aA] "=&v"(aA), [aB] "=&v"(aB), [n] "+s"(n),
              [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [wc] "+v"(wcv), [h] "+v"(h)
            : [cb] "v"(cb), [C] "v"(C), [stage] "v"(stage), [ring] "v"(ring), [V] "n"(V) : "scc", "vcc");
    } else if (V == 7) {   // pure VALU: 16 dependent-in-pairs instructions, no LDS (issue cost per instruction)
        asm volatile(
            "1:\n\t"
            "v_ffbh_u32 %[t], %[sA]\n\tv_ffbh_u32 %[t2], %[sB]\n\tv_sub_u32 %[t], %[C], %[t]\n\tv_sub_u32 %[t2], %[C], %[t2]\n\t"
            "v_alignbit_b32 %[sA], %[sA], %[h], %[t]\n\tv_alignbit_b32 %[sB], %[sB], %[h], %[t2]\n\t"
            "v_lshl_add_u32 %[aA], %[sA], 1, %[cb]\n\tv_lshl_add_u32 %[aB], %[sB], 1, %[cb]\n\t"
            "v_ffbh_u32 %[t], %[aA]\n\tv_ffbh_u32 %[t2], %[aB]\n\tv_sub_u32 %[t], %[C], %[t]\n\tv_sub_u32 %[t2], %[C], %[t2]\n\t"
            "v_alignbit_b32 %[sA], %[sA], %[h], %[t]\n\tv_alignbit_b32 %[sB], %[sB], %[h], %[t2]\n\t"
            "v_lshl_add_u32 %[aA], %[sA], 1, %[cb]\n\tv_lshl_add_u32 %[aB], %[sB], 1, %[cb]\n\t"
            "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b"
            : [sA] "+v"(sA), [sB] "+v"(sB), [t] "=&v"(t), [t2] "=&v"(t2), [aA] "=&v"(aA), [aB] "=&v"(aB), [n] "+s"(n)
            : [cb] "v"(cb), [C] "v"(C), [h] "v"(h) : "scc");
        eA = aA; eB = aB;
    } else if (V == 8) {   // pure VALU: 16 instructions in ONE dependent chain
        asm volatile(
            "1:\n\t"
            "v_ffbh_u32 %[t], %[sA]\n\tv_sub_u32 %[t], %[C], %[t]\n\tv_alignbit_b32 %[sA], %[sA], %[h], %[t]\n\tv_lshl_add_u32 %[sA], %[sA], 1, %[cb]\n\t"
            "v_ffbh_u32 %[t], %[sA]\n\tv_sub_u32 %[t], %[C], %[t]\n\tv_alignbit_b32 %[sA], %[sA], %[h], %[t]\n\tv_lshl_add_u32 %[sA], %[sA], 1, %[cb]\n\t"
            "v_ffbh_u32 %[t], %[sA]\n\tv_sub_u32 %[t], %[C], %[t]\n\tv_alignbit_b32 %[sA], %[sA], %[h], %[t]\n\tv_lshl_add_u32 %[sA], %[sA], 1, %[cb]\n\t"
            "v_ffbh_u32 %[t], %[sA]\n\tv_sub_u32 %[t], %[C], %[t]\n\tv_alignbit_b32 %[sA], %[sA], %[h], %[t]\n\tv_lshl_add_u32 %[sA], %[sA], 1, %[cb]\n\t"
            "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b"
            : [sA] "+v"(sA), [t] "=&v"(t), [n] "+s"(n) : [cb] "v"(cb), [C] "v"(C), [h] "v"(h) : "scc");
        eA = sA; eB = sB;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = sA + sB + eA + eB + x0 + x1 + x2 + x3 + wcv + h;
    if (lane == 0) cyc[wv] = t1 - t0;
}

template <int V> int run(const char *name, const uint16_t *d_tab, uint32_t *d_out, unsigned long long *d_cyc, int waves) {
    const int iters = 20000;
    CK(hipFuncSetAttribute((const void *)k_pair<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WAVE_LDS));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_pair<V>, dim3(1), dim3(64 * waves), 4 * WAVE_LDS, 0, d_tab, d_out, iters, d_cyc);
        CK(hipDeviceSynchronize());
    }
    unsigned long long c[4];
    CK(hipMemcpy(c, d_cyc, sizeof c, hipMemcpyDeviceToHost));
    printf("%-58s waves %d: %.1f ticks per iteration\n", name, waves, (double)c[0] / iters);
    return 0;
}

int main() {
    std::vector<uint16_t> tab(8192);
    uint32_t x = 12345;
    for (auto &v : tab) { x = x * 1664525u + 1013904223u; v = (uint16_t)(1 + (x >> 8) % 16383); }
    uint16_t *d_tab; uint32_t *d_out; unsigned long long *d_cyc;
    CK(hipMalloc(&d_tab, 16384)); CK(hipMalloc(&d_out, 4096)); CK(hipMalloc(&d_cyc, 64));
    CK(hipMemcpy(d_tab, tab.data(), 16384, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0, 4096));
    for (int waves : { 1, 4 }) {
        if (run<1>("1 one chain (5 instr + wait)", d_tab, d_out, d_cyc, waves)) return 1;
        if (run<2>("2 two chains staggered (10 instr + 2 waits)", d_tab, d_out, d_cyc, waves)) return 1;
        if (run<3>("3 two chains, entries requested together (10 + 2 waits)", d_tab, d_out, d_cyc, waves)) return 1;
        if (run<4>("4 staggered + window upkeep + 2 writes + window read", d_tab, d_out, d_cyc, waves)) return 1;
        if (run<5>("5 as 4 without the stage writes", d_tab, d_out, d_cyc, waves)) return 1;
        if (run<6>("6 as 4 without the window read", d_tab, d_out, d_cyc, waves)) return 1;
        if (run<7>("7 16 VALU, two interleaved chains, no LDS", d_tab, d_out, d_cyc, waves)) return 1;
        if (run<8>("8 16 VALU, one dependent chain, no LDS", d_tab, d_out, d_cyc, waves)) return 1;
    }
    return 0;
}
