// ubench_chain.hip -- microbenchmarks of the serial tANS-decode chain step on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_chain ubench_chain.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// 1. pure dependent LDS chain
__global__ void __launch_bounds__(64) k_lds_chain(const uint32_t *tab, uint32_t *out, int iters, unsigned long long *cyc) {
    __shared__ uint32_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = tab[i];
    __syncthreads();
    uint32_t x = out[0] & 8191;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) x = s[x] & 8191;
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// 2. two interleaved chains
__global__ void __launch_bounds__(64) k_lds_chain2(const uint32_t *tab, uint32_t *out, int iters, unsigned long long *cyc) {
    __shared__ uint32_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = tab[i];
    __syncthreads();
    uint32_t x = out[0] & 8191, y = out[1] & 8191;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { uint32_t a = s[x], b = s[y]; x = a & 8191; y = (b + (a >> 13)) & 8191; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + y; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// 3. chain + bfe from a scalar window with readfirstlane + s_lshl (no refill)
__global__ void __launch_bounds__(64) k_chain_win(const uint32_t *tab, uint32_t *out, int iters, unsigned long long *cyc) {
    __shared__ uint32_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = tab[i];
    __syncthreads();
    uint32_t x = out[0] & 8191, y = out[1] & 8191;
    uint64_t W = 0x123456789abcdef0ull;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        uint32_t a = s[x], b = s[y];
        uint32_t hi = (uint32_t)(W >> 32);
        uint32_t n0 = a & 15, n1 = b & 15, t = n0 + n1;
        x = ((a >> 8) + __builtin_amdgcn_ubfe(hi, 32 - n0, n0)) & 8191;
        y = ((b >> 8) + __builtin_amdgcn_ubfe(hi, 32 - t, n1)) & 8191;
        uint32_t T = __builtin_amdgcn_readfirstlane(t);
        W = (W << T) | (W >> (64 - T));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + y + (uint32_t)W; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// 4. same with the window in VGPRs (no scalar hop)
__global__ void __launch_bounds__(64) k_chain_vwin(const uint32_t *tab, uint32_t *out, int iters, unsigned long long *cyc) {
    __shared__ uint32_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = tab[i];
    __syncthreads();
    uint32_t x = out[0] & 8191, y = out[1] & 8191;
    uint64_t W = 0x123456789abcdef0ull + threadIdx.x * 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        uint32_t a = s[x], b = s[y];
        uint32_t hi = (uint32_t)(W >> 32);
        uint32_t n0 = a & 15, n1 = b & 15, t = n0 + n1;
        x = ((a >> 8) + __builtin_amdgcn_ubfe(hi, 32 - n0, n0)) & 8191;
        y = ((b >> 8) + __builtin_amdgcn_ubfe(hi, 32 - t, n1)) & 8191;
        W = (W << t) | (W >> (64 - t));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + y + (uint32_t)W; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// 5. dependent scalar-cache (s_load) chain through constant memory
__global__ void __launch_bounds__(64) k_sload_chain(const uint32_t *__restrict__ tab, uint32_t *out, int iters, unsigned long long *cyc) {
    uint32_t x = __builtin_amdgcn_readfirstlane(out[0] & 8191);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { x = __builtin_amdgcn_readfirstlane(x); x = tab[x] & 8191; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// 6. dependent global (L2) chain
__global__ void __launch_bounds__(64) k_gl_chain(const uint32_t *tab, uint32_t *out, int iters, unsigned long long *cyc) {
    uint32_t x = (out[0] + threadIdx.x * 0) & 8191;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) x = __builtin_nontemporal_load(&tab[x]) & 8191;
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}


// 7..: ablation of the real decode loop.  FEAT bit0: symbol reads + stage write, bit1: branch-free
// refill with readlane, bit2: refill di-- branch, bit3: flush branch + store
template <int FEAT>
__global__ void __launch_bounds__(64) k_abl(const uint32_t *tab, uint32_t *out, int iters, unsigned long long *cyc) {
    __shared__ uint32_t s[8192]; __shared__ uint16_t sym[8192]; __shared__ uint32_t stage[64];
    for (int i = threadIdx.x; i < 8192; i += 64) { uint32_t e = tab[i]; uint32_t nb = e & 15; s[i] = ((e >> 8) & 8191) << 16 | (32 - nb) << 8 | nb; sym[i] = (uint16_t)e; }
    __syncthreads();
    uint32_t x = (out[0] & 8191) << 2, y = (out[1] & 8191) << 2;
    uint64_t W = 0x123456789abcdef0ull; uint32_t avail = 64; int64_t di = 1000000;
    uint32_t buf_a = tab[threadIdx.x], buf_b = tab[threadIdx.x + 64];
    const char *cb = (const char *)s; const char *sb = (const char *)sym;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        uint32_t a = *(const uint32_t *)(cb + x), b = *(const uint32_t *)(cb + y);
        uint32_t hi = (uint32_t)(W >> 32);
        uint32_t n0 = a & 0xFF, n1 = b & 0xFF;
        if (FEAT & 1) {
            uint32_t sa = *(const uint16_t *)(sb + (x >> 1)), sbb = *(const uint16_t *)(sb + (y >> 1));
            stage[(i & 63)] = sa | (sbb << 16);
        }
        uint32_t b0 = hi >> ((a >> 8) & 0xFF);
        uint32_t b1 = (hi << n0) >> ((b >> 8) & 0xFF);
        x = ((b0 << 2) + ((a >> 16) << 2)) & 32764;
        y = ((b1 << 2) + ((b >> 16) << 2)) & 32764;
        uint32_t T = __builtin_amdgcn_readfirstlane(n0 + n1);
        if (FEAT & 2) {
            W <<= T; avail -= T;
            const uint32_t nd = __builtin_amdgcn_readlane(buf_a, (int)((uint32_t)di & 63u));
            const bool need = avail < 32;
            const uint64_t add = (uint64_t)nd << ((32u - avail) & 63u);
            W |= need ? add : 0ull;
            avail += need ? 32u : 0u;
            if (FEAT & 4) {
                if (need) {
                    if (((uint32_t)di & 63u) == 0u) { buf_a = buf_b; buf_b = tab[(threadIdx.x + (uint32_t)di) & 8191]; }
                    di--;
                }
            } else di -= need ? 1 : 0;
        } else {
            W = (W << T) | (W >> (64 - T));
        }
        if (FEAT & 8) {
            if (((i + 1) & 63) == 0) out[64 + ((i >> 6) & 7) * 64 + threadIdx.x] = stage[threadIdx.x];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + y + (uint32_t)W + buf_a + stage[threadIdx.x]; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K> int run(const char *name, K k, const uint32_t *d_tab, uint32_t *d_out, unsigned long long *d_cyc, int iters, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_tab, d_out, iters, d_cyc);   // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_tab, d_out, iters, d_cyc);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    printf("%-16s blocks=%4d  %8.2f ns/iter  %7.1f memtime-ticks/iter  (%.3f ms)\n", name, blocks, ms * 1e6 / iters, (double)cyc / iters, ms);
    return 0;
}

int main() {
    std::vector<uint32_t> tab(8192);
    uint32_t r = 12345;
    for (auto &v : tab) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0x1FFF00) | (1 + (r >> 28) % 12); }
    uint32_t *d_tab, *d_out; unsigned long long *d_cyc;
    CK(hipMalloc(&d_tab, 8192 * 4)); CK(hipMalloc(&d_out, 65536)); CK(hipMalloc(&d_cyc, 8));
    CK(hipMemcpy(d_tab, tab.data(), 8192 * 4, hipMemcpyHostToDevice)); CK(hipMemset(d_out, 0, 65536));
    const int iters = 1000000;
    run("abl0", k_abl<0>, d_tab, d_out, d_cyc, iters, 1);
    run("abl1(sym+stage)", k_abl<1>, d_tab, d_out, d_cyc, iters, 1);
    run("abl2(refill)", k_abl<2>, d_tab, d_out, d_cyc, iters, 1);
    run("abl3", k_abl<3>, d_tab, d_out, d_cyc, iters, 1);
    run("abl7(+di branch)", k_abl<7>, d_tab, d_out, d_cyc, iters, 1);
    run("abl15(+flush)", k_abl<15>, d_tab, d_out, d_cyc, iters, 1);
    run("abl8", k_abl<8>, d_tab, d_out, d_cyc, iters, 1);
    run("abl4|2", k_abl<6>, d_tab, d_out, d_cyc, iters, 1);
    for (int blocks : {1}) {
        run("lds_chain", k_lds_chain, d_tab, d_out, d_cyc, iters, blocks);
        run("lds_chain2", k_lds_chain2, d_tab, d_out, d_cyc, iters, blocks);
        run("chain_win", k_chain_win, d_tab, d_out, d_cyc, iters, blocks);
        run("chain_vwin", k_chain_vwin, d_tab, d_out, d_cyc, iters, blocks);
        run("sload_chain", k_sload_chain, d_tab, d_out, d_cyc, iters, blocks);
        run("gl_chain", k_gl_chain, d_tab, d_out, d_cyc, iters / 4, blocks);
    }
    return 0;
}
