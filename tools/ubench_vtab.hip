#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t v32u __attribute__((ext_vector_type(32)));
// table of 8192 u16 entries = 4096 dwords = 64 VGPRs x 64 lanes; entry i lives in dword i>>1: reg (i>>1)>>6, lane (i>>1)&63
__global__ void __launch_bounds__(64) k_vtab(const uint32_t *tab, uint32_t *out, int iters) {
    const uint32_t lane = threadIdx.x;
    v32u ta, tb;
#pragma unroll
    for (int r = 0; r < 32; r++) { ta[r] = tab[r * 64 + lane]; tb[r] = tab[(32 + r) * 64 + lane]; }
    uint32_t s0 = blockIdx.x & 8191, s1 = (blockIdx.x * 7 + 3) & 8191;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            uint32_t &s = k ? s1 : s0;
            const uint32_t d = s >> 1, r = d >> 6, l = d & 63;
            const uint32_t x = (r < 32) ? ta[r] : tb[r - 32];
            const uint32_t e2 = __builtin_amdgcn_readlane(x, l);
            const uint32_t e = (s & 1) ? (e2 >> 16) : (e2 & 0xFFFF);
            s = (e + i) & 8191;
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[blockIdx.x * 4] = s0; out[blockIdx.x * 4 + 1] = s1; out[blockIdx.x * 4 + 2] = (uint32_t)(t1 - t0); }
}
int main() {
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = (i * 2654435761u) & 0x1FFF1FFF;
    uint32_t *d_tab, *d_out; hipMalloc(&d_tab, 4096 * 4); hipMalloc(&d_out, 65536 * 16);
    hipMemcpy(d_tab, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int blocks : {1, 256, 1024, 2048, 4096, 5120}) {
        hipLaunchKernelGGL(k_vtab, dim3(blocks), dim3(64), 0, 0, d_tab, d_out, iters);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL(k_vtab, dim3(blocks), dim3(64), 0, 0, d_tab, d_out, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<uint32_t> o(blocks * 4); hipMemcpy(o.data(), d_out, blocks * 16, hipMemcpyDeviceToHost);
        printf("blocks %5d: %.3f ms, ticks/pair (wave 0) %.1f, pairs/s total %.3e\n", blocks, ms, (double)o[2] / iters, (double)blocks * iters / (ms * 1e-3));
    }
    return 0;
}
