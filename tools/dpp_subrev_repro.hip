// Reproducer: on ROCm 7.2 / gfx950, v_subrev_u32_dpp computes dpp(src1) - src0, not src1 - dpp(src0) as LLVM assumes when it folds a
// lane permutation into `x - dpp(y)`; v_mov / v_sub / v_add DPP forms are as documented.  Build: hipcc --offload-arch=gfx950 -O3.
// Expected on a correct toolchain: subrev_dpp = q - t[3 of the quad] (996, 1996, ...).  Measured on MI355X: 3999, 3998, 3997, 3996.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t *in, const uint32_t *qin, uint32_t *o) {
    const uint32_t lane = threadIdx.x;
    uint32_t t = in[lane], q = qin[lane], r0, r1, r2, r3;
    asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r0) : "v"(t));
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r1) : "v"(t), "v"(q));
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r2) : "v"(t), "v"(q));
    asm volatile("s_nop 4\n\tv_add_u32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r3) : "v"(t), "v"(q));
    o[lane] = r0; o[64 + lane] = r1; o[128 + lane] = r2; o[192 + lane] = r3;
}
int main() {
    uint32_t h[64], q[64], a[256], *d, *dq, *da;
    for (int i = 0; i < 64; i++) { h[i] = i + 1; q[i] = 1000 * (i + 1); }
    (void)hipMalloc(&d, 256); (void)hipMalloc(&dq, 256); (void)hipMalloc(&da, 1024);
    (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dq, q, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, dq, da);
    (void)hipMemcpy(a, da, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; i++) printf("lane %d: t=%u q=%u | mov_dpp=%u subrev_dpp=%d sub_dpp=%d add_dpp=%u\n", i, h[i], q[i], a[i], (int)a[64 + i], (int)a[128 + i], a[192 + i]);
}
