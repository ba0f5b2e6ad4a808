// lds_order.hip -- litmus: do the LDS writes of ONE wave become visible to another wave of the workgroup in program order WITHOUT
// an s_waitcnt between them?  (kernels_lf4.hip publishes a macroblock's samples and then a progress flag; with in-order LDS the
// producer need not wait for its stores to retire before it writes the flag.)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_order.hip -o /tmp/lds_order && /tmp/lds_order
// Producer wave: iteration i writes five ds_write_b128 + seventeen ds_write_b32 of the value i all over a 16 KB area, then the flag
// i -- no wait.  Consumer waves (the other SIMDs, and one on the producer's own SIMD): read the flag, then the data; data older
// than the flag that was seen is a violation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long *viol, unsigned long long *checks, int iters) {
    __shared__ __attribute__((aligned(16))) int data[4096 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096 + 64; i += blockDim.x) data[i] = 0;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)data;
    const uint32_t flag = base + 4096 * 4;
    if (wave == 0) {
        for (int i = 1; i <= iters; ++i) {
            const v4i v = {i, i, i, i};
            const uint32_t a = base + lane * 16;      // five b128 stores: 5 KB
            asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n ds_write_b128 %0, %1 offset:4096"
                         ::"v"(a), "v"(v) : "memory");
            const uint32_t b = base + 5120 + lane * 4;   // seventeen b32 stores at a row stride, like P2's
            asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:528\n ds_write_b32 %0, %1 offset:1056\n ds_write_b32 %0, %1 offset:1584\n"
                         "ds_write_b32 %0, %1 offset:2112\n ds_write_b32 %0, %1 offset:2640\n ds_write_b32 %0, %1 offset:3168\n ds_write_b32 %0, %1 offset:3696\n"
                         "ds_write_b32 %0, %1 offset:4224\n ds_write_b32 %0, %1 offset:4752\n ds_write_b32 %0, %1 offset:5280\n ds_write_b32 %0, %1 offset:5808\n"
                         "ds_write_b32 %0, %1 offset:6336\n ds_write_b32 %0, %1 offset:6864\n ds_write_b32 %0, %1 offset:7392\n ds_write_b32 %0, %1 offset:7920\n ds_write_b32 %0, %1 offset:8448"
                         ::"v"(b), "v"(i) : "memory");
            asm volatile("ds_write_b32 %0, %1" ::"v"(flag), "v"(i) : "memory");      // the flag: every lane, the same word, NO wait in front
            if ((i & 63) == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (only so that the counter does not saturate)
        }
        return;
    }
    unsigned long long bad = 0, n = 0;
    int seen = 0;
    while (seen < iters) {
        int f;
        asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(flag) : "memory");
        v4i d0, d1;
        int e0, e1;
        asm volatile("ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4\n ds_read_b32 %2, %5 offset:8448\n ds_read_b32 %3, %5\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(d0), "=&v"(d1), "=&v"(e0), "=&v"(e1) : "v"(base + lane * 16), "v"(base + 5120 + lane * 4) : "memory");
        bad += (d0.x < f) + (d0.w < f) + (d1.x < f) + (d1.w < f) + (e0 < f) + (e1 < f);
        n += 6;
        seen = __builtin_amdgcn_readfirstlane(f);
    }
    atomicAdd(viol, bad);
    atomicAdd(checks, n);
}
int main() {
    unsigned long long *d, h[2] = {0, 0};
    (void)hipMalloc(&d, 16);
    (void)hipMemset(d, 0, 16);
    for (int waves : {2, 4, 5, 8})       // 5: wave 4 shares the producer's SIMD
        hipLaunchKernelGGL(k, dim3(64), dim3(64 * waves), 0, 0, d, d + 1, 400000);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("LDS write order seen by other waves, no s_waitcnt between data and flag: %llu violations in %llu checks\n", h[0], h[1]);
    return h[0] != 0;
}
