// Micro-benchmark: chip-wide integer VALU throughput (wave64 instructions per ns) by wall clock (hipEvents) at
// 1, 2, 4, 8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_peak.hip -o valu_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x * 7 + 1, b = threadIdx.x * 13 + 5, c = threadIdx.x ^ 0x55, d = threadIdx.x + 99;
    uint32_t e = a + 1, f = b + 2, g = c + 3, h = d + 4;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 1) { REP16(asm volatile("v_dot4_i32_i8 %0, %1, %2, %0\n v_dot4_i32_i8 %3, %4, %5, %3\n v_dot4_i32_i8 %6, %1, %5, %6\n v_dot4_i32_i8 %7, %4, %2, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 2) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_sad_u16 %3, %3, %4, %2\n v_dot2_i32_i16 %5, %1, %6, %5\n v_add3_u32 %7, %7, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 3) { REP16(asm volatile("v_pk_add_i16 %0, %0, %1\n v_pk_add_i16 %2, %2, %3\n v_pk_add_i16 %4, %4, %5\n v_pk_add_i16 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}
template <int OP>
void run(const char *name) {
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-22s", name);
    for (int wgs = 1; wgs <= 8; wgs *= 2) {      // workgroups of 4 waves per CU = waves per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(256 * wgs), dim3(256), 0, 0, out, 10);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * wgs), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = 256.0 * wgs * 4 * iters * 64;
        printf("  %dw/SIMD: %.3f winstr/ns (%.2f ms)", wgs, instr / (ms * 1e6), ms);
    }
    printf("\n");
    hipFree(out);
}
int main() {
    run<0>("v_add_u32"); run<1>("v_dot4_i32_i8"); run<2>("perm/sad/dot2/add3 mix"); run<3>("v_pk_add_i16");
    return 0;
}
