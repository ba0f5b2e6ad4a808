// Micro-benchmark: issue cost of the integer VALU instructions the search/loop-filter kernels lean on,
// for 1..8 waves per SIMD (gfx950).  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(uint32_t *out, unsigned long long *cyc, int iters) {
    uint32_t a = threadIdx.x * 7 + 1, b = threadIdx.x * 13 + 5, c = threadIdx.x ^ 0x55, d = threadIdx.x + 99;
    uint32_t e = a + 1, f = b + 2, g = c + 3, h = d + 4;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 1) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 2) { REP16(asm volatile("v_dot2_i32_i16 %0, %1, %2, %0\n v_dot2_i32_i16 %3, %4, %5, %3\n v_dot2_i32_i16 %6, %1, %5, %6\n v_dot2_i32_i16 %7, %4, %2, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 3) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %3, %3, %4, %2\n v_perm_b32 %5, %5, %6, %2\n v_perm_b32 %7, %7, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 4) { REP16(asm volatile("v_mad_i32_i24 %0, %0, %1, %2\n v_mad_i32_i24 %3, %3, %4, %2\n v_mad_i32_i24 %5, %5, %6, %2\n v_mad_i32_i24 %7, %7, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 5) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 6) { REP16(asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_3\n v_sub_u32_sdwa %6, %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n v_sub_u32_sdwa %7, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 7) { REP16(asm volatile("v_pk_add_i16 %0, %0, %1\n v_pk_add_i16 %2, %2, %3\n v_pk_add_i16 %4, %4, %5\n v_pk_add_i16 %6, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 8) { REP16(asm volatile("v_sad_u16 %0, %0, %1, %2\n v_sad_u16 %3, %3, %4, %2\n v_sad_u16 %5, %5, %6, %2\n v_sad_u16 %7, %7, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 9) { REP16(asm volatile("v_med3_i32 %0, %0, %1, %2\n v_med3_i32 %3, %3, %4, %2\n v_med3_i32 %5, %5, %6, %2\n v_med3_i32 %7, %7, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 10) { REP16(asm volatile("v_dot4_i32_i8 %0, %1, %2, %0\n v_dot4_i32_i8 %3, %4, %5, %3\n v_dot4_i32_i8 %6, %1, %5, %6\n v_dot4_i32_i8 %7, %4, %2, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 11) { REP16(asm volatile("v_alignbyte_b32 %0, %0, %1, %2\n v_alignbyte_b32 %3, %3, %4, %2\n v_alignbyte_b32 %5, %5, %6, %2\n v_alignbyte_b32 %7, %7, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 12) { REP16(asm volatile("v_ashrrev_i32 %0, 3, %0\n v_ashrrev_i32 %1, 3, %1\n v_ashrrev_i32 %2, 3, %2\n v_ashrrev_i32 %3, 3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 13) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %3, %3, %4, %2\n v_add3_u32 %5, %5, %6, %2\n v_add3_u32 %7, %7, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char *name) {
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 8 * 64 * 4 * 4); hipMalloc(&cyc, 4096 * 8);
    const int iters = 200;
    printf("%-16s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {      // waves per SIMD: block = 256*wps threads on one CU... use 256 blocks (1/CU)
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * (wps > 4 ? 4 : wps)), 0, 0, out, cyc, iters);   // up to 1024 threads = 4 waves/SIMD
        hipDeviceSynchronize();
        unsigned long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        const int w = wps > 4 ? 4 : wps;
        printf("  %dw/simd: %.2f cyc/instr/wave, %.2f cyc/instr/SIMD", w, avg / (iters * 64.0), avg / (iters * 64.0) / w);
        if (wps >= 4) break;
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("v_add_u32 indep"); run<1>("v_add_u32 dep"); run<2>("v_dot2_i32_i16"); run<3>("v_perm_b32"); run<4>("v_mad_i32_i24");
    run<5>("v_mul_lo_u32"); run<6>("v_sub_sdwa"); run<7>("v_pk_add_i16"); run<8>("v_sad_u16"); run<9>("v_med3_i32");
    run<10>("v_dot4_i32_i8"); run<11>("v_alignbyte"); run<12>("v_ashrrev"); run<13>("v_add3_u32");
    return 0;
}
