// lone_wave.hip -- what ONE wave on a SIMD pays per instruction when its instructions depend on each other: the regime of the loop
// filter's worker waves (kernels_lf4.hip), whose frame time is (dependent steps) x (cycles the chain's wave needs per step).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/lone_wave.hip -o scripts/ubench/bin/lone_wave && scripts/ubench/bin/lone_wave
// One workgroup of one wave per launch (nothing else on the part); s_memtime around 64 x 64 instructions; cycles per instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define T0 asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1 asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory")

template <int OP>
__global__ void k(uint32_t *out, unsigned long long *cyc, int iters) {
    __shared__ uint32_t lds[4096];
    uint32_t a = threadIdx.x * 7 + 1, b = threadIdx.x * 13 + 5, c = threadIdx.x ^ 0x55, d = threadIdx.x + 99;
    uint32_t e = a + 1, f = b + 2, g = c + 3, h = d + 4;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u q4 = {a, b, c, d};
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 4 + 256) & 0x3fff);      // a byte-offset chain for the pointer chase
    __syncthreads();
    uint32_t p = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds + threadIdx.x * 4;
    unsigned long long t0, t1;
    T0;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
        if (OP == 2) { REP16(asm volatile("v_med3_i32 %0, %0, %1, %2\n v_med3_i32 %0, %0, %1, %2\n v_med3_i32 %0, %0, %1, %2\n v_med3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (OP == 3) { REP16(asm volatile("v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %4, %5\n v_med3_i32 %2, %2, %4, %5\n v_med3_i32 %3, %3, %4, %5" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));) }
        if (OP == 4) { REP16(asm volatile("v_sad_u16 %0, %0, %1, %2\n v_sad_u16 %0, %0, %1, %2\n v_sad_u16 %0, %0, %1, %2\n v_sad_u16 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        // compare -> select through VCC, the selected value feeding the next compare (what `mask ? w : 0` compiles to)
        if (OP == 5) { REP16(asm volatile("v_cmp_le_i32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %0, vcc\n v_cmp_le_i32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %0, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");) }
        // ... through an SGPR pair (VOP3 forms)
        if (OP == 6) { REP16(asm volatile("v_cmp_le_i32 s[20:21], %0, %1\n v_cndmask_b32 %0, %2, %0, s[20:21]\n v_cmp_le_i32 s[20:21], %0, %1\n v_cndmask_b32 %0, %2, %0, s[20:21]" : "+v"(a) : "v"(b), "v"(c) : "s20", "s21");) }
        // the same selection by arithmetic: m = (b - a) >> 31 (all ones where a > b); a = a & ~m | c & m  -> sub, ashr, bfi
        if (OP == 7) { REP16(asm volatile("v_sub_u32 %3, %1, %0\n v_ashrrev_i32 %3, 31, %3\n v_bfi_b32 %0, %3, %2, %0\n v_sub_u32 %3, %1, %0\n v_ashrrev_i32 %3, 31, %3\n v_bfi_b32 %0, %3, %2, %0" : "+v"(a) : "v"(b), "v"(c), "v"(d));) }
        // compares whose selects do NOT depend on them immediately: two independent compare/select pairs interleaved
        if (OP == 8) { REP16(asm volatile("v_cmp_le_i32 vcc, %0, %2\n v_cmp_le_i32 s[20:21], %1, %2\n v_cndmask_b32 %0, %3, %0, vcc\n v_cndmask_b32 %1, %3, %1, s[20:21]" : "+v"(a), "+v"(d) : "v"(b), "v"(c) : "vcc", "s20", "s21");) }
        // a typical filter slice: sub, med3, mad_i24, med3, add, ashr, sub  (all dependent)
        if (OP == 9) { REP16(asm volatile("v_sub_u32 %0, %0, %1\n v_med3_i32 %0, %0, %2, %3\n v_mad_i32_i24 %0, %1, 3, %0\n v_med3_i32 %0, %0, %2, %3\n v_add_u32 %0, 4, %0\n v_ashrrev_i32 %0, 3, %0\n v_sub_u32 %0, %1, %0\n v_max3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c), "v"(d));) }
        // LDS: dependent ds_read_b32 (pointer chase), dependent ds_read_b128, write + fence
        if (OP == 10) { REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(p)::"memory");) }
        if (OP == 11) { REP16(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(q4) : "v"(p) : "memory");) }
        if (OP == 12) { REP16(asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(p), "v"(a) : "memory");) }
        if (OP == 13) { REP16(asm volatile("ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(p), "v"(q4) : "memory");) }
        // twenty ds_read_b32 at immediate offsets, one wait (P2's load), per 4: counted as 4 "instructions" = 5 reads + wait
        if (OP == 14) { REP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:528\n ds_read_b32 %2, %4 offset:1056\n ds_read_b32 %3, %4 offset:1584\n ds_read_b32 %0, %4 offset:2112\n s_waitcnt lgkmcnt(0)" : "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h) : "v"(p) : "memory");) }
        // v_readfirstlane -> scalar use -> VALU (the poll's pattern)
        if (OP == 15) { REP16(asm volatile("v_readfirstlane_b32 s20, %0\n s_add_u32 s20, s20, 1\n v_add_u32 %0, s20, %0\n v_readfirstlane_b32 s20, %0\n s_add_u32 s20, s20, 1\n v_add_u32 %0, s20, %0" : "+v"(a)::"s20");) }
        // packed 16-bit, dependent
        if (OP == 16) { REP16(asm volatile("v_pk_add_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %2\n v_pk_sub_i16 %0, %0, %1\n v_pk_min_i16 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));) }
    }
    T1;
    out[threadIdx.x] = a + b + c + d + e + f + g + h + p + q4.x + q4.w;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char *name, int per_rep) {
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    const int iters = 64;
    unsigned long long best = ~0ull;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        if (h < best) best = h;
    }
    printf("%-52s %8.2f shader cycles per instruction\n", name, (double)best / (iters * 16.0 * per_rep));
    fflush(stdout);
    hipFree(out); hipFree(cyc);
}
int main(int argc, char **argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
#define RUN(n, name, per) if (only < 0 || only == n) run<n>(name, per)
    RUN(0, "v_add_u32, dependent", 4);
    RUN(1, "v_add_u32, four independent chains", 4);
    RUN(2, "v_med3_i32, dependent", 4);
    RUN(3, "v_med3_i32, four independent chains", 4);
    RUN(4, "v_sad_u16, dependent", 4);
    RUN(5, "v_cmp -> v_cndmask through vcc, dependent", 4);
    RUN(6, "v_cmp -> v_cndmask through s[20:21], dependent", 4);
    RUN(7, "sub / ashr 31 / bfi (select by arithmetic), dep", 6);
    RUN(8, "two cmp + two cndmask, interleaved", 4);
    RUN(9, "filter slice (sub med3 mad med3 add ashr sub max3)", 8);
    RUN(10, "ds_read_b32 pointer chase (read + wait)", 4);
    RUN(11, "ds_read_b128 + wait", 4);
    RUN(12, "ds_write_b32 + wait", 4);
    RUN(13, "ds_write_b128 + wait", 4);
    RUN(14, "five ds_read_b32 + one wait", 1);
    RUN(15, "readfirstlane -> s_add -> v_add, dependent", 6);
    RUN(16, "packed 16-bit add/max/sub/min, dependent", 4);
    return 0;
}
