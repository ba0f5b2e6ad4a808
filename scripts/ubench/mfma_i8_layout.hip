// mfma_i8_layout.hip -- the operand and result lane maps of v_mfma_i32_32x32x32_i8 checked with random signed bytes against a host loop
// (the quarter-pel search runs its six-tap passes on it, kernels_s2.hip): hipcc --offload-arch=gfx950 -O3 mfma_i8_layout.hip && ./a.out
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// A[32][32] int8 row-major, B[32][32] int8 (k rows, n cols) -> D[32][32] int32 by one v_mfma_i32_32x32x32_i8, assumed maps:
// lane l: A[row l&31][k = 16*(l>>5) + j], B[k = 16*(l>>5) + j][col l&31], j = 0..15; D reg r: row (r&3)+8*(r>>2)+4*(l>>5), col l&31
__global__ void k(const int8_t *A, const int8_t *B, int *D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v4i a, b;
    int8_t ta[16], tb[16];
    for (int j = 0; j < 16; ++j) { ta[j] = A[r * 32 + 16 * h + j]; tb[j] = B[(16 * h + j) * 32 + r]; }
    __builtin_memcpy(&a, ta, 16);
    __builtin_memcpy(&b, tb, 16);
    v16i c;
    for (int i = 0; i < 16; ++i) c[i] = 64;
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}
int main() {
    int8_t hA[1024], hB[1024]; int hD[1024], ref[1024];
    unsigned s = 12345;
    for (int i = 0; i < 1024; ++i) { s = s * 1664525u + 1013904223u; hA[i] = (int8_t)(s >> 24); s = s * 1664525u + 1013904223u; hB[i] = (int8_t)(s >> 24); }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { int acc = 64; for (int kk = 0; kk < 32; ++kk) acc += hA[m * 32 + kk] * hB[kk * 32 + n]; ref[m * 32 + n] = acc; }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
    printf("mismatches %d of 1024\n", bad);
    return bad != 0;
}
