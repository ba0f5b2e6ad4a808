// light_clock.hip -- which shader clock does a kernel of a few waves get?  A lone wave runs a dependent chain of
// 24-bit multiplies / adds and reads the shader-cycle counter (s_memtime) and the constant 100 MHz wall clock around it;
// then the same while a second stream keeps every CU busy.  Also: cycles per dependent instruction of that chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void chain(int n, unsigned *out, unsigned long long *t) {
    unsigned r = 200u + threadIdx.x, acc = 0;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {   // 6 dependent VALU instructions per iteration
        const unsigned s = 1u + (__umul24(r - 1u, 173u) >> 8);
        r = (r - s) | 128u;
        acc += r;
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = w1 - w0; }
}
__global__ void burn(int n, unsigned *out) {
    unsigned a = threadIdx.x, b = blockIdx.x;
    for (int i = 0; i < n; ++i) { a = a * 1664525u + b; b = b * 22695477u + a; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b;
}
int main() {
    unsigned *o1, *o2; unsigned long long *t;
    CK(hipMalloc(&o1, 1 << 20)); CK(hipMalloc(&o2, 64 << 20)); CK(hipMalloc(&t, 4096));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int N = 200000;
    for (int pass = 0; pass < 3; ++pass) {
        const bool loaded = pass == 1;
        if (loaded) for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(burn, dim3(4096), dim3(256), 0, s2, 60000, o2);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(chain, dim3(pass == 2 ? 64 : 1), dim3(64), 0, s1, N, o1, t);
        CK(hipStreamSynchronize(s1));
        unsigned long long h[2];
        CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
        CK(hipDeviceSynchronize());
        const double us = h[1] / 100.0;
        printf("%-34s: %.0f us, shader clock %.0f MHz, %.2f cycles per dependent instruction\n",
               pass == 0 ? "one wave, GPU otherwise idle" : (pass == 1 ? "one wave, every CU busy beside it" : "64 waves on 64 CUs, otherwise idle"),
               us, h[0] / us, (double)h[0] / (6.0 * N));
    }
    return 0;
}
