// concurrency.hip -- how many kernels of different streams does the part run at once?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/concurrency.hip -o scripts/ubench/bin/concurrency; GPU_MAX_HW_QUEUES=24 scripts/ubench/bin/concurrency
// N streams, each gets `reps` launches of a kernel that spins for `us` microseconds on `wgs` workgroups.  If C of them run at
// once the whole thing takes N * reps * us / C.
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void spin(unsigned long long ticks, unsigned *sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned x = threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 12345u) *sink = x;
}
static double run(int nstreams, int wgs, int us, int reps, std::vector<hipStream_t> &st, unsigned *sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
        for (int s = 0; s < nstreams; ++s) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, st[s], (unsigned long long)us * 100, sink);
    hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return ms;
}
int main() {
    unsigned *sink; hipMalloc(&sink, 4);
    std::vector<hipStream_t> st(64);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s\n", q ? q : "(default 4)");
    for (int wgs : {1, 9, 256, 4096}) {
        for (int n : {1, 2, 4, 8, 16, 24, 32}) {
            const int us = 200, reps = 10;
            const double ms = run(n, wgs, us, reps, st, sink);
            printf("  %4d workgroups x %2d streams: %7.2f ms for %d x %d us each -> %.1f kernels at once\n", wgs, n, ms, reps, us, n * reps * us / 1000.0 / ms);
        }
    }
    return 0;
}
