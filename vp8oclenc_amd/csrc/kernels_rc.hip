// kernels_rc.hip -- the two O(pixels) host scans the reference runs per frame to produce parameters of the
// inter path: get_loopfilter_strength (src/vp8enc.cpp:96-127) and the chroma differences of
// scene_change (:265-282).  At the frame rates of this path (0.2 ms per 1080p frame) a single-threaded scan of
// 2 Mpixel on the host costs 25-50 frame times, and the pixels are already in HBM -- so they are two reductions
// on the device copy of the current frame.
#include "vp8hip_dev.h"

namespace vp8 {

namespace {

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
    return v;
}

// out[0] += sum of Y, out[1] += sum over interior pixels of (p - (8 neighbours)/8)^2.  Both are the reference's
// `int` accumulators: they are kept modulo 2^32 here (order-independent), which is what its overflow does.
__global__ __launch_bounds__(256) void k_lf_strength(Plane y, uint32_t *out) {
    const int row = blockIdx.y;
    const uint8_t *p = y.p + (ptrdiff_t)row * y.stride;
    uint32_t s = 0, d = 0;
    const bool inner_row = row >= 1 && row < y.h - 1;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < y.w; x += gridDim.x * 256) {
        const int c = p[x];
        s += (uint32_t)c;
        if (inner_row && x >= 1 && x < y.w - 1) {
            const uint8_t *u = p - y.stride, *w = p + y.stride;
            const int a = (u[x - 1] + u[x] + u[x + 1] + p[x - 1] + p[x + 1] + w[x - 1] + w[x] + w[x + 1]) / 8;
            d += (uint32_t)((c - a) * (c - a));
        }
    }
    s = wave_sum(s);
    d = wave_sum(d);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], s);
        atomicAdd(&out[1], d);
    }
}

// out[2] += sum |a.U - b.U|, out[3] += sum |a.V - b.V| (four pixels per thread)
__global__ __launch_bounds__(256) void k_chroma_sad(Plane au, Plane av, Plane bu, Plane bv, uint32_t *out) {
    const int pl = blockIdx.z, row = blockIdx.y;
    const Plane &a = pl == 0 ? au : av, &b = pl == 0 ? bu : bv;
    uint32_t s = 0;
    for (int x = (blockIdx.x * 256 + threadIdx.x) * 4; x < a.w; x += gridDim.x * 1024) {
        const uint32_t va = *reinterpret_cast<const uint32_t *>(a.p + (ptrdiff_t)row * a.stride + x);
        const uint32_t vb = *reinterpret_cast<const uint32_t *>(b.p + (ptrdiff_t)row * b.stride + x);
        s = __builtin_amdgcn_sad_u8(va, vb, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[2 + pl], s);
}

}  // namespace

void launch_lf_strength(hipStream_t s, const Frame &cur, uint32_t *stats) {
    hipMemsetAsync(stats, 0, 8, s);
    hipLaunchKernelGGL(k_lf_strength, dim3((cur.Y[0].w + 1023) / 1024, cur.Y[0].h), dim3(256), 0, s, cur.Y[0], stats);
}

void launch_chroma_sad(hipStream_t s, const Frame &cur, const Frame &prev, uint32_t *stats) {
    hipMemsetAsync(stats + 2, 0, 8, s);
    hipLaunchKernelGGL(k_chroma_sad, dim3((cur.U.w + 1023) / 1024, cur.U.h, 2), dim3(256), 0, s, cur.U, cur.V, prev.U, prev.V, stats);
}

}  // namespace vp8
