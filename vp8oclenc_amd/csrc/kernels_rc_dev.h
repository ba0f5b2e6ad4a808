// kernels_rc_dev.h -- the device side of the loop-filter strength scan (get_loopfilter_strength, src/vp8enc.cpp:96-127) and of
// prepare_segments_data (:129-221) behind it, shared by kernels_rc.hip (the scan as a launch of its own) and kernels_me.hip (the
// scan riding in a batch's pyramid launch: one link less in every frame's chain, see k_pyramid).
#pragma once
#include "vp8hip_dev.h"

namespace vp8 {
namespace rc {

constexpr int ROWS_PER_BLOCK = 8;     // k_lf_strength: 8 pixel rows x the whole width per workgroup
constexpr int MAX_PARTIALS = 2048;    // per quantity; height 8192 / 8 rows = 1024 workgroups

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *s_red) {   // sum over 256 threads, valid in thread 0
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t r = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    return r;
}

// partial[2b] = sum of Y, partial[2b+1] = sum over interior pixels of (p - (8 neighbours)/8)^2 for the rows of
// workgroup b.  Both are the reference's `int` accumulators, kept modulo 2^32 (order-independent), which is what
// their overflow does.  A thread owns four adjacent columns and slides a three-row window down its rows.
__device__ __forceinline__ void lf_strength_body(const Plane &y, uint32_t *partial, int wg) {   // wg: which ROWS_PER_BLOCK rows
    __shared__ uint32_t s_red[4];
    const int r0 = wg * ROWS_PER_BLOCK;
    uint32_t s = 0, d = 0;
    for (int x = threadIdx.x * 4; x < y.w; x += 1024) {
        // three dwords per row: columns x-4..x-1, x..x+3, x+4..x+7 (the margin makes every load legal)
        uint32_t a[3], b[3], c[3];
        const uint8_t *p = y.p + (ptrdiff_t)(r0 - 1) * y.stride + x - 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) { a[k] = *reinterpret_cast<const uint32_t *>(p + 4 * k); b[k] = *reinterpret_cast<const uint32_t *>(p + y.stride + 4 * k); }
        for (int r = r0; r < r0 + ROWS_PER_BLOCK && r < y.h; ++r) {
            const uint8_t *q = y.p + (ptrdiff_t)(r + 1) * y.stride + x - 4;
#pragma unroll
            for (int k = 0; k < 3; ++k) c[k] = *reinterpret_cast<const uint32_t *>(q + 4 * k);
            s = __builtin_amdgcn_sad_u8(b[1], 0u, s);
            if (r >= 1 && r < y.h - 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (x + i < 1 || x + i >= y.w - 1) continue;
                    // pixel i of the strip: bytes 3+i, 4+i, 5+i of the 12-byte rows
                    auto px = [&](const uint32_t (&w)[3], int j) { return byte_of(w[j >> 2], j & 3); };
                    const int nb = (px(a, 3 + i) + px(a, 4 + i) + px(a, 5 + i) + px(b, 3 + i) + px(b, 5 + i) + px(c, 3 + i) +
                                    px(c, 4 + i) + px(c, 5 + i)) / 8;
                    const int e = px(b, 4 + i) - nb;
                    d += (uint32_t)(e * e);
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { a[k] = b[k]; b[k] = c[k]; }
        }
    }
    s = block_sum(s, s_red);
    d = block_sum(d, s_red);
    if (threadIdx.x == 0) { partial[2 * wg] = s; partial[2 * wg + 1] = d; }
}
// stats[o], stats[o+1] = sums of the even / odd partials (one workgroup of 256)
__device__ __forceinline__ void fold(const uint32_t *partial, int nblocks, uint32_t *s_red, uint32_t &even, uint32_t &odd) {
    uint32_t e = 0, o = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { e += partial[2 * i]; o += partial[2 * i + 1]; }
    even = block_sum(e, s_red);
    odd = block_sum(o, s_red);
}
// get_loopfilter_strength's closing arithmetic (vp8enc.cpp:100-103,119-123) + prepare_segments_data
// (vp8enc.cpp:129-221) on the device: the frame loop then needs no host round trip for its parameters
struct SegArgs { int n, ni, is_key, q0, q1, q2, q3, qi_min; };
__device__ __forceinline__ void auto_segments_body(const uint32_t *partial, int nblocks, uint32_t *stats, SegData *sd,
                                                   int32_t *strength_out, const SegArgs &g) {
    __shared__ uint32_t s_red[4];
    const int n = g.n, ni = g.ni, is_key = g.is_key, q0 = g.q0, q1 = g.q1, q2 = g.q2, q3 = g.q3, qi_min = g.qi_min;
    uint32_t sum, dev;
    fold(partial, nblocks, s_red, sum, dev);
    if (threadIdx.x != 0) return;
    stats[0] = sum;
    stats[1] = dev;
    int avg = (int32_t)sum;
    avg += n / 2;
    avg /= n;
    const int reductor = (avg * 5 / 255) + 3;
    int div = (int32_t)dev;
    div += ni / 2;
    div /= ni;
    int sharpness = div / 8;
    sharpness = sharpness > 7 ? 7 : sharpness;
    strength_out[0] = reductor;
    strength_out[1] = sharpness;
    strength_out[2] = sharpness;                    // video.loop_filter_sharpness in force (check_SSIM may raise it to 7)
    const int refqi[4] = {q0, q1, q2, q3};
    fill_segment_data(sd, is_key, refqi, qi_min, reductor, sharpness, false);
}
// the scan and its closing arithmetic in ONE launch: the workgroup that finishes last (a counter that is zero at rest) folds
// the partial sums of all of them
struct ScanCore { uint32_t *partial, *done, *stats; SegData *sd; int32_t *strength_out; SegArgs g; };
struct StrengthItem { Plane y; ScanCore c; };
// wg of nwg workgroups of 256 threads, whatever launch they are part of
__device__ __forceinline__ void strength_segments_body(const Plane &y, const ScanCore &a, int wg, int nwg) {
    uint32_t *partial = a.partial, *done = a.done, *stats = a.stats;
    SegData *sd = a.sd;
    int32_t *strength_out = a.strength_out;
    const SegArgs &g = a.g;
    __shared__ uint32_t s_last;
    lf_strength_body(y, partial, wg);
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(done, 1u) == (uint32_t)nwg - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) *done = 0;
    auto_segments_body(partial, nwg, stats, sd, strength_out, g);
}

}  // namespace rc
}  // namespace vp8
