// kernels_rc.hip -- the two O(pixels) host scans the reference runs per frame to produce parameters of the
// inter path: get_loopfilter_strength (src/vp8enc.cpp:96-127) and the chroma differences of
// scene_change (:265-282), plus prepare_segments_data (:129-221) chained behind the first one.  At the frame
// rates of this path (0.2 ms per 1080p frame) a single-threaded scan of 2 Mpixel on the host costs 25-50 frame
// times, and the pixels are already in HBM -- so they are reductions on the device copy of the current frame.
// No same-address atomics: every workgroup writes one partial sum, the consumer adds them up (a first version
// with one atomicAdd per wave onto two words cost 0.1 ms per frame).
#include <cstdlib>
#include <stdlib.h>
#include <string.h>

#include "vp8hip_dev.h"

namespace vp8 {

namespace {

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
__device__ __forceinline__ void lf_strength_body(const Plane &y, uint32_t *partial) {
    __shared__ uint32_t s_red[4];
    const int r0 = blockIdx.x * ROWS_PER_BLOCK;
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
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s; partial[2 * blockIdx.x + 1] = d; }
}
__global__ __launch_bounds__(256) void k_lf_strength(Plane y, uint32_t *partial) { lf_strength_body(y, partial); }

// partial[2b + plane] = sum |a - b| over the rows of workgroup b (8 chroma rows, four pixels per load)
__global__ __launch_bounds__(256) void k_chroma_sad(Plane au, Plane av, Plane bu, Plane bv, uint32_t *partial) {
    __shared__ uint32_t s_red[4];
    const int pl = blockIdx.y;
    const Plane &a = pl == 0 ? au : av, &b = pl == 0 ? bu : bv;
    const int r0 = blockIdx.x * ROWS_PER_BLOCK;
    uint32_t s = 0;
    for (int r = r0; r < r0 + ROWS_PER_BLOCK && r < a.h; ++r)
        for (int x = threadIdx.x * 4; x < a.w; x += 1024)
            s = __builtin_amdgcn_sad_u8(*reinterpret_cast<const uint32_t *>(a.p + (ptrdiff_t)r * a.stride + x),
                                        *reinterpret_cast<const uint32_t *>(b.p + (ptrdiff_t)r * b.stride + x), s);
    s = block_sum(s, s_red);
    if (threadIdx.x == 0) partial[2 * blockIdx.x + pl] = s;
}

// stats[o], stats[o+1] = sums of the even / odd partials (one workgroup of 256)
__device__ __forceinline__ void fold(const uint32_t *partial, int nblocks, uint32_t *s_red, uint32_t &even, uint32_t &odd) {
    uint32_t e = 0, o = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { e += partial[2 * i]; o += partial[2 * i + 1]; }
    even = block_sum(e, s_red);
    odd = block_sum(o, s_red);
}
__global__ __launch_bounds__(256) void k_fold(const uint32_t *partial, int nblocks, uint32_t *stats, int o) {
    __shared__ uint32_t s_red[4];
    uint32_t e, d;
    fold(partial, nblocks, s_red, e, d);
    if (threadIdx.x == 0) { stats[o] = e; stats[o + 1] = d; }
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
__global__ __launch_bounds__(256) void k_auto_segments(const uint32_t *partial, int nblocks, uint32_t *stats, SegData *sd,
                                                       int32_t *strength_out, SegArgs g) {
    auto_segments_body(partial, nblocks, stats, sd, strength_out, g);
}
// the scan and its closing arithmetic in ONE launch: the workgroup that finishes last (a counter that is zero at rest) folds
// the partial sums of all of them
struct StrengthItem { Plane y; uint32_t *partial, *done, *stats; SegData *sd; int32_t *strength_out; SegArgs g; };
__device__ __forceinline__ void strength_segments_body(const StrengthItem &a) {
    const Plane &y = a.y;
    uint32_t *partial = a.partial, *done = a.done, *stats = a.stats;
    SegData *sd = a.sd;
    int32_t *strength_out = a.strength_out;
    const SegArgs &g = a.g;
    __shared__ uint32_t s_last;
    lf_strength_body(y, partial);
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(done, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) *done = 0;
    auto_segments_body(partial, (int)gridDim.x, stats, sd, strength_out, g);
}
__global__ __launch_bounds__(256) void k_strength_segments(StrengthItem a) { strength_segments_body(a); }
__global__ __launch_bounds__(256) void k_strength_segments_b(BatchOf<StrengthItem> b) { strength_segments_body(b.item[blockIdx.z]); }

}  // namespace

size_t rc_partial_words() { return 2 * MAX_PARTIALS + 4; }   // + the completion counter of k_strength_segments (zero at rest)

void launch_auto_segments(hipStream_t s, const Frame &cur, uint32_t *partial, uint32_t *stats, SegData *sd, int32_t *strength_out,
                          int is_key, const int32_t refqi[4], int qi_min) {
    static const bool split = [] { const char *v = getenv("VP8HIP_SPLIT_SEGMENTS"); return v && v[0] && v[0] != '0'; }();   // A/B
    const Plane &y = cur.Y[0];
    const int nb = (y.h + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    const SegArgs g{y.w * y.h, (y.h - 1) * (y.w - 1), is_key, refqi[0], refqi[1], refqi[2], refqi[3], qi_min};
    if (!split) {
        hipLaunchKernelGGL(k_strength_segments, dim3(nb), dim3(256), 0, s, StrengthItem{y, partial, partial + 2 * MAX_PARTIALS, stats, sd, strength_out, g});
    } else {
        hipLaunchKernelGGL(k_lf_strength, dim3(nb), dim3(256), 0, s, y, partial);
        hipLaunchKernelGGL(k_auto_segments, dim3(1), dim3(256), 0, s, partial, nb, stats, sd, strength_out, g);
    }
}

void launch_auto_segments_batch(hipStream_t s, const Frame *const *cur, uint32_t *const *partial, uint32_t *const *stats, SegData *const *sd,
                                int32_t *const *strength_out, const int *is_key, const int32_t (*refqi)[4], int qi_min, int n) {
    BatchOf<StrengthItem> b;
    b.n = n;
    const Plane &y0 = cur[0]->Y[0];
    const int nb = (y0.h + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    for (int i = 0; i < n; ++i) {
        const Plane &y = cur[i]->Y[0];
        const SegArgs g{y.w * y.h, (y.h - 1) * (y.w - 1), is_key[i], refqi[i][0], refqi[i][1], refqi[i][2], refqi[i][3], qi_min};
        b.item[i] = StrengthItem{y, partial[i], partial[i] + 2 * MAX_PARTIALS, stats[i], sd[i], strength_out[i], g};
    }
    static const bool skip = experiment_skip("scan");
    if (skip) return;   // timing experiment only
    hipLaunchKernelGGL(k_strength_segments_b, dim3(nb, 1, n), dim3(256), 0, s, b);
}

void launch_lf_strength(hipStream_t s, const Frame &cur, uint32_t *partial, uint32_t *stats) {
    const Plane &y = cur.Y[0];
    const int nb = (y.h + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipLaunchKernelGGL(k_lf_strength, dim3(nb), dim3(256), 0, s, y, partial);
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, s, partial, nb, stats, 0);
}

void launch_chroma_sad(hipStream_t s, const Frame &cur, const Frame &prev, uint32_t *partial, uint32_t *stats) {
    const int nb = (cur.U.h + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipLaunchKernelGGL(k_chroma_sad, dim3(nb, 2), dim3(256), 0, s, cur.U, cur.V, prev.U, prev.V, partial);
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, s, partial, nb, stats, 2);
}

}  // namespace vp8
