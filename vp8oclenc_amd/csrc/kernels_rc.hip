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
#include "kernels_rc_dev.h"

namespace vp8 {

namespace {

using namespace rc;

__global__ __launch_bounds__(256) void k_lf_strength(Plane y, uint32_t *partial) { lf_strength_body(y, partial, blockIdx.x); }

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

__global__ __launch_bounds__(256) void k_fold(const uint32_t *partial, int nblocks, uint32_t *stats, int o) {
    __shared__ uint32_t s_red[4];
    uint32_t e, d;
    fold(partial, nblocks, s_red, e, d);
    if (threadIdx.x == 0) { stats[o] = e; stats[o + 1] = d; }
}

__global__ __launch_bounds__(256) void k_auto_segments(const uint32_t *partial, int nblocks, uint32_t *stats, SegData *sd,
                                                       int32_t *strength_out, SegArgs g) {
    auto_segments_body(partial, nblocks, stats, sd, strength_out, g);
}
__global__ __launch_bounds__(256) void k_strength_segments(StrengthItem a) { strength_segments_body(a.y, a.c, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void k_strength_segments_b(BatchOf<StrengthItem> b) { strength_segments_body(b.item[blockIdx.z].y, b.item[blockIdx.z].c, blockIdx.x, gridDim.x); }

}  // namespace

size_t rc_partial_words() { return 2 * MAX_PARTIALS + 4; }   // + the completion counter of k_strength_segments (zero at rest)

void launch_auto_segments(hipStream_t s, const Frame &cur, uint32_t *partial, uint32_t *stats, SegData *sd, int32_t *strength_out,
                          int is_key, const int32_t refqi[4], int qi_min) {
    static const bool split = [] { const char *v = getenv("VP8HIP_SPLIT_SEGMENTS"); return v && v[0] && v[0] != '0'; }();   // A/B
    const Plane &y = cur.Y[0];
    const int nb = (y.h + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    const SegArgs g{y.w * y.h, (y.h - 1) * (y.w - 1), is_key, refqi[0], refqi[1], refqi[2], refqi[3], qi_min};
    if (!split) {
        hipLaunchKernelGGL(k_strength_segments, dim3(nb), dim3(256), 0, s, StrengthItem{y, ScanCore{partial, partial + 2 * MAX_PARTIALS, stats, sd, strength_out, g}});
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
        b.item[i] = StrengthItem{y, ScanCore{partial[i], partial[i] + 2 * MAX_PARTIALS, stats[i], sd[i], strength_out[i], g}};
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
