// kernels_lf.hip -- VP8 normal loop filter of a whole frame (Y, U, V) in one launch for gfx950.
//
// Parity target: loop_filter_frame_luma / _chroma, CPU_kernels.cl:970-1075, :1333-1439, with the
// edge filters filter_mb_edge8 / filter_b_edge8 (:829-926).  Those run the macroblocks of a plane
// in raster order on one CPU thread; the order only matters through these dependencies:
//   MB(x,y) needs MB(x-1,y) finished and MB(x+1,y-1) finished (its left-edge filter rewrites the
//   right three columns of MB(x,y-1), which MB(x,y)'s top-edge filter reads).
// So MB rows run as a wavefront with a two-macroblock lag.  One wave per macroblock row handles the
// luma MB (lanes 0-15) and the two chroma MBs (lanes 16-23, 24-31) of each position; a per-row
// progress counter in HBM carries the dependency.  Pixels cross between rows through HBM with
// sc1 (agent-scope, L1-bypassing, write-through) loads and stores, so no cache fence is needed per
// step (MI355X_MICROARCH.md, "Valid forms").  Inside the wave a macroblock is filtered in an LDS tile.
//
// Quirks kept: (a) the reference hands the q registers of one edge to the next edge as p registers
// WITHOUT the saturation its stores apply (:1024, :1062), so out-of-range values survive inside a
// macroblock row/column; (b) loop_filter_level == 0 makes the reference leave the whole plane
// (:990): every macroblock from the first such one on (raster order) stays unfiltered.
#include "vp8hip_dev.h"

namespace vp8 {

__device__ __forceinline__ int c128(int v) { return iclamp(v, -128, 127); }

struct EdgeRegs { int p3, p2, p1, p0, q0, q1, q2, q3; };

__device__ __forceinline__ bool lf_mask(const EdgeRegs &e, int int_lim, int edge_lim) {
    const bool over = (iabs(e.p3 - e.p2) > int_lim) | (iabs(e.p2 - e.p1) > int_lim) | (iabs(e.p1 - e.p0) > int_lim) |
                      (iabs(e.q1 - e.q0) > int_lim) | (iabs(e.q2 - e.q1) > int_lim) | (iabs(e.q3 - e.q2) > int_lim) |
                      ((iabs(e.p0 - e.q0) * 2 + iabs(e.p1 - e.q1) / 2) > edge_lim);
    return !over;
}

// filter_mb_edge8, CPU_kernels.cl:829-883
__device__ __forceinline__ void filter_mb_edge(EdgeRegs &e, int mb_lim, int int_lim, int hev_thr) {
    const bool mask = lf_mask(e, int_lim, mb_lim);
    const bool hev = (iabs(e.p1 - e.p0) > hev_thr) | (iabs(e.q1 - e.q0) > hev_thr);
    int w = c128(e.p1 - e.q1);
    w = c128(w + (e.q0 - e.p0) * 3);
    w = mask ? w : 0;
    int a = hev ? w : 0;
    const int b = c128(a + 3) >> 3;
    a = c128(a + 4) >> 3;
    e.q0 -= a;
    e.p0 += b;
    w = hev ? 0 : w;
    a = c128((w * 27 + 63) >> 7);
    e.q0 -= a;
    e.p0 += a;
    a = c128((w * 18 + 63) >> 7);
    e.q1 -= a;
    e.p1 += a;
    a = c128((w * 9 + 63) >> 7);
    e.q2 -= a;
    e.p2 += a;
}

// filter_b_edge8, CPU_kernels.cl:885-926
__device__ __forceinline__ void filter_b_edge(EdgeRegs &e, int b_lim, int int_lim, int hev_thr) {
    const bool mask = lf_mask(e, int_lim, b_lim);
    const bool hev = (iabs(e.p1 - e.p0) > hev_thr) | (iabs(e.q1 - e.q0) > hev_thr);
    int a = c128(e.p1 - e.q1);
    a = hev ? a : 0;
    a = c128(a + (e.q0 - e.p0) * 3);
    a = mask ? a : 0;
    const int b = c128(a + 3) >> 3;
    a = c128(a + 4) >> 3;
    e.q0 -= a;
    e.p0 += b;
    a = (a + 1) >> 1;
    a = hev ? 0 : a;
    e.q1 -= a;
    e.p1 += a;
}

__device__ __forceinline__ uint8_t px(int u) { return (uint8_t)sat8(u + 128); }

// One line of samples across the four edges of a macroblock: t[0..3] = the four samples before the
// MB edge, t[4..] = the macroblock's own.  `stride` in bytes between consecutive samples.
__device__ __forceinline__ void filter_line(uint8_t *t, int stride, int msz, bool has_mb_edge, bool inner,
                                            int mb_lim, int b_lim, int int_lim, int hev_thr) {
    EdgeRegs e;
    e.q0 = (int)t[4 * stride] - 128;
    e.q1 = (int)t[5 * stride] - 128;
    e.q2 = (int)t[6 * stride] - 128;
    e.q3 = (int)t[7 * stride] - 128;
    if (has_mb_edge) {
        e.p3 = (int)t[0] - 128;
        e.p2 = (int)t[1 * stride] - 128;
        e.p1 = (int)t[2 * stride] - 128;
        e.p0 = (int)t[3 * stride] - 128;
        filter_mb_edge(e, mb_lim, int_lim, hev_thr);
        t[1 * stride] = px(e.p2);
        t[2 * stride] = px(e.p1);
        t[3 * stride] = px(e.p0);
        t[4 * stride] = px(e.q0);
        t[5 * stride] = px(e.q1);
        t[6 * stride] = px(e.q2);
    }
    if (inner) {
        for (int k = 4; k < msz; k += 4) {
            e.p3 = e.q0; e.p2 = e.q1; e.p1 = e.q2; e.p0 = e.q3;  // registers carried, not re-read
            uint8_t *u = t + (4 + k) * stride;
            e.q0 = (int)u[0] - 128;
            e.q1 = (int)u[stride] - 128;
            e.q2 = (int)u[2 * stride] - 128;
            e.q3 = (int)u[3 * stride] - 128;
            filter_b_edge(e, b_lim, int_lim, hev_thr);
            u[-2 * stride] = px(e.p1);
            u[-1 * stride] = px(e.p0);
            u[0] = px(e.q0);
            u[stride] = px(e.q1);
        }
    }
}

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct LFArgs {
    Plane Y, U, V;
    MBOut o;
    const SegData *sd;
    int32_t *progress;  // [mbh] macroblocks finished per row; zeroed before the launch
    int mbw, mbh;
};

// tile geometry: luma 20 rows x 24 B (20 used), chroma 12 rows x 12 B
constexpr int TY_STRIDE = 24, TC_STRIDE = 12;
constexpr int TY_BYTES = 20 * TY_STRIDE, TC_BYTES = 12 * TC_STRIDE;

__global__ __launch_bounds__(64) void k_loop_filter(LFArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[TY_BYTES + 2 * TC_BYTES];
    const int row = blockIdx.x;
    const int lane = threadIdx.x;
    const int first_lf0 = *a.o.first_lf0;
    const int32_t *SD = a.sd->v;
    // this lane's filtering role
    const int pl = lane < 16 ? 0 : (lane < 24 ? 1 : (lane < 32 ? 2 : 3));
    const int li = pl == 0 ? lane : (pl == 1 ? lane - 16 : lane - 24);
    const int msz = pl == 0 ? 16 : 8;
    const int tstride = pl == 0 ? TY_STRIDE : TC_STRIDE;
    uint8_t *tile = s_tile + (pl == 0 ? 0 : (pl == 1 ? TY_BYTES : TY_BYTES + TC_BYTES));

    for (int x = 0; x < a.mbw; ++x) {
        if (row > 0) {  // wait for MB(x+1, row-1)
            const int need = imin(x + 2, a.mbw);
            while (__hip_atomic_load(&a.progress[row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need)
                __builtin_amdgcn_s_sleep(1);
        }
        const int mb = row * a.mbw + x;
        if (mb < first_lf0) {
            // ---- HBM -> LDS: luma 20x20 at (16x-4,16row-4), chroma 12x12 at (8x-4,8row-4) ----------
            for (int i = lane; i < 100 + 36 + 36; i += 64) {
                if (i < 100) {
                    const int r = i / 5, j = i % 5;
                    const uint32_t *g = reinterpret_cast<const uint32_t *>(a.Y.p + (ptrdiff_t)(16 * row - 4 + r) * a.Y.stride + 16 * x - 4) + j;
                    *reinterpret_cast<uint32_t *>(s_tile + r * TY_STRIDE + 4 * j) = ld_sc1(g);
                } else {
                    const int c = (i - 100) / 36, k = (i - 100) % 36;
                    const int r = k / 3, j = k % 3;
                    const Plane &cp = c == 0 ? a.U : a.V;
                    const uint32_t *g = reinterpret_cast<const uint32_t *>(cp.p + (ptrdiff_t)(8 * row - 4 + r) * cp.stride + 8 * x - 4) + j;
                    *reinterpret_cast<uint32_t *>(s_tile + TY_BYTES + c * TC_BYTES + r * TC_STRIDE + 4 * j) = ld_sc1(g);
                }
            }
            __syncthreads();
            const int seg = a.o.seg[mb];
            const int32_t *sd = SD + seg * SD_INTS;
            const int int_lim = (int16_t)sd[SD_INTERIOR_LIMIT], mb_lim = (int16_t)sd[SD_MBEDGE_LIMIT];
            const int b_lim = (int16_t)sd[SD_SUB_BEDGE_LIMIT], hev_thr = (int16_t)sd[SD_HEV_THRESHOLD];
            const bool inner = a.o.mask[mb] != 0;
            // vertical edges: one lane per pixel row, samples along x
            if (pl < 3) filter_line(tile + (4 + li) * tstride, 1, msz, x > 0, inner, mb_lim, b_lim, int_lim, hev_thr);
            __syncthreads();
            // horizontal edges: one lane per pixel column, samples along y
            if (pl < 3) filter_line(tile + 4 + li, tstride, msz, row > 0, inner, mb_lim, b_lim, int_lim, hev_thr);
            __syncthreads();
            // ---- LDS -> HBM (whole tile: nobody else touches these pixels while we hold them) -------
            for (int i = lane; i < 100 + 36 + 36; i += 64) {
                if (i < 100) {
                    const int r = i / 5, j = i % 5;
                    uint32_t *g = reinterpret_cast<uint32_t *>(a.Y.p + (ptrdiff_t)(16 * row - 4 + r) * a.Y.stride + 16 * x - 4) + j;
                    st_sc1(g, *reinterpret_cast<const uint32_t *>(s_tile + r * TY_STRIDE + 4 * j));
                } else {
                    const int c = (i - 100) / 36, k = (i - 100) % 36;
                    const int r = k / 3, j = k % 3;
                    const Plane &cp = c == 0 ? a.U : a.V;
                    uint32_t *g = reinterpret_cast<uint32_t *>(cp.p + (ptrdiff_t)(8 * row - 4 + r) * cp.stride + 8 * x - 4) + j;
                    st_sc1(g, *reinterpret_cast<const uint32_t *>(s_tile + TY_BYTES + c * TC_BYTES + r * TC_STRIDE + 4 * j));
                }
            }
        }
        // publish: stores drained, then the counter (same wave, program order)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (lane == 0) __hip_atomic_store(&a.progress[row], x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// first macroblock (raster order) whose segment has loop_filter_level == 0, CPU_kernels.cl:990
__global__ __launch_bounds__(256) void k_first_lf0(MBOut o, const SegData *sd, int mbs) {
    const int mb = blockIdx.x * 256 + threadIdx.x;
    if (mb < mbs && sd->v[o.seg[mb] * SD_INTS + SD_LOOP_FILTER_LEVEL] == 0) atomicMin(o.first_lf0, mb);
}

void launch_loop_filter(hipStream_t s, const Frame &recon, const MBOut &o, const SegData *d_sd, int32_t *progress,
                        int mbw, int mbh) {
    LFArgs a;
    a.Y = recon.Y[0];
    a.U = recon.U;
    a.V = recon.V;
    a.o = o;
    a.sd = d_sd;
    a.progress = progress;
    a.mbw = mbw;
    a.mbh = mbh;
    hipMemsetAsync(progress, 0, sizeof(int32_t) * mbh, s);
    hipMemsetAsync(o.first_lf0, 0x7f, 4, s);  // 0x7f7f7f7f: "none"
    hipLaunchKernelGGL(k_first_lf0, dim3((mbw * mbh + 255) / 256), dim3(256), 0, s, o, d_sd, mbw * mbh);
    hipLaunchKernelGGL(k_loop_filter, dim3(mbh), dim3(64), 0, s, a);
}

}  // namespace vp8
