// vp8hip_dev.h -- shared declarations of the gfx950 inter-frame kernels (internal, not the ABI).
//
// Surfaces: every plane lives in HBM with a PAD-pixel margin on all four sides and a row
// stride that is a multiple of 64 B, so pixel (0,0) is 32-byte aligned, rows of an 8x8 block
// are one aligned 8-byte load, and a search window that hangs over the frame edge is an
// ordinary load (the margin of a reference plane holds the replicated edge = the reference's
// CLK_ADDRESS_CLAMP_TO_EDGE sampler, src/GPU_kernels.cl:562).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vp8 {

constexpr int PAD = 32;        // allocated margin (pixels) around every plane
constexpr int EXT = 8;         // replicated-edge width actually filled (max reach of any filter: 3)
constexpr int SD_INTS = 11;    // ints per segment_data, src/vp8enc.h:80-92
enum { SD_Y_AC_I = 0, SD_Y_DC_IDELTA, SD_Y2_DC_IDELTA, SD_Y2_AC_IDELTA, SD_UV_DC_IDELTA, SD_UV_AC_IDELTA,
       SD_LOOP_FILTER_LEVEL, SD_MBEDGE_LIMIT, SD_SUB_BEDGE_LIMIT, SD_INTERIOR_LIMIT, SD_HEV_THRESHOLD };

struct Plane {
    uint8_t *p;   // address of pixel (0,0)
    int stride;   // bytes per row
    int w, h;
};

struct Frame {
    Plane Y[5];   // Y[l] = luma downsampled by 2^l
    Plane U, V;
};

struct RefSet {            // what one search/predict launch needs to know about the references
    Frame ref[3];          // LAST, GOLDEN, ALTREF
    int use[3];
};

// per-reference motion state: two short2 nets (ping-pong, src/init.h:672-854) and the block costs
struct NetSet {
    int16_t *net[3][2];
    int32_t *bdiff[3];
};

struct MBOut {
    int32_t *parts, *ref, *seg, *nz, *mask;
    int16_t *vec;      // [MBs][4][2]
    int16_t *coeffs;   // [MBs][25][16]
    float *ssim;
    int32_t *first_lf0; // smallest MB index whose segment has loop_filter_level == 0 (INT_MAX if none)
};

struct SegData { int32_t v[4 * SD_INTS]; };

// ---- launchers (kernels_*.hip) ---------------------------------------------------------------
void launch_border(hipStream_t s, const Frame &f);
void launch_downsample(hipStream_t s, const Plane *src, const Plane *dst, int nsurf);
void launch_reset_nets(hipStream_t s, const NetSet &n, int b8);
void launch_pyramid(hipStream_t s, const Frame *a, const Frame *b);   // all four levels of one or two frames
void launch_pack(hipStream_t s, const Frame &f, const void *y, const void *u, const void *v);
void launch_search1(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, int level,
                    int src_idx, int net_width);
void launch_search2(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, uint32_t *dbg = nullptr,
                    int dbg_block = -1);
void launch_select(hipStream_t s, const NetSet &nets, const MBOut &o, int mbw, int mbh, int use_golden,
                   int use_altref);
void launch_mb(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, const Frame &recon,
               const MBOut &o, const SegData *d_sd, float ssim_target, int mbw, int mbh);
void launch_filter_mask(hipStream_t s, const MBOut &o, const SegData *d_sd, int mbs);
void launch_loop_filter(hipStream_t s, const Frame &recon, const MBOut &o, const SegData *d_sd, int32_t *progress,
                        int mbw, int mbh);   // first version: one wave per MB row, hand-off through HBM (kept for A/B)
void launch_loop_filter2(hipStream_t s, const Frame &recon, const MBOut &o, const SegData *d_sd, int32_t *progress,
                         int mbw, int mbh, unsigned launch_no);  // banded wavefront in LDS (the one the library uses)

// ---- device helpers ---------------------------------------------------------------------------
#if defined(__HIPCC__)
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int byte_of(uint32_t w, int k) { return (int)((w >> (8 * k)) & 0xffu); }

__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) {  // byte-aligned dword load
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint2 ld_u64(const uint8_t *p) {     // byte-aligned 8-byte load
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// The block-match metric (src/GPU_kernels.cl:85-190, weight_opt): sum of |coefficients| of a 4x4
// forward transform of the difference block, DC/4.  Column pass keeps the reference's quirk (its b1
// is overwritten; rows 1 and 3 use the raw r2).  d = 4 rows x 4 columns, row-major.
__device__ __forceinline__ int weight4x4(const int d[16]) {
    int R[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int r0 = d[c], r1 = d[4 + c], r2 = d[8 + c], r3 = d[12 + c];
        const int a = (r0 + r3) * 8;
        const int dd = (r0 - r3) * 8;
        const int cc = (r1 - r2) * 8;
        R[c] = a + cc;
        R[8 + c] = a - cc;
        R[4 + c] = (r2 * 2217 + dd * 5352 + 14500) >> 12;
        R[12 + c] = (dd * 2217 - r2 * 5352 + 7500) >> 12;
    }
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e0 = R[4 * i], e1 = R[4 * i + 1], e2 = R[4 * i + 2], e3 = R[4 * i + 3];
        const int a1 = e0 + e3, d1 = e0 - e3, b1 = e1 + e2, c1 = e1 - e2;
        const int o0 = (a1 + b1 + 7) >> 4;
        const int o2 = (a1 - b1 + 7) >> 4;
        const int o1 = ((c1 * 2217 + d1 * 5352 + 12000) >> 16) + (d1 != 0);
        const int o3 = (d1 * 2217 - c1 * 5352 + 51000) >> 16;
        sum += (i == 0 ? (iabs(o0) >> 2) : iabs(o0)) + iabs(o1) + iabs(o2) + iabs(o3);
    }
    return sum;
}

// weight of the 4x4 block whose rows are the byte quads c[r] (current) and p[r] (candidate)
__device__ __forceinline__ int weight_quads(const uint32_t c[4], const uint32_t p[4]) {
    int d[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) d[4 * r + k] = byte_of(c[r], k) - byte_of(p[r], k);
    return weight4x4(d);
}

// VP8 six-tap filters by 1/8-pel phase, src/GPU_kernels.cl:563-572
static __device__ __constant__ const int8_t k_sixtap[8][8] = {
    {0, 0, 127, 0, 0, 0, 0, 0}, /* phase 0 is the identity: handled explicitly (tap 128 does not fit) */
    {0, -6, 123, 12, -1, 0, 0, 0},   {2, -11, 108, 36, -8, 1, 0, 0}, {0, -9, 93, 50, -6, 0, 0, 0},
    {3, -16, 77, 77, -16, 3, 0, 0},  {0, -6, 50, 93, -9, 0, 0, 0},   {1, -8, 36, 108, -11, 2, 0, 0},
    {0, -1, 12, 123, -6, 0, 0, 0},
};
__device__ __forceinline__ void load_taps(int phase, int f[6]) {
#pragma unroll
    for (int t = 0; t < 6; ++t) f[t] = k_sixtap[phase][t];
    if (phase == 0) f[2] = 128;
}
// sat8(s >> 7), written as clamp-then-shift.  The shift-then-saturate form is pattern-matched by
// hipcc (ROCm 7.2) into v_ashr_pk_u8_i32, whose result the compiler then ORs with further bytes as
// if bits 31:16 were zero; on gfx950 they keep the previous contents of the destination VGPR
// (0xffff after a negative sum), which corrupted two neighbouring pixels.  Found by the parity tests.
__device__ __forceinline__ int sat8_shr7(int s) { return iclamp(s, 0, 32767) >> 7; }
// (sum + 64)/128 with C truncation toward zero
__device__ __forceinline__ int div128(int s) { return s >= 0 ? (s >> 7) : -((-s) >> 7); }
#endif

}  // namespace vp8
