// kernels_lf4.hip -- VP8 normal loop filter: banded wavefront over a band-tall pixel plane in LDS, the dependency chain on
// worker waves that do nothing but filter, everything else on helper waves (gfx950).
//
// Same arithmetic and ordering semantics as loop_filter_frame_luma/_chroma (CPU_kernels.cl:970-1075, :1333-1439; edge
// filters :829-926).  The filter is a chain of dependent edge filters: MB(x,y) needs MB(x-1,y) complete and, for its
// horizontal edges only, the vertical MB edge of MB(x+1,y-1) -- eight edge filters per macroblock in sequence, mb_w + mb_h
// macroblock steps per frame.  A lone wave issues one instruction of any kind per 4-5.5 cycles, so the frame time is
// (steps) x (instructions the chain's wave issues per step).  The previous form (kernels_lf3.hip in the history: byte tiles
// and strips in LDS, the worker itself prefetching, unpacking, packing, transposing through bytes, draining to HBM and
// keeping strip rings) issued about 620 per step, 300 of them the filters.  Here:
//
//   * A workgroup owns a band of ROWS macroblock rows and keeps it as ONE pixel plane in LDS, a dword per sample, already
//     carrying the +256 bias the filters work with and saturated the way the reference's stores saturate: 4 + 16*ROWS luma
//     rows (the four rows above the band first) by a ring of RING macroblocks, and the same for U|V side by side.  Both
//     phases of a macroblock step address that plane directly:
//       P1 = vertical edges, lane = pixel row:   five ds_read_b128, filter, five ds_write_b128 (in place);
//       P2 = horizontal edges, lane = column:    twenty ds_read_b32 at immediate row offsets, filter, seventeen ds_write_b32.
//     No tile, no strips, no packing or unpacking, no copies between rows: the row below reads the bottom rows of the row
//     above where they lie.
//   * P2 of MB(x,y) needs only P1 of MB(x+1,y-1), so row y runs ONE macroblock behind row y-1 (x = S - r at step S) with
//     the hand-off in the middle of the step; a wave runs two rows (32 lanes each: 0-15 luma, 16-23 U, 24-31 V).
//   * Helper waves: a PORTER per row pair loads the macroblocks a few steps ahead of its worker, expands them into the
//     plane and leaves a parameter record per macroblock (the limits of its segment, which edges apply), and packs
//     macroblocks that have become final back to bytes and stores them; the PUBLISHER hands the band's bottom rows to the
//     next band as tagged 8-byte granules in a hand-off buffer in HBM (sc1), whose LOADER polls the granules themselves.
//     The workers never touch global memory and never wait on vmcnt.
//   * Branch-free edge filters (an edge that does not apply has interior limit -1), samples carry +256 so |a-b| is one
//     v_sad_u16 even when an unsaturated carry (reference quirk, :1024/:1062) dips below zero.
//   * One LDS poll per step (middle of the step) covers every dependency of a worker.
// Every device-side wait is bounded (VP8HIP_ERR_TIMEOUT).
// History (1080p, one frame): v1 one wave per row through HBM 1.9 ms; v2 banded 0.71 ms; v3 (byte tiles, worker does
// everything) 0.31 ms; this file: see DESIGN.md section 4.
#include <stdlib.h>
#include <string.h>

#include "vp8hip_dev.h"

namespace vp8 {

namespace lf4 {

constexpr int WORKERS = 4;             // worker waves per band (one per SIMD)
constexpr int ROWS = 2 * WORKERS;      // MB rows per band
constexpr int RING = 8;                // ring length of the plane in macroblocks (a power of two)
constexpr int BIAS = 256;
// the plane: one stride for luma and chroma rows, so that P2's row offsets are the same immediates in every lane
constexpr int LS = RING * 16 * 4 + 16;         // 528 B: 128 luma samples (or 64 U | 4 pad | 64 V) + padding against bank conflicts
constexpr int VOFF = RING * 8 * 4 + 16;        // V's samples inside a chroma row
constexpr int Y_ROWS = 4 + 16 * ROWS, C_ROWS = 4 + 8 * ROWS;
constexpr int Y_BYTES = Y_ROWS * LS, C_BYTES = C_ROWS * LS;
constexpr int PAR_BYTES = 32;                  // parameter record of a macroblock
constexpr int BIG = 0x3fffffff;

// waves: WORKERS workers, then one porter per worker (wave WORKERS + i lands on the SIMD of worker i), loader, publisher
enum { W_PORTER = WORKERS, W_LOADER = 2 * WORKERS, W_PUBLISHER, NWAVES };
enum { F_FEED = WORKERS, F_TOP = 2 * WORKERS, F_TOPDRAIN, F_PUB, F_ABORT, NFLAGS };   // flag[0..WORKERS-1] = 2*step + phase of each worker; flag[F_FEED + i] = steps porter i has fed

__device__ __forceinline__ int ad(int a, int b) { return (int)__builtin_amdgcn_sad_u16((uint32_t)a, (uint32_t)b, 0u); }
__device__ __forceinline__ int c128(int v) { return iclamp(v, -128, 127); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return imax(imax(a, b), c); }
struct EdgeRegs { int p3, p2, p1, p0, q0, q1, q2, q3; };
struct Limits { int mb_delta, b_delta, hev_thr; };   // *_delta = interior limit - 2*edge limit - 1, see edge_masks

// 2|p0-q0| + (|p1-q1| >> 1) <= E  <=>  |p1-q1| + 4|p0-q0| <= 2E + 1  <=>  that sum + (I - 2E - 1) <= I, so the
// edge test joins the six interior tests (each |a-b| <= I) in one max3 and one compare.  edge_delta = I-2E-1.
// I == -1 switches the edge off: the interior differences are >= 0, so the mask can never be true.
__device__ __forceinline__ void edge_masks(const EdgeRegs &e, int int_lim, int edge_delta, int hev_thr, bool &mask,
                                           bool &hev) {
    const int d10 = ad(e.p1, e.p0), dq10 = ad(e.q1, e.q0);
    const int m1 = max3i(ad(e.p3, e.p2), ad(e.p2, e.p1), d10);
    const int m2 = max3i(dq10, ad(e.q2, e.q1), ad(e.q3, e.q2));
    const int edge = (int)__builtin_amdgcn_sad_u16((uint32_t)e.p1, (uint32_t)e.q1, (uint32_t)((ad(e.p0, e.q0) << 2) + edge_delta));
    mask = max3i(m1, m2, edge) <= int_lim;
    hev = imax(d10, dq10) > hev_thr;
}
__device__ __forceinline__ void filter_mb_edge(EdgeRegs &e, const Limits &L, int int_lim) {  // :829-883
    bool mask, hev;
    edge_masks(e, int_lim, L.mb_delta, L.hev_thr, mask, hev);
    int w = c128(e.p1 - e.q1);
    w = c128(w + __mul24(e.q0 - e.p0, 3));   // (|q0 - p0| < 2^11: v_mad_i32_i24, not the 64-bit multiply-add hipcc picks for `* 3`)
    w = mask ? w : 0;
    int a = imin(hev ? w : 0, 123);   // min(a + 4, 127) >> 3 and min(a + 3, 127) >> 3 are both 15 from 123 on: one min for the two
    const int b = (a + 3) >> 3;
    a = (a + 4) >> 3;
    e.q0 -= a; e.p0 += b;
    w = hev ? 0 : w;
    a = (w * 27 + 63) >> 7; e.q0 -= a; e.p0 += a;
    a = (w * 18 + 63) >> 7; e.q1 -= a; e.p1 += a;
    a = (w * 9 + 63) >> 7;  e.q2 -= a; e.p2 += a;
}
__device__ __forceinline__ void filter_b_edge(EdgeRegs &e, const Limits &L, int int_lim) {  // :885-926
    bool mask, hev;
    edge_masks(e, int_lim, L.b_delta, L.hev_thr, mask, hev);
    int a = c128(e.p1 - e.q1);
    a = hev ? a : 0;
    a = iclamp(a + __mul24(e.q0 - e.p0, 3), -128, 123);   // the clamp to 127 and the two min(.., 127) >> 3 behind it in one (see filter_mb_edge)
    a = mask ? a : 0;
    const int b = (a + 3) >> 3;
    a = (a + 4) >> 3;
    e.q0 -= a; e.p0 += b;
    a = (a + 1) >> 1;
    a = hev ? 0 : a;
    e.q1 -= a; e.p1 += a;
}

// One line of biased samples t[0..19] (t[0..3] precede the macroblock edge) through the MB edge and the three
// inner edges, each under its own interior limit (-1 = edge switched off).  t[] receives the UNSATURATED results (the reference saturates when it
// stores); the p/q registers handed from edge to edge stay unsaturated too (:1024, :1062).  t[0], t[18], t[19] are never written.
__device__ __forceinline__ void filter_line(int (&t)[20], const Limits &L, int il_mb, int il4, int il8) {
    EdgeRegs e;
    e.p3 = t[0]; e.p2 = t[1]; e.p1 = t[2]; e.p0 = t[3];
    e.q0 = t[4]; e.q1 = t[5]; e.q2 = t[6]; e.q3 = t[7];
    filter_mb_edge(e, L, il_mb);
    t[1] = e.p2; t[2] = e.p1; t[3] = e.p0;
    t[4] = e.q0; t[5] = e.q1; t[6] = e.q2;
#pragma unroll
    for (int k = 4; k < 16; k += 4) {
        e.p3 = e.q0; e.p2 = e.q1; e.p1 = e.q2; e.p0 = e.q3;
        e.q0 = t[4 + k]; e.q1 = t[5 + k]; e.q2 = t[6 + k]; e.q3 = t[7 + k];
        filter_b_edge(e, L, k == 4 ? il4 : il8);
        t[2 + k] = e.p1; t[3 + k] = e.p0; t[4 + k] = e.q0; t[5 + k] = e.q1;
    }
}

// ---- the same line in two parts ------------------------------------------------------------------------------------------
// A lone wave issues an instruction every 5.3 cycles whether or not it depends on the one before (scripts/ubench/lone_wave.hip),
// and an LDS round trip costs it 50-130 cycles of nothing: what a step can win is the waits.  Of a line's twenty samples the four in
// front of the macroblock edge arrive LATE in both phases (P1: the columns P2 of the previous step has just written; P2: the rows
// of the macroblock row above, behind the step's poll); the sixteen of the macroblock itself are there early.  And an inner edge
// depends on the edge before it only through |p3 - p2| and |p2 - p1| (the edge at 4 also through p1, which the macroblock edge
// moves): its hev, its filter value before the mask and six of its eight interior differences are functions of samples nobody
// has touched.  line_pre computes all of that from t[4..19] while the late samples are in flight; line_post is the chain.
// Same operations on the same values as filter_line (max3 regrouped: max is associative), so the results are bit for bit the same.
struct BPre { int pre_max, a, hv; };            // an inner edge whose p1, p0 and q side are as loaded (the edges at 8 and 12); hv = max(|p1-p0|, |q1-q0|)
struct LinePre {
    int mb_dq10, mb_m2;                          // macroblock edge: |q1-q0|, max3(|q1-q0|, |q2-q1|, |q3-q2|)
    int e4_dq10, e4_m2, e4_e0, e4_qp3;           // edge at 4: the same two, (|p0-q0| << 2) + b_delta, 3 (q0 - p0)
    BPre e8, e12;
};
__device__ __forceinline__ BPre b_edge_pre(int p1, int p0, int q0, int q1, int q2, int q3, const Limits &L) {
    const int d10 = ad(p1, p0), dq10 = ad(q1, q0);
    const int m2 = max3i(dq10, ad(q2, q1), ad(q3, q2));
    const int edge = (int)__builtin_amdgcn_sad_u16((uint32_t)p1, (uint32_t)q1, (uint32_t)((ad(p0, q0) << 2) + L.b_delta));
    BPre r;
    r.pre_max = max3i(d10, m2, edge);
    r.hv = imax(d10, dq10);
    int a = c128(p1 - q1);
    a = r.hv > L.hev_thr ? a : 0;
    r.a = iclamp(a + __mul24(q0 - p0, 3), -128, 123);
    return r;
}
__device__ __forceinline__ void b_edge_post(EdgeRegs &e, const BPre &pre, int int_lim, int hev_thr) {
    const bool mask = max3i(ad(e.p3, e.p2), ad(e.p2, e.p1), pre.pre_max) <= int_lim;
    int a = mask ? pre.a : 0;
    const int b = (a + 3) >> 3;
    a = (a + 4) >> 3;
    e.q0 -= a; e.p0 += b;
    a = (a + 1) >> 1;
    a = pre.hv > hev_thr ? 0 : a;
    e.q1 -= a; e.p1 += a;
}
// the value exists HERE: the compiler may neither compute it later (it sinks what only a later block uses) nor move it across
#define LF_KEEP(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ void line_pre(const int (&t)[20], const Limits &L, LinePre &pre) {
    pre.mb_dq10 = ad(t[5], t[4]);
    pre.mb_m2 = max3i(pre.mb_dq10, ad(t[6], t[5]), ad(t[7], t[6]));
    pre.e4_dq10 = ad(t[9], t[8]);
    pre.e4_m2 = max3i(pre.e4_dq10, ad(t[10], t[9]), ad(t[11], t[10]));
    pre.e4_e0 = (ad(t[7], t[8]) << 2) + L.b_delta;
    pre.e4_qp3 = __mul24(t[8] - t[7], 3);
    pre.e8 = b_edge_pre(t[10], t[11], t[12], t[13], t[14], t[15], L);
    pre.e12 = b_edge_pre(t[14], t[15], t[16], t[17], t[18], t[19], L);
    LF_KEEP(pre.mb_dq10); LF_KEEP(pre.mb_m2); LF_KEEP(pre.e4_dq10); LF_KEEP(pre.e4_m2); LF_KEEP(pre.e4_e0); LF_KEEP(pre.e4_qp3);
    LF_KEEP(pre.e8.pre_max); LF_KEEP(pre.e8.a); LF_KEEP(pre.e8.hv); LF_KEEP(pre.e12.pre_max); LF_KEEP(pre.e12.a); LF_KEEP(pre.e12.hv);
}
// t[0..3] have arrived.  Writes t[1..17] (unsaturated, like filter_line).
__device__ __forceinline__ void line_post(int (&t)[20], const Limits &L, int il_mb, int il4, int il8, const LinePre &pre) {
    EdgeRegs e;
    e.p3 = t[0]; e.p2 = t[1]; e.p1 = t[2]; e.p0 = t[3];
    e.q0 = t[4]; e.q1 = t[5]; e.q2 = t[6]; e.q3 = t[7];
    {   // the macroblock edge (filter_mb_edge with its q-side differences taken from pre)
        const int d10 = ad(e.p1, e.p0);
        const int m1 = max3i(ad(e.p3, e.p2), ad(e.p2, e.p1), d10);
        const int edge = (int)__builtin_amdgcn_sad_u16((uint32_t)e.p1, (uint32_t)e.q1, (uint32_t)((ad(e.p0, e.q0) << 2) + L.mb_delta));
        const bool mask = max3i(m1, pre.mb_m2, edge) <= il_mb;
        const bool hev = imax(d10, pre.mb_dq10) > L.hev_thr;
        int w = c128(e.p1 - e.q1);
        w = c128(w + __mul24(e.q0 - e.p0, 3));
        w = mask ? w : 0;
        int a = imin(hev ? w : 0, 123);
        const int b = (a + 3) >> 3;
        a = (a + 4) >> 3;
        e.q0 -= a; e.p0 += b;
        w = hev ? 0 : w;
        a = (w * 27 + 63) >> 7; e.q0 -= a; e.p0 += a;
        a = (w * 18 + 63) >> 7; e.q1 -= a; e.p1 += a;
        a = (w * 9 + 63) >> 7;  e.q2 -= a; e.p2 += a;
    }
    t[1] = e.p2; t[2] = e.p1; t[3] = e.p0;
    t[4] = e.q0; t[5] = e.q1; t[6] = e.q2;
    {   // the edge at 4: p3, p2, p1 are the macroblock edge's q0, q1, q2
        e.p3 = e.q0; e.p2 = e.q1; e.p1 = e.q2; e.p0 = e.q3;
        e.q0 = t[8]; e.q1 = t[9]; e.q2 = t[10]; e.q3 = t[11];
        const int d10 = ad(e.p1, e.p0);
        const int m1 = max3i(ad(e.p3, e.p2), ad(e.p2, e.p1), d10);
        const int edge = (int)__builtin_amdgcn_sad_u16((uint32_t)e.p1, (uint32_t)e.q1, (uint32_t)pre.e4_e0);
        const bool mask = max3i(m1, pre.e4_m2, edge) <= il4;
        const bool hev = imax(d10, pre.e4_dq10) > L.hev_thr;
        int a = c128(e.p1 - e.q1);
        a = hev ? a : 0;
        a = iclamp(a + pre.e4_qp3, -128, 123);
        a = mask ? a : 0;
        const int b = (a + 3) >> 3;
        a = (a + 4) >> 3;
        e.q0 -= a; e.p0 += b;
        a = (a + 1) >> 1;
        a = hev ? 0 : a;
        e.q1 -= a; e.p1 += a;
        t[6] = e.p1; t[7] = e.p0; t[8] = e.q0; t[9] = e.q1;
    }
    e.p3 = e.q0; e.p2 = e.q1; e.p1 = t[10]; e.p0 = t[11];
    e.q0 = t[12]; e.q1 = t[13]; e.q2 = t[14]; e.q3 = t[15];
    b_edge_post(e, pre.e8, il8, L.hev_thr);
    t[10] = e.p1; t[11] = e.p0; t[12] = e.q0; t[13] = e.q1;
    e.p3 = e.q0; e.p2 = e.q1; e.p1 = t[14]; e.p0 = t[15];
    e.q0 = t[16]; e.q1 = t[17]; e.q2 = t[18]; e.q3 = t[19];
    b_edge_post(e, pre.e12, il8, L.hev_thr);
    t[14] = e.p1; t[15] = e.p0; t[16] = e.q0; t[17] = e.q1;
}

// biased sample -> biased saturated sample; its low byte is the pixel (BIAS = 256)
__device__ __forceinline__ int satb(int v) { return iclamp(v, BIAS, BIAS + 255); }
// four plane dwords (biased saturated samples) -> their four bytes
__device__ __forceinline__ uint32_t pack4(const int4 &v) {
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)v.y, (uint32_t)v.x, 0x0c0c0400u);
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)v.w, (uint32_t)v.z, 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
// four bytes -> four plane dwords
__device__ __forceinline__ int4 unpack4(uint32_t w) {
    return make_int4(byte_of(w, 0) | BIAS, byte_of(w, 1) | BIAS, byte_of(w, 2) | BIAS, byte_of(w, 3) | BIAS);
}

// Global memory through address-space-1 pointers: the plane pointer of a lane is a per-lane choice among the three planes, made
// from integers (see the descriptors below), and a pointer the compiler cannot place is a FLAT access -- which counts on vmcnt
// AND lgkmcnt and puts an `s_waitcnt vmcnt(0) lgkmcnt(0)` behind itself.
typedef __attribute__((address_space(1))) uint8_t g_u8;
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 gld128(const g_u8 *p) { const v4u_t v = *(const __attribute__((address_space(1))) v4u_t *)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 gld64(const g_u8 *p) { const v2u_t v = *(const __attribute__((address_space(1))) v2u_t *)p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ void gst128(g_u8 *p, const uint4 &v) { *(__attribute__((address_space(1))) v4u_t *)p = v4u_t{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void gst64(g_u8 *p, const uint2 &v) { *(__attribute__((address_space(1))) v2u_t *)p = v2u_t{v.x, v.y}; }
__device__ __forceinline__ uint32_t ld_sc1(const g_u8 *p) {
    return __hip_atomic_load((const __attribute__((address_space(1))) uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// an aligned 8-byte granule {data, tag}: one store, one load -- the tag cannot arrive without its data
__device__ __forceinline__ void st64_sc1(uint2 *p, uint32_t data, uint32_t tag) {
    __hip_atomic_store((__attribute__((address_space(1))) unsigned long long *)(uintptr_t)p, (unsigned long long)data | ((unsigned long long)tag << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint2 ld64_sc1(const uint2 *p) {
    const unsigned long long v = __hip_atomic_load((const __attribute__((address_space(1))) unsigned long long *)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ void st_sc1(g_u8 *p, uint32_t v) {
    __hip_atomic_store((__attribute__((address_space(1))) uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS through address-space-3 pointers made from byte offsets: ds_read / ds_write with immediate offsets, never flat
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v4i_t lds_int4;
typedef __attribute__((address_space(3))) v2i_t lds_int2;
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) volatile int lds_flag_t;
__device__ __forceinline__ int4 ld128(uint32_t a) { const v4i_t v = *(lds_int4 *)(uintptr_t)a; return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ int2 ld64(uint32_t a) { const v2i_t v = *(lds_int2 *)(uintptr_t)a; return make_int2(v.x, v.y); }
__device__ __forceinline__ int ld32(uint32_t a) { return *(lds_int *)(uintptr_t)a; }
__device__ __forceinline__ void st128(uint32_t a, const int4 &v) { *(lds_int4 *)(uintptr_t)a = v4i_t{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void st64(uint32_t a, const int2 &v) { *(lds_int2 *)(uintptr_t)a = v2i_t{v.x, v.y}; }
__device__ __forceinline__ void st32(uint32_t a, int v) { *(lds_int *)(uintptr_t)a = v; }
// everything this wave has written to / read from LDS is done, and the compiler moves no memory access across this point
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct Args {
    Plane Y, U, V;
    MBOut o;
    SegData *sd;      // read; written only by the verdict workgroup when check_SSIM's filter update applies (chk)
    LfCheck chk;
    int32_t *gprog;   // the progress buffer (diagnostic stamps, error word, clock words)
    uint2 *handoff;   // [bands][mbw][32] {four samples, tag}: the bottom four rows of a band on their way to the next band
    int gbase;        // tags only grow: launch n uses the range (n*(mbw+2), (n+1)*(mbw+2)], so no memset
    int mbw, mbh, nbands;
    int32_t *err;     // set to 1 if a bounded wait expired (the host reports VP8HIP_ERR_TIMEOUT)
    int stall_test;   // test hook: publishers count from a wrong base, so every later band must time out
};

struct Shared {
    uint8_t Y[Y_BYTES];                           // plane row 4 + 16 r + j = pixel row j of local MB row r; rows 0-3 = the rows above the band
    uint8_t C[C_BYTES];                           // plane row 4 + 8 r + j; U at column 0, V at VOFF
    uint8_t dummy[64 * 64];                       // 64 B per lane: where stores of lanes that have nothing to store go
    uint8_t par[ROWS * RING * PAR_BYTES];         // {il P1 MB edge, il inner edges, il P2 MB edge, hev threshold, mb_delta, b_delta, -, -}
    int flag[16];                                 // worker progress, F_FEED .. F_ABORT.  Read and written through `flag` below.
    SegData sd;                                   // the segment data check_SSIM's filter update gives, when it applies (chk)
    float red[NWAVES];
    int repl;
    int first_lf0;                                // first macroblock whose segment has loop_filter_level 0 (:990)
    int4 lim[4];                                  // per segment: {interior limit, mb_delta, b_delta, hev threshold} (struct Limits)
};

// Every wait in this kernel is bounded (dispatch order and co-residency of workgroups are not architecturally
// guaranteed): a wait that is still unsatisfied after SPIN_LIMIT polls (>= 0.3 s; a frame takes < 1 ms) raises the
// workgroup's abort flag and the error word in HBM, and every wave that sees the flag leaves the kernel.  The
// frame is then invalid -- reported as VP8HIP_ERR_TIMEOUT -- but nothing hangs.
constexpr int SPIN_LIMIT = 1 << 22;
#define LF_WAIT(cond_unsatisfied, nap)                                              \
    {                                                                               \
        int spins_ = 0;                                                             \
        while ((cond_unsatisfied) && !flag[F_ABORT]) {                              \
            __builtin_amdgcn_s_sleep(nap);                                          \
            if (++spins_ > SPIN_LIMIT / (nap)) { flag[F_ABORT] = 1; *a.err = 1; }   \
        }                                                                           \
        if (flag[F_ABORT]) return;                                                  \
        asm volatile("" ::: "memory");                                              \
    }

// The workgroup behind the last band, present when check_SSIM rides in the launch: what check_SSIM reports (vp8enc.cpp:237-258:
// replaced count, the raster-order float sum / count, the minimum), the updated segment data back to where the entropy stage
// reads them, and the verdict to the host.  The sum must be the reference's -- one float accumulator over the macroblocks in
// raster order -- so the values are staged in LDS by all threads (the plane this workgroup has no other use for)
// and one thread adds them, four per ds_read_b128.
__device__ __forceinline__ void verdict_workgroup(const Args &a, Shared &sh, bool updated) {   // (inlined: a call would put the argument block into scratch memory)
    constexpr int NT = NWAVES * 64, CHUNK = 8192;
    static_assert(sizeof(sh.Y) >= CHUNK * sizeof(float), "staging area");
    float *s_val = reinterpret_cast<float *>(&sh.Y[0]);
    const int mbs = a.mbw * a.mbh, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) sh.repl = 0;
    int repl = 0;
    float mn = 2.0f, sum = 0.0f;
    for (int base = 0; base < mbs; base += CHUNK) {
        const int n = imin(CHUNK, mbs - base);
        __syncthreads();
        for (int i = threadIdx.x; i < CHUNK; i += NT) {
            float v = 0.0f;
            if (i < n) {
                v = a.o.ssim[base + i];
                repl += a.chk.is_inter[base + i] == 0;
                mn = v < mn ? v : mn;
            }
            s_val[i] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float4 *q = reinterpret_cast<const float4 *>(s_val);
            int i = 0;
            for (; i + 32 <= n; i += 32) {   // eight reads in flight, then the 32 dependent additions
                float4 v[8];
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) v[k2] = q[(i >> 2) + k2];
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) sum = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(sum, v[k2].x), v[k2].y), v[k2].z), v[k2].w);
            }
            for (; i + 4 <= n; i += 4) {
                const float4 v = q[i >> 2];
                sum = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(sum, v.x), v.y), v.z), v.w);
            }
            for (; i < n; ++i) sum = __fadd_rn(sum, s_val[i]);
        }
    }
    // with no macroblock flagged the fallback left is_inter untouched (stale): nothing was replaced
    const bool fallback_ran = __builtin_nontemporal_load(a.o.flags) != 0;
    if (fallback_ran) atomicAdd(&sh.repl, repl);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const float o = __shfl_xor(mn, m, 64); mn = o < mn ? o : mn; }
    __syncthreads();            // (sh.red was last read before this function)
    if (lane == 0) sh.red[wave] = mn;
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int w = 0; w < NWAVES; ++w) mn = sh.red[w] < mn ? sh.red[w] : mn;
    if (updated) {
        for (int i = 0; i < 4 * SD_INTS; ++i) a.sd->v[i] = sh.sd.v[i];
        a.chk.strength[2] = 7;      // video.loop_filter_sharpness after prepare_segments_data(1, 7)
    }
    a.o.flags[0] = 0;               // the fallback has run (the launch before this one): zero at rest
    const int32_t w[5] = {sh.repl, __float_as_int(__fdiv_rn(sum, (float)mbs)), __float_as_int(mn), *a.err, updated ? 1 : 0};
    for (int i = 0; i < 5; ++i) {
        a.chk.stats[i] = w[i];
        __hip_atomic_store(&a.chk.verdict[i], w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(&a.chk.verdict[5], (int32_t)a.chk.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // the host polls this word
}

#ifdef LF_STAMPS
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#else
#define STAMP(v)
#endif

__device__ __forceinline__ void loop_filter4_body(const Args &a) {
    __shared__ __attribute__((aligned(16))) Shared sh;
    const int band = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    lds_flag_t *const flag = (lds_flag_t *)sh.flag;
    if (threadIdx.x < 16) flag[threadIdx.x] = 0;
    if (threadIdx.x == 0) sh.first_lf0 = 0x7fffffff;
    // check_SSIM's tail in this launch (vp8enc.cpp:252-261): `if (min1 > 0.95) prepare_segments_data(1, 7)`.  Every workgroup
    // takes the frame's minimum SSIM itself (8 160 floats at 1080p: a few microseconds) and, above 0.95, filters with the
    // segment data that call produces -- nobody waits for a kernel that would have done it.
    const int32_t *sdv = a.sd->v;
    if (a.chk.on) {
        float mn = 2.0f;
        const int mbs_all = a.mbw * a.mbh;
        for (int i = threadIdx.x; i < mbs_all; i += NWAVES * 64) { const float v = a.o.ssim[i]; mn = v < mn ? v : mn; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const float o = __shfl_xor(mn, m, 64); mn = o < mn ? o : mn; }
        if (lane == 0) sh.red[wave] = mn;
        __syncthreads();
        mn = sh.red[0];
#pragma unroll
        for (int w = 1; w < NWAVES; ++w) mn = sh.red[w] < mn ? sh.red[w] : mn;
        if (mn > 0.95f) {   // (the reference compares with the double 0.95: no float lies between 0.95f and 0.95)
            if (threadIdx.x == 0) {
                const int refqi[4] = {a.chk.refqi[0], a.chk.refqi[1], a.chk.refqi[2], a.chk.refqi[3]};
                fill_segment_data(&sh.sd, 0, refqi, a.chk.qi_min, a.chk.strength[0], a.chk.strength[1], true);
            }
            sdv = sh.sd.v;
            __syncthreads();
        }
        if (band >= a.nbands) {
            verdict_workgroup(a, sh, sdv != a.sd->v);
            return;
        }
    } else if (band >= a.nbands) {
        return;
    }
    if (threadIdx.x < 4) {   // a table read per macroblock: selecting among four registers by a per-lane index compiles to branches
        const int32_t *sd = sdv + threadIdx.x * SD_INTS;
        const int il = sd[SD_INTERIOR_LIMIT] & 0xff;
        sh.lim[threadIdx.x] = make_int4(il, il - (sd[SD_MBEDGE_LIMIT] & 0xff) * 2 - 1, il - (sd[SD_SUB_BEDGE_LIMIT] & 0xff) * 2 - 1,
                                        sd[SD_HEV_THRESHOLD] & 0xff);
    }
    // The kernel's own clock (constant 100 MHz): band 0 stamps the start, the drainer of the last band adds end - start to an
    // accumulator the host reads with the profile (vp8hip_profile_read_clock).
    // hipEvents around a launch also count the time its packet waits for the queue when many streams share the part.
    unsigned long long *clk = reinterpret_cast<unsigned long long *>(a.err + 4);   // {start, sum of ticks, launches, sum of shader-clock cycles per tick x 1000, launches left out of that sum, launches whose last wave changed slots}
    // (the start is SUBTRACTED from the sum of ticks here and the end added by the frame's last wave: no wave has to read the start back.
    // The word is meaningful when no launch is in flight, which is when the host reads it.)
    if (band == 0 && threadIdx.x == 0) {
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        atomicAdd(clk + 1, 0ull - t_start);
#ifdef LF_STAMPS
        __hip_atomic_store(clk, t_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    const unsigned long long cyc0 = __builtin_amdgcn_s_memtime(), tick0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t hwid0 = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    __syncthreads();
    const int mbw = a.mbw, mbh = a.mbh;
    const int band_row0 = band * ROWS;
    {
        // CPU_kernels.cl:990: a macroblock whose segment has level 0 ends the plane.  Levels are >= 1 for
        // every quantizer the host produces, so the scan over segment ids runs only if one IS zero.
        const bool any0 = sdv[SD_LOOP_FILTER_LEVEL] == 0 || sdv[SD_INTS + SD_LOOP_FILTER_LEVEL] == 0 ||
                          sdv[2 * SD_INTS + SD_LOOP_FILTER_LEVEL] == 0 || sdv[3 * SD_INTS + SD_LOOP_FILTER_LEVEL] == 0;
        if (any0) {
            int first = 0x7fffffff;
            for (int mb = threadIdx.x; mb < mbw * mbh; mb += NWAVES * 64)
                if (sdv[a.o.seg[mb] * SD_INTS + SD_LOOP_FILTER_LEVEL] == 0) { first = mb; break; }
            if (first != 0x7fffffff) atomicMin(&sh.first_lf0, first);
            __syncthreads();
        }
    }
    // byte offsets of the pieces of `sh` in LDS
#define LDS_OFFSET(member) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)&(member)[0])
    const uint32_t Y0 = LDS_OFFSET(sh.Y), C0 = LDS_OFFSET(sh.C), D0 = LDS_OFFSET(sh.dummy), P0 = LDS_OFFSET(sh.par);
    const bool publishes = band + 1 < a.nbands;   // a next band exists: every row of this band is real
    const int rows_real = imin(ROWS, mbh - band_row0);   // MB rows of this band inside the frame
    // lane roles shared by the waves that work on row pairs: half-wave = row, 0-15 luma, 16-23 U, 24-31 V
    const int half = lane >> 5, l32 = lane & 31;
    const int pl = l32 < 16 ? 0 : (l32 < 24 ? 1 : 2);
    const int li = pl == 0 ? l32 : (pl == 1 ? l32 - 16 : l32 - 24);
    const int msz = pl == 0 ? 16 : 8;
    // (values, not a reference to one of the three descriptors: through a per-lane reference every use is a vector load from the
    // kernel-argument segment with an s_waitcnt vmcnt(0) behind it -- which in the drainer waits for every store in flight)
    // (... and the three descriptors pinned in scalar registers first: a select between two loads becomes a load from a selected
    // address again)
    uint64_t yp_ = (uint64_t)a.Y.p, up_ = (uint64_t)a.U.p, vp_ = (uint64_t)a.V.p;
    int ys_ = a.Y.stride, us_ = a.U.stride, vs_ = a.V.stride;
    asm volatile("" : "+s"(yp_), "+s"(up_), "+s"(vp_), "+s"(ys_), "+s"(us_), "+s"(vs_));
    struct { g_u8 *p; int stride; } P;
    P.p = (g_u8 *)(pl == 0 ? yp_ : (pl == 1 ? up_ : vp_));
    P.stride = pl == 0 ? ys_ : (pl == 1 ? us_ : vs_);
    const uint32_t plane0 = pl == 0 ? Y0 : (pl == 1 ? C0 : C0 + VOFF);
    const int mb_shift = pl == 0 ? 6 : 5;         // bytes of a macroblock inside a plane row: 64 / 32

    // ---------------------------------------------------------------------------------------------
    // porter waves, one per row pair (= per worker, on the worker's SIMD: a worker alone leaves a third of its SIMD's issue
    // slots unused, and four porters with a row pair each fit into those; ONE feeder and ONE drainer for the whole band each
    // needed more slots than one SIMD has left and the workers waited for them).  Iteration T of porter i:
    //   feed  macroblock T - r of its rows r (HBM bytes -> plane dwords, and the macroblock's parameter record), the load
    //         issued PIPE iterations earlier;
    //   drain macroblock T - LAG - r: final after step T - LAG + 1 of the workers (its last three columns by P1 of macroblock
    //         x + 1, its last three rows by P2 of the row below) -- plane dwords -> bytes -> HBM.  The bottom four rows of a
    //         band that publishes go the publisher's way (and are stored by the next band's loader).
    // The order is fixed, so the slot a feed overwrites (macroblock x - RING) was drained RING - LAG iterations ago, and the
    // compiler can count the loads and stores in flight (a wait for a load never waits for a younger store).
    // ---------------------------------------------------------------------------------------------
    if (wave >= W_PORTER) __builtin_amdgcn_s_setprio(1);   // helpers: ahead of other kernels' waves on the SIMD (a video's other streams), behind the workers
    if (wave >= W_PORTER && wave < W_PORTER + WORKERS) {
        const int pi = wave - W_PORTER;
        const int r = 2 * pi + half, gr = band_row0 + r;
        if (band_row0 + 2 * pi >= mbh) return;       // neither row is inside the frame (its worker has left as well)
        const bool row_real = gr < mbh;
        const int first_lf0 = sh.first_lf0;
        constexpr int PIPE = 3, LAG = 5;
        static_assert(LAG + PIPE <= RING, "a feed must find its slot drained");
        const g_u8 *grow = P.p + (ptrdiff_t)(gr * msz + li) * P.stride;            // this lane's pixel row in the frame
        const uint32_t lrow = plane0 + (uint32_t)(4 + msz * r + li) * LS;          // ... and in the plane
        const bool drains = row_real && !(publishes && r == ROWS - 1 && li >= msz - 4);
        const bool par_lane = l32 == 0;                                            // lane 0 of each half: the parameter record of its row
        const int32_t *pseg = a.o.seg + imin(gr, mbh - 1) * mbw, *pmask = a.o.mask + imin(gr, mbh - 1) * mbw;
        const uint32_t lpar = P0 + (uint32_t)(r * RING) * PAR_BYTES;
        const int T0 = 2 * pi, T1 = mbw + 2 * pi + 1;     // feeds: T0 <= T < T1 (the second row's last macroblock in iteration mbw + 2 pi)
        uint4 px[PIPE];
        int seg[PIPE], maskv[PIPE];
#define FEED_ISSUE(p, T_)                                                                                          \
    {                                                                                                              \
        const int x = (T_) - r;                                                                                    \
        px[p] = make_uint4(0, 0, 0, 0);                                                                            \
        seg[p] = maskv[p] = 0;                                                                                     \
        if (row_real & (x >= 0) & (x < mbw)) {                                                                     \
            px[p] = gld128(grow + x * msz);   /* chroma lanes use 8 of the 16 bytes; at the right frame edge the rest is margin */ \
            if (par_lane) { seg[p] = pseg[x]; maskv[p] = pmask[x]; }                                               \
        }                                                                                                          \
    }
#pragma unroll
        for (int p = 0; p < PIPE; ++p) FEED_ISSUE(p, T0 + p)
        if (lane == 0) flag[F_FEED + pi] = T0;
        for (int Tb = T0; Tb < T1 + LAG; Tb += PIPE) {
#pragma unroll
            for (int p = 0; p < PIPE; ++p) {
                const int T = Tb + p;
                if (T < T1) {
                    // the bottom rows of a publishing band's last row are read by the publisher: macroblock x - RING must have gone
                    if (publishes && pi == WORKERS - 1) LF_WAIT(flag[F_PUB] < T - (ROWS - 1) - (RING - 1), 4)
                    const int x = T - r;
                    if (row_real & (x >= 0) & (x < mbw)) {
                        const uint32_t d = lrow + ((uint32_t)(x & (RING - 1)) << mb_shift);
                        st128(d, unpack4(px[p].x));
                        st128(d + 16, unpack4(px[p].y));
                        if (pl == 0) { st128(d + 32, unpack4(px[p].z)); st128(d + 48, unpack4(px[p].w)); }
                        if (par_lane) {
                            const int4 lim = sh.lim[seg[p] & 3];
                            const bool do_filter = (gr * mbw + x) < first_lf0;
                            const int il = lim.x;
                            const uint32_t dp = lpar + (uint32_t)(x & (RING - 1)) * PAR_BYTES;
                            st128(dp, make_int4((do_filter & (x > 0)) ? il : -1, (do_filter & (maskv[p] != 0)) ? il : -1, (do_filter & (gr > 0)) ? il : -1, lim.w));
                            st64(dp + 16, make_int2(lim.y, lim.z));
                        }
                    }
                    lds_fence();
                    if (lane == 0) flag[F_FEED + pi] = T + 1;
                    FEED_ISSUE(p, T + PIPE)
                }
                const int D = T - LAG;
                if (D >= T0 && D < T1) {
                    const int need = 2 * (D + 1) + 2, below = imin(pi + 1, WORKERS - 1);
                    LF_WAIT(flag[pi] < need || flag[below] < need, 4)
                    if (band > 0 && pi == 0) {
                        // the four rows above the band over macroblock D: final now that row 0 has filtered across them (P2 of step D)
                        if ((half == 0) & (li < 4) & (D < mbw)) {
                            const uint32_t sl = plane0 + (uint32_t)li * LS + ((uint32_t)(D & (RING - 1)) << mb_shift);
                            uint4 v;
                            v.x = pack4(ld128(sl));
                            v.y = pack4(ld128(sl + 16));
                            g_u8 *g = P.p + (ptrdiff_t)(band_row0 * msz - 4 + li) * P.stride + D * msz;
                            if (pl == 0) {
                                v.z = pack4(ld128(sl + 32));
                                v.w = pack4(ld128(sl + 48));
                                gst128(g, v);
                            } else {
                                gst64(g, make_uint2(v.x, v.y));
                            }
                        }
                        lds_fence();
                        if (lane == 0) flag[F_TOPDRAIN] = D + 1;
                    }
                    const int x = D - r;
                    if (drains & (x >= 0) & (x < mbw)) {
                        const uint32_t sl = lrow + ((uint32_t)(x & (RING - 1)) << mb_shift);
                        uint4 v;
                        v.x = pack4(ld128(sl));
                        v.y = pack4(ld128(sl + 16));
                        g_u8 *g = const_cast<g_u8 *>(grow) + x * msz;
                        if (pl == 0) {
                            v.z = pack4(ld128(sl + 32));
                            v.w = pack4(ld128(sl + 48));
                            gst128(g, v);
                        } else {
                            gst64(g, make_uint2(v.x, v.y));
                        }
                    }
                }
            }
        }
        if (band + 1 == a.nbands && pi == (rows_real - 1) / 2 && lane == 0) {   // the frame's last rows: this wave is the last to finish real work
            // (this is the tail of every frame's chain: no load to wait for, no 64-bit division -- fire-and-forget additions only)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            atomicAdd(clk + 1, t1);
            atomicAdd(clk + 2, 1ull);
            // the shader clock this wave saw while it ran: s_memtime cycles per 100 MHz tick (MI355X_MICROARCH.md, DVFS (6)), x 1000
            const float cyc_f = (float)(__builtin_amdgcn_s_memtime() - cyc0), tick_f = (float)(t1 - tick0 + 1);
            const unsigned long long ratio = (unsigned long long)(cyc_f * 1000.0f / tick_f);
            const uint32_t hwid1 = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
            // a wave that was context-switched (the hardware scheduler rotating an oversubscribed set of queues) comes back on
            // another slot, whose cycle counter is another one: such launches are counted, not averaged
            if (ratio > 100000ull) atomicAdd(clk + 4, 1ull); else atomicAdd(clk + 3, ratio);
            if (hwid1 != hwid0) atomicAdd(clk + 5, 1ull);
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------
    // publisher wave: the bottom four rows of the band's last row, macroblock by macroblock as they become final for the band
    // below (their last three columns by P1 of macroblock x + 1), to the hand-off buffer in HBM: 32 lanes, one aligned 8-byte
    // granule {four samples, tag} each, written through (sc1).  The tag (gbase + x + 1) says "this launch's macroblock x": the
    // loader below polls the granules themselves, one round trip, no counter and no wait for a store to retire.
    // ---------------------------------------------------------------------------------------------
    const int hk = li;                                               // (loader and publisher, lanes < 32) dword index inside the plane's 4 rows
    const int hnd = pl == 0 ? 4 : 2;                                  // dwords per row
    const int hr = hk / hnd, hj = hk % hnd;
    if (wave == W_PUBLISHER) {
        if (!publishes) return;
        const uint32_t sp = plane0 + (uint32_t)(4 + msz * (ROWS - 1) + (msz - 4) + hr) * LS + 16 * hj;
        uint2 *out = a.handoff + (size_t)band * mbw * 32 + lane;
        const uint32_t tag0 = (uint32_t)(a.gbase + 1 - (a.stall_test ? (1 << 20) : 0));
        for (int x = 0; x < mbw; ++x) {
            const int done = x + 1 < mbw ? 2 * (x + ROWS) + 1 : 2 * (x + ROWS - 1) + 2;
            LF_WAIT(flag[WORKERS - 1] < done, 2)
            int4 q = make_int4(0, 0, 0, 0);
            if (lane < 32) q = ld128(sp + ((uint32_t)(x & (RING - 1)) << mb_shift));
            lds_fence();
            if (lane == 0) flag[F_PUB] = x + 1;     // the plane's slot is free (what the last porter waits for)
            if (lane < 32) st64_sc1(out + (size_t)x * 32, pack4(q), tag0 + (uint32_t)x);
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------
    // loader wave: the previous band's bottom rows, from the hand-off buffer -> plane rows 0-3.  The granule of macroblock
    // x + 1 is requested before macroblock x is expanded; a granule whose tag is not this launch's yet is asked for again.
    // (The rows' final store, once row 0 has filtered across them, is porter 0's.)
    // ---------------------------------------------------------------------------------------------
    if (wave == W_LOADER) {
        if (band == 0) return;
        const uint32_t sp = plane0 + (uint32_t)hr * LS + 16 * hj;
        const uint2 *in = a.handoff + (size_t)(band - 1) * mbw * 32 + (lane & 31);
        const uint32_t tag0 = (uint32_t)(a.gbase + 1);
        uint2 cur = ld64_sc1(in);
        for (int x = 0; x < mbw; ++x) {
            uint2 nxt = make_uint2(0, 0);
            if (x + 1 < mbw) nxt = ld64_sc1(in + (size_t)(x + 1) * 32);
            {
                int spins = 0;
                while (__builtin_amdgcn_readfirstlane(__any((int)(cur.y != tag0 + (uint32_t)x)))) {
                    if (flag[F_ABORT]) return;
                    if (++spins > SPIN_LIMIT / 8) { flag[F_ABORT] = 1; *a.err = 1; }
                    __builtin_amdgcn_s_sleep(2);
                    cur = ld64_sc1(in + (size_t)x * 32);
                }
            }
            // ring space: the slot holds macroblock x - RING, which porter 0 stores in its drain step x - RING
            LF_WAIT(flag[F_TOPDRAIN] < x - (RING - 1), 2)
            if (lane < 32) st128(sp + ((uint32_t)(x & (RING - 1)) << mb_shift), unpack4(cur.x));
            lds_fence();
            if (lane == 0) flag[F_TOP] = x + 1;
            cur = nxt;
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------
    // worker waves: two MB rows each, nothing but the two phases
    // ---------------------------------------------------------------------------------------------
    // the helper waves share SIMDs with the workers: let them issue only in idle slots
#ifndef LF_PRIO
#define LF_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(LF_PRIO);
    const int r = 2 * wave + half;              // local MB row
    const int gr = band_row0 + r;               // global MB row
    const bool row_real = gr < mbh;
    if (band_row0 + 2 * wave >= mbh) {          // neither row of this wave is inside the frame
        if (lane == 0) flag[wave] = BIG;
        return;
    }
    const uint32_t ringmask = pl == 0 ? RING * 64 - 1 : RING * 32 - 1;
    const uint32_t p1_base = plane0 + (uint32_t)(4 + msz * r + li) * LS;    // P1: this lane's pixel row
    const uint32_t p2_base = plane0 + (uint32_t)(msz * r) * LS + 4 * li;     // P2: this lane's column, from the row four above the macroblock
    const uint32_t par_base = P0 + (uint32_t)(r * RING) * PAR_BYTES;
    const uint32_t dummy = D0 + lane * 64;
    const int chroma_m1 = pl == 0 ? 0 : -1;
    const bool top_wave = wave == 0 && band > 0;
    const int steps = mbw + 2 * wave + 2;       // the wave's second row does its last macroblock in step mbw + 2 wave
    // the feeder's first step
    {
        int spins = 0;
        while (flag[F_FEED + wave] < 1) {
            if (flag[F_ABORT]) return;
            if (++spins > SPIN_LIMIT) { flag[F_ABORT] = 1; *a.err = 1; }
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    }
#ifdef LF_STAMPS
    unsigned long long st_p1 = 0, st_wait = 0, st_p2 = 0, st_spins = 0, st_t0, st_t1, tl0 = 0, tl64 = 0;
#endif
    // The step, arranged around its LDS round trips (a lone wave: 5.3 cycles per instruction of any kind, 50-130 cycles per round trip;
    // LDS operations of one wave execute in program order -- scripts/ubench/lds_order.hip -- so a flag written right behind the data
    // is seen behind them and no store is ever waited for):
    //   P1    operands requested half a step earlier (the macroblock's own sixteen samples and its parameter record, behind the poll)
    //         and at the end of the previous step (the four samples to the left, behind P2's stores): line_pre, line_post, stores,
    //         flag 2S+1
    //   poll  the three flags are read, then the macroblock's own sixteen rows for P2 (written by this wave: no wait needed);
    //         line_pre of P2 runs while all of that is in flight; only then are the flags looked at
    //   P2    the four rows above and the next step's P1 operands are requested; line_post, stores, the next step's left samples
    //         requested, flag 2S+2
    int4 n_pa = make_int4(0, 0, 0, 0), n_v0 = n_pa, n_v1 = n_pa, n_v2 = n_pa, n_v3 = n_pa, n_v4 = n_pa;
    int2 n_pb = make_int2(0, 0);
// (Unconditional: an LDS read of a slot that holds nothing yet is harmless -- the values are used under `on` only -- and reads under a
// predicate cost a register move per destination to merge the two paths.)
#define P1_REQUEST(S_)                                                                        \
    {                                                                                         \
        const uint32_t sl = (uint32_t)((S_) - r) & (RING - 1);                                \
        const uint32_t sn = p1_base + (sl << mb_shift), pn = par_base + sl * PAR_BYTES;       \
        n_pa = ld128(pn); n_pb = ld64(pn + 16);                                               \
        n_v1 = ld128(sn); n_v2 = ld128(sn + 16); n_v3 = ld128(sn + 32); n_v4 = ld128(sn + 48); \
    }
// the four samples to the left of step S_'s macroblock (the previous macroblock's last columns, which P2 has just written)
#define P1_REQUEST_LEFT(S_) n_v0 = ld128(p1_base + (((((uint32_t)((S_) - r) & (RING - 1)) << mb_shift) - 16) & ringmask));
    P1_REQUEST(0)
    P1_REQUEST_LEFT(0)
    const int up = imax(wave - 1, 0);
    for (int S = 0; S < steps; ++S) {
        STAMP(st_t0);
        const int x = S - r;
        // (`&`, not `&&`: one predicate, one exec mask -- short-circuit evaluation nests the regions)
        const bool on = row_real & (x >= 0) & (x < mbw);
        const uint32_t slot = (uint32_t)x & (RING - 1);
        const uint32_t colB = slot << mb_shift, colA = (colB - 16) & ringmask;
        Limits L;
        int il_p1 = -1, il_in = -1, il_p2 = -1;
        L.hev_thr = n_pa.w; L.mb_delta = n_pb.x; L.b_delta = n_pb.y;
        // ---- P1: vertical edges, lane = pixel row -------------------------------------------------
        if (on) {
            const uint32_t s = p1_base + colB;
            const int4 v1 = n_v1, v2 = n_v2, v3 = n_v3, v4 = n_v4;
            il_p1 = n_pa.x; il_in = n_pa.y; il_p2 = n_pa.z;
            int t[20] = {0, 0, 0, 0, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w, v4.x, v4.y, v4.z, v4.w};
            LinePre pre;
            line_pre(t, L, pre);
            const int4 v0 = n_v0;
            t[0] = v0.x; t[1] = v0.y; t[2] = v0.z; t[3] = v0.w;
            line_post(t, L, il_p1, il_in, il_in | chroma_m1, pre);
#pragma unroll
            for (int k = 1; k < 18; ++k) t[k] = satb(t[k]);
            // macroblock 0 has nothing to its left (the slot belongs to a macroblock the porter may be bringing in); chroma
            // lanes own eight columns
            st128(x > 0 ? p1_base + colA : dummy, make_int4(t[0], t[1], t[2], t[3]));
            st128(s, make_int4(t[4], t[5], t[6], t[7]));
            st128(s + 16, make_int4(t[8], t[9], t[10], t[11]));
            const uint32_t sh_ = pl == 0 ? s : dummy;
            st128(sh_ + 32, make_int4(t[12], t[13], t[14], t[15]));
            st128(sh_ + 48, make_int4(t[16], t[17], t[18], t[19]));
        }
        asm volatile("" ::: "memory");
        flag[wave] = 2 * S + 1;   // (every lane, the same word: no exec mask to set up and restore.  Behind the stores: LDS executes a wave's operations in order)
        // ---- the one poll of the step ------------------------------------------------------------
        const int need_up = 2 * S + 1;                           // P1 of the rows above (their macroblock x+1)
        const int need_feed = imin(S + 2, mbw + 2 * wave + 1);   // the macroblocks of the next step's P1 (this wave's porter)
        const int need_top = imin(S + 1, mbw);                   // the rows above the band over this step's macroblock of row 0
        int f_up = need_up, f_top = need_top;
        if (wave > 0) f_up = flag[up];
        if (top_wave) f_top = flag[F_TOP];
        int f_feed = flag[F_FEED + wave];
        // P2's own sixteen rows (this wave's P1 has just written them: in order behind its stores), and everything of P2 that does
        // not need the four rows above -- while the flags travel
        const uint32_t s2 = p2_base + colB;
        int t2[20];
#pragma unroll
        for (int k = 4; k < 20; ++k) t2[k] = ld32(s2 + k * LS);   // chroma lanes: rows 12-19 are don't-care
        LinePre pre2;
        line_pre(t2, L, pre2);
        __builtin_amdgcn_sched_barrier(0);      // (the flags are looked at behind this work, not in front of it: they need the time)
        STAMP(st_t1);
#ifdef LF_STAMPS
        st_p1 += st_t1 - st_t0; st_t0 = st_t1;
#endif
        // Everything the poll compares is the same in all lanes; readfirstlane says so to the compiler, which otherwise
        // builds the loop out of exec-mask bookkeeping.
        if (!__builtin_amdgcn_readfirstlane((int)((f_up >= need_up) & (f_feed >= need_feed) & (f_top >= need_top)))) {
            // What this wave waits for is the work of a wave that shares its SIMD (its porter) or of the wave above: a polling
            // wave at high priority takes the issue slots its own supplier needs.
            __builtin_amdgcn_s_setprio(0);
            for (int spins = 0;; ++spins) {
                if (wave > 0) f_up = flag[up];
                if (top_wave) f_top = flag[F_TOP];
                f_feed = flag[F_FEED + wave];
                const int f_abort = flag[F_ABORT];
                const bool ok = (f_up >= need_up) & (f_feed >= need_feed) & (f_top >= need_top);
                const int state = __builtin_amdgcn_readfirstlane(f_abort ? 2 : (ok ? 1 : 0));
                if (state == 1) break;
                if (state == 2) return;
#ifdef LF_STAMPS
                st_spins += (f_feed < need_feed) ? 1000 : 1;
#endif
                if (spins > SPIN_LIMIT) { flag[F_ABORT] = 1; *a.err = 1; }
                if (spins < 32) asm volatile("s_nop 3"); else __builtin_amdgcn_s_sleep(1);   // the flag is usually a few hundred cycles away: a tight poll first, naps when it is not
            }
            __builtin_amdgcn_s_setprio(LF_PRIO);
        }
        asm volatile("" ::: "memory");
#ifdef LF_STAMPS
        if (wave == 0 && (S == 0 || S == 64)) { unsigned long long tt = __builtin_amdgcn_s_memrealtime(); if (S == 0) tl0 = tt; else tl64 = tt; }
#endif
        // the four rows above the macroblock (the row above has done P1 of its macroblock x + 1: what the poll has seen), then the
        // next step's P1 operands
#pragma unroll
        for (int k = 0; k < 4; ++k) t2[k] = ld32(s2 + k * LS);
        P1_REQUEST(S + 1)
        STAMP(st_t1);
#ifdef LF_STAMPS
        st_wait += st_t1 - st_t0; st_t0 = st_t1;
#endif
        // ---- P2: horizontal edges, lane = pixel column ---------------------------------------------
        if (on) {
            line_post(t2, L, il_p2, il_in, il_in | chroma_m1, pre2);
#pragma unroll
            for (int k = 1; k < 18; ++k) t2[k] = satb(t2[k]);
#pragma unroll
            for (int k = 1; k < 12; ++k) st32(s2 + k * LS, t2[k]);   // rows 1-3: the bottom of the row above (row 0 of the frame: rows nobody reads)
            if (pl == 0) {
#pragma unroll
                for (int k = 12; k < 18; ++k) st32(s2 + k * LS, t2[k]);
            }
        }
        asm volatile("" ::: "memory");
        P1_REQUEST_LEFT(S + 1)    // (behind P2's stores, which wrote those columns)
        flag[wave] = 2 * S + 2;
        STAMP(st_t1);
#ifdef LF_STAMPS
        st_p2 += st_t1 - st_t0;
#endif
    }
    flag[wave] = BIG;
#ifdef LF_STAMPS
    if (lane == 0 && band < 2) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.gprog + 1024) + (band * WORKERS + wave) * 4;
        o[0] = st_wait; o[1] = st_p1; o[2] = st_p2; o[3] = st_spins;
    }
    if (lane == 0 && wave == 0 && band < 16) {   // the band's timeline, 100 MHz ticks since the launch's start
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.gprog + 1024) + 8 * 4 + band * 3;
        const unsigned long long t0 = __hip_atomic_load(clk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o[0] = tl0 - t0; o[1] = tl64 - t0; o[2] = __builtin_amdgcn_s_memrealtime() - t0;
    }
#endif
}

__global__ __launch_bounds__(NWAVES * 64) void k_loop_filter4(Args a) { loop_filter4_body(a); }
static_assert(sizeof(BatchOf<Args>) <= 4096, "a batch's argument blocks travel in the 4 KiB kernel-argument segment");
__global__ __launch_bounds__(NWAVES * 64) void k_loop_filter4_b(BatchOf<Args> b) { loop_filter4_body(b.item[blockIdx.z]); }

}  // namespace lf4

size_t loop_filter4_handoff_bytes(int mbw, int mbh) { return (size_t)((mbh + lf4::ROWS - 1) / lf4::ROWS) * mbw * 32 * sizeof(uint2); }

static lf4::Args loop_filter4_args(hipStream_t s, const Frame &recon, const MBOut &o, SegData *d_sd, int32_t *progress, void *handoff, int mbw, int mbh,
                                   unsigned launch_no, int stall_test, const LfCheck *chk) {
    lf4::Args a;
    if (chk) a.chk = *chk;
    else a.chk.on = 0;
    a.Y = recon.Y[0];
    a.U = recon.U;
    a.V = recon.V;
    a.o = o;
    a.sd = d_sd;
    a.gprog = progress;
    a.handoff = static_cast<uint2 *>(handoff);
    a.mbw = mbw;
    a.mbh = mbh;
    a.nbands = (mbh + lf4::ROWS - 1) / lf4::ROWS;
    // the hand-off tags are never reset: every launch tags inside its own window (wraps after ~2^31/(mbw+2)
    // launches; the host zeroes the buffer when the window index wraps)
    const unsigned window = 0x7fffffffu / (unsigned)(mbw + 2) - 1;
    const unsigned n = launch_no % window;
    if (n == 0) (void)hipMemsetAsync(handoff, 0, loop_filter4_handoff_bytes(mbw, mbh), s);
    a.gbase = (int)(n * (unsigned)(mbw + 2));
    a.err = progress + LF_ERR_WORD;
    a.stall_test = stall_test;
    return a;
}
static bool lf_skip() {
    static const bool skip = experiment_skip("lf");
    return skip;   // timing experiment only
}

void launch_loop_filter4(hipStream_t s, const Frame &recon, const MBOut &o, SegData *d_sd, int32_t *progress, void *handoff,
                         int mbw, int mbh, unsigned launch_no, int stall_test, const LfCheck *chk) {
    const lf4::Args a = loop_filter4_args(s, recon, o, d_sd, progress, handoff, mbw, mbh, launch_no, stall_test, chk);
    if (lf_skip()) return;
    VP8_LAUNCH(lf4::k_loop_filter4, dim3(a.nbands + (a.chk.on ? 1 : 0)), dim3(lf4::NWAVES * 64), 0, s, a);   // + the verdict workgroup
}

void launch_loop_filter4_batch(hipStream_t s, const Frame *const *recon, const MBOut *const *o, SegData *const *d_sd,
                               int32_t *const *progress, void *const *handoff, int mbw, int mbh, const unsigned *launch_no, int n, const LfCheck *chk) {
    BatchOf<lf4::Args> b;
    b.n = n;
    bool any = false;
    for (int i = 0; i < n; ++i) {
        b.item[i] = loop_filter4_args(s, *recon[i], *o[i], d_sd[i], progress[i], handoff[i], mbw, mbh, launch_no[i], 0, chk ? &chk[i] : nullptr);
        any = any || b.item[i].chk.on;
    }
    if (lf_skip()) return;
    VP8_LAUNCH(lf4::k_loop_filter4_b, dim3(b.item[0].nbands + (any ? 1 : 0), 1, n), dim3(lf4::NWAVES * 64), 0, s, b);
}

}  // namespace vp8
