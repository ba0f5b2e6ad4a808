// kernels_lf2.hip -- VP8 normal loop filter, banded wavefront in LDS (gfx950).
//
// Same arithmetic and ordering semantics as loop_filter_frame_luma/_chroma (CPU_kernels.cl:970-1075,
// :1333-1439; edge filters :829-926) -- see kernels_lf.hip for the dependency argument: MB(x,y) needs
// MB(x-1,y) and MB(x+1,y-1).  The frame-long critical path is (mb_w + 2*mb_h) macroblock steps of
// 8 dependent edge filters each, so the design goal is: NOTHING but ALU and LDS on that path.
//
//   * A workgroup owns a band of ROWS consecutive MB rows.  A wave runs two rows (32 lanes each:
//     lanes 0-15 luma rows/columns, 16-23 U, 24-31 V); row r handles MB x = S - 2r at step S, so the
//     second row of a wave depends only on the wave's own previous step.
//   * The only data a row needs from the row above -- its bottom four pixel rows -- lives in an LDS
//     ring ("strip"); waves publish steps done in LDS counters (data ready / ring space).
//   * A macroblock's own pixels are still untouched by any filter when its turn comes, so they are
//     prefetched from HBM a step ahead with plain loads; each step writes back the 16x16 (8x8) block
//     shifted by (-4,-4), which is exactly the set of pixels no later filter touches.
//   * Between bands the bottom strip travels through the frame itself (write-through sc1 stores + an
//     HBM counter); a loader wave per band copies it into the LDS ring ahead of use, so that latency
//     is paid once per band, not per step.
// Quirks kept: unsaturated register carry between the edges of one line (:1024,:1062) and the
// "level 0 leaves the plane" exit (:990) via first_lf0.
#include "vp8hip_dev.h"

namespace vp8 {

namespace lf2 {

constexpr int WORKERS = 4;             // worker waves per band
constexpr int ROWS = 2 * WORKERS;      // MB rows per band
constexpr int RING_MB = 16;            // strip ring length in macroblocks
constexpr int RWY = RING_MB * 16, RWC = RING_MB * 8;   // ring widths in pixels
constexpr int STRIP_BYTES = 4 * RWY + 2 * 4 * RWC;     // Y, U, V bottom strips of one MB row
constexpr int TILE_YS = 24, TILE_CS = 12;              // work-tile row strides
constexpr int TILE_BYTES = 16 * TILE_YS + 2 * 8 * TILE_CS;
constexpr int TILE_SLOTS = 4;           // steps a finished tile stays in LDS for the writer wave

// ---- edge filters -------------------------------------------------------------------------------
// The reference works on u = pixel - 128 (CPU_kernels.cl:928-956); every quantity it forms is a
// difference of two samples or sample +- correction, so the same arithmetic runs on the pixel values
// themselves.  Here samples carry a +BIAS so that they stay positive even when an unsaturated
// register (reference quirk, :1024/:1062) dips below 0, which lets |a-b| be one v_sad_u16.
// Clamps that cannot trigger are dropped: (27w+63)>>7 etc. lie in [-27,27] for w in [-128,127], and
// a+3 / a+4 can only exceed the upper bound.
constexpr int BIAS = 64;
__device__ __forceinline__ int c128(int v) { return iclamp(v, -128, 127); }
__device__ __forceinline__ int ad(int a, int b) { return (int)__builtin_amdgcn_sad_u16((uint32_t)a, (uint32_t)b, 0u); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return imax(imax(a, b), c); }
struct EdgeRegs { int p3, p2, p1, p0, q0, q1, q2, q3; };

__device__ __forceinline__ void edge_masks(const EdgeRegs &e, int int_lim, int edge_lim, int hev_thr, bool &mask,
                                           bool &hev) {
    const int d10 = ad(e.p1, e.p0), dq10 = ad(e.q1, e.q0);
    const int m1 = max3i(ad(e.p3, e.p2), ad(e.p2, e.p1), d10);
    const int m2 = max3i(dq10, ad(e.q2, e.q1), ad(e.q3, e.q2));
    const int edge = ad(e.p0, e.q0) * 2 + (ad(e.p1, e.q1) >> 1);
    mask = (imax(m1, m2) <= int_lim) & (edge <= edge_lim);
    hev = imax(d10, dq10) > hev_thr;
}
__device__ __forceinline__ void filter_mb_edge(EdgeRegs &e, int mb_lim, int int_lim, int hev_thr) {  // :829-883
    bool mask, hev;
    edge_masks(e, int_lim, mb_lim, hev_thr, mask, hev);
    int w = c128(e.p1 - e.q1);
    w = c128(w + (e.q0 - e.p0) * 3);
    w = mask ? w : 0;
    int a = hev ? w : 0;
    const int b = imin(a + 3, 127) >> 3;
    a = imin(a + 4, 127) >> 3;
    e.q0 -= a; e.p0 += b;
    w = hev ? 0 : w;
    a = (w * 27 + 63) >> 7; e.q0 -= a; e.p0 += a;
    a = (w * 18 + 63) >> 7; e.q1 -= a; e.p1 += a;
    a = (w * 9 + 63) >> 7;  e.q2 -= a; e.p2 += a;
}
__device__ __forceinline__ void filter_b_edge(EdgeRegs &e, int b_lim, int int_lim, int hev_thr) {  // :885-926
    bool mask, hev;
    edge_masks(e, int_lim, b_lim, hev_thr, mask, hev);
    int a = c128(e.p1 - e.q1);
    a = hev ? a : 0;
    a = c128(a + (e.q0 - e.p0) * 3);
    a = mask ? a : 0;
    const int b = imin(a + 3, 127) >> 3;
    a = imin(a + 4, 127) >> 3;
    e.q0 -= a; e.p0 += b;
    a = (a + 1) >> 1;
    a = hev ? 0 : a;
    e.q1 -= a; e.p1 += a;
}

// One line of biased samples t[0..msz+3] (t[0..3] precede the macroblock edge) through the MB edge and
// the inner edges.  t[] receives the UNSATURATED results (saturation = the reference's store happens
// when the line is packed, pack4); the p/q registers handed from edge to edge stay unsaturated too.
__device__ __forceinline__ void filter_line(int (&t)[20], int msz, bool has_mb_edge, bool inner, int mb_lim,
                                            int b_lim, int int_lim, int hev_thr) {
    EdgeRegs e;
    e.q0 = t[4]; e.q1 = t[5]; e.q2 = t[6]; e.q3 = t[7];
    if (has_mb_edge) {
        e.p3 = t[0]; e.p2 = t[1]; e.p1 = t[2]; e.p0 = t[3];
        filter_mb_edge(e, mb_lim, int_lim, hev_thr);
        t[1] = e.p2; t[2] = e.p1; t[3] = e.p0;
        t[4] = e.q0; t[5] = e.q1; t[6] = e.q2;
    }
#pragma unroll
    for (int k = 4; k < 16; k += 4) {
        if (inner && k < msz) {
            e.p3 = e.q0; e.p2 = e.q1; e.p1 = e.q2; e.p0 = e.q3;
            e.q0 = t[4 + k]; e.q1 = t[5 + k]; e.q2 = t[6 + k]; e.q3 = t[7 + k];
            filter_b_edge(e, b_lim, int_lim, hev_thr);
            t[2 + k] = e.p1; t[3 + k] = e.p0; t[4 + k] = e.q0; t[5 + k] = e.q1;
        }
    }
}

// four biased samples -> four saturated bytes (the reference's convert_uchar_sat on store)
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    return (uint32_t)sat8(a - BIAS) | ((uint32_t)sat8(b - BIAS) << 8) | ((uint32_t)sat8(c - BIAS) << 16) |
           ((uint32_t)sat8(d - BIAS) << 24);
}
__device__ __forceinline__ int ub(uint32_t w, int k) { return byte_of(w, k) + BIAS; }

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct Args {
    Plane Y, U, V;
    MBOut o;
    const SegData *sd;
    int32_t *gprog;   // [bands] gbase + macroblock steps finished by the LAST row of each band
    int gbase;        // counters only grow: launch n uses the range (n*(mbw+2), (n+1)*(mbw+2)], so no memset
    int mbw, mbh, nbands;
};

struct Shared {
    uint8_t strip[ROWS + 1][STRIP_BYTES];   // strip[r] = bottom rows of the MB row ABOVE local row r
    uint8_t tile[ROWS][TILE_SLOTS][TILE_BYTES];   // work tiles, one slot per step in flight (writer drains)
    volatile int prog[WORKERS];             // steps completed by each worker wave
    volatile int top_ready;                 // macroblocks of strip[0] delivered by the loader
    volatile int pub_done;                  // macroblocks of strip[ROWS] handed to the next band
    volatile int wr_done;                   // steps whose finished blocks the writer has read out of LDS
    int first_lf0;                          // first macroblock whose segment has loop_filter_level 0 (:990)
};

constexpr int NWAVES = WORKERS + 3;         // workers + loader + publisher + writer

__global__ __launch_bounds__(NWAVES * 64) void k_loop_filter2(Args a) {
    __shared__ __attribute__((aligned(16))) Shared sh;
    const int band = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < WORKERS) sh.prog[threadIdx.x] = 0;
    if (threadIdx.x == 0) { sh.top_ready = 0; sh.pub_done = 0; sh.wr_done = 0; sh.first_lf0 = 0x7fffffff; }
    __syncthreads();
    const int mbw = a.mbw, mbh = a.mbh;
    const int band_row0 = band * ROWS;
    {
        // CPU_kernels.cl:990: a macroblock whose segment has level 0 ends the plane.  Levels are >= 1 for
        // every quantizer the host produces, so the scan over segment ids runs only if one IS zero.
        const int32_t *sdv = a.sd->v;
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

    // ---------------------------------------------------------------------------------------------
    // publisher wave: bottom strip of the band's last row (strip[ROWS]) -> the frame (sc1, write-
    // through) -> HBM counter.  Keeps the store drain (s_waitcnt vmcnt(0)) off the workers' path.
    // ---------------------------------------------------------------------------------------------
    if (wave == WORKERS + 1) {
        if (band + 1 >= a.nbands) return;
        // lane < 44: one dword of 4 rows x (5 + 3 + 3) dwords = columns x0-4 .. x0+msz-1 of Y, U, V
        const int pl = lane < 20 ? 0 : (lane < 32 ? 1 : 2);
        const int k = pl == 0 ? lane : (pl == 1 ? lane - 20 : lane - 32);
        const int ndw = pl == 0 ? 5 : 3;
        const int rr = k / ndw, j = k % ndw;
        const Plane &P = pl == 0 ? a.Y : (pl == 1 ? a.U : a.V);
        const int msz = pl == 0 ? 16 : 8, rw = pl == 0 ? RWY : RWC;
        const int y = (band_row0 + ROWS - 1) * msz + (msz - 4) + rr;
        const uint8_t *sp = sh.strip[ROWS] + (pl == 0 ? 0 : (pl == 1 ? 4 * RWY : 4 * RWY + 4 * RWC)) + rr * rw;
        for (int x = 0; x <= mbw; ++x) {
            const int done_step = x + 2 * (ROWS - 1) + 1;   // steps the last wave has finished once MB x is through
            while (sh.prog[WORKERS - 1] < done_step) __builtin_amdgcn_s_sleep(1);
            if (lane < 44) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(sp + ((x * msz - 4 + 4 * j) & (rw - 1)));
                st_sc1(reinterpret_cast<uint32_t *>(P.p + (ptrdiff_t)y * P.stride + x * msz - 4) + j, v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                sh.pub_done = x + 1;
                __hip_atomic_store(&a.gprog[band], a.gbase + x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------
    // loader wave: previous band's bottom strip (in the frame, written with sc1) -> strip[0]
    // ---------------------------------------------------------------------------------------------
    if (wave == WORKERS) {
        if (band == 0) return;
        const int l = lane & 31;
        // lane l < 32: one dword of the 4 x (16 + 8 + 8) pixels above macroblock x
        const int pl = l < 16 ? 0 : (l < 24 ? 1 : 2);
        const int k = pl == 0 ? l : (pl == 1 ? l - 16 : l - 24);     // dword index inside the plane's 4 rows
        const int nd = pl == 0 ? 4 : 2;                               // dwords per row
        const int r = k / nd, j = k % nd;
        const Plane &P = pl == 0 ? a.Y : (pl == 1 ? a.U : a.V);
        const int msz = pl == 0 ? 16 : 8, rw = pl == 0 ? RWY : RWC;
        const int y = band_row0 * msz - 4 + r;
        uint8_t *sp = sh.strip[0] + (pl == 0 ? 0 : (pl == 1 ? 4 * RWY : 4 * RWY + 4 * RWC)) + r * rw;
        for (int x = 0; x < mbw; ++x) {
            const int need = imin(x + 2, mbw + 1);
            while (__hip_atomic_load(&a.gprog[band - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.gbase < need)
                __builtin_amdgcn_s_sleep(2);
            // ring space: the WRITER still reads strip[0] for row 0's finished blocks (row 0: step == x),
            // and it may trail the workers by TILE_SLOTS-1 steps
            while (sh.wr_done < x - (RING_MB - 3)) __builtin_amdgcn_s_sleep(1);
            if (lane < 32) {
                const uint32_t v = ld_sc1(reinterpret_cast<const uint32_t *>(P.p + (ptrdiff_t)y * P.stride + x * msz) + j);
                *reinterpret_cast<uint32_t *>(sp + ((x * msz + 4 * j) & (rw - 1))) = v;
            }
            lds_fence();
            if (lane == 0) sh.top_ready = x + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------
    // writer wave: after every step, the blocks that became final -- 16x16 (8x8) shifted by (-4,-4):
    // four pixel rows from the strip above + msz-4 rows of the row's tile slot -- go LDS -> HBM.
    // Workers therefore issue no global stores and never wait behind one.
    // ---------------------------------------------------------------------------------------------
    if (wave == WORKERS + 2) {
        const int half = lane >> 5, l32 = lane & 31;
        const int pl = l32 < 16 ? 0 : (l32 < 24 ? 1 : 2);
        const int li = pl == 0 ? l32 : (pl == 1 ? l32 - 16 : l32 - 24);
        const int msz = pl == 0 ? 16 : 8, nd = msz / 4;
        const Plane &P = pl == 0 ? a.Y : (pl == 1 ? a.U : a.V);
        const int rw = pl == 0 ? RWY : RWC;
        const int tstride = pl == 0 ? TILE_YS : TILE_CS;
        const int strip_off = pl == 0 ? 0 : (pl == 1 ? 4 * RWY : 4 * RWY + 4 * RWC);
        const int tile_off = pl == 0 ? 0 : (pl == 1 ? 16 * TILE_YS : 16 * TILE_YS + 8 * TILE_CS);
        const int steps = mbw + 1 + 2 * (ROWS - 1);
        for (int S = 0; S < steps; ++S) {
            for (int w = 0; w < WORKERS; ++w)
                while (sh.prog[w] < S + 1) __builtin_amdgcn_s_sleep(1);
            for (int rp = 0; rp < ROWS; rp += 2) {
                const int r = rp + half, gr = band_row0 + r;
                const int x = S - 2 * r;
                const bool row_real = gr < mbh;
                if (gr > mbh || x < 0 || x > mbw) continue;
                const int x0 = x * msz, yy = gr * msz - 4 + li;
                const bool from_top = li < 4;
                if (!(from_top ? gr > 0 : row_real) || yy < 0) continue;
                const uint8_t *top = sh.strip[r] + strip_off;
                const uint8_t *tile = sh.tile[r][S & (TILE_SLOTS - 1)] + tile_off;
                uint32_t v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nd)
                        v[j] = from_top ? *reinterpret_cast<const uint32_t *>(top + li * rw + ((x0 - 4 + 4 * j) & (rw - 1)))
                                        : reinterpret_cast<const uint32_t *>(tile + (li - 4) * tstride)[j];
                uint8_t *g = P.p + (ptrdiff_t)yy * P.stride + x0 - 4;
                if (from_top && r == 0 && band > 0) {
                    // these four pixel rows share cache lines with the previous band's strip hand-off:
                    // every access to them inside this launch is sc1
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < nd) st_sc1(reinterpret_cast<uint32_t *>(g) + j, v[j]);
                } else if (pl == 0) {
                    __builtin_memcpy(g, v, 16);
                } else {
                    __builtin_memcpy(g, v, 8);
                }
            }
            lds_fence();   // our LDS reads of this step are complete before the slot is released
            if (lane == 0) sh.wr_done = S + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------
    // worker waves
    // ---------------------------------------------------------------------------------------------
    const int half = lane >> 5, l32 = lane & 31;
    const int r = 2 * wave + half;              // local MB row
    const int gr = band_row0 + r;               // global MB row (gr == mbh: virtual row that only flushes)
    const bool row_real = gr < mbh, row_any = gr <= mbh;
    const int pl = l32 < 16 ? 0 : (l32 < 24 ? 1 : 2);
    const int li = pl == 0 ? l32 : (pl == 1 ? l32 - 16 : l32 - 24);
    const int msz = pl == 0 ? 16 : 8, nd = msz / 4;
    const Plane &P = pl == 0 ? a.Y : (pl == 1 ? a.U : a.V);
    const int rw = pl == 0 ? RWY : RWC;
    const int tstride = pl == 0 ? TILE_YS : TILE_CS;
    const int strip_off = pl == 0 ? 0 : (pl == 1 ? 4 * RWY : 4 * RWY + 4 * RWC);
    const int tile_off = pl == 0 ? 0 : (pl == 1 ? 16 * TILE_YS : 16 * TILE_YS + 8 * TILE_CS);
    uint8_t *top = sh.strip[r] + strip_off;         // 4 rows x rw: bottom of the row above
    uint8_t *bot = sh.strip[r + 1] + strip_off;     // 4 rows x rw: our own bottom rows
    const int y0 = gr * msz;
    const bool has_top = gr > 0;
    const bool publishes = band + 1 < a.nbands;   // a next band exists: every row of this band is real
    const int first_lf0 = sh.first_lf0;
    // segment parameters packed per segment: int_lim | mb_lim<<8 | b_lim<<16 | hev<<24 (all < 256)
    uint32_t sdp[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int32_t *sd = a.sd->v + s * SD_INTS;
        sdp[s] = (uint32_t)(sd[SD_INTERIOR_LIMIT] & 0xff) | ((uint32_t)(sd[SD_MBEDGE_LIMIT] & 0xff) << 8) |
                 ((uint32_t)(sd[SD_SUB_BEDGE_LIMIT] & 0xff) << 16) | ((uint32_t)(sd[SD_HEV_THRESHOLD] & 0xff) << 24);
    }

    // prefetch of macroblock 0
    uint4 nxt = make_uint4(0, 0, 0, 0);
    int nxt_seg = 0, nxt_mask = 0;
    if (row_real) {
        const uint8_t *g = P.p + (ptrdiff_t)(y0 + li) * P.stride;
        if (pl == 0) nxt = *reinterpret_cast<const uint4 *>(g);
        else { const uint2 t2 = *reinterpret_cast<const uint2 *>(g); nxt.x = t2.x; nxt.y = t2.y; }
        nxt_seg = a.o.seg[gr * mbw];
        nxt_mask = a.o.mask[gr * mbw];
    }
    uint32_t left4 = 0;
    const int steps = mbw + 1 + 2 * (ROWS - 1);
#ifdef LF2_STAMPS
    unsigned long long st_wait = 0, st_p1 = 0, st_p2 = 0, st_wb = 0, st_t0, st_t1;
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#else
#define STAMP(v)
#endif
    for (int S = 0; S < steps; ++S) {
        STAMP(st_t0);
        // ---- dependencies --------------------------------------------------------------------
        if (wave > 0) while (sh.prog[wave - 1] < S) __builtin_amdgcn_s_sleep(1);                  // data from above
        // ring space below: our second row may not lap the row under it, nor the writer wave that still
        // reads that row's strip up to TILE_SLOTS-1 steps later (derivation in DESIGN.md, loop filter)
        if (wave + 1 < WORKERS) while (sh.prog[wave + 1] < S + 10 - RING_MB) __builtin_amdgcn_s_sleep(1);
        // last wave: its second row's strip is drained by the publisher wave (x2 = S - 2*(ROWS-1))
        if (wave + 1 == WORKERS && publishes) while (sh.pub_done < S - 2 * (ROWS - 1) - (RING_MB - 2)) __builtin_amdgcn_s_sleep(1);
        while (sh.wr_done < S - (TILE_SLOTS - 1)) __builtin_amdgcn_s_sleep(1);   // the writer has drained tile slot S % TILE_SLOTS
        uint8_t *tile = sh.tile[r][S & (TILE_SLOTS - 1)] + tile_off;
        const int x = S - 2 * r;
        const bool act = row_any && x >= 0 && x <= mbw;
        const bool mbstep = act && row_real && x < mbw;    // a real macroblock (else: flush column / flush row)
        if (wave == 0 && band > 0 && act && half == 0) {   // r == 0 (a real row or the virtual flush row)
            const int need = imin(x + 1, mbw);
            while (sh.top_ready < need) __builtin_amdgcn_s_sleep(1);
        }
#ifdef LF2_STAMPS
        STAMP(st_t1); st_wait += st_t1 - st_t0; st_t0 = st_t1;
#endif
        const int x0 = x * msz;
        const uint4 own = nxt;
        const int seg = nxt_seg, maskv = nxt_mask;
        if (mbstep && x + 1 < mbw) {   // prefetch the next macroblock of this row
            const uint8_t *g = P.p + (ptrdiff_t)(y0 + li) * P.stride + x0 + msz;
            if (pl == 0) nxt = *reinterpret_cast<const uint4 *>(g);
            else { const uint2 t2 = *reinterpret_cast<const uint2 *>(g); nxt.x = t2.x; nxt.y = t2.y; }
            nxt_seg = a.o.seg[gr * mbw + x + 1];
            nxt_mask = a.o.mask[gr * mbw + x + 1];
        }
        const uint32_t sp = seg == 0 ? sdp[0] : (seg == 1 ? sdp[1] : (seg == 2 ? sdp[2] : sdp[3]));
        const int int_lim = sp & 0xff, mb_lim = (sp >> 8) & 0xff, b_lim = (sp >> 16) & 0xff, hev_thr = sp >> 24;
        const bool do_filter = mbstep && (gr * mbw + x) < first_lf0;
        const bool inner = maskv != 0;
        uint32_t *trow = reinterpret_cast<uint32_t *>(tile + li * tstride);
        // ---- phase 1: vertical edges, lane = pixel row, in registers ----------------------------
        if (mbstep) {
            int t[20];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = ub(left4, k);
#pragma unroll
            for (int k = 0; k < 16; ++k) t[4 + k] = ub(k < 4 ? own.x : (k < 8 ? own.y : (k < 12 ? own.z : own.w)), k & 3);
            if (do_filter) filter_line(t, msz, x > 0, inner, mb_lim, b_lim, int_lim, hev_thr);
#pragma unroll
            for (int j = 0; j < 5; ++j)
                if (j <= nd) trow[j] = pack4(t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3]);
        } else if (act && row_real) {
            trow[0] = left4;   // flush column: only the carried four columns are meaningful
        }
        lds_fence();
#ifdef LF2_STAMPS
        STAMP(st_t1); st_p1 += st_t1 - st_t0; st_t0 = st_t1;
#endif
        // ---- phase 2: horizontal edges, lane = pixel column -------------------------------------
        if (mbstep && do_filter) {
            int t[20];
            const int rc = (x0 + li) & (rw - 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = (int)top[k * rw + rc] + BIAS;
#pragma unroll
            for (int k = 0; k < 16; ++k) t[4 + k] = k < msz ? (int)tile[k * tstride + 4 + li] + BIAS : BIAS;
            filter_line(t, msz, has_top, inner, mb_lim, b_lim, int_lim, hev_thr);
            if (has_top) {
                top[1 * rw + rc] = (uint8_t)sat8(t[1] - BIAS); top[2 * rw + rc] = (uint8_t)sat8(t[2] - BIAS); top[3 * rw + rc] = (uint8_t)sat8(t[3] - BIAS);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < msz) tile[k * tstride + 4 + li] = (uint8_t)sat8(t[4 + k] - BIAS);
        }
        lds_fence();
#ifdef LF2_STAMPS
        STAMP(st_t1); st_p2 += st_t1 - st_t0; st_t0 = st_t1;
#endif
        // ---- write-back -----------------------------------------------------------------------
        if (act) {
            if (row_real) {
                // our bottom four rows -> strip of the row below (columns x0-4 .. x0+msz-1)
                if (li >= msz - 4) {
                    const int br = li - (msz - 4);
#pragma unroll
                    for (int j = 0; j < 5; ++j)
                        if (j <= nd) {
                            const uint32_t v = trow[j];
                            *reinterpret_cast<uint32_t *>(bot + br * rw + ((x0 - 4 + 4 * j) & (rw - 1))) = v;
                        }
                }
                left4 = trow[nd];   // columns msz-4 .. msz-1 of this macroblock, as filtered so far
            }
        }
        lds_fence();
        if (lane == 0) sh.prog[wave] = S + 1;
#ifdef LF2_STAMPS
        STAMP(st_t1); st_wb += st_t1 - st_t0;
#endif
    }
#ifdef LF2_STAMPS
    if (lane == 0 && band < 4) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.gprog + 1024) + (band * WORKERS + wave) * 4;
        o[0] = st_wait; o[1] = st_p1; o[2] = st_p2; o[3] = st_wb;
    }
#endif
}

}  // namespace lf2

void launch_loop_filter2(hipStream_t s, const Frame &recon, const MBOut &o, const SegData *d_sd, int32_t *progress,
                         int mbw, int mbh, unsigned launch_no) {
    lf2::Args a;
    a.Y = recon.Y[0];
    a.U = recon.U;
    a.V = recon.V;
    a.o = o;
    a.sd = d_sd;
    a.gprog = progress;
    a.mbw = mbw;
    a.mbh = mbh;
    a.nbands = (mbh + 1 + lf2::ROWS - 1) / lf2::ROWS;   // + the virtual flush row
    // band counters are never reset: every launch counts inside its own window (wraps after ~2^31/(mbw+2)
    // launches; the host zeroes the buffer when the window index wraps)
    const unsigned window = 0x7fffffffu / (unsigned)(mbw + 2) - 1;
    const unsigned n = launch_no % window;
    if (n == 0) hipMemsetAsync(progress, 0, sizeof(int32_t) * (a.nbands + 1), s);
    a.gbase = (int)(n * (unsigned)(mbw + 2));
    hipLaunchKernelGGL(lf2::k_loop_filter2, dim3(a.nbands), dim3(lf2::NWAVES * 64), 0, s, a);
}

}  // namespace vp8
