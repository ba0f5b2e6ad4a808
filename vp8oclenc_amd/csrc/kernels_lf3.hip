// kernels_lf3.hip -- VP8 normal loop filter, banded wavefront in LDS, one-step row lag (gfx950).
//
// Same arithmetic and ordering semantics as loop_filter_frame_luma/_chroma (CPU_kernels.cl:970-1075,
// :1333-1439; edge filters :829-926).  The filter is a chain of dependent edge filters: MB(x,y) needs
// MB(x-1,y) complete and, for its horizontal edges only, the vertical MB edge of MB(x+1,y-1).  A single wave
// issues one VALU instruction every ~5.5 cycles no matter what (scripts/ubench/valu_rates.hip), so the frame
// time is (steps on the critical path) x (instructions per step); both are what this version cuts:
//
//   * Each macroblock step has two phases: P1 = vertical edges (lane = pixel row, registers), P2 = horizontal
//     edges (lane = pixel column, through an LDS tile).  P2 of MB(x,y) needs only P1 of MB(x+1,y-1), so row y
//     runs ONE macroblock behind row y-1 (x = S - r at step S) with a hand-off in the middle of the step:
//     mb_w + mb_h steps per frame instead of mb_w + 2*mb_h.
//   * A workgroup owns a band of ROWS MB rows; a wave runs two rows (32 lanes each: 0-15 luma, 16-23 U,
//     24-31 V), so every second hand-off is inside a wave and costs nothing.
//   * Branch-free edge filters: edges that do not apply (frame border, chroma lanes, skipped inner edges,
//     the level-0 exit) run with their mask forced off instead of being jumped over, which removes the
//     divergent control flow (and its register shuffling) from the instruction stream.
//   * Samples carry +256 in registers: |a-b| is one v_sad_u16 even when an unsaturated carry (reference
//     quirk, :1024/:1062) dips below zero, and the saturated byte is just the low byte of a clamp.
//   * Bottom strips of each row live in an LDS ring.  A finished 16x16 block (shifted by (-4,-4)) is stored to
//     HBM by its own row at the top of the NEXT step, right behind the prefetch of the next macroblock, so the
//     stores have a full step to retire before the wave waits on vmcnt again (gfx9 counts loads and stores in
//     one counter).  The strip between bands goes through the frame (sc1 = write-through) with a loader and a
//     publisher wave per band; the loader also stores the rows it loaded once row 0 has filtered across them,
//     because a write-through store takes longer than a step to retire.
//   * One LDS poll per step (middle of the step) covers every dependency.
// History (1080p, one frame): v1 one wave per row through HBM 1.9 ms; v2 (git history) banded, two-step lag,
// writer wave 0.71 ms; this file 0.36 ms.
#include <stdlib.h>
#include <string.h>

#include "vp8hip_dev.h"

namespace vp8 {

namespace lf3 {

constexpr int WORKERS = 4;             // worker waves per band (one per SIMD)
constexpr int ROWS = 2 * WORKERS;      // MB rows per band
constexpr int RING_MB = 16;            // strip ring length in macroblocks
// One layout for all three planes (chroma simply uses half of it), so that every LDS access of the worker
// loop is base + immediate offset and nothing in it depends on the plane of the lane:
constexpr int SROW = RING_MB * 16;                     // strip row stride; ring width = RING_MB * msz pixels
constexpr int STRIP_PLANE = 4 * SROW;                  // four pixel rows per plane
constexpr int STRIP_BYTES = 3 * STRIP_PLANE;           // Y, U, V bottom strips of one MB row
constexpr int TILE_S = 24;                             // work-tile row stride: 4 carried columns + 16 + pad
constexpr int TILE_PLANE = 16 * TILE_S;
constexpr int TILE_BYTES = 3 * TILE_PLANE;
constexpr int TILE_SLOTS = 2;          // a finished tile is drained to HBM at the top of the next step
constexpr int BIAS = 256;

enum { F_TOP = WORKERS, F_PUB, F_ABORT = 7 };   // flag[0..WORKERS-1] = 2*step + phase of each worker

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
    w = c128(w + (e.q0 - e.p0) * 3);
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
    a = iclamp(a + (e.q0 - e.p0) * 3, -128, 123);   // the clamp to 127 and the two min(.., 127) >> 3 behind it in one (see filter_mb_edge)
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
// stores); the p/q registers handed from edge to edge stay unsaturated too (:1024, :1062).
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

// biased sample -> biased saturated sample; its low byte is the pixel (BIAS = 256)
__device__ __forceinline__ int satb(int v) { return iclamp(v, BIAS, BIAS + 255); }
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)satb(b), (uint32_t)satb(a), 0x0c0c0400u);
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)satb(d), (uint32_t)satb(c), 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
__device__ __forceinline__ int ub(uint32_t w, int k) { return byte_of(w, k) | BIAS; }

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int WAIT_LGKM0 = 0xc07f;   // s_waitcnt lgkmcnt(0) as the builtin's immediate: the compiler's own waitcnt pass sees it
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct Args {
    Plane Y, U, V;
    MBOut o;
    SegData *sd;      // read; written only by the verdict workgroup when check_SSIM's filter update applies (chk)
    LfCheck chk;
    int32_t *gprog;   // [bands] gbase + macroblocks of the band's bottom strip published so far
    int gbase;        // counters only grow: launch n uses the range (n*(mbw+2), (n+1)*(mbw+2)], so no memset
    int mbw, mbh, nbands;
    int32_t *err;     // set to 1 if a bounded wait expired (the host reports VP8HIP_ERR_TIMEOUT)
    int stall_test;   // test hook: publishers count from a wrong base, so every later band must time out
};

struct Shared {
    uint8_t strip[ROWS + 1][STRIP_BYTES];         // strip[r] = bottom rows of the MB row ABOVE local row r
    uint8_t tile[ROWS][TILE_SLOTS][TILE_BYTES];   // work tiles: this step's and the previous one's (being drained)
    int flag[8];                                  // worker progress, F_TOP, F_PUB; [F_ABORT]: a bounded wait expired somewhere in
                                                  // this workgroup, everybody leaves.  Read and written through `flag` below.
    uint32_t dummy[WORKERS * 64];                 // sink for stores of lanes that have nothing to store
    SegData sd;                                   // the segment data check_SSIM's filter update gives, when it applies (chk)
    float red[8];
    int repl;
    int first_lf0;                                // first macroblock whose segment has loop_filter_level 0 (:990)
    int4 lim[4];                                  // per segment: {interior limit, mb_delta, b_delta, hev threshold} (struct Limits)
};
// The flags are polled: the accesses must be volatile, and a volatile access through HIP's generic pointers stays a FLAT
// instruction (the address-space inference pass leaves volatile accesses alone) -- a flat load that resolves to LDS takes the
// vector-memory path, returns on vmcnt behind the wave's prefetch loads and block stores, and four of them one after the other
// were the 450 cycles of every step's poll.  Through an LDS-qualified pointer they are ds_read / ds_write on lgkmcnt.
typedef __attribute__((address_space(3))) volatile int lds_flag_t;

constexpr int NWAVES = WORKERS + 2;         // workers + loader + publisher

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
    }

// The workgroup behind the last band, present when check_SSIM rides in the launch: what check_SSIM reports (vp8enc.cpp:237-258:
// replaced count, the raster-order float sum / count, the minimum), the updated segment data back to where the entropy stage
// reads them, and the verdict to the host.  The sum must be the reference's -- one float accumulator over the macroblocks in
// raster order -- so the values are staged in LDS by all threads (the strips and tiles this workgroup has no other use for)
// and one thread adds them, four per ds_read_b128.
__device__ __forceinline__ void verdict_workgroup(const Args &a, Shared &sh, bool updated) {   // (inlined: a call would put the argument block into scratch memory)
    constexpr int NT = NWAVES * 64, CHUNK = 8192;
    static_assert(sizeof(sh.strip) + sizeof(sh.tile) >= CHUNK * sizeof(float), "staging area");
    float *s_val = reinterpret_cast<float *>(&sh.strip[0][0]);
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

__device__ __forceinline__ void loop_filter3_body(const Args &a) {
    __shared__ __attribute__((aligned(16))) Shared sh;
    const int band = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    lds_flag_t *const flag = (lds_flag_t *)sh.flag;
    if (threadIdx.x < 8) flag[threadIdx.x] = 0;
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
    // The kernel's own clock (constant 100 MHz): band 0 stamps the start, the wave that runs the frame's last row (the
    // virtual flush row) adds end - start to an accumulator the host reads with the profile (vp8hip_profile_read_clock).
    // hipEvents around a launch also count the time its packet waits for the queue when many streams share the part.
    unsigned long long *clk = reinterpret_cast<unsigned long long *>(a.err + 4);   // {start, sum of ticks, launches, sum of shader-clock cycles per tick x 1000, launches left out of that sum, launches whose last wave changed slots}
    if (band == 0 && threadIdx.x == 0) __hip_atomic_store(clk, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        const int msz = pl == 0 ? 16 : 8, rmask = RING_MB * msz - 1;
        const int y = (band_row0 + ROWS - 1) * msz + (msz - 4) + rr;
        const uint8_t *sp = sh.strip[ROWS] + pl * STRIP_PLANE + rr * SROW;
        for (int x = 0; x <= mbw; ++x) {
            const int done = 2 * (x + ROWS - 1) + 2;   // the last row has finished macroblock x
            LF_WAIT(flag[WORKERS - 1] < done, 3)
            if (lane < 44) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(sp + ((x * msz - 4 + 4 * j) & rmask));
                st_sc1(reinterpret_cast<uint32_t *>(P.p + (ptrdiff_t)y * P.stride + x * msz - 4) + j, v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                flag[F_PUB] = x + 1;
                __hip_atomic_store(&a.gprog[band], a.gbase + x + 1 - (a.stall_test ? (1 << 20) : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        const int msz = pl == 0 ? 16 : 8, rmask = RING_MB * msz - 1;
        const int y = band_row0 * msz - 4 + r;
        uint8_t *sp = sh.strip[0] + pl * STRIP_PLANE + r * SROW;
        uint8_t *gp = P.p + (ptrdiff_t)y * P.stride + 4 * j;
        // The four pixel rows above this band share cache lines with the previous band's hand-off, so every access
        // to them inside the launch is sc1 -- including their final store once row 0 has filtered across them.
        // That store is done here, not by worker 0: a write-through store takes longer than a step to retire and
        // would sit in front of every vmcnt wait of the worker.  Block m = columns m0-4 .. m0+msz-5, final when
        // row 0 has finished macroblock m (row 0: step == macroblock).
#define DRAIN_TOP(m)                                                                                        \
    {                                                                                                       \
        LF_WAIT(flag[0] < 2 * (m) + 2, 8)                                                                    \
        if (lane < 32) st_sc1(reinterpret_cast<uint32_t *>(gp + (m) * msz - 4),                             \
                              *reinterpret_cast<const uint32_t *>(sp + (((m) * msz - 4 + 4 * j) & rmask))); \
    }
        for (int x = 0; x < mbw; ++x) {
            // columns x0+13..15 are final once the previous band's last row has run P1 of macroblock x+1
            const int need = imin(x + 2, mbw + 1);
            LF_WAIT(__hip_atomic_load(&a.gprog[band - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.gbase < need, 2)
            // ring space: the slot still holds macroblock x-RING_MB, whose last four columns belong to the block
            // of macroblock x-RING_MB+1
            if (x >= RING_MB - 1) DRAIN_TOP(x - (RING_MB - 1))
            if (lane < 32) *reinterpret_cast<uint32_t *>(sp + ((x * msz + 4 * j) & rmask)) = ld_sc1(reinterpret_cast<const uint32_t *>(gp + x * msz));
            lds_fence();
            if (lane == 0) flag[F_TOP] = x + 1;
        }
        for (int m = imax(mbw - (RING_MB - 1), 0); m <= mbw; ++m) DRAIN_TOP(m)
        return;
    }

    // ---------------------------------------------------------------------------------------------
    // worker waves
    // ---------------------------------------------------------------------------------------------
    // the loader and publisher waves share SIMDs with workers 0 and 1: let them issue only in idle slots
    __builtin_amdgcn_s_setprio(3);
    const int half = lane >> 5, l32 = lane & 31;
    const int r = 2 * wave + half;              // local MB row
    const int gr = band_row0 + r;               // global MB row (gr == mbh: virtual row that only flushes)
    const bool row_real = gr < mbh, row_any = gr <= mbh;
    const int pl = l32 < 16 ? 0 : (l32 < 24 ? 1 : 2);
    const int li = pl == 0 ? l32 : (pl == 1 ? l32 - 16 : l32 - 24);
    const int msz = pl == 0 ? 16 : 8, nd = msz / 4;
    const Plane &P = pl == 0 ? a.Y : (pl == 1 ? a.U : a.V);
    const int rmask = RING_MB * msz - 1;
    uint8_t *top = sh.strip[r] + pl * STRIP_PLANE;       // 4 rows: bottom of the row above
    uint8_t *bot = sh.strip[r + 1] + pl * STRIP_PLANE;   // 4 rows: our own bottom rows
    // P1 hands columns x0-4..x0-1 of the bottom four pixel rows to the row below; the other lanes aim the
    // same store at a private dummy word instead of branching around it
    const bool bottom_lane = li >= msz - 4;
    uint8_t *botw = bottom_lane ? bot + (li - (msz - 4)) * SROW : reinterpret_cast<uint8_t *>(&sh.dummy[lane]);
    const int botw_mask = bottom_lane ? rmask : 0;
    const int tile_lane = pl * TILE_PLANE + li * TILE_S;    // this lane's row of the tile (P1)
    const int tile_col = pl * TILE_PLANE + 4 + li;          // this lane's column of the tile (P2)
    // Drain: the block that became final in a step -- 16x16 (8x8) shifted by (-4,-4) = four pixel rows of the
    // strip above (lanes li < 4) + msz-4 rows of the tile (lanes li >= 4) -- is stored at the top of the NEXT
    // step, right behind the prefetch, so the stores have a whole step to retire before anything waits on vmcnt.
    const bool from_top = li < 4;
    const bool drain_lane = row_any && (from_top ? gr > 0 && !(r == 0 && band > 0) : row_real);   // (the loader stores those)
    const uint8_t *dr_src = from_top ? top + li * SROW : sh.tile[r][0] + pl * TILE_PLANE + (li - 4) * TILE_S;
    const int dr_slot = from_top ? 0 : TILE_BYTES;          // tile lanes alternate between the two slots
    const int dr_and = from_top ? rmask : 0xffff;            // strip lanes wrap around the ring
    const int dr_col = from_top ? -1 : 0;                    // ... and start at column x0-4
    uint8_t *dr_g = P.p + (ptrdiff_t)(gr * msz - 4 + li) * P.stride - 4;
    const bool has_top = gr > 0;
    const bool publishes = band + 1 < a.nbands;   // a next band exists: every row of this band is real
    const int first_lf0 = sh.first_lf0;
    // Prefetch of macroblock 0.  Every lane loads 16 bytes (chroma lanes use 8 of them; at the right frame edge
    // the rest is margin).  The loads stay inside a branch on purpose: hoisted to the top of the loop body, hipcc
    // parks an s_waitcnt vmcnt(0) right behind them (measured: +700 cycles per step).
    const uint8_t *pf_p = P.p + (ptrdiff_t)(imin(gr, mbh - 1) * msz + li) * P.stride;
    const int32_t *pf_seg = a.o.seg + imin(gr, mbh - 1) * mbw, *pf_mask = a.o.mask + imin(gr, mbh - 1) * mbw;
    uint4 nxt = make_uint4(0, 0, 0, 0);
    int nxt_seg = 0, nxt_mask = 0;
    if (row_real) {
        nxt = *reinterpret_cast<const uint4 *>(pf_p);
        nxt_seg = pf_seg[0];
        nxt_mask = pf_mask[0];
    }
    uint32_t left4 = 0;
    const int steps = mbw + ROWS + 1;   // + one step that only drains
#ifdef LF_STAMPS
    unsigned long long st_p1 = 0, st_wait = 0, st_p2 = 0, st_wb = 0, st_t0, st_t1;
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#else
#define STAMP(v)
#endif
    for (int S = 0; S < steps; ++S) {
        STAMP(st_t0);
        uint8_t *tile = sh.tile[r][S & 1];
        const int x = S - r;
        // (`&`, not `&&`: one predicate, one exec mask -- short-circuit evaluation nests the regions)
        const bool p1_on = row_real & (x >= 0) & (x <= mbw);   // a real macroblock or the flush column behind the last one
        const bool mbstep = p1_on & (x < mbw);
        const int x0 = x * msz;
        // the prefetched macroblock is unpacked HERE, before the next prefetch is issued into the same registers: taking a
        // copy of the sixteen bytes + segment + mask instead cost nine moves per step
        int t[20];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[4 + k] = ub(k < 4 ? nxt.x : (k < 8 ? nxt.y : (k < 12 ? nxt.z : nxt.w)), k & 3);
        const int seg = nxt_seg, maskv = nxt_mask;
        // (Under a predicate on purpose.  Unconditional loads from a clamped position would save the moves that keep the old
        // registers alive for the lanes that do not load, but hipcc then waits for the loads it has just issued -- s_waitcnt
        // vmcnt(4) and vmcnt(3) a few instructions further down: +9 % on the whole kernel.)
        if (mbstep & (x + 1 < mbw)) {   // prefetch the next macroblock of this row
            nxt = *reinterpret_cast<const uint4 *>(pf_p + x0 + msz);
            nxt_seg = pf_seg[x + 1];
            nxt_mask = pf_mask[x + 1];
        }
        if (drain_lane & (x >= 1) & (x <= mbw + 1)) {   // the block of macroblock x-1 (or the flush column)
            const int c0 = ((x - 1) * msz - 4) & dr_col;
            const uint8_t *src = dr_src + ((S - 1) & 1) * dr_slot;
            uint32_t v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const uint32_t *>(src + ((c0 + 4 * j) & dr_and));
            uint32_t *g = reinterpret_cast<uint32_t *>(dr_g + (x - 1) * msz);
            *reinterpret_cast<uint2 *>(g) = make_uint2(v[0], v[1]);
            if (pl == 0) *reinterpret_cast<uint2 *>(g + 2) = make_uint2(v[2], v[3]);
        }
        const int4 lim = sh.lim[seg & 3];
        Limits L;
        // an edge that does not apply gets interior limit -1: its mask can never be true
        const int int_lim = lim.x;
        L.mb_delta = lim.y;
        L.b_delta = lim.z;
        L.hev_thr = lim.w;
        const bool do_filter = mbstep & ((gr * mbw + x) < first_lf0);
        const bool en_in = do_filter & (maskv != 0);
        const int il4 = en_in ? int_lim : -1;
        const int il8 = en_in && pl == 0 ? int_lim : -1;
        uint32_t *trow = reinterpret_cast<uint32_t *>(tile + tile_lane);
        // ---- P1: vertical edges, lane = pixel row, in registers ---------------------------------
        // The flush column (x == mbw) takes the same path with every edge off: the filters are then the identity and the
        // carried four columns land in the tile's first dword unchanged; the rest of its tile row is margin.
        if (p1_on) {
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = ub(left4, k);
            filter_line(t, L, do_filter & (x > 0) ? int_lim : -1, il4, il8);
            const uint32_t d0 = pack4(t[0], t[1], t[2], t[3]);
            trow[0] = d0;
#pragma unroll
            for (int j = 1; j < 5; ++j) trow[j] = pack4(t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3]);
            // the row below reads columns x0-4..x0-1 of our bottom rows in P2 of this very step
            *reinterpret_cast<uint32_t *>(botw + ((x0 - 4) & botw_mask)) = d0;
        }
        __builtin_amdgcn_s_waitcnt(WAIT_LGKM0);
        flag[wave] = 2 * S + 1;   // (every lane, the same word: no exec mask to set up and restore)
        STAMP(st_t1);
#ifdef LF_STAMPS
        st_p1 += st_t1 - st_t0; st_t0 = st_t1;
#endif
        // ---- the one poll of the step ------------------------------------------------------------
        {
            const int need_up = 2 * S + 1;                       // P1 of the rows above (their macroblock x+1)
            // ring space below: our second row is about to overwrite, in its bottom strip, the slot of macroblock
            // x-RING_MB, whose last columns the wave below stores at the top of its step S-(RING_MB-3)
            const int need_dn = wave + 1 < WORKERS ? 2 * (S - (RING_MB - 3)) + 1 : 0;
            const int x_r0 = S - 2 * wave;                        // macroblock of this wave's first row
            const bool top_dep = wave == 0 && band > 0 && x_r0 >= 0 && x_r0 <= mbw && band_row0 <= mbh;
            const int need_top = imin(x_r0 + 1, mbw);
            // last wave: the publisher must have drained what the second row is about to overwrite in strip[ROWS]
            const int need_pub = (wave + 1 == WORKERS && publishes) ? S - (ROWS - 1) - (RING_MB - 2) : 0;
            // Everything the poll compares is the same in all lanes; readfirstlane says so to the compiler, which otherwise
            // builds the loop out of exec-mask bookkeeping (a third of the poll's instructions on the path of every step).
            const int up = imax(wave - 1, 0), dn = imin(wave + 1, WORKERS - 1);
            for (int spins = 0;; ++spins) {
                // (unconditional loads: five ds_read_b32 in flight at once)
                const int f_up = flag[up], f_dn = flag[dn], f_top = flag[F_TOP], f_pub = flag[F_PUB], f_abort = flag[F_ABORT];
                const bool ok = (wave == 0 || f_up >= need_up) && (wave + 1 == WORKERS || f_dn >= need_dn) && (!top_dep || f_top >= need_top) &&
                                f_pub >= need_pub;
                const int state = __builtin_amdgcn_readfirstlane(f_abort ? 2 : (ok ? 1 : 0));
                if (state == 1) break;
                if (state == 2) return;
                if (spins > SPIN_LIMIT) { flag[F_ABORT] = 1; *a.err = 1; }
                if (spins < 32) asm volatile("s_nop 3"); else __builtin_amdgcn_s_sleep(1);   // the flag is usually a few hundred cycles away: a tight poll first, naps when it is not
            }
        }
        STAMP(st_t1);
#ifdef LF_STAMPS
        st_wait += st_t1 - st_t0; st_t0 = st_t1;
#endif
        // ---- P2: horizontal edges, lane = pixel column ---------------------------------------------
        if (mbstep) {
            int t[20];
            const int rc = (x0 + li) & rmask;
            uint8_t *tp = top + rc, *bp = bot + rc, *tc = tile + tile_col;
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = (int)tp[k * SROW];
#pragma unroll
            for (int k = 0; k < 16; ++k) t[4 + k] = (int)tc[k * TILE_S];   // chroma lanes: rows 8-15 are don't-care
            __builtin_amdgcn_s_waitcnt(WAIT_LGKM0);   // one wait for the twenty loads instead of one per use
#pragma unroll
            for (int k = 0; k < 20; ++k) t[k] |= BIAS;
            filter_line(t, L, do_filter & has_top ? int_lim : -1, il4, il8);
            // rows 1-3 of the strip above (row 0 of the frame: a scratch strip nobody reads)
            tp[1 * SROW] = (uint8_t)satb(t[1]); tp[2 * SROW] = (uint8_t)satb(t[2]); tp[3 * SROW] = (uint8_t)satb(t[3]);
            int s[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { s[k] = satb(t[4 + k]); tc[k * TILE_S] = (uint8_t)s[k]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) bp[j * SROW] = (uint8_t)(pl == 0 ? s[12 + j] : s[4 + j]);   // our bottom rows -> row below
            left4 = trow[nd];   // columns msz-4 .. msz-1 of this macroblock after both phases (next P1's left side)
        }
        __builtin_amdgcn_s_waitcnt(WAIT_LGKM0);
        flag[wave] = 2 * S + 2;
        STAMP(st_t1);
#ifdef LF_STAMPS
        st_p2 += st_t1 - st_t0;
#endif
    }
#ifdef LF_STAMPS
    if (lane == 0 && band < 4) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.gprog + 1024) + (band * WORKERS + wave) * 4;
        o[0] = st_wait; o[1] = st_p1; o[2] = st_p2; o[3] = st_wb;
    }
#endif
    if (gr == mbh && l32 == 0) {   // the frame's last row: this wave is the last to finish real work
        const unsigned long long t0 = __hip_atomic_load(clk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(clk + 1, t1 - t0);
        atomicAdd(clk + 2, 1ull);
        // the shader clock this wave saw while it ran: s_memtime cycles per 100 MHz tick (MI355X_MICROARCH.md, DVFS (6))
        const unsigned long long ratio = (__builtin_amdgcn_s_memtime() - cyc0) * 1000ull / (t1 - tick0 + 1);
        const uint32_t hwid1 = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        // a wave that was context-switched (the hardware scheduler rotating an oversubscribed set of queues) comes back on
        // another slot, whose cycle counter is another one: such launches are counted, not averaged
        if (ratio > 100000ull) atomicAdd(clk + 4, 1ull); else atomicAdd(clk + 3, ratio);
        if (hwid1 != hwid0) atomicAdd(clk + 5, 1ull);
    }
}

__global__ __launch_bounds__(NWAVES * 64) void k_loop_filter3(Args a) { loop_filter3_body(a); }
static_assert(sizeof(BatchOf<Args>) <= 4096, "a batch's argument blocks travel in the 4 KiB kernel-argument segment");
__global__ __launch_bounds__(NWAVES * 64) void k_loop_filter3_b(BatchOf<Args> b) { loop_filter3_body(b.item[blockIdx.z]); }

}  // namespace lf3

static lf3::Args loop_filter3_args(hipStream_t s, const Frame &recon, const MBOut &o, SegData *d_sd, int32_t *progress, int mbw, int mbh,
                                   unsigned launch_no, int stall_test, const LfCheck *chk) {
    lf3::Args a;
    if (chk) a.chk = *chk;
    else a.chk.on = 0;
    a.Y = recon.Y[0];
    a.U = recon.U;
    a.V = recon.V;
    a.o = o;
    a.sd = d_sd;
    a.gprog = progress;
    a.mbw = mbw;
    a.mbh = mbh;
    a.nbands = (mbh + 1 + lf3::ROWS - 1) / lf3::ROWS;   // + the virtual flush row
    // band counters are never reset: every launch counts inside its own window (wraps after ~2^31/(mbw+2)
    // launches; the host zeroes the buffer when the window index wraps)
    const unsigned window = 0x7fffffffu / (unsigned)(mbw + 2) - 1;
    const unsigned n = launch_no % window;
    if (n == 0) hipMemsetAsync(progress, 0, sizeof(int32_t) * (a.nbands + 1), s);
    a.gbase = (int)(n * (unsigned)(mbw + 2));
    a.err = progress + LF_ERR_WORD;
    a.stall_test = stall_test;
    return a;
}
static bool lf_skip() {
    static const bool skip = experiment_skip("lf");
    return skip;   // timing experiment only
}

void launch_loop_filter3(hipStream_t s, const Frame &recon, const MBOut &o, SegData *d_sd, int32_t *progress,
                         int mbw, int mbh, unsigned launch_no, int stall_test, const LfCheck *chk) {
    const lf3::Args a = loop_filter3_args(s, recon, o, d_sd, progress, mbw, mbh, launch_no, stall_test, chk);
    if (lf_skip()) return;
    VP8_LAUNCH(lf3::k_loop_filter3, dim3(a.nbands + (a.chk.on ? 1 : 0)), dim3(lf3::NWAVES * 64), 0, s, a);   // + the verdict workgroup
}

void launch_loop_filter3_batch(hipStream_t s, const Frame *const *recon, const MBOut *const *o, SegData *const *d_sd,
                               int32_t *const *progress, int mbw, int mbh, const unsigned *launch_no, int n, const LfCheck *chk) {
    BatchOf<lf3::Args> b;
    b.n = n;
    bool any = false;
    for (int i = 0; i < n; ++i) {
        b.item[i] = loop_filter3_args(s, *recon[i], *o[i], d_sd[i], progress[i], mbw, mbh, launch_no[i], 0, chk ? &chk[i] : nullptr);
        any = any || b.item[i].chk.on;
    }
    if (lf_skip()) return;
    VP8_LAUNCH(lf3::k_loop_filter3_b, dim3(b.item[0].nbands + (any ? 1 : 0), 1, n), dim3(lf3::NWAVES * 64), 0, s, b);
}

}  // namespace vp8
