// kernels_me.hip -- motion estimation side of the inter-frame path for gfx950:
//   packing of a new frame into its padded surface, edge replication, pyramid, hierarchical full-pel search
//   (the quarter-pel search is kernels_s2.hip, the reference choice is inside k_mb).
// Behaviour follows the reference kernels cited at each kernel (paths under the reference's src/); the parallel
// decomposition is new (k_search1: a lane per candidate row of an 8x8 block walking five candidates over
// transposed columns, the winner by a shuffle minimum of packed (cost, index) keys).  Every kernel has a single
// and a batched form (blockIdx.z = context, vp8hip_dev.h).
#include <stdlib.h>
#include <string.h>

#include "vp8hip_dev.h"

namespace vp8 {

// ------------------------------------------------------------------------------------------------
// Edge replication of a reference frame's Y/U/V planes (replaces the clamp-to-edge image sampler,
// GPU_kernels.cl:562).  grid = (max rows + 2*EXT, 3 planes), block = 64.
// ------------------------------------------------------------------------------------------------
struct BorderItem { Plane y, u, v; };
// one row (-EXT .. h + EXT - 1) of one plane by the 64 lanes of a wave.  Rows start 32-byte aligned and plane widths are multiples
// of 8 (vp8hip_create: frame sizes are multiples of 16), so a row above or below the plane is copied eight bytes per lane and step
// and either margin is ONE eight-byte store of the edge sample (byte by byte a full 1080p row was 30 dependent rounds per lane, and
// the 48 such rows of a frame made the pyramid-and-edges launch twice as long as the pyramid alone).
static_assert(EXT == 8, "border_row stores a margin as one eight-byte word");
__device__ __forceinline__ void border_row(const Plane &pl, int row, int lane) {
    if (row >= pl.h + EXT) return;
    uint8_t *dst = pl.p + (ptrdiff_t)row * pl.stride;
    const uint8_t *src = pl.p + (ptrdiff_t)iclamp(row, 0, pl.h - 1) * pl.stride;
    if (row < 0 || row >= pl.h)
        for (int x = lane * 8; x < pl.w; x += 512) *reinterpret_cast<uint2 *>(dst + x) = *reinterpret_cast<const uint2 *>(src + x);
    if (lane < 2) {
        const uint32_t s4 = (uint32_t)src[lane ? pl.w - 1 : 0] * 0x01010101u;
        *reinterpret_cast<uint2 *>(dst + (lane ? pl.w : -EXT)) = make_uint2(s4, s4);
    }
}
__device__ __forceinline__ void border_body(const BorderItem &a) {
    border_row(blockIdx.y == 0 ? a.y : (blockIdx.y == 1 ? a.u : a.v), (int)blockIdx.x - EXT, threadIdx.x);
}

__global__ __launch_bounds__(64) void k_border(BorderItem a) { border_body(a); }

void launch_border(hipStream_t s, const Frame &f) {
    dim3 grid(f.Y[0].h + 2 * EXT, 3);
    VP8_LAUNCH(k_border, grid, dim3(64), 0, s, BorderItem{f.Y[0], f.U, f.V});
}

// ------------------------------------------------------------------------------------------------
// downsample_x2, GPU_kernels.cl:429-451: dst = (a+b+c+d+2)/4 of each 2x2.
// The whole pyramid in one launch (replaces 4 x downsample_x2 per surface, inter_part.h:11-33).
// A workgroup takes a 64x64 tile of the full-resolution plane and produces the 32x32, 16x16, 8x8 and
// 4x4 tiles below it; every level is computed from the ROUNDED level above it, exactly like the
// reference's cascade of launches.  grid = (ceil(W/64), ceil(H/64), surfaces), block = 256.
// ------------------------------------------------------------------------------------------------
// A surface that has just come out of the loop filter also needs its replicated edges before it serves as a reference; the
// pyramid reads the interior only, so the two do not depend on each other and share the launch (bit z of border_mask: the rows
// of workgroups behind the tiles do surface z's edges, a wave per row of a plane) -- one link less in every frame's chain.
struct PyrArgs { Frame f[2 * MAX_BATCH]; uint32_t border_mask; };   // blockIdx.z picks the surface: one or two of a context, or those of a batch
static int border_jobs(const Frame &f) { return (f.Y[0].h + 2 * EXT) + 2 * (f.U.h + 2 * EXT); }

__global__ __launch_bounds__(256) void k_pyramid(PyrArgs a) {
    __shared__ uint8_t s2[16][16];
    __shared__ uint8_t s3[8][8];
    const Frame &f = a.f[blockIdx.z];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const Plane &P0 = f.Y[0];
    const int tiles_y = (P0.h + 63) / 64;
    // A tile is 64 bytes wide, a line of memory 128: horizontal neighbours share every line they read, and the rows they write in
    // the levels below are 32, 16, 8 and 4 bytes of one line.  Tiles are therefore handed out so that an XCD gets a run of
    // consecutive tiles (xcd_band, vp8hip_dev.h): the two halves of a line meet in ONE L2.
    const int tile = xcd_band((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, tiles_y * (int)gridDim.x);
    const int bx = tile % (int)gridDim.x, by = tile / (int)gridDim.x;
    if ((int)blockIdx.y >= tiles_y) {
        if (!((a.border_mask >> blockIdx.z) & 1)) return;
        int job = (((int)blockIdx.y - tiles_y) * (int)gridDim.x + (int)blockIdx.x) * 4 + (t >> 6);
        const int ny = P0.h + 2 * EXT, nc = f.U.h + 2 * EXT;
        if (job < ny) border_row(P0, job - EXT, t & 63);
        else if (job < ny + nc) border_row(f.U, job - ny - EXT, t & 63);
        else if (job < ny + 2 * nc) border_row(f.V, job - ny - nc - EXT, t & 63);
        return;
    }
    // 4x4 source pixels -> 2x2 of level 1 -> 1 of level 2
    const int sx = imin(bx * 64 + 4 * tx, P0.w - 4), sy0 = by * 64 + 4 * ty;
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = *reinterpret_cast<const uint32_t *>(P0.p + (ptrdiff_t)imin(sy0 + k, P0.h - 1) * P0.stride + sx);
    int l1[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            l1[j][i] = (byte_of(r[2 * j], 2 * i) + byte_of(r[2 * j], 2 * i + 1) + byte_of(r[2 * j + 1], 2 * i) +
                        byte_of(r[2 * j + 1], 2 * i + 1) + 2) >> 2;
    const Plane &P1 = f.Y[1];
    const int x1 = bx * 32 + 2 * tx, y1 = by * 32 + 2 * ty;
    if (x1 < P1.w) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (y1 + j < P1.h)
                *reinterpret_cast<uint16_t *>(P1.p + (ptrdiff_t)(y1 + j) * P1.stride + x1) = (uint16_t)(l1[j][0] | (l1[j][1] << 8));
    }
    const int l2 = (l1[0][0] + l1[0][1] + l1[1][0] + l1[1][1] + 2) >> 2;
    const Plane &P2 = f.Y[2];
    const int x2 = bx * 16 + tx, y2 = by * 16 + ty;
    if (x2 < P2.w && y2 < P2.h) P2.p[(ptrdiff_t)y2 * P2.stride + x2] = (uint8_t)l2;
    s2[ty][tx] = (uint8_t)l2;
    __syncthreads();
    if (t < 64) {
        const int ux = t & 7, uy = t >> 3;
        const int l3 = (s2[2 * uy][2 * ux] + s2[2 * uy][2 * ux + 1] + s2[2 * uy + 1][2 * ux] + s2[2 * uy + 1][2 * ux + 1] + 2) >> 2;
        const Plane &P3 = f.Y[3];
        const int x3 = bx * 8 + ux, y3 = by * 8 + uy;
        if (x3 < P3.w && y3 < P3.h) P3.p[(ptrdiff_t)y3 * P3.stride + x3] = (uint8_t)l3;
        s3[uy][ux] = (uint8_t)l3;
    }
    __syncthreads();
    if (t < 16) {
        const int ux = t & 3, uy = t >> 2;
        const int l4 = (s3[2 * uy][2 * ux] + s3[2 * uy][2 * ux + 1] + s3[2 * uy + 1][2 * ux] + s3[2 * uy + 1][2 * ux + 1] + 2) >> 2;
        const Plane &P4 = f.Y[4];
        const int x4 = bx * 4 + ux, y4 = by * 4 + uy;
        if (x4 < P4.w && y4 < P4.h) P4.p[(ptrdiff_t)y4 * P4.stride + x4] = (uint8_t)l4;
    }
}

static dim3 pyramid_grid(const Frame &f, int nframes, uint32_t border_mask) {
    const int gx = (f.Y[0].w + 63) / 64, gy = (f.Y[0].h + 63) / 64;
    const int extra = border_mask ? ((border_jobs(f) + 3) / 4 + gx - 1) / gx : 0;
    return dim3(gx, gy + extra, nframes);
}
void launch_pyramid(hipStream_t s, const Frame *a, const Frame *b, uint32_t border_mask) {
    PyrArgs p;
    p.f[0] = *a;
    p.f[1] = b ? *b : *a;
    p.border_mask = border_mask;
    VP8_LAUNCH(k_pyramid, pyramid_grid(*a, b ? 2 : 1, border_mask), dim3(256), 0, s, p);
}
void launch_pyramid_batch(hipStream_t s, const Frame *const *f, int nframes, uint32_t border_mask) {
    if (nframes <= 0) return;
    PyrArgs p;
    for (int i = 0; i < nframes; ++i) p.f[i] = *f[i];
    p.border_mask = border_mask;
    VP8_LAUNCH(k_pyramid, pyramid_grid(*f[0], nframes, border_mask), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------------
// Tight HBM-resident Y,U,V planes -> the padded surfaces of a frame, one launch.  A thread copies `per` units of 8 bytes,
// 256 units apart (a workgroup = per x 2 KB): with one unit per thread a 1080p frame is 1530 workgroups of next to no work,
// and under load it is the dispatch of workgroups, not the copying, that such a launch waits for (same box, headline with 1 / 8 /
// 16 / 32 units per thread: 62.0 / 62.4 / 62.5 / 62.5 M MB/s).
// ------------------------------------------------------------------------------------------------
// The source may be smaller than the coded ("wrk") size -- 1920x1080 in, 1920x1088 coded: copy_with_padding, encIO.h:141-196,
// happens here.  Rows below the source repeat its last row, samples to its right repeat the row's last sample (what the
// reference does for Y and U and means for V: its V lines read and write U, :180-183, so V's right padding is never
// written -- which bites only when the width is not a multiple of 16, none of BASELINE's configs; there this is the
// intended result, not the reference's undefined one; tests/test_padding.py shows both).  sw, sh: luma size of the source; ssy, ssc: its row strides.
struct PackItem { Plane py, pu, pv; const uint8_t *sy, *su, *sv; int sw, sh, ssy, ssc; };
__device__ __forceinline__ void pack_body(const PackItem &a, int per) {
    const Plane &py = a.py, &pu = a.pu, &pv = a.pv;
    const uint8_t *sy = a.sy, *su = a.su, *sv = a.sv;
    const int ny = (py.w >> 3) * py.h, nc = (pu.w >> 3) * pu.h;
    for (int k = 0; k < per; ++k) {
        int i = ((int)blockIdx.x * per + k) * 256 + (int)threadIdx.x;
        const Plane *pl = &py;
        const uint8_t *src = sy;
        int sw = a.sw, sh = a.sh, ss = a.ssy;
        if (i >= ny) {
            i -= ny;
            pl = &pu;
            src = su;
            sw >>= 1; sh >>= 1; ss = a.ssc;
            if (i >= nc) { i -= nc; pl = &pv; src = sv; }
            if (i >= nc) return;
        }
        const int upr = pl->w >> 3;     // 8-byte units per row
        const int y = i / upr, x = (i % upr) * 8;
        const uint8_t *row = src + (size_t)(y < sh ? y : sh - 1) * ss;
        uint2 v;
        if (x + 8 <= sw && ((ss | (int)(reinterpret_cast<uintptr_t>(src) & 7)) & 7) == 0) {
            v = *reinterpret_cast<const uint2 *>(row + x);
        } else {          // the unit hangs over the source's right edge, or the source rows are not 8-byte aligned
            uint32_t w[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j >> 2] |= (uint32_t)row[x + j < sw ? x + j : sw - 1] << (8 * (j & 3));
            v = make_uint2(w[0], w[1]);
        }
        *reinterpret_cast<uint2 *>(pl->p + (ptrdiff_t)y * pl->stride + x) = v;
    }
}

__global__ __launch_bounds__(256) void k_pack_b(BatchOf<PackItem> b, int per) { pack_body(b.item[blockIdx.z], per); }
static int pack_units_per_thread() {
    static const int per = [] { const char *e = getenv("VP8HIP_PACK_UNITS"); const int v = e ? atoi(e) : 16; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    return per;
}

static PackItem pack_item(const Frame &f, const void *y, const void *u, const void *v, int sw, int sh, int ssy, int ssc) {
    if (sw <= 0) { sw = f.Y[0].w; sh = f.Y[0].h; }
    if (ssy <= 0) { ssy = sw; ssc = sw >> 1; }
    return PackItem{f.Y[0], f.U, f.V, (const uint8_t *)y, (const uint8_t *)u, (const uint8_t *)v, sw, sh, ssy, ssc};
}
void launch_pack(hipStream_t s, const Frame &f, const void *y, const void *u, const void *v, int sw, int sh, int ssy, int ssc) {
    const int n = (f.Y[0].w >> 3) * f.Y[0].h + 2 * ((f.U.w >> 3) * f.U.h);
    // (a batch of one: the by-value form of the kernel picks the plane through a pointer into its argument block, which hipcc
    // answers by copying the block to scratch memory)
    BatchOf<PackItem> b;
    b.n = 1;
    b.item[0] = pack_item(f, y, u, v, sw, sh, ssy, ssc);
    const int per = pack_units_per_thread();
    VP8_LAUNCH(k_pack_b, dim3((n + 256 * per - 1) / (256 * per), 1, 1), dim3(256), 0, s, b, per);
}
void launch_pack_batch(hipStream_t s, const Frame *const *f, const void *const *y, const void *const *u, const void *const *v, int n,
                       int sw, int sh) {
    BatchOf<PackItem> b;
    b.n = n;
    for (int i = 0; i < n; ++i) b.item[i] = pack_item(*f[i], y[i], u[i], v[i], sw, sh, 0, 0);
    const int units = (f[0]->Y[0].w >> 3) * f[0]->Y[0].h + 2 * ((f[0]->U.w >> 3) * f[0]->U.h);
    static const bool skip = experiment_skip("pack");
    if (skip) return;   // timing experiment only
    const int per = pack_units_per_thread();
    VP8_LAUNCH(k_pack_b, dim3((units + 256 * per - 1) / (256 * per), 1, n), dim3(256), 0, s, b, per);
}

// ------------------------------------------------------------------------------------------------
// luma_search_1step, GPU_kernels.cl:459-560.  One 8x8 block of one pyramid level per 32 lanes,
// lane = candidate dxy (25 of 32 used).  Cost = sum of 4 weight4x4 (ushort, wraps) + MV penalty;
// out-of-frame candidates can never win (Diff|0x7fff >= initial MinDiff) and are skipped; the
// sequential "first strict minimum" is min over (cost<<8 | dxy).
// grid = (ceil(nblk/8), enabled refs), block = 256.
// ------------------------------------------------------------------------------------------------
struct Search1Args {
    Plane cur;
    Plane ref[3];
    const int16_t *src[3];
    int16_t *dst[3];
    int refmap[3];
    int nrefs;      // enabled references of this context (a batched launch is sized for the largest of its contexts)
    int net_width, w, h, pixel_rate, rate_shift, nblk, bw;
    int pbw, pbh;   // block grid of the coarser level (whose cells of src[] were written this frame)
    uint32_t bw_inv;   // ceil(2^32 / bw)
};

// Two mappings of the same work:
//   SPLIT = false: five lanes per 8x8 block, lane j = candidate row dy = j-2, a loop over the four 4x4 sub-blocks;
//                  12 blocks per wave.  Fewest instructions (the per-block preamble is paid once): the throughput form.
//   SPLIT = true:  twenty lanes per block = (sub-block sb) x (dy); the four sub-block costs of a candidate meet in two
//                  shuffles; 3 blocks per wave.  The same arithmetic in four times as many waves, each a quarter as long
//                  (the preamble is paid per sub-block: +29 % instructions): the latency form.  On the coarse levels,
//                  which have a few hundred blocks, a launch lasts as long as ONE wave: 12 us per level with the loop,
//                  5-6 us split (single video, frame after frame: 73 -> 45 us of a 540 us frame for the five levels).
// In either, a lane loads the 8-byte reference rows of a sub-block at its dy once and walks the five dx candidates over
// them in registers (the window bytes are shared by the five candidates of a row).
template <bool SPLIT>
struct S1Map {
    static constexpr int LANES_PER_BLOCK = SPLIT ? 20 : 5;
    static constexpr int BLOCKS_PER_WAVE = SPLIT ? 3 : 12;
    static constexpr int BLOCKS_PER_WG = 4 * BLOCKS_PER_WAVE;
};

// cost of the five dx candidates of one 4x4 sub-block (sx, sy) for this lane's dy row, added to acc[]
__device__ __forceinline__ void s1_subblock(const uint8_t *cp, int cstride, const uint8_t *rp, int rstride, int acc[5]) {
    uint32_t c[4], q0[4], q1[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        c[y] = *reinterpret_cast<const uint32_t *>(cp + (ptrdiff_t)y * cstride) ^ 0x80808080u;
        const uint2 q = ld_u64(rp + (ptrdiff_t)y * rstride);
        q0[y] = q.x ^ 0x80808080u; q1[y] = q.y ^ 0x80808080u;
    }
    // rows -> columns (byte r = row r), pixels biased: the form weight_cols_pre wants.  Candidate dx = i reads
    // columns i..i+3 of the eight: no byte alignment per candidate, and the current block's share of the metric
    // (16 dot4) is computed once for the five of them.
    uint32_t cc[4], col[8];
    transpose4x4(c, cc);
    transpose4x4(q0, col);
    transpose4x4(q1, col + 4);
    int pre[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) weight_pre_column(cc[k], pre + 4 * k);
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] += weight_cols_pre(pre, col + i);
}

// The same with the current block's share of the metric taken from LDS (pre16: the 16 ints weight_pre_column x 4 of this sub-block, made once
// per wave by s1_make_pre below): the five dy lanes of a block -- and every reference -- need the same 16 dot4, 8 permutes and 4 biases per
// sub-block; as instructions of the wave they cost what they cost one lane.
__device__ __forceinline__ void s1_subblock_pre(const int *pre16, const uint8_t *rp, int rstride, int acc[5]) {
    uint32_t q0[4], q1[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const uint2 q = ld_u64(rp + (ptrdiff_t)y * rstride);
        q0[y] = q.x ^ 0x80808080u; q1[y] = q.y ^ 0x80808080u;
    }
    uint32_t col[8];
    transpose4x4(q0, col);
    transpose4x4(q1, col + 4);
    int pre[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int4 v = *reinterpret_cast<const int4 *>(pre16 + 4 * k);
        pre[4 * k] = v.x; pre[4 * k + 1] = v.y; pre[4 * k + 2] = v.z; pre[4 * k + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] += weight_cols_pre(pre, col + i);
}
// ... made by the wave for its twelve blocks: lane = (block slot, sub-block), 48 of the 64 lanes; 16 ints each into pre_lds[slot][sub-block][16]
constexpr int S1_PRE_INTS = 12 * 4 * 16;
__device__ __forceinline__ void s1_make_pre(const Plane &cur, int cx, int cy, int sb, bool on, int *dst16) {
    const int sx = (sb >> 1) * 4, sy = (sb & 1) * 4;      // the order of the cost loop: (0,0), (0,+4 rows), (+4 cols,0), (+4,+4)
    const uint8_t *cp = cur.p + (ptrdiff_t)(cy + sy) * cur.stride + cx + sx;
    uint32_t c[4], cc[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) c[y] = *reinterpret_cast<const uint32_t *>(cp + (ptrdiff_t)y * cur.stride) ^ 0x80808080u;
    transpose4x4(c, cc);
    int pre[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) weight_pre_column(cc[k], pre + 4 * k);
    if (on) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<int4 *>(dst16 + 4 * k) = make_int4(pre[4 * k], pre[4 * k + 1], pre[4 * k + 2], pre[4 * k + 3]);
    }
}

// One 8x8 block of one level against one reference: the lanes of the block (M::LANES_PER_BLOCK of them: `sub` is the lane's number in
// the block, `lane` its number in the wave) search the 25 candidates around the scaled parent vector `pv` (packed short2, 0 = none).
// Returns the block's vector as the net holds it (packed short2, already multiplied by pixel_rate); valid in the lane with sub == 0.
// `live` false: the lanes take part in the shuffles on harmless in-frame data.
template <bool SPLIT, bool PRE_LDS = false>
__device__ __forceinline__ uint32_t search1_block(const Search1Args &a, int r, int cx, int cy, uint32_t pv, bool live, int sub, int lane,
                                                  const int *pre_lds = nullptr /* PRE_LDS (loop form): this block's [sub-block][16] from s1_make_pre */) {
    const int sb0 = sub / 5, j = sub - 5 * sb0;       // SPLIT: this lane's sub-block; otherwise sb0 = 0
    // vector / pixel_rate truncates toward zero (:495-500)
    int v0x = (int16_t)(pv & 0xffffu), v0y = (int16_t)(pv >> 16);
    const int rmask = a.pixel_rate - 1;
    v0x = (v0x + ((v0x >> 31) & rmask)) >> a.rate_shift;
    v0y = (v0y + ((v0y >> 31) & rmask)) >> a.rate_shift;
    if (a.pixel_rate > 8) v0x = v0y = 0;

    const int py = (int16_t)(cy + v0y + (j - 2));
    const bool row_valid = live && py >= 0 && py <= a.h - 8;
    const int xb = cx + v0x - 2;                       // x of candidate dx = -2
    // rows that cannot hold a valid candidate are read at a harmless in-frame position
    const bool loadable = row_valid && xb >= -8 && xb <= a.w - 4;
    const int lx = loadable ? xb : cx, ly = loadable ? py : cy;

    const uint8_t *cp = a.cur.p + (ptrdiff_t)cy * a.cur.stride + cx;
    const uint8_t *rp = a.ref[r].p + (ptrdiff_t)ly * a.ref[r].stride + lx;
    int acc[5] = {0, 0, 0, 0, 0};
    if (SPLIT) {
        const int sx = (sb0 >> 1) * 4, sy = (sb0 & 1) * 4;   // sub-block order (0,0),(0,+4 rows),(+4 cols,0),(+4,+4), :456-458
        s1_subblock(cp + (ptrdiff_t)sy * a.cur.stride + sx, a.cur.stride, rp + (ptrdiff_t)sy * a.ref[r].stride + sx, a.ref[r].stride, acc);
#pragma unroll
        for (int i = 0; i < 5; ++i) {   // lanes sub, sub+5, sub+10, sub+15 of the block: the sums land on sb0 = 0
            acc[i] += __shfl(acc[i], lane + 10, 64);
            acc[i] += __shfl(acc[i], lane + 5, 64);
        }
    } else {
        // one sub-block at a time (loop NOT unrolled: the register footprint decides how many waves a SIMD holds)
        if (PRE_LDS) {
#pragma unroll 1
            for (int sb = 0; sb < 4; ++sb) {
                const int sx = (sb >> 1) * 4, sy = (sb & 1) * 4;
                s1_subblock_pre(pre_lds + 16 * sb, rp + (ptrdiff_t)sy * a.ref[r].stride + sx, a.ref[r].stride, acc);
            }
        } else {
#pragma unroll 1
            for (int sb = 0; sb < 4; ++sb) {
                const int sx = (sb >> 1) * 4, sy = (sb & 1) * 4;
                s1_subblock(cp + (ptrdiff_t)sy * a.cur.stride + sx, a.cur.stride, rp + (ptrdiff_t)sy * a.ref[r].stride + sx, a.ref[r].stride, acc);
            }
        }
    }
    const int pen_mask = a.pixel_rate < 4 ? -1 : 0;     // x 32 on the two finest levels, x 0 above (a shift and a mask: `* pen_scale' with a scale the compiler cannot see compiles to a 64-bit multiply-add per candidate)
    const int pen_y = iabs(iabs(py - cy) - v0y);
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int px = (int16_t)(xb + i);
        const bool valid = loadable && px >= 0 && px <= a.w - 8;
        // :542-543 (sic): |displacement| minus the signed parent vector, only on the two finest levels
        const int diff = (acc[i] + (((iabs(iabs(px - cx) - v0x) + pen_y) << 5) & pen_mask)) & 0xffff;   // ushort accumulator
        const uint32_t key = (valid && diff < 0x7fff) ? ((uint32_t)diff << 8) | (uint32_t)(j * 5 + i) : 0xffffffffu;
        best = key < best ? key : best;
    }
    // minimum over the five dy lanes of the block (its sb = 0 lanes hold the full sums): (cost << 8 | dxy) = the
    // reference's first strict minimum
#pragma unroll
    for (int off = 1; off <= 4; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl((int)best, lane + off, 64);
        if (j + off < 5 && o < best) best = o;
    }
    int bx, by;  // best position minus block position
    if (best != 0xffffffffu) {
        const int k = best & 0xff;
        bx = v0x + (k % 5 - 2);
        by = v0y + (k / 5 - 2);
    } else {  // nothing accepted: "vector" still holds the scaled parent, :501,:551-556
        bx = (int16_t)(v0x - cx);
        by = (int16_t)(v0y - cy);
    }
    const uint32_t ox16 = (uint16_t)(int16_t)((int16_t)bx * (int16_t)a.pixel_rate);
    const uint32_t oy16 = (uint16_t)(int16_t)((int16_t)by * (int16_t)a.pixel_rate);
    return ox16 | (oy16 << 16);
}

// PRE_LDS (loop form): the current blocks' share of the metric made once per wave into LDS (s1_make_pre) instead of by every lane of a block
// REF_LOOP: a workgroup searches its blocks in EVERY enabled reference, one after the other (grid y = 1), instead of a workgroup per reference:
// the block arithmetic and the current blocks' share of the metric are made once for the two or three of them (batches; one video keeps a
// workgroup per reference: there a launch is as long as one wave)
template <bool SPLIT, bool PRE_LDS = false, bool REF_LOOP = false>
__device__ __forceinline__ void search1_body(const Search1Args &a) {
    using M = S1Map<SPLIT>;
    if (!REF_LOOP && (int)blockIdx.y >= a.nrefs) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / M::LANES_PER_BLOCK, sub = lane - M::LANES_PER_BLOCK * grp;
    const int b_raw = (xcd_band(blockIdx.x, gridDim.x) * 4 + wave) * M::BLOCKS_PER_WAVE + grp;
    const bool live = grp < M::BLOCKS_PER_WAVE && b_raw < a.nblk;
    const int b = live ? b_raw : a.nblk - 1;
    const int by = a.bw == 1 ? b : (int)__umulhi((uint32_t)b, a.bw_inv), bx = b - by * a.bw;   // b / bw: bw_inv = ceil(2^32 / bw), exact for b * bw < 2^32
    const int cx = bx * 8, cy = by * 8;
    // The reference zeroes the nets every frame (reset_vectors, :404-427) because parent cells beyond the
    // coarser level's block grid are read but never written; reading them as 0 here is the same thing
    // without the extra kernel.
    const int parent = (cy >> 4) * a.net_width + (cx >> 4);
    const bool parent_written = (cx >> 4) < a.pbw && (cy >> 4) < a.pbh;
    const int *pre_lds = nullptr;
    if (!SPLIT && PRE_LDS) {
        // the current blocks' share of the metric, once per wave into LDS: lane = (block slot, sub-block).  The stages hand over inside the wave
        // (LDS executes a wave's operations in order): no barrier, the compiler is kept from moving the reads up
        __shared__ __attribute__((aligned(16))) int s_pre[4][S1_PRE_INTS];
        const int slot = lane >> 2, sb = lane & 3;
        const int tb_raw = (xcd_band(blockIdx.x, gridDim.x) * 4 + wave) * M::BLOCKS_PER_WAVE + slot;
        const int tb = tb_raw < a.nblk ? tb_raw : a.nblk - 1;
        const int tby = a.bw == 1 ? tb : (int)__umulhi((uint32_t)tb, a.bw_inv), tbx = tb - tby * a.bw;
        s1_make_pre(a.cur, tbx * 8, tby * 8, sb, slot < M::BLOCKS_PER_WAVE, &s_pre[wave][(slot < M::BLOCKS_PER_WAVE ? slot : 0) * 64 + sb * 16]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pre_lds = &s_pre[wave][(grp < M::BLOCKS_PER_WAVE ? grp : 0) * 64];
    }
    const int nr = REF_LOOP ? a.nrefs : 1;
#pragma unroll 1
    for (int ri = 0; ri < nr; ++ri) {
        const int r = a.refmap[REF_LOOP ? ri : (int)blockIdx.y];
        const uint32_t pv = parent_written ? reinterpret_cast<const uint32_t *>(a.src[r])[parent] : 0u;
        const uint32_t out = search1_block<SPLIT, !SPLIT && PRE_LDS>(a, r, cx, cy, pv, live, sub, lane, pre_lds);
        if (sub == 0 && live) {
            const int cell = (cy >> 3) * a.net_width + (cx >> 3);
            reinterpret_cast<uint32_t *>(a.dst[r])[cell] = out;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Levels 4, 3, 2 and 1 of one reference's search in ONE launch, for a single video coded frame after frame: there the five levels are
// five links of the frame's dependency chain, each of the coarse four lasting as long as one wave of it plus a launch (4.5 + 4.8 + 5.0 +
// 7.8 us at 1080p: profiles/r05_single_stream_timeline.txt), and nothing else fills the part.  A workgroup takes a tile of 4 x 4 level-1
// blocks and computes the tile's ANCESTORS itself -- the one level-4 block, the one level-3 block and the 2 x 2 level-2 blocks whose
// vectors the tile descends from (the parent of block (bx, by) is block (bx / 2, by / 2) of the level above) -- keeping their vectors in
// LDS: 6 block searches on top of 16, no workgroup waits for another, and the chain has one link where it had four.  The same routine on
// the same inputs (search1_block), so the level-1 net -- the only one the next launch reads -- is the one four launches leave.
// ------------------------------------------------------------------------------------------------
// FINEST: level 0 as well -- the 8 x 8 level-0 blocks under the tile, in the loop form (five lanes per block, twelve blocks per wave:
// the tile's 64 blocks are one pass of the workgroup's six waves), their parents the tile's own level-1 vectors.  The whole hierarchical
// search of a reference is then ONE link of the chain.
struct CoarseArgs { Search1Args lv[5]; };   // lv[l] = level l
constexpr int COARSE_WAVES = 6;             // 18 block slots of 20 lanes: the tile's 16 level-1 blocks in one pass
// the tile's work: L0..L4 = the five levels' arguments (L0 read only with FINEST), r = the reference's slot in them
// TOP_ONLY: levels 4, 3 and 2 only (two waves are enough); the level-2 net leaves the workgroup and levels 1 and 0 are launches of their own
template <bool FINEST, bool TOP_ONLY = false>
__device__ __forceinline__ void coarse_tile(const Search1Args &L0, const Search1Args &L1, const Search1Args &L2, const Search1Args &L3, const Search1Args &L4, int r) {
    using M = S1Map<true>;
    __shared__ uint32_t mv4, mv3, mv2[4], mv1[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / M::LANES_PER_BLOCK, sub = lane - M::LANES_PER_BLOCK * grp;
    const int slot = grp < M::BLOCKS_PER_WAVE ? wave * M::BLOCKS_PER_WAVE + grp : 99;     // block slot of this lane in the workgroup, 0..17
    const int tiles_x = (L1.bw + 3) / 4;
    const int tile = xcd_band(blockIdx.x, gridDim.x);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    // ---- level 4: block (tx / 2, ty / 2) ----
    {
        const int bx = tx >> 1, by = ty >> 1;
        const bool exists = bx < L4.bw && by * L4.bw + bx < L4.nblk;
        if (wave == 0) {     // (whole waves only: the block's lanes shuffle among themselves)
            const bool live = slot == 0 && exists;
            const uint32_t out = search1_block<true>(L4, r, (live ? bx : 0) * 8, (live ? by : 0) * 8, 0u, live, sub, lane);
            if (slot == 0 && sub == 0) mv4 = exists ? out : 0u;      // (a cell beyond the coarser grid reads as zero: reset_vectors, :404-427)
        }
    }
    __syncthreads();
    // ---- level 3: block (tx, ty) ----
    {
        const bool exists = tx < L3.bw && ty * L3.bw + tx < L3.nblk;
        if (wave == 0) {
            const bool live = slot == 0 && exists;
            const uint32_t out = search1_block<true>(L3, r, (live ? tx : 0) * 8, (live ? ty : 0) * 8, mv4, live, sub, lane);
            if (slot == 0 && sub == 0) mv3 = exists ? out : 0u;
        }
    }
    __syncthreads();
    // ---- level 2: blocks (2 tx + i, 2 ty + j) ----
    if (wave < 2) {
        const int i = slot & 1, j = (slot >> 1) & 1;
        const int bx = 2 * tx + i, by = 2 * ty + j;
        const bool exists = slot < 4 && bx < L2.bw && by * L2.bw + bx < L2.nblk;
        const uint32_t out = search1_block<true>(L2, r, (exists ? bx : 0) * 8, (exists ? by : 0) * 8, mv3, exists, sub, lane);
        if (slot < 4 && sub == 0) mv2[slot] = exists ? out : 0u;
        if (TOP_ONLY && slot < 4 && sub == 0 && exists) reinterpret_cast<uint32_t *>(L2.dst[r])[by * L2.net_width + bx] = out;
    }
    if (TOP_ONLY) return;
    __syncthreads();
    // ---- level 1: the tile's 4 x 4 blocks; the only net that leaves the workgroup ----
    {
        const int i = slot & 3, j = (slot >> 2) & 3;
        const int bx = 4 * tx + i, by = 4 * ty + j;
        const bool live = slot < 16 && bx < L1.bw && by * L1.bw + bx < L1.nblk;
        const uint32_t pv = mv2[(((j >> 1) & 1) << 1) | ((i >> 1) & 1)];
        const uint32_t out = search1_block<true>(L1, r, (live ? bx : 0) * 8, (live ? by : 0) * 8, pv, live, sub, lane);
        if (sub == 0 && live) reinterpret_cast<uint32_t *>(L1.dst[r])[by * L1.net_width + bx] = out;
        if (FINEST && sub == 0 && slot < 16) mv1[slot] = live ? out : 0u;
    }
    if (!FINEST) return;
    __syncthreads();
    // ---- level 0: the 8 x 8 blocks under the tile ----
    {
        using M0 = S1Map<false>;
        const int grp0 = lane / M0::LANES_PER_BLOCK, sub0 = lane - M0::LANES_PER_BLOCK * grp0;
        const int slot0 = grp0 < M0::BLOCKS_PER_WAVE ? wave * M0::BLOCKS_PER_WAVE + grp0 : 99;     // 0..71
        const int i = slot0 & 7, j = (slot0 >> 3) & 7;
        const int bx = 8 * tx + i, by = 8 * ty + j;
        const bool live = slot0 < 64 && bx < L0.bw && by * L0.bw + bx < L0.nblk;
        const uint32_t pv = mv1[((j >> 1) << 2) | (i >> 1)];
        const uint32_t out = search1_block<false>(L0, r, (live ? bx : 0) * 8, (live ? by : 0) * 8, pv, live, sub0, lane);
        if (sub0 == 0 && live) reinterpret_cast<uint32_t *>(L0.dst[r])[by * L0.net_width + bx] = out;
    }
}

template <bool FINEST>
__global__ __launch_bounds__(64 * COARSE_WAVES) void k_search1_coarse(CoarseArgs a) {
    if ((int)blockIdx.y >= a.lv[1].nrefs) return;
    coarse_tile<FINEST>(a.lv[0], a.lv[1], a.lv[2], a.lv[3], a.lv[4], a.lv[1].refmap[blockIdx.y]);
}

// The same for the members of a BATCH (blockIdx.z = member): ONE launch where the batched path had four (levels 4-1) or five.  Eight members'
// Search1Args of four or five levels do not fit the 4 KiB kernel-argument segment, so what all members share -- the levels' geometry: the
// members are contexts of one size, their surfaces and nets are laid out alike -- travels once and a member brings its addresses only.
struct CoarseGeom { int w, h, cur_stride, ref_stride; };
struct CoarseMember {
    const uint8_t *cur[5];          // pixel (0, 0) of level l of the current frame
    const uint8_t *ref[3][5];       // ... of reference r
    int16_t *dst1[3], *dst0[3];     // the level-1 and level-0 nets of reference r (where launch_search1_batch leaves them)
    int refmap[3], nrefs;
};
struct CoarseBatchArgs { CoarseGeom g[5]; int net_width, n; CoarseMember m[MAX_BATCH]; };
static_assert(sizeof(CoarseBatchArgs) <= 4096, "a batch's argument block travels in the 4 KiB kernel-argument segment");
__device__ __forceinline__ Search1Args coarse_level(const CoarseBatchArgs &a, const CoarseMember &m, int l, int r) {
    Search1Args s;
    const CoarseGeom &g = a.g[l];
    s.cur = Plane{const_cast<uint8_t *>(m.cur[l]), g.cur_stride, g.w, g.h};
    s.ref[0] = Plane{const_cast<uint8_t *>(m.ref[r][l]), g.ref_stride, g.w, g.h};
    s.dst[0] = l == 1 ? m.dst1[r] : m.dst0[r];        // (only levels 1 and 0 leave the workgroup)
    s.net_width = a.net_width;
    s.w = g.w;
    s.h = g.h;
    s.pixel_rate = 1 << l;
    s.rate_shift = l;
    s.bw = g.w / 8;
    s.nblk = (g.w / 8) * (g.h / 8);
    return s;
}
__global__ __launch_bounds__(128) void k_search1_top_b(CoarseBatchArgs a) {
    const CoarseMember &m = a.m[blockIdx.z];
    if ((int)blockIdx.y >= m.nrefs) return;
    const int r = m.refmap[blockIdx.y];
    const Search1Args L1 = coarse_level(a, m, 1, r), L2 = coarse_level(a, m, 2, r), L3 = coarse_level(a, m, 3, r), L4 = coarse_level(a, m, 4, r);
    coarse_tile<false, true>(L1, L1, L2, L3, L4, 0);      // (level 1 gives the tile grid; its blocks are not searched here)
}
template <bool FINEST>
__global__ __launch_bounds__(64 * COARSE_WAVES) void k_search1_coarse_b(CoarseBatchArgs a) {
    const CoarseMember &m = a.m[blockIdx.z];
    if ((int)blockIdx.y >= m.nrefs) return;
    const int r = m.refmap[blockIdx.y];
    const Search1Args L0 = coarse_level(a, m, 0, r), L1 = coarse_level(a, m, 1, r), L2 = coarse_level(a, m, 2, r), L3 = coarse_level(a, m, 3, r),
                      L4 = coarse_level(a, m, 4, r);
    coarse_tile<FINEST>(L0, L1, L2, L3, L4, 0);       // (slot 0 of the locals holds reference r)
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void k_search1(Search1Args a) { search1_body<SPLIT>(a); }
static_assert(sizeof(BatchOf<Search1Args>) <= 4096 && sizeof(PyrArgs) <= 4096 && sizeof(BatchOf<PackItem>) <= 4096, "a batch's argument blocks travel in the 4 KiB kernel-argument segment");
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_search1_b(BatchOf<Search1Args> b) { search1_body<SPLIT>(b.item[blockIdx.z]); }
__global__ __launch_bounds__(256) void k_search1_pl(Search1Args a) { search1_body<false, true>(a); }
__global__ __launch_bounds__(256) void k_search1_pl_b(BatchOf<Search1Args> b) { search1_body<false, true>(b.item[blockIdx.z]); }
__global__ __launch_bounds__(256) void k_search1_plr_b(BatchOf<Search1Args> b) { search1_body<false, true, true>(b.item[blockIdx.z]); }
// VP8HIP_S1_REF_LOOP=0: a workgroup per reference in batches too (same-box A/B runs)
static bool search1_ref_loop() {
    static const bool on = [] { const char *v = getenv("VP8HIP_S1_REF_LOOP"); return !(v && v[0] == '0'); }();
    return on;
}
// VP8HIP_S1_PRE_LDS=0: the loop form as it was (every lane makes the current block's share itself); same-box A/B runs
static bool search1_pre_lds() {
    static const bool on = [] { const char *v = getenv("VP8HIP_S1_PRE_LDS"); return !(v && v[0] == '0'); }();
    return on;
}

static Search1Args search1_args(const Frame &cur, const RefSet &refs, const NetSet &nets, int level, int src_idx, int net_width) {
    Search1Args a;
    a.cur = cur.Y[level];
    int n = 0;
    for (int r = 0; r < 3; ++r) {
        a.ref[r] = refs.ref[r].Y[level];
        a.src[r] = nets.net[r][src_idx];
        a.dst[r] = nets.net[r][src_idx ^ 1];
        if (refs.use[r]) a.refmap[n++] = r;
    }
    a.nrefs = n;
    for (int i = n; i < 3; ++i) a.refmap[i] = 0;
    a.net_width = net_width;
    a.w = a.cur.w;
    a.h = a.cur.h;
    a.pixel_rate = 1 << level;
    a.rate_shift = level;
    a.bw = a.w / 8;
    a.nblk = (a.w / 8) * (a.h / 8);
    a.bw_inv = a.bw > 0 ? (uint32_t)(((1ull << 32) + a.bw - 1) / a.bw) : 0;
    a.pbw = level < 4 ? cur.Y[level + 1].w / 8 : 0;
    a.pbh = level < 4 ? cur.Y[level + 1].h / 8 : 0;
    return a;
}

int persistent_workgroups() {
    static const int n = [] { const char *v = getenv("VP8HIP_PERSIST"); return v && v[0] ? atoi(v) : 0; }();
    return n;
}

static bool search1_skip() {
    static const bool skip = experiment_skip("s1");
    return skip;   // timing experiment only
}
// fewer waves than the chip has SIMDs: the launch is as long as one wave whatever else runs -> the short-wave form.
// VP8HIP_S1_SPLIT=0/1 forces one form (same-box A/B runs)
static bool search1_split(size_t blocks_times_refs, bool latency) {
    static const int forced = [] { const char *v = getenv("VP8HIP_S1_SPLIT"); return v && v[0] ? (v[0] == '1' ? 1 : 0) : -1; }();
    return forced >= 0 ? forced == 1 : (latency || blocks_times_refs < (size_t)12 * 1024);
}

void launch_search1(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, int level, int src_idx,
                    int net_width, bool latency) {
    const Search1Args a = search1_args(cur, refs, nets, level, src_idx, net_width);
    const int n = a.nrefs;
    if (a.nblk <= 0 || n == 0 || search1_skip()) return;
    if (search1_split((size_t)a.nblk * n, latency))
        VP8_LAUNCH(k_search1<true>, dim3((a.nblk + S1Map<true>::BLOCKS_PER_WG - 1) / S1Map<true>::BLOCKS_PER_WG, n), dim3(256), 0, s, a);
    else if (search1_pre_lds())
        VP8_LAUNCH(k_search1_pl, dim3((a.nblk + S1Map<false>::BLOCKS_PER_WG - 1) / S1Map<false>::BLOCKS_PER_WG, n), dim3(256), 0, s, a);
    else
        VP8_LAUNCH(k_search1<false>, dim3((a.nblk + S1Map<false>::BLOCKS_PER_WG - 1) / S1Map<false>::BLOCKS_PER_WG, n), dim3(256), 0, s, a);
}

// levels 4..1 (finest = false) or 4..0 in one launch (k_search1_coarse); the level-1 and level-0 nets land where launch_search1 leaves them
void launch_search1_coarse(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, int net_width, bool finest) {
    CoarseArgs a;
    for (int l = 0; l <= 4; ++l) a.lv[l] = search1_args(cur, refs, nets, l, l & 1, net_width);   // src of level l: 0 for 4, 2 and 0; 1 for 3 and 1
    const Search1Args &L0 = a.lv[0], &L1 = a.lv[1];
    if (L0.nblk <= 0 || L0.nrefs == 0 || search1_skip()) return;
    // tiles of 4 x 4 level-1 blocks = 8 x 8 level-0 blocks; a frame of one macroblock row or column has level-0 blocks without a level-1 block above them
    const int tiles_x = finest ? (L0.bw + 7) / 8 : (L1.bw + 3) / 4, tiles_y = finest ? (L0.h / 8 + 7) / 8 : (L1.h / 8 + 3) / 4;
    if (tiles_x * tiles_y <= 0) return;
    if (finest) VP8_LAUNCH(k_search1_coarse<true>, dim3(tiles_x * tiles_y, L0.nrefs), dim3(64 * COARSE_WAVES), 0, s, a);
    else VP8_LAUNCH(k_search1_coarse<false>, dim3(tiles_x * tiles_y, L0.nrefs), dim3(64 * COARSE_WAVES), 0, s, a);
}

// levels 4..1 (finest = false) or 4..0 of every member of a batch in ONE launch; false = the members' surfaces are not laid out alike
// (never so for contexts of one size: the caller then launches level by level)
bool launch_search1_coarse_batch(hipStream_t s, const Frame *const *cur, const RefSet *refs, const NetSet *const *nets, int net_width, int n, bool finest, bool top_only) {
    CoarseBatchArgs a;
    a.net_width = net_width;
    a.n = n;
    int maxrefs = 0;
    for (int l = 0; l <= 4; ++l) {
        const Plane &c = cur[0]->Y[l];
        a.g[l] = CoarseGeom{c.w, c.h, c.stride, refs[0].ref[0].Y[l].stride};
    }
    for (int i = 0; i < n; ++i) {
        CoarseMember &m = a.m[i];
        int k = 0;
        for (int r = 0; r < 3; ++r) {
            for (int l = 0; l <= 4; ++l) {
                const Plane &rp = refs[i].ref[r].Y[l], &cp = cur[i]->Y[l];
                if (rp.stride != a.g[l].ref_stride || cp.stride != a.g[l].cur_stride || cp.w != a.g[l].w || cp.h != a.g[l].h || rp.w != cp.w || rp.h != cp.h) return false;
                m.ref[r][l] = rp.p;
                if (r == 0) m.cur[l] = cp.p;
            }
            m.dst1[r] = nets[i]->net[r][0];       // launch_search1_batch's ping-pong: level 4 reads net 0 ... level 1 writes net 0, level 0 writes net 1
            m.dst0[r] = nets[i]->net[r][1];
            if (refs[i].use[r]) m.refmap[k++] = r;
        }
        m.nrefs = k;
        for (int j = k; j < 3; ++j) m.refmap[j] = 0;
        maxrefs = k > maxrefs ? k : maxrefs;
    }
    const int bw0 = a.g[0].w / 8, bh0 = a.g[0].h / 8, bw1 = a.g[1].w / 8, bh1 = a.g[1].h / 8;
    if (bw0 * bh0 <= 0 || maxrefs == 0 || search1_skip()) return true;
    const int tiles_x = finest ? (bw0 + 7) / 8 : (bw1 + 3) / 4, tiles_y = finest ? (bh0 + 7) / 8 : (bh1 + 3) / 4;
    if (tiles_x * tiles_y <= 0) return !finest ? true : false;
    if (top_only) {     // tiles of 2 x 2 level-2 blocks = one level-3 block (the tile grid is the same: 4 x 4 level-1 blocks)
        VP8_LAUNCH(k_search1_top_b, dim3(tiles_x * tiles_y, maxrefs, n), dim3(128), 0, s, a);
        return true;
    }
    if (finest) VP8_LAUNCH(k_search1_coarse_b<true>, dim3(tiles_x * tiles_y, maxrefs, n), dim3(64 * COARSE_WAVES), 0, s, a);
    else VP8_LAUNCH(k_search1_coarse_b<false>, dim3(tiles_x * tiles_y, maxrefs, n), dim3(64 * COARSE_WAVES), 0, s, a);
    return true;
}

void launch_search1_batch(hipStream_t s, const Frame *const *cur, const RefSet *refs, const NetSet *const *nets, int level, int src_idx,
                          int net_width, int n) {
    BatchOf<Search1Args> b;
    b.n = n;
    int maxrefs = 0, totrefs = 0;
    for (int i = 0; i < n; ++i) {
        b.item[i] = search1_args(*cur[i], refs[i], *nets[i], level, src_idx, net_width);
        maxrefs = b.item[i].nrefs > maxrefs ? b.item[i].nrefs : maxrefs;
        totrefs += b.item[i].nrefs;
    }
    const int nblk = b.item[0].nblk;
    if (nblk <= 0 || maxrefs == 0 || search1_skip()) return;
    if (search1_split((size_t)nblk * totrefs, false))
        VP8_LAUNCH(k_search1_b<true>, dim3((nblk + S1Map<true>::BLOCKS_PER_WG - 1) / S1Map<true>::BLOCKS_PER_WG, maxrefs, n), dim3(256), 0, s, b);
    else if (search1_pre_lds() && search1_ref_loop())     // (70 registers, seven waves per SIMD; held to 64 it spills six and is no faster: 73.4-73.6 either way)
        VP8_LAUNCH(k_search1_plr_b, dim3((nblk + S1Map<false>::BLOCKS_PER_WG - 1) / S1Map<false>::BLOCKS_PER_WG, 1, n), dim3(256), 0, s, b);
    else if (search1_pre_lds())
        VP8_LAUNCH(k_search1_pl_b, dim3((nblk + S1Map<false>::BLOCKS_PER_WG - 1) / S1Map<false>::BLOCKS_PER_WG, maxrefs, n), dim3(256), 0, s, b);
    else
        VP8_LAUNCH(k_search1_b<false>, dim3((nblk + S1Map<false>::BLOCKS_PER_WG - 1) / S1Map<false>::BLOCKS_PER_WG, maxrefs, n), dim3(256), 0, s, b);
}

// test tap: the block-match metric on caller-supplied difference blocks (n x 16 ints)
__global__ __launch_bounds__(256) void k_weight_tap(const int32_t *d, int n, int32_t *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // the production form (weight_cols_pre) on a (current, prediction) byte pair with current - prediction = d:
    // columns as dwords, biased by -128
    uint32_t cc[4], pp[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cc[c] = pp[c] = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = d[i * 16 + 4 * r + c];
            const uint32_t cb = (uint32_t)(v > 0 ? v : 0) ^ 0x80u, pb = (uint32_t)(v > 0 ? 0 : -v) ^ 0x80u;
            cc[c] |= cb << (8 * r);
            pp[c] |= pb << (8 * r);
        }
    }
    int pre[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) weight_pre_column(cc[c], pre + 4 * c);
    out[i] = weight_cols_pre(pre, pp);
}
void launch_weight_tap(hipStream_t s, const int32_t *d, int n, int32_t *out) {
    hipLaunchKernelGGL(k_weight_tap, dim3((n + 255) / 256), dim3(256), 0, s, d, n, out);
}

}  // namespace vp8
