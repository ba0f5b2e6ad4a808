// kernels_me.hip -- motion estimation side of the inter-frame path for gfx950:
//   edge replication, pyramid, hierarchical full-pel search, quarter-pel search, reference choice.
// Behaviour follows the reference kernels cited at each kernel (paths under the reference's src/);
// the parallel decomposition is new: one 32-lane half-wave per 8x8 block with one search
// candidate per lane, the winner found with a wave-level packed (cost,index) minimum.
#include "vp8hip_dev.h"

namespace vp8 {

// ------------------------------------------------------------------------------------------------
// Edge replication of a reference frame's Y/U/V planes (replaces the clamp-to-edge image sampler,
// GPU_kernels.cl:562).  grid = (max rows + 2*EXT, 3 planes), block = 64.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_border(Plane py, Plane pu, Plane pv) {
    const Plane pl = blockIdx.y == 0 ? py : (blockIdx.y == 1 ? pu : pv);
    const int row = (int)blockIdx.x - EXT;
    if (row >= pl.h + EXT) return;
    uint8_t *dst = pl.p + (ptrdiff_t)row * pl.stride;
    if (row >= 0 && row < pl.h) {
        for (int t = threadIdx.x; t < 2 * EXT; t += 64) {
            const int x = t < EXT ? t - EXT : pl.w + (t - EXT);
            dst[x] = dst[t < EXT ? 0 : pl.w - 1];
        }
    } else {
        const uint8_t *src = pl.p + (ptrdiff_t)iclamp(row, 0, pl.h - 1) * pl.stride;
        for (int x = -EXT + (int)threadIdx.x; x < pl.w + EXT; x += 64) dst[x] = src[iclamp(x, 0, pl.w - 1)];
    }
}

void launch_border(hipStream_t s, const Frame &f) {
    dim3 grid(f.Y[0].h + 2 * EXT, 3);
    hipLaunchKernelGGL(k_border, grid, dim3(64), 0, s, f.Y[0], f.U, f.V);
}

// ------------------------------------------------------------------------------------------------
// downsample_x2, GPU_kernels.cl:429-451: dst = (a+b+c+d+2)/4 of each 2x2.  One thread per output
// pixel pair-of-rows column; blockIdx.y selects the surface (current / LAST are done together).
// ------------------------------------------------------------------------------------------------
struct DownArgs { Plane src[2], dst[2]; };

__global__ __launch_bounds__(256) void k_downsample(DownArgs a) {
    const Plane s = a.src[blockIdx.y], d = a.dst[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.w * d.h) return;
    const int x = i % d.w, y = i / d.w;
    const uint8_t *p = s.p + (ptrdiff_t)(2 * y) * s.stride + 2 * x;
    const int v = p[0] + p[1] + p[s.stride] + p[s.stride + 1] + 2;
    d.p[(ptrdiff_t)y * d.stride + x] = (uint8_t)(v >> 2);
}

void launch_downsample(hipStream_t s, const Plane *src, const Plane *dst, int nsurf) {
    DownArgs a;
    for (int i = 0; i < 2; ++i) {
        a.src[i] = src[i < nsurf ? i : 0];
        a.dst[i] = dst[i < nsurf ? i : 0];
    }
    const int n = dst[0].w * dst[0].h;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_downsample, dim3((n + 255) / 256, nsurf), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// The whole pyramid in one launch (replaces 4 x downsample_x2 per surface, inter_part.h:11-33).
// A workgroup takes a 64x64 tile of the full-resolution plane and produces the 32x32, 16x16, 8x8 and
// 4x4 tiles below it; every level is computed from the ROUNDED level above it, exactly like the
// reference's cascade of launches.  grid = (ceil(W/64), ceil(H/64), surfaces), block = 256.
// ------------------------------------------------------------------------------------------------
struct PyrArgs { Frame f[2]; };

__global__ __launch_bounds__(256) void k_pyramid(PyrArgs a) {
    __shared__ uint8_t s2[16][16];
    __shared__ uint8_t s3[8][8];
    const Frame &f = a.f[blockIdx.z];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const Plane &P0 = f.Y[0];
    // 4x4 source pixels -> 2x2 of level 1 -> 1 of level 2
    const int sx = imin(blockIdx.x * 64 + 4 * tx, P0.w - 4), sy0 = blockIdx.y * 64 + 4 * ty;
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
    const int x1 = blockIdx.x * 32 + 2 * tx, y1 = blockIdx.y * 32 + 2 * ty;
    if (x1 < P1.w) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (y1 + j < P1.h)
                *reinterpret_cast<uint16_t *>(P1.p + (ptrdiff_t)(y1 + j) * P1.stride + x1) = (uint16_t)(l1[j][0] | (l1[j][1] << 8));
    }
    const int l2 = (l1[0][0] + l1[0][1] + l1[1][0] + l1[1][1] + 2) >> 2;
    const Plane &P2 = f.Y[2];
    const int x2 = blockIdx.x * 16 + tx, y2 = blockIdx.y * 16 + ty;
    if (x2 < P2.w && y2 < P2.h) P2.p[(ptrdiff_t)y2 * P2.stride + x2] = (uint8_t)l2;
    s2[ty][tx] = (uint8_t)l2;
    __syncthreads();
    if (t < 64) {
        const int ux = t & 7, uy = t >> 3;
        const int l3 = (s2[2 * uy][2 * ux] + s2[2 * uy][2 * ux + 1] + s2[2 * uy + 1][2 * ux] + s2[2 * uy + 1][2 * ux + 1] + 2) >> 2;
        const Plane &P3 = f.Y[3];
        const int x3 = blockIdx.x * 8 + ux, y3 = blockIdx.y * 8 + uy;
        if (x3 < P3.w && y3 < P3.h) P3.p[(ptrdiff_t)y3 * P3.stride + x3] = (uint8_t)l3;
        s3[uy][ux] = (uint8_t)l3;
    }
    __syncthreads();
    if (t < 16) {
        const int ux = t & 3, uy = t >> 2;
        const int l4 = (s3[2 * uy][2 * ux] + s3[2 * uy][2 * ux + 1] + s3[2 * uy + 1][2 * ux] + s3[2 * uy + 1][2 * ux + 1] + 2) >> 2;
        const Plane &P4 = f.Y[4];
        const int x4 = blockIdx.x * 4 + ux, y4 = blockIdx.y * 4 + uy;
        if (x4 < P4.w && y4 < P4.h) P4.p[(ptrdiff_t)y4 * P4.stride + x4] = (uint8_t)l4;
    }
}

void launch_pyramid(hipStream_t s, const Frame *a, const Frame *b) {
    PyrArgs p;
    p.f[0] = *a;
    p.f[1] = b ? *b : *a;
    hipLaunchKernelGGL(k_pyramid, dim3((a->Y[0].w + 63) / 64, (a->Y[0].h + 63) / 64, b ? 2 : 1), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------------
// Tight HBM-resident Y,U,V planes -> the padded surfaces of a frame, one launch (8 bytes per thread).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack(Plane py, Plane pu, Plane pv, const uint8_t *sy, const uint8_t *su,
                                              const uint8_t *sv) {
    int i = blockIdx.x * 256 + threadIdx.x;
    const int ny = (py.w >> 3) * py.h, nc = (pu.w >> 3) * pu.h;
    const Plane *pl = &py;
    const uint8_t *src = sy;
    if (i >= ny) {
        i -= ny;
        pl = &pu;
        src = su;
        if (i >= nc) { i -= nc; pl = &pv; src = sv; }
        if (i >= nc) return;
    }
    const int upr = pl->w >> 3;     // 8-byte units per row
    const int y = i / upr, x = (i % upr) * 8;
    *reinterpret_cast<uint2 *>(pl->p + (ptrdiff_t)y * pl->stride + x) = *reinterpret_cast<const uint2 *>(src + (size_t)y * pl->w + x);
}

void launch_pack(hipStream_t s, const Frame &f, const void *y, const void *u, const void *v) {
    const int n = (f.Y[0].w >> 3) * f.Y[0].h + 2 * ((f.U.w >> 3) * f.U.h);
    hipLaunchKernelGGL(k_pack, dim3((n + 255) / 256), dim3(256), 0, s, f.Y[0], f.U, f.V, (const uint8_t *)y,
                       (const uint8_t *)u, (const uint8_t *)v);
}

// ------------------------------------------------------------------------------------------------
// reset_vectors, GPU_kernels.cl:404-427
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reset_nets(NetSet n, int b8) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= b8) return;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        reinterpret_cast<uint32_t *>(n.net[r][0])[i] = 0;
        reinterpret_cast<uint32_t *>(n.net[r][1])[i] = 0;
        n.bdiff[r][i] = 0x7fffffff;
    }
}

void launch_reset_nets(hipStream_t s, const NetSet &n, int b8) {
    hipLaunchKernelGGL(k_reset_nets, dim3((b8 + 255) / 256), dim3(256), 0, s, n, b8);
}

// minimum of a packed key over the 32 lanes of a half-wave
__device__ __forceinline__ uint32_t halfwave_min(uint32_t key) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)key, m, 32);
        key = o < key ? o : key;
    }
    return key;
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
    int net_width, w, h, pixel_rate, rate_shift, nblk, bw;
    int pbw, pbh;   // block grid of the coarser level (whose cells of src[] were written this frame)
};

// Mapping: five lanes per 8x8 block, lane j = candidate row dy = j-2; the lane loads its eight
// 12-byte reference rows once and walks the five dx candidates over them in registers (the window
// bytes are shared by the five candidates of a row).  12 blocks per wave (60 of 64 lanes busy).
constexpr int S1_BLOCKS_PER_WAVE = 12;
constexpr int S1_BLOCKS_PER_WG = 4 * S1_BLOCKS_PER_WAVE;

__global__ __launch_bounds__(256) void k_search1(Search1Args a) {
    const int r = a.refmap[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / 5, j = lane - 5 * grp;
    const int b_raw = (blockIdx.x * 4 + wave) * S1_BLOCKS_PER_WAVE + grp;
    const bool live = grp < S1_BLOCKS_PER_WAVE && b_raw < a.nblk;
    const int b = live ? b_raw : a.nblk - 1;
    const int cx = (b % a.bw) * 8, cy = (b / a.bw) * 8;
    // The reference zeroes the nets every frame (reset_vectors, :404-427) because parent cells beyond the
    // coarser level's block grid are read but never written; reading them as 0 here is the same thing
    // without the extra kernel.  vector / pixel_rate truncates toward zero (:495-500).
    const int parent = (cy >> 4) * a.net_width + (cx >> 4);
    const bool parent_written = (cx >> 4) < a.pbw && (cy >> 4) < a.pbh;
    const uint32_t pv = parent_written ? reinterpret_cast<const uint32_t *>(a.src[r])[parent] : 0u;
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
    // One 4x4 sub-block at a time (loop NOT unrolled): 4 current dwords + 4 x 8 reference bytes feed
    // the five dx candidates, whose costs accumulate in acc[].  Measured on MI355X
    // (scripts/ubench/valu_rates.hip): one wave issues a VALU instruction every ~5.5 cycles whatever
    // the instruction, and throughput scales linearly to >= 4 waves per SIMD -- so the register
    // footprint (waves per SIMD), not the instruction mix, decides the speed of this kernel.
    int acc[5] = {0, 0, 0, 0, 0};
#pragma unroll 1
    for (int sb = 0; sb < 4; ++sb) {
        const int sx = (sb >> 1) * 4, sy = (sb & 1) * 4;
        uint32_t c[4], q0[4], q1[4];
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            c[y] = *reinterpret_cast<const uint32_t *>(cp + (ptrdiff_t)(sy + y) * a.cur.stride + sx);
            const uint2 q = ld_u64(rp + (ptrdiff_t)(sy + y) * a.ref[r].stride + sx);
            q0[y] = q.x; q1[y] = q.y;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            uint32_t p[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) p[y] = i == 0 ? q0[y] : (i == 4 ? q1[y] : __builtin_amdgcn_alignbyte(q1[y], q0[y], i));
            acc[i] += weight_quads(c, p);
        }
    }
    const int pen_scale = a.pixel_rate < 4 ? 32 : 0;
    const int pen_y = iabs(iabs(py - cy) - v0y);
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int px = (int16_t)(xb + i);
        const bool valid = loadable && px >= 0 && px <= a.w - 8;
        // :542-543 (sic): |displacement| minus the signed parent vector, only on the two finest levels
        const int diff = (acc[i] + (iabs(iabs(px - cx) - v0x) + pen_y) * pen_scale) & 0xffff;   // ushort accumulator
        const uint32_t key = (valid && diff < 0x7fff) ? ((uint32_t)diff << 8) | (uint32_t)(j * 5 + i) : 0xffffffffu;
        best = key < best ? key : best;
    }
    // minimum over the five lanes of the block: (cost << 8 | dxy) = the reference's first strict minimum
#pragma unroll
    for (int off = 1; off <= 4; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl((int)best, lane + off, 64);
        if (j + off < 5 && o < best) best = o;
    }
    if (j == 0 && live) {
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
        const int cell = (cy >> 3) * a.net_width + (cx >> 3);
        reinterpret_cast<uint32_t *>(a.dst[r])[cell] = ox16 | (oy16 << 16);
    }
}

void launch_search1(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, int level, int src_idx,
                    int net_width) {
    Search1Args a;
    a.cur = cur.Y[level];
    int n = 0;
    for (int r = 0; r < 3; ++r) {
        a.ref[r] = refs.ref[r].Y[level];
        a.src[r] = nets.net[r][src_idx];
        a.dst[r] = nets.net[r][src_idx ^ 1];
        if (refs.use[r]) a.refmap[n++] = r;
    }
    for (int i = n; i < 3; ++i) a.refmap[i] = 0;
    a.net_width = net_width;
    a.w = a.cur.w;
    a.h = a.cur.h;
    a.pixel_rate = 1 << level;
    a.rate_shift = level;
    a.bw = a.w / 8;
    a.nblk = (a.w / 8) * (a.h / 8);
    a.pbw = level < 4 ? cur.Y[level + 1].w / 8 : 0;
    a.pbh = level < 4 ? cur.Y[level + 1].h / 8 : 0;
    if (a.nblk <= 0 || n == 0) return;
    hipLaunchKernelGGL(k_search1, dim3((a.nblk + S1_BLOCKS_PER_WG - 1) / S1_BLOCKS_PER_WG, n), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// luma_search_2step, GPU_kernels.cl:1068-1203 (+construct_opt1/2 :776-1066).
// v0 = 4*net is a whole-pel vector, so the 25 candidates q = 4c+v0+(dx,dy), dx,dy in -2..2 use only
// five x cases and five y cases: (integer offset, 1/8-pel phase) = (-1,4) (-1,6) (0,0) (0,2) (0,4).
// Per block (32 lanes): stage the 14x14 reference window in LDS, filter the 14 rows once for each of
// the five x cases (saturated to u8, exactly what every construct_opt call would recompute), then
// lane k applies only its vertical taps, takes the 4 sub-block costs and the winner is a packed min.
// Lane 25 is the explicit zero-MV candidate (changelog.txt:93).  grid = (ceil(b8/8), refs), block 256.
// ------------------------------------------------------------------------------------------------
struct Search2Args {
    Plane cur;
    Plane ref[3];
    const int16_t *net_in[3];
    int16_t *net_out[3];
    int32_t *bdiff[3];
    int refmap[3];
    int w, h, nblk, bw;
    uint32_t *dbg;   // test tap: per-candidate prediction and cost of block dbg_block (or nullptr)
    int dbg_block;
};

__device__ __forceinline__ int case_phase(int c) { return c == 0 ? 4 : (c == 1 ? 6 : (c == 2 ? 0 : (c == 3 ? 2 : 4))); }

__global__ __launch_bounds__(256) void k_search2(Search2Args a) {
    __shared__ uint32_t s_win[8][14 * 5];
    __shared__ uint32_t s_H[8][5 * 14 * 2];
    const int r = a.refmap[blockIdx.y];
    const int g = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const int b = imin(blockIdx.x * 8 + g, a.nblk - 1);
    const bool live = blockIdx.x * 8 + g < a.nblk;
    const int cx = (b % a.bw) * 8, cy = (b / a.bw) * 8;
    const uint32_t nv = reinterpret_cast<const uint32_t *>(a.net_in[r])[b];
    const int nx = (int16_t)(nv & 0xffffu), ny = (int16_t)(nv >> 16);
    const int v0x = (int16_t)(nx * 4), v0y = (int16_t)(ny * 4);
    // window origin; a garbage vector (possible only when every candidate is out of frame) is clamped
    // so that the loads stay inside the allocated margin
    const int Lx = iclamp(cx + nx, 3 - EXT, a.w + EXT - 11), Ly = iclamp(cy + ny, 3 - EXT, a.h + EXT - 11);
    const Plane rf = a.ref[r];
    const int ax = (Lx - 3) & ~3, o = (Lx - 3) & 3;
    for (int idx = lane; idx < 70; idx += 32) {
        const int row = idx / 5, j = idx % 5;
        s_win[g][idx] = *reinterpret_cast<const uint32_t *>(rf.p + (ptrdiff_t)(Ly - 3 + row) * rf.stride + ax + 4 * j);
    }
    __syncthreads();
    // horizontal pass: (x case, window row) pairs
    for (int pi = lane; pi < 70; pi += 32) {
        const int xc = pi / 14, row = pi % 14;
        const int xo = xc < 2 ? -1 : 0, phx = case_phase(xc);
        const int s0 = o + xo + 1;  // byte of the row that holds tap 0 of output column 0
        uint32_t w[6];
#pragma unroll
        for (int j = 0; j < 5; ++j) w[j] = s_win[g][row * 5 + j];
        w[5] = 0;
        const bool j0 = (s0 >> 2) != 0;
        const int sh = s0 & 3;
        uint32_t q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = __builtin_amdgcn_alignbyte(j0 ? w[j + 2] : w[j + 1], j0 ? w[j + 1] : w[j], sh);
        int bb[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) bb[i] = byte_of(q[i >> 2], i & 3);
        uint32_t out[2] = {0, 0};
        if (phx == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) out[c >> 2] |= (uint32_t)bb[c + 2] << (8 * (c & 3));
        } else {
            int f[6];
            load_taps(phx, f);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                int sum = 64;
#pragma unroll
                for (int t = 0; t < 6; ++t) sum += bb[c + t] * f[t];
                out[c >> 2] |= (uint32_t)sat8_shr7(sum) << (8 * (c & 3));  // negative sums saturate to 0 either way
            }
        }
        s_H[g][pi * 2] = out[0];
        s_H[g][pi * 2 + 1] = out[1];
    }
    __syncthreads();

    const int k = lane;
    const int dx = k % 5 - 2, dy = k / 5 - 2;
    int qx = (int16_t)(cx * 4 + v0x + dx), qy = (int16_t)(cy * 4 + v0y + dy);
    if (k == 25) { qx = cx * 4; qy = cy * 4; }
    const bool valid = live && k < 26 && qx >= 0 && qx <= a.w * 4 - 32 && qy >= 0 && qy <= a.h * 4 - 32;
    uint32_t p_lo[8], p_hi[8];
    if (k < 25) {
        const int xc = k % 5, yc = k / 5;
        const int rb = (yc < 2 ? -1 : 0) + 1, phy = case_phase(yc);
        const uint32_t *H = &s_H[g][xc * 28];
        if (phy == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { p_lo[i] = H[(rb + i + 2) * 2]; p_hi[i] = H[(rb + i + 2) * 2 + 1]; }
        } else {
            int f[6];
            load_taps(phy, f);
            uint32_t hl[13], hh[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) { hl[i] = H[(rb + i) * 2]; hh[i] = H[(rb + i) * 2 + 1]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    int s1 = 64, s2 = 64;
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        s1 += byte_of(hl[i + t], c) * f[t];
                        s2 += byte_of(hh[i + t], c) * f[t];
                    }
                    lo |= (uint32_t)sat8_shr7(s1) << (8 * c);
                    hi |= (uint32_t)sat8_shr7(s2) << (8 * c);
                }
                p_lo[i] = lo; p_hi[i] = hi;
            }
        }
    } else {  // zero MV: whole-pel, both passes are the identity
        const uint8_t *zp = rf.p + (ptrdiff_t)cy * rf.stride + cx;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint2 z = *reinterpret_cast<const uint2 *>(zp + (ptrdiff_t)i * rf.stride);
            p_lo[i] = z.x; p_hi[i] = z.y;
        }
    }
    const uint8_t *cp = a.cur.p + (ptrdiff_t)cy * a.cur.stride + cx;
    uint32_t c_lo[8], c_hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint2 c = *reinterpret_cast<const uint2 *>(cp + (ptrdiff_t)i * a.cur.stride);
        c_lo[i] = c.x; c_hi[i] = c.y;
    }
    int diff = weight_quads(c_lo, p_lo) + weight_quads(c_lo + 4, p_lo + 4) + weight_quads(c_hi, p_hi) +
               weight_quads(c_hi + 4, p_hi + 4);
    if (k < 25) diff += (iabs(dx) + iabs(dy)) * 32;  // :1176-1178
    if (a.dbg && live && b == a.dbg_block && k < 26) {
        uint32_t *d = a.dbg + k * 18;
        for (int i = 0; i < 8; ++i) { d[2 * i] = p_lo[i]; d[2 * i + 1] = p_hi[i]; }
        d[16] = (uint32_t)diff;
        d[17] = valid;
    }
    if (a.dbg && live && b == a.dbg_block) {
        for (int i = lane; i < 70; i += 32) a.dbg[26 * 18 + i] = s_win[g][i];
        for (int i = lane; i < 140; i += 32) a.dbg[26 * 18 + 70 + i] = s_H[g][i];
        if (lane == 0) { a.dbg[26 * 18 + 210] = (uint32_t)Lx; a.dbg[26 * 18 + 211] = (uint32_t)Ly; a.dbg[26 * 18 + 212] = (uint32_t)o; }
    }
    uint32_t key = (valid && diff < 0x7fff) ? ((uint32_t)diff << 8) | (uint32_t)k : 0xffffffffu;
    key = halfwave_min(key);
    if (lane == 0 && live) {
        int bqx = (int16_t)(a.w * 4 - 32), bqy = (int16_t)(a.h * 4 - 32), md = 0x7fff;  // :1136-1137
        if (key != 0xffffffffu) {
            const int kk = key & 0xff;
            md = (int)(key >> 8);
            bqx = kk == 25 ? cx * 4 : (int16_t)(cx * 4 + v0x + (kk % 5 - 2));
            bqy = kk == 25 ? cy * 4 : (int16_t)(cy * 4 + v0y + (kk / 5 - 2));
        }
        const int vx = (int16_t)(bqx - cx * 4), vy = (int16_t)(bqy - cy * 4);
        if ((vx != 0) | (vy != 0)) md -= (iabs(vx - v0x) + iabs(vy - v0y)) * 32;  // :1195-1197
        reinterpret_cast<uint32_t *>(a.net_out[r])[b] = (uint32_t)(uint16_t)vx | ((uint32_t)(uint16_t)vy << 16);
        a.bdiff[r][b] = md;
    }
}

void launch_search2_v1(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, uint32_t *dbg, int dbg_block) {
    Search2Args a;
    a.cur = cur.Y[0];
    int n = 0;
    for (int r = 0; r < 3; ++r) {
        a.ref[r] = refs.ref[r].Y[0];
        a.net_in[r] = nets.net[r][1];   // vnet2 holds the 1x result, init.h:832-854
        a.net_out[r] = nets.net[r][0];
        a.bdiff[r] = nets.bdiff[r];
        if (refs.use[r]) a.refmap[n++] = r;
    }
    for (int i = n; i < 3; ++i) a.refmap[i] = 0;
    a.w = a.cur.w;
    a.h = a.cur.h;
    a.bw = a.w / 8;
    a.nblk = a.w * a.h / 64;
    a.dbg = dbg;
    a.dbg_block = dbg_block;
    hipLaunchKernelGGL(k_search2, dim3((a.nblk + 7) / 8, n), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// select_reference (GPU_kernels.cl:1205-1283) + pack_8x8_into_16x16 (:1346-1366), one thread per MB
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_select(NetSet n, MBOut o, int mbw, int mbs, int use_golden, int use_altref) {
    const int mb = blockIdx.x * 256 + threadIdx.x;
    if (mb >= mbs) return;
    const int b8w = mbw * 2;
    const int b = ((mb / mbw) * 2) * b8w + (mb % mbw) * 2;
    const int idx[4] = {b, b + 1, b + b8w, b + b8w + 1};
    int diff1 = n.bdiff[0][idx[0]] + n.bdiff[0][idx[1]] + n.bdiff[0][idx[2]] + n.bdiff[0][idx[3]];
    int diff2 = 0x7fffffff;
    if (use_altref == 1) diff2 = n.bdiff[2][idx[0]] + n.bdiff[2][idx[1]] + n.bdiff[2][idx[2]] + n.bdiff[2][idx[3]];
    int ref = diff1 <= diff2 ? 0 : 2;
    diff1 = diff1 <= diff2 ? diff1 : diff2;
    diff2 = 0x7fffffff;
    if (use_golden == 1) diff2 = n.bdiff[1][idx[0]] + n.bdiff[1][idx[1]] + n.bdiff[1][idx[2]] + n.bdiff[1][idx[3]];
    ref = diff1 <= diff2 ? ref : 1;
    const uint32_t *net = reinterpret_cast<const uint32_t *>(n.net[ref][0]);
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = net[idx[k]];
    o.ref[mb] = ref;
    *reinterpret_cast<uint4 *>(o.vec + 8 * mb) = make_uint4(v[0], v[1], v[2], v[3]);
    o.parts[mb] = (v[1] == v[0] && v[2] == v[0] && v[3] == v[0]) ? 0 : 1;
}

// test tap: the block-match metric on caller-supplied difference blocks (n x 16 ints)
__global__ __launch_bounds__(256) void k_weight_tap(const int32_t *d, int n, int32_t *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = d[i * 16 + k];
    out[i] = weight4x4(v);
}
void launch_weight_tap(hipStream_t s, const int32_t *d, int n, int32_t *out) {
    hipLaunchKernelGGL(k_weight_tap, dim3((n + 255) / 256), dim3(256), 0, s, d, n, out);
}

void launch_select(hipStream_t s, const NetSet &nets, const MBOut &o, int mbw, int mbh, int use_golden, int use_altref) {
    const int mbs = mbw * mbh;
    hipLaunchKernelGGL(k_select, dim3((mbs + 255) / 256), dim3(256), 0, s, nets, o, mbw, mbs, use_golden, use_altref);
}

}  // namespace vp8
