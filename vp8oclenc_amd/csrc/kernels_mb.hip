// kernels_mb.hip -- per-macroblock reconstruction loop of the inter-frame path for gfx950.
//
// One fused kernel replaces the reference's per-frame sequence (inter_part.h:268-378)
//   prepare_predictors_and_residual x9, then for each segment LQ..UQ:
//   dct4x4 x3, wht4x4_iwht4x4, idct4x4 x3, count_SSIM x3, gather_SSIM
// and prepare_filter_mask (loop_filter.h:25-46).  Every step is local to a macroblock (the SSIM
// that gates the next segment pass is the macroblock's own), so the whole segment loop runs inside
// the kernel with predictor, residual and coefficients in registers: the predictor/residual planes
// of the reference never exist in HBM.  Mapping: 32 lanes per macroblock, lane b < 24 owns 4x4
// block b (0-15 Y raster, 16-19 U, 20-23 V) -- the block index of macroblock_coeffs_t.
#include <stdlib.h>
#include <string.h>

#include "vp8hip_dev.h"

namespace vp8 {

// (quantiser tables k_dc_q / k_ac_q and qi(): vp8hip_dev.h)

// C division x / q (truncating toward zero, GPU_kernels.cl:1478-1481) by a quantiser that is the same for a
// whole pass: floor(n/q) = (n*M) >> 24 with M = floor(2^24/q) + 1 is exact for n < 2^15 and 2 <= q < 512
// (error term n*(M*q - 2^24)/(q*2^24) < 2^-9 < 1/q).  Every |coefficient| here is below 2^15 (fdct <= 2040,
// WHT <= 16320) and every quantiser below 512 (y2ac <= 440).  (n*M) >> 24 = the HIGH half of n * (M << 8): one
// v_mul_hi_u32 (M < 2^23 for q >= 2, so M << 8 fits; the 64-bit product by two 24-bit multiplies and a funnel shift was
// seven instructions per coefficient).  q == 1 (the DC of a 16x16 macroblock) is passed through.  M << 8 comes from a
// table: the quantisers are table entries and small functions of them, all below 512.
struct TDiv { uint32_t M8; bool one; };
struct Recip24 { uint32_t m8[512]; };
static constexpr Recip24 make_recip24() {
    Recip24 t{};
    for (uint32_t q = 0; q < 512; ++q) t.m8[q] = q < 2 ? 0u : (((1u << 24) / q + 1u) << 8);
    return t;
}
static __device__ __constant__ const Recip24 k_recip24 = make_recip24();
__device__ __forceinline__ TDiv tdiv_make(int q) { return TDiv{k_recip24.m8[q & 511], q == 1}; }
__device__ __forceinline__ int tdiv(int x, const TDiv &d) {
    const int s = x >> 31;
    const uint32_t n = (uint32_t)((x ^ s) - s);
    const uint32_t r = d.one ? n : __umulhi(n, d.M8);
    return ((int)r ^ s) - s;
}

// `construct`, GPU_kernels.cl:574-774: separable six-tap on a 4x4 block at integer position (ix,iy)
// with 1/8-pel phases (fx,fy).  Of the nine horizontally filtered lines the first six are saturated
// to u8, the last three are narrowed with a plain (uchar) cast (wrap mod 256) -- :702-708,727-733,752-758.
//
// Both passes on v_dot4_i32_i8 (the 32-bit multiply-add form of the first version was a third of the kernel's issue
// cycles): pixels as signed bytes p-128 (every tap set sums to 128: sum p*f = sum (p-128)*f + 128*128), taps pre-shifted
// into the four byte positions a 4-sample row needs (the block is loaded from its own address, so output c starts at byte
// c of the line).  The horizontal results are packed per line, transposed (3 x eight v_perm) into columns whose bytes
// run down the six vertical taps, and the four output columns are transposed back into the rows the transform reads.
// Phase 0 (whole-pel: tap 128 does not fit a signed byte) selects the unfiltered bytes instead, per line / per column.
constexpr uint32_t pk8t(int a, int b, int c, int d) {
    return (uint32_t)(a & 255) | ((uint32_t)(b & 255) << 8) | ((uint32_t)(c & 255) << 16) | ((uint32_t)(d & 255) << 24);
}
// six taps f0..f5 placed at byte offset c = 0..3: entries [2c], [2c+1] and, for c = 3, [8]
#define P6(f0, f1, f2, f3, f4, f5)                                                                          \
    {pk8t(f0, f1, f2, f3), pk8t(f4, f5, 0, 0), pk8t(0, f0, f1, f2), pk8t(f3, f4, f5, 0), pk8t(0, 0, f0, f1),  \
     pk8t(f2, f3, f4, f5), pk8t(0, 0, 0, f0),  pk8t(f1, f2, f3, f4), pk8t(f5, 0, 0, 0),  0, 0, 0}
static __device__ __constant__ const uint32_t K_P6[8][12] = {   // 1/8-pel phases 0..7, GPU_kernels.cl:563-572 (phase 0 never used)
    P6(0, 0, 0, 0, 0, 0),       P6(0, -6, 123, 12, -1, 0), P6(2, -11, 108, 36, -8, 1), P6(0, -9, 93, 50, -6, 0),
    P6(3, -16, 77, 77, -16, 3), P6(0, -6, 50, 93, -9, 0),  P6(1, -8, 36, 108, -11, 2), P6(0, -1, 12, 123, -6, 0)};
constexpr int KB6 = 128 * 128 + 64;   // undo the -128 pixel bias + rounding
__device__ __forceinline__ int d4k(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, true); }
__device__ __forceinline__ int d4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }
// four 6-tap sums of a 12-byte line (three dwords of biased bytes), outputs c = 0..3 start at byte c
__device__ __forceinline__ void six_tap4(uint32_t w0, uint32_t w1, uint32_t w2, const uint32_t t[9], int s[4]) {
    s[0] = d4(w1, t[1], d4k(w0, t[0], KB6));
    s[1] = d4(w1, t[3], d4k(w0, t[2], KB6));
    s[2] = d4(w1, t[5], d4k(w0, t[4], KB6));
    s[3] = d4(w2, t[8], d4(w1, t[7], d4k(w0, t[6], KB6)));
}
__device__ __forceinline__ uint32_t pack_sat4(const int s[4]) {   // sat8(s >> 7) of four sums, byte c = sum c
    const uint32_t lo = __builtin_amdgcn_ashr_pk_u8_i32(s[0], s[1], 7), hi = __builtin_amdgcn_ashr_pk_u8_i32(s[2], s[3], 7);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
// rows out[r] (byte c = column c) of the 4x4 predictor.  CONFORMANT (vp8hip_conformant_stream, NOT the reference): all nine
// lines saturated, as RFC 6386 section 18.3 has it and as every decoder will predict.
template <bool CONFORMANT>
__device__ __forceinline__ void predict4x4(const Plane &rf, int ix, int iy, int fx, int fy, uint32_t out[4]) {
    uint32_t tx[9], ty[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { tx[i] = K_P6[fx][i]; ty[i] = K_P6[fy][i]; }
    uint32_t Hb[9];   // horizontally filtered lines, four biased bytes each
#pragma unroll
    for (int L = 0; L < 9; ++L) {
        const uint8_t *p = rf.p + (ptrdiff_t)(iy - 2 + L) * rf.stride + (ix - 2);
        const uint32_t w0 = ld_u32(p) ^ 0x80808080u, w1 = ld_u32(p + 4) ^ 0x80808080u, w2 = ld_u32(p + 8) ^ 0x80808080u;
        int sm[4];
        six_tap4(w0, w1, w2, tx, sm);
        uint32_t h;
        if (L < 6 || CONFORMANT) {
            h = pack_sat4(sm);                    // saturated lines, :600-680
        } else {                                  // the last three: C division toward zero, then a plain (uchar) cast, :702-758
            h = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) h |= (uint32_t)(div128(sm[c]) & 0xff) << (8 * c);
        }
        h ^= 0x80808080u;
        Hb[L] = fx == 0 ? __builtin_amdgcn_alignbyte(w1, w0, 2) : h;   // whole-pel x: bytes 2..5 of the line as they are
    }
    uint32_t ca[4], cb[4], oc[4];   // columns: rows 0-3, rows 4-7 (row 8 comes from Hb[8])
    transpose4x4(Hb, ca);
    transpose4x4(Hb + 4, cb);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int sm[4];
        six_tap4(ca[c], cb[c], Hb[8] >> (8 * c), ty, sm);   // only byte 0 of the third dword meets a non-zero tap
        const uint32_t v = pack_sat4(sm);
        oc[c] = fy == 0 ? (__builtin_amdgcn_alignbyte(cb[c], ca[c], 2) ^ 0x80808080u) : v;   // whole-pel y: rows 2..5 of the column
    }
    transpose4x4(oc, out);
}

// dct4x4, GPU_kernels.cl:1417-1476: libvpx fdct constants, vertical pass first
__device__ __forceinline__ void fdct4x4(const int in[16], int out[16]) {
    int L[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int r0 = in[c], r1 = in[4 + c], r2 = in[8 + c], r3 = in[12 + c];
        const int a1 = (r0 + r3) * 8, d1 = (r0 - r3) * 8, b1 = (r1 + r2) * 8, c1 = (r1 - r2) * 8;
        L[c] = a1 + b1;
        L[8 + c] = a1 - b1;
        // |c1|,|d1| <= 4080 here and <= 16320 in the row pass: both rotations are one v_dot2_i32_i16 each
        const uint32_t xy = pk16(c1, d1);
        L[4 + c] = dot2(xy, K_ROT_A, 14500) >> 12;     // c1*2217 + d1*5352 + 14500
        L[12 + c] = dot2(xy, K_ROT_B, 7500) >> 12;     // d1*2217 - c1*5352 + 7500
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e0 = L[4 * i], e1 = L[4 * i + 1], e2 = L[4 * i + 2], e3 = L[4 * i + 3];
        const int a1 = e0 + e3, d1 = e0 - e3, b1 = e1 + e2, c1 = e1 - e2;
        out[4 * i] = (a1 + b1 + 7) >> 4;
        out[4 * i + 2] = (a1 - b1 + 7) >> 4;
        const uint32_t xy = pk16(c1, d1);
        out[4 * i + 1] = (dot2(xy, K_ROT_A, 12000) >> 16) + (d1 != 0);
        out[4 * i + 3] = dot2(xy, K_ROT_B, 51000) >> 16;
    }
}

// dequant_and_iDCT, GPU_kernels.cl:192-255 (inputs already dequantised)
__device__ __forceinline__ void idct4x4(int L[16]) {
    int T[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i0 = L[c], i1 = L[4 + c], i2 = L[8 + c], i3 = L[12 + c];
        const int a1 = i0 + i2, b1 = i0 - i2;
        // operands are far below 2^23; the low 32 bits of the 24x24 product are the reference's int product
        const int c1 = (__mul24(i1, 35468) >> 16) - (i3 + (__mul24(i3, 20091) >> 16));
        const int d1 = (i1 + (__mul24(i1, 20091) >> 16)) + (__mul24(i3, 35468) >> 16);
        T[c] = a1 + d1;
        T[12 + c] = a1 - d1;
        T[4 + c] = b1 + c1;
        T[8 + c] = b1 - c1;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i0 = T[4 * r], i1 = T[4 * r + 1], i2 = T[4 * r + 2], i3 = T[4 * r + 3];
        const int a1 = i0 + i2, b1 = i0 - i2;
        // operands are far below 2^23; the low 32 bits of the 24x24 product are the reference's int product
        const int c1 = (__mul24(i1, 35468) >> 16) - (i3 + (__mul24(i3, 20091) >> 16));
        const int d1 = (i1 + (__mul24(i1, 20091) >> 16)) + (__mul24(i3, 35468) >> 16);
        L[4 * r] = (a1 + d1 + 4) >> 3;
        L[4 * r + 3] = (a1 - d1 + 4) >> 3;
        L[4 * r + 1] = (b1 + c1 + 4) >> 3;
        L[4 * r + 2] = (b1 - c1 + 4) >> 3;
    }
}

// WHT_and_quant + dequant_and_iWHT, GPU_kernels.cl:257-401, spread over the sixteen luma lanes of the macroblock: lane
// L = 4r + c holds element (r, c) of the 4x4 block of luma DCs.  Both transforms are two passes of the same four-point
// pattern -- out[i] = v0 + s1(i) v1 + s2(i) v2 + s3(i) v3 with the signs of row i of {++++, ++--, +--+, +-+-} -- first down
// the column, then along the row (forward), first along the row, then down the column (inverse): four gathers of four
// lanes, three multiply-adds each, ONE division per lane.  (Every lane used to gather all sixteen values and run the whole
// round trip with its sixteen divisions: 300 instructions per lane, a seventh of the kernel.)
// x: this lane's DC in; returns this lane's reconstructed DC, q = this lane's element of the quantised second-order block.
__device__ __forceinline__ int had4(int v0, int v1, int v2, int v3, int i) {
    const int s1 = i >= 2 ? -1 : 1, s2 = (i == 1 || i == 2) ? -1 : 1, s3 = (i & 1) ? -1 : 1;
    return v0 + __mul24(s1, v1) + __mul24(s2, v2) + __mul24(s3, v3);
}
// (gathers by ds_bpermute with the eight byte addresses made once: __shfl recomputes its lane arithmetic, four instructions,
// at each of the sixteen calls)
__device__ __forceinline__ int wht_roundtrip_lane(int x, int lane, int dc_q, int ac_q, const TDiv &ddc, const TDiv &dac, int &q) {
    const int L = lane & 15, r = L >> 2, c = L & 3;
    const int base = (int)(threadIdx.x & 32u) * 4;             // byte address (in the wave) of the macroblock's lane 0
    const int col = base + 4 * c, row = base + 4 * (L & 12);   // ... of lane c (top of this lane's column), of lane 4r (head of its row)
    auto gather_col = [&](int v, int i) { return had4(__builtin_amdgcn_ds_bpermute(col, v), __builtin_amdgcn_ds_bpermute(col + 16, v),
                                                      __builtin_amdgcn_ds_bpermute(col + 32, v), __builtin_amdgcn_ds_bpermute(col + 48, v), i); };
    auto gather_row = [&](int v, int i) { return had4(__builtin_amdgcn_ds_bpermute(row, v), __builtin_amdgcn_ds_bpermute(row + 4, v),
                                                      __builtin_amdgcn_ds_bpermute(row + 8, v), __builtin_amdgcn_ds_bpermute(row + 12, v), i); };
    // forward: columns (T[4r + c] from X[c], X[4 + c], X[8 + c], X[12 + c]), then rows
    int o = gather_row(gather_col(x, r), c);
    o += (o > 0);
    o >>= 1;
    q = tdiv(o, L == 0 ? ddc : dac);
    const int xq = __mul24(q, L == 0 ? dc_q : ac_q);
    // inverse: rows, then columns with (. + 3) >> 3
    return (gather_col(gather_row(xq, c), r) + 3) >> 3;
}

// zig-zag position of raster coefficient k: coeff[inv_zigzag[k]] = L[k], GPU_kernels.cl:1489
__device__ __forceinline__ constexpr int inv_zigzag(int k) {
    constexpr int t[16] = {0, 1, 5, 6, 2, 4, 7, 12, 3, 8, 11, 13, 9, 10, 14, 15};
    return t[k];
}
// the same for an index known only at run time: sixteen nibbles
__device__ __forceinline__ int inv_zigzag_rt(int k) { return (int)((0xfea9db83c7426510ull >> (4 * k)) & 15); }
__device__ __forceinline__ void store_zigzag(int16_t *dst, const int L[16]) {
    int z[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) z[inv_zigzag(k)] = L[k];
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = (uint32_t)(uint16_t)z[2 * k] | ((uint32_t)(uint16_t)z[2 * k + 1] << 16);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    d4[0] = make_uint4(w[0], w[1], w[2], w[3]);
    d4[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// Pointers only where a Frame would do: every surface of a context has the same layout (carve_frame), so one stride / size
// pair per plane kind serves the five frames, and eight contexts' blocks fit the 4 KiB of a batched launch's arguments
// (five Frames with their pyramids were 840 B of a 1 KB block).
struct MBPlanes { uint8_t *y, *u, *v; };
struct MBArgs {
    MBPlanes cur, ref0, ref1, ref2, recon;   // (single members, not arrays: a three-way select over an array member becomes a
                                             // dynamic index, and a dynamically indexed argument block is copied to scratch memory)
    int ystride, cstride, yw, yh, cw, ch;
    int32_t *o_parts, *o_ref, *o_seg, *o_nz, *o_mask;
    int16_t *o_vec, *o_coeffs;
    float *o_ssim;
    int32_t *o_flag;      // set when a macroblock ends below the SSIM target: check_SSIM's fallback has work (zero at rest)
    const SegData *sd;
    const int32_t *bdiff0, *bdiff1, *bdiff2;
    const int16_t *vnet0, *vnet1, *vnet2;
    float ssim_target;
    int mbw, mbs;
    int use_golden, use_altref;
};

// LDS per macroblock: current and reconstructed pixels, plane-major (Y 16x16, U 8x8, V 8x8)
// + two float arrays of per-pixel SSIM terms, stored in the order the reference's float4 lanes accumulate them:
// plane base (0 / 256 / 320) + component * n/4 + sequence index, so a chain reads consecutive floats.
struct MBTile { uint8_t cur[384]; uint8_t rec[384]; float fa[384]; float fb[384]; };

// sums of the pixels of a 384-byte tile: Y over the 32 lanes, U in lanes 0-15, V in lanes 16-31 (exact
// integers, so the means equal the reference's float accumulation, which is exact too: sums < 2^24)
__device__ __forceinline__ void tile_sums(const uint32_t *t32, int lane, int &sy, int &sc) {
    sy = (int)__builtin_amdgcn_sad_u8(t32[lane], 0u, __builtin_amdgcn_sad_u8(t32[lane + 32], 0u, 0u));
    sc = (int)__builtin_amdgcn_sad_u8(t32[64 + lane], 0u, 0u);
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) sy += __shfl_xor(sy, m, 32);
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) sc += __shfl_xor(sc, m, 32);
}
// one accumulation chain of n4 terms in the reference's order (first term assigned, then term + acc)
__device__ __forceinline__ float chain(const float *p, int n4) {
    float acc = 0.0f;
    for (int q = 0; q < n4; q += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(p + q);
        acc = q == 0 ? v.x : __fadd_rn(v.x, acc);
        acc = __fadd_rn(v.y, acc);
        acc = __fadd_rn(v.z, acc);
        acc = __fadd_rn(v.w, acc);
    }
    return acc;
}
__device__ __forceinline__ float sum4(float acc, int lane) {   // (((c0 + c1) + c2) + c3) of the four components
    const float s0 = __shfl(acc, (lane & ~3) + 0, 32), s1 = __shfl(acc, (lane & ~3) + 1, 32);
    const float s2 = __shfl(acc, (lane & ~3) + 2, 32), s3 = __shfl(acc, (lane & ~3) + 3, 32);
    return __fadd_rn(__fadd_rn(__fadd_rn(s0, s1), s2), s3);
}

// SSIM in the reference's float4 order (count_SSIM_luma/_chroma, :1610-2095): lane-component `comp` accumulates
// pixels (r, 4j+comp), r outer, j inner; n = 16 (luma) or 8 (chroma).  The per-pixel terms (deviation squares and
// products) are computed by all 32 lanes and parked in LDS in chain order; only the 64- (16-) term accumulation
// chains, whose order is the result, run on the 12 chain lanes.  All float ops are written so that nothing can
// be contracted (the reference's mad is a*b+c, unfused).
// All LDS traffic of a macroblock stays inside its own 32 lanes = half of one wave, so program order is the
// only synchronisation needed: there is no workgroup barrier in this kernel.
template <bool CONFORMANT>
__device__ __forceinline__ void mb_body(const MBArgs &a, MBTile *s_t) {
    const int g = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const int mb_raw = xcd_band(blockIdx.x, gridDim.x) * 8 + g;   // an XCD's workgroups on one band of the frame (vp8hip_dev.h)
    const bool live = mb_raw < a.mbs;
    const int mb = live ? mb_raw : a.mbs - 1;
    const int mbx = mb % a.mbw, mby = mb / a.mbw;
    const int32_t *SD = a.sd->v;

    const bool blk = lane < 24;
    const int plane = lane < 16 ? 0 : (lane < 20 ? 1 : 2);
    const int bi = lane < 16 ? lane : (lane - 16) & 3;       // block index inside its plane
    const int bw = plane == 0 ? 4 : 2;                        // blocks per MB row of the plane
    const int bx = blk ? bi % bw : 0, by = blk ? bi / bw : 0;
    const int msz = plane == 0 ? 16 : 8;
    const int posx = mbx * msz + bx * 4, posy = mby * msz + by * 4;
    // select_reference (GPU_kernels.cl:1205-1283) + pack_8x8_into_16x16 (:1346-1366), per macroblock:
    // cheapest of the three references by the sum of its four block costs (ties: LAST over ALTREF, then
    // that over GOLDEN), its four vectors, 16x16 iff they are equal.  Every lane evaluates it (same
    // addresses across the 32 lanes: one broadcast load each), lane 0 stores the outputs.
    const int b8w = a.mbw * 2;
    const int cell0 = (mby * 2) * b8w + mbx * 2;
    const int cidx[4] = {cell0, cell0 + 1, cell0 + b8w, cell0 + b8w + 1};
    int diff1 = a.bdiff0[cidx[0]] + a.bdiff0[cidx[1]] + a.bdiff0[cidx[2]] + a.bdiff0[cidx[3]];
    int diff2 = 0x7fffffff;
    if (a.use_altref == 1)
        diff2 = a.bdiff2[cidx[0]] + a.bdiff2[cidx[1]] + a.bdiff2[cidx[2]] + a.bdiff2[cidx[3]];
    int ref = diff1 <= diff2 ? 0 : 2;
    diff1 = diff1 <= diff2 ? diff1 : diff2;
    diff2 = 0x7fffffff;
    if (a.use_golden == 1)
        diff2 = a.bdiff1[cidx[0]] + a.bdiff1[cidx[1]] + a.bdiff1[cidx[2]] + a.bdiff1[cidx[3]];
    ref = diff1 <= diff2 ? ref : 1;
    const uint32_t *vnet = reinterpret_cast<const uint32_t *>(ref == 0 ? a.vnet0 : (ref == 1 ? a.vnet1 : a.vnet2));
    uint32_t mbv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mbv[k] = vnet[cidx[k]];
    const int parts = (mbv[1] == mbv[0] && mbv[2] == mbv[0] && mbv[3] == mbv[0]) ? 0 : 1;
    if (lane == 0 && live) {
        a.o_ref[mb] = ref;
        a.o_parts[mb] = parts;
        *reinterpret_cast<uint4 *>(a.o_vec + 8 * mb) = make_uint4(mbv[0], mbv[1], mbv[2], mbv[3]);
    }
    // (pointer by pointer: selecting one of three argument STRUCTS by a per-lane index sends the argument block to scratch memory)
    uint8_t *const rpy = ref == 0 ? a.ref0.y : (ref == 1 ? a.ref1.y : a.ref2.y);
    uint8_t *const rpu = ref == 0 ? a.ref0.u : (ref == 1 ? a.ref1.u : a.ref2.u);
    uint8_t *const rpv = ref == 0 ? a.ref0.v : (ref == 1 ? a.ref1.v : a.ref2.v);
    const int pstride = plane == 0 ? a.ystride : a.cstride, pw = plane == 0 ? a.yw : a.cw, ph = plane == 0 ? a.yh : a.ch;
    const Plane cp{plane == 0 ? a.cur.y : (plane == 1 ? a.cur.u : a.cur.v), pstride, pw, ph};
    const Plane rc{plane == 0 ? a.recon.y : (plane == 1 ? a.recon.u : a.recon.v), pstride, pw, ph};
    const Plane rp{plane == 0 ? rpy : (plane == 1 ? rpu : rpv), pstride, pw, ph};
    const int tile_off = plane == 0 ? 0 : (plane == 1 ? 256 : 320);

    // The predictor stays packed (four dwords of bytes) and the current block lives in the LDS tile: a pass re-forms the
    // residual from them, and everything a pass produces (coefficients, block 24, reconstruction) is stored as the pass
    // produces it -- the last pass to run is the one that counts, GPU_kernels.cl:1391 -- so that none of the 16-entry
    // arrays stays alive across the float SSIM section (193 VGPRs and two waves per SIMD before; see k_mb below).
    uint32_t predw[4] = {0, 0, 0, 0};
    if (blk) {
        // prepare_predictors_and_residual, :1285-1344: vector of the block's 8x8 quadrant
        const int quad = plane == 0 ? (by >> 1) * 2 + (bx >> 1) : by * 2 + bx;
        const uint32_t vv = quad == 0 ? mbv[0] : (quad == 1 ? mbv[1] : (quad == 2 ? mbv[2] : mbv[3]));
        const int vx = (int16_t)(vv & 0xffffu), vy = (int16_t)(vv >> 16);
        const int gsh = plane == 0 ? 2 : 3, gm = plane == 0 ? 3 : 7;
        const int fxp = posx * (gm + 1) + vx, fyp = posy * (gm + 1) + vy;  // >= 0 for every in-frame vector
        const int dx = (fxp & gm) * (plane == 0 ? 2 : 1), dy = (fyp & gm) * (plane == 0 ? 2 : 1);
        const int ix = iclamp(fxp >> gsh, 2 - EXT, rp.w + EXT - 7), iy = iclamp(fyp >> gsh, 2 - EXT, rp.h + EXT - 7);
        predict4x4<CONFORMANT>(rp, ix, iy, dx, dy, predw);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t cw = *reinterpret_cast<const uint32_t *>(cp.p + (ptrdiff_t)(posy + r) * cp.stride + posx);
            *reinterpret_cast<uint32_t *>(&s_t[g].cur[tile_off + (by * 4 + r) * msz + bx * 4]) = cw;
        }
    }
    // ---- SSIM terms of the current frame: mean (exact in float) and variance chain ---------------
    // chain lanes 0..11: plane sp = lane>>2, float4 component comp = lane&3
    const int sp = (lane >> 2) < 3 ? (lane >> 2) : 2, comp = lane & 3;
    const int n4 = sp == 0 ? 64 : 16;                                       // terms per chain
    const int chain_off = (sp == 0 ? 0 : (sp == 1 ? 256 : 320)) + comp * n4;
    const float area = sp == 0 ? 256.0f : 64.0f;
    // term lanes (all 32): Y dwords `lane` and `lane+32` (sequence index = dword index), chroma dword `lane&15`
    // of U (lanes 0-15) or V (16-31); byte k of a dword belongs to component k
    const uint32_t *cur32 = reinterpret_cast<const uint32_t *>(s_t[g].cur);
    const uint32_t *rec32 = reinterpret_cast<const uint32_t *>(s_t[g].rec);
    const int coff = (lane < 16 ? 256 : 320) + (lane & 15);
    float *fa = s_t[g].fa, *fb = s_t[g].fb;
    float M1y, M1c, M1, D1;
    {
        int sy, sc;
        tile_sums(cur32, lane, sy, sc);
        M1y = __fdiv_rn((float)sy, 256.0f);
        M1c = __fdiv_rn((float)sc, 64.0f);
        const float mu = __shfl(M1c, 0, 32), mv = __shfl(M1c, 16, 32);
        M1 = sp == 0 ? M1y : (sp == 1 ? mu : mv);
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const uint32_t w = cur32[h < 2 ? lane + 32 * h : 64 + lane];
            const float m = h < 2 ? M1y : M1c;
            const int off = h < 2 ? lane + 32 * h : coff, cs = h < 2 ? 64 : 16;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = __fsub_rn((float)byte_of(w, k), m);
                fa[off + k * cs] = __fmul_rn(t, t);
            }
        }
        D1 = __fdiv_rn(sum4(chain(fa + chain_off, n4), lane), area);
    }

    float ssim = -2.0f;  // pack_8x8_into_16x16, :1352
    int seg_final = a.o_seg[mb];
    int nz_blk = 0;            // this lane's share of prepare_filter_mask's count, from the last pass that ran
    bool any_pass = false;

    for (int seg = 3; seg >= 0; --seg) {        // inter_part.h:329
        // dct4x4 gate, :1391 (same value in all 32 lanes of a macroblock; the two macroblocks of a wave may
        // leave the loop at different passes)
        if (ssim > a.ssim_target) break;
        {
        any_pass = true;
        seg_final = seg;
        const int i = SD[seg * SD_INTS + SD_Y_AC_I];
        int dc_q, ac_q;
        if (plane == 0) {                        // :1394-1408
            ac_q = k_ac_q[i];
            dc_q = parts == 0 ? 1 : k_dc_q[qi(SD[SD_Y_DC_IDELTA] + i)];
        } else {
            dc_q = imin(k_dc_q[qi(SD[SD_UV_DC_IDELTA] + i)], 132);
            ac_q = k_ac_q[qi(SD[SD_UV_AC_IDELTA] + i)];
        }
        int coef[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) coef[k] = 0;
        if (blk) {
            int res[16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t cw = *reinterpret_cast<const uint32_t *>(&s_t[g].cur[tile_off + (by * 4 + r) * msz + bx * 4]);
#pragma unroll
                for (int c = 0; c < 4; ++c) res[4 * r + c] = byte_of(cw, c) - byte_of(predw[r], c);
            }
            fdct4x4(res, coef);
            const TDiv ddc = tdiv_make(dc_q), dac = tdiv_make(ac_q);
            coef[0] = tdiv(coef[0], ddc);        // truncating, :1478-1481
#pragma unroll
            for (int k = 1; k < 16; ++k) coef[k] = tdiv(coef[k], dac);
        }
        nz_blk = 0;
        if (parts == 0) {                        // wht4x4_iwht4x4, :1498-1543 (every lane takes part in the gathers)
            const int y2dc = k_dc_q[qi(SD[SD_Y2_DC_IDELTA] + i)] * 2;
            const int y2ac = imax(31 * k_ac_q[qi(SD[SD_Y2_AC_IDELTA] + i)] / 20, 8);
            int q24 = 0;
            const int nd = wht_roundtrip_lane((int16_t)coef[0], lane, y2dc, y2ac, tdiv_make(y2dc), tdiv_make(y2ac), q24);
            if (lane < 16) {
                coef[0] = (int16_t)nd;           // stored as short, :1537
                // block 24, element by element, and its share of the non-zero count (CPU_kernels.cl:800-819)
                if (live) a.o_coeffs[((size_t)mb * 25 + 24) * 16 + inv_zigzag_rt(lane)] = (int16_t)q24;
                nz_blk += iabs((int16_t)q24);
            }
        }
        if (blk) {                               // idct4x4, :1545-1608
            if (live) store_zigzag(a.o_coeffs + ((size_t)mb * 25 + lane) * 16, coef);
            // prepare_filter_mask, CPU_kernels.cl:800-819
#pragma unroll
            for (int k = 1; k < 16; ++k) nz_blk += iabs((int16_t)coef[k]);
            if (plane != 0 || parts != 0) nz_blk += iabs((int16_t)coef[0]);
            int L[16];
            L[0] = __mul24((int16_t)coef[0], dc_q);
#pragma unroll
            for (int k = 1; k < 16; ++k) L[k] = __mul24((int16_t)coef[k], ac_q);
            idct4x4(L);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uint32_t w = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) w |= (uint32_t)sat8(L[4 * r + c] + byte_of(predw[r], c)) << (8 * c);
                *reinterpret_cast<uint32_t *>(&s_t[g].rec[tile_off + (by * 4 + r) * msz + bx * 4]) = w;
                if (live) *reinterpret_cast<uint32_t *>(rc.p + (ptrdiff_t)(posy + r) * rc.stride + posx) = w;
            }
        }
        }
        {
        // count_SSIM_*, gather_SSIM
        float M2, D, C;
        {
            int sy, sc;
            tile_sums(rec32, lane, sy, sc);
            const float M2y = __fdiv_rn((float)sy, 256.0f), M2c = __fdiv_rn((float)sc, 64.0f);
            const float mu = __shfl(M2c, 0, 32), mv = __shfl(M2c, 16, 32);
            M2 = sp == 0 ? M2y : (sp == 1 ? mu : mv);
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const int di = h < 2 ? lane + 32 * h : 64 + lane;
                const uint32_t wc = cur32[di], wr = rec32[di];
                const float m1 = h < 2 ? M1y : M1c, m2 = h < 2 ? M2y : M2c;
                const int off = h < 2 ? lane + 32 * h : coff, cs = h < 2 ? 64 : 16;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t1 = __fsub_rn((float)byte_of(wc, k), m1);
                    const float t2 = __fsub_rn((float)byte_of(wr, k), m2);
                    fa[off + k * cs] = __fmul_rn(t2, t2);
                    fb[off + k * cs] = __fmul_rn(t1, t2);
                }
            }
            D = __fadd_rn(D1, __fdiv_rn(sum4(chain(fa + chain_off, n4), lane), area));
            C = __fdiv_rn(sum4(chain(fb + chain_off, n4), lane), area);
        }
        const float k1 = 0.01f * 0.01f * 255 * 255, k2 = 0.03f * 0.03f * 255 * 255;
        const float num = __fmul_rn(__fadd_rn(__fmul_rn(M1, __fmul_rn(M2, 2.0f)), k1), __fadd_rn(__fmul_rn(C, 2.0f), k2));
        const float den = __fmul_rn(__fadd_rn(__fmul_rn(M1, M1), __fadd_rn(__fmul_rn(M2, M2), k1)), __fadd_rn(D, k2));
        float metric = __fdiv_rn(num, den);
        float dm = __fsub_rn(M1, M2);
        dm = dm < 0 ? -dm : dm;
        dm = dm > 4 ? __fmul_rn(0.02f, dm) : 0.0f;
        metric = __fsub_rn(metric, dm);
        const float m1 = __shfl(metric, 0, 32), m2 = __shfl(metric, 4, 32), m3 = __shfl(metric, 8, 32);
        ssim = __fdiv_rn(__fadd_rn(__fadd_rn(m1, m2), m3), 3.0f);
        }
    }

    // ---- results -------------------------------------------------------------------------------
    int nz = any_pass ? nz_blk : 0;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) nz += __shfl_xor(nz, m, 32);
    if (lane == 0 && live) {
        if (any_pass) {
            a.o_seg[mb] = seg_final;
            a.o_nz[mb] = nz;
            a.o_mask[mb] = (parts != 0 || nz > 0) ? -1 : 0;
        }
        a.o_ssim[mb] = ssim;
        if (ssim < a.ssim_target) *a.o_flag = 1;   // (vp8enc.cpp:244: what the fallback acts on; rare, so no store otherwise)
    }
}

// ---- the same macroblock loop with every lane of the workgroup on a 4x4 block ---------------------------------------------------
// The kernel is bound by what its waves ISSUE, and in the form above a wave issues the block phase (prediction, transforms,
// quantisation, reconstruction: two thirds of its instructions) for 2 x 24 blocks on 64 lanes, and each of its three accumulation
// chains (16 dependent steps of four additions) for 2 x 12 chains.  Here a workgroup of THREE waves takes eight macroblocks:
//   waves 0, 1   the luma blocks of macroblocks 0-3 / 4-7 (16 lanes per macroblock = one DPP row: its sums are row reductions);
//   wave 2       the chroma blocks of all eight (8 lanes per macroblock: U 0-3, V 0-3);
// every lane owns a block, so eight macroblocks take three block phases where they took four.  A plane's SSIM terms are made by
// the lanes of its blocks (the pixels they read are their own wave's), and the chains are dealt over the workgroup by LENGTH:
// the 2 x 32 luma chains (64 terms: `variance' and `covariance' of 8 macroblocks x 4 components) are wave 0, the 2 x 64 chroma
// chains (16 terms) waves 1 and 2 -- 24 chain steps of a wave per eight macroblocks where there were 128.  The three planes of a
// macroblock now live in different waves: they meet in LDS (means, variances, the plane's metric, the chroma blocks' share of the
// non-zero count) behind two workgroup barriers per pass, and the pass loop ends for the whole workgroup when a pass found no
// macroblock below the target (a word in LDS the active macroblocks set before the first barrier).  A macroblock that has
// left the loop is masked out of the block phase; its chains are summed again from the terms its last pass left (the same bits).
// Same operations on the same values in the same order per macroblock: the outputs are the form above's, bit for bit.
struct MBShare {
    float M1[8][4], M2[8][4], D1[8][4], metric[8][4];   // [macroblock slot][plane]
    int nzc[8];                                         // the chroma blocks' share of prepare_filter_mask's count
    int any[2];                                         // [pass & 1]: a macroblock of the workgroup ran this pass
};
struct MBTileP { MBTile t; uint32_t skew[4]; };          // (a tile is 60 x 64 B: without the 16 bytes the chains of eight macroblocks read the same banks)
// (every pattern used here gives every lane a source; with "old" a constant hipcc folds the move into the addition that follows: v_add_u32_dpp)
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ int quad_sum(int v) {         // over the four lanes of a quad, in every lane
    v += dpp_i<0xb1>(v);                                 // quad_perm [1,0,3,2]
    return v + dpp_i<0x4e>(v);                           // quad_perm [2,3,0,1]
}
__device__ __forceinline__ int half_row_sum(int v) { v = quad_sum(v); return v + dpp_i<0x141>(v); }   // eight lanes: + row_half_mirror
__device__ __forceinline__ int row_sum(int v) { v = half_row_sum(v); return v + dpp_i<0x140>(v); }     // sixteen lanes: + row_mirror
__device__ __forceinline__ float sum4_quad(float acc) {  // (((c0 + c1) + c2) + c3) of the quad's four components, in every lane
    return __fadd_rn(__fadd_rn(__fadd_rn(dpp_f<0x00>(acc), dpp_f<0x55>(acc)), dpp_f<0xaa>(acc)), dpp_f<0xff>(acc));
}
template <int N4>
__device__ __forceinline__ float chain_n(const float *p) {
    float acc = 0.0f;
#pragma unroll 4
    for (int q = 0; q < N4; q += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(p + q);
        acc = q == 0 ? v.x : __fadd_rn(v.x, acc);
        acc = __fadd_rn(v.y, acc);
        acc = __fadd_rn(v.z, acc);
        acc = __fadd_rn(v.w, acc);
    }
    return acc;
}
// wht_roundtrip_lane for a macroblock whose sixteen luma lanes are lanes base16 .. base16 + 15 of the wave
__device__ __forceinline__ int wht_roundtrip_row(int x, int L, int base_bytes, int dc_q, int ac_q, const TDiv &ddc, const TDiv &dac, int &q) {
    const int r = L >> 2, c = L & 3;
    const int col = base_bytes + 4 * c, row = base_bytes + 4 * (L & 12);
    auto gather_col = [&](int v, int i) { return had4(__builtin_amdgcn_ds_bpermute(col, v), __builtin_amdgcn_ds_bpermute(col + 16, v),
                                                      __builtin_amdgcn_ds_bpermute(col + 32, v), __builtin_amdgcn_ds_bpermute(col + 48, v), i); };
    auto gather_row = [&](int v, int i) { return had4(__builtin_amdgcn_ds_bpermute(row, v), __builtin_amdgcn_ds_bpermute(row + 4, v),
                                                      __builtin_amdgcn_ds_bpermute(row + 8, v), __builtin_amdgcn_ds_bpermute(row + 12, v), i); };
    int o = gather_row(gather_col(x, r), c);
    o += (o > 0);
    o >>= 1;
    q = tdiv(o, L == 0 ? ddc : dac);
    const int xq = __mul24(q, L == 0 ? dc_q : ac_q);
    return (gather_col(gather_row(xq, c), r) + 3) >> 3;
}

template <bool CONFORMANT>
__device__ __forceinline__ void mb_body_packed(const MBArgs &a, MBTileP *s_t, MBShare *s_x) {
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wl = threadIdx.x & 63;
    const bool luma = w < 2;
    const int g = luma ? w * 4 + (wl >> 4) : wl >> 3;          // the macroblock's slot in the workgroup
    const int plane = luma ? 0 : 1 + ((wl >> 2) & 1);
    const int bi = luma ? wl & 15 : wl & 3;                     // block index inside its plane
    const int blkidx = luma ? bi : 12 + 4 * plane + bi;        // ... of macroblock_coeffs_t: 0-15 Y, 16-19 U, 20-23 V
    const int mb_raw = xcd_band(blockIdx.x, gridDim.x) * 8 + g;
    const bool live = mb_raw < a.mbs;
    const int mb = live ? mb_raw : a.mbs - 1;
    const int mbx = mb % a.mbw, mby = mb / a.mbw;
    const int32_t *SD = a.sd->v;
    const int bw = luma ? 4 : 2;
    const int bx = bi % bw, by = bi / bw;
    const int msz = luma ? 16 : 8;
    const int posx = mbx * msz + bx * 4, posy = mby * msz + by * 4;
    if (threadIdx.x == 0) s_x->any[0] = s_x->any[1] = 0;
    // select_reference + pack_8x8_into_16x16, as in mb_body: every lane evaluates its macroblock's
    const int b8w = a.mbw * 2;
    const int cell0 = (mby * 2) * b8w + mbx * 2;
    const int cidx[4] = {cell0, cell0 + 1, cell0 + b8w, cell0 + b8w + 1};
    int diff1 = a.bdiff0[cidx[0]] + a.bdiff0[cidx[1]] + a.bdiff0[cidx[2]] + a.bdiff0[cidx[3]];
    int diff2 = 0x7fffffff;
    if (a.use_altref == 1)
        diff2 = a.bdiff2[cidx[0]] + a.bdiff2[cidx[1]] + a.bdiff2[cidx[2]] + a.bdiff2[cidx[3]];
    int ref = diff1 <= diff2 ? 0 : 2;
    diff1 = diff1 <= diff2 ? diff1 : diff2;
    diff2 = 0x7fffffff;
    if (a.use_golden == 1)
        diff2 = a.bdiff1[cidx[0]] + a.bdiff1[cidx[1]] + a.bdiff1[cidx[2]] + a.bdiff1[cidx[3]];
    ref = diff1 <= diff2 ? ref : 1;
    const uint32_t *vnet = reinterpret_cast<const uint32_t *>(ref == 0 ? a.vnet0 : (ref == 1 ? a.vnet1 : a.vnet2));
    uint32_t mbv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mbv[k] = vnet[cidx[k]];
    const int parts = (mbv[1] == mbv[0] && mbv[2] == mbv[0] && mbv[3] == mbv[0]) ? 0 : 1;
    const bool head = luma && bi == 0;                          // the lane that stores its macroblock's outputs
    if (head && live) {
        a.o_ref[mb] = ref;
        a.o_parts[mb] = parts;
        *reinterpret_cast<uint4 *>(a.o_vec + 8 * mb) = make_uint4(mbv[0], mbv[1], mbv[2], mbv[3]);
    }
    uint8_t *const rpy = ref == 0 ? a.ref0.y : (ref == 1 ? a.ref1.y : a.ref2.y);
    uint8_t *const rpu = ref == 0 ? a.ref0.u : (ref == 1 ? a.ref1.u : a.ref2.u);
    uint8_t *const rpv = ref == 0 ? a.ref0.v : (ref == 1 ? a.ref1.v : a.ref2.v);
    const int pstride = luma ? a.ystride : a.cstride, pw = luma ? a.yw : a.cw, ph = luma ? a.yh : a.ch;
    const Plane cp{plane == 0 ? a.cur.y : (plane == 1 ? a.cur.u : a.cur.v), pstride, pw, ph};
    const Plane rc{plane == 0 ? a.recon.y : (plane == 1 ? a.recon.u : a.recon.v), pstride, pw, ph};
    const Plane rp{plane == 0 ? rpy : (plane == 1 ? rpu : rpv), pstride, pw, ph};
    const int tile_off = plane == 0 ? 0 : (plane == 1 ? 256 : 320);
    MBTile &T = s_t[g].t;

    uint32_t predw[4];
    {
        const int quad = luma ? (by >> 1) * 2 + (bx >> 1) : by * 2 + bx;
        const uint32_t vv = quad == 0 ? mbv[0] : (quad == 1 ? mbv[1] : (quad == 2 ? mbv[2] : mbv[3]));
        const int vx = (int16_t)(vv & 0xffffu), vy = (int16_t)(vv >> 16);
        const int gsh = luma ? 2 : 3, gm = luma ? 3 : 7;
        const int fxp = posx * (gm + 1) + vx, fyp = posy * (gm + 1) + vy;
        const int dx = (fxp & gm) * (luma ? 2 : 1), dy = (fyp & gm) * (luma ? 2 : 1);
        const int ix = iclamp(fxp >> gsh, 2 - EXT, rp.w + EXT - 7), iy = iclamp(fyp >> gsh, 2 - EXT, rp.h + EXT - 7);
        predict4x4<CONFORMANT>(rp, ix, iy, dx, dy, predw);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t cw = *reinterpret_cast<const uint32_t *>(cp.p + (ptrdiff_t)(posy + r) * cp.stride + posx);
            *reinterpret_cast<uint32_t *>(&T.cur[tile_off + (by * 4 + r) * msz + bx * 4]) = cw;
        }
    }
    // ---- SSIM terms of the current frame, by the lanes of the plane: dwords bi + dstep * h of its 64 (16) ----
    const uint32_t *cur32 = reinterpret_cast<const uint32_t *>(T.cur) + (tile_off >> 2);
    const uint32_t *rec32 = reinterpret_cast<const uint32_t *>(T.rec) + (tile_off >> 2);
    const int dstep = luma ? 16 : 4, cs = luma ? 64 : 16;   // the plane's dwords per lane step; floats per float4 component of its chains
    const float area = luma ? 256.0f : 64.0f;
    float *fa = T.fa + tile_off + bi, *fb = T.fb + tile_off + bi;
    auto plane_sum = [&](const uint32_t *t32) {
        uint32_t s = 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) s = __builtin_amdgcn_sad_u8(t32[bi + dstep * h], 0u, s);
        return luma ? row_sum((int)s) : quad_sum((int)s);
    };
    const float M1 = __fdiv_rn((float)plane_sum(cur32), area);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const uint32_t wd = cur32[bi + dstep * h];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = __fsub_rn((float)byte_of(wd, k), M1);
            fa[dstep * h + k * cs] = __fmul_rn(t, t);
        }
    }
    if (bi == 0) s_x->M1[g][plane] = M1;
    __syncthreads();
    if (w == 0) {
        if (wl < 32) {
            const int m = wl >> 2, comp = wl & 3;
            const float s = sum4_quad(chain_n<64>(s_t[m].t.fa + comp * 64));
            if (comp == 0) s_x->D1[m][0] = __fdiv_rn(s, 256.0f);
        }
    } else if (w == 1) {
        const int m = wl >> 3, p = 1 + ((wl >> 2) & 1), comp = wl & 3;
        const float s = sum4_quad(chain_n<16>(s_t[m].t.fa + (p == 1 ? 256 : 320) + comp * 16));
        if (comp == 0) s_x->D1[m][p] = __fdiv_rn(s, 64.0f);
    }
    __syncthreads();

    float ssim = -2.0f;  // pack_8x8_into_16x16, :1352
    int seg_final = a.o_seg[mb];
    int nz_y = 0, nz_c = 0;    // prepare_filter_mask's count, luma and chroma blocks, from the last pass this macroblock ran
    bool any_pass = false;
    for (int seg = 3, round = 0; seg >= 0; --seg, ++round) {        // inter_part.h:329
        const bool act = !(ssim > a.ssim_target);                   // dct4x4 gate, :1391
        if (act) {
            any_pass = true;
            seg_final = seg;
            const int i = SD[seg * SD_INTS + SD_Y_AC_I];
            int dc_q, ac_q;
            if (luma) {                              // :1394-1408
                ac_q = k_ac_q[i];
                dc_q = parts == 0 ? 1 : k_dc_q[qi(SD[SD_Y_DC_IDELTA] + i)];
            } else {
                dc_q = imin(k_dc_q[qi(SD[SD_UV_DC_IDELTA] + i)], 132);
                ac_q = k_ac_q[qi(SD[SD_UV_AC_IDELTA] + i)];
            }
            int coef[16];
            {
                int res[16];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t cw = *reinterpret_cast<const uint32_t *>(&T.cur[tile_off + (by * 4 + r) * msz + bx * 4]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) res[4 * r + c] = byte_of(cw, c) - byte_of(predw[r], c);
                }
                fdct4x4(res, coef);
                const TDiv ddc = tdiv_make(dc_q), dac = tdiv_make(ac_q);
                coef[0] = tdiv(coef[0], ddc);        // truncating, :1478-1481
#pragma unroll
                for (int k = 1; k < 16; ++k) coef[k] = tdiv(coef[k], dac);
            }
            int nz_blk = 0;
            if (luma && parts == 0) {                // wht4x4_iwht4x4, :1498-1543: the sixteen lanes of the macroblock's row
                const int y2dc = k_dc_q[qi(SD[SD_Y2_DC_IDELTA] + i)] * 2;
                const int y2ac = imax(31 * k_ac_q[qi(SD[SD_Y2_AC_IDELTA] + i)] / 20, 8);
                int q24 = 0;
                const int nd = wht_roundtrip_row((int16_t)coef[0], bi, (wl & 48) * 4, y2dc, y2ac, tdiv_make(y2dc), tdiv_make(y2ac), q24);
                coef[0] = (int16_t)nd;               // stored as short, :1537
                if (live) a.o_coeffs[((size_t)mb * 25 + 24) * 16 + inv_zigzag_rt(bi)] = (int16_t)q24;
                nz_blk += iabs((int16_t)q24);
            }
            {                                        // idct4x4, :1545-1608
                if (live) store_zigzag(a.o_coeffs + ((size_t)mb * 25 + blkidx) * 16, coef);
#pragma unroll
                for (int k = 1; k < 16; ++k) nz_blk += iabs((int16_t)coef[k]);
                if (!luma || parts != 0) nz_blk += iabs((int16_t)coef[0]);
                int L[16];
                L[0] = __mul24((int16_t)coef[0], dc_q);
#pragma unroll
                for (int k = 1; k < 16; ++k) L[k] = __mul24((int16_t)coef[k], ac_q);
                idct4x4(L);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uint32_t wd = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) wd |= (uint32_t)sat8(L[4 * r + c] + byte_of(predw[r], c)) << (8 * c);
                    *reinterpret_cast<uint32_t *>(&T.rec[tile_off + (by * 4 + r) * msz + bx * 4]) = wd;
                    if (live) *reinterpret_cast<uint32_t *>(rc.p + (ptrdiff_t)(posy + r) * rc.stride + posx) = wd;
                }
            }
            if (luma) nz_y = row_sum(nz_blk);
            else {
                const int n = half_row_sum(nz_blk);
                if ((wl & 7) == 0) s_x->nzc[g] = n;
            }
            // count_SSIM_*: the plane's mean and its terms
            const float M2 = __fdiv_rn((float)plane_sum(rec32), area);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const uint32_t wc = cur32[bi + dstep * h], wr = rec32[bi + dstep * h];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t1 = __fsub_rn((float)byte_of(wc, k), M1);
                    const float t2 = __fsub_rn((float)byte_of(wr, k), M2);
                    fa[dstep * h + k * cs] = __fmul_rn(t2, t2);
                    fb[dstep * h + k * cs] = __fmul_rn(t1, t2);
                }
            }
            if (bi == 0) {
                s_x->M2[g][plane] = M2;
                s_x->any[round & 1] = 1;
            }
        }
        __syncthreads();
        if (!s_x->any[round & 1]) break;             // (the same word for every lane of the workgroup: they all leave, or none)
        if (threadIdx.x == 0) s_x->any[(round + 1) & 1] = 0;
        {   // the chains by length: wave 0 the luma chains of the eight macroblocks, waves 1 and 2 the chroma chains of four each;
            // lanes 0-31 the squares (`variance'), 32-63 the products (`covariance')
            const int comp = wl & 3;
            const int m = w == 0 ? (wl >> 2) & 7 : (w - 1) * 4 + ((wl >> 3) & 3);
            const int p = w == 0 ? 0 : 1 + ((wl >> 2) & 1);
            const float *src = (wl < 32 ? s_t[m].t.fa : s_t[m].t.fb) + (p == 0 ? 0 : (p == 1 ? 256 : 320)) + comp * (w == 0 ? 64 : 16);
            const float acc = w == 0 ? chain_n<64>(src) : chain_n<16>(src);
            const float s = __fdiv_rn(sum4_quad(acc), w == 0 ? 256.0f : 64.0f);
            const float C = __shfl_xor(s, 32, 64);
            const float D = __fadd_rn(s_x->D1[m][p], s);
            const float m1v = s_x->M1[m][p], m2v = s_x->M2[m][p];
            const float k1 = 0.01f * 0.01f * 255 * 255, k2 = 0.03f * 0.03f * 255 * 255;
            const float num = __fmul_rn(__fadd_rn(__fmul_rn(m1v, __fmul_rn(m2v, 2.0f)), k1), __fadd_rn(__fmul_rn(C, 2.0f), k2));
            const float den = __fmul_rn(__fadd_rn(__fmul_rn(m1v, m1v), __fadd_rn(__fmul_rn(m2v, m2v), k1)), __fadd_rn(D, k2));
            float metric = __fdiv_rn(num, den);
            float dm = __fsub_rn(m1v, m2v);
            dm = dm < 0 ? -dm : dm;
            dm = dm > 4 ? __fmul_rn(0.02f, dm) : 0.0f;
            metric = __fsub_rn(metric, dm);
            if (wl < 32 && comp == 0) s_x->metric[m][p] = metric;
        }
        if (act && luma) nz_c = s_x->nzc[g];
        __syncthreads();
        const float4 mt = *reinterpret_cast<const float4 *>(s_x->metric[g]);
        ssim = __fdiv_rn(__fadd_rn(__fadd_rn(mt.x, mt.y), mt.z), 3.0f);
    }

    // ---- results -------------------------------------------------------------------------------
    if (head && live) {
        if (any_pass) {
            const int nz = nz_y + nz_c;
            a.o_seg[mb] = seg_final;
            a.o_nz[mb] = nz;
            a.o_mask[mb] = (parts != 0 || nz > 0) ? -1 : 0;
        }
        a.o_ssim[mb] = ssim;
        if (ssim < a.ssim_target) *a.o_flag = 1;
    }
}

// One kernel for one context and for a batch (blockIdx.z = member; a single context is a batch of one).  The argument block
// is read through the kernel-argument pointer: a by-value MBArgs whose members are picked by a per-lane index (the reference
// of the macroblock) is copied to scratch memory by hipcc (296 B per lane, 21 -> 38 us per 1080p frame when it happened).
// Register budget: 127 VGPRs = four waves per SIMD, nothing spilled (history: 193 VGPRs and two waves; forcing three or
// four waves on that body by spilling was slower, 42.8-49.5 against 48.7-49.6 M MB/s).
__global__ __launch_bounds__(256, 2) void k_mb_b(BatchOf<MBArgs> b) {
    __shared__ __attribute__((aligned(16))) MBTile s_t[8];
    mb_body<false>(b.item[blockIdx.z], s_t);
}
// the same with the format's predictor instead of the reference's (vp8hip_conformant_stream): its own kernel, so that the
// reference path stays the code that was measured
__global__ __launch_bounds__(256, 2) void k_mb_b_conformant(BatchOf<MBArgs> b) {
    __shared__ __attribute__((aligned(16))) MBTile s_t[8];
    mb_body<true>(b.item[blockIdx.z], s_t);
}

// the packed form: 192 threads per eight macroblocks (VP8HIP_MB_PACKED=0: the form above, for same-box A/B runs)
__global__ __launch_bounds__(192, 2) void k_mb_p(BatchOf<MBArgs> b) {
    __shared__ __attribute__((aligned(16))) MBTileP s_t[8];
    __shared__ __attribute__((aligned(16))) MBShare s_x;
    mb_body_packed<false>(b.item[blockIdx.z], s_t, &s_x);
}
__global__ __launch_bounds__(192, 2) void k_mb_p_conformant(BatchOf<MBArgs> b) {
    __shared__ __attribute__((aligned(16))) MBTileP s_t[8];
    __shared__ __attribute__((aligned(16))) MBShare s_x;
    mb_body_packed<true>(b.item[blockIdx.z], s_t, &s_x);
}
static bool mb_packed() {
    static const bool on = [] { const char *v = getenv("VP8HIP_MB_PACKED"); return !(v && v[0] == '0'); }();
    return on;
}

static MBArgs mb_args(const Frame &cur, const RefSet &refs, const NetSet &nets, const Frame &recon, const MBOut &o, const SegData *d_sd,
                      float ssim_target, int mbw, int mbh) {
    MBArgs a;
    auto planes = [](const Frame &f) { return MBPlanes{f.Y[0].p, f.U.p, f.V.p}; };
    a.cur = planes(cur);
    a.ref0 = planes(refs.ref[0]); a.ref1 = planes(refs.ref[1]); a.ref2 = planes(refs.ref[2]);
    a.recon = planes(recon);
    a.ystride = cur.Y[0].stride; a.yw = cur.Y[0].w; a.yh = cur.Y[0].h;
    a.cstride = cur.U.stride; a.cw = cur.U.w; a.ch = cur.U.h;
    a.o_parts = o.parts; a.o_ref = o.ref; a.o_seg = o.seg; a.o_nz = o.nz; a.o_mask = o.mask;
    a.o_vec = o.vec; a.o_coeffs = o.coeffs; a.o_ssim = o.ssim; a.o_flag = o.flags;
    a.sd = d_sd;
    a.bdiff0 = nets.bdiff[0]; a.bdiff1 = nets.bdiff[1]; a.bdiff2 = nets.bdiff[2];
    a.vnet0 = nets.net[0][0]; a.vnet1 = nets.net[1][0]; a.vnet2 = nets.net[2][0];
    a.ssim_target = ssim_target;
    a.mbw = mbw;
    a.mbs = mbw * mbh;
    a.use_golden = refs.use[1];
    a.use_altref = refs.use[2];
    return a;
}
static bool mb_skip() {
    static const bool skip = experiment_skip("mb");
    return skip;   // timing experiment only (what the frame costs without this kernel); never set in production
}

void launch_mb(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, const Frame &recon,
               const MBOut &o, const SegData *d_sd, float ssim_target, int mbw, int mbh, bool conformant) {
    const MBArgs a = mb_args(cur, refs, nets, recon, o, d_sd, ssim_target, mbw, mbh);
    if (mb_skip()) return;
    BatchOf<MBArgs> b;
    b.n = 1;
    b.item[0] = a;
    if (mb_packed()) {
        if (conformant) VP8_LAUNCH(k_mb_p_conformant, dim3((a.mbs + 7) / 8, 1, 1), dim3(192), 0, s, b);
        else VP8_LAUNCH(k_mb_p, dim3((a.mbs + 7) / 8, 1, 1), dim3(192), 0, s, b);
        return;
    }
    if (conformant) VP8_LAUNCH(k_mb_b_conformant, dim3((a.mbs + 7) / 8, 1, 1), dim3(256), 0, s, b);
    else VP8_LAUNCH(k_mb_b, dim3((a.mbs + 7) / 8, 1, 1), dim3(256), 0, s, b);
}

void launch_mb_batch(hipStream_t s, const Frame *const *cur, const RefSet *refs, const NetSet *const *nets, const Frame *const *recon,
                     const MBOut *const *o, const SegData *const *d_sd, float ssim_target, int mbw, int mbh, int n, bool conformant) {
    static_assert(sizeof(BatchOf<MBArgs>) <= 4096, "the batch travels in the kernel arguments");
    BatchOf<MBArgs> b;
    b.n = n;
    for (int i = 0; i < n; ++i) b.item[i] = mb_args(*cur[i], refs[i], *nets[i], *recon[i], *o[i], d_sd[i], ssim_target, mbw, mbh);
    if (mb_skip()) return;
    if (mb_packed()) {
        if (conformant) VP8_LAUNCH(k_mb_p_conformant, dim3((b.item[0].mbs + 7) / 8, 1, n), dim3(192), 0, s, b);
        else VP8_LAUNCH(k_mb_p, dim3((b.item[0].mbs + 7) / 8, 1, n), dim3(192), 0, s, b);
        return;
    }
    if (conformant) VP8_LAUNCH(k_mb_b_conformant, dim3((b.item[0].mbs + 7) / 8, 1, n), dim3(256), 0, s, b);
    else VP8_LAUNCH(k_mb_b, dim3((b.item[0].mbs + 7) / 8, 1, n), dim3(256), 0, s, b);
}

// ------------------------------------------------------------------------------------------------
// prepare_filter_mask, CPU_kernels.cl:782-827, recomputed from the device copy of the coefficients
// (after the host changed them: vp8hip_upload_mb_data).  32 lanes per macroblock.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_filter_mask(MBOut o, const SegData *sd, int mbs) {
    const int lane = threadIdx.x & 31;
    const int mb_raw = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = mb_raw < mbs;
    const int mb = live ? mb_raw : mbs - 1;
    const int parts = o.parts[mb];
    int nz = 0;
    if (lane < 25) {
        const int16_t *c = o.coeffs + ((size_t)mb * 25 + lane) * 16;
        int s = 0;
#pragma unroll
        for (int k = 1; k < 16; ++k) s += iabs(c[k]);
        if (lane < 16) nz = s + (parts != 0 ? iabs(c[0]) : 0);
        else if (lane < 24) nz = s + iabs(c[0]);
        else nz = parts == 0 ? s + iabs(c[0]) : 0;
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) nz += __shfl_xor(nz, m, 32);
    if (lane == 0 && live) {
        o.nz[mb] = nz;
        o.mask[mb] = (parts != 0 || nz > 0) ? -1 : 0;
    }
}

void launch_filter_mask(hipStream_t s, const MBOut &o, const SegData *d_sd, int mbs) {
    hipLaunchKernelGGL(k_filter_mask, dim3((mbs + 7) / 8), dim3(256), 0, s, o, d_sd, mbs);
}

}  // namespace vp8
