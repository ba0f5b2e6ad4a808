// kernels_intra.hip -- the reference's HOST intra path on the device:
//   key frames            intra_transform / predict_and_transform_mb, src/intra_part.h:517-741,1089-1109
//   intra fallback        check_SSIM -> test_inter_on_intra, src/vp8enc.cpp:231-263, src/intra_part.h:855-1087
//
// The reference walks the macroblocks in raster order on one CPU thread: a macroblock predicts from the
// reconstruction of its left, top-left, top and top-right neighbours, and inside a macroblock each of the 16 luma
// 4x4 blocks predicts from the blocks before it.  Here one wavefront owns one macroblock ROW and the rows run as a
// lag-2 wavefront over the frame: row r may code macroblock c once row r-1 has finished c+1 (progress counters in
// HBM, data and counters at agent scope like the loop filter's band hand-off).  Inside the wavefront the 64 lanes are
// 10 candidate modes x 4 pixel rows: every lane predicts one row of one B_PRED mode, the 4x4 transforms run on
// quads (row pass in the lane, LDS transpose, column pass in the lane), all ten quads quantise and reconstruct, and
// only the quad of the winning mode (first strict minimum of the reference's `weight`) writes.  Chroma (TM_PRED)
// codes its 8 blocks at once on 8 quads.  Everything stays in LDS until the macroblock is committed.
//
// In fallback mode a macroblock is touched only if its SSIM is below the target, and a row never waits for rows
// above it unless it has such a macroblock: untouched macroblocks are final as the inter kernels left them.
#include <stdlib.h>

#include "vp8hip_dev.h"

namespace vp8 {
namespace {

constexpr int IMG_S = 32;                 // bytes per row of the luma work tile: x = -4 .. 27 at byte x + 4
constexpr int CIMG_S = 16;                // chroma work tiles: x = -4 .. 11
constexpr int INTRA_SPIN_LIMIT = 1 << 21; // polls (~1 us each) before a wait gives up

struct Sh {
    uint8_t img[17 * IMG_S];       // rows y = -1 .. 15: neighbours in row 0 / bytes 0..3, the reconstruction inside
    uint8_t cimg[2][9 * CIMG_S];
    uint8_t srcY[256];
    uint8_t srcC[2][64];
    int16_t tr[64 * 4];            // 4x4 transposes, one 8-byte row per lane
    int16_t coef[24 * 16];         // quantised coefficients of the macroblock, already in zigzag order
};

// Sub-block predictors over one edge array e[0..14] = { L3, L3, L2, L1, L0, TL, T0 .. T7, T7 } (the doubled ends turn
// the three "3x" taps of B_HE/B_LD/B_HU, src/intra_part.h:330,352,499, into the ordinary 1-2-1 filter).  An entry is
// kind << 4 | k:  kind 0 = (e[k-1] + 2 e[k] + e[k+1] + 2) >> 2,  kind 1 = (e[k] + e[k+1] + 1) >> 1,  kind 2 = e[k];
// on the device all three are the first form with the taps (e[k], e[k+1], e[k]) resp. (e[k], e[k], e[k]).
#define F3(k) (0x00 | (k))
#define F2(k) (0x10 | (k))
#define CP(k) (0x20 | (k))
__device__ __constant__ const uint8_t k_bpred[8][16] = {
    /* B_VE */ {F3(6), F3(7), F3(8), F3(9), F3(6), F3(7), F3(8), F3(9), F3(6), F3(7), F3(8), F3(9), F3(6), F3(7), F3(8), F3(9)},
    /* B_HE */ {F3(4), F3(4), F3(4), F3(4), F3(3), F3(3), F3(3), F3(3), F3(2), F3(2), F3(2), F3(2), F3(1), F3(1), F3(1), F3(1)},
    /* B_LD */ {F3(7), F3(8), F3(9), F3(10), F3(8), F3(9), F3(10), F3(11), F3(9), F3(10), F3(11), F3(12), F3(10), F3(11), F3(12), F3(13)},
    /* B_RD */ {F3(5), F3(6), F3(7), F3(8), F3(4), F3(5), F3(6), F3(7), F3(3), F3(4), F3(5), F3(6), F3(2), F3(3), F3(4), F3(5)},
    /* B_VR */ {F2(5), F2(6), F2(7), F2(8), F3(5), F3(6), F3(7), F3(8), F3(4), F2(5), F2(6), F2(7), F3(3), F3(5), F3(6), F3(7)},
    /* B_VL */ {F2(6), F2(7), F2(8), F2(9), F3(7), F3(8), F3(9), F3(10), F2(7), F2(8), F2(9), F3(11), F3(8), F3(9), F3(10), F3(12)},
    /* B_HD */ {F2(4), F3(5), F3(6), F3(7), F2(3), F3(4), F2(4), F3(5), F2(2), F3(3), F2(3), F3(4), F2(1), F3(2), F2(2), F3(3)},
    /* B_HU */ {F2(3), F3(3), F2(2), F3(2), F2(2), F3(2), F2(1), F3(1), F2(1), F3(1), CP(1), CP(1), CP(1), CP(1), CP(1), CP(1)},
};
#undef F3
#undef F2
#undef CP
// byte offset of e[k] in the luma tile relative to the block's top-left neighbour (row 4*br, byte 4*bc + 3)
__device__ __constant__ const uint8_t k_eoff[15] = {4 * IMG_S, 4 * IMG_S, 3 * IMG_S, 2 * IMG_S, IMG_S, 0, 1, 2, 3, 4, 5, 6, 7, 8, 8};
// position of raster coefficient j in the reference's zigzag_block order (src/intra_part.h:13-37)
__device__ __constant__ const uint8_t k_zzpos[16] = {0, 1, 5, 6, 2, 4, 7, 12, 3, 8, 11, 13, 9, 10, 14, 15};

// 2^32 / q + 1 for every quantizer step: trunc(t / q) = mulhi(t, M) - (t >> 31) for |t| < 2^32 / q
struct Recip { uint32_t m[285]; };
constexpr Recip make_recip() {
    Recip r{};
    for (int q = 0; q < 285; ++q) r.m[q] = q < 2 ? 0u : (uint32_t)((1ull << 32) / (unsigned)q) + 1u;
    return r;
}
__device__ __constant__ const Recip k_recip = make_recip();

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
// sums over the 16 lanes of a row land in its lane 15; over the wavefront in lane 63
__device__ __forceinline__ int row_sum(int v) {
    v += dpp<0xB1>(0, v);    // quad_perm [1,0,3,2]
    v += dpp<0x4E>(0, v);    // quad_perm [2,3,0,1]
    v += dpp<0x114>(0, v);   // row_shr:4
    v += dpp<0x118>(0, v);   // row_shr:8
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
    v = row_sum(v);
    v += dpp<0x142, 0xa>(0, v);   // row_bcast:15
    v += dpp<0x143, 0xc>(0, v);   // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }   // LDS operations of one wavefront execute in order

__device__ __forceinline__ uint32_t ld_agent(const uint8_t *p) {
    return __hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint8_t *p, uint32_t v) {
    __hip_atomic_store(reinterpret_cast<uint32_t *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Steps { int y_dc, y_ac, uv_dc, uv_ac; };   // prepare_segments_data, src/vp8enc.cpp:164-187 (deltas of segment 0)
__device__ __forceinline__ Steps steps_of(const SegData *sd, int id) {
    const int base = sd->v[id * SD_INTS + SD_Y_AC_I];
    Steps s;
    s.y_ac = k_ac_q[qi(base)];
    s.y_dc = k_dc_q[qi(base + sd->v[SD_Y_DC_IDELTA])];
    s.uv_dc = imin(k_dc_q[qi(base + sd->v[SD_UV_DC_IDELTA])], 132);
    s.uv_ac = k_ac_q[qi(base + sd->v[SD_UV_AC_IDELTA])];
    return s;
}

// quantizer of one lane: coefficient row 0 of its column is DC for column 0; everything else is AC
struct LaneQ { int q0, h0, qa, ha; uint32_t m0, ma; };
__device__ __forceinline__ LaneQ lane_q(int dc, int ac, int col) {
    LaneQ q;
    q.qa = ac; q.ha = ac / 2; q.ma = k_recip.m[ac];
    q.q0 = col == 0 ? dc : ac; q.h0 = q.q0 / 2; q.m0 = k_recip.m[q.q0];
    return q;
}
__device__ __forceinline__ int quant1(int w, int sign_of, int h, uint32_t m) {   // quant4x4, src/intra_part.h:212-250
    const int t = w + (sign_of < 0 ? -h : h);
    return __mulhi(t, (int)m) - (t >> 31);
}

// One 4x4 unit on a quad of lanes (lane i of the quad = pixel row i on entry, coefficient column i in the middle):
// residual -> DCT4x4 (:114-157) -> weight (:159-210) -> quant4x4 -> dequantise + iDCT4x4 (:42-111) -> reconstruction.
// p = this lane's predictor row, sdw = its source row.  Returns the packed reconstructed row; qc[j] = quantised
// coefficient 4*j + (lane & 3); weight = the reference's `weight` of the residual, identical in the four lanes.
__device__ __forceinline__ uint32_t code_unit(int16_t *tr, int lane, const int p[4], uint32_t sdw, const LaneQ &q, int qc[4], int &weight) {
    const int d0 = byte_of(sdw, 0) - p[0], d1 = byte_of(sdw, 1) - p[1], d2 = byte_of(sdw, 2) - p[2], d3 = byte_of(sdw, 3) - p[3];
    const int s03 = d0 + d3, s12 = d1 + d2, m12 = d1 - d2, m03 = d0 - d3;
    const int o0 = (s03 + s12) * 8, o2 = (s03 - s12) * 8;
    const int o1 = (__mul24(m12, 17736) + __mul24(m03, 42816) + 14500) >> 12;   // c1 = m12 << 3, d1 = m03 << 3
    const int o3 = (__mul24(m03, 17736) - __mul24(m12, 42816) + 7500) >> 12;
    int16_t *mine = tr + lane * 4;                          // tr: 64 x 4 shorts of this wavefront
    const int16_t *col = tr + (lane & ~3) * 4 + (lane & 3);
    *reinterpret_cast<uint2 *>(mine) = make_uint2(pk16(o0, o1), pk16(o2, o3));
    lds_order();
    const int v0 = col[0], v1 = col[4], v2 = col[8], v3 = col[12];
    lds_order();
    const int a = v0 + v3, b = v1 + v2, c = v1 - v2, d = v0 - v3;
    const int w0 = (a + b + 7) >> 4, w2 = (a - b + 7) >> 4;
    const int w1 = ((__mul24(c, 2217) + __mul24(d, 5352) + 12000) >> 16) + (d != 0);
    const int w3 = (__mul24(d, 2217) - __mul24(c, 5352) + 51000) >> 16;
    {
        const int t0 = (lane & 3) == 0 ? w0 / 4 : w0;
        int s = iabs(t0) + iabs(w1) + iabs(w2) + iabs(w3);
        s += dpp<0xB1>(0, s);
        s += dpp<0x4E>(0, s);
        weight = s;
    }
    // coefficient 11 (row 2, column 3) is rounded by the sign of coefficient 10 (row 2, column 2), :227
    const int w2sign = dpp<0xA4>(0, w2);   // quad_perm [0,1,2,2]
    qc[0] = quant1(w0, w0, q.h0, q.m0);
    qc[1] = quant1(w1, w1, q.ha, q.ma);
    qc[2] = quant1(w2, w2sign, q.ha, q.ma);
    qc[3] = quant1(w3, w3, q.ha, q.ma);
    const int x0 = __mul24(qc[0], q.q0), x1 = __mul24(qc[1], q.qa), x2 = __mul24(qc[2], q.qa), x3 = __mul24(qc[3], q.qa);
    {
        const int ia = x0 + x2, ib = x0 - x2;
        const int ic = ((__mul24(x1, 35468)) >> 16) - (x3 + ((__mul24(x3, 20091)) >> 16));
        const int id = (x1 + ((__mul24(x1, 20091)) >> 16)) + ((__mul24(x3, 35468)) >> 16);
        *reinterpret_cast<uint2 *>(mine) = make_uint2(pk16(ia + id, ib + ic), pk16(ib - ic, ia - id));   // 16-bit stores, as :54-75
    }
    lds_order();
    const int t0 = col[0], t1 = col[4], t2 = col[8], t3 = col[12];
    lds_order();
    const int ha = t0 + t2, hb = t0 - t2;
    const int hc = ((__mul24(t1, 35468)) >> 16) - (t3 + ((__mul24(t3, 20091)) >> 16));
    const int hd = (t1 + ((__mul24(t1, 20091)) >> 16)) + ((__mul24(t3, 35468)) >> 16);
    const int y0 = sat8(((ha + hd + 4) >> 3) + p[0]), y1 = sat8(((hb + hc + 4) >> 3) + p[1]);
    const int y2 = sat8(((hb - hc + 4) >> 3) + p[2]), y3 = sat8(((ha - hd + 4) >> 3) + p[3]);
    return (uint32_t)y0 | ((uint32_t)y1 << 8) | ((uint32_t)y2 << 16) | ((uint32_t)y3 << 24);
}

// per-lane constants of the luma mode decision: lane = 4 * mode + pixel row, modes 0..9 (lanes 40..63 idle)
struct LaneK {
    int a[4], b[4], c[4];   // tile offsets of the three taps of each of the 4 pixels
    int zz[4];              // zigzag position of coefficient 4*j + (lane & 3)
    int mode, row;
    bool is_dc, is_tm, active;
};
__device__ __forceinline__ LaneK lane_consts(int lane) {
    LaneK k;
    k.mode = lane >> 2;
    k.row = lane & 3;
    k.active = k.mode < 10;
    k.is_dc = k.mode == 0;
    k.is_tm = k.mode == 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ea = 5, eb = 5, ec = 5;
        if (k.mode == 0) ea = 1 + j;                        // B_DC: the four left neighbours; the top row comes as a dword
        else if (k.mode == 1) { ea = 4 - k.row; eb = 5; }   // B_TM: L[row] and TL
        else if (k.mode < 10) {
            const int e = k_bpred[k.mode - 2][4 * k.row + j], kind = e >> 4, kk = e & 15;
            ea = kind == 0 ? kk - 1 : kk;
            eb = kind == 1 ? kk + 1 : kk;
            ec = kind == 0 ? kk + 1 : kk;
        }
        k.a[j] = k_eoff[ea];
        k.b[j] = k_eoff[eb];
        k.c[j] = k_eoff[ec];
        k.zz[j] = k_zzpos[4 * j + (lane & 3)];
    }
    return k;
}

// one luma 4x4 block: pick_luma_predictor (:252-515) + transform + reconstruction; returns the chosen mode
template <int BR, int BC>
__device__ __forceinline__ int luma_block(Sh &sh, int lane, const LaneK &k, const LaneQ &q) {
    constexpr int B = 4 * BR * IMG_S + 4 * BC + 3;   // the block's top-left neighbour
    int A[4], Bt[4], C[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A[j] = sh.img[B + k.a[j]];
        Bt[j] = sh.img[B + k.b[j]];
        C[j] = sh.img[B + k.c[j]];
    }
    const uint32_t tdw = *reinterpret_cast<const uint32_t *>(&sh.img[B + 1]);                          // T0..T3
    const uint32_t sdw = *reinterpret_cast<const uint32_t *>(&sh.srcY[(4 * BR + k.row) * 16 + 4 * BC]);
    const int dc = (int)(__builtin_amdgcn_sad_u8(tdw, 0u, (uint32_t)(A[0] + A[1] + A[2] + A[3])) + 4u) >> 3;
    const int tm_d = A[0] - Bt[0];
    int p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = (A[j] + 2 * Bt[j] + C[j] + 2) >> 2;
        const int tm = sat8(byte_of(tdw, j) + tm_d);
        p[j] = k.is_dc ? dc : (k.is_tm ? tm : g);
    }
    int qc[4], w;
    const uint32_t rec = code_unit(sh.tr, lane, p, sdw, q, qc, w);
    // first strict minimum in mode order = minimum of (weight, mode)
    int key = k.active ? (w << 4) | k.mode : 0x7fffffff;
    key = imin(key, dpp<0x114>(0x7fffffff, key));
    key = imin(key, dpp<0x118>(0x7fffffff, key));
    key = imin(key, dpp<0x142, 0xa>(0x7fffffff, key));
    key = imin(key, dpp<0x143, 0xc>(0x7fffffff, key));
    const int best = __builtin_amdgcn_readlane(key, 63) & 15;
    if (k.mode == best) {
        *reinterpret_cast<uint32_t *>(&sh.img[(4 * BR + 1 + k.row) * IMG_S + 4 + 4 * BC]) = rec;
        int16_t *cf = sh.coef + (4 * BR + BC) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) cf[k.zz[j]] = (int16_t)qc[j];
    }
    lds_order();
    return best;
}

// the 8 chroma blocks (TM_PRED from the macroblock's edges, :674-739) on lanes 0..31: quad = plane * 4 + block
__device__ __forceinline__ void chroma_blocks(Sh &sh, int lane, const LaneK &k, int uv_dc, int uv_ac) {
    const int l = lane & 31, pl = l >> 4, bb = (l >> 2) & 3, br = bb >> 1, bc = bb & 1, i = l & 3;
    const uint8_t *t = sh.cimg[pl];
    const uint32_t tdw = *reinterpret_cast<const uint32_t *>(t + 4 + 4 * bc);
    const int dl = (int)t[(4 * br + i + 1) * CIMG_S + 3] - (int)t[3];
    const uint32_t sdw = *reinterpret_cast<const uint32_t *>(&sh.srcC[pl][(4 * br + i) * 8 + 4 * bc]);
    int p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = sat8(byte_of(tdw, j) + dl);
    const LaneQ q = lane_q(uv_dc, uv_ac, lane & 3);
    int qc[4], w;
    const uint32_t rec = code_unit(sh.tr, lane, p, sdw, q, qc, w);
    if (lane < 32) {
        *reinterpret_cast<uint32_t *>(&sh.cimg[pl][(4 * br + i + 1) * CIMG_S + 4 + 4 * bc]) = rec;
        int16_t *cf = sh.coef + (16 + 4 * pl + bb) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) cf[k.zz[j]] = (int16_t)qc[j];
    }
    lds_order();
}

// count_SSIM_16x16 (src/intra_part.h:744-853) of the work tiles against the source tiles.  The reference's integer
// accumulators are NOT reset between planes: M1 enters chroma as |M1 - M2|, M2/D1/D2/C with their luma results.
// Central moments come from raw moments (exact in integers): sum (x - m)^2 = sum x^2 - 2 m sum x + n m^2.
__device__ __forceinline__ int tdiv_pow2(int v, int sh) { return (v + ((v >> 31) & ((1 << sh) - 1))) >> sh; }   // C division
__device__ float mb_ssim(Sh &sh, int lane) {
    const float c1 = 0.01f * 0.01f * 255 * 255, c2 = 0.03f * 0.03f * 255 * 255;
    int M1 = 0, M2 = 0, D1 = 0, D2 = 0, C = 0;
    float ssim = 0.0f;
    // raw moments of the three planes: luma on all lanes, U on lanes 0..15, V on lanes 16..31
    const uint32_t ty = *reinterpret_cast<const uint32_t *>(&sh.img[((lane >> 2) + 1) * IMG_S + 4 + 4 * (lane & 3)]);
    const uint32_t sy = *reinterpret_cast<const uint32_t *>(&sh.srcY[lane * 4]);
    const int l = lane & 31, pl = l >> 4;
    const uint32_t tc = *reinterpret_cast<const uint32_t *>(&sh.cimg[pl][(((l >> 1) & 7) + 1) * CIMG_S + 4 + 4 * (l & 1)]);
    const uint32_t sc = *reinterpret_cast<const uint32_t *>(&sh.srcC[pl][(l & 15) * 4]);
    int S1[3], S2[3], Q1[3], Q2[3], X[3];
    S1[0] = wave_sum((int)__builtin_amdgcn_sad_u8(ty, 0u, 0u));
    S2[0] = wave_sum((int)__builtin_amdgcn_sad_u8(sy, 0u, 0u));
    Q1[0] = wave_sum((int)__builtin_amdgcn_udot4(ty, ty, 0u, false));
    Q2[0] = wave_sum((int)__builtin_amdgcn_udot4(sy, sy, 0u, false));
    X[0] = wave_sum((int)__builtin_amdgcn_udot4(ty, sy, 0u, false));
    {
        const int a = row_sum((int)__builtin_amdgcn_sad_u8(tc, 0u, 0u)), b = row_sum((int)__builtin_amdgcn_sad_u8(sc, 0u, 0u));
        const int cq = row_sum((int)__builtin_amdgcn_udot4(tc, tc, 0u, false)), dq = row_sum((int)__builtin_amdgcn_udot4(sc, sc, 0u, false));
        const int e = row_sum((int)__builtin_amdgcn_udot4(tc, sc, 0u, false));
        S1[1] = __builtin_amdgcn_readlane(a, 15); S1[2] = __builtin_amdgcn_readlane(a, 31);
        S2[1] = __builtin_amdgcn_readlane(b, 15); S2[2] = __builtin_amdgcn_readlane(b, 31);
        Q1[1] = __builtin_amdgcn_readlane(cq, 15); Q1[2] = __builtin_amdgcn_readlane(cq, 31);
        Q2[1] = __builtin_amdgcn_readlane(dq, 15); Q2[2] = __builtin_amdgcn_readlane(dq, 31);
        X[1] = __builtin_amdgcn_readlane(e, 15); X[2] = __builtin_amdgcn_readlane(e, 31);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int n = p ? 64 : 256, lg = p ? 6 : 8;
        M1 = (M1 + S1[p] + n / 2) >> lg;     // sums are non-negative
        M2 = (M2 + S2[p] + n / 2) >> lg;
        D1 = tdiv_pow2(D1 + Q1[p] - 2 * M1 * S1[p] + n * M1 * M1 + n / 2, lg);
        D2 = tdiv_pow2(D2 + Q2[p] - 2 * M2 * S2[p] + n * M2 * M2 + n / 2, lg);
        C = tdiv_pow2(C + X[p] - M1 * S2[p] - M2 * S1[p] + n * M1 * M2 + n / 2, lg);
        const float m1 = (float)M1, m2 = (float)M2;
        const float num = (m1 * m2 * 2 + c1) * ((float)C * 2 + c2);
        const float den = (m1 * m1 + m2 * m2 + c1) * ((float)D1 + (float)D2 + c2);
        ssim = p == 0 ? __fdiv_rn(num, den) : ssim + __fdiv_rn(num, den);
        M1 = iabs(M1 - M2);
        ssim -= M1 > 4 ? (float)M1 * 0.02f : 0.0f;
    }
    return __fdiv_rn(ssim, 3.0f);
}

struct IntraArgs {
    Plane cy, cu, cv;       // current frame
    Plane ry, ru, rv;       // reconstruction (unfiltered), read for the neighbours and written
    MBOut o;
    const SegData *sd;
    int32_t *modes;         // [MBs][16] sub-block modes of the last attempt (e_data.mode)
    int32_t *is_inter;      // [MBs] 0 where check_SSIM replaced the macroblock
    int32_t *prog;          // [mbh] macroblocks of the row that are final, + gen_base
    uint32_t gen_base;      // launch number * 1024: a counter left by an earlier launch reads as negative progress
    const int32_t *flagged; // fallback riding in the frame's chain (vp8hip_check_ssim_async): *flagged == 0 = k_mb left no macroblock
                            // below the target, nothing to do and nothing to initialise; nullptr = always run
    int32_t *err;           // set to 1 when a bounded wait expired (shared with the loop filter: VP8HIP_ERR_TIMEOUT)
    float target;
    int key;                // 1: key frame (every macroblock, segment 0); 0: fallback of an inter frame
    int mbw, mbh;
    int stall_test;         // test hook: row 0 never publishes, so every other row must run into its bounded wait
    int modes_of_kept;      // 0: modes of the last attempt MADE (the reference); 1: of the attempt KEPT (decodable)
};

// Row progress carries the launch's generation, so the counters need no clearing between launches (the clearing was a
// command of its own in every frame's chain): whatever an earlier launch left is below this launch's base.
__device__ __forceinline__ int prog_load(const IntraArgs &a, int r) {
    return (int)((uint32_t)__hip_atomic_load(&a.prog[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.gen_base);
}
__device__ __forceinline__ void prog_store(const IntraArgs &a, int r, int c) {
    __hip_atomic_store(&a.prog[r], (int32_t)(a.gen_base + (uint32_t)c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// prepare_filter_mask (CPU_kernels.cl:782-827) of a macroblock the fallback has just replaced: its parts are 4x4, so the
// mask is set and the count is the sum of |coefficient| over blocks 0..23 (sh.coef); all 64 lanes, result in every lane
__device__ __forceinline__ int replaced_nz(const Sh &sh, int lane) {
    const uint32_t *lc = reinterpret_cast<const uint32_t *>(sh.coef);
    int s = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t w = lc[lane + 64 * i];
        s += iabs((int)(int16_t)(w & 0xffffu)) + iabs((int)(int16_t)(w >> 16));
    }
    return wave_sum(s);
}

__global__ __launch_bounds__(64) void k_intra(IntraArgs a) {
    __shared__ __attribute__((aligned(16))) Sh sh;
    const int lane = threadIdx.x, r = blockIdx.x, mbw = a.mbw;
    const LaneK k = lane_consts(lane);
    int32_t *err = a.err;
    if (a.stall_test && r == 0) return;
    const int mb_row0 = r * mbw;
    if (!a.key) {   // frame loop + check_SSIM defaults: every macroblock inter, no modes (vp8enc.cpp:437-438)
        for (int i = lane; i < mbw; i += 64) a.is_inter[mb_row0 + i] = 1;
        for (int i = lane; i < mbw * 16; i += 64) a.modes[(size_t)mb_row0 * 16 + i] = 0;
    }
    int c = 0;
    while (c < mbw) {
        if (!a.key) {   // next macroblock below the target
            const int idx = c + lane;
            const bool f = idx < mbw && a.o.ssim[mb_row0 + idx] < a.target;
            const unsigned long long m = __ballot(f);
            if (!m) { c += 64; continue; }
            c += __builtin_ctzll(m);
        }
        if (lane == 0) prog_store(a, r, c);   // everything before c is final
        const int mb = mb_row0 + c;
        // ---- source tiles --------------------------------------------------------------------------------------
        *reinterpret_cast<uint32_t *>(&sh.srcY[lane * 4]) =
            *reinterpret_cast<const uint32_t *>(a.cy.p + (ptrdiff_t)(16 * r + (lane >> 2)) * a.cy.stride + 16 * c + 4 * (lane & 3));
        if (lane < 32) {
            const Plane &P = lane < 16 ? a.cu : a.cv;
            const int l = lane & 15;
            *reinterpret_cast<uint32_t *>(&sh.srcC[lane >> 4][l * 4]) =
                *reinterpret_cast<const uint32_t *>(P.p + (ptrdiff_t)(8 * r + (l >> 1)) * P.stride + 8 * c + 4 * (l & 1));
        }
        // ---- neighbours: the row above must have finished its macroblock c + 1 -----------------------------------
        if (r > 0) {
            const int need = imin(c + 2, mbw);
            int spins = 0;
            while (prog_load(a, r - 1) < need) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > INTRA_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;
                }
            }
        }
        if (lane < 44) {
            // lanes 0..5 / 6..8 / 9..11: the row above (x = -4 .. 19 / -4 .. 7); lanes 12..27 / 28..35 / 36..43: the column left
            const bool top = lane < 12;
            const int pl = top ? (lane < 6 ? 0 : (lane < 9 ? 1 : 2)) : (lane < 28 ? 0 : (lane < 36 ? 1 : 2));
            const int j = top ? (pl == 0 ? lane : (pl == 1 ? lane - 6 : lane - 9)) : (pl == 0 ? lane - 12 : (pl == 1 ? lane - 28 : lane - 36));
            const Plane &P = pl == 0 ? a.ry : (pl == 1 ? a.ru : a.rv);
            const int msz = pl == 0 ? 16 : 8;
            const bool last_col_ar = top && pl == 0 && j == 5 && c == mbw - 1;      // no macroblock above-right: repeat top[15] (:596-601)
            const int x = top ? msz * c - 4 + 4 * (last_col_ar ? 4 : j) : msz * c - 4;
            const int y = top ? msz * r - 1 : msz * r + j;
            uint32_t v = ld_agent(P.p + (ptrdiff_t)y * P.stride + x);
            if (last_col_ar) v = (v >> 24) * 0x01010101u;
            if (top && r == 0) v = 0x7f7f7f7fu;                                     // 127 above the frame, also in the corner (:575-589)
            else if (c == 0 && (!top || j == 0)) v = 0x81818181u;                   // 129 left of the frame (:540-551)
            uint8_t *dst = pl == 0 ? sh.img : sh.cimg[pl - 1];
            const int S = pl == 0 ? IMG_S : CIMG_S;
            *reinterpret_cast<uint32_t *>(dst + (top ? 4 * j : (j + 1) * S)) = v;
            if (top && pl == 0 && j == 5) {   // blocks 7, 11, 15 take T4..T7 from the row above the macroblock too (:630, top_pred_Y[16..19])
                *reinterpret_cast<uint32_t *>(sh.img + 4 * IMG_S + 20) = v;
                *reinterpret_cast<uint32_t *>(sh.img + 8 * IMG_S + 20) = v;
                *reinterpret_cast<uint32_t *>(sh.img + 12 * IMG_S + 20) = v;
            }
        }
        lds_order();
        float cur_ssim = a.key ? 0.0f : a.o.ssim[mb];
        // ---- attempts: key frame = segment 0 once; fallback = AQ (2), HQ (1), UQ (0) while below the target --------
        for (int att = 0; att < (a.key ? 1 : 3); ++att) {
            const int seg = a.key ? 0 : 2 - att;
            if (!a.key && !(cur_ssim < a.target)) break;
            const Steps st = steps_of(a.sd, seg);
            const LaneQ ql = lane_q(st.y_dc, st.y_ac, lane & 3);
            int mymode = 0;
#define LUMA(BR, BC)                                                   \
            {                                                          \
                const int m_ = luma_block<BR, BC>(sh, lane, k, ql);    \
                if (lane == 4 * BR + BC) mymode = m_;                  \
            }
            LUMA(0, 0) LUMA(0, 1) LUMA(0, 2) LUMA(0, 3)
            LUMA(1, 0) LUMA(1, 1) LUMA(1, 2) LUMA(1, 3)
            LUMA(2, 0) LUMA(2, 1) LUMA(2, 2) LUMA(2, 3)
            LUMA(3, 0) LUMA(3, 1) LUMA(3, 2) LUMA(3, 3)
#undef LUMA
            chroma_blocks(sh, lane, k, st.uv_dc, st.uv_ac);
            if (lane < 16 && !a.modes_of_kept) a.modes[(size_t)mb * 16 + lane] = mymode;   // e_data.mode is overwritten by every attempt (:970)
            bool commit = true;
            float s = 0.0f;
            if (!a.key) {
                s = mb_ssim(sh, lane);
                commit = s > cur_ssim;
            }
            if (commit) {
                if (lane < 16 && a.modes_of_kept) a.modes[(size_t)mb * 16 + lane] = mymode;
                cur_ssim = s;
                const int nz = a.key ? 0 : replaced_nz(sh, lane);
                st_agent(a.ry.p + (ptrdiff_t)(16 * r + (lane >> 2)) * a.ry.stride + 16 * c + 4 * (lane & 3),
                         *reinterpret_cast<const uint32_t *>(&sh.img[((lane >> 2) + 1) * IMG_S + 4 + 4 * (lane & 3)]));
                if (lane < 32) {
                    const Plane &P = lane < 16 ? a.ru : a.rv;
                    const int l = lane & 15;
                    st_agent(P.p + (ptrdiff_t)(8 * r + (l >> 1)) * P.stride + 8 * c + 4 * (l & 1),
                             *reinterpret_cast<const uint32_t *>(&sh.cimg[lane >> 4][((l >> 1) + 1) * CIMG_S + 4 + 4 * (l & 1)]));
                }
                uint32_t *gc = reinterpret_cast<uint32_t *>(a.o.coeffs + (size_t)mb * 400);
                const uint32_t *lc = reinterpret_cast<const uint32_t *>(sh.coef);
#pragma unroll
                for (int i = 0; i < 3; ++i) gc[lane + 64 * i] = lc[lane + 64 * i];   // blocks 0..23; block 24 (Y2) stays
                if (lane == 0) {
                    a.o.parts[mb] = 2;    // are4x4
                    a.o.seg[mb] = seg;
                    if (!a.key) {
                        a.o.ssim[mb] = s;
                        a.is_inter[mb] = 0;
                        a.o.nz[mb] = nz;
                        a.o.mask[mb] = -1;
                    }
                }
            }
        }
        ++c;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) prog_store(a, r, c);
    }
    if (lane == 0) prog_store(a, r, mbw);
}

// ---------------------------------------------------------------------------------------------------------------------
// check_SSIM's fallback with the three attempts of a macroblock side by side.  The reference tries segment AQ, then
// HQ, then UQ, each only while the macroblock is still below the target -- but an attempt never reads what an earlier
// one committed (a macroblock is not its own neighbour), so the three are independent computations and only the
// DECISIONS are sequential.  Three wavefronts code the three attempts at once into their own tiles; then every
// thread replays the reference's chain on the three SSIM values: which attempts "ran" (the last of them owns
// e_data.mode), which one is kept (the last that raised the SSIM).  A flagged macroblock costs one attempt's time
// instead of up to three.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void intra_check3_body(const IntraArgs &a) {
    __shared__ __attribute__((aligned(16))) Sh sh3[3];
    __shared__ float s_ssim[3];
    __shared__ int s_abort;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = blockIdx.x, mbw = a.mbw;
    Sh &sh = sh3[wave];
    const LaneK k = lane_consts(lane);
    const int mb_row0 = r * mbw;
    if (a.stall_test && r == 0) return;
    if (a.flagged && __builtin_nontemporal_load(a.flagged) == 0) {   // (is_inter / modes are then not read either: replaced == 0)
        // the row's counter still gets this launch's number: counters carry the launch number modulo 2^22, and a row left
        // unstamped for that many launches (long GOPs with nothing below the target) would read as complete in a later one
        if (threadIdx.x == 0) prog_store(a, r, mbw);
        return;
    }
    for (int i = threadIdx.x; i < mbw; i += 192) a.is_inter[mb_row0 + i] = 1;
    for (int i = threadIdx.x; i < mbw * 16; i += 192) a.modes[(size_t)mb_row0 * 16 + i] = 0;
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    const int seg = 2 - wave;                         // AQ, HQ, UQ
    const Steps st = steps_of(a.sd, seg);
    const LaneQ ql = lane_q(st.y_dc, st.y_ac, lane & 3);
    int c = 0;
    while (c < mbw) {
        {   // next macroblock below the target (every wave reads the same values and arrives at the same c)
            const int idx = c + lane;
            const bool f = idx < mbw && a.o.ssim[mb_row0 + idx] < a.target;
            const unsigned long long m = __ballot(f);
            if (!m) { c += 64; continue; }
            c += __builtin_ctzll(m);
        }
        if (threadIdx.x == 0) prog_store(a, r, c);
        const int mb = mb_row0 + c;
        const float cur0 = a.o.ssim[mb];   // read before anybody may replace it
        *reinterpret_cast<uint32_t *>(&sh.srcY[lane * 4]) =
            *reinterpret_cast<const uint32_t *>(a.cy.p + (ptrdiff_t)(16 * r + (lane >> 2)) * a.cy.stride + 16 * c + 4 * (lane & 3));
        if (lane < 32) {
            const Plane &P = lane < 16 ? a.cu : a.cv;
            const int l = lane & 15;
            *reinterpret_cast<uint32_t *>(&sh.srcC[lane >> 4][l * 4]) =
                *reinterpret_cast<const uint32_t *>(P.p + (ptrdiff_t)(8 * r + (l >> 1)) * P.stride + 8 * c + 4 * (l & 1));
        }
        if (r > 0 && wave == 0) {
            const int need = imin(c + 2, mbw);
            int spins = 0;
            while (prog_load(a, r - 1) < need) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > INTRA_SPIN_LIMIT || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (lane == 0) {
                        __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_abort = 1;
                    }
                    break;
                }
            }
        }
        __syncthreads();
        if (s_abort) return;
        if (lane < 44) {   // neighbours, as in k_intra (each wave fills its own tile)
            const bool top = lane < 12;
            const int pl = top ? (lane < 6 ? 0 : (lane < 9 ? 1 : 2)) : (lane < 28 ? 0 : (lane < 36 ? 1 : 2));
            const int j = top ? (pl == 0 ? lane : (pl == 1 ? lane - 6 : lane - 9)) : (pl == 0 ? lane - 12 : (pl == 1 ? lane - 28 : lane - 36));
            const Plane &P = pl == 0 ? a.ry : (pl == 1 ? a.ru : a.rv);
            const int msz = pl == 0 ? 16 : 8;
            const bool last_col_ar = top && pl == 0 && j == 5 && c == mbw - 1;
            const int x = top ? msz * c - 4 + 4 * (last_col_ar ? 4 : j) : msz * c - 4;
            const int y = top ? msz * r - 1 : msz * r + j;
            uint32_t v = ld_agent(P.p + (ptrdiff_t)y * P.stride + x);
            if (last_col_ar) v = (v >> 24) * 0x01010101u;
            if (top && r == 0) v = 0x7f7f7f7fu;
            else if (c == 0 && (!top || j == 0)) v = 0x81818181u;
            uint8_t *dst = pl == 0 ? sh.img : sh.cimg[pl - 1];
            const int S = pl == 0 ? IMG_S : CIMG_S;
            *reinterpret_cast<uint32_t *>(dst + (top ? 4 * j : (j + 1) * S)) = v;
            if (top && pl == 0 && j == 5) {
                *reinterpret_cast<uint32_t *>(sh.img + 4 * IMG_S + 20) = v;
                *reinterpret_cast<uint32_t *>(sh.img + 8 * IMG_S + 20) = v;
                *reinterpret_cast<uint32_t *>(sh.img + 12 * IMG_S + 20) = v;
            }
        }
        lds_order();
        int mymode = 0;
#define LUMA(BR, BC)                                                   \
        {                                                              \
            const int m_ = luma_block<BR, BC>(sh, lane, k, ql);        \
            if (lane == 4 * BR + BC) mymode = m_;                      \
        }
        LUMA(0, 0) LUMA(0, 1) LUMA(0, 2) LUMA(0, 3)
        LUMA(1, 0) LUMA(1, 1) LUMA(1, 2) LUMA(1, 3)
        LUMA(2, 0) LUMA(2, 1) LUMA(2, 2) LUMA(2, 3)
        LUMA(3, 0) LUMA(3, 1) LUMA(3, 2) LUMA(3, 3)
#undef LUMA
        chroma_blocks(sh, lane, k, st.uv_dc, st.uv_ac);
        const float s = mb_ssim(sh, lane);
        if (lane == 0) s_ssim[wave] = s;
        __syncthreads();
        // the reference's decision chain (vp8enc.cpp:245-250, intra_part.h:1058-1086) on the three results
        float cur = cur0;
        int last_run = -1, kept = -1;
        for (int t = 0; t < 3; ++t) {
            if (!(cur < a.target)) break;
            last_run = t;
            if (s_ssim[t] > cur) { kept = t; cur = s_ssim[t]; }
        }
        if (wave == (a.modes_of_kept ? kept : last_run) && lane < 16) a.modes[(size_t)mb * 16 + lane] = mymode;   // e_data.mode: of the last attempt made (:970)
        if (wave == kept) {
            const int nz = replaced_nz(sh, lane);
            st_agent(a.ry.p + (ptrdiff_t)(16 * r + (lane >> 2)) * a.ry.stride + 16 * c + 4 * (lane & 3),
                     *reinterpret_cast<const uint32_t *>(&sh.img[((lane >> 2) + 1) * IMG_S + 4 + 4 * (lane & 3)]));
            if (lane < 32) {
                const Plane &P = lane < 16 ? a.ru : a.rv;
                const int l = lane & 15;
                st_agent(P.p + (ptrdiff_t)(8 * r + (l >> 1)) * P.stride + 8 * c + 4 * (l & 1),
                         *reinterpret_cast<const uint32_t *>(&sh.cimg[lane >> 4][((l >> 1) + 1) * CIMG_S + 4 + 4 * (l & 1)]));
            }
            uint32_t *gc = reinterpret_cast<uint32_t *>(a.o.coeffs + (size_t)mb * 400);
            const uint32_t *lc = reinterpret_cast<const uint32_t *>(sh.coef);
#pragma unroll
            for (int i = 0; i < 3; ++i) gc[lane + 64 * i] = lc[lane + 64 * i];
            if (lane == 0) {
                a.o.parts[mb] = 2;
                a.o.seg[mb] = seg;
                a.o.ssim[mb] = cur;
                a.is_inter[mb] = 0;
                a.o.nz[mb] = nz;        // prepare_filter_mask of the replaced macroblock: no launch of its own afterwards
                a.o.mask[mb] = -1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ++c;
        __syncthreads();   // the kept attempt's stores are complete: the row may advance
        if (threadIdx.x == 0) prog_store(a, r, c);
    }
    if (threadIdx.x == 0) prog_store(a, r, mbw);
}
__global__ __launch_bounds__(192) void k_intra_check3(IntraArgs a) { intra_check3_body(a); }
// the same for the members of a batch: blockIdx.z = member
__global__ __launch_bounds__(192) void k_intra_check3_b(BatchOf<IntraArgs> b) { intra_check3_body(b.item[blockIdx.z]); }

// ---------------------------------------------------------------------------------------------------------------------
// Key frames, pipelined at 4x4-block granularity.  In k_intra one wavefront walks the 16 luma blocks of a macroblock
// one after the other, so the frame's critical path is 16 * (mbw + 2 mbh) block steps.  A luma block only needs the
// block to its left, the row of pixels above it and four pixels above-right -- so here a workgroup still owns one
// macroblock row but splits it over five wavefronts: wave k (0..3) codes the k-th row of 4x4 blocks of EVERY
// macroblock of the row, running two blocks behind wave k-1 (one where the above-right pixels come from the
// macroblock row above, :630), and wave 4 codes the chroma blocks and publishes the row's progress.  The rows of
// pixels the waves hand to each other live in LDS for the whole width (bot[k] = the pixel row above block row k), the
// hand-off is a counter in LDS; only the macroblock row's last pixel row crosses to the next workgroup through HBM.
// Critical path: 4 mbw + ~14 mbh block steps (1080p: 2.8 ms -> 1 ms).  A key frame commits every macroblock, which
// is what makes this legal; check_SSIM's fallback decides per macroblock and stays on k_intra.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int K4_WAVES = 5;
#define K4_WAIT(cond_unsatisfied, nap)                                                                  \
    {                                                                                                   \
        int spins_ = 0;                                                                                 \
        while ((cond_unsatisfied) && !*abort_flag) {                                                    \
            __builtin_amdgcn_s_sleep(nap);                                                              \
            if (++spins_ > 8 * INTRA_SPIN_LIMIT) {   /* LDS polls are ~0.1 us: about a second */          \
                *abort_flag = 1;                                                                        \
                __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);               \
            }                                                                                           \
        }                                                                                               \
        if (*abort_flag) return;                                                                        \
    }

__device__ __forceinline__ void intra_key4_body(const IntraArgs &a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = blockIdx.x, mbw = a.mbw;
    const int BW = mbw * 16 + 16;                      // bytes per pixel row: x = -4 .. W + 11 at byte x + 4
    uint8_t *bot = smem;                               // [5][BW]  bot[k] = pixel row above block row k (bot[4]: the row's last)
    uint8_t *lcol = bot + 5 * BW;                      // [4][4]   right column of each luma wave's previous block
    uint8_t *E = lcol + 16;                            // [4][16]  edge array of the block in work
    uint8_t *src = E + 64;                             // [4][2][64] source rows of the macroblock in work / the next one
    int16_t *tr = reinterpret_cast<int16_t *>(src + 512);           // [5][256] transposes
    uint8_t *cimg = reinterpret_cast<uint8_t *>(tr + K4_WAVES * 256);   // [2][9 * 16] chroma tiles as in k_intra
    uint8_t *csrc = cimg + 2 * 9 * CIMG_S;             // [2][64]
    int16_t *ccoef = reinterpret_cast<int16_t *>(csrc + 128);       // [8][16]
    volatile int *done = reinterpret_cast<volatile int *>(ccoef + 128);   // [4] blocks finished by each luma wave
    volatile int *abort_flag = done + 4;
    if (a.stall_test && r == 0) return;
    for (int i = threadIdx.x; i < 5 * BW; i += 64 * K4_WAVES) {
        const int k = i / BW, x = i % BW;
        bot[i] = k == 0 ? 127 : (x < 4 ? 129 : 0);     // above the frame 127; left of it 129 (the corners of block rows 1..3)
    }
    if (threadIdx.x < 16) lcol[threadIdx.x] = 129;
    if (threadIdx.x < 5) done[threadIdx.x] = 0;        // done[4] is the abort flag
    __syncthreads();
    const Steps st = steps_of(a.sd, 0);

    if (wave < 4) {
        // ---- luma block row `wave` ---------------------------------------------------------------------------------
        LaneK k = lane_consts(lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // taps as indices into the edge array instead of tile offsets
            int ea = 5, eb = 5, ec = 5;
            if (k.mode == 0) ea = 1 + j;
            else if (k.mode == 1) { ea = 4 - k.row; eb = 5; }
            else if (k.mode < 10) {
                const int e = k_bpred[k.mode - 2][4 * k.row + j], kind = e >> 4, kk = e & 15;
                ea = kind == 0 ? kk - 1 : kk;
                eb = kind == 1 ? kk + 1 : kk;
                ec = kind == 0 ? kk + 1 : kk;
            }
            k.a[j] = ea; k.b[j] = eb; k.c[j] = ec;
        }
        const LaneQ ql = lane_q(st.y_dc, st.y_ac, lane & 3);
        uint8_t *myE = E + 16 * wave, *mytop = bot + wave * BW, *mybot = bot + (wave + 1) * BW, *mycol = lcol + 4 * wave;
        uint8_t *mysrc = src + 128 * wave;
        int16_t *mytr = tr + 256 * wave;
        const int y0 = 16 * r + 4 * wave;
        // lane l < 16 fetches source row l >> 2, dword l & 3 of a macroblock
        auto fetch = [&](int c) -> uint32_t {
            return *reinterpret_cast<const uint32_t *>(a.cy.p + (ptrdiff_t)(y0 + ((lane & 15) >> 2)) * a.cy.stride + 16 * c + 4 * (lane & 3));
        };
        uint32_t nxt = fetch(0);
        if (lane < 16) *reinterpret_cast<uint32_t *>(mysrc + 4 * lane) = nxt;
        for (int c = 0; c < mbw; ++c) {
            if (c + 1 < mbw) nxt = fetch(c + 1);        // in flight while this macroblock is coded
            const uint8_t *s_mb = mysrc + 64 * (c & 1);
            if (wave == 0 && r > 0) {
                // the pixel row above this macroblock row, x = 16c - 4 .. 16c + 19, once the row above has finished c + 1
                const int need = imin(c + 2, mbw);
                K4_WAIT(prog_load(a, r - 1) < need, 2)
                if (lane < 6) {
                    const bool ar_last = lane == 5 && c == mbw - 1;   // no macroblock above-right: repeat top[15] (:596-601)
                    uint32_t v = ld_agent(a.ry.p + (ptrdiff_t)(16 * r - 1) * a.ry.stride + 16 * c - 4 + 4 * (ar_last ? 4 : lane));
                    if (ar_last) v = (v >> 24) * 0x01010101u;
                    if (c == 0 && lane == 0) v = 0x81818181u;         // top-left of the first macroblock: 129 (:540-551)
                    *reinterpret_cast<uint32_t *>(bot + 16 * c + 4 * lane) = v;
                }
                lds_order();
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * c + q;
                if (wave > 0) {
                    const int need = imin(q == 3 ? j + 1 : j + 2, 4 * mbw);
                    K4_WAIT(done[wave - 1] < need, 1)
                }
                // edge array { L3, L3, L2, L1, L0, TL, T0 .. T7, T7 }: T4..T7 of a macroblock's last column come from the
                // row above the macroblock (bot[0]), like top_pred_Y[16..19] in the reference
                if (lane < 15) {
                    int v;
                    if (lane < 5) v = mycol[lane < 2 ? 3 : 4 - lane];
                    else {
                        const int t = lane == 14 ? 8 : lane - 5;          // 0 = TL, 1..8 = T0..T7
                        const uint8_t *row = (q == 3 && t >= 5) ? bot : mytop;
                        v = row[4 * j + 3 + t];
                    }
                    myE[lane] = (uint8_t)v;
                }
                lds_order();
                int A[4], Bt[4], C[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    A[t] = myE[k.a[t]];
                    Bt[t] = myE[k.b[t]];
                    C[t] = myE[k.c[t]];
                }
                const uint32_t tdw = *reinterpret_cast<const uint32_t *>(mytop + 4 * j + 4);
                const uint32_t sdw = *reinterpret_cast<const uint32_t *>(s_mb + 16 * k.row + 4 * q);
                const int dc = (int)(__builtin_amdgcn_sad_u8(tdw, 0u, (uint32_t)(A[0] + A[1] + A[2] + A[3])) + 4u) >> 3;
                const int tm_d = A[0] - Bt[0];
                int p[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int g = (A[t] + 2 * Bt[t] + C[t] + 2) >> 2;
                    const int tm = sat8(byte_of(tdw, t) + tm_d);
                    p[t] = k.is_dc ? dc : (k.is_tm ? tm : g);
                }
                int qc[4], w;
                const uint32_t rec = code_unit(mytr, lane, p, sdw, ql, qc, w);
                int key = k.active ? (w << 4) | k.mode : 0x7fffffff;
                key = imin(key, dpp<0x114>(0x7fffffff, key));
                key = imin(key, dpp<0x118>(0x7fffffff, key));
                key = imin(key, dpp<0x142, 0xa>(0x7fffffff, key));
                key = imin(key, dpp<0x143, 0xc>(0x7fffffff, key));
                const int best = __builtin_amdgcn_readlane(key, 63) & 15;
                const int mb = r * mbw + c, blk = 4 * wave + q;
                if (k.mode == best) {
                    *reinterpret_cast<uint32_t *>(a.ry.p + (ptrdiff_t)(y0 + k.row) * a.ry.stride + 4 * j) = rec;
                    if (k.row == 3) *reinterpret_cast<uint32_t *>(mybot + 4 * j + 4) = rec;
                    mycol[k.row] = (uint8_t)(rec >> 24);
                    int16_t *cf = a.o.coeffs + ((size_t)mb * 25 + blk) * 16;
#pragma unroll
                    for (int t = 0; t < 4; ++t) cf[k.zz[t]] = (int16_t)qc[t];
                }
                if (lane == 0) a.modes[(size_t)mb * 16 + blk] = best;
                lds_order();
                if (lane == 0) done[wave] = j + 1;
            }
            if (c + 1 < mbw && lane < 16) *reinterpret_cast<uint32_t *>(mysrc + 64 * ((c + 1) & 1) + 4 * lane) = nxt;
            lds_order();
        }
        return;
    }

    // ---- wave 4: chroma (TM_PRED, 8 blocks at once) and the row's progress ---------------------------------------------
    const LaneK k = lane_consts(lane);
    const LaneQ qc_l = lane_q(st.uv_dc, st.uv_ac, lane & 3);
    int16_t *mytr = tr + 256 * 4;
    for (int c = 0; c < mbw; ++c) {
        const int mb = r * mbw + c;
        if (lane < 32) {
            const Plane &P = lane < 16 ? a.cu : a.cv;
            const int l = lane & 15;
            *reinterpret_cast<uint32_t *>(csrc + 64 * (lane >> 4) + 4 * l) =
                *reinterpret_cast<const uint32_t *>(P.p + (ptrdiff_t)(8 * r + (l >> 1)) * P.stride + 8 * c + 4 * (l & 1));
        }
        if (r > 0) K4_WAIT(prog_load(a, r - 1) < c + 1, 2)
        if (lane < 22) {
            // lanes 0..2 / 3..5: the chroma row above (x = -4 .. 7); lanes 6..13 / 14..21: the column left (first macroblock: 129)
            const bool top = lane < 6;
            const int pl = top ? lane / 3 : (lane - 6) / 8, jj = top ? lane % 3 : (lane - 6) % 8;
            const Plane &P = pl == 0 ? a.ru : a.rv;
            uint32_t v;
            if (top) {
                v = r == 0 ? 0x7f7f7f7fu : ld_agent(P.p + (ptrdiff_t)(8 * r - 1) * P.stride + 8 * c - 4 + 4 * jj);
                if (r > 0 && c == 0 && jj == 0) v = 0x81818181u;
                *reinterpret_cast<uint32_t *>(cimg + pl * 9 * CIMG_S + 4 * jj) = v;
            } else {
                // left neighbours: the right column of this wave's previous macroblock, still in the tile
                const uint8_t *t = cimg + pl * 9 * CIMG_S + (jj + 1) * CIMG_S;
                const int px = c == 0 ? 129 : t[4 + 7];
                cimg[pl * 9 * CIMG_S + (jj + 1) * CIMG_S + 3] = (uint8_t)px;
            }
        }
        lds_order();
        {
            const int l = lane & 31, pl = l >> 4, bb = (l >> 2) & 3, br = bb >> 1, bc = bb & 1, i = l & 3;
            const uint8_t *t = cimg + pl * 9 * CIMG_S;
            const uint32_t tdw = *reinterpret_cast<const uint32_t *>(t + 4 + 4 * bc);
            const int dl = (int)t[(4 * br + i + 1) * CIMG_S + 3] - (int)t[3];
            const uint32_t sdw = *reinterpret_cast<const uint32_t *>(csrc + 64 * pl + (4 * br + i) * 8 + 4 * bc);
            int p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = sat8(byte_of(tdw, j) + dl);
            int qc[4], w;
            const uint32_t rec = code_unit(mytr, lane, p, sdw, qc_l, qc, w);
            lds_order();
            if (lane < 32) {
                *reinterpret_cast<uint32_t *>(cimg + pl * 9 * CIMG_S + (4 * br + i + 1) * CIMG_S + 4 + 4 * bc) = rec;
                int16_t *cf = ccoef + (4 * pl + bb) * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) cf[k.zz[j]] = (int16_t)qc[j];
            }
            lds_order();
        }
        if (lane < 32) {
            const Plane &P = lane < 16 ? a.ru : a.rv;
            const int l = lane & 15;
            st_agent(P.p + (ptrdiff_t)(8 * r + (l >> 1)) * P.stride + 8 * c + 4 * (l & 1),
                     *reinterpret_cast<const uint32_t *>(cimg + (lane >> 4) * 9 * CIMG_S + ((l >> 1) + 1) * CIMG_S + 4 + 4 * (l & 1)));
        }
        reinterpret_cast<uint32_t *>(a.o.coeffs + ((size_t)mb * 25 + 16) * 16)[lane] = reinterpret_cast<const uint32_t *>(ccoef)[lane];   // blocks 16..23
        if (lane == 0) {
            a.o.parts[mb] = 2;   // are4x4
            a.o.seg[mb] = 0;
        }
        // publish: the luma row below block row 3 goes to HBM at agent scope (wave 3's own stores are not waited for)
        K4_WAIT(done[3] < 4 * c + 4, 1)
        if (lane < 4)
            st_agent(a.ry.p + (ptrdiff_t)(16 * r + 15) * a.ry.stride + 16 * c + 4 * lane, *reinterpret_cast<const uint32_t *>(bot + 4 * BW + 16 * c + 4 + 4 * lane));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) prog_store(a, r, c + 1);
    }
}
#undef K4_WAIT

// what check_SSIM reports (src/vp8enc.cpp:237-258): replaced count, the raster-order float sum / count, the minimum
// The sum must be the reference's: one float accumulator over the macroblocks in raster order.  The values are staged
// in LDS by all threads, 8192 at a time, so that the one summing thread reads four per ds_read_b128 instead of
// waiting for HBM once per element.
constexpr int STATS_CHUNK = 8192;
struct SsimStats { int replaced; float mean, mn; };
// all 256 threads; the result is valid in thread 0
__device__ __forceinline__ SsimStats ssim_stats_body(const float *ssim, const int32_t *is_inter, int mbs) {
    __shared__ int s_repl;
    __shared__ float s_min[256];
    __shared__ __attribute__((aligned(16))) float s_val[STATS_CHUNK];
    if (threadIdx.x == 0) s_repl = 0;
    int repl = 0;
    float mn = 2.0f, sum = 0.0f;
    for (int base = 0; base < mbs; base += STATS_CHUNK) {
        const int n = imin(STATS_CHUNK, mbs - base);
        __syncthreads();
        for (int i = threadIdx.x; i < STATS_CHUNK; i += 256) {
            float v = 0.0f;
            if (i < n) {
                v = ssim[base + i];
                repl += is_inter[base + i] == 0;
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
    atomicAdd(&s_repl, repl);
    s_min[threadIdx.x] = mn;
    __syncthreads();
    SsimStats r{0, 0.0f, 2.0f};
    if (threadIdx.x == 0) {
        for (int i = 0; i < 256; ++i) mn = s_min[i] < mn ? s_min[i] : mn;
        r.replaced = s_repl;
        r.mean = __fdiv_rn(sum, (float)mbs);
        r.mn = mn;
    }
    return r;
}

__global__ __launch_bounds__(64 * K4_WAVES) void k_intra_key4(IntraArgs a) { intra_key4_body(a); }
// the key frames of several members of a batch in one launch (blockIdx.z = member): a key frame is a wavefront of mb_h workgroups a
// millisecond and a half long, and a batch whose members all start a GOP -- every chunk of a file does, together -- ran six of them
// one after the other on its stream
__global__ __launch_bounds__(64 * K4_WAVES) void k_intra_key4_b(BatchOf<IntraArgs> b) { intra_key4_body(b.item[blockIdx.z]); }
__global__ __launch_bounds__(256) void k_ssim_stats(const float *ssim, const int32_t *is_inter, int mbs, const int32_t *err, int32_t *out) {
    const SsimStats r = ssim_stats_body(ssim, is_inter, mbs);
    if (threadIdx.x == 0) {
        out[0] = r.replaced;
        out[1] = __float_as_int(r.mean);
        out[2] = __float_as_int(r.mn);
        out[3] = *err;   // the time-out flag of the wavefront kernels travels with the statistics: one read-back
    }
}

}  // namespace

static IntraArgs intra_args(const Frame &cur, const Frame &recon, const MBOut &o, const SegData *d_sd, int32_t *modes, int32_t *is_inter,
                            int32_t *prog, unsigned gen, int32_t *err, float target, int key, int mbw, int mbh, int stall_test, int modes_of_kept) {
    IntraArgs a;
    a.cy = cur.Y[0]; a.cu = cur.U; a.cv = cur.V;
    a.ry = recon.Y[0]; a.ru = recon.U; a.rv = recon.V;
    a.o = o;
    a.sd = d_sd;
    a.modes = modes;
    a.is_inter = is_inter;
    a.prog = prog;
    a.gen_base = gen * 1024u;   // progress within a launch stays below 1024 (512 macroblocks per row at most)
    a.flagged = nullptr;
    a.err = err;
    a.target = target;
    a.key = key;
    a.mbw = mbw;
    a.mbh = mbh;
    a.stall_test = stall_test;
    a.modes_of_kept = modes_of_kept;
    return a;
}

void launch_intra(hipStream_t s, const Frame &cur, const Frame &recon, const MBOut &o, const SegData *d_sd, int32_t *modes,
                  int32_t *is_inter, int32_t *prog, unsigned gen, int32_t *err, float target, int key, int mbw, int mbh, int stall_test,
                  int modes_of_kept) {
    const IntraArgs a = intra_args(cur, recon, o, d_sd, modes, is_inter, prog, gen, err, target, key, mbw, mbh, stall_test, modes_of_kept);
    static const bool legacy_key = getenv("VP8HIP_INTRA_KEY_MB") != nullptr;   // A/B switch: key frames on the per-macroblock wavefront
    static const bool legacy_check = getenv("VP8HIP_INTRA_CHECK_1WAVE") != nullptr;   // A/B switch: the three attempts one after the other
    if (key && !legacy_key) {
        const size_t shmem = 5 * (size_t)(mbw * 16 + 16) + 3856;
        hipLaunchKernelGGL(k_intra_key4, dim3(mbh), dim3(64 * K4_WAVES), shmem, s, a);
    } else if (!key && !legacy_check) {
        hipLaunchKernelGGL(k_intra_check3, dim3(mbh), dim3(192), 0, s, a);
    } else {
        hipLaunchKernelGGL(k_intra, dim3(mbh), dim3(64), 0, s, a);
    }
}

void launch_intra_key_batch(hipStream_t s, const CheckItem *items, int n, int mbw, int mbh) {
    BatchOf<IntraArgs> b;
    b.n = n;
    for (int i = 0; i < n; ++i) {
        const CheckItem &c = items[i];
        b.item[i] = intra_args(*c.cur, *c.recon, *c.o, c.sd, c.modes, c.is_inter, c.prog, c.gen, c.err, 0.0f, 1, mbw, mbh, 0, 0);
    }
    const size_t shmem = 5 * (size_t)(mbw * 16 + 16) + 3856;
    hipLaunchKernelGGL(k_intra_key4_b, dim3(mbh, 1, n), dim3(64 * K4_WAVES), shmem, s, b);
}

void launch_ssim_stats(hipStream_t s, const MBOut &o, const int32_t *is_inter, int mbs, const int32_t *err, int32_t *out) {
    hipLaunchKernelGGL(k_ssim_stats, dim3(1), dim3(256), 0, s, o.ssim, is_inter, mbs, err, out);
}

void launch_check_fallback(hipStream_t s, const CheckItem *items, int n, float target, int mbw, int mbh, int modes_of_kept) {
    static_assert(sizeof(BatchOf<IntraArgs>) <= 4096, "a batch's argument blocks travel in the 4 KiB kernel-argument segment");
    BatchOf<IntraArgs> b;
    b.n = n;
    for (int i = 0; i < n; ++i) {
        const CheckItem &c = items[i];
        b.item[i] = intra_args(*c.cur, *c.recon, *c.o, c.sd, c.modes, c.is_inter, c.prog, c.gen, c.err, target, 0, mbw, mbh, 0, modes_of_kept);
        b.item[i].flagged = c.o->flags;
    }
    hipLaunchKernelGGL(k_intra_check3_b, dim3(mbh, 1, n), dim3(192), 0, s, b);
}

}  // namespace vp8
