// kernels_s2.hip -- quarter-pel refinement (luma_search_2step, src/GPU_kernels.cl:1094-1203): the two six-tap passes as int8
// products on the matrix cores, the block match on the vector ALUs, lane = candidate.
//
// The search kernels are VALU-issue bound on gfx950 (scripts/ubench/valu_cost.hip, profiles/r04_issue_cycles_by_opcode.json), so
// the kernel is built around what the vector pipe has to issue.  Pixels travel as signed bytes p - 128: every tap set sums to 128,
// so a filter pass computes sum((p - 128) * f) = sum(p * f) - 128 * 128 and sat_i8((sum + 64) >> 7) IS the biased byte of the
// reference's clamped sample (v_ashr_pk_i8_i32: no bias to undo).
//   * 32 lanes per 8x8 block (2 blocks per wave, 8 per workgroup), one reference per blockIdx.y.
//   * The 16 x 32-byte window around the 1x winner is staged in LDS by ONE unaligned 16-byte global load per lane (the window
//     starts at its own first byte).
//   * Horizontal pass = one v_mfma_i32_32x32x32_i8: A = the two blocks' window rows, B = the taps of the four fractional x cases
//     x 8 columns (a constant operand table, make_bh below), the rounding 64 as one more product (k = 31: mfma_round).  A lane comes out with a
//     column and four groups of four consecutive rows = four dwords of the TRANSPOSED H array, which go to LDS.
//   * Vertical pass = three MFMAs over the wave's 2 x 5 x 8 columns: A = the taps of the four y cases (make_av), B = 32 columns of
//     16 bytes, one ds_read_b128 each; a lane comes out with, per y case, one dword of the prediction [column][row half].  The
//     whole-pel cases are copies.
//   * Then lane k = candidate (dx, dy) of the 25 (+ the zero-vector candidate): the block-match metric on four 4x4 blocks, the
//     current block's share of its column pass made once per block into LDS (weight_pre_column) and added by every candidate with
//     one dot4 per quantity (weight_cols_pre, vp8hip_dev.h); the minimum over the 32 lanes by four DPP steps + row_bcast.
// The operand and result lane maps of the MFMA are pinned by scripts/ubench/mfma_i8_layout.hip, the results by the whole parity
// suite.  History per 1080p frame and reference: 32-bit multiply-adds 0.156 ms; both passes on v_dot4_i32_i8 0.091 (892 vector
// instructions per wave; git history, ae32ece^); this form 681 instructions, 56 us per chunk of three references.
#include <stdlib.h>
#include <string.h>

#include "vp8hip_dev.h"
#include "kernels_rc_dev.h"

namespace vp8 {

namespace {

constexpr uint32_t pk8(int a, int b, int c, int d) {
    return (uint32_t)(a & 255) | ((uint32_t)(b & 255) << 8) | ((uint32_t)(c & 255) << 16) | ((uint32_t)(d & 255) << 24);
}
// the six taps of quarter-pel offset -2..2 (phases 4, 6, (0), 2, 4 of the 1/8-pel table, GPU_kernels.cl:563-572); case 2 is the copy
constexpr int TAPS[5][6] = {{3, -16, 77, 77, -16, 3}, {1, -8, 36, 108, -11, 2}, {0, 0, 128, 0, 0, 0}, {2, -11, 108, 36, -8, 1}, {3, -16, 77, 77, -16, 3}};
// The two six-tap passes as products on the matrix cores (v_mfma_i32_32x32x32_i8: D[32][32] = A[32][32] . B[32][32] + C, signed bytes
// in, int32 out -- exact; lane maps checked by scripts/ubench/mfma_i8_layout.hip).  A filter pass IS a banded matrix product:
//   horizontal: D[window row][(x case, column)] = sum_k window[row][k] * BH[k][(x case, column)],  BH[k][n] = tap_xc[k - c - s]
//   vertical:   D[(y case, row)][column]        = sum_k AV[(y case, row)][k] * H[k][column],       AV[m][k] = tap_yc[k - i - s]
// (s = 1 for the positive offsets, whose six taps start one sample later).  A lane holds 16 consecutive k of one row (A) or one
// column (B): lanes 0-31 k = 0..15, lanes 32-63 k = 16..31.
struct OperandTable { uint32_t w[64][4]; };
constexpr OperandTable make_bh() {   // B of the horizontal pass: lane l -> column n = l & 31 = xi * 8 + c, k = 16 * (l >> 5) + j
    OperandTable t{};
    for (int l = 0; l < 64; ++l) {
        const int n = l & 31, xi = n >> 3, c = n & 7, xc = xi + (xi >> 1), s = xc >= 2 ? 1 : 0;
        for (int j = 0; j < 16; ++j) {
            const int k = 16 * (l >> 5) + j, tap = k - c - s;
            const int v = k == 31 ? 64 : (tap >= 0 && tap < 6) ? TAPS[xc][tap] : 0;      // k = 31: the rounding (see mfma_round)
            t.w[l][j >> 2] |= (uint32_t)(v & 255) << (8 * (j & 3));
        }
    }
    return t;
}
constexpr OperandTable make_av() {   // A of the vertical pass: lane l -> row m = l & 31 = f * 8 + i (y case f + (f >> 1)), k = 16 * (l >> 5) + j
    OperandTable t{};
    for (int l = 0; l < 64; ++l) {
        const int m = l & 31, f = m >> 3, i = m & 7, yc = f + (f >> 1), s = yc >= 2 ? 1 : 0;
        for (int j = 0; j < 16; ++j) {
            const int k = 16 * (l >> 5) + j, tap = k - i - s;
            const int v = k == 31 ? 64 : (tap >= 0 && tap < 6) ? TAPS[yc][tap] : 0;
            t.w[l][j >> 2] |= (uint32_t)(v & 255) << (8 * (j & 3));
        }
    }
    return t;
}
static __device__ __constant__ const OperandTable K_BH = make_bh();
static __device__ __constant__ const OperandTable K_AV = make_av();

constexpr int HT_XC = 36;            // dwords per x case in the transposed H array: 8 columns x 16 B + 16 B bank skew
constexpr int V_STRIDE = 20;           // dwords per candidate in the V array: 8 columns x 8 B + 16 B so that the b128 reads of 16 lanes miss each other's banks
constexpr int WIN_ROW = 8;             // dwords per row of the staged window: 20 bytes loaded, 32 so that a row half is one aligned ds_read_b128
// Every LDS array of the kernel is indexed by the block's slot g, and the two blocks of a wave sit in ITS lanes: the stages
// hand data over inside a wave, so a wave-level "my LDS writes have landed" is all the synchronisation there is
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct S2Args {
    Plane cur;
    Plane ref[3];
    const int16_t *net_in[3];
    int16_t *net_out[3];
    int32_t *bdiff[3];
    int refmap[3];
    int nrefs;       // enabled references of this context (a batched launch is sized for the largest of its contexts)
    int w, h, nblk, bw;
    uint32_t bw_inv;   // ceil(2^32 / bw)
    unsigned long long *clk;   // launch clock (launch_clock_end, vp8hip_dev.h) or nullptr; a batched launch uses its first member's
};

// minimum over the 32 lanes of a block, valid in its lanes 16..31: four DPP steps inside the 16-lane rows (the lanes a step
// pairs hold, after it, the same value: mirrors do as well as butterflies), then lane 15 of the lower row into the upper one.
// (__shfl_xor is a ds_bpermute_b32 with five instructions of address arithmetic around it, per step.)
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);   // lanes without a source keep their own value
}
// (a lane without a source takes the minimum's identity: written so, hipcc folds the move into the minimum -- one v_min_u32_dpp per step where
// "keeps its own value" cost a copy, the move and the minimum)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_or_max(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t halfwave_min_upper(uint32_t key) {
    key = umin(key, dpp_or_max<0xb1>(key));         // quad_perm [1,0,3,2]
    key = umin(key, dpp_or_max<0x4e>(key));         // quad_perm [2,3,0,1]
    key = umin(key, dpp_or_max<0x141>(key));        // row_half_mirror
    key = umin(key, dpp_or_max<0x140>(key));        // row_mirror
    return umin(key, dpp_or_max<0x142, 0xa>(key));  // row_bcast:15 into rows 1 and 3
}

// sat_i8(a >> 7) in byte 0, sat_i8(b >> 7) in byte 1 (v_ashr_pk_i8_i32; as a 16-bit value the undefined bits 31:16 stay explicit).
// Pixels travel as signed bytes p - 128 and every tap set sums to 128, so a pass computes sum((p-128) f) = sum(p f) - 128 * 128, and
//     sat_i8((sum(p f) - 16384 + 64) >> 7) = sat_u8((sum(p f) + 64) >> 7) - 128:
// the signed saturation of the biased sum IS the biased byte of the reference's clamped sample; the rounding 64 rides in the product (mfma_round).
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short ashr7_pk_i8(int a, int b) { return __builtin_amdgcn_ashr_pk_i8_i32(a, b, 7); }
__device__ __forceinline__ uint32_t pack4(unsigned short lo, unsigned short hi) {
    const us2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, v);
}
// four consecutive results of one lane (rows 4q .. 4q + 3 of its column) as four biased bytes
__device__ __forceinline__ uint32_t round_pack4(const v16i &acc, int q) {
    return pack4(ashr7_pk_i8(acc[4 * q], acc[4 * q + 1]), ashr7_pk_i8(acc[4 * q + 2], acc[4 * q + 3]));
}
// The rounding 64 of a pass rides in the product: k = 31 of the tap operand is 64 (make_bh, make_av; no tap reaches that far) and k = 31 of
// the data operand is made 1 here -- byte 15 of the lanes of the upper k half, which otherwise meets a zero tap.  (As the C input the
// constant cost sixteen registers: hipcc does not fold a splat into the instruction's inline-constant field.)  `upper`: 0 or -1.
__device__ __forceinline__ v4i one_at_k31(v4i d, int upper) {
    d.w = (int)(((uint32_t)d.w & (0xffffffffu >> (8 & upper))) | (0x01000000u & (uint32_t)upper));
    return d;
}
__device__ __forceinline__ v16i mfma_round(v4i a, v4i b) {
    const v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}

// SPREAD: the cost phase's 26 x 4 (candidate, 4x4 block) tasks of a block spread over ALL 32 lanes of its half-wave and, for what is left, over
// one wave of the workgroup.  With lane = candidate, 6 of a block's 32 lanes idle through four rounds of the metric (the wave issues them all the
// same); here the idle lanes take the FOURTH 4x4 block of candidates 0..17 during the first three rounds, and the fourth 4x4 block of candidates
// 18..25 of all eight blocks -- 64 tasks -- is one round of wave 0 alone: 3.25 rounds per wave on average instead of 4.  The costs meet in LDS
// (in the H array's dead bytes), two workgroup barriers around wave 0's extra round.  Same integer sums in another order: the same result.
// tn: the thread's number again, for what a workgroup that takes several groups (search2_groups) should NOT keep across its loop: every
// register of lane arithmetic kept is one a wave of the SIMD cannot have, and the cheap ones (two or three instructions to make) are made again
template <bool SPREAD>
__device__ __forceinline__ void search2_body(const S2Args &a, int wg_x, int ref_idx, int tn) {
    if (ref_idx >= a.nrefs) return;
    __shared__ __attribute__((aligned(16))) uint32_t s_HT[8][5 * HT_XC];
    __shared__ __attribute__((aligned(16))) uint32_t s_cz[8][32];   // [0..15] current block, [16..31] zero-MV block, both as [column][row half], biased bytes
    __shared__ __attribute__((aligned(16))) uint32_t s_V[8][25 * V_STRIDE];   // vertical pass results: [y case * 5 + x case][column][row half]
    uint32_t(*s_win)[25 * V_STRIDE] = s_V;   // the staged window (16 rows x 32 B) is dead once the horizontal pass has read it: same bytes
    // the current block's share of the metric, [4x4 block][column][R0,R2,X,Y]: made after the vertical pass, in the bytes of the H array
    // (dead by then; the LDS per workgroup decides how many workgroups a CU holds: 22.8 KB = seven, with an array of its own six --
    // 68.9-69.1 against 68.4-68.9 M MB/s on one box, scripts/ab_build.sh; eight, with the V array's bank skew given up, adds nothing)
    int(*s_pre)[5 * HT_XC] = reinterpret_cast<int(*)[5 * HT_XC]>(s_HT);
    const int r = a.refmap[ref_idx];
    const int g = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const int b = imin(wg_x * 8 + g, a.nblk - 1);
    const bool live = wg_x * 8 + g < a.nblk;
    const int by = a.bw == 1 ? b : (int)__umulhi((uint32_t)b, a.bw_inv), bx = b - by * a.bw;   // b / bw: bw_inv = ceil(2^32 / bw), exact for b * bw < 2^32
    const int cx = bx * 8, cy = by * 8;
    const uint32_t nv = reinterpret_cast<const uint32_t *>(a.net_in[r])[b];
    const int nx = (int16_t)(nv & 0xffffu), ny = (int16_t)(nv >> 16);
    const int v0x = (int16_t)(nx * 4), v0y = (int16_t)(ny * 4);
    // window origin; a garbage vector (possible only when every candidate is out of frame) is clamped
    // so that the loads stay inside the allocated margin
    const int Lx = iclamp(cx + nx, 3 - EXT, a.w + EXT - 11), Ly = iclamp(cy + ny, 3 - EXT, a.h + EXT - 11);
    const Plane rf = a.ref[r];
    const int wl = threadIdx.x & 63, kh = wl >> 5, gp = g & ~1;   // lane of the wave, its k half in an MFMA operand, the wave's first slot
    const int g_n = tn >> 5, lane_n = tn & 31, wl_n = tn & 63, kh_n = wl_n >> 5, gp_n = g_n & ~1;
    // The window from its own first byte (Lx - 3: any alignment; the part's global loads need none): lane = (row, half) takes 16
    // bytes, 16 rows of 32 bytes -- the six-tap passes need 14 x 19, the rest stays inside the planes' allocated margin (PAD) and
    // meets zero taps.  One load and one ds_write_b128 per lane, no loop.
    {
        const int row = lane_n >> 1, half = lane_n & 1;
        v4i v;
        __builtin_memcpy(&v, rf.p + (ptrdiff_t)(Ly - 3 + row) * rf.stride + (Lx - 3) + 16 * half, 16);
        *reinterpret_cast<v4i *>(&s_win[g_n][row * WIN_ROW + 4 * half]) = v ^ (int)0x80808080u;
    }
    {   // current block (lanes 0-15) and zero-MV block (16-31): one dword each, scattered as column bytes
        const int sel = lane_n >> 4, row = (lane_n >> 1) & 7, half = lane_n & 1;
        const uint8_t *base = sel ? rf.p : a.cur.p;
        const int stride = sel ? rf.stride : a.cur.stride;
        const uint32_t v = *reinterpret_cast<const uint32_t *>(base + (ptrdiff_t)(cy + row) * stride + cx + 4 * half) ^ 0x80808080u;
        uint8_t *cz = reinterpret_cast<uint8_t *>(s_cz[g_n]) + sel * 64 + half * 32 + row;
#pragma unroll
        for (int j = 0; j < 4; ++j) cz[j * 8] = (uint8_t)(v >> (8 * j));
    }
    lds_fence();

    // ---- horizontal pass: ONE MFMA for the wave's two blocks ------------------------------------------------------------
    // A = the windows (row m = 16 * block + window row; rows 14, 15 and bytes 20..31 of a row hold whatever the LDS held: they
    // meet zero taps or land in bytes nobody reads), B = the taps of the four fractional x cases (K_BH).  A lane comes out with
    // column n = (x case, c) and four groups of four consecutive rows: each group one dword of the TRANSPOSED H array.
    {
        const int m = wl & 31, m_n = wl_n & 31;
        const v4i aw = *reinterpret_cast<const v4i *>(&s_win[gp_n + (m_n >> 4)][(m_n & 15) * WIN_ROW + 4 * kh_n]);
        const v4i bh = *reinterpret_cast<const v4i *>(K_BH.w[wl_n]);
        const v16i acc = mfma_round(one_at_k31(aw, -kh_n), bh);
        const int xi = m >> 3, c = m & 7, xc = xi + (xi >> 1);
        uint32_t *ht = &s_HT[gp][xc * HT_XC + c * 4 + kh];   // rows 4 * kh .. of column c; + 2: rows 8 + 4 * kh ..; next slot: the other block
#pragma unroll
        for (int q = 0; q < 4; ++q) ht[(q >> 1) * (5 * HT_XC) + 2 * (q & 1)] = round_pack4(acc, q);
    }
    {   // whole-pel x case: column c of the window, rows 4*rg..4*rg+3 (rows 14, 15 -- staged like the others -- only ever meet zero taps)
        const int c = lane_n & 7, rg = lane_n >> 3;
        const uint8_t *wb = reinterpret_cast<const uint8_t *>(s_win[g_n]) + 3 + c + rg * (16 * WIN_ROW);
        uint32_t v = 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) v |= (uint32_t)wb[rr * (4 * WIN_ROW)] << (8 * rr);
        s_HT[g_n][2 * HT_XC + c * 4 + rg] = v;
    }
    lds_fence();

    // ---- vertical pass: three MFMAs for the wave's 2 x 5 x 8 columns ------------------------------------------------------
    // A = the taps of the four fractional y cases (K_AV: row m = (y case, output row)), B = 32 columns of the H arrays, 16 bytes
    // each (the lanes of the upper k half read the same column: their A entries are zero).  A lane comes out with its column and, per
    // y case, the four rows 4 * kh .. 4 * kh + 3: one dword of the prediction [column][row half].  The whole-pel y case is a copy.
    {
        const v4i av = *reinterpret_cast<const v4i *>(K_AV.w[wl_n]);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int ng = 32 * t + (wl & 31);
            const bool on = t < 2 || (wl & 31) < 16;
            const int blk = ng >= 40 ? 1 : 0, rem = on ? ng - 40 * blk : 0, xc = rem >> 3, c = rem & 7;
            const v4i hv = *reinterpret_cast<const v4i *>(&s_HT[gp + blk][xc * HT_XC + c * 4]);
            const v16i acc = mfma_round(av, one_at_k31(hv, -kh_n));
            if (on) {
                uint32_t *sv = &s_V[gp + blk][xc * V_STRIDE + c * 2 + kh];
#pragma unroll
                for (int f = 0; f < 4; ++f) sv[(f + (f >> 1)) * 5 * V_STRIDE] = round_pack4(acc, f);
                // whole-pel dy: rows 3..10 of the column, the lower lane half rows 3..6, the upper 7..10
                sv[2 * 5 * V_STRIDE] = kh ? __builtin_amdgcn_alignbyte((uint32_t)hv[2], (uint32_t)hv[1], 3)
                                          : __builtin_amdgcn_alignbyte((uint32_t)hv[1], (uint32_t)hv[0], 3);
            }
        }
    }
    lds_fence();
    {   // the current block's share of the metric (vp8hip_dev.h, weight_pre_column): 16 columns x 4 quantities, two per lane.
        // Order = the order the cost loop below walks the 4x4 blocks: q = (m*2 + n)*4 + j  <->  column 4n+j, row half m
        const int q = lane_n & 15, m = q >> 3, n = (q >> 2) & 1, j = q & 3;
        const uint32_t ccol = s_cz[g_n][(4 * n + j) * 2 + m];
        int *pre = &s_pre[g_n][q * 4 + (lane_n >> 4) * 2];
        pre[0] = dot4s(ccol, lane_n < 16 ? K_W_R0 : K_W_X, 0);
        pre[1] = dot4s(ccol, lane_n < 16 ? K_W_R2 : K_W_Y, 0);
    }
    lds_fence();

    // ---- cost: lane = candidate --------------------------------------------------------------------
    const int k = lane;
    const int dx = k % 5 - 2, dy = k / 5 - 2;
    int qx = (int16_t)(cx * 4 + v0x + dx), qy = (int16_t)(cy * 4 + v0y + dy);
    if (k == 25) { qx = cx * 4; qy = cy * 4; }
    const bool valid = live && k < 26 && qx >= 0 && qx <= a.w * 4 - 32 && qy >= 0 && qy <= a.h * 4 - 32;
    int diff = 0;
    if (SPREAD) {
        int *q3 = &s_pre[g][64];          // [candidate]: the cost of its fourth 4x4 block, made by another lane (the H array's bytes behind the pre table)
        const bool helper = lane >= 26;
        // (branch-free: a lane's own candidate or, for the idle lanes, candidate hb + j; the idle lanes' costs go to q3, everybody else's
        // store lands in the table's unused word 31)
        const uint32_t *own = k < 25 ? &s_V[g][k * V_STRIDE] : &s_cz[g][16];
        const uint32_t *hsrc = &s_V[g][(lane - 26) * 3 * V_STRIDE + 9];     // idle lanes: candidate 3 (lane - 26) + j, 4x4 block 3 (n = m = 1)
        int *hdst = &q3[(lane - 26) * 3];
        auto metric = [&](const uint32_t *src, const int *pre16) {
            int pre[16];
            uint32_t pp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int4 v = *reinterpret_cast<const int4 *>(pre16 + 4 * j);
                pre[4 * j] = v.x; pre[4 * j + 1] = v.y; pre[4 * j + 2] = v.z; pre[4 * j + 3] = v.w;
                pp[j] = src[2 * j];
            }
            return weight_cols_pre(pre, pp);
        };
#pragma unroll
        for (int j = 0; j < 3; ++j) {       // rounds 0..2: a candidate's own lane its 4x4 blocks 0..2, the idle lanes block 3 of candidates 0..17
            const int c = metric(helper ? hsrc + j * V_STRIDE : own + (8 * (j & 1) + (j >> 1)), &s_pre[g][helper ? 48 : 16 * j]);
            *(helper ? hdst + j : &q3[31]) = c;
            diff += helper ? 0 : c;
        }
        __syncthreads();                    // every wave's V / pre / zero-MV arrays (and the idle lanes' costs) are in LDS
        if ((int)(threadIdx.x >> 6) == (wg_x & 3)) {      // ONE wave (taking turns from workgroup to workgroup: the waves of a workgroup sit on different SIMDs): block 3 of candidates 18..25 of all eight block slots
            const int slot = wl >> 3, cand = 18 + (wl & 7);
            s_pre[slot][64 + cand] = metric((cand < 25 ? &s_V[slot][cand * V_STRIDE] : &s_cz[slot][16]) + 9, &s_pre[slot][48]);
        }
        __syncthreads();
        diff += q3[k < 26 ? k : 31];
    } else {
    uint32_t P[8][2];
    {
        // candidates 0..24: their prediction from the producer; lane 25 (zero MV: whole-pel, both passes are the identity)
        // and the idle lanes read the zero-MV block
        const uint32_t *src = k < 25 ? &s_V[g][k * V_STRIDE] : &s_cz[g][16];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + 4 * c4);
            P[2 * c4][0] = v.x; P[2 * c4][1] = v.y; P[2 * c4 + 1][0] = v.z; P[2 * c4 + 1][1] = v.w;
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            int pre[16];
            uint32_t pp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int4 v = *reinterpret_cast<const int4 *>(&s_pre[g][((m * 2 + n) * 4 + j) * 4]);
                pre[4 * j] = v.x; pre[4 * j + 1] = v.y; pre[4 * j + 2] = v.z; pre[4 * j + 3] = v.w;
                pp[j] = P[4 * n + j][m];
            }
            diff += weight_cols_pre(pre, pp);
        }
    }
    if (k < 25) diff += (iabs(dx) + iabs(dy)) * 32;  // :1176-1178
    uint32_t key = (valid && diff < 0x7fff) ? ((uint32_t)diff << 8) | (uint32_t)k : 0xffffffffu;
    key = halfwave_min_upper(key);
    if (lane == 16 && live) {   // (every lane of the block holds the block's position and vector; this one also the minimum)
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

// ITER: a workgroup takes ITER consecutive groups of eight blocks, one after the other.  A third of what a wave issues for a group does not
// depend on the group -- the lane's place in the operand tables of the two passes and the tables themselves, its LDS addresses in every stage, its
// candidate's offset and penalty -- and the compiler keeps all of it in registers across the loop (the price: registers, i.e. waves per SIMD).
template <bool SPREAD, int ITER>
__device__ __forceinline__ void search2_groups(const S2Args &a, int grp, int ref_idx) {
    if (ITER == 1) { search2_body<SPREAD>(a, grp, ref_idx, (int)threadIdx.x); return; }
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        int tn = (int)threadIdx.x;
        asm volatile("" : "+v"(tn));     // (the same number, as far as the compiler can tell a new one per group)
        search2_body<SPREAD>(a, grp * ITER + it, ref_idx, tn);
    }
}
template <bool SPREAD, int ITER>
__global__ __launch_bounds__(256, 4) void k_search2(S2Args a) {   // (a register budget of 128 also makes the MFMAs write VGPRs: no v_accvgpr_read per result)
    launch_clock_begin(a.clk);
    search2_groups<SPREAD, ITER>(a, xcd_band(blockIdx.x, gridDim.x), blockIdx.y);
    launch_clock_end(a.clk);
}
static_assert(sizeof(BatchOf<S2Args>) <= 4096, "a batch's argument blocks travel in the 4 KiB kernel-argument segment");
template <bool SPREAD, int ITER>
__global__ __launch_bounds__(256, 4) void k_search2_b(BatchOf<S2Args> b) {
    launch_clock_begin(b.item[0].clk);
    search2_groups<SPREAD, ITER>(b.item[blockIdx.z], xcd_band(blockIdx.x, gridDim.x), blockIdx.y);
    launch_clock_end(b.item[0].clk);
}
// The same launch carrying the loop-filter strength scans of its members' NEW frames (kernels_rc_dev.h): the workgroups behind the
// nbx block groups of reference 0.  With the part full a launch of the scan's own holds the batch's stream for 0.4-0.7 ms where its
// work is 15 us (every workgroup waits for a place) -- the headline without that launch: +4 %, scripts/ab_flags_experiments.sh -- and
// the shorter launches of the chain cannot absorb it either (in the pyramid's launch it made THAT link the long one: -2 %).  This is
// the frame's longest launch, the scan's result is wanted by k_mb, which comes next, and the scan's 136 workgroups per frame are
// spread through the launch's 12 000 per member.
struct S2Scans { rc::ScanCore item[MAX_BATCH]; uint32_t mask; int nbx, wgs; };
static_assert(sizeof(BatchOf<S2Args>) + sizeof(S2Scans) <= 4096, "the kernel-argument segment");
// (waves_per_eu: with the scans' body in the same function the register allocator otherwise settles at 76 -- six waves per SIMD; told to, it fits the same code into 70)
template <bool SPREAD, int ITER>
__global__ __attribute__((amdgpu_waves_per_eu(7, 8))) __launch_bounds__(256) void k_search2_bs(BatchOf<S2Args> b, S2Scans sc) {
    if ((int)blockIdx.x >= sc.nbx) {
        const int wg = (int)blockIdx.x - sc.nbx;
        if (blockIdx.y == 0 && ((sc.mask >> blockIdx.z) & 1) && wg < sc.wgs) rc::strength_segments_body(b.item[blockIdx.z].cur, sc.item[blockIdx.z], wg, sc.wgs);
        return;
    }
    launch_clock_begin(b.item[0].clk);
    search2_groups<SPREAD, ITER>(b.item[blockIdx.z], xcd_band(blockIdx.x, sc.nbx), blockIdx.y);
    launch_clock_end(b.item[0].clk);
}
// Persistent form: a grid no larger than what the part holds at once, every workgroup walking the (context, reference,
// block group) space with a stride.  A command-processor pipe stays busy with a launch until its last workgroup is
// placed -- for a grid of tens of thousands of workgroups on a full chip that is the kernel's whole duration, and the
// pipe's other queues wait (the kernel trace of eight busy streams shows 3.6 kernels running and every stream idle half of
// the time, ~0.26 ms between a kernel's end and its successor's start).  A grid that fits is placed at once.
__global__ __launch_bounds__(256, 4) void k_search2_p(BatchOf<S2Args> b, int nbx, int maxrefs, int total) {
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const int item = w / (nbx * maxrefs), rem = w - item * (nbx * maxrefs);
        const int ref_idx = rem / nbx, wg_x = rem - ref_idx * nbx;
        search2_body<false>(b.item[item], wg_x, ref_idx, (int)threadIdx.x);
        lds_fence();   // the next round reuses this workgroup's LDS: every read of this round has returned
    }
}

}  // namespace

static S2Args search2_args(const Frame &cur, const RefSet &refs, const NetSet &nets, unsigned long long *clk) {
    S2Args a;
    a.clk = clk;
    a.cur = cur.Y[0];
    int n = 0;
    for (int r = 0; r < 3; ++r) {
        a.ref[r] = refs.ref[r].Y[0];
        a.net_in[r] = nets.net[r][1];   // vnet2 holds the 1x result, init.h:832-854
        a.net_out[r] = nets.net[r][0];
        a.bdiff[r] = nets.bdiff[r];
        if (refs.use[r]) a.refmap[n++] = r;
    }
    a.nrefs = n;
    for (int i = n; i < 3; ++i) a.refmap[i] = 0;
    a.w = a.cur.w;
    a.h = a.cur.h;
    a.bw = a.w / 8;
    a.bw_inv = (uint32_t)(((1ull << 32) + a.bw - 1) / a.bw);
    a.nblk = a.w * a.h / 64;
    return a;
}
// VP8HIP_S2_SPREAD=0: lane = candidate through all four rounds, as it was (same-box A/B runs)
static bool search2_spread() {
    static const bool on = [] { const char *v = getenv("VP8HIP_S2_SPREAD"); return !(v && v[0] == '0'); }();
    return on;
}
// VP8HIP_S2_ITER=1/2/4: groups of eight blocks a workgroup takes one after the other (same-box A/B runs).  Batches: four.  One video: one --
// its launch is a few rounds of workgroups long, and workgroups four times as long make its end ragged (53.0 against 54.3 us)
static int search2_iter(bool batch) {
    static const int forced = [] { const char *v = getenv("VP8HIP_S2_ITER"); const int k = v && v[0] ? atoi(v) : 0; return k == 1 || k == 2 || k == 4 ? k : 0; }();
    return forced ? forced : (batch ? 4 : 1);
}
static bool search2_skip() {
    static const bool skip = experiment_skip("s2");
    return skip;   // timing experiment only
}

void launch_search2(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, unsigned long long *clk) {
    const S2Args a = search2_args(cur, refs, nets, clk);
    if (a.nrefs == 0 || search2_skip()) return;
    const int nbx = (a.nblk + 7) / 8;
    if (!search2_spread()) VP8_LAUNCH((k_search2<false, 1>), dim3(nbx, a.nrefs), dim3(256), 0, s, a);
    else if (search2_iter(false) == 4) VP8_LAUNCH((k_search2<true, 4>), dim3((nbx + 3) / 4, a.nrefs), dim3(256), 0, s, a);
    else if (search2_iter(false) == 2) VP8_LAUNCH((k_search2<true, 2>), dim3((nbx + 1) / 2, a.nrefs), dim3(256), 0, s, a);
    else VP8_LAUNCH((k_search2<true, 1>), dim3(nbx, a.nrefs), dim3(256), 0, s, a);
}

bool launch_search2_batch(hipStream_t s, const Frame *const *cur, const RefSet *refs, const NetSet *const *nets, int n, unsigned long long *clk,
                          const ScanRequest *const *scan) {
    BatchOf<S2Args> b;
    b.n = n;
    int maxrefs = 0;
    for (int i = 0; i < n; ++i) {
        b.item[i] = search2_args(*cur[i], refs[i], *nets[i], clk);
        maxrefs = b.item[i].nrefs > maxrefs ? b.item[i].nrefs : maxrefs;
    }
    if (maxrefs == 0 || search2_skip()) return false;
    const int nbx = (b.item[0].nblk + 7) / 8;
    const int persist = persistent_workgroups();
    if (persist > 0 && nbx * maxrefs * n > persist) {
        VP8_LAUNCH(k_search2_p, dim3(persist), dim3(256), 0, s, b, nbx, maxrefs, nbx * maxrefs * n);
        return false;
    }
    S2Scans sc;
    sc.mask = 0;
    for (int i = 0; scan && i < n; ++i) {
        if (!scan[i]) continue;
        const ScanRequest &q = *scan[i];
        const Plane &y = b.item[i].cur;
        sc.mask |= 1u << i;
        sc.item[i] = rc::ScanCore{q.partial, q.partial + 2 * rc::MAX_PARTIALS, q.stats, q.sd, q.strength_out,
                                  rc::SegArgs{y.w * y.h, (y.h - 1) * (y.w - 1), q.is_key, q.refqi[0], q.refqi[1], q.refqi[2], q.refqi[3], q.qi_min}};
    }
    const int iter = search2_spread() ? search2_iter(true) : 1, ngrp = (nbx + iter - 1) / iter;     // workgroups that search: each takes `iter` groups of eight blocks
    if (!sc.mask) {
        if (!search2_spread()) VP8_LAUNCH((k_search2_b<false, 1>), dim3(nbx, maxrefs, n), dim3(256), 0, s, b);
        else if (iter == 4) VP8_LAUNCH((k_search2_b<true, 4>), dim3(ngrp, maxrefs, n), dim3(256), 0, s, b);
        else if (iter == 2) VP8_LAUNCH((k_search2_b<true, 2>), dim3(ngrp, maxrefs, n), dim3(256), 0, s, b);
        else VP8_LAUNCH((k_search2_b<true, 1>), dim3(nbx, maxrefs, n), dim3(256), 0, s, b);
        return false;
    }
    sc.nbx = ngrp;
    sc.wgs = (b.item[0].h + rc::ROWS_PER_BLOCK - 1) / rc::ROWS_PER_BLOCK;
    if (!search2_spread()) VP8_LAUNCH((k_search2_bs<false, 1>), dim3(nbx + sc.wgs, maxrefs, n), dim3(256), 0, s, b, sc);
    else if (iter == 4) VP8_LAUNCH((k_search2_bs<true, 4>), dim3(ngrp + sc.wgs, maxrefs, n), dim3(256), 0, s, b, sc);
    else if (iter == 2) VP8_LAUNCH((k_search2_bs<true, 2>), dim3(ngrp + sc.wgs, maxrefs, n), dim3(256), 0, s, b, sc);
    else VP8_LAUNCH((k_search2_bs<true, 1>), dim3(nbx + sc.wgs, maxrefs, n), dim3(256), 0, s, b, sc);
    return true;
}

}  // namespace vp8
