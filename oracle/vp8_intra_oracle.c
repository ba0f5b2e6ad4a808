/*
 * vp8_intra_oracle.c -- CPU restatement of the reference's HOST intra path: key frames
 * (intra_transform, src/intra_part.h:1089-1109) and the per-macroblock intra fallback of inter frames
 * (check_SSIM, src/vp8enc.cpp:231-263 -> test_inter_on_intra, src/intra_part.h:855-1087).
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the checker of the HIP path (kernels_intra.hip); never linked into or called from the product.
 * Pinned against the reference's own code compiled from /root/reference (oracle/ref_host_driver.cpp ->
 * oracle/_ref/libvp8refhost.so) by tests/test_intra_oracle.py, and against tests/golden/intra/.
 *
 * Structure differs from the reference on purpose (one macroblock coder shared by both callers, table-driven
 * sub-block predictors over one edge array); the arithmetic, the order of decisions and the quirks are the
 * reference's:
 *   - quant4x4 rounds coefficient 11 by the sign of coefficient 10 (src/intra_part.h:227)
 *   - count_SSIM_16x16 does not reset its accumulators between planes (src/intra_part.h:796-851)
 *   - test_inter_on_intra overwrites e_data.mode[] even when the attempt is rejected (:970)
 *   - block 24 of a replaced macroblock is an uninitialised stack copy there (:858,1066); here it is left as it was
 */
#include <stdint.h>
#include <string.h>

#include "vp8_oracle.h"

static const uint8_t dc_qlookup[128] = {
    4,   5,   6,   7,   8,   9,   10,  10,  11,  12,  13,  14,  15,  16,  17,  17,  18,  19,  20,  20,  21,  21,
    22,  22,  23,  23,  24,  25,  25,  26,  27,  28,  29,  30,  31,  32,  33,  34,  35,  36,  37,  37,  38,  39,
    40,  41,  42,  43,  44,  45,  46,  46,  47,  48,  49,  50,  51,  52,  53,  54,  55,  56,  57,  58,  59,  60,
    61,  62,  63,  64,  65,  66,  67,  68,  69,  70,  71,  72,  73,  74,  75,  76,  76,  77,  78,  79,  80,  81,
    82,  83,  84,  85,  86,  87,  88,  89,  91,  93,  95,  96,  98,  100, 101, 102, 104, 106, 108, 110, 112, 114,
    116, 118, 122, 124, 126, 128, 130, 132, 134, 136, 138, 140, 143, 145, 148, 151, 154, 157};
static const int16_t ac_qlookup[128] = {
    4,   5,   6,   7,   8,   9,   10,  11,  12,  13,  14,  15,  16,  17,  18,  19,  20,  21,  22,  23,  24,  25,
    26,  27,  28,  29,  30,  31,  32,  33,  34,  35,  36,  37,  38,  39,  40,  41,  42,  43,  44,  45,  46,  47,
    48,  49,  50,  51,  52,  53,  54,  55,  56,  57,  58,  60,  62,  64,  66,  68,  70,  72,  74,  76,  78,  80,
    82,  84,  86,  88,  90,  92,  94,  96,  98,  100, 102, 104, 106, 108, 110, 112, 114, 116, 119, 122, 125, 128,
    131, 134, 137, 140, 143, 146, 149, 152, 155, 158, 161, 164, 167, 170, 173, 177, 181, 185, 189, 193, 197, 201,
    205, 209, 213, 217, 221, 225, 229, 234, 239, 245, 249, 254, 259, 264, 269, 274, 279, 284};

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---- forward transform shared by DCT4x4 (:114-157) and weight (:159-210); every store narrows to 16 bits ---- */
static void fdct(const int16_t in[16], int16_t out[16]) {
    for (int r = 0; r < 4; ++r) {
        const int16_t *p = in + 4 * r;
        const int a = (p[0] + p[3]) * 8, b = (p[1] + p[2]) * 8, c = (p[1] - p[2]) * 8, d = (p[0] - p[3]) * 8;
        out[4 * r + 0] = (int16_t)(a + b);
        out[4 * r + 2] = (int16_t)(a - b);
        out[4 * r + 1] = (int16_t)((c * 2217 + d * 5352 + 14500) >> 12);
        out[4 * r + 3] = (int16_t)((d * 2217 - c * 5352 + 7500) >> 12);
    }
    for (int k = 0; k < 4; ++k) {
        const int a = out[k] + out[12 + k], b = out[4 + k] + out[8 + k], c = out[4 + k] - out[8 + k], d = out[k] - out[12 + k];
        out[k] = (int16_t)((a + b + 7) >> 4);
        out[8 + k] = (int16_t)((a - b + 7) >> 4);
        out[4 + k] = (int16_t)(((c * 2217 + d * 5352 + 12000) >> 16) + (d != 0));
        out[12 + k] = (int16_t)((d * 2217 - c * 5352 + 51000) >> 16);
    }
}

/* weight, src/intra_part.h:159-210 */
int vp8o_host_weight(const int16_t r[16]) {
    int16_t t[16];
    fdct(r, t);
    t[0] = (int16_t)(t[0] / 4);
    int s = 0;
    for (int i = 0; i < 16; ++i) s += t[i] < 0 ? -t[i] : t[i];
    return s;
}

/* quant4x4, src/intra_part.h:212-250 */
static void quant(int16_t c[16], int dc_q, int ac_q) {
    int neg[16];
    for (int i = 0; i < 16; ++i) neg[i] = c[i] < 0;
    neg[11] = neg[10];   /* the reference tests coefficient 10 again; 10 keeps its sign after its own rounding */
    for (int i = 0; i < 16; ++i) {
        const int q = i ? ac_q : dc_q;
        c[i] = (int16_t)(c[i] + (neg[i] ? (-q) / 2 : q / 2));
    }
    for (int i = 0; i < 16; ++i) c[i] = (int16_t)(c[i] / (int16_t)(i ? ac_q : dc_q));
}

/* iDCT4x4 with dequantisation, src/intra_part.h:42-111: columns first, 16-bit intermediate */
static void idct_add(const int16_t c[16], const uint8_t pred[16], uint8_t out[16], int dc_q, int ac_q) {
    int16_t t[16];
    for (int k = 0; k < 4; ++k) {
        const int i0 = c[k] * (k ? ac_q : dc_q), i4 = c[4 + k] * ac_q, i8 = c[8 + k] * ac_q, i12 = c[12 + k] * ac_q;
        const int a = i0 + i8, b = i0 - i8;
        const int cc = ((i4 * 35468) >> 16) - (i12 + ((i12 * 20091) >> 16));
        const int d = (i4 + ((i4 * 20091) >> 16)) + ((i12 * 35468) >> 16);
        t[k] = (int16_t)(a + d);
        t[12 + k] = (int16_t)(a - d);
        t[4 + k] = (int16_t)(b + cc);
        t[8 + k] = (int16_t)(b - cc);
    }
    for (int r = 0; r < 4; ++r) {
        const int16_t *p = t + 4 * r;
        const int a = p[0] + p[2], b = p[0] - p[2];
        const int cc = ((p[1] * 35468) >> 16) - (p[3] + ((p[3] * 20091) >> 16));
        const int d = (p[1] + ((p[1] * 20091) >> 16)) + ((p[3] * 35468) >> 16);
        const int v[4] = {a + d, b + cc, b - cc, a - d};
        for (int k = 0; k < 4; ++k) {
            const int16_t s = (int16_t)(((v[k] + 4) >> 3) + pred[4 * r + k]);
            out[4 * r + k] = (uint8_t)clampi(s, 0, 255);
        }
    }
}

/* zigzag_block, src/intra_part.h:13-37: out[i] = in[zz[i]] */
static void zigzag(int16_t c[16]) {
    static const uint8_t zz[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
    int16_t t[16];
    memcpy(t, c, sizeof t);
    for (int i = 0; i < 16; ++i) c[i] = t[zz[i]];
}

/* ---- sub-block predictors ------------------------------------------------------------------------------------
 * Edge array e[0..14] = { L3, L3, L2, L1, L0, TL, T0, T1, T2, T3, T4, T5, T6, T7, T7 } (the doubled ends turn the
 * reference's three "3x" special cases, :330, :352, :499, into the ordinary 1-2-1 filter).  Entry = kind << 4 | k:
 *   kind 0: (e[k-1] + 2 e[k] + e[k+1] + 2) >> 2     kind 1: (e[k] + e[k+1] + 1) >> 1     kind 2: e[k]
 * Rows of the table: B_VE, B_HE, B_LD, B_RD, B_VR, B_VL, B_HD, B_HU (src/intra_part.h:296-512). */
#define F3(k) (0x00 | (k))
#define F2(k) (0x10 | (k))
#define CP(k) (0x20 | (k))
static const uint8_t bpred_tab[8][16] = {
    /* VE */ {F3(6), F3(7), F3(8), F3(9), F3(6), F3(7), F3(8), F3(9), F3(6), F3(7), F3(8), F3(9), F3(6), F3(7), F3(8), F3(9)},
    /* HE */ {F3(4), F3(4), F3(4), F3(4), F3(3), F3(3), F3(3), F3(3), F3(2), F3(2), F3(2), F3(2), F3(1), F3(1), F3(1), F3(1)},
    /* LD */ {F3(7), F3(8), F3(9), F3(10), F3(8), F3(9), F3(10), F3(11), F3(9), F3(10), F3(11), F3(12), F3(10), F3(11), F3(12), F3(13)},
    /* RD */ {F3(5), F3(6), F3(7), F3(8), F3(4), F3(5), F3(6), F3(7), F3(3), F3(4), F3(5), F3(6), F3(2), F3(3), F3(4), F3(5)},
    /* VR */ {F2(5), F2(6), F2(7), F2(8), F3(5), F3(6), F3(7), F3(8), F3(4), F2(5), F2(6), F2(7), F3(3), F3(5), F3(6), F3(7)},
    /* VL */ {F2(6), F2(7), F2(8), F2(9), F3(7), F3(8), F3(9), F3(10), F2(7), F2(8), F2(9), F3(11), F3(8), F3(9), F3(10), F3(12)},
    /* HD */ {F2(4), F3(5), F3(6), F3(7), F2(3), F3(4), F2(4), F3(5), F2(2), F3(3), F2(3), F3(4), F2(1), F3(2), F2(2), F3(3)},
    /* HU */ {F2(3), F3(3), F2(2), F3(2), F2(2), F3(2), F2(1), F3(1), F2(1), F3(1), CP(1), CP(1), CP(1), CP(1), CP(1), CP(1)},
};

static void bpred(int mode, const int e[15], uint8_t p[16]) {
    if (mode == 0) {   /* B_DC_PRED :264-273 */
        int v = 4;
        for (int i = 0; i < 4; ++i) v += e[6 + i] + e[4 - i];
        for (int i = 0; i < 16; ++i) p[i] = (uint8_t)(v >> 3);
    } else if (mode == 1) {   /* B_TM_PRED :275-283 */
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) p[4 * r + c] = (uint8_t)clampi(e[6 + c] + e[4 - r] - e[5], 0, 255);
    } else {
        const uint8_t *t = bpred_tab[mode - 2];
        for (int i = 0; i < 16; ++i) {
            const int k = t[i] & 15, kind = t[i] >> 4;
            p[i] = (uint8_t)(kind == 0 ? (e[k - 1] + 2 * e[k] + e[k + 1] + 2) >> 2 : kind == 1 ? (e[k] + e[k + 1] + 1) >> 1 : e[k]);
        }
    }
}

/* pick_luma_predictor, src/intra_part.h:252-515: first strict minimum of the weight over the ten modes in enum
 * order.  (Its early return at weight 0, :295, changes nothing: no later weight is negative.) */
int vp8o_pick_luma_predictor(const uint8_t orig[16], uint8_t pred[16], int16_t resid[16], const int16_t top[8],
                             const int16_t left[4], int top_left) {
    int e[15];
    e[0] = e[1] = left[3]; e[2] = left[2]; e[3] = left[1]; e[4] = left[0]; e[5] = top_left;
    for (int i = 0; i < 8; ++i) e[6 + i] = top[i];
    e[14] = top[7];
    int best = -1, best_w = 0;
    for (int m = 0; m < 10; ++m) {
        uint8_t p[16];
        int16_t r[16];
        bpred(m, e, p);
        for (int i = 0; i < 16; ++i) r[i] = (int16_t)(orig[i] - p[i]);
        const int w = (int16_t)vp8o_host_weight(r);
        if (best < 0 || w < best_w) {
            best = m; best_w = w;
            memcpy(pred, p, 16);
            memcpy(resid, r, 32);
        }
    }
    return best;
}

/* count_SSIM_16x16, src/intra_part.h:744-853.  The integer accumulators carry over from plane to plane: after
 * luma M1 holds |M1 - M2|, M2 the luma mean of frame 2, D1/D2/C the luma (co)variances, and the chroma sums are
 * added on top of them. */
float vp8o_count_ssim_16x16(const uint8_t *y1, const uint8_t *u1, const uint8_t *v1, int w1, const uint8_t *y2,
                            const uint8_t *u2, const uint8_t *v2, int w2) {
    const float c1 = 0.01f * 0.01f * 255 * 255, c2 = 0.03f * 0.03f * 255 * 255;
    int M1 = 0, M2 = 0, D1 = 0, D2 = 0, C = 0;
    float ssim = 0.0f;
    for (int pl = 0; pl < 3; ++pl) {
        const uint8_t *a = pl == 0 ? y1 : pl == 1 ? u1 : v1, *b = pl == 0 ? y2 : pl == 1 ? u2 : v2;
        const int n = pl ? 8 : 16, sa = pl ? w1 / 2 : w1, sb = pl ? w2 / 2 : w2, cnt = n * n;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) { M1 += a[i * sa + j]; M2 += b[i * sb + j]; }
        M1 = (M1 + cnt / 2) / cnt;
        M2 = (M2 + cnt / 2) / cnt;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const int t1 = a[i * sa + j] - M1, t2 = b[i * sb + j] - M2;
                D1 += t1 * t1; D2 += t2 * t2; C += t1 * t2;
            }
        D1 = (D1 + cnt / 2) / cnt; D2 = (D2 + cnt / 2) / cnt; C = (C + cnt / 2) / cnt;
        const float m1 = (float)M1, m2 = (float)M2;
        const float num = (m1 * m2 * 2 + c1) * ((float)C * 2 + c2);
        const float den = (m1 * m1 + m2 * m2 + c1) * ((float)D1 + (float)D2 + c2);
        if (pl == 0) ssim = num / den; else ssim += num / den;
        M1 -= M2;
        M1 = M1 < 0 ? -M1 : M1;
        ssim -= M1 > 4 ? (float)M1 * 0.02f : 0.0f;
    }
    return ssim / 3;
}

/* quantizer steps of one segment as prepare_segments_data derives them, src/vp8enc.cpp:164-187 */
typedef struct { int y_dc, y_ac, uv_dc, uv_ac; } steps_t;
static steps_t steps_of(const int32_t sd[44], int id) {
    const int base = sd[11 * id + 0];
    steps_t s;
    s.y_ac = ac_qlookup[clampi(base, 0, 127)];
    s.y_dc = dc_qlookup[clampi(base + sd[1], 0, 127)];
    s.uv_dc = dc_qlookup[clampi(base + sd[4], 0, 127)];
    s.uv_ac = ac_qlookup[clampi(base + sd[5], 0, 127)];
    if (s.uv_dc > 132) s.uv_dc = 132;
    return s;
}

typedef struct {
    int16_t coeffs[24][16];   /* raster order inside a block, not yet zigzagged */
    uint8_t y[256], u[64], v[64];
    int32_t mode[16];
} intra_mb_t;

/* One macroblock: B_PRED luma with a mode decision per 4x4 block, TM_PRED chroma, transform, quantisation,
 * reconstruction -- the body shared by predict_and_transform_mb (:517-741) and test_inter_on_intra (:855-1056).
 * Neighbours come from the reconstruction planes (tight stride `width`). */
static void intra_mb(int mb_row, int mb_col, int width, int mb_width, const uint8_t *cy, const uint8_t *cu, const uint8_t *cv,
                     const uint8_t *ry, const uint8_t *ru, const uint8_t *rv, steps_t q, intra_mb_t *o) {
    const int cw = width / 2;
    const int yo = (mb_row * width + mb_col) * 16, co = mb_row * 8 * cw + mb_col * 8;
    int16_t topY[20], leftY[16], topU[8], topV[8], leftU[8], leftV[8];
    int tlY, tlU, tlV;
    /* frame edges: 129 to the left, 127 above, 127 in the corner (:540-616) */
    for (int i = 0; i < 16; ++i) leftY[i] = mb_col ? ry[yo - 1 + i * width] : 129;
    for (int i = 0; i < 8; ++i) {
        leftU[i] = mb_col ? ru[co - 1 + i * cw] : 129;
        leftV[i] = mb_col ? rv[co - 1 + i * cw] : 129;
    }
    for (int i = 0; i < 16; ++i) topY[i] = mb_row ? ry[yo - width + i] : 127;
    for (int i = 0; i < 4; ++i) topY[16 + i] = !mb_row ? 127 : (mb_col < mb_width - 1 ? ry[yo - width + 16 + i] : topY[15]);
    for (int i = 0; i < 8; ++i) {
        topU[i] = mb_row ? ru[co - cw + i] : 127;
        topV[i] = mb_row ? rv[co - cw + i] : 127;
    }
    if (!mb_row) tlY = tlU = tlV = 127;
    else if (!mb_col) tlY = tlU = tlV = 129;
    else { tlY = ry[yo - width - 1]; tlU = ru[co - cw - 1]; tlV = rv[co - cw - 1]; }

    for (int br = 0; br < 4; ++br) {
        const int next_row_tl = leftY[4 * br + 3];
        for (int bc = 0; bc < 4; ++bc) {
            const int b = 4 * br + bc;
            uint8_t orig[16], pred[16], rec[16];
            int16_t resid[16];
            for (int i = 0; i < 4; ++i) memcpy(orig + 4 * i, cy + yo + (4 * br + i) * width + 4 * bc, 4);
            o->mode[b] = vp8o_pick_luma_predictor(orig, pred, resid, topY + 4 * bc, leftY + 4 * br, tlY);
            fdct(resid, o->coeffs[b]);
            quant(o->coeffs[b], q.y_dc, q.y_ac);
            idct_add(o->coeffs[b], pred, rec, q.y_dc, q.y_ac);
            for (int i = 0; i < 4; ++i) memcpy(o->y + (4 * br + i) * 16 + 4 * bc, rec + 4 * i, 4);
            tlY = topY[4 * bc + 3];
            for (int i = 0; i < 4; ++i) { leftY[4 * br + i] = rec[4 * i + 3]; topY[4 * bc + i] = rec[12 + i]; }
        }
        tlY = next_row_tl;
    }
    for (int pl = 0; pl < 2; ++pl) {
        const uint8_t *src = pl ? cv : cu;
        const int16_t *top = pl ? topV : topU, *left = pl ? leftV : leftU;
        const int tl = pl ? tlV : tlU;
        uint8_t *dst = pl ? o->v : o->u;
        for (int b = 0; b < 4; ++b) {
            const int br = b >> 1, bc = b & 1;
            uint8_t pred[16], rec[16];
            int16_t resid[16];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    pred[4 * i + j] = (uint8_t)clampi(top[4 * bc + j] + left[4 * br + i] - tl, 0, 255);
                    resid[4 * i + j] = (int16_t)(src[co + (4 * br + i) * cw + 4 * bc + j] - pred[4 * i + j]);
                }
            int16_t *c = o->coeffs[16 + 4 * pl + b];
            fdct(resid, c);
            quant(c, q.uv_dc, q.uv_ac);
            idct_add(c, pred, rec, q.uv_dc, q.uv_ac);
            for (int i = 0; i < 4; ++i) memcpy(dst + (4 * br + i) * 8 + 4 * bc, rec + 4 * i, 4);
        }
    }
}

static void commit_mb(const intra_mb_t *m, int mb, int mb_row, int mb_col, int width, uint8_t *ry, uint8_t *ru, uint8_t *rv,
                      int16_t *coeffs) {
    const int cw = width / 2;
    for (int i = 0; i < 16; ++i) memcpy(ry + (mb_row * 16 + i) * width + mb_col * 16, m->y + 16 * i, 16);
    for (int i = 0; i < 8; ++i) {
        memcpy(ru + (mb_row * 8 + i) * cw + mb_col * 8, m->u + 8 * i, 8);
        memcpy(rv + (mb_row * 8 + i) * cw + mb_col * 8, m->v + 8 * i, 8);
    }
    for (int b = 0; b < 24; ++b) {
        int16_t *c = coeffs + ((size_t)mb * 25 + b) * 16;
        memcpy(c, m->coeffs[b], 32);
        zigzag(c);
    }
}

/* intra_transform, src/intra_part.h:1089-1109: every macroblock in raster order, segment intra_segment (0) */
void vp8o_intra_transform(int width, int height, const uint8_t *cy, const uint8_t *cu, const uint8_t *cv, const int32_t sd[44],
                          uint8_t *ry, uint8_t *ru, uint8_t *rv, int16_t *coeffs, int32_t *parts, int32_t *seg, int32_t *modes) {
    const int mbw = width / 16, mbh = height / 16;
    const steps_t q = steps_of(sd, 0);
    intra_mb_t m;
    for (int r = 0; r < mbh; ++r)
        for (int c = 0; c < mbw; ++c) {
            const int mb = r * mbw + c;
            intra_mb(r, c, width, mbw, cy, cu, cv, ry, ru, rv, q, &m);
            commit_mb(&m, mb, r, c, width, ry, ru, rv, coeffs);
            parts[mb] = 2;   /* are4x4 */
            seg[mb] = 0;
            memcpy(modes + 16 * mb, m.mode, sizeof m.mode);
        }
}

/* vp8o_conformant (vp8_oracle.c; NOT the reference's behaviour, off by default), first half: keep the sub-block modes of
 * the attempt that was KEPT instead of those of the last attempt MADE.  The reference writes e_data.mode on every attempt
 * (src/intra_part.h:964) but coefficients and reconstruction only when the attempt is kept (:1058-1086), so a macroblock
 * whose AQ attempt was kept and whose HQ / UQ attempts then failed goes into the stream with modes that do not belong to its
 * coefficients: a decoder reconstructs something else than the encoder did (tests/test_decode_roundtrip.py shows it). */
#define g_modes_of_kept vp8o_conformant

/* check_SSIM, src/vp8enc.cpp:231-263: every macroblock below the target is tried as intra in segments AQ (2),
 * HQ (1), UQ (0), each only while it is still below; an attempt is kept when its SSIM beats the current one. */
void vp8o_check_ssim(int width, int height, float ssim_target, const uint8_t *cy, const uint8_t *cu, const uint8_t *cv,
                     const int32_t sd[44], uint8_t *ry, uint8_t *ru, uint8_t *rv, int16_t *coeffs, int32_t *parts, int32_t *seg,
                     float *ssim, int32_t *is_inter, int32_t *modes, int32_t *replaced, float *new_ssim, float *min_ssim) {
    const int mbw = width / 16, mbh = height / 16, cw = width / 2;
    static const int order[3] = {2, 1, 0};
    intra_mb_t m;
    float sum = 0.0f, mn = 2.0f;
    int repl = 0;
    memset(modes, 0, sizeof(int32_t) * 16 * (size_t)mbw * mbh);
    for (int r = 0; r < mbh; ++r)
        for (int c = 0; c < mbw; ++c) {
            const int mb = r * mbw + c;
            is_inter[mb] = 1;
            for (int k = 0; k < 3; ++k) {
                if (!(ssim[mb] < ssim_target)) continue;
                intra_mb(r, c, width, mbw, cy, cu, cv, ry, ru, rv, steps_of(sd, order[k]), &m);
                if (!g_modes_of_kept) memcpy(modes + 16 * mb, m.mode, sizeof m.mode);
                const float s = vp8o_count_ssim_16x16(m.y, m.u, m.v, 16, cy + (r * width + c) * 16, cu + r * 8 * cw + c * 8,
                                                      cv + r * 8 * cw + c * 8, width);
                if (s > ssim[mb]) {
                    parts[mb] = 2;
                    ssim[mb] = s;
                    seg[mb] = order[k];
                    commit_mb(&m, mb, r, c, width, ry, ru, rv, coeffs);
                    is_inter[mb] = 0;
                    if (g_modes_of_kept) memcpy(modes + 16 * mb, m.mode, sizeof m.mode);
                }
            }
            repl += !is_inter[mb];
            sum += ssim[mb];
            mn = ssim[mb] < mn ? ssim[mb] : mn;
        }
    *replaced = repl;
    *new_ssim = sum / (float)(mbw * mbh);
    *min_ssim = mn;
}
