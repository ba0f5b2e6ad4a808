/*
 * vp8_oracle.c -- CPU restatement of the vp8oclenc inter-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY (parity checker + timed CPU baseline); see vp8_oracle.h.
 * Written from the semantics of the reference kernels, function by function; every
 * function cites the reference lines it follows (paths relative to /root/reference).
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -fwrapv).
 */
#include "vp8_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/GPU_kernels.cl:58-80 */
static const int dc_qlookup[128] = {
    4,   5,   6,   7,   8,   9,   10,  10,  11,  12,  13,  14,  15,  16,  17,  17,  18,  19,  20,  20,  21,  21,
    22,  22,  23,  23,  24,  25,  25,  26,  27,  28,  29,  30,  31,  32,  33,  34,  35,  36,  37,  37,  38,  39,
    40,  41,  42,  43,  44,  45,  46,  46,  47,  48,  49,  50,  51,  52,  53,  54,  55,  56,  57,  58,  59,  60,
    61,  62,  63,  64,  65,  66,  67,  68,  69,  70,  71,  72,  73,  74,  75,  76,  76,  77,  78,  79,  80,  81,
    82,  83,  84,  85,  86,  87,  88,  89,  91,  93,  95,  96,  98,  100, 101, 102, 104, 106, 108, 110, 112, 114,
    116, 118, 122, 124, 126, 128, 130, 132, 134, 136, 138, 140, 143, 145, 148, 151, 154, 157,
};
static const int ac_qlookup[128] = {
    4,   5,   6,   7,   8,   9,   10,  11,  12,  13,  14,  15,  16,  17,  18,  19,  20,  21,  22,  23,  24,  25,
    26,  27,  28,  29,  30,  31,  32,  33,  34,  35,  36,  37,  38,  39,  40,  41,  42,  43,  44,  45,  46,  47,
    48,  49,  50,  51,  52,  53,  54,  55,  56,  57,  58,  60,  62,  64,  66,  68,  70,  72,  74,  76,  78,  80,
    82,  84,  86,  88,  90,  92,  94,  96,  98,  100, 102, 104, 106, 108, 110, 112, 114, 116, 119, 122, 125, 128,
    131, 134, 137, 140, 143, 146, 149, 152, 155, 158, 161, 164, 167, 170, 173, 177, 181, 185, 189, 193, 197, 201,
    205, 209, 213, 217, 221, 225, 229, 234, 239, 245, 249, 254, 259, 264, 269, 274, 279, 284,
};
/* src/GPU_kernels.cl:563-572: VP8 six-tap filters indexed by 1/8-pel phase */
static const int sixtap[8][6] = {
    {0, 0, 128, 0, 0, 0},   {0, -6, 123, 12, -1, 0}, {2, -11, 108, 36, -8, 1}, {0, -9, 93, 50, -6, 0},
    {3, -16, 77, 77, -16, 3}, {0, -6, 50, 93, -9, 0},  {1, -8, 36, 108, -11, 2}, {0, -1, 12, 123, -6, 0},
};
static const int inv_zigzag[16] = {0, 1, 5, 6, 2, 4, 7, 12, 3, 8, 11, 13, 9, 10, 14, 15};

static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int sat8(int v) { return clampi(v, 0, 255); }
static inline int clamp_qi(int v) { return clampi(v, 0, 127); }

int vp8o_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* tests at 1080p / 4K raise the team size after the library was loaded (OMP_NUM_THREADS is read once, at load) */
void vp8o_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- weight_opt, src/GPU_kernels.cl:85-190 ------------------------------------------------
 * Column pass keeps the reference's quirk: b1 is discarded, rows 1 and 3 use the raw r2. */
int vp8o_weight(const int d[16]) {
    int R[16];
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
    for (int i = 0; i < 4; ++i) {
        const int e0 = R[4 * i], e1 = R[4 * i + 1], e2 = R[4 * i + 2], e3 = R[4 * i + 3];
        const int a1 = e0 + e3, d1 = e0 - e3, b1 = e1 + e2, c1 = e1 - e2;
        const int o0 = (a1 + b1 + 7) >> 4;
        const int o2 = (a1 - b1 + 7) >> 4;
        const int o1 = ((c1 * 2217 + d1 * 5352 + 12000) >> 16) + (d1 != 0);
        const int o3 = (d1 * 2217 - c1 * 5352 + 51000) >> 16;
        sum += (i == 0 ? iabs(o0) / 4 : iabs(o0)) + iabs(o1) + iabs(o2) + iabs(o3);
    }
    return sum;
}

/* ---- downsample_x2, src/GPU_kernels.cl:429-451 -------------------------------------------- */
void vp8o_downsample_x2(const uint8_t *src, uint8_t *dst, int src_w, int src_h) {
    const int n = src_w * src_h / 4;
    const int hw = src_w / 2;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < n; ++b) {
        const int x = (b % hw) * 2, y = (b / hw) * 2;
        const int i = y * src_w + x;
        const int s = src[i] + src[i + 1] + src[i + src_w] + src[i + src_w + 1] + 2;
        dst[(y / 2) * hw + x / 2] = (uint8_t)(s / 4);
    }
}

/* ---- luma_search_1step, src/GPU_kernels.cl:459-560 ---------------------------------------- */
static const int dx1[4] = {0, 0, 4, 4}, dy1[4] = {0, 4, 0, 4}; /* :456-457 */

static inline int weight4x4_u8(const uint8_t *a, int sa, const uint8_t *b, int sb) {
    int d[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) d[4 * r + c] = (int)a[r * sa + c] - (int)b[r * sb + c];
    return vp8o_weight(d);
}

void vp8o_luma_search_1step(const uint8_t *cur, const uint8_t *ref, const int16_t *src_net, int16_t *dst_net,
                            int net_width, int width, int height, int pixel_rate) {
    const int cut_width = (width / 8) * 8;
    const int nblk = (width / 8) * (height / 8);
#pragma omp parallel for schedule(dynamic, 64)
    for (int b = 0; b < nblk; ++b) {
        const int16_t cx = (int16_t)((b % (cut_width / 8)) * 8);
        const int16_t cy = (int16_t)((b / (cut_width / 8)) * 8);
        /* parent cell of the previous (coarser) level, :495-500 */
        const int16_t hx = (int16_t)(cx / 2), hy = (int16_t)(cy / 2);
        const int parent = (hy / 8) * net_width + (hx / 8);
        int16_t v0x = (int16_t)(src_net[2 * parent] / (int16_t)pixel_rate);
        int16_t v0y = (int16_t)(src_net[2 * parent + 1] / (int16_t)pixel_rate);
        if (pixel_rate > 8) v0x = v0y = 0;
        const int cell = (cy / 8) * net_width + (cx / 8);

        uint16_t MinDiff = 0x7fff;
        int16_t bx = v0x, by = v0y; /* "vector" starts as the scaled parent, :501,:551-556 */
        for (int dxy = 0; dxy < 25; ++dxy) {
            const int16_t px = (int16_t)(cx + v0x + ((dxy % 5) - 2));
            const int16_t py = (int16_t)(cy + v0y + ((dxy / 5) - 2));
            /* out-of-frame candidates get Diff |= 0x7fff and can never pass Diff < MinDiff, :546-553 */
            if (px < 0 || px > width - 8 || py < 0 || py > height - 8) continue;
            uint16_t Diff = 0; /* ushort accumulator, wraps mod 2^16, :472,:538 */
            for (int j = 0; j < 4; ++j) {
                const uint8_t *c4 = cur + (cy + dy1[j]) * width + (cx + dx1[j]);
                const uint8_t *r4 = ref + (py + dy1[j]) * width + (px + dx1[j]);
                Diff = (uint16_t)(Diff + weight4x4_u8(c4, width, r4, width));
            }
            /* :542-543 (sic: |total displacement| minus the signed parent vector) */
            const int pen = (iabs(iabs(px - cx) - v0x) + iabs(iabs(py - cy) - v0y)) * (pixel_rate < 4) * 64 / 2;
            Diff = (uint16_t)(Diff + pen);
            if (Diff < MinDiff) {
                bx = px;
                by = py;
                MinDiff = Diff;
            }
        }
        dst_net[2 * cell] = (int16_t)((int16_t)(bx - cx) * (int16_t)pixel_rate);
        dst_net[2 * cell + 1] = (int16_t)((int16_t)(by - cy) * (int16_t)pixel_rate);
    }
}

/* ---- six-tap helpers ------------------------------------------------------------------------
 * read_imageui with CLK_ADDRESS_CLAMP_TO_EDGE, src/GPU_kernels.cl:562 */
static inline int refpix(const uint8_t *ref, int w, int h, int x, int y) {
    return ref[clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)];
}
/* (sum p[i]*f[i] + 64)/128 with C truncating division, e.g. :593 */
static inline int tap6h(const uint8_t *ref, int w, int h, int x, int y, const int *f) {
    int s = 64;
    for (int t = 0; t < 6; ++t) s += refpix(ref, w, h, x - 2 + t, y) * f[t];
    return s / 128;
}

/* construct_opt1 (:776-943) == construct_opt2 (:945-1066) numerically: all nine horizontally
 * filtered lines are saturated to u8, then the vertical pass is saturated. */
static void interp4x4_sat(const uint8_t *ref, int w, int h, int ix, int iy, int fx, int fy, int out[16]) {
    int H[9][4];
    for (int L = 0; L < 9; ++L)
        for (int c = 0; c < 4; ++c) H[L][c] = sat8(tap6h(ref, w, h, ix + c, iy - 2 + L, sixtap[fx]));
    for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 4; ++c) {
            int s = 64;
            for (int t = 0; t < 6; ++t) s += H[i + t][c] * sixtap[fy][t];
            out[4 * i + c] = sat8(s / 128);
        }
}

/* copy_with_padding, src/encIO.h:141-196 (see vp8_oracle.h for the chroma lines) */
static void pad_plane(const uint8_t *s, int sw, int sh, uint8_t *d, int w, int h) {
    for (int y = 0; y < sh; ++y) {                       /* :152-185: copy a line, extend it to the right with its last sample */
        memcpy(d + (size_t)y * w, s + (size_t)y * sw, (size_t)sw);
        for (int x = sw; x < w; ++x) d[(size_t)y * w + x] = s[(size_t)y * sw + sw - 1];
    }
    for (int y = sh; y < h; ++y) memcpy(d + (size_t)y * w, d + (size_t)(sh - 1) * w, (size_t)w);   /* :186-200: last line downwards */
}
void vp8o_copy_with_padding(const uint8_t *sy, const uint8_t *su, const uint8_t *sv, int src_w, int src_h, uint8_t *dy, uint8_t *du,
                            uint8_t *dv, int w, int h) {
    pad_plane(sy, src_w, src_h, dy, w, h);
    pad_plane(su, src_w / 2, src_h / 2, du, w / 2, h / 2);
    pad_plane(sv, src_w / 2, src_h / 2, dv, w / 2, h / 2);
}

int vp8o_conformant = 0;   /* see vp8_oracle.h: NOT the reference, off by default */
void vp8o_set_conformant_stream(int on) { vp8o_conformant = on; }

/* construct (:574-774): lines 0..5 saturated (:600,616,632,648,664,680), lines 6..8 narrowed
 * with a plain (uchar) cast, i.e. wrapped mod 256 (:702-708, :727-733, :752-758).  A decoder saturates all nine (RFC 6386
 * section 18.3), so where one of the three lines below a 4x4 block overshoots, the encoder predicts from other samples than the
 * decoder will: with vp8o_conformant the format's rule is used. */
static void interp4x4_construct(const uint8_t *ref, int w, int h, int ix, int iy, int fx, int fy, int out[16]) {
    if (vp8o_conformant) {
        interp4x4_sat(ref, w, h, ix, iy, fx, fy, out);
        return;
    }
    int H[9][4];
    for (int L = 0; L < 9; ++L)
        for (int c = 0; c < 4; ++c) {
            const int v = tap6h(ref, w, h, ix + c, iy - 2 + L, sixtap[fx]);
            H[L][c] = (L < 6) ? sat8(v) : (int)(uint8_t)v;
        }
    for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 4; ++c) {
            int s = 64;
            for (int t = 0; t < 6; ++t) s += H[i + t][c] * sixtap[fy][t];
            out[4 * i + c] = sat8(s / 128);
        }
}

/* ---- luma_search_2step, src/GPU_kernels.cl:1068-1203 -------------------------------------- */
static const int dx4[4] = {0, 0, 16, 16}, dy4[4] = {0, 16, 0, 16}; /* :454-455 */

void vp8o_luma_search_2step(const uint8_t *cur, const uint8_t *ref, const int16_t *net, int16_t *ref_net,
                            int32_t *ref_Bdiff, int width, int height) {
    const int nblk = width * height / 64;
    const int bw = width / 8;
#pragma omp parallel for schedule(dynamic, 64)
    for (int b = 0; b < nblk; ++b) {
        const int16_t v0x = (int16_t)(net[2 * b] * 4), v0y = (int16_t)(net[2 * b + 1] * 4); /* qpel */
        const int pcx = (b % bw) * 8, pcy = (b / bw) * 8;
        const int16_t cx4 = (int16_t)(pcx * 4), cy4 = (int16_t)(pcy * 4);
        int MinDiff = 0x7fff;
        int16_t bestx = (int16_t)(width * 4 - 32), besty = (int16_t)(height * 4 - 32); /* :1136-1137 */
        for (int k = 0; k < 26; ++k) {
            int16_t qx = (int16_t)(cx4 + v0x + ((k % 5) - 2));
            int16_t qy = (int16_t)(cy4 + v0y + ((k / 5) - 2));
            if (k == 25) { /* explicit zero-MV candidate, :1149-1150 */
                qx = cx4;
                qy = cy4;
            }
            if (qx < 0 || qx > width * 4 - 32 || qy < 0 || qy > height * 4 - 32) continue; /* :1180-1183 */
            const int fx = (qx % 4) * 2, fy = (qy % 4) * 2;
            int Diff = 0;
            for (int j = 0; j < 4; ++j) {
                const int ix = (qx + dx4[j]) / 4, iy = (qy + dy4[j]) / 4;
                int p[16], d[16];
                interp4x4_sat(ref, width, height, ix, iy, fx, fy, p);
                const uint8_t *c4 = cur + (pcy + dy4[j] / 4) * width + pcx + dx4[j] / 4;
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 4; ++c) d[4 * r + c] = (int)c4[r * width + c] - p[4 * r + c];
                Diff += vp8o_weight(d);
            }
            if (k != 25) Diff += (iabs(qx - cx4 - v0x) + iabs(qy - cy4 - v0y)) * 64 / 2; /* :1176-1178 */
            if (Diff < MinDiff) {
                bestx = qx;
                besty = qy;
                MinDiff = Diff;
            }
        }
        const int16_t vx = (int16_t)(bestx - cx4), vy = (int16_t)(besty - cy4);
        /* penalty taken back only for a non-zero vector, :1193-1197 */
        if ((vx != 0) | (vy != 0)) MinDiff -= (iabs(vx - v0x) + iabs(vy - v0y)) * 64 / 2;
        ref_net[2 * b] = vx;
        ref_net[2 * b + 1] = vy;
        ref_Bdiff[b] = MinDiff;
    }
}

/* ---- select_reference, src/GPU_kernels.cl:1205-1283 ---------------------------------------- */
void vp8o_select_reference(const int16_t *last_net, const int16_t *golden_net, const int16_t *altref_net,
                           const int32_t *last_Bdiff, const int32_t *golden_Bdiff, const int32_t *altref_Bdiff,
                           int32_t *MB_ref, int16_t *MB_vectors, int width, int height, int use_golden,
                           int use_altref) {
    const int mb_width = width / 16, mb_count = mb_width * (height / 16);
    const int b8w = mb_width * 2;
    for (int mb = 0; mb < mb_count; ++mb) {
        const int b = ((mb / mb_width) * 2) * b8w + (mb % mb_width) * 2;
        const int idx[4] = {b, b + 1, b + b8w, b + b8w + 1};
        int diff1 = last_Bdiff[idx[0]] + last_Bdiff[idx[1]] + last_Bdiff[idx[2]] + last_Bdiff[idx[3]];
        int diff2 = 0x7fffffff;
        if (use_altref == 1)
            diff2 = altref_Bdiff[idx[0]] + altref_Bdiff[idx[1]] + altref_Bdiff[idx[2]] + altref_Bdiff[idx[3]];
        int ref = (diff1 <= diff2) ? VP8O_LAST : VP8O_ALTREF;
        diff1 = (diff1 <= diff2) ? diff1 : diff2;
        diff2 = 0x7fffffff;
        if (use_golden == 1)
            diff2 = golden_Bdiff[idx[0]] + golden_Bdiff[idx[1]] + golden_Bdiff[idx[2]] + golden_Bdiff[idx[3]];
        ref = (diff1 <= diff2) ? ref : VP8O_GOLDEN;
        const int16_t *net = (ref == VP8O_LAST) ? last_net : (ref == VP8O_GOLDEN ? golden_net : altref_net);
        MB_ref[mb] = ref;
        for (int k = 0; k < 4; ++k) {
            MB_vectors[8 * mb + 2 * k] = net[2 * idx[k]];
            MB_vectors[8 * mb + 2 * k + 1] = net[2 * idx[k] + 1];
        }
    }
}

/* ---- pack_8x8_into_16x16, src/GPU_kernels.cl:1346-1366 ------------------------------------- */
void vp8o_pack_8x8_into_16x16(const int16_t *MB_vectors, int32_t *MB_parts, float *MB_SSIM, int mb_count) {
    for (int mb = 0; mb < mb_count; ++mb) {
        const int16_t *v = MB_vectors + 8 * mb;
        MB_SSIM[mb] = -2.0f;
        int same = 1;
        for (int k = 1; k < 4; ++k) same &= (v[2 * k] == v[0]) && (v[2 * k + 1] == v[1]);
        MB_parts[mb] = same ? VP8O_16x16 : VP8O_8x8;
    }
}

/* ---- prepare_predictors_and_residual, src/GPU_kernels.cl:1285-1344 ------------------------- */
void vp8o_prepare_predictors_and_residual(const uint8_t *cur, const uint8_t *ref, uint8_t *predictor,
                                          int16_t *residual, const int32_t *MB_ref, const int16_t *MB_vectors,
                                          int width, int height, int plane, int ref_id) {
    const int mb_size = (plane == 0) ? 16 : 8;
    const int nblk = (width / 4) * (height / 4);
    const int g = (plane == 0) ? 4 : 8;
#pragma omp parallel for schedule(dynamic, 256)
    for (int b = 0; b < nblk; ++b) {
        const int posx = (b % (width / 4)) * 4, posy = (b / (width / 4)) * 4;
        const int mb = (posy / mb_size) * (width / mb_size) + posx / mb_size;
        if (MB_ref[mb] != ref_id) continue;
        const int qx = (posx % mb_size) / (mb_size / 2), qy = (posy % mb_size) / (mb_size / 2);
        const int vi = qy * 2 + qx;
        const int vx = MB_vectors[8 * mb + 2 * vi], vy = MB_vectors[8 * mb + 2 * vi + 1];
        int dx = (posx * g + vx) % g, dy = (posy * g + vy) % g;
        dx *= (plane == 0) ? 2 : 1;
        dy *= (plane == 0) ? 2 : 1;
        /* selected vectors are always in-frame, so dx,dy >= 0; guard the table index anyway */
        dx = ((dx % 8) + 8) % 8;
        dy = ((dy % 8) + 8) % 8;
        const int ix = (posx * g + vx) / g, iy = (posy * g + vy) / g;
        int p[16];
        interp4x4_construct(ref, width, height, ix, iy, dx, dy, p);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                const int i = (posy + r) * width + posx + c;
                predictor[i] = (uint8_t)p[4 * r + c];
                residual[i] = (int16_t)((int)cur[i] - p[4 * r + c]);
            }
    }
}

/* quantizer set shared by dct4x4 (:1394-1408) and idct4x4 (:1568-1582) */
static void block_quantizers(const int32_t *SD, int segment_id, int is16x16, int plane, int *dc_q, int *ac_q) {
    const int i = SD[segment_id * SD_INTS + SD_Y_AC_I];
    if (plane == 0) {
        *ac_q = ac_qlookup[i];
        *dc_q = is16x16 ? 1 : dc_qlookup[clamp_qi(SD[SD_Y_DC_IDELTA] + i)];
    } else {
        int uv_dc = dc_qlookup[clamp_qi(SD[SD_UV_DC_IDELTA] + i)];
        if (uv_dc > 132) uv_dc = 132;
        *dc_q = uv_dc;
        *ac_q = ac_qlookup[clamp_qi(SD[SD_UV_AC_IDELTA] + i)];
    }
}

/* ---- dct4x4, src/GPU_kernels.cl:1368-1496 -------------------------------------------------- */
static void fdct4x4(const int in[16], int out[16]) {
    int L[16];
    /* vertical pass first (:1417-1429), libvpx vp8_short_fdct4x4 constants */
    for (int c = 0; c < 4; ++c) {
        const int r0 = in[c], r1 = in[4 + c], r2 = in[8 + c], r3 = in[12 + c];
        const int a1 = (r0 + r3) * 8, d1 = (r0 - r3) * 8, b1 = (r1 + r2) * 8, c1 = (r1 - r2) * 8;
        L[c] = a1 + b1;
        L[8 + c] = a1 - b1;
        L[4 + c] = (c1 * 2217 + d1 * 5352 + 14500) >> 12;
        L[12 + c] = (d1 * 2217 - c1 * 5352 + 7500) >> 12;
    }
    for (int i = 0; i < 4; ++i) { /* :1431-1476 */
        const int e0 = L[4 * i], e1 = L[4 * i + 1], e2 = L[4 * i + 2], e3 = L[4 * i + 3];
        const int a1 = e0 + e3, d1 = e0 - e3, b1 = e1 + e2, c1 = e1 - e2;
        out[4 * i] = (a1 + b1 + 7) >> 4;
        out[4 * i + 2] = (a1 - b1 + 7) >> 4;
        out[4 * i + 1] = ((c1 * 2217 + d1 * 5352 + 12000) >> 16) + (d1 != 0);
        out[4 * i + 3] = (d1 * 2217 - c1 * 5352 + 51000) >> 16;
    }
}

void vp8o_dct4x4(const int16_t *residual, int16_t *MB, int32_t *MB_segment_id, const int32_t *MB_parts,
                 const float *MB_SSIM, int width, int height, const int32_t *SD, int segment_id,
                 float SSIM_target, int plane) {
    const int mb_size = (plane == 0) ? 16 : 8;
    const int nblk = (width / 4) * (height / 4);
#pragma omp parallel for schedule(static)
    for (int b = 0; b < nblk; ++b) {
        const int posx = (b % (width / 4)) * 4, posy = (b / (width / 4)) * 4;
        const int mb = (posy / mb_size) * (width / mb_size) + posx / mb_size;
        if (MB_SSIM[mb] > SSIM_target) continue; /* :1391 */
        MB_segment_id[mb] = segment_id;          /* same value from every block/plane, :1393 */
        int dc_q, ac_q;
        block_quantizers(SD, segment_id, MB_parts[mb] == VP8O_16x16, plane, &dc_q, &ac_q);
        int in[16], L[16];
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) in[4 * r + c] = residual[(posy + r) * width + posx + c];
        fdct4x4(in, L);
        L[0] /= dc_q; /* truncating division, :1478-1481 */
        for (int k = 1; k < 16; ++k) L[k] /= ac_q;
        int bn = ((posy % mb_size) / 4) * (mb_size / 4) + (posx % mb_size) / 4;
        bn += (plane == 1) ? 16 : 0;
        bn += (plane == 2) ? 20 : 0;
        int16_t *dst = MB + ((size_t)mb * 25 + bn) * 16;
        for (int k = 0; k < 16; ++k) dst[inv_zigzag[k]] = (int16_t)L[k];
    }
}

/* ---- wht4x4_iwht4x4, src/GPU_kernels.cl:1498-1543 (+ :257-401) ----------------------------- */
void vp8o_wht4x4_iwht4x4(int16_t *MB, const int32_t *MB_segment_id, const int32_t *MB_parts, const int32_t *SD,
                         int segment_id, int mb_count) {
#pragma omp parallel for schedule(static)
    for (int mb = 0; mb < mb_count; ++mb) {
        if (MB_segment_id[mb] != segment_id) continue;
        if (MB_parts[mb] != VP8O_16x16) continue;
        const int i = SD[segment_id * SD_INTS + SD_Y_AC_I];
        const int y2_dc_q = dc_qlookup[clamp_qi(SD[SD_Y2_DC_IDELTA] + i)] * 2;
        int y2_ac_q = 31 * ac_qlookup[clamp_qi(SD[SD_Y2_AC_IDELTA] + i)] / 20;
        if (y2_ac_q < 8) y2_ac_q = 8;
        int16_t *m = MB + (size_t)mb * 400;
        int X[16], T[16];
        for (int k = 0; k < 16; ++k) X[k] = m[k * 16]; /* the 16 luma DCs, raster */
        /* WHT_and_quant :257-339: vertical butterflies, then per-row butterflies */
        for (int c = 0; c < 4; ++c) {
            const int a = X[c] + X[12 + c], b = X[4 + c] + X[8 + c], cc = X[4 + c] - X[8 + c], d = X[c] - X[12 + c];
            T[c] = a + b;
            T[4 + c] = cc + d;
            T[8 + c] = a - b;
            T[12 + c] = d - cc;
        }
        for (int r = 0; r < 4; ++r) {
            const int x = T[4 * r], y = T[4 * r + 1], z = T[4 * r + 2], w = T[4 * r + 3];
            const int a1 = x + w, b1 = y + z, c1 = y - z, d1 = x - w;
            int o[4] = {a1 + b1, c1 + d1, a1 - b1, d1 - c1};
            for (int k = 0; k < 4; ++k) {
                o[k] += (o[k] > 0);
                o[k] >>= 1;
                X[4 * r + k] = o[k] / ((r == 0 && k == 0) ? y2_dc_q : y2_ac_q);
            }
        }
        for (int k = 0; k < 16; ++k) m[24 * 16 + inv_zigzag[k]] = (int16_t)X[k];
        /* dequant_and_iWHT :341-401: per-row pass first, then vertical, (x+3)>>3 */
        for (int k = 0; k < 16; ++k) X[k] *= (k == 0) ? y2_dc_q : y2_ac_q;
        for (int r = 0; r < 4; ++r) {
            const int x = X[4 * r], y = X[4 * r + 1], z = X[4 * r + 2], w = X[4 * r + 3];
            const int a1 = x + w, b1 = y + z, c1 = y - z, d1 = x - w;
            T[4 * r] = a1 + b1;
            T[4 * r + 1] = c1 + d1;
            T[4 * r + 2] = a1 - b1;
            T[4 * r + 3] = d1 - c1;
        }
        for (int c = 0; c < 4; ++c) {
            const int a = T[c] + T[12 + c], b = T[4 + c] + T[8 + c], cc = T[4 + c] - T[8 + c], d = T[c] - T[12 + c];
            X[c] = (a + b + 3) >> 3;
            X[4 + c] = (cc + d + 3) >> 3;
            X[8 + c] = (a - b + 3) >> 3;
            X[12 + c] = (d - cc + 3) >> 3;
        }
        for (int k = 0; k < 16; ++k) m[k * 16] = (int16_t)X[k];
    }
}

/* ---- idct4x4, src/GPU_kernels.cl:1545-1608 (+dequant_and_iDCT :192-255) -------------------- */
static void dequant_idct4x4(int L[16], int dc_q, int ac_q) {
    int T[16];
    L[0] *= dc_q;
    for (int k = 1; k < 16; ++k) L[k] *= ac_q;
    for (int c = 0; c < 4; ++c) { /* vertical */
        const int i0 = L[c], i1 = L[4 + c], i2 = L[8 + c], i3 = L[12 + c];
        const int a1 = i0 + i2, b1 = i0 - i2;
        const int c1 = ((i1 * 35468) >> 16) - (i3 + ((i3 * 20091) >> 16));
        const int d1 = (i1 + ((i1 * 20091) >> 16)) + ((i3 * 35468) >> 16);
        T[c] = a1 + d1;
        T[12 + c] = a1 - d1;
        T[4 + c] = b1 + c1;
        T[8 + c] = b1 - c1;
    }
    for (int r = 0; r < 4; ++r) { /* horizontal */
        const int i0 = T[4 * r], i1 = T[4 * r + 1], i2 = T[4 * r + 2], i3 = T[4 * r + 3];
        const int a1 = i0 + i2, b1 = i0 - i2;
        const int c1 = ((i1 * 35468) >> 16) - (i3 + ((i3 * 20091) >> 16));
        const int d1 = (i1 + ((i1 * 20091) >> 16)) + ((i3 * 35468) >> 16);
        L[4 * r] = (a1 + d1 + 4) >> 3;
        L[4 * r + 3] = (a1 - d1 + 4) >> 3;
        L[4 * r + 1] = (b1 + c1 + 4) >> 3;
        L[4 * r + 2] = (b1 - c1 + 4) >> 3;
    }
}

void vp8o_idct4x4(uint8_t *recon, const uint8_t *predictor, const int16_t *MB, const int32_t *MB_segment_id,
                  const int32_t *MB_parts, int width, int height, const int32_t *SD, int segment_id, int plane) {
    const int mb_size = (plane == 0) ? 16 : 8;
    const int nblk = (width / 4) * (height / 4);
#pragma omp parallel for schedule(static)
    for (int b = 0; b < nblk; ++b) {
        const int x = (b % (width / 4)) * 4, y = (b / (width / 4)) * 4;
        const int mb = (y / mb_size) * (width / mb_size) + x / mb_size;
        if (MB_segment_id[mb] != segment_id) continue;
        int dc_q, ac_q;
        block_quantizers(SD, segment_id, MB_parts[mb] == VP8O_16x16, plane, &dc_q, &ac_q);
        int bn = ((y % mb_size) / 4) * (mb_size / 4) + (x % mb_size) / 4;
        bn += (plane == 1) ? 16 : 0;
        bn += (plane == 2) ? 20 : 0;
        const int16_t *src = MB + ((size_t)mb * 25 + bn) * 16;
        int L[16];
        for (int k = 0; k < 16; ++k) L[k] = src[inv_zigzag[k]];
        dequant_idct4x4(L, dc_q, ac_q);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                const int i = (y + r) * width + x + c;
                recon[i] = (uint8_t)sat8(L[4 * r + c] + predictor[i]);
            }
    }
}

/* ---- count_SSIM_luma / _chroma, src/GPU_kernels.cl:1610-1971, :1973-2095 -------------------
 * float4 lane accumulation in raster order then lane sum; mad() modelled as unfused a*b+c
 * (build with -ffp-contract=off). n = mb_size (16 or 8). */
void vp8o_count_SSIM(const uint8_t *f1, const uint8_t *f2, const int32_t *MB_segment_id, float *metric, int width,
                     int height, int segment_id, int n) {
    const int mbw = width / n, mb_count = mbw * (height / n);
    const float c1 = 0.01f * 0.01f * 255 * 255;
    const float c2 = 0.03f * 0.03f * 255 * 255;
    const float area = (float)(n * n);
#pragma omp parallel for schedule(static)
    for (int mb = 0; mb < mb_count; ++mb) {
        if (MB_segment_id[mb] != segment_id) continue;
        const uint8_t *a = f1 + (size_t)(mb / mbw) * n * width + (mb % mbw) * n;
        const uint8_t *b = f2 + (size_t)(mb / mbw) * n * width + (mb % mbw) * n;
        float IL[4];
        float M1, M2, D, C;
#define LANESUM() (((IL[0] + IL[1]) + IL[2]) + IL[3])
        for (int l = 0; l < 4; ++l) IL[l] = 0.0f;
        for (int r = 0; r < n; ++r)
            for (int j = 0; j < n / 4; ++j)
                for (int l = 0; l < 4; ++l) IL[l] += (float)a[r * width + 4 * j + l];
        M1 = LANESUM() / area;
        for (int l = 0; l < 4; ++l) IL[l] = 0.0f;
        for (int r = 0; r < n; ++r)
            for (int j = 0; j < n / 4; ++j)
                for (int l = 0; l < 4; ++l) {
                    const float t = (float)a[r * width + 4 * j + l] - M1;
                    IL[l] = (r == 0 && j == 0) ? t * t : t * t + IL[l];
                }
        D = LANESUM() / area;
        for (int l = 0; l < 4; ++l) IL[l] = 0.0f;
        for (int r = 0; r < n; ++r)
            for (int j = 0; j < n / 4; ++j)
                for (int l = 0; l < 4; ++l) IL[l] += (float)b[r * width + 4 * j + l];
        M2 = LANESUM() / area;
        for (int l = 0; l < 4; ++l) IL[l] = 0.0f;
        for (int r = 0; r < n; ++r)
            for (int j = 0; j < n / 4; ++j)
                for (int l = 0; l < 4; ++l) {
                    const float t = (float)b[r * width + 4 * j + l] - M2;
                    IL[l] = (r == 0 && j == 0) ? t * t : t * t + IL[l];
                }
        D += LANESUM() / area;
        for (int r = 0; r < n; ++r)
            for (int j = 0; j < n / 4; ++j)
                for (int l = 0; l < 4; ++l) {
                    const float t = ((float)a[r * width + 4 * j + l] - M1) * ((float)b[r * width + 4 * j + l] - M2);
                    IL[l] = (r == 0 && j == 0) ? t : IL[l] + t;
                }
        C = LANESUM() / area;
#undef LANESUM
        C = (M1 * (M2 * 2) + c1) * (C * 2 + c2) / ((M1 * M1 + (M2 * M2 + c1)) * (D + c2));
        D = M1 - M2;
        D = (D < 0) ? -D : D;
        D = (D > 4) ? 0.02f * D : 0.0f; /* :1959-1966 */
        C -= D;
        metric[mb] = C;
    }
}

void vp8o_gather_SSIM(const float *m1, const float *m2, const float *m3, float *MB_SSIM, int mb_count) {
    for (int mb = 0; mb < mb_count; ++mb) MB_SSIM[mb] = (m1[mb] + m2[mb] + m3[mb]) / 3;
}

/* ---- prepare_filter_mask, src/CPU_kernels.cl:782-827 --------------------------------------- */
void vp8o_prepare_filter_mask(const int16_t *MB, int32_t *MB_non_zero_coeffs, const int32_t *MB_parts,
                              int32_t *mb_mask, int width, int height) {
    const int mb_count = (width / 16) * (height / 16);
#pragma omp parallel for schedule(static)
    for (int mb = 0; mb < mb_count; ++mb) {
        const int16_t *m = MB + (size_t)mb * 400;
        const int split = MB_parts[mb];
        int coeffs = 0;
        for (int b = 0; b < 16; ++b)
            for (int i = 1; i < 16; ++i) coeffs += iabs(m[b * 16 + i]);
        for (int b = 16; b < 24; ++b)
            for (int i = 0; i < 16; ++i) coeffs += iabs(m[b * 16 + i]);
        if (split == VP8O_16x16) {
            for (int i = 0; i < 16; ++i) coeffs += iabs(m[24 * 16 + i]);
        } else {
            for (int b = 0; b < 16; ++b) coeffs += iabs(m[b * 16]);
        }
        MB_non_zero_coeffs[mb] = coeffs;
        mb_mask[mb] = ((split != VP8O_16x16) || (coeffs > 0)) ? -1 : 0;
    }
}

/* ---- loop filter, src/CPU_kernels.cl:829-926 (edge filters), :970-1075, :1333-1439 ---------
 * Works on u = pixel-128 held in "registers" (int here, short8 there).  The reference keeps
 * the q registers of one edge as the p registers of the next edge WITHOUT the saturation the
 * store applies, so values outside [-128,127] survive inside a macroblock row/column. */
static inline int c128(int v) { return clampi(v, -128, 127); }

static void filter_mb_edge(const int *p3, int *p2, int *p1, int *p0, int *q0, int *q1, int *q2, const int *q3,
                           int mb_lim, int int_lim, int hev_thr) {
    for (int k = 0; k < 8; ++k) {
        int mask = (iabs(p3[k] - p2[k]) > int_lim) | (iabs(p2[k] - p1[k]) > int_lim) | (iabs(p1[k] - p0[k]) > int_lim) |
                   (iabs(q1[k] - q0[k]) > int_lim) | (iabs(q2[k] - q1[k]) > int_lim) | (iabs(q3[k] - q2[k]) > int_lim) |
                   ((iabs(p0[k] - q0[k]) * 2 + iabs(p1[k] - q1[k]) / 2) > mb_lim);
        mask = !mask;
        const int hev = (iabs(p1[k] - p0[k]) > hev_thr) | (iabs(q1[k] - q0[k]) > hev_thr);
        int w = c128(p1[k] - q1[k]);
        w = c128(w + (q0[k] - p0[k]) * 3);
        w = mask ? w : 0;
        int a = hev ? w : 0;
        const int b = c128(a + 3) >> 3;
        a = c128(a + 4) >> 3;
        q0[k] -= a;
        p0[k] += b;
        w = hev ? 0 : w;
        a = c128((w * 27 + 63) >> 7);
        q0[k] -= a;
        p0[k] += a;
        a = c128((w * 18 + 63) >> 7);
        q1[k] -= a;
        p1[k] += a;
        a = c128((w * 9 + 63) >> 7);
        q2[k] -= a;
        p2[k] += a;
    }
}

static void filter_b_edge(const int *p3, const int *p2, int *p1, int *p0, int *q0, int *q1, const int *q2,
                          const int *q3, int b_lim, int int_lim, int hev_thr) {
    for (int k = 0; k < 8; ++k) {
        int mask = (iabs(p3[k] - p2[k]) > int_lim) | (iabs(p2[k] - p1[k]) > int_lim) | (iabs(p1[k] - p0[k]) > int_lim) |
                   (iabs(q1[k] - q0[k]) > int_lim) | (iabs(q2[k] - q1[k]) > int_lim) | (iabs(q3[k] - q2[k]) > int_lim) |
                   ((iabs(p0[k] - q0[k]) * 2 + iabs(p1[k] - q1[k]) / 2) > b_lim);
        mask = !mask;
        const int hev = (iabs(p1[k] - p0[k]) > hev_thr) | (iabs(q1[k] - q0[k]) > hev_thr);
        int a = c128(p1[k] - q1[k]);
        a = hev ? a : 0;
        a = c128(a + (q0[k] - p0[k]) * 3);
        a = mask ? a : 0;
        const int b = c128(a + 3) >> 3;
        a = c128(a + 4) >> 3;
        q0[k] -= a;
        p0[k] += b;
        a = (a + 1) >> 1;
        a = hev ? 0 : a;
        q1[k] -= a;
        p1[k] += a;
    }
}

/* 8 samples spaced `step` apart starting at frame[pos]: read8p/write8p (:928-956) and the
 * vload8/vstore8 forms (:1041-1057) are the same thing with step = width or 1 */
static inline void rd8(const uint8_t *frame, int pos, int step, int *v) {
    for (int k = 0; k < 8; ++k) v[k] = (int)frame[pos + k * step] - 128;
}
static inline void wr8(uint8_t *frame, int pos, int step, const int *v) {
    for (int k = 0; k < 8; ++k) frame[pos + k * step] = (uint8_t)sat8(v[k] + 128);
}

void vp8o_loop_filter_frame(uint8_t *frame, const int32_t *MB_segment_ids, const int32_t *mb_mask,
                            const int32_t *SD, int width, int height, int mb_size) {
    const int mb_width = width / mb_size, mb_count = mb_width * (height / mb_size);
    int p3[8], p2[8], p1[8], p0[8], q0[8], q1[8], q2[8], q3[8];
    for (int mb = 0; mb < mb_count; ++mb) {
        const int32_t *sd = SD + MB_segment_ids[mb] * SD_INTS;
        if (sd[SD_LOOP_FILTER_LEVEL] == 0) return; /* sic: leaves the whole plane, :990 */
        const int int_lim = (int16_t)sd[SD_INTERIOR_LIMIT], mb_lim = (int16_t)sd[SD_MBEDGE_LIMIT];
        const int b_lim = (int16_t)sd[SD_SUB_BEDGE_LIMIT], hev_thr = (int16_t)sd[SD_HEV_THRESHOLD];
        const int x0 = (mb % mb_width) * mb_size, y0 = (mb / mb_width) * mb_size;
        /* dir 0: vertical edges (filter along x, 8 rows at a time); dir 1: horizontal edges */
        for (int dir = 0; dir < 2; ++dir) {
            const int along = dir == 0 ? 1 : width;  /* step across the edge */
            const int lane = dir == 0 ? width : 1;   /* step between the 8 parallel samples */
            const int e0 = dir == 0 ? x0 : y0;       /* coordinate of the MB edge */
            for (int g = 0; g < mb_size; g += 8) {
                const int base = dir == 0 ? (y0 + g) * width + x0 : y0 * width + x0 + g;
                rd8(frame, base, lane, q0);
                rd8(frame, base + along, lane, q1);
                rd8(frame, base + 2 * along, lane, q2);
                rd8(frame, base + 3 * along, lane, q3);
                if (e0 > 0) {
                    rd8(frame, base - 4 * along, lane, p3);
                    rd8(frame, base - 3 * along, lane, p2);
                    rd8(frame, base - 2 * along, lane, p1);
                    rd8(frame, base - 1 * along, lane, p0);
                    filter_mb_edge(p3, p2, p1, p0, q0, q1, q2, q3, mb_lim, int_lim, hev_thr);
                    wr8(frame, base - 3 * along, lane, p2);
                    wr8(frame, base - 2 * along, lane, p1);
                    wr8(frame, base - 1 * along, lane, p0);
                    wr8(frame, base, lane, q0);
                    wr8(frame, base + along, lane, q1);
                    wr8(frame, base + 2 * along, lane, q2);
                }
                for (int e = 4; e < mb_size && mb_mask[mb]; e += 4) {
                    memcpy(p3, q0, sizeof p3); /* registers carried, not re-read, :1024 */
                    memcpy(p2, q1, sizeof p2);
                    memcpy(p1, q2, sizeof p1);
                    memcpy(p0, q3, sizeof p0);
                    const int eb = base + e * along;
                    rd8(frame, eb, lane, q0);
                    rd8(frame, eb + along, lane, q1);
                    rd8(frame, eb + 2 * along, lane, q2);
                    rd8(frame, eb + 3 * along, lane, q3);
                    filter_b_edge(p3, p2, p1, p0, q0, q1, q2, q3, b_lim, int_lim, hev_thr);
                    wr8(frame, eb - 2 * along, lane, p1);
                    wr8(frame, eb - 1 * along, lane, p0);
                    wr8(frame, eb, lane, q0);
                    wr8(frame, eb + along, lane, q1);
                }
            }
        }
    }
}

/* ==============================================================================================
 * Whole-frame driver: src/inter_part.h:1-384 + src/loop_filter.h
 * ============================================================================================== */
typedef struct {
    uint8_t *Y[5]; /* Y[0] full res, Y[l] downsampled by 2^l */
    uint8_t *U, *V;
} o_frame;

struct vp8o_ctx {
    int W, H, mbs, b8;
    float ssim_target;
    int32_t SD[44];
    o_frame ref[3]; /* LAST, GOLDEN, ALTREF: separate copies as in the reference */
    o_frame cur;
    int16_t *net[3][2]; /* vnet1, vnet2 per reference */
    int32_t *bdiff[3];
    int32_t *MB_parts, *MB_ref, *MB_seg, *MB_nz, *mb_mask;
    int16_t *MB_vec, *MB_coeffs;
    float *MB_SSIM, *metric[3];
    uint8_t *pred[3], *recon[3];
    int16_t *resid[3];
};

static void frame_alloc(o_frame *f, int W, int H) {
    for (int l = 0; l < 5; ++l) f->Y[l] = (uint8_t *)calloc((size_t)(W >> l) * (H >> l) + 64, 1);
    f->U = (uint8_t *)calloc((size_t)W * H / 4, 1);
    f->V = (uint8_t *)calloc((size_t)W * H / 4, 1);
}
static void frame_free(o_frame *f) {
    for (int l = 0; l < 5; ++l) free(f->Y[l]);
    free(f->U);
    free(f->V);
}
static void frame_copy(o_frame *d, const o_frame *s, int W, int H) {
    for (int l = 0; l < 5; ++l) memcpy(d->Y[l], s->Y[l], (size_t)(W >> l) * (H >> l));
    memcpy(d->U, s->U, (size_t)W * H / 4);
    memcpy(d->V, s->V, (size_t)W * H / 4);
}
static void frame_pyramid(o_frame *f, int W, int H) {
    for (int l = 1; l < 5; ++l) vp8o_downsample_x2(f->Y[l - 1], f->Y[l], W >> (l - 1), H >> (l - 1));
}

vp8o_ctx *vp8o_create(int width, int height, float ssim_target) {
    if (width % 16 || height % 16 || width < 16 || height < 16) return NULL;
    vp8o_ctx *c = (vp8o_ctx *)calloc(1, sizeof *c);
    c->W = width;
    c->H = height;
    c->mbs = (width / 16) * (height / 16);
    c->b8 = c->mbs * 4;
    c->ssim_target = ssim_target;
    for (int r = 0; r < 3; ++r) {
        frame_alloc(&c->ref[r], width, height);
        c->net[r][0] = (int16_t *)calloc((size_t)c->b8 * 2, sizeof(int16_t));
        c->net[r][1] = (int16_t *)calloc((size_t)c->b8 * 2, sizeof(int16_t));
        c->bdiff[r] = (int32_t *)calloc(c->b8, sizeof(int32_t));
        c->metric[r] = (float *)calloc(c->mbs, sizeof(float));
    }
    frame_alloc(&c->cur, width, height);
    c->MB_parts = (int32_t *)calloc(c->mbs, 4);
    c->MB_ref = (int32_t *)calloc(c->mbs, 4);
    c->MB_seg = (int32_t *)calloc(c->mbs, 4);
    c->MB_nz = (int32_t *)calloc(c->mbs, 4);
    c->mb_mask = (int32_t *)calloc(c->mbs, 4);
    c->MB_vec = (int16_t *)calloc((size_t)c->mbs * 8, 2);
    c->MB_coeffs = (int16_t *)calloc((size_t)c->mbs * 400, 2);
    c->MB_SSIM = (float *)calloc(c->mbs, 4);
    for (int p = 0; p < 3; ++p) {
        const size_t n = p == 0 ? (size_t)width * height : (size_t)width * height / 4;
        c->pred[p] = (uint8_t *)calloc(n, 1);
        c->recon[p] = (uint8_t *)calloc(n, 1);
        c->resid[p] = (int16_t *)calloc(n, 2);
    }
    return c;
}

void vp8o_destroy(vp8o_ctx *c) {
    if (!c) return;
    for (int r = 0; r < 3; ++r) {
        frame_free(&c->ref[r]);
        free(c->net[r][0]);
        free(c->net[r][1]);
        free(c->bdiff[r]);
        free(c->metric[r]);
    }
    frame_free(&c->cur);
    free(c->MB_parts);
    free(c->MB_ref);
    free(c->MB_seg);
    free(c->MB_nz);
    free(c->mb_mask);
    free(c->MB_vec);
    free(c->MB_coeffs);
    free(c->MB_SSIM);
    for (int p = 0; p < 3; ++p) {
        free(c->pred[p]);
        free(c->recon[p]);
        free(c->resid[p]);
    }
    free(c);
}

void vp8o_upload_last(vp8o_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v) {
    memcpy(c->ref[0].Y[0], y, (size_t)c->W * c->H);
    memcpy(c->ref[0].U, u, (size_t)c->W * c->H / 4);
    memcpy(c->ref[0].V, v, (size_t)c->W * c->H / 4);
}

void vp8o_set_segments(vp8o_ctx *c, const int32_t sd[44]) { memcpy(c->SD, sd, sizeof c->SD); }

void vp8o_inter_transform(vp8o_ctx *c, const uint8_t *cur_y, const uint8_t *cur_u, const uint8_t *cur_v,
                          int prev_is_golden, int prev_is_altref, int use_golden, int use_altref, vp8o_results *out) {
    const int W = c->W, H = c->H, CW = W / 2, CH = H / 2;
    const int net_width = (W / 16) * 2;
    const int use[3] = {1, use_golden, use_altref};
    memcpy(c->cur.Y[0], cur_y, (size_t)W * H);
    memcpy(c->cur.U, cur_u, (size_t)CW * CH);
    memcpy(c->cur.V, cur_v, (size_t)CW * CH);

    /* prepare_GPU_buffers, src/inter_part.h:1-94 */
    for (int r = 0; r < 3; ++r) { /* reset_vectors :5-7 */
        memset(c->net[r][0], 0, (size_t)c->b8 * 4);
        memset(c->net[r][1], 0, (size_t)c->b8 * 4);
        for (int i = 0; i < c->b8; ++i) c->bdiff[r][i] = 0x7fffffff;
    }
    frame_pyramid(&c->ref[0], W, H); /* :11-33 */
    frame_pyramid(&c->cur, W, H);
    if (prev_is_golden) frame_copy(&c->ref[1], &c->ref[0], W, H); /* :35-42, :72-77 */
    if (prev_is_altref) frame_copy(&c->ref[2], &c->ref[0], W, H); /* :43-50, :78-83 */

    /* hierarchical search, src/inter_part.h:110-236; ping-pong per src/init.h:672-854 */
    for (int r = 0; r < 3; ++r) {
        if (!use[r]) continue;
        int src = 0;
        for (int l = 4; l >= 0; --l) {
            vp8o_luma_search_1step(c->cur.Y[l], c->ref[r].Y[l], c->net[r][src], c->net[r][src ^ 1], net_width, W >> l,
                                   H >> l, 1 << l);
            src ^= 1;
        }
        /* after 5 levels the result sits in vnet2 (index 1); 2-step reads vnet2, writes vnet1 */
        vp8o_luma_search_2step(c->cur.Y[0], c->ref[r].Y[0], c->net[r][1], c->net[r][0], c->bdiff[r], W, H);
    }
    vp8o_select_reference(c->net[0][0], c->net[1][0], c->net[2][0], c->bdiff[0], c->bdiff[1], c->bdiff[2], c->MB_ref,
                          c->MB_vec, W, H, use_golden, use_altref);
    vp8o_pack_8x8_into_16x16(c->MB_vec, c->MB_parts, c->MB_SSIM, c->mbs);

    /* predictors, src/inter_part.h:268-321 */
    for (int r = 0; r < 3; ++r) {
        if (!use[r]) continue;
        vp8o_prepare_predictors_and_residual(c->cur.Y[0], c->ref[r].Y[0], c->pred[0], c->resid[0], c->MB_ref,
                                             c->MB_vec, W, H, 0, r);
        vp8o_prepare_predictors_and_residual(c->cur.U, c->ref[r].U, c->pred[1], c->resid[1], c->MB_ref, c->MB_vec,
                                             CW, CH, 1, r);
        vp8o_prepare_predictors_and_residual(c->cur.V, c->ref[r].V, c->pred[2], c->resid[2], c->MB_ref, c->MB_vec,
                                             CW, CH, 2, r);
    }
    /* segment loop LQ..UQ, src/inter_part.h:329-378 */
    for (int seg = 3; seg >= 0; --seg) {
        vp8o_dct4x4(c->resid[0], c->MB_coeffs, c->MB_seg, c->MB_parts, c->MB_SSIM, W, H, c->SD, seg, c->ssim_target, 0);
        vp8o_dct4x4(c->resid[1], c->MB_coeffs, c->MB_seg, c->MB_parts, c->MB_SSIM, CW, CH, c->SD, seg, c->ssim_target, 1);
        vp8o_dct4x4(c->resid[2], c->MB_coeffs, c->MB_seg, c->MB_parts, c->MB_SSIM, CW, CH, c->SD, seg, c->ssim_target, 2);
        vp8o_wht4x4_iwht4x4(c->MB_coeffs, c->MB_seg, c->MB_parts, c->SD, seg, c->mbs);
        vp8o_idct4x4(c->recon[0], c->pred[0], c->MB_coeffs, c->MB_seg, c->MB_parts, W, H, c->SD, seg, 0);
        vp8o_idct4x4(c->recon[1], c->pred[1], c->MB_coeffs, c->MB_seg, c->MB_parts, CW, CH, c->SD, seg, 1);
        vp8o_idct4x4(c->recon[2], c->pred[2], c->MB_coeffs, c->MB_seg, c->MB_parts, CW, CH, c->SD, seg, 2);
        vp8o_count_SSIM(c->cur.Y[0], c->recon[0], c->MB_seg, c->metric[0], W, H, seg, 16);
        vp8o_count_SSIM(c->cur.U, c->recon[1], c->MB_seg, c->metric[1], CW, CH, seg, 8);
        vp8o_count_SSIM(c->cur.V, c->recon[2], c->MB_seg, c->metric[2], CW, CH, seg, 8);
        vp8o_gather_SSIM(c->metric[0], c->metric[1], c->metric[2], c->MB_SSIM, c->mbs);
    }
    if (out) {
        const size_t n = c->mbs;
        if (out->prefilter_Y) memcpy(out->prefilter_Y, c->recon[0], (size_t)W * H);
        if (out->prefilter_U) memcpy(out->prefilter_U, c->recon[1], (size_t)CW * CH);
        if (out->prefilter_V) memcpy(out->prefilter_V, c->recon[2], (size_t)CW * CH);
        if (out->MB_parts) memcpy(out->MB_parts, c->MB_parts, n * 4);
        if (out->MB_reference_frame) memcpy(out->MB_reference_frame, c->MB_ref, n * 4);
        if (out->MB_vectors) memcpy(out->MB_vectors, c->MB_vec, n * 16);
        if (out->MB_coeffs) memcpy(out->MB_coeffs, c->MB_coeffs, n * 800);
        if (out->MB_segment_id) memcpy(out->MB_segment_id, c->MB_seg, n * 4);
        if (out->MB_SSIM) memcpy(out->MB_SSIM, c->MB_SSIM, n * 4);
    }
}

/* prepare_filter_mask_and_non_zero_coeffs() + do_loop_filter(): src/loop_filter.h:25-55 then :140-183
 * (three planes concurrently, one thread each), after which the filtered reconstruction is LAST. */
void vp8o_loop_filter(vp8o_ctx *c, vp8o_results *out) {
    const int W = c->W, H = c->H, CW = W / 2, CH = H / 2;
    vp8o_prepare_filter_mask(c->MB_coeffs, c->MB_nz, c->MB_parts, c->mb_mask, W, H);
#pragma omp parallel sections
    {
#pragma omp section
        vp8o_loop_filter_frame(c->recon[0], c->MB_seg, c->mb_mask, c->SD, W, H, 16);
#pragma omp section
        vp8o_loop_filter_frame(c->recon[1], c->MB_seg, c->mb_mask, c->SD, CW, CH, 8);
#pragma omp section
        vp8o_loop_filter_frame(c->recon[2], c->MB_seg, c->mb_mask, c->SD, CW, CH, 8);
    }
    /* the filtered reconstruction is the next LAST, src/vp8enc.cpp:395-401 */
    memcpy(c->ref[0].Y[0], c->recon[0], (size_t)W * H);
    memcpy(c->ref[0].U, c->recon[1], (size_t)CW * CH);
    memcpy(c->ref[0].V, c->recon[2], (size_t)CW * CH);
    if (out) {
        const size_t n = c->mbs;
        if (out->MB_non_zero_coeffs) memcpy(out->MB_non_zero_coeffs, c->MB_nz, n * 4);
        if (out->mb_mask) memcpy(out->mb_mask, c->mb_mask, n * 4);
        if (out->recon_Y) memcpy(out->recon_Y, c->recon[0], (size_t)W * H);
        if (out->recon_U) memcpy(out->recon_U, c->recon[1], (size_t)CW * CH);
        if (out->recon_V) memcpy(out->recon_V, c->recon[2], (size_t)CW * CH);
    }
}

/* host-side changes between transform and loop filter (vp8hip_upload_mb_data / vp8hip_upload_recon) */
void vp8o_upload_mb_data(vp8o_ctx *c, const int16_t *coeffs, const int32_t *parts, const int32_t *seg) {
    if (coeffs) memcpy(c->MB_coeffs, coeffs, (size_t)c->mbs * 800);
    if (parts) memcpy(c->MB_parts, parts, (size_t)c->mbs * 4);
    if (seg) memcpy(c->MB_seg, seg, (size_t)c->mbs * 4);
}
void vp8o_upload_recon(vp8o_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v) {
    memcpy(c->recon[0], y, (size_t)c->W * c->H);
    memcpy(c->recon[1], u, (size_t)c->W * c->H / 4);
    memcpy(c->recon[2], v, (size_t)c->W * c->H / 4);
}

const int16_t *vp8o_debug_net(const vp8o_ctx *c, int ref, int which) { return c->net[ref][which - 1]; }
const int32_t *vp8o_debug_bdiff(const vp8o_ctx *c, int ref) { return c->bdiff[ref]; }
const uint8_t *vp8o_debug_pyramid(const vp8o_ctx *c, int ref, int level) {
    return ref == 3 ? c->cur.Y[level] : c->ref[ref].Y[level];
}
