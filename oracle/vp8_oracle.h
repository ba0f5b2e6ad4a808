/*
 * vp8_oracle.h -- CPU restatement of the vp8oclenc inter-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file (and everything under oracle/) is the
 * parity checker and the timed CPU baseline.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path
 * (vp8oclenc_amd/csrc, libvp8hip.so) never links or calls it.
 *
 * Parity pin: every function below is checked bit-for-bit (SSIM: 1e-4) against the
 * reference's own kernels (src/GPU_kernels.cl, src/CPU_kernels.cl) compiled for
 * x86 by oracle/build_ref.sh into oracle/_ref/libvp8ref.so (tests/test_oracle_vs_ref.py),
 * and against the golden vectors that build produced (tests/golden/).  The reference
 * ships no tests or golden vectors of its own.
 *
 * Each function cites the reference kernel it restates (paths relative to
 * /root/reference).  All planes are tightly packed, row-major, stride == width.
 */
#ifndef VP8_ORACLE_H
#define VP8_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/vp8enc.h:80-92 -- 11 ints per segment, 4 segments */
enum {
    SD_Y_AC_I = 0, SD_Y_DC_IDELTA, SD_Y2_DC_IDELTA, SD_Y2_AC_IDELTA, SD_UV_DC_IDELTA,
    SD_UV_AC_IDELTA, SD_LOOP_FILTER_LEVEL, SD_MBEDGE_LIMIT, SD_SUB_BEDGE_LIMIT,
    SD_INTERIOR_LIMIT, SD_HEV_THRESHOLD, SD_INTS = 11
};

enum { VP8O_LAST = 0, VP8O_GOLDEN = 1, VP8O_ALTREF = 2 };
enum { VP8O_16x16 = 0, VP8O_8x8 = 1 };

/* src/GPU_kernels.cl:85-190 (weight_opt): cost of one 4x4 difference block, d[16] row-major */
int vp8o_weight(const int d[16]);

/* src/GPU_kernels.cl:429-451 */
void vp8o_downsample_x2(const uint8_t *src, uint8_t *dst, int src_w, int src_h);

/* src/GPU_kernels.cl:459-560.  nets are short2 cells: net[2*cell+0]=x, [2*cell+1]=y */
void vp8o_luma_search_1step(const uint8_t *cur, const uint8_t *ref, const int16_t *src_net,
                            int16_t *dst_net, int net_width, int width, int height, int pixel_rate);

/* src/GPU_kernels.cl:1068-1203 (+construct_opt1/2 :776-1066); ref addressed clamp-to-edge */
void vp8o_luma_search_2step(const uint8_t *cur, const uint8_t *ref, const int16_t *net,
                            int16_t *ref_net, int32_t *ref_Bdiff, int width, int height);

/* src/GPU_kernels.cl:1205-1283 */
void vp8o_select_reference(const int16_t *last_net, const int16_t *golden_net, const int16_t *altref_net,
                           const int32_t *last_Bdiff, const int32_t *golden_Bdiff, const int32_t *altref_Bdiff,
                           int32_t *MB_ref, int16_t *MB_vectors, int width, int height,
                           int use_golden, int use_altref);

/* src/GPU_kernels.cl:1346-1366 */
void vp8o_pack_8x8_into_16x16(const int16_t *MB_vectors, int32_t *MB_parts, float *MB_SSIM, int mb_count);

/* src/GPU_kernels.cl:1285-1344 (+construct :574-774).  plane: 0=Y 1=U 2=V; width/height of THAT plane */
void vp8o_prepare_predictors_and_residual(const uint8_t *cur, const uint8_t *ref, uint8_t *predictor,
                                          int16_t *residual, const int32_t *MB_ref, const int16_t *MB_vectors,
                                          int width, int height, int plane, int ref_id);

/* src/GPU_kernels.cl:1368-1496.  MB: short[mb_count][25][16] */
void vp8o_dct4x4(const int16_t *residual, int16_t *MB, int32_t *MB_segment_id, const int32_t *MB_parts,
                 const float *MB_SSIM, int width, int height, const int32_t *SD, int segment_id,
                 float SSIM_target, int plane);

/* src/GPU_kernels.cl:1498-1543 */
void vp8o_wht4x4_iwht4x4(int16_t *MB, const int32_t *MB_segment_id, const int32_t *MB_parts,
                         const int32_t *SD, int segment_id, int mb_count);

/* src/GPU_kernels.cl:1545-1608 */
void vp8o_idct4x4(uint8_t *recon, const uint8_t *predictor, const int16_t *MB, const int32_t *MB_segment_id,
                  const int32_t *MB_parts, int width, int height, const int32_t *SD, int segment_id, int plane);

/* src/GPU_kernels.cl:1610-1971 (luma, mb 16x16) and :1973-2095 (chroma, 8x8); float */
void vp8o_count_SSIM(const uint8_t *f1, const uint8_t *f2, const int32_t *MB_segment_id, float *metric,
                     int width, int height, int segment_id, int mb_size);
/* src/GPU_kernels.cl:2097-2105 */
void vp8o_gather_SSIM(const float *m1, const float *m2, const float *m3, float *MB_SSIM, int mb_count);

/* src/CPU_kernels.cl:782-827 */
void vp8o_prepare_filter_mask(const int16_t *MB, int32_t *MB_non_zero_coeffs, const int32_t *MB_parts,
                              int32_t *mb_mask, int width, int height);

/* src/CPU_kernels.cl:970-1075 (mb_size 16) and :1333-1439 (mb_size 8); width/height of the plane */
void vp8o_loop_filter_frame(uint8_t *frame, const int32_t *MB_segment_ids, const int32_t *mb_mask,
                            const int32_t *SD, int width, int height, int mb_size);

/* Coefficient entropy stage (vp8_entropy_oracle.c), src/CPU_kernels.cl:347-778; all partitions in one call.
 * coeff_probs / coeff_probs_denom: uint[num_partitions][4][8][3][11]; third_context: uchar[MBs][25];
 * after vp8o_num_div_denom the first [4][8][3][11] of coeff_probs are the frame's probabilities. */
void vp8o_count_probs(const int16_t *MB, const int32_t *MB_non_zero_coeffs, const int32_t *MB_parts,
                      uint32_t *coeff_probs, uint32_t *coeff_probs_denom, uint8_t *third_context, int mb_height,
                      int mb_width, int num_partitions);
void vp8o_num_div_denom(uint32_t *coeff_probs, const uint32_t *coeff_probs_denom, int num_partitions);
void vp8o_encode_coefficients(const int16_t *MB, const int32_t *MB_non_zero_coeffs, const int32_t *MB_parts,
                              uint8_t *output, int32_t *partition_sizes, const uint8_t *third_context,
                              const uint32_t *coeff_probs, int mb_height, int mb_width, int num_partitions,
                              int partition_step);

/* Host intra path (vp8_intra_oracle.c): src/intra_part.h and check_SSIM of src/vp8enc.cpp:231-263.  Planes have
 * stride = width; coeffs [MBs][25][16] (blocks 0..23 written, zigzag order); modes [MBs][16]; sd = 4 x 11 ints. */
int vp8o_host_weight(const int16_t r[16]);
int vp8o_pick_luma_predictor(const uint8_t orig[16], uint8_t pred[16], int16_t resid[16], const int16_t top[8],
                             const int16_t left[4], int top_left);
float vp8o_count_ssim_16x16(const uint8_t *y1, const uint8_t *u1, const uint8_t *v1, int w1, const uint8_t *y2,
                            const uint8_t *u2, const uint8_t *v2, int w2);
void vp8o_intra_transform(int width, int height, const uint8_t *cy, const uint8_t *cu, const uint8_t *cv, const int32_t sd[44],
                          uint8_t *ry, uint8_t *ru, uint8_t *rv, int16_t *coeffs, int32_t *parts, int32_t *seg, int32_t *modes);
void vp8o_check_ssim(int width, int height, float ssim_target, const uint8_t *cy, const uint8_t *cu, const uint8_t *cv,
                     const int32_t sd[44], uint8_t *ry, uint8_t *ru, uint8_t *rv, int16_t *coeffs, int32_t *parts, int32_t *seg,
                     float *ssim, int32_t *is_inter, int32_t *modes, int32_t *replaced, float *new_ssim, float *min_ssim);

/* ------------------------------------------------------------------------------------------
 * Whole inter frame, in the enqueue order of src/inter_part.h:96-384 followed by
 * src/loop_filter.h:25-55,140-183.  One context keeps the three references and their pyramids
 * the way prepare_GPU_buffers() (src/inter_part.h:1-94) rotates them.
 * ------------------------------------------------------------------------------------------ */
typedef struct vp8o_ctx vp8o_ctx;

typedef struct {
    int32_t *MB_parts;            /* [MBs] */
    int32_t *MB_reference_frame;  /* [MBs] */
    int16_t *MB_vectors;          /* [MBs][4][2] qpel */
    int16_t *MB_coeffs;           /* [MBs][25][16] */
    int32_t *MB_segment_id;       /* [MBs] */
    float   *MB_SSIM;             /* [MBs] */
    int32_t *MB_non_zero_coeffs;  /* [MBs] */
    int32_t *mb_mask;             /* [MBs] */
    uint8_t *recon_Y, *recon_U, *recon_V; /* loop-filtered reconstruction */
    uint8_t *prefilter_Y, *prefilter_U, *prefilter_V; /* recon before the loop filter (may be NULL) */
} vp8o_results;

vp8o_ctx *vp8o_create(int width, int height, float ssim_target);
void vp8o_destroy(vp8o_ctx *c);
/* LAST := these planes (key-frame reconstruction or host-modified recon), src/vp8enc.cpp:395-401 */
void vp8o_upload_last(vp8o_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v);
void vp8o_set_segments(vp8o_ctx *c, const int32_t sd[44]);
/* prepare_GPU_buffers + inter_transform on cur (src/inter_part.h:1-384).  Fills the MB_* members and
 * prefilter_* of *out (NULL members are skipped). */
void vp8o_inter_transform(vp8o_ctx *c, const uint8_t *cur_y, const uint8_t *cur_u, const uint8_t *cur_v,
                          int prev_is_golden, int prev_is_altref, int use_golden, int use_altref,
                          vp8o_results *out);
/* prepare_filter_mask + loop filter (src/loop_filter.h); afterwards the filtered reconstruction is
 * LAST.  Fills MB_non_zero_coeffs, mb_mask, recon_* of *out. */
void vp8o_loop_filter(vp8o_ctx *c, vp8o_results *out);
void vp8o_upload_mb_data(vp8o_ctx *c, const int16_t *coeffs, const int32_t *parts, const int32_t *seg);
void vp8o_upload_recon(vp8o_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v);
/* stage outputs of the last vp8o_inter_frame (for parity tests): level 0..4 = /16,/8,/4,/2,/1 */
const int16_t *vp8o_debug_net(const vp8o_ctx *c, int ref, int which /*1 or 2*/);
const int32_t *vp8o_debug_bdiff(const vp8o_ctx *c, int ref);
const uint8_t *vp8o_debug_pyramid(const vp8o_ctx *c, int ref /*0..2, 3 = current*/, int level /*0..4 = 1x..1/16*/);
/* copy_with_padding, src/encIO.h:141-196: tight src_w x src_h planes -> tight w x h planes (w, h = the padded "wrk" size).
 * Y, U and the bottom rows as the reference has them; the right padding of V as the reference MEANS it (its V lines,
 * :180-183, read from U and write into U's next row, so with a width that needs padding V's is never written -- undefined,
 * and none of BASELINE's configs has such a width; pinned and shown in tests/test_padding.py). */
void vp8o_copy_with_padding(const uint8_t *sy, const uint8_t *su, const uint8_t *sv, int src_w, int src_h, uint8_t *dy, uint8_t *du,
                            uint8_t *dv, int w, int h);
int vp8o_num_threads(void);
void vp8o_set_num_threads(int n);
/* NOT the reference, default 0.  1 = the two places where the reference's encoder and a decoder of its stream part ways
 * are closed: check_SSIM keeps the sub-block modes of the attempt it kept (vp8_intra_oracle.c), and the predictor saturates
 * all nine first-pass lines as the format says instead of wrapping the last three (vp8_oracle.c, interp4x4_construct).
 * Exists to prove that these two are the whole difference (tests/test_decode_roundtrip.py) and to check the product's
 * opt-in of the same meaning, vp8hip_conformant_stream. */
extern int vp8o_conformant;
void vp8o_set_conformant_stream(int on);

#ifdef __cplusplus
}
#endif
#endif
