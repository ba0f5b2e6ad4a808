/*
 * ref_driver.c -- NDRange loops around the reference's own kernels (compiled for x86 by
 * oracle/build_ref.sh).  TEST INFRASTRUCTURE ONLY.
 *
 * Exports ref_<kernel>() with the same argument lists as vp8o_<kernel>() in vp8_oracle.h so a
 * test can run the reference kernel and the restatement on the same buffers.  Global sizes
 * follow the host code: src/inter_part.h:5,11-33,110-236,251-378, src/loop_filter.h:30-32,143.
 * `__local` arrays are function-scope statics in the x86 objects: single-threaded only.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const uint8_t *data; int w, h; } ref_image;
/* work-item ids read by the built-ins in ref_shim.cl */
static size_t g_gid, g_lsz = 256;
size_t ref_gid(void) { return g_gid; }
size_t ref_lsz(void) { return g_lsz; }
static void ref_set_ids(size_t gid, size_t lsz) { g_gid = gid; g_lsz = lsz; }

/* kernel entry points, src/GPU_kernels.cl:404,429,459,1070,1205,1285,1346,1368,1498,1545,1610,1973,2097 */
void downsample_x2(const uint8_t *src, uint8_t *dst, int src_w, int src_h);
void luma_search_1step(const uint8_t *cur, const uint8_t *prev, const int16_t *src_net, int16_t *dst_net,
                       int net_width, int width, int height, int pixel_rate);
void luma_search_2step(const uint8_t *cur, const ref_image *ref, const int16_t *net, int16_t *ref_net,
                       int32_t *ref_Bdiff, int width, int height);
void select_reference(const int16_t *l, const int16_t *g, const int16_t *a, const int32_t *lB, const int32_t *gB,
                      const int32_t *aB, int32_t *MB_ref, int16_t *MB_vec, int width, int use_golden,
                      int use_altref);
void pack_8x8_into_16x16(const int16_t *MB_vec, int32_t *MB_parts, float *MB_SSIM);
void prepare_predictors_and_residual(const uint8_t *cur, const ref_image *ref, uint8_t *pred, int16_t *resid,
                                     const int32_t *MB_ref, const int16_t *MB_vec, int width, int plane, int ref_id);
void dct4x4(const int16_t *resid, int16_t *MB, int32_t *MB_seg, const int32_t *MB_parts, const float *MB_SSIM,
            int width, const int32_t *SD, int seg, float ssim_target, int plane);
void wht4x4_iwht4x4(int16_t *MB, const float *MB_SSIM, int32_t *MB_seg, const int32_t *MB_parts, const int32_t *SD,
                    int seg);
void idct4x4(uint8_t *recon, const uint8_t *pred, const int16_t *MB, const int32_t *MB_seg, const int32_t *MB_parts,
             int width, const int32_t *SD, int seg, int plane);
void count_SSIM_luma(const uint8_t *f1, const uint8_t *f2, const int32_t *MB_seg, float *metric, int width, int seg);
void count_SSIM_chroma(const uint8_t *f1, const uint8_t *f2, const int32_t *MB_seg, float *metric, int cwidth,
                       int seg);
void gather_SSIM(const float *m1, const float *m2, const float *m3, float *MB_SSIM);
/* src/CPU_kernels.cl:782,970,1333 */
void prepare_filter_mask(const int16_t *MB, int32_t *nz, const int32_t *MB_parts, int32_t *mb_mask, int width,
                         int height, int parts);
void loop_filter_frame_luma(uint8_t *frame, const int32_t *seg, const int32_t *mask, const int32_t *SD, int width,
                            int height);
void loop_filter_frame_chroma(uint8_t *frame, const int32_t *seg, const int32_t *mask, const int32_t *SD, int width,
                              int height);

#define NDRANGE(n, lsz, call)                       \
    for (size_t gid_ = 0; gid_ < (size_t)(n); ++gid_) { \
        ref_set_ids(gid_, (lsz));                   \
        call;                                       \
    }

static size_t round256(size_t n) { return (n % 256) ? n + 256 - (n % 256) : n; }

void ref_downsample_x2(const uint8_t *src, uint8_t *dst, int src_w, int src_h) {
    NDRANGE((size_t)src_w * src_h / 4, 256, downsample_x2(src, dst, src_w, src_h));
}

/* The kernel evaluates out-of-frame candidates before masking them (src/GPU_kernels.cl:525-549),
 * so the reference plane is copied into the middle of a guard buffer. */
void ref_luma_search_1step(const uint8_t *cur, const uint8_t *ref, const int16_t *src_net, int16_t *dst_net,
                           int net_width, int width, int height, int pixel_rate) {
    const size_t plane = (size_t)width * height;
    const size_t guard = 4 * plane + 65536;
    uint8_t *buf = (uint8_t *)calloc(plane + 2 * guard, 1);
    uint8_t *curbuf = (uint8_t *)calloc(plane + 4096, 1);
    memcpy(buf + guard, ref, plane);
    memcpy(curbuf, cur, plane);
    const size_t n = round256((size_t)(width / 8) * (height / 8));
    NDRANGE(n, 256, luma_search_1step(curbuf, buf + guard, src_net, dst_net, net_width, width, height, pixel_rate));
    free(buf);
    free(curbuf);
}

void ref_luma_search_2step(const uint8_t *cur, const uint8_t *ref, const int16_t *net, int16_t *ref_net,
                           int32_t *ref_Bdiff, int width, int height) {
    ref_image img = {ref, width, height};
    const size_t n = round256((size_t)width * height / 64);
    NDRANGE(n, 256, luma_search_2step(cur, &img, net, ref_net, ref_Bdiff, width, height));
}

void ref_select_reference(const int16_t *last_net, const int16_t *golden_net, const int16_t *altref_net,
                          const int32_t *last_Bdiff, const int32_t *golden_Bdiff, const int32_t *altref_Bdiff,
                          int32_t *MB_ref, int16_t *MB_vectors, int width, int height, int use_golden,
                          int use_altref) {
    NDRANGE((size_t)(width / 16) * (height / 16), 256,
            select_reference(last_net, golden_net, altref_net, last_Bdiff, golden_Bdiff, altref_Bdiff, MB_ref,
                             MB_vectors, width, use_golden, use_altref));
}

void ref_pack_8x8_into_16x16(const int16_t *MB_vectors, int32_t *MB_parts, float *MB_SSIM, int mb_count) {
    NDRANGE(mb_count, 256, pack_8x8_into_16x16(MB_vectors, MB_parts, MB_SSIM));
}

void ref_prepare_predictors_and_residual(const uint8_t *cur, const uint8_t *ref, uint8_t *predictor,
                                         int16_t *residual, const int32_t *MB_ref, const int16_t *MB_vectors,
                                         int width, int height, int plane, int ref_id) {
    ref_image img = {ref, width, height};
    NDRANGE((size_t)(width / 4) * (height / 4), 256,
            prepare_predictors_and_residual(cur, &img, predictor, residual, MB_ref, MB_vectors, width, plane, ref_id));
}

void ref_dct4x4(const int16_t *residual, int16_t *MB, int32_t *MB_segment_id, const int32_t *MB_parts,
                const float *MB_SSIM, int width, int height, const int32_t *SD, int segment_id, float SSIM_target,
                int plane) {
    NDRANGE((size_t)(width / 4) * (height / 4), 256,
            dct4x4(residual, MB, MB_segment_id, MB_parts, MB_SSIM, width, SD, segment_id, SSIM_target, plane));
}

void ref_wht4x4_iwht4x4(int16_t *MB, const int32_t *MB_segment_id, const int32_t *MB_parts, const int32_t *SD,
                        int segment_id, int mb_count) {
    NDRANGE(mb_count, 256, wht4x4_iwht4x4(MB, NULL, (int32_t *)MB_segment_id, MB_parts, SD, segment_id));
}

void ref_idct4x4(uint8_t *recon, const uint8_t *predictor, const int16_t *MB, const int32_t *MB_segment_id,
                 const int32_t *MB_parts, int width, int height, const int32_t *SD, int segment_id, int plane) {
    NDRANGE((size_t)(width / 4) * (height / 4), 256,
            idct4x4(recon, predictor, MB, MB_segment_id, MB_parts, width, SD, segment_id, plane));
}

void ref_count_SSIM(const uint8_t *f1, const uint8_t *f2, const int32_t *MB_segment_id, float *metric, int width,
                    int height, int segment_id, int mb_size) {
    const size_t n = (size_t)(width / mb_size) * (height / mb_size);
    if (mb_size == 16) {
        NDRANGE(n, 256, count_SSIM_luma(f1, f2, MB_segment_id, metric, width, segment_id));
    } else {
        NDRANGE(n, 256, count_SSIM_chroma(f1, f2, MB_segment_id, metric, width, segment_id));
    }
}

void ref_gather_SSIM(const float *m1, const float *m2, const float *m3, float *MB_SSIM, int mb_count) {
    NDRANGE(mb_count, 256, gather_SSIM(m1, m2, m3, MB_SSIM));
}

void ref_prepare_filter_mask(const int16_t *MB, int32_t *MB_non_zero_coeffs, const int32_t *MB_parts,
                             int32_t *mb_mask, int width, int height) {
    NDRANGE(4, 1, prepare_filter_mask(MB, MB_non_zero_coeffs, MB_parts, mb_mask, width, height, 4));
}

void ref_loop_filter_frame(uint8_t *frame, const int32_t *MB_segment_ids, const int32_t *mb_mask, const int32_t *SD,
                           int width, int height, int mb_size) {
    ref_set_ids(0, 1);
    if (mb_size == 16)
        loop_filter_frame_luma(frame, MB_segment_ids, mb_mask, SD, width, height);
    else
        loop_filter_frame_chroma(frame, MB_segment_ids, mb_mask, SD, width, height);
}

/* ---- coefficient entropy stage, src/CPU_kernels.cl:347-778 (run on the CPU device by the reference,
 * one work-item per partition: src/vp8enc.cpp:48-94) ------------------------------------------------- */
void count_probs(const int16_t *MB, const int32_t *nz, const int32_t *MB_parts, uint32_t *coeff_probs,
                 uint32_t *coeff_probs_denom, uint8_t *third_context, int mb_height, int mb_width,
                 int num_partitions, int partition_step);
void num_div_denom(uint32_t *coeff_probs, const uint32_t *coeff_probs_denom, int num_partitions);
void encode_coefficients(const int16_t *MB, const int32_t *nz, const int32_t *MB_parts, uint8_t *output,
                         int32_t *partition_sizes, const uint8_t *third_context, const uint32_t *coeff_probs,
                         int mb_height, int mb_width, int num_partitions, int partition_step);

void ref_count_probs(const int16_t *MB, const int32_t *nz, const int32_t *MB_parts, uint32_t *coeff_probs,
                     uint32_t *coeff_probs_denom, uint8_t *third_context, int mb_height, int mb_width,
                     int num_partitions) {
    NDRANGE(num_partitions, 1, count_probs(MB, nz, MB_parts, coeff_probs, coeff_probs_denom, third_context, mb_height,
                                           mb_width, num_partitions, 0));
}
void ref_num_div_denom(uint32_t *coeff_probs, const uint32_t *coeff_probs_denom, int num_partitions) {
    NDRANGE(num_partitions, 1, num_div_denom(coeff_probs, coeff_probs_denom, num_partitions));
}
void ref_encode_coefficients(const int16_t *MB, const int32_t *nz, const int32_t *MB_parts, uint8_t *output,
                             int32_t *partition_sizes, const uint8_t *third_context, const uint32_t *coeff_probs,
                             int mb_height, int mb_width, int num_partitions, int partition_step) {
    NDRANGE(num_partitions, 1, encode_coefficients(MB, nz, MB_parts, output, partition_sizes, third_context,
                                                   coeff_probs, mb_height, mb_width, num_partitions, partition_step));
}
