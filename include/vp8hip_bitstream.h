/*
 * vp8hip_bitstream.h -- the host half of the reference's entropy stage and its container, as a C ABI:
 * frame header + per-macroblock modes / motion vectors (first partition), frame assembly, IVF.
 * Implemented in vp8oclenc_amd/csrc/vp8_bitstream.cpp (plain C++, no GPU code, no global state).
 *
 * What each entry point replaces in the reference:
 *   vp8bs_default_probs   the fallback loop of entropy_encode(),   src/vp8enc.cpp:69-76
 *   vp8bs_encode_header   encode_header(),                         src/entropy_host.cpp:709-1256
 *                         (with write_mv :125-207, bool_encode_inter_mb_modes_and_mvs :209-443,
 *                          count_mv / count_mv_probs :445-707 and the boolean encoder :20-110)
 *   vp8bs_gather_frame    gather_frame(),                          src/encIO.h:1-30
 *   vp8bs_ivf_*           write_output_header / write_output_file, src/encIO.h:32-139
 *
 * Together with vp8hip_count_probs / vp8hip_encode_coefficients (include/vp8hip.h) this turns the device results
 * of a frame into the bytes the reference writes to its .ivf file.
 */
#ifndef VP8HIP_BITSTREAM_H
#define VP8HIP_BITSTREAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VP8BS_NUM_COEFF_PROBS 1056   /* [4][8][3][11] */

typedef struct {
    int32_t width, height;            /* video.dst_width / dst_height: written into key frames (:1244-1247) */
    int32_t mb_width, mb_height;
    int32_t is_key, is_golden, is_altref;          /* frames.current_is_{key,golden,altref}_frame */
    int32_t loop_filter_type;         /* video.loop_filter_type (0, init.h:1583) */
    int32_t loop_filter_sharpness;    /* video.loop_filter_sharpness as left by prepare_segments_data */
    int32_t partitions_log2;          /* video.number_of_partitions_ind: 0..3 */
    int32_t skip_prob;                /* frames.skip_prob, loop_filter.h:37-44 (vp8host_skip_prob) */
    int32_t replaced;                 /* frames.replaced of check_SSIM (0 when it did not run) */
    const int32_t *segments;          /* frames.segments_data: 4 x 11 ints */
    const int32_t *MB_segment_id;     /* [MBs] */
    const int32_t *MB_non_zero_coeffs;/* [MBs] */
    const int32_t *MB_reference_frame;/* [MBs] 0 LAST, 1 GOLDEN, 2 ALTREF (inter frames) */
    const int32_t *MB_parts;          /* [MBs] 0 = 16x16, 1 = 8x8 (inter macroblocks) */
    const int16_t *MB_vectors;        /* [MBs][4][2]: x, y in quarter pixels (inter frames) */
    const int32_t *is_inter_mb;       /* [MBs] e_data.is_inter_mb, or NULL = every macroblock inter (inter frames) */
    const int32_t *modes;             /* [MBs][16] e_data.mode: sub-block modes of intra macroblocks, or NULL if there are none */
    const uint32_t *new_probs;        /* [4][8][3][11] after vp8bs_default_probs */
    const uint32_t *new_probs_denom;  /* [4][8][3][11] partition 0's denominators (< 2 = context never seen) */
} vp8bs_frame;

/* contexts that were never seen take the format's default probability (src/vp8enc.cpp:69-76) */
void vp8bs_default_probs(uint32_t *new_probs, const uint32_t *new_probs_denom);

/* First partition with its uncompressed chunk (3 bytes, 10 for a key frame).  Returns its size
 * (= frames.encoded_frame_size after encode_header), 0 if `capacity` is too small, or (size_t)-1 if the partition has 512 KB
 * or more, which the frame tag's 19-bit size field cannot say (VP8HIP_ERR_FORMAT in vp8hip.h).  out_mv_probs (may be NULL)
 * receives the frame's 2 x 19 motion-vector probabilities (new_mv_context). */
size_t vp8bs_encode_header(const vp8bs_frame *f, uint8_t *out, size_t capacity, uint8_t *out_mv_probs);

/* Appends the partition sizes (3 bytes each, all but the last) and the coefficient partitions to a frame that holds
 * `header_size` bytes.  Partition p lies at partitions + p * partition_step.  Returns the frame size, 0 if it does
 * not fit. */
size_t vp8bs_gather_frame(uint8_t *frame, size_t header_size, size_t capacity, int num_partitions,
                          const uint8_t *partitions, size_t partition_step, const int32_t *partition_sizes);

/* IVF container: 32-byte file header, 12-byte frame header (little endian).  Return the bytes written. */
size_t vp8bs_ivf_file_header(uint8_t out[32], int width, int height, uint32_t framerate, uint32_t timescale,
                             uint32_t frame_count);
size_t vp8bs_ivf_frame_header(uint8_t out[12], uint32_t frame_size, uint64_t timestamp);

#ifdef __cplusplus
}
#endif
#endif
