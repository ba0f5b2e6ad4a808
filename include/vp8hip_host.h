/*
 * vp8hip_host.h -- host-side mirror (C++ behind a C ABI, no GPU needed) of the reference host
 * code that PRODUCES the parameters of the inter-frame path and sequences it.  The reference keeps
 * this logic in vp8enc.cpp / init.h around its OpenCL calls; a maintainer who swaps the OpenCL calls
 * for include/vp8hip.h keeps those functions as they are.  They are restated here so that the
 * parity tests and bench.py drive the device path with the same numbers the reference would.
 */
#ifndef VP8HIP_HOST_H
#define VP8HIP_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ParseArgs quantizer ladders, init.h:1585-1603.  Index 0..3 = UQ, HQ, AQ, LQ. */
void vp8host_quantizer_ladders(int qi_min, int qi_max, int32_t lastqi[4], int32_t altrefqi[4]);

/* get_loopfilter_strength(), vp8enc.cpp:96-127: brightness-based divisor and sharpness (0..7) */
void vp8host_loopfilter_strength(const uint8_t *cur_y, int width, int height, int32_t *reductor, int32_t *sharpness);

/* scene_change(), vp8enc.cpp:265-311: the decision on the two chroma differences (vp8hip_chroma_change), with the
 * reference's hold-over (its function-static `holdover`) and frames.last_key_detect in `st`.  Returns 1 when the
 * current frame must be coded as a key frame; the caller then sets st->last_key_detect = frame_number
 * (intra_transform does, intra_part.h:1091-1098). */
typedef struct { int32_t holdover, last_key_detect; } vp8host_scene_state;
int vp8host_scene_change(vp8host_scene_state *st, int Udiff, int Vdiff, int frame_number);

/* prepare_segments_data(), vp8enc.cpp:129-221.  refqi = lastqi or altrefqi (vp8enc.cpp:149-151);
 * update_filter/shrpnss: the second call made from check_SSIM (vp8enc.cpp:260-261). */
void vp8host_prepare_segments_data(int is_key_frame, const int32_t refqi[4], int qi_min, int reductor,
                                   int sharpness, int update_filter, int shrpnss, int32_t sd[44]);

/* The input format: YUV4MPEG2.  OpenYUV420FileAndParseHeader(), init.h:1610-1737, on the first `size` bytes of the stream
 * (128 are plenty: the reference keeps the header in a 128-byte array): the magic word, then the FIRST THREE tags that start
 * with W, H or F -- width, height, frame rate num:denom rounded to (num + denom / 2) / denom -- each ended by a space, then
 * everything up to and including the first "FRAME\n" (a FRAME line with parameters is refused, :1724-1726).  Returns 0 and
 * the offset of the first frame's samples, or -1 as the reference does (not YUV4MPEG2, no size, no plain FRAME line, or the
 * buffer ends first).  Frames follow as tight I420 of width x height, each followed by the next one's 6-byte "FRAME\n"
 * (get_yuv420_frame checks bytes 0 and 4 of it, encIO.h:243-248: vp8host_y4m_frame_marker_ok). */
int vp8host_y4m_parse_header(const uint8_t *data, size_t size, int32_t *width, int32_t *height, int32_t *framerate, size_t *first_frame_offset);
int vp8host_y4m_frame_marker_ok(const uint8_t marker[6]);

/* frames.skip_prob, loop_filter.h:37-44 */
int vp8host_skip_prob(const int32_t *MB_non_zero_coeffs, int mb_count);

/* frame-type state machine, vp8enc.cpp:340-344, 364-374 and intra_part.h:1091-1098 */
typedef struct {
    int32_t gop_size, altref_range;
    int32_t frame_number, frames_until_key, frames_until_altref;
    int32_t golden_frame_number, altref_frame_number;
    int32_t current_is_key, current_is_golden, current_is_altref;
    int32_t prev_is_key, prev_is_golden, prev_is_altref;
} vp8host_gop;

void vp8host_gop_init(vp8host_gop *g, int gop_size, int altref_range);
/* advance to the next input frame; fills the current_* / prev_* flags */
void vp8host_gop_next(vp8host_gop *g);
/* what intra_transform() does to the counters when the current frame is (or is forced to be) a key frame */
void vp8host_gop_key_coded(vp8host_gop *g);
/* flags of inter_transform(), inter_part.h:103-104 */
void vp8host_gop_inter_flags(const vp8host_gop *g, int32_t *use_golden, int32_t *use_altref);
/* ++frames.frame_number at the end of the loop body, vp8enc.cpp:487 */
void vp8host_gop_frame_done(vp8host_gop *g);

#ifdef __cplusplus
}
#endif
#endif
