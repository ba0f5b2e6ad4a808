/*
 * vp8hip.h -- C ABI of the MI355X (gfx950) inter-frame path of vp8oclenc.
 *
 * Drop-in boundary: each entry point replaces a group of OpenCL calls that the reference host
 * driver makes (citations are file:line under the reference's src/).  The library owns every
 * device allocation; the host owns every host buffer; no host pointer is kept after a call
 * returns.  All calls are made from one host thread per context.  Return value: 0 on success,
 * negative vp8hip_status on failure (the reference stores cl_int errors in device.state_gpu,
 * inter_part.h:380).  There is no CPU fallback: without a usable HIP device vp8hip_create fails.
 *
 * Plane convention at the boundary: tightly packed 8-bit planes of the padded ("wrk") size,
 * width and height multiples of 16 (init.h:381-389), chroma planes (W/2)x(H/2) -- exactly
 * hostFrameBuffers.current_Y/U/V and reconstructed_Y/U/V (vp8enc.h:364-372).
 */
#ifndef VP8HIP_H
#define VP8HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vp8hip_ctx vp8hip_ctx;

typedef enum {
    VP8HIP_OK = 0,
    VP8HIP_ERR_ARG = -1,       /* bad size / NULL pointer */
    VP8HIP_ERR_NO_DEVICE = -2, /* no HIP device, or device_ordinal out of range */
    VP8HIP_ERR_HIP = -3,       /* a HIP runtime call failed (vp8hip_last_hip_error) */
    VP8HIP_ERR_STATE = -4,     /* call out of order (e.g. loop filter before any transform) */
    VP8HIP_ERR_ARCH = -5,      /* device is not gfx950: the kernels are built for MI355X only */
    VP8HIP_ERR_TIMEOUT = -6,   /* a bounded device-side wait of the loop filter expired: the frame is invalid
                                  (reported by vp8hip_synchronize / vp8hip_download_*; the context stays usable) */
    VP8HIP_ERR_OVERFLOW = -7,  /* vp8hip_encode_coefficients: output or scratch too small for this frame */
    VP8HIP_ERR_FORMAT = -8     /* the frame cannot be written as VP8: its first partition has 512 KB or more and the frame tag
                                  has 19 bits for that size (RFC 6386 section 9.1) -- key frames of about 7000x4000 and up.  The
                                  reference writes the low 19 bits (entropy_host.cpp:1237-1241) and emits a frame no decoder
                                  can read; here the call fails */
} vp8hip_status;

/* segment_data[4], vp8enc.h:80-92: 11 ints per segment */
#define VP8HIP_SD_INTS 44

/* Host pointers filled by vp8hip_download_results; any member may be NULL (skipped).
 * Layouts: vp8enc.h:105-120, 378-382. */
typedef struct {
    int32_t *MB_parts;           /* [MBs] 0=16x16 1=8x8                         inter_part.h:263 */
    int32_t *MB_reference_frame; /* [MBs] 0 LAST 1 GOLDEN 2 ALTREF              inter_part.h:264 */
    int16_t *MB_vectors;         /* [MBs][4]{x,y} qpel, TL TR BL BR             inter_part.h:265 */
    int16_t *MB_coeffs;          /* [MBs][25][16] zig-zag order                 vp8enc.cpp:422   */
    int32_t *MB_segment_id;      /* [MBs]                                        vp8enc.cpp:423   */
    float *MB_SSIM;              /* [MBs]                                        vp8enc.cpp:424   */
    uint8_t *recon_Y;            /* reconstruction BEFORE the loop filter        vp8enc.cpp:431   */
    uint8_t *recon_U;            /*                                              vp8enc.cpp:432   */
    uint8_t *recon_V;            /*                                              vp8enc.cpp:433   */
} vp8hip_results;

/* init_all() GPU half (init.h:133-312, 430-582, 595-1166) / finalize() (vp8enc.cpp:501-708).
 * width,height: padded luma size.  ssim_target: video.SSIM_target (init.h:1512,1576). */
int vp8hip_create(vp8hip_ctx **out, int width, int height, float ssim_target, int device_ordinal);
void vp8hip_destroy(vp8hip_ctx *ctx);

/* clEnqueueWriteBuffer(current_frame_Y/U/V), vp8enc.cpp:386-388 */
int vp8hip_upload_current(vp8hip_ctx *ctx, const uint8_t *y, const uint8_t *u, const uint8_t *v);
/* The NEXT frame's planes started on their way early, while the current frame is coded: tight planes of the source size (one copy when
 * U follows Y and V follows U in memory), asynchronous when they are page-locked (vp8hip_host_alloc), into a staging buffer of the
 * context's.  The vp8hip_upload_current that names the same three pointers then copies nothing: it packs from the staging buffer, behind
 * the copy.  The planes stay unchanged until that vp8hip_upload_current has returned.  Touches nothing of the frame under way. */
int vp8hip_prefetch_current(vp8hip_ctx *ctx, const uint8_t *y, const uint8_t *u, const uint8_t *v);
/* same, from planes already resident in this device's memory (tight stride); async on the ctx stream */
int vp8hip_set_current_device(vp8hip_ctx *ctx, const void *d_y, const void *d_u, const void *d_v);

/* LAST := these (already loop-filtered) planes: vp8enc.cpp:395-401, intra_part.h:1112-1125.
 * Needed after a key frame and whenever the host changed the reconstruction. */
int vp8hip_upload_last(vp8hip_ctx *ctx, const uint8_t *y, const uint8_t *u, const uint8_t *v);
int vp8hip_set_last_device(vp8hip_ctx *ctx, const void *d_y, const void *d_u, const void *d_v);

/* The two per-frame host scans that produce parameters of this path, on the device copy of the
 * current frame -- at 0.2 ms per 1080p frame a single-threaded 2-Mpixel scan on the host would be the bottleneck.
 * Both block until the values are back.
 * get_loopfilter_strength(), vp8enc.cpp:96-127: reductor and sharpness of the current luma plane. */
int vp8hip_loopfilter_strength(vp8hip_ctx *ctx, int32_t *reductor, int32_t *sharpness);
/* scene_change()'s inputs, vp8enc.cpp:265-282: mean absolute difference of the U and V planes between this current
 * frame and the previous current frame (the context keeps both; 0,0 while there is only one).  The decision
 * logic with its hold-over stays on the host: vp8host_scene_change(), include/vp8hip_host.h. */
int vp8hip_chroma_change(vp8hip_ctx *ctx, int32_t *Udiff, int32_t *Vdiff);
/* ... in two halves: _async enqueues the scan behind the current frame's pack and returns; _result waits for the two sums only (they
 * arrive in page-locked memory; not for the stream) and returns what vp8hip_chroma_change returns.  A second _async before a _result
 * replaces the first; _result without a pending _async: VP8HIP_ERR_STATE.  For a host that hands the next frame over early (while the
 * previous frame's loop filter runs) and wants scene_change()'s verdict without a round trip in its critical path. */
int vp8hip_chroma_change_async(vp8hip_ctx *ctx);
int vp8hip_chroma_change_result(vp8hip_ctx *ctx, int32_t *Udiff, int32_t *Vdiff);

/* get_loopfilter_strength() + prepare_segments_data() (vp8enc.cpp:96-127, 129-221) evaluated on the device for the
 * current frame: the segment data of the following vp8hip_inter_transform / vp8hip_loop_filter are produced without
 * any host round trip (asynchronous on the context's stream) -- what a frame loop at several thousand frames per
 * second needs.  refqi = the lastqi or altrefqi ladder (vp8enc.cpp:149-151).  vp8hip_get_segments reads back the
 * segment data in force (the frame header needs them) and the strength pair; it blocks. */
int vp8hip_auto_segments(vp8hip_ctx *ctx, int is_key_frame, const int32_t refqi[4], int qi_min);
int vp8hip_get_segments(vp8hip_ctx *ctx, int32_t sd[VP8HIP_SD_INTS], int32_t *reductor, int32_t *sharpness);

/* clEnqueueWriteBuffer(segments_data_gpu), vp8enc.cpp:224 */
int vp8hip_set_segments(vp8hip_ctx *ctx, const int32_t sd[VP8HIP_SD_INTS]);

/* prepare_GPU_buffers() + inter_transform(), inter_part.h:1-94, 96-384.  Flags as computed at
 * inter_part.h:103-104 and vp8enc.cpp:364-366.  Asynchronous; results via vp8hip_download_results. */
int vp8hip_inter_transform(vp8hip_ctx *ctx, int prev_is_golden, int prev_is_altref, int use_golden,
                           int use_altref);

/* the clEnqueueReadBuffer group at inter_part.h:263-265 and vp8enc.cpp:422-433, followed by the
 * clFinish at vp8enc.cpp:439-440: returns when the copies are complete */
int vp8hip_download_results(vp8hip_ctx *ctx, const vp8hip_results *r);

/* Host-side changes made between the transform and the loop filter (per-MB intra fallback,
 * intra_part.h:1063-1084; key frames): any pointer may be NULL = keep the device copy.
 * vp8enc.cpp:460-470, loop_filter.h:7, 63-65 */
int vp8hip_upload_mb_data(vp8hip_ctx *ctx, const int16_t *MB_coeffs, const int32_t *MB_parts,
                          const int32_t *MB_segment_id);
int vp8hip_upload_recon(vp8hip_ctx *ctx, const uint8_t *y, const uint8_t *u, const uint8_t *v);

/* ---- the host intra path on the device -----------------------------------------------------------
 * The reference codes key frames and the intra fallback of inter frames on one CPU thread, which forces the
 * reconstruction and all coefficients across the bus twice per frame (vp8enc.cpp:422-433, 460-470;
 * loop_filter.h:7,63-65).  Both run here as a row wavefront on the current frame already in HBM.
 *
 * intra_transform()'s loop, intra_part.h:1089-1109 (predict_and_transform_mb, :517-741): the current frame as a key
 * frame -- B_PRED luma with the mode of every 4x4 block picked by pick_luma_predictor (:252-515), TM_PRED chroma,
 * quantizers of segment 0 of the segment data in force (vp8hip_set_segments / vp8hip_auto_segments with
 * is_key_frame).  Leaves coefficients (blocks 0..23, zigzag), MB_parts = are4x4, MB_segment_id = 0 and the
 * unfiltered reconstruction where vp8hip_inter_transform leaves them; continue with vp8hip_prepare_filter_mask and
 * vp8hip_loop_filter, after which the next vp8hip_inter_transform takes prev_is_golden = prev_is_altref = 1
 * (intra_part.h:1091-1098).  Asynchronous. */
int vp8hip_intra_transform(vp8hip_ctx *ctx);
/* check_SSIM(), vp8enc.cpp:231-263, on the results of the preceding vp8hip_inter_transform: every macroblock whose
 * SSIM is below the context's ssim_target is tried as intra in segments AQ, HQ, UQ (test_inter_on_intra,
 * intra_part.h:855-1087) and replaced -- coefficients, MB_parts, MB_segment_id, MB_SSIM, reconstruction -- when that
 * scores higher.  Returns frames.replaced, frames.new_SSIM and the minimum SSIM (`min1`; the reference updates
 * the filter parameters when it exceeds 0.95, :260).  Blocks until the three values are back. */
int vp8hip_check_ssim(vp8hip_ctx *ctx, int32_t *replaced, float *new_ssim, float *min_ssim);
/* The same without the host in the middle -- what a frame loop at thousands of frames per second needs.  The call enqueues
 * check_SSIM's fallback (a launch whose workgroups leave at once unless the transform has flagged a macroblock below the
 * target -- not made at all when the context's target is -1 or lower, which no macroblock's SSIM can lie below; it also renews
 * the filter mask and non-zero count of what it replaces: no vp8hip_prepare_filter_mask needed) and
 * arms the NEXT vp8hip_loop_filter / vp8hip_batch_loop_filter call, whose launch then carries the rest: every band of the
 * filter takes the frame's minimum SSIM and, above 0.95, filters with the segment data `prepare_segments_data(1, 7)` gives
 * (:260-261; from the strength pair vp8hip_auto_segments left on the device, so the frame's segment data must come from that
 * call; refqi / qi_min as for it), and one extra workgroup writes those segment data back for the entropy stage, sums the
 * frame's SSIM in the reference's order and hands replaced / new_SSIM / min SSIM to the host through memory the host polls.
 * vp8hip_check_ssim_result collects them when the host next needs them -- the reference's "redo as key frame" decision
 * (:443-453), which the native frame loop takes at the start of the NEXT call (include/vp8hip_driver.h) -- and returns as soon
 * as the verdict workgroup has run, a few microseconds into the filter's launch.  filter_updated: 1 when the segment data were
 * rewritten (video.loop_filter_sharpness is then 7).  VP8HIP_ERR_STATE while no armed loop filter has been launched.
 * vp8hip_check_ssim_ready: 1 if _result would not wait. */
int vp8hip_check_ssim_async(vp8hip_ctx *ctx, const int32_t refqi[4], int qi_min);
int vp8hip_check_ssim_result(vp8hip_ctx *ctx, int32_t *replaced, float *new_ssim, float *min_ssim, int32_t *filter_updated);
int vp8hip_check_ssim_ready(const vp8hip_ctx *ctx);
/* e_data[].mode[16] of the last vp8hip_intra_transform / vp8hip_check_ssim (the sub-block modes the header coder
 * writes; after check_ssim: of the LAST attempt on a macroblock, as in the reference -- see
 * vp8hip_conformant_stream -- and 0 where none was made) and
 * e_data[].is_inter_mb (check_ssim only).  Either pointer may be NULL.  After vp8hip_check_ssim_async both are defined only
 * if the verdict says that macroblocks were replaced (with none below the target the fallback does not touch them, and the
 * header coder is not to read them: use_intra_info = 0 gives the reference's bits). */
int vp8hip_download_intra(vp8hip_ctx *ctx, int32_t *modes, int32_t *is_inter_mb);
/* NOT the reference's behaviour, off by default: with on = 1 the emitted stream decodes, in any VP8 decoder, to exactly the
 * reconstruction the encoder keeps as its references.  The reference's does not, in two places (found by decoding the frames
 * with a decoder written from RFC 6386, tests/vp8_decode.py, tests/test_decode_roundtrip.py):
 *  1. `construct` (GPU_kernels.cl:702-758) wraps the last three of the nine first-pass lines of a 4x4 predictor to 8 bits
 *     where the format saturates them (RFC 6386 section 18.3): wherever the six-tap filter overshoots on one of those lines --
 *     hard edges, text, graphics -- the encoder predicts from other samples than every decoder will;
 *  2. check_SSIM writes e_data.mode on every attempt (intra_part.h:964) but coefficients and reconstruction only when the
 *     attempt is kept (:1058-1086): a macroblock whose AQ attempt was kept and whose HQ / UQ attempts then failed goes out with
 *     sub-block modes that do not belong to its coefficients.
 * Either error lives on through inter prediction until the next key frame.  on = 1 saturates all nine lines and keeps the
 * modes of the attempt that was KEPT; the output is then no longer the reference's byte for byte wherever one of the two
 * cases occurs (and identical where none does).  All members of a batch must agree. */
int vp8hip_conformant_stream(vp8hip_ctx *ctx, int on);
/* copy_with_padding(), encIO.h:141-196, on the device: after this call the planes handed to vp8hip_upload_current /
 * vp8hip_set_current_device / vp8hip_batch_set_current_device are tight planes of src_width x src_height (both even, less than
 * 16 below the coded size the context was created with -- 1920x1080 for a 1920x1088 context), and the step that brings them
 * into the context's surfaces repeats the last row downwards and every row's last sample to the right.  0, 0 = back to planes
 * of the coded size.  (Reconstruction planes -- vp8hip_upload_last, vp8hip_set_last_device, downloads -- always have the coded
 * size.)  Identical to the reference whenever the width needs no padding, which covers every BASELINE config; for other
 * widths the reference never writes V's right padding (:180-183 read and write U instead) and this is what it means.
 * All members of a batch must have the same source size (vp8hip_batch_create and the batched launch check it). */
int vp8hip_set_source_size(vp8hip_ctx *ctx, int src_width, int src_height);

/* prepare_filter_mask_and_non_zero_coeffs(), loop_filter.h:25-55.  nz_out: [MBs] or NULL.
 * (vp8hip_inter_transform already produced mask and counts for its own coefficients; this call
 * recomputes them from the device copy, e.g. after vp8hip_upload_mb_data.) */
int vp8hip_prepare_filter_mask(vp8hip_ctx *ctx, int32_t *nz_out);

/* do_loop_filter(), loop_filter.h:185-190: normal loop filter on the reconstruction, in place,
 * after which that reconstruction IS the LAST reference of the next vp8hip_inter_transform. */
int vp8hip_loop_filter(vp8hip_ctx *ctx);
/* on = 1: the context gets a second stream, and work that does not depend on the filtered frame runs beside the loop filter:
 * the entropy stage of the same frame (vp8hip_count_probs ... vp8hip_encode_frame), the next frame's upload, parameter scan
 * and GOLDEN / ALTREF searches.  The filter stays on the stream the frame was coded on and the CONTEXT moves to the other one
 * until a call needs the filtered frame (it then moves back, behind the filter): a video's frame-to-frame dependency chain
 * is launches of one stream.  vp8hip_stream() names the stream the next call will use.  A single video coded frame after
 * frame: 0.60 -> 0.43 ms per 1080p frame.  Off by default: a host that runs many contexts side by side (GOP chunks) already
 * keeps the device busy and advances them in batches (vp8hip_batch_create), which excludes this mode. */
int vp8hip_filter_overlap(vp8hip_ctx *ctx, int on);

/* ---- coefficient entropy stage: the consumer of the coefficient buffer -------------------------
 * The reference runs it on a CPU OpenCL device, one work-item per partition (vp8enc.cpp:48-94).  Here the
 * statistics are a device histogram over all blocks.  Input = the context's current coefficients, MB_parts
 * and non-zero counts (as left by vp8hip_inter_transform, or vp8hip_upload_mb_data + vp8hip_prepare_filter_mask).
 *
 * count_probs + num_div_denom + the two read-backs of vp8enc.cpp:58-69.
 *   new_probs[4][8][3][11]       probability (1..255) of a zero branch per context, summed over the partitions;
 *   new_probs_denom[4][8][3][11] partition 0's denominators (1 + branches seen), which is what the reference's
 *                                host code inspects to fall back to the default probabilities (:70-76).
 * num_partitions: 1, 2, 4 or 8 (partition p owns macroblock rows p, p+n, ...). */
#define VP8HIP_NUM_COEFF_PROBS 1056
int vp8hip_count_probs(vp8hip_ctx *ctx, int num_partitions, uint32_t *new_probs, uint32_t *new_probs_denom);

/* The write-back of the final probabilities, encode_coefficients and the read of the partitions (vp8enc.cpp:77-81,
 * gather at entropy_host.cpp): codes the macroblock rows of every partition with the boolean coder, on the
 * device, and returns partition p at partitions + p * partition_step with its length in partition_sizes[p].
 * coeff_probs[4][8][3][11] (low byte used) = new_probs after the host's default-probability fallback.
 * Must follow vp8hip_count_probs for the same coefficients and num_partitions (it reuses the block contexts).
 * VP8HIP_ERR_OVERFLOW if a partition does not fit partition_step (nothing is written then).  The device scratch
 * starts at 64 bools per 4x4 block on average and is doubled -- the frame is then coded again -- up to the 304 a
 * block can produce at most, so no frame is refused for the device's sake. */
int vp8hip_encode_coefficients(vp8hip_ctx *ctx, const uint32_t *coeff_probs, int num_partitions, int partition_step,
                               uint8_t *partitions, int32_t *partition_sizes);

/* encode_header(), entropy_host.cpp:709-1256 -- the first partition (frame header, then segment id / skip flag /
 * reference frame / motion-vector mode and vectors or intra modes of every macroblock) -- coded on the device from
 * the results that are already there, instead of on one host thread after downloading them (2-3 ms per 1080p frame
 * there).  Uses MB_segment_id, the non-zero counts, MB_reference_frame, MB_parts, MB_vectors as the transform and
 * vp8hip_prepare_filter_mask left them, the segment data in force, the coefficient statistics of the preceding
 * vp8hip_count_probs (the header transmits the probability of every context that occurred) and, for key frames or
 * with use_intra_info, e_data.mode / is_inter_mb of vp8hip_intra_transform / vp8hip_check_ssim.  skip_prob,
 * prob_intra, prob_last/prob_gf, the segment-map and motion-vector probabilities are derived on the device as
 * encode_header derives them.  Writes the partition with its 3- or 10-byte uncompressed chunk to `out`; *size = its
 * size (frames.encoded_frame_size after encode_header).  Blocks.  The host implementation of the same function is
 * vp8bs_encode_header (include/vp8hip_bitstream.h); vp8bs_gather_frame appends the coefficient partitions. */
#define VP8HIP_SHARPNESS_ON_DEVICE INT32_MIN
typedef struct {
    int32_t is_key, is_golden, is_altref;    /* frames.current_is_{key,golden,altref}_frame */
    int32_t loop_filter_type;                /* video.loop_filter_type (0) */
    int32_t loop_filter_sharpness;           /* video.loop_filter_sharpness; VP8HIP_SHARPNESS_ON_DEVICE = the value vp8hip_auto_segments
                                                computed.  (Not -1: get_loopfilter_strength's `int` accumulator overflows on large noisy
                                                frames, vp8enc.cpp:112-126, and the sharpness it then leaves is NEGATIVE -- -1 on a 1080p
                                                frame of noise; the reference writes its low three bits into the header, and so must this.) */
    int32_t partitions_log2;                 /* video.number_of_partitions_ind */
    int32_t width, height;                   /* video.dst_width/height for key frames; 0 = the coded size */
    int32_t use_intra_info;                  /* inter frames: 1 = vp8hip_check_ssim ran on this frame */
} vp8hip_header_params;
int vp8hip_encode_header(vp8hip_ctx *ctx, const vp8hip_header_params *params, uint8_t *out, size_t capacity, size_t *size);

/* entropy_encode() + gather_frame() (vp8enc.cpp:48-94, encIO.h:1-30) in one call and entirely on the device:
 * count_probs, num_div_denom, the default-probability fallback, encode_coefficients, encode_header, then the frame
 * is assembled in `out` -- first partition, the sizes of all coefficient partitions but the last, the partitions.
 * Nine kernel launches, the last of which writes the finished frame into pinned host memory itself (no copy command),
 * instead of the 21 launches and eight blocking transfers of the step-by-step calls; any frame
 * size (the step-by-step vp8hip_encode_coefficients stops at 2^20 4x4 blocks).  params->partitions_log2 is ignored (derived from num_partitions).  *size = bytes of the finished frame:
 * what the reference hands to write_output_file(). */
int vp8hip_encode_frame(vp8hip_ctx *ctx, int num_partitions, const vp8hip_header_params *params, uint8_t *out, size_t capacity,
                        size_t *size);

/* init_all() allocates everything before the first frame (init.h:430-593); the entropy stage's scratch (about 100 MB at 1080p)
 * and the pinned frame buffer are made when the first frame is asked for, or here -- a host that wants no allocation inside its
 * frame loop calls this once after vp8hip_create. */
int vp8hip_reserve_frame_path(vp8hip_ctx *ctx);
/* ... sized for the densest frame there can be (304 bools per 4x4 block instead of 64; about 270 MB at 1080p): no frame ever has to
 * be coded a second time because the scratch was too small.  For a caller that starts the next frame before it takes the bytes of
 * this one (vp8hip_encode_frame_begin below). */
int vp8hip_reserve_frame_path_dense(vp8hip_ctx *ctx);

/* The same in two halves, for a host thread that drives several contexts (GOP chunks) or wants the next frame under way before
 * it takes this one's bytes: _begin enqueues the whole entropy stage and the read-back and returns at once; _end waits for the
 * stage (not for whatever was enqueued behind it) and fills `out`.  VP8HIP_ERR_STATE from _begin while a frame is pending,
 * from _end when none is.
 * With vp8hip_filter_overlap the stage runs on a stream of its own beside the frame's loop filter, and the NEXT frame may be
 * started between _begin and _end (vp8hip_set_current_device ... vp8hip_loop_filter: its side work runs beside filter and stage,
 * its macroblock kernel waits for the stage): one video then costs a frame what it costs without frames out.  Only the recode
 * after a scratch overflow is lost that way (its input is overwritten): _end then returns VP8HIP_ERR_STATE -- call
 * vp8hip_reserve_frame_path_dense once and it cannot happen.  Without the overlap mode no other call may be made on the context
 * between the two. */
int vp8hip_encode_frame_begin(vp8hip_ctx *ctx, int num_partitions, const vp8hip_header_params *params);
int vp8hip_encode_frame_end(vp8hip_ctx *ctx, uint8_t *out, size_t capacity, size_t *size);

/* filtered planes = the current LAST (debug.h:8-36 dump; host intra fallback input) */
int vp8hip_download_last(vp8hip_ctx *ctx, uint8_t *y, uint8_t *u, uint8_t *v);

int vp8hip_synchronize(vp8hip_ctx *ctx);
/* hipStream_t the context launches on (for event timing by the caller) */
void *vp8hip_stream(vp8hip_ctx *ctx);
int vp8hip_last_hip_error(const vp8hip_ctx *ctx);
/* Hardware queues the HIP runtime of this process multiplexes its streams onto.  The runtime reads GPU_MAX_HW_QUEUES once, at the
 * process's first HIP call (default 4), and streams that share a queue serialise -- so the LIBRARY sets it when it is loaded (a
 * constructor: setenv("GPU_MAX_HW_QUEUES", "16", 0), i.e. unless the host exported a value of its own): a host that links or dlopens
 * libvp8hip.so before it touches the GPU runs on 16 queues without knowing any of this (the reference creates the queues it needs
 * itself, init.h:1162-1165).  16: measured optimum; beyond 24 queues per process the hardware scheduler rotates them and
 * context-switches running waves.  Returns the value in force as far as the library can tell: what the environment said when the
 * runtime initialised.  If the runtime was ALREADY initialised when the library was loaded (a host that made HIP calls first) the
 * constructor changes nothing, says so on stderr once (VP8HIP_QUIET=1 silences it) and this returns what the environment held then
 * (4 if nothing).  vp8hip_inter_transform prints one line when more contexts launch on streams of their own than there are queues.
 * SIDE EFFECTS of that constructor, stated: (1) setenv() in the host process -- safe when the library is loaded at program start or
 * before the host has made threads; a host that dlopen()s it while other threads may call getenv() exports GPU_MAX_HW_QUEUES itself or
 * sets VP8HIP_NO_ENV=1, with which the library touches nothing; (2) the variable is inherited by the host's child processes. */
int vp8hip_hw_queues(void);

const char *vp8hip_status_string(int status);
/* The ABI of this header as MAJOR * 1000 + MINOR: MAJOR changes when an existing entry point or struct changes its meaning or
 * layout (vp8drv_config grew in round 2: 2; vp8drv_stats grew by refs_searched and vp8drv_default_config turned check_ssim on
 * -- return values and counters provisional until vp8drv_resolve -- in round 3: 3), MINOR when entry points are added.  A host
 * built against an older header checks it once after loading the library.  3001: the shard, device-memory and frame-check entry points;
 * 3002: vp8drv_encode_video_device; 3003: vp8hip_import_last, vp8hip_group_*, the load-time hardware-queue setting;
 * vp8drv_frame_check folds position in (4: its values change). */
#define VP8HIP_ABI_VERSION 4008
int vp8hip_abi_version(void);
/* 1 if this build of the library honours the timing-experiment switches that leave work out of a launch or a wait
 * (VP8HIP_EXPERIMENT_SKIP, VP8HIP_EXPERIMENT_SKIP_ENT, VP8HIP_EXPERIMENT_NOWAIT, VP8DRV_EXPERIMENT_READY_FIRST; built with
 * -DVP8HIP_EXPERIMENTS, results are then garbage on purpose); 0 for the shipped build, in which they are constants and no
 * environment can take work out of a run.  bench.py refuses to print a line from a build that answers 1. */
int vp8hip_experiments_compiled_in(void);
/* the head-of-frame stream mode of batches in force in this process: 0 none (default), 1 per batch, 2 one for all (VP8HIP_BATCH_PREP) */
int vp8hip_batch_prep_mode(void);

/* ---- device memory for a caller that has none of its own ---------------------------------------------------------------
 * vp8hip_set_current_device / vp8hip_set_last_device / the batched forms take planes that are already in this device's memory.
 * A host that is not a GPU program itself (the reference's main(), bench.py, the tests) gets such memory here, so that the process
 * needs no second GPU runtime beside the one this library was built for (PyTorch ships its own copy of the HIP runtime; both in
 * one process was where a one-in-twenty teardown crash of round 3 lived).  Plain hipMalloc / hipMemcpy / hipDeviceSynchronize
 * on device_ordinal; the copies block. */
int vp8hip_device_count(void);
int vp8hip_device_alloc(int device_ordinal, size_t bytes, void **out);
int vp8hip_device_free(int device_ordinal, void *p);
int vp8hip_device_upload(int device_ordinal, void *dst, const void *src, size_t bytes);
int vp8hip_device_download(int device_ordinal, void *dst, const void *src, size_t bytes);
int vp8hip_device_synchronize(int device_ordinal);
/* page-locked host memory (hipHostMalloc): source planes handed to vp8hip_prefetch_current / vp8hip_batch_upload_current / _prefetch_current from it
 * are copied asynchronously (vp8hip_upload_current itself returns when its copy is done, whatever the memory) */
int vp8hip_host_alloc(int device_ordinal, size_t bytes, void **out);
int vp8hip_host_free(int device_ordinal, void *p);
int vp8hip_device_mem_info(int device_ordinal, size_t *free_bytes, size_t *total_bytes);
/* "dddd:bb:dd.f" of the device (for pinning the host threads to its NUMA node); len >= 16 */
int vp8hip_device_pci_bus_id(int device_ordinal, char *out, int len);
/* version of the HIP runtime this process's library calls land in (hipRuntimeGetVersion), e.g. 70226015 */
int vp8hip_runtime_version(void);


#ifdef __cplusplus
}
#endif

/* the rest of the ABI, in headers of their own (they include this one): several contexts at a time, and the measurement taps */
#include "vp8hip_multi.h"
#include "vp8hip_taps.h"

#endif
