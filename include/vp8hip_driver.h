/*
 * vp8hip_driver.h -- the reference's frame loop (main(), src/vp8enc.cpp:351-488) as native host code over
 * include/vp8hip.h and include/vp8hip_host.h.
 *
 * The reference host is C++ and this is its C++ counterpart (vp8oclenc_amd/csrc/vp8_driver.cpp): frame-type
 * state machine, per-frame loop-filter strength and segment data, key frames (intra_transform), inter_transform,
 * check_SSIM with its intra fallback, its filter-parameter update and its "redo as key frame" decision, filter
 * mask, loop filter -- in the reference's order, every stage on the device.  scene_change()'s decision is the caller's
 * (force_key; vp8hip_chroma_change + vp8host_scene_change produce it) or, with cfg.scene_detect, made here the same way.  Not here: the header/MV entropy
 * coder and the container.
 * vp8oclenc_amd/driver.py is the same loop in Python for the parity tests (it also runs the CPU oracle).
 */
#ifndef VP8HIP_DRIVER_H
#define VP8HIP_DRIVER_H

#include <stddef.h>
#include <stdint.h>

#include "vp8hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vp8drv vp8drv;

typedef struct {
    int32_t gop_size;        /* -g, frames between key frames (init.h:1575) */
    int32_t altref_range;    /* frames between altref frames (vp8enc.cpp:340-344) */
    int32_t qi_min, qi_max;  /* -qmin / -qmax -> the two quantizer ladders (init.h:1585-1603) */
    float ssim_target;       /* -SSIM-target (init.h:1512,1576) */
    int32_t device_params;   /* 1: loop-filter strength + segment data on the device, no host round trip
                                (vp8hip_auto_segments); 0: host mirror on the caller's luma plane */
    int32_t check_ssim;      /* 1: check_SSIM after every inter_transform (vp8enc.cpp:231-263, 442-453) -- what the reference does:
                                intra fallback, filter update when even the worst macroblock is above 0.95, redo as key frame.
                                With device_params the call does not wait for it (vp8hip_check_ssim_async): fallback, statistics
                                and filter update run on the device in front of the loop filter, and the one decision that is the
                                host's -- redo as key frame -- is taken when the verdict is next needed: at the start of the next
                                vp8drv_encode_frame_* / vp8drv_batch_encode_frame_device, in vp8drv_get_frame[_begin], or in
                                vp8drv_resolve.  Until then the return value "inter frame" and the counters of vp8drv_get_stats
                                are provisional.  (A bounded device-side wait of the SAME frame's loop filter that expires is
                                reported one step later still: the verdict workgroup samples the error word when the filter's
                                launch starts, so a time-out inside that launch shows in the next verdict or in
                                vp8hip_synchronize / vp8drv_get_frame, whichever comes first.)  With device_params == 0 the
                                host sits in the middle, as in the reference.
                                0: skip it (the reference's loop minus the filter update; kept for A/B runs) */
    int32_t num_partitions;  /* -partitions: 1, 2, 4 or 8 coefficient partitions (init.h:1451-1469, default 1) */
    int32_t display_width, display_height;   /* video.dst_width/height written into key frames; 0 = the coded size */
    int32_t host_bitstream;  /* vp8drv_get_frame: 0 = the whole entropy stage on the device (vp8hip_encode_frame);
                                1 = first partition on the host (vp8bs_encode_header) after downloading what it reads,
                                the way the reference does it -- kept as the cross-check */
    int32_t overlap_filter;  /* 1: loop filter on its own stream, side by side with the frame's entropy stage
                                (vp8hip_filter_overlap); for a single video coded frame after frame.  Default 0 */
    int32_t ref_mask;        /* which of the two optional references inter frames may search besides LAST: bit 0 GOLDEN,
                                bit 1 ALTREF, ANDed onto the reference's own rule (inter_part.h:103-104).  Default 3 (= the
                                reference); 0 = LAST only (BASELINE configs[1]) */
    int32_t conformant_stream;   /* 1 = vp8hip_conformant_stream: the stream decodes to the encoder's own reconstruction (the
                                format's predictor, the kept attempt's modes); 0 = the reference's stream, byte for byte.
                                Default 0 */
    int32_t scene_detect;        /* 1: scene_change() (vp8enc.cpp:265-311, 408-416) inside the loop: the chroma differences to the
                                previous input frame on the device (vp8hip_chroma_change; blocks for two words per frame),
                                the decision with its hold-over on the host (vp8host_scene_change); a detected cut is coded
                                as a key frame.  0 (default): the caller's force_key alone decides */
    int32_t src_width, src_height;   /* size of the frames handed to vp8drv_encode_frame_* when it is below the coded size
                                (video.src_* against video.wrk_*, init.h:375-392): copy_with_padding (encIO.h:141-196) then
                                runs on the device (vp8hip_set_source_size) and key frames carry this size as the display
                                size unless display_width/height say otherwise.  0 = frames of the coded size.  Needs
                                device_params = 1 */
} vp8drv_config;

void vp8drv_default_config(vp8drv_config *cfg);   /* the reference's defaults: 150, 5, 0, 48, -1, 1, 1, 1, 0, 0, 0, 0, 3, 0, 0, 0, 0 */

int vp8drv_create(vp8drv **out, int width, int height, int device_ordinal, const vp8drv_config *cfg);
void vp8drv_destroy(vp8drv *d);
vp8hip_ctx *vp8drv_context(vp8drv *d);   /* for downloads, the entropy stage, profiling */

/* One iteration of the while-loop body, vp8enc.cpp:351-488, for a frame whose planes are already in this
 * device's memory (tight stride).  Returns 1 = coded as a key frame, 0 = inter frame, < 0 = vp8hip_status.
 * force_key: the caller's scene_change() verdict (vp8enc.cpp:408-416).  Asynchronous unless check_ssim is on. */
int vp8drv_encode_frame_device(vp8drv *d, const void *d_y, const void *d_u, const void *d_v, int force_key);
/* same for host planes (blocks for the upload).  With device_params == 0 the host mirror scans y. */
int vp8drv_encode_frame_host(vp8drv *d, const uint8_t *y, const uint8_t *u, const uint8_t *v, int force_key);
/* The frame AFTER the one just handed in, started on its way to the device (vp8hip_prefetch_current): a reader that is one frame ahead
 * calls this right after vp8drv_encode_frame_host and hands the same pointers to the next vp8drv_encode_frame_host. */
int vp8drv_prefetch_frame_host(vp8drv *d, const uint8_t *y, const uint8_t *u, const uint8_t *v);
/* ... and handed over altogether -- made the context's current frame (a pack from the prefetch's staging buffer), with cfg.scene_detect its
 * chroma scan started (vp8hip_chroma_change_async) -- while the frame just coded is still in its loop filter: a reader that is a frame ahead
 * calls this after vp8drv_get_frame_begin (or vp8drv_resolve) of the frame just coded and hands the same pointers to the next
 * vp8drv_encode_frame_host, which then uploads nothing and waits for no scan: the new frame's side work is enqueued a hundred microseconds
 * earlier, early enough to run beside the previous frame's loop filter (scripts/native/y4m_to_ivf.cpp).  Takes the open verdict first. */
int vp8drv_stage_frame_host(vp8drv *d, const uint8_t *y, const uint8_t *u, const uint8_t *v);

/* With check_ssim: waits for the verdict on the frame just coded and, if it sends the frame back (vp8enc.cpp:443-453), codes it
 * again as a key frame.  Returns 1 if the last frame ended as a key frame, 0 if as an inter frame, < 0 = vp8hip_status.  Implied by
 * the next vp8drv_encode_frame_* and by vp8drv_get_frame[_begin]; a no-op when nothing is open. */
int vp8drv_resolve(vp8drv *d);
/* 1 if the next call on this driver (this batch) would not wait for a verdict: a host that advances several chunks from one
 * thread takes the ones that are ready first instead of waiting for them in a fixed order */
int vp8drv_ready(const vp8drv *d);

/* The frame just coded, as bytes: entropy_encode() + gather_frame() of the reference (vp8enc.cpp:48-94, 476-481;
 * encIO.h:1-30) -- coefficient statistics, coefficient partitions and the first partition (frame header, macroblock
 * modes, motion vectors) coded on the device and assembled into `out` (vp8hip_encode_frame; with host_bitstream the
 * first partition comes from the host coder of include/vp8hip_bitstream.h instead).  Call it after vp8drv_encode_frame_* and before the next one.
 * Blocks.  *size = bytes written; VP8HIP_ERR_OVERFLOW if `capacity` is too small.  With vp8bs_ivf_file_header /
 * vp8bs_ivf_frame_header around the frames this is the reference's .ivf output, byte for byte. */
int vp8drv_get_frame(vp8drv *d, uint8_t *out, size_t capacity, size_t *size);
/* The same in two halves (device entropy stage only: VP8HIP_ERR_STATE with host_bitstream): _begin enqueues the
 * stage and returns, _end waits and fills `out`.  One host thread can thus keep many GOP chunks in flight:
 * encode_frame + get_frame_begin on every chunk, then get_frame_end on every chunk.  With cfg.overlap_filter (one video frame
 * after frame) the NEXT frame may be started between the two -- encode(t), get_frame_begin(t), encode(t + 1), get_frame_end(t): the
 * stage of frame t runs beside its loop filter and beside frame t + 1's input side (vp8hip_encode_frame_begin, include/vp8hip.h;
 * call vp8hip_reserve_frame_path_dense(vp8drv_context(d)) once); without it no other call on this driver between the two. */
int vp8drv_get_frame_begin(vp8drv *d);
int vp8drv_get_frame_end(vp8drv *d, uint8_t *out, size_t capacity, size_t *size);

/* Several GOP chunks advanced one frame at a time, every stage ONE launch for all of them (vp8hip_batch_* in vp8hip.h): what a
 * host with many chunks in flight uses instead of one stream per chunk -- the part runs four to five kernels at once, so
 * sixteen narrow launches queue where four wide ones fill it.  Members: up to VP8HIP_MAX_BATCH drivers of one geometry
 * with device_params = 1, overlap_filter = 0, scene_detect = 0 and the same qi_min / qi_max / num_partitions / check_ssim.  One call = vp8drv_encode_frame_device on every member
 * (force_key / was_key indexed by member, either may be NULL); a member whose frame is a key frame takes its ordinary key-frame
 * path on the shared stream.  vp8drv_get_frame[_begin/_end] per member afterwards, as usual. */
typedef struct vp8drv_batch vp8drv_batch;
int vp8drv_batch_create(vp8drv_batch **out, vp8drv *const *drv, int n);
void vp8drv_batch_destroy(vp8drv_batch *b);      /* the drivers stay */
int vp8drv_batch_encode_frame_device(vp8drv_batch *b, const int *members /* NULL = all; 0 = this member sits the call out */,
                                     const void *const *y, const void *const *u, const void *const *v, const int *force_key, int *was_key);
/* ... with the members' frames in host memory (vp8hip_batch_upload_current: tight planes of the source size, page-locked for the copies
 * to be asynchronous, unchanged until the next call on this batch has returned) */
int vp8drv_batch_encode_frame_host(vp8drv_batch *b, const int *members, const void *const *y, const void *const *u, const void *const *v,
                                   const int *force_key, int *was_key);
/* the members' NEXT frames started on their way (vp8hip_batch_prefetch_current; y[i] NULL: nothing for member i): hand the same pointers
 * to the next vp8drv_batch_encode_frame_host */
int vp8drv_batch_prefetch_frame_host(vp8drv_batch *b, const uint8_t *const *y, const uint8_t *const *u, const uint8_t *const *v);
/* vp8drv_get_frame_begin for the members' frames in one set of launches (src/vp8enc.cpp:48-94 for up to four chunks at
 * once); then vp8drv_get_frame_end on every member */
int vp8drv_batch_get_frame_begin(vp8drv_batch *b, const int *members);
int vp8drv_batch_ready(const vp8drv_batch *b);
/* One frame on EVERY batch of `batches` -- vp8drv_batch_encode_frame_device(batches[k], NULL, y[k], u[k], v[k], NULL, was_key[k])
 * (was_key may be NULL, and so may any was_key[k]) -- in the order of the array, each batch waiting for its own members' check_SSIM
 * verdicts: the loop a host with many chunks in flight would otherwise write itself, in one call.  (The fixed order is on
 * purpose: serving whichever batch is ready first lets the batches bunch up and was 3-6 % slower.) */
int vp8drv_batches_encode_frame_device(vp8drv_batch *const *batches, int nbatches, const void *const *const *y, const void *const *const *u,
                                       const void *const *const *v, int *const *was_key);
/* `nframes` frames on every batch, ONE HOST THREAD PER BATCH (started and joined here): the batches share nothing, and with
 * check_SSIM in the loop each one's next frame waits for the verdict on its previous one -- a thread of its own keeps one
 * batch's wait from holding up the others' streams (same box, 48 chunks in 8 batches: one thread serving the batches in turn
 * 55-59 M MB/s, a thread per batch 60.3-60.4, check_SSIM off 60.7-60.9).  Frame t of member i of batch k is
 * frames[(start[k][i] + t) % nd]: `frames` = nd device-resident frames as {y, u, v} pointer triples, shared by all chunks (a
 * transcoder's ring of decoded frames; the bench's synthetic sequence).  The threads start 200 us apart: batches that start together
 * from an idle device stay in lockstep, all their loop filters running at once with nothing wide beside them (5-8 % slower).
 * keys_out[k][i] (may be NULL) counts member i's key frames.  bytes_out (may be NULL): with it every frame is also delivered as
 * bytes -- vp8drv_batch_get_frame_begin for the batch, vp8drv_get_frame_end per member into a buffer of the thread -- and
 * bytes_out[k][i] receives the sum of member i's frame sizes: the loop of a transcoder that writes the frames away.
 * check_out (may be NULL; needs bytes_out): check_out[k][i] is folded with a checksum of every frame of member i as it is
 * delivered, in order -- vp8drv_frame_check(previous value, frame, size); start it at 0 -- so that a caller can hold the bytes of
 * a whole run against a second coding of the same frames without keeping them (bench.py's self-check).
 * Every thread ends with its members' last verdicts taken (vp8drv_resolve) and their stream synchronised: a bounded device-side wait that
 * expired anywhere in the run -- the last frame's loop filter included -- is THIS call's VP8HIP_ERR_TIMEOUT.
 * Returns the first error of any batch, or VP8HIP_OK. */
int vp8drv_batches_encode_frames_device(vp8drv_batch *const *batches, int nbatches, int nframes, const void *const (*frames)[3], int nd,
                                        const int *const *start, int *const *keys_out, uint64_t *const *bytes_out, uint64_t *const *check_out);
/* The same with the nd frames in HOST memory ({y, u, v} tight planes of the source size; page-locked -- vp8hip_host_alloc -- for the copies
 * to overlap the device's work): every frame crosses the host-device link on its way in (vp8hip_batch_upload_current), and with bytes_out
 * on its way out as well -- the whole-job rate WITH the link in it, the reference's own hand-over (vp8enc.cpp:386-388, 476-481). */
int vp8drv_batches_encode_frames_host(vp8drv_batch *const *batches, int nbatches, int nframes, const void *const (*frames)[3], int nd,
                                      const int *const *start, int *const *keys_out, uint64_t *const *bytes_out, uint64_t *const *check_out);
/* ONE video, `nframes` frames, frame after frame with the frames out -- the loop of scripts/native/y4m_to_ivf.cpp for frames that are
 * already in device memory: encode(t), take frame t - 1's bytes, enqueue frame t's entropy stage, take frame t's verdict; frame t
 * is frames[(start + t) % nd].  With overlap_filter the stage of a frame runs on a stream of its own beside its loop filter and the
 * next frame's side work (vp8hip_encode_frame_begin).  Because frame t + 1 is under way before frame t's bytes are taken, a frame denser
 * than the coder's scratch could not be coded again: the call sizes the scratch for the densest frame there can be
 * (vp8hip_reserve_frame_path_dense; a no-op when the caller has done it) -- about 270 MB per context at 1080p and four times that at
 * 4K, allocated behind a stream synchronisation, so a caller that times this call reserves beforehand.  The frames
 * are laid end to end into `out` (capacity bytes), sizes[t] = frame t's size; keys (may be NULL) counts the key frames, frames sent
 * back by check_SSIM included.  out == NULL (sizes is then not used): the same video WITHOUT frames out -- encode after encode, the
 * last verdict taken at the end and the context synchronised (vp8hip_synchronize: a bounded device-side wait that expired inside the last
 * frame's loop filter is reported by THIS call, VP8HIP_ERR_TIMEOUT) -- for callers that code several videos side by side from a thread each and want no interpreter in
 * the loop (bench.py's config3_literal).  A native loop because the host's reaction times are on the path: every microsecond between a
 * frame's verdict and the enqueue of its stage moves the stage further under the next frame's LAST search (one video with frames
 * out, 1080p: 0.383 ms per frame from Python). */
int vp8drv_encode_video_device(vp8drv *d, int nframes, const void *const (*frames)[3], int nd, int start, uint8_t *out, size_t capacity,
                               uint32_t *sizes, int *keys);
/* the fold above: f = h * 0x9E3779B97F4A7C15 + size; for every little-endian 64-bit word w of the frame, in order (the tail
 * zero-padded): f = (rotl(f, 5) ^ w) * 0x100000001B3; h' = f.  Position-dependent: words that trade places change the value. */
uint64_t vp8drv_frame_check(uint64_t h, const uint8_t *frame, size_t size);

/* counters and the flags inter_transform was given for the last inter frame (tests, logs) */
typedef struct {
    int32_t frame_number, inter_frames, key_frames;
    int32_t last_use_golden, last_use_altref, last_prev_is_golden, last_prev_is_altref, last_was_altref;
    int32_t redone_as_key;            /* inter frames recoded as key frames by check_SSIM's verdict (vp8enc.cpp:443-453) */
    int32_t last_replaced;            /* frames.replaced, frames.new_SSIM and min1 of the last check_SSIM */
    float last_new_ssim, last_min_ssim;
    int32_t scene_changes;            /* encStat.scene_changes_by_color: key frames forced by scene_detect (vp8enc.cpp:411) */
    int32_t refs_searched;            /* references searched by all inter frames so far (1 + use_golden + use_altref each) */
} vp8drv_stats;
void vp8drv_get_stats(const vp8drv *d, vp8drv_stats *s);

#ifdef __cplusplus
}
#endif
#endif
