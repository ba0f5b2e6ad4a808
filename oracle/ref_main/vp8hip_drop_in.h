// vp8hip_drop_in.h -- the host side of INTEGRATION.md section 2 as code: what oracle/ref_main/build.sh splices into the
// reference's OWN host program (src/vp8enc.cpp with its headers, compiled from where they lie under /root/reference) so that
// its main() drives libvp8hip.so instead of two OpenCL devices.  TEST INFRASTRUCTURE: it proves the drop-in, nothing in the
// product includes it.  Everything here is written against the reference's globals (`device`, `video`, `frames`,
// src/vp8enc.cpp:5-12) and is included right behind them; the reference lines each function stands for are cited, and the
// line-addressed edits that delete those lines and call these functions are in build.sh.  No text of the reference is in this
// file.
//
// Two builds from the same edits:
//   default                  every stage of the frame loop on the device (INTEGRATION.md "optional" blocks: key frames,
//                            check_SSIM's fallback, the whole entropy stage with the first partition) -- no OpenCL device needed
//   -DVP8HIP_KEEP_HOST_STAGES the reference's own host code keeps what it does on the host today: intra_transform /
//                            predict_and_transform_mb for key frames, check_SSIM with test_inter_on_intra, encode_header and
//                            gather_frame; the library replaces exactly the OpenCL traffic (uploads, inter_transform, read-backs,
//                            filter mask, loop filter, count_probs / encode_coefficients of the CPU device)
//   -DVP8HIP_FAST            (a third build of the same patch; oracle/_ref/vp8oclenc_hip_fast) the reference's control flow -- its frame-type
//                            state machine, scene_change()'s decision, check_SSIM's "redo as key frame", its file formats -- with the
//                            library's ASYNCHRONOUS entry points and device-side scans where the two builds above block or scan on
//                            the host: frames are read one ahead by a reader thread into page-locked buffers (get_yuv420_frame's
//                            fread) and padded on the device (vp8hip_set_source_size instead of copy_with_padding), the next frame
//                            is started on its way early (vp8hip_prefetch_current), get_loopfilter_strength + prepare_segments_data
//                            run on the device (vp8hip_auto_segments), scene_change()'s two sums likewise (vp8hip_chroma_change),
//                            check_SSIM does not wait (vp8hip_check_ssim_async; the verdict is taken where the frame's bytes are
//                            next needed), and a frame's bytes leave ONE ITERATION LATE (vp8hip_encode_frame_begin / _end: frame
//                            t + 1 is under way while frame t's entropy stage and loop filter run).
// All three must write the same .ivf (tests/test_ref_main.py).
#pragma once
#include <stdint.h>
#include <string.h>

#include "vp8hip.h"
#include "vp8hip_host.h"

#include <time.h>
#include <unistd.h>
#ifdef VP8HIP_FAST
#include <condition_variable>
#include <mutex>
#include <thread>
#endif

static vp8hip_ctx *hip_ctx = NULL;
static bool hip_cur_on_device = false;     // (fast build) this iteration's frame has been handed to the device (once: an upload also rotates the previous frame scene_change compares with)
// the frame loop by the host's clock -- from the end of init_all() to the start of finalize(), i.e. main()'s while loop with its reads and
// writes (vp8enc.cpp:351-488) -- printed on stderr by hip_finalize(): what bench.py's drop_in leg reports next to the process's wall time
static struct timespec hip_loop_t0;
static double hip_seconds_in[8];      // time inside the library's calls, by call site (VP8HIP_DROP_IN_TIMELINE=1)
static const char *const hip_site_name[8] = {"upload_current", "set_segments", "inter_transform", "download/check_ssim", "filter_mask", "loop_filter", "entropy_encode", "intra_transform"};
static int hip_timeline = 0;
static inline double hip_now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
struct hip_site_timer {
    int site; double t0;
    hip_site_timer(int s) : site(s), t0(hip_timeline ? hip_now() : 0.0) {}
    ~hip_site_timer() { if (hip_timeline) hip_seconds_in[site] += hip_now() - t0; }
};

#define HIP_CK(call)                                                                                                   \
    do {                                                                                                               \
        const int rc_ = (call);                                                                                        \
        if (rc_ != VP8HIP_OK) {   /* the reference parks cl_int errors in device.state_gpu (inter_part.h:380) and goes on */ \
            device.state_gpu = rc_;                                                                                    \
            printf("%s -> %d (%s)\n", #call, rc_, vp8hip_status_string(rc_));                                          \
            fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, vp8hip_status_string(rc_));                                 \
            fflush(NULL);                                                                                              \
            _exit(3);       /* (no static destructors: the fast build's reader thread may be in the middle of a read) */ \
        }                                                                                                              \
    } while (0)

#ifdef VP8HIP_FAST
// ---------------------------------------------------------------------------------------------------------------------------------
// The reader: get_yuv420_frame()'s fread (encIO.h:204-254) on a thread of its own, one to two frames ahead, into a ring of page-locked
// buffers (vp8hip_host_alloc: the copies to the device are then asynchronous).  The frames keep their SOURCE size: copy_with_padding
// (encIO.h:141-202) runs on the device (vp8hip_set_source_size), with the reference's own quirks (REFERENCE_DEFECTS.md).
static void hip_upload_current();
static void hip_prefetch_next();
enum { HIP_RING = 6 };      // the frame being coded, the one handed over early, the one on its way (prefetch), the one being read, and slack
static uint8_t *hip_ring[HIP_RING];
static int hip_ring_state[HIP_RING];       // what get_yuv420_frame would have returned for the frame in this slot: 1, 0 (end of stream), -1 (broken)
static size_t hip_ring_head = 0, hip_ring_tail = 0, hip_ring_freed = 0;      // filled by the reader / handed to main() / given back by main()
static std::mutex hip_ring_m;
static std::condition_variable hip_ring_cv;
static std::thread hip_reader;
static bool hip_reader_stop = false;
static int hip_cur_slot = -1;
static size_t hip_prefetched_seq = (size_t)-1;      // the frame (by its number in the stream) that has been started on its way
static int hip_early_slot = -1;            // the NEXT frame, already handed to the device (hip_early_next) while this one's loop filter runs
static bool hip_chroma_async = false;      // ... with scene_change()'s scan under way (vp8hip_chroma_change_async)
static bool hip_frame_pending = false;     // a frame is between vp8hip_encode_frame_begin and _end
static bool hip_have_lagged = false;       // frames.encoded_frame holds the PREVIOUS iteration's frame, not yet written

static void hip_reader_main()
{
    const size_t full = (size_t)video.src_frame_size_luma + 2 * (size_t)video.src_frame_size_chroma;
    for (;;) {
        size_t slot;
        {
            std::unique_lock<std::mutex> l(hip_ring_m);
            hip_ring_cv.wait(l, [] { return hip_reader_stop || hip_ring_head - hip_ring_freed < HIP_RING; });
            if (hip_reader_stop) return;
            slot = hip_ring_head % HIP_RING;
        }
        int state = 1;
        if (fread(hip_ring[slot], 1, full, input_file.handle) != full) state = 0;      // encIO.h:216-223: a short read ends the stream
        else {
            char buf[6];                                                                 // :243-248: the NEXT frame's "FRAME\n"
            const size_t m = fread(buf, 1, 6, input_file.handle);
            if (m > 0 && (buf[0] != 'F' || buf[4] != 'E')) state = -1;
        }
        {
            std::lock_guard<std::mutex> l(hip_ring_m);
            hip_ring_state[slot] = state;
            ++hip_ring_head;
        }
        hip_ring_cv.notify_all();
        if (state != 1) return;
    }
}

static cl_uchar *hip_own_current[3];       // init_all()'s allocations behind frames.current_* (init.h:419-421): finalize() frees them
static int hip_fast_init()
{
    hip_own_current[0] = frames.current_Y; hip_own_current[1] = frames.current_U; hip_own_current[2] = frames.current_V;
    HIP_CK(vp8hip_filter_overlap(hip_ctx, 1));      // one video: the loop filter beside the next frame's input side, the entropy stage beside both
    if (video.wrk_width != video.src_width || video.wrk_height != video.src_height)
        HIP_CK(vp8hip_set_source_size(hip_ctx, video.src_width, video.src_height));
    HIP_CK(vp8hip_reserve_frame_path_dense(hip_ctx));   // the next frame is started before this one's bytes are taken: no frame may need a second coding
    const size_t full = (size_t)video.src_frame_size_luma + 2 * (size_t)video.src_frame_size_chroma;
    for (int k = 0; k < HIP_RING; ++k) HIP_CK(vp8hip_host_alloc((int)device.gpu_preferred_platform_number, full, (void **)&hip_ring[k]));
    hip_reader = std::thread(hip_reader_main);
    return 1;
}

static void hip_fast_shutdown()
{
    {
        std::lock_guard<std::mutex> l(hip_ring_m);
        hip_reader_stop = true;
    }
    hip_ring_cv.notify_all();
    if (hip_reader.joinable()) hip_reader.join();
    frames.current_Y = hip_own_current[0]; frames.current_U = hip_own_current[1]; frames.current_V = hip_own_current[2];
    for (int k = 0; k < HIP_RING; ++k) vp8hip_host_free((int)device.gpu_preferred_platform_number, hip_ring[k]);
}

// get_yuv420_frame()'s body, encIO.h:206-253: the next frame, already read; the previous one's buffer goes back to the reader
static int hip_get_frame()
{
    std::unique_lock<std::mutex> l(hip_ring_m);
    if (hip_cur_slot >= 0) {        // (its upload has returned: the planes are the host's again)
        ++hip_ring_freed;
        hip_cur_slot = -1;
        hip_ring_cv.notify_all();
    }
    size_t slot;
    if (hip_early_slot >= 0) {      // taken out of the ring (and handed to the device) by hip_early_next() already
        slot = (size_t)hip_early_slot;
        hip_early_slot = -1;
    } else {
        hip_ring_cv.wait(l, [] { return hip_ring_tail < hip_ring_head; });
        slot = hip_ring_tail % HIP_RING;
        const int state = hip_ring_state[slot];
        if (state == -1) printf("broken stream!\n");
        if (state != 1) return state;
        ++hip_ring_tail;
        hip_cur_on_device = false;
        hip_chroma_async = false;
    }
    hip_cur_slot = (int)slot;
    frames.tmp_Y = hip_ring[slot];
    frames.tmp_U = frames.tmp_Y + video.src_frame_size_luma;
    frames.tmp_V = frames.tmp_U + video.src_frame_size_chroma;
    frames.current_Y = frames.tmp_Y;      // source size; the device pads
    frames.current_U = frames.tmp_U;
    frames.current_V = frames.tmp_V;
    return 1;
}

// The NEXT frame, if the reader has it, handed to the device NOW -- at the end of an iteration, while this frame's loop filter has most of
// its time in front of it: its upload (a pack from the staging buffer the prefetch filled), the following frame started on its way, and
// scene_change()'s scan enqueued (vp8hip_chroma_change_async).  The next iteration then finds frame and answer on the device, and its side
// work (parameter scan, pyramid, GOLDEN / ALTREF searches) is enqueued early enough to run beside the filter instead of behind it.
static void hip_early_next()
{
    size_t slot;
    {
        std::lock_guard<std::mutex> l(hip_ring_m);
        if (hip_early_slot >= 0 || !(hip_ring_tail < hip_ring_head) || hip_ring_state[hip_ring_tail % HIP_RING] != 1) return;
        slot = hip_ring_tail % HIP_RING;
        ++hip_ring_tail;
    }
    hip_early_slot = (int)slot;
    uint8_t *y = hip_ring[slot];
    hip_site_timer t_(0);
    HIP_CK(vp8hip_upload_current(hip_ctx, y, y + video.src_frame_size_luma, y + video.src_frame_size_luma + video.src_frame_size_chroma));
    hip_cur_on_device = true;
    hip_prefetch_next();
    HIP_CK(vp8hip_chroma_change_async(hip_ctx));
    hip_chroma_async = true;
}

// the frame after the current one, if the reader has it: started on its way to the device now (vp8hip_prefetch_current); the upload that
// names the same planes in the next iteration then copies nothing
static void hip_prefetch_next()
{
    size_t seq;
    {
        std::lock_guard<std::mutex> l(hip_ring_m);
        if (!(hip_ring_tail < hip_ring_head) || hip_ring_state[hip_ring_tail % HIP_RING] != 1) return;
        seq = hip_ring_tail;
    }
    if (seq == hip_prefetched_seq) return;
    uint8_t *y = hip_ring[seq % HIP_RING];
    HIP_CK(vp8hip_prefetch_current(hip_ctx, y, y + video.src_frame_size_luma, y + video.src_frame_size_luma + video.src_frame_size_chroma));
    hip_prefetched_seq = seq;
}

// get_loopfilter_strength() + prepare_segments_data(), vp8enc.cpp:96-127 + 131-227, on the device for the current frame (no host scan, no
// round trip).  update_filter (check_SSIM's prepare_segments_data(1, 7), :260-261) never arrives here: vp8hip_check_ssim_async does it.
static void hip_prepare_segments(const int update_filter, const int shrpnss)
{
    (void)update_filter; (void)shrpnss;
    hip_upload_current();       // key frames reach this call before anything of theirs is on the device (vp8enc.cpp:379-383): the scan needs the frame
    hip_site_timer t_(1);
    const cl_int *refqi = frames.current_is_altref_frame ? video.altrefqi : video.lastqi;     // :149-151
    HIP_CK(vp8hip_auto_segments(hip_ctx, frames.current_is_key_frame, (const int32_t *)refqi, (int)video.qi_min));
}

// scene_change()'s two sums, vp8enc.cpp:270-284, on the device (the decision with its hold-over stays the reference's)
static void hip_chroma_diffs(int *Udiff, int *Vdiff)
{
    int32_t u = 0, v = 0;
    hip_site_timer t_(3);
    if (hip_chroma_async) {
        hip_chroma_async = false;
        HIP_CK(vp8hip_chroma_change_result(hip_ctx, &u, &v));
    } else
        HIP_CK(vp8hip_chroma_change(hip_ctx, &u, &v));
    *Udiff = u;
    *Vdiff = v;
}
#endif   // VP8HIP_FAST

// init_all()'s device half, init.h:430-1276 (and :107-374, the platform / program / kernel objects): one context.  The three
// host arrays that were mapped OpenCL buffers (vp8enc.cpp:355-361) become plain allocations.
static int hip_init()
{
    frames.MB = (macroblock_coeffs_t *)malloc(sizeof(macroblock_coeffs_t) * video.mb_count);
    frames.reconstructed_Y = (cl_uchar *)malloc(video.wrk_frame_size_luma);
    frames.reconstructed_U = (cl_uchar *)malloc(video.wrk_frame_size_chroma);
    frames.reconstructed_V = (cl_uchar *)malloc(video.wrk_frame_size_chroma);
    video.partition_step = video.partition_step / video.number_of_partitions;   // init.h:1190
    const int rc = vp8hip_create(&hip_ctx, video.wrk_width, video.wrk_height, video.SSIM_target, (int)device.gpu_preferred_platform_number);
    if (rc != VP8HIP_OK) {
        printf("no usable MI355X: %s\n", vp8hip_status_string(rc));     // where the reference says "no GPU device found" (init.h:146-150)
        return -1;
    }
#ifdef VP8HIP_FAST
    if (hip_fast_init() < 0) return -1;
#endif
    hip_timeline = getenv("VP8HIP_DROP_IN_TIMELINE") != NULL;
    clock_gettime(CLOCK_MONOTONIC, &hip_loop_t0);
    return 1;
}

// finalize(), vp8enc.cpp:505-681: ~170 clRelease* calls
static void hip_finalize()
{
    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double loop_s = (double)(t1.tv_sec - hip_loop_t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - hip_loop_t0.tv_nsec);
    fprintf(stderr, "vp8hip_drop_in: frame loop %d frames %.6f s\n", (int)frames.frame_number, loop_s);
    if (hip_timeline) {
        double in_calls = 0.0;
        for (int i = 0; i < 8; ++i) in_calls += hip_seconds_in[i];
        for (int i = 0; i < 8; ++i) fprintf(stderr, "vp8hip_drop_in: in %-20s %.6f s\n", hip_site_name[i], hip_seconds_in[i]);
        fprintf(stderr, "vp8hip_drop_in: in the reference's own host code (reader, padding, scans, frame types, writer) %.6f s\n", loop_s - in_calls);
    }
#ifdef VP8HIP_FAST
    hip_fast_shutdown();
#endif
    vp8hip_destroy(hip_ctx);
    free(frames.MB);
    free(frames.reconstructed_Y);
    free(frames.reconstructed_U);
    free(frames.reconstructed_V);
}

// prepare_segments_data()'s tail, vp8enc.cpp:222-227: clEnqueueWriteBuffer(segments_data_gpu / _cpu)
static void hip_set_segments() { hip_site_timer t_(1); HIP_CK(vp8hip_set_segments(hip_ctx, (const int32_t *)frames.segments_data)); }

// main(), vp8enc.cpp:386-406: three clEnqueueWriteBuffer(current_frame_Y/U/V); the re-upload of the filtered reconstruction
// (:389-404) has no counterpart -- it never leaves the device
static void hip_upload_current()
{
    hip_site_timer t_(0);
#ifdef VP8HIP_FAST
    if (hip_cur_on_device) return;
    hip_cur_on_device = true;
#endif
    HIP_CK(vp8hip_upload_current(hip_ctx, frames.current_Y, frames.current_U, frames.current_V));
#ifdef VP8HIP_FAST
    hip_prefetch_next();
#endif
}

// inter_part.h:1-384: prepare_GPU_buffers() + inter_transform(), ~82 kernel launches on three queues
static void inter_transform()
{
    const int use_golden = !frames.prev_is_golden_frame;                                                                        // :103
    const int use_altref = (!frames.prev_is_altref_frame) && (frames.altref_frame_number != frames.golden_frame_number);        // :104
    hip_site_timer t_(2);
    HIP_CK(vp8hip_inter_transform(hip_ctx, frames.prev_is_golden_frame, frames.prev_is_altref_frame, use_golden, use_altref));
}

// main(), vp8enc.cpp:421-434 and inter_part.h:263-265: nine clEnqueueReadBuffer
static void hip_download_results()
{
    hip_site_timer t_(3);
#ifdef VP8HIP_KEEP_HOST_STAGES
    vp8hip_results r;
    r.MB_parts = frames.MB_parts;
    r.MB_reference_frame = frames.MB_reference_frame;
    r.MB_vectors = (int16_t *)frames.MB_vectors;
    r.MB_coeffs = (int16_t *)frames.MB;
    r.MB_segment_id = frames.MB_segment_id;
    r.MB_SSIM = frames.MB_SSIM;
    r.recon_Y = frames.reconstructed_Y;
    r.recon_U = frames.reconstructed_U;
    r.recon_V = frames.reconstructed_V;
    HIP_CK(vp8hip_download_results(hip_ctx, &r));
#endif      // (on the device nothing of it is needed on the host)
}

#ifndef VP8HIP_KEEP_HOST_STAGES
// check_SSIM()'s loop and mean, vp8enc.cpp:242-257 (test_inter_on_intra, intra_part.h:855-1087, per macroblock below the target)
static void hip_check_ssim(float *min1, float *min2)
{
#ifdef VP8HIP_FAST
    // nobody waits: the fallback, the statistics and the filter update (:260-261) run on the device in front of the loop filter; what the
    // host has to decide -- "redo as key frame", :443-453 -- it decides in hip_entropy_encode(), where the frame's bytes are next needed
    hip_site_timer t_(3);
    const cl_int *refqi = frames.current_is_altref_frame ? video.altrefqi : video.lastqi;
    HIP_CK(vp8hip_check_ssim_async(hip_ctx, (const int32_t *)refqi, (int)video.qi_min));
    frames.replaced = 0;            // provisional: main()'s own test at :443 must not fire now
    frames.new_SSIM = 2.0f;
    *min1 = 0.0f;                   // (so that check_SSIM()'s own prepare_segments_data(1, 7) stays out: the device has done it)
    *min2 = 0.0f;
#else
    int32_t replaced = 0;
    float new_ssim = 0.0f, mn = 2.0f;
    hip_site_timer t_(3);
    HIP_CK(vp8hip_check_ssim(hip_ctx, &replaced, &new_ssim, &mn));
    frames.replaced = replaced;
    frames.new_SSIM = new_ssim;
    *min1 = mn;
    *min2 = mn;     // (the minimum before the fallback: printed only)
#endif
}

// intra_transform()'s loop and uploads, intra_part.h:1100-1126 (predict_and_transform_mb per macroblock on the host, then the
// reconstruction to the device)
static void hip_intra_transform()
{
    hip_site_timer t_(7);
#ifdef VP8HIP_FAST
    hip_upload_current();       // (a no-op by now: prepare_segments_data() stands in front of every intra_transform())
#else
    HIP_CK(vp8hip_upload_current(hip_ctx, frames.current_Y, frames.current_U, frames.current_V));   // key frames were host-only
#endif
    HIP_CK(vp8hip_intra_transform(hip_ctx));
}
#else
// intra_transform()'s uploads, intra_part.h:1112-1126: the host-coded key frame goes to the device
static void hip_upload_intra_results()
{
    hip_site_timer t_(7);
    HIP_CK(vp8hip_upload_mb_data(hip_ctx, (const int16_t *)frames.MB, frames.MB_parts, frames.MB_segment_id));
    HIP_CK(vp8hip_upload_recon(hip_ctx, frames.reconstructed_Y, frames.reconstructed_U, frames.reconstructed_V));
}
#endif

// main(), vp8enc.cpp:457-470: coefficients / parts / segment ids (and, LF on the CPU device, the reconstruction) handed to the
// device that filters -- after check_SSIM's fallback may have changed them on the host
static void hip_upload_host_results()
{
#ifdef VP8HIP_KEEP_HOST_STAGES
    if (!frames.current_is_key_frame && frames.replaced > 0) {    // (key frames: intra_transform() has uploaded already)
        HIP_CK(vp8hip_upload_mb_data(hip_ctx, (const int16_t *)frames.MB, frames.MB_parts, frames.MB_segment_id));
        HIP_CK(vp8hip_upload_recon(hip_ctx, frames.reconstructed_Y, frames.reconstructed_U, frames.reconstructed_V));
    }
#endif
}

// loop_filter.h:1-55: prepare_filter_mask on either device + the read-back of the non-zero counts + skip_prob (:37-44)
static void prepare_filter_mask_and_non_zero_coeffs()
{
    hip_site_timer t_(4);
#ifdef VP8HIP_FAST
    // no read-back: the counts stay where the entropy stage reads them; inter frames got mask and counts from vp8hip_inter_transform /
    // vp8hip_check_ssim_async, key frames need the call
    if (frames.current_is_key_frame) HIP_CK(vp8hip_prepare_filter_mask(hip_ctx, NULL));
#else
    HIP_CK(vp8hip_prepare_filter_mask(hip_ctx, frames.MB_non_zero_coeffs));
    frames.skip_prob = vp8host_skip_prob(frames.MB_non_zero_coeffs, video.mb_count);
#endif
}

// loop_filter.h:57-190: do_loop_filter() on either device; the filtered frame IS the next LAST
static void do_loop_filter()
{
    if (video.GOP_size < 2) return;     // :59, :142
    hip_site_timer t_(5);
    HIP_CK(vp8hip_loop_filter(hip_ctx));
}

// debug.h:12-23: the filtered reconstruction for the dump
static void hip_download_last() { HIP_CK(vp8hip_download_last(hip_ctx, frames.reconstructed_Y, frames.reconstructed_U, frames.reconstructed_V)); }

extern void encode_header(cl_uchar *const partition);   // entropy_host.cpp:709
#ifdef VP8HIP_FAST
static void intra_transform();      // intra_part.h:1089 (the reference's own: GOP counters, then hip_intra_transform())

// after main()'s loop: the last frame's bytes (nothing was coded behind it)
static void hip_flush_last_frame()
{
    hip_have_lagged = false;
    frames.encoded_frame_size = 0;
    if (!hip_frame_pending) return;
    size_t n = 0;
    HIP_CK(vp8hip_encode_frame_end(hip_ctx, frames.encoded_frame, (size_t)((video.src_frame_size_luma + 2 * video.src_frame_size_chroma) << 1), &n));
    frames.encoded_frame_size = (cl_uint)n;
    hip_frame_pending = false;
    hip_have_lagged = true;
}
#endif

// entropy_encode()'s body, vp8enc.cpp:50-91
static void hip_entropy_encode()
{
    hip_site_timer t_(6);
#ifdef VP8HIP_KEEP_HOST_STAGES
    // count_probs + num_div_denom on the CPU device and their two read-backs (:58-68)
    HIP_CK(vp8hip_count_probs(hip_ctx, (int)video.number_of_partitions, (uint32_t *)frames.new_probs, (uint32_t *)frames.new_probs_denom));
    for (int i = 0; i < 4; ++i)               // contexts no bool was coded in take the default probability (:69-76)
        for (int j = 0; j < 8; ++j)
            for (int k = 0; k < 3; ++k)
                for (int l = 0; l < 11; ++l)
                    if (frames.new_probs_denom[i][j][k][l] < 2) frames.new_probs[i][j][k][l] = k_default_coeff_probs[i][j][k][l];
    // the write-back of the probabilities + encode_coefficients (:77-81); the partitions land where gather_frame() picks them up
    HIP_CK(vp8hip_encode_coefficients(hip_ctx, (const uint32_t *)frames.new_probs, (int)video.number_of_partitions, (int)video.partition_step,
                                      frames.partitions, (int32_t *)frames.partition_sizes));
    encode_header(frames.encoded_frame);      // :84, the reference's own first-partition coder, untouched
#elif defined(VP8HIP_FAST)
    const size_t cap = (size_t)((video.src_frame_size_luma + 2 * video.src_frame_size_chroma) << 1);    // init.h:407
    // (a) the PREVIOUS frame's bytes: its stage ran beside its loop filter and beside this frame's searches
    hip_have_lagged = false;
    frames.encoded_frame_size = 0;
    if (hip_frame_pending) {
        size_t n = 0;
        HIP_CK(vp8hip_encode_frame_end(hip_ctx, frames.encoded_frame, cap, &n));
        frames.encoded_frame_size = (cl_uint)n;
        hip_frame_pending = false;
        hip_have_lagged = true;
    }
    // (b) this frame's verdict (a few microseconds into its loop filter's launch) and the reference's decision on it, vp8enc.cpp:443-453
    int32_t replaced = 0;
    if (!frames.current_is_key_frame) {
        int32_t updated = 0;
        float new_ssim = 0.0f, mn = 2.0f;
        HIP_CK(vp8hip_check_ssim_result(hip_ctx, &replaced, &new_ssim, &mn, &updated));
        frames.replaced = replaced;
        frames.new_SSIM = new_ssim;
        if (video.print_info) printf("%d>AvgSSIM=%f; MinSSIM=%f; repl:%d ", frames.frame_number, frames.new_SSIM, mn, frames.replaced);
        if ((frames.replaced > (video.mb_count / 6)) || (frames.new_SSIM < video.SSIM_target)) {
            if (frames.new_SSIM < video.SSIM_target) ++encStat.scene_changes_by_ssim;
            else ++encStat.scene_changes_by_replaced;
            frames.current_is_key_frame = 1;        // redo as intra: the current frame is still on the device
            hip_prepare_segments(0, 0);
            intra_transform();
            if (video.print_info) printf("\nkey frame FORCED by bad inter-result: replaced(%d) and SSIM(%f)!\n", frames.replaced, frames.new_SSIM);
            HIP_CK(vp8hip_prepare_filter_mask(hip_ctx, NULL));
            if (video.GOP_size >= 2) HIP_CK(vp8hip_loop_filter(hip_ctx));
            replaced = 0;
        }
    }
    // (c) this frame's entropy stage, enqueued; its bytes are taken in the next iteration (or by hip_flush_last_frame)
    vp8hip_header_params hp;
    hp.is_key = frames.current_is_key_frame;
    hp.is_golden = frames.current_is_golden_frame;
    hp.is_altref = frames.current_is_altref_frame;
    hp.loop_filter_type = video.loop_filter_type;
    hp.loop_filter_sharpness = VP8HIP_SHARPNESS_ON_DEVICE;      // what vp8hip_auto_segments / the filter update left on the device
    hp.partitions_log2 = 0;
    hp.width = video.dst_width;
    hp.height = video.dst_height;
    hp.use_intra_info = !frames.current_is_key_frame && replaced > 0;     // (with nothing replaced the fallback did not even initialise them: the same bits)
    HIP_CK(vp8hip_encode_frame_begin(hip_ctx, (int)video.number_of_partitions, &hp));
    hip_frame_pending = true;
    // (d) ... and the next frame, early
    if (!getenv("VP8HIP_DROP_IN_NO_EARLY")) hip_early_next();
#else
    // ... and, on the device, everything up to the finished frame: gather_frame() (encIO.h:1-30) has nothing left to do
    vp8hip_header_params hp;
    hp.is_key = frames.current_is_key_frame;
    hp.is_golden = frames.current_is_golden_frame;
    hp.is_altref = frames.current_is_altref_frame;
    hp.loop_filter_type = video.loop_filter_type;
    hp.loop_filter_sharpness = video.loop_filter_sharpness;
    hp.partitions_log2 = 0;
    hp.width = video.dst_width;
    hp.height = video.dst_height;
    hp.use_intra_info = !frames.current_is_key_frame;        // check_SSIM ran on every inter frame (:442)
    size_t n = 0;
    HIP_CK(vp8hip_encode_frame(hip_ctx, (int)video.number_of_partitions, &hp, frames.encoded_frame,
                               (size_t)((video.src_frame_size_luma + 2 * video.src_frame_size_chroma) << 1), &n));   // init.h:407
    frames.encoded_frame_size = (cl_uint)n;
#endif
}
