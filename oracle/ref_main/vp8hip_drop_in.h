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
// Both must write the same .ivf (tests/test_ref_main.py).
#pragma once
#include <stdint.h>
#include <string.h>

#include "vp8hip.h"
#include "vp8hip_host.h"

#include <time.h>

static vp8hip_ctx *hip_ctx = NULL;
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
            exit(3);                                                                                                   \
        }                                                                                                              \
    } while (0)

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
static void hip_upload_current() { hip_site_timer t_(0); HIP_CK(vp8hip_upload_current(hip_ctx, frames.current_Y, frames.current_U, frames.current_V)); }

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
    int32_t replaced = 0;
    float new_ssim = 0.0f, mn = 2.0f;
    hip_site_timer t_(3);
    HIP_CK(vp8hip_check_ssim(hip_ctx, &replaced, &new_ssim, &mn));
    frames.replaced = replaced;
    frames.new_SSIM = new_ssim;
    *min1 = mn;
    *min2 = mn;     // (the minimum before the fallback: printed only)
}

// intra_transform()'s loop and uploads, intra_part.h:1100-1126 (predict_and_transform_mb per macroblock on the host, then the
// reconstruction to the device)
static void hip_intra_transform()
{
    hip_site_timer t_(7);
    HIP_CK(vp8hip_upload_current(hip_ctx, frames.current_Y, frames.current_U, frames.current_V));   // key frames were host-only
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
    HIP_CK(vp8hip_prepare_filter_mask(hip_ctx, frames.MB_non_zero_coeffs));
    frames.skip_prob = vp8host_skip_prob(frames.MB_non_zero_coeffs, video.mb_count);
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
