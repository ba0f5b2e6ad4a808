/* vp8hip_taps.h -- measurement and debug taps of libvp8hip.so (bench.py, the tests; NOT part of the reference boundary; included by vp8hip.h): per-kernel timing, intermediate buffers by name. */
#ifndef VP8HIP_TAPS_H
#define VP8HIP_TAPS_H

#include "vp8hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement taps (bench.py / tests; not part of the reference boundary) -------------- */
typedef enum {
    VP8HIP_K_PACK = 0,      /* tight planes -> padded surfaces */
    VP8HIP_K_DOWNSAMPLE,    /* downsample_x2                     GPU_kernels.cl:429  */
    VP8HIP_K_SEARCH1_L4,    /* luma_search_1step, 1/16           GPU_kernels.cl:459  */
    VP8HIP_K_SEARCH1_L3,
    VP8HIP_K_SEARCH1_L2,
    VP8HIP_K_SEARCH1_L1,
    VP8HIP_K_SEARCH1_L0,    /* full resolution: the kernel BASELINE.json's roofline target names */
    VP8HIP_K_SEARCH2,       /* luma_search_2step                 GPU_kernels.cl:1068 */
    VP8HIP_K_SELECT,        /* select_reference + pack_8x8_into_16x16 */
    VP8HIP_K_MB,            /* predictors + dct/quant/wht/idct + SSIM + filter mask */
    VP8HIP_K_FILTER_MASK,   /* prepare_filter_mask (recompute)   CPU_kernels.cl:782  */
    VP8HIP_K_LOOP_FILTER,   /* loop_filter_frame_luma/_chroma    CPU_kernels.cl:970,1333 */
    VP8HIP_K_BORDER,        /* edge replication of a new reference */
    VP8HIP_K_ENT_COUNT,     /* count_probs + num_div_denom       CPU_kernels.cl:536,764 */
    VP8HIP_K_ENT_ENCODE,    /* encode_coefficients               CPU_kernels.cl:347 */
    VP8HIP_K_INTRA,         /* key frame / check_SSIM fallback   intra_part.h:517-1109 */
    VP8HIP_K_HDR_ENCODE,    /* encode_header (first partition)   entropy_host.cpp:709 */
    VP8HIP_K_COUNT
} vp8hip_kernel_id;

/* time the kernels whose bit is set in mask (1u << id) with hipEvents on the ctx stream */
int vp8hip_profile_enable(vp8hip_ctx *ctx, uint32_t mask);
/* blocks until the stream is idle; total_ms[id] / launches[id] since the last read (arrays of VP8HIP_K_COUNT) */
int vp8hip_profile_read(vp8hip_ctx *ctx, double *total_ms, int64_t *launches);
/* The loop filter's duration by the kernel's own clock (s_memrealtime: start of its first band to the end of its last row),
 * summed since the last call.  With many contexts in flight the HIP events of vp8hip_profile_read also count the time a
 * packet waits for its queue to be scheduled; this figure does not, and it is what a rocprofv3 kernel trace shows. */
int vp8hip_profile_read_clock(vp8hip_ctx *ctx, double *loop_filter_ms, int64_t *loop_filter_launches, double *shader_clock_ghz /* may be NULL:
    the shader clock those launches ran at (s_memtime cycles per s_memrealtime tick, averaged over the launches) */);
/* Among the launches counted by the last vp8hip_profile_read_clock: how many had the wave that runs the frame's last row END on
 * another hardware slot than it started on -- it was context-switched, which happens when the process holds more queues than
 * the part's scheduler keeps resident (24 on MI355X: GPU_MAX_HW_QUEUES plus what torch / RCCL create).  0 when healthy. */
int64_t vp8hip_profile_context_switches(const vp8hip_ctx *ctx);
/* k_search2's launches by the kernel's own clock (earliest workgroup start to latest workgroup end, sampled every 64th
 * workgroup) since the last call; a batched launch counts once, on the batch's first member. */
int vp8hip_profile_read_search2_clock(vp8hip_ctx *ctx, double *ms, int64_t *launches);
/* the stamping costs about 1 % of throughput: off until asked for (on = 1), per context (a batch follows its first member) */
int vp8hip_profile_search2_clock(vp8hip_ctx *ctx, int on);

/* stage outputs of the last vp8hip_inter_transform, for parity tests */
typedef enum {
    VP8HIP_DBG_NET1 = 0,   /* ref, -      : short2[b8]   (after the 2-step search: qpel vectors)   */
    VP8HIP_DBG_NET2,       /* ref, -      : short2[b8]   (after the 1x 1-step search: full-pel)    */
    VP8HIP_DBG_BDIFF,      /* ref, -      : int[b8]                                               */
    VP8HIP_DBG_PYRAMID,    /* ref(3=cur), level 0..4 : tight (W>>l)x(H>>l) plane                   */
    VP8HIP_DBG_MB_MASK,    /* -           : int[MBs]                                              */
    VP8HIP_DBG_MB_NZ,      /* -           : int[MBs]                                              */
    VP8HIP_DBG_THIRD_CONTEXT, /* -        : uchar[MBs][25] (entries of coded macroblocks, after vp8hip_count_probs) */
    VP8HIP_DBG_CURRENT_CHROMA /* ref 0 = U, 1 = V : tight (W/2)x(H/2) plane of the current frame (after copy_with_padding) */
} vp8hip_debug_id;
int vp8hip_debug_download(vp8hip_ctx *ctx, int what, int ref, int level, void *dst, size_t bytes);


#ifdef __cplusplus
}
#endif
#endif
