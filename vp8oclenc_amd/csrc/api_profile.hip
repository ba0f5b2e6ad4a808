// api_profile.hip -- per-kernel timers, the kernels' own clocks, debug taps and test hooks (nothing a host needs to code a frame).
//
#include "vp8hip_ctx.h"

using namespace vp8;

namespace vp8 {

int prof_collect(vp8hip_ctx *c) {
    if (c->ev_used == 0) return VP8HIP_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->ev_used; i += 2) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        c->prof_ms[c->ev_kernel[i / 2]] += ms;
        c->prof_n[c->ev_kernel[i / 2]] += 1;
    }
    c->ev_used = 0;
    return VP8HIP_OK;
}

}  // namespace vp8

extern "C" {

int vp8hip_profile_enable(vp8hip_ctx *c, uint32_t mask) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c) return VP8HIP_ERR_ARG;
    int rc = prof_collect(c);
    c->prof_mask = mask;
    return rc;
}

int vp8hip_profile_read(vp8hip_ctx *c, double *total_ms, int64_t *launches) {
    USE_DEVICE(c);
    if (!c) return VP8HIP_ERR_ARG;
    int rc = prof_collect(c);
    if (rc) return rc;
    for (int k = 0; k < VP8HIP_K_COUNT; ++k) {
        if (total_ms) total_ms[k] = c->prof_ms[k];
        if (launches) launches[k] = c->prof_n[k];
        c->prof_ms[k] = 0;
        c->prof_n[k] = 0;
    }
    return VP8HIP_OK;
}

int vp8hip_profile_read_clock(vp8hip_ctx *c, double *loop_filter_ms, int64_t *loop_filter_launches, double *shader_clock_ghz) {
    USE_DEVICE(c);
    if (!c || !loop_filter_ms || !loop_filter_launches) return VP8HIP_ERR_ARG;
    JOIN_LF(c);
    unsigned long long clk[6] = {0, 0, 0, 0, 0, 0};
    int32_t *base = c->d_progress + LF_ERR_WORD + 4;
    HIPCHK(c, hipMemcpyAsync(clk, base, sizeof(clk), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(base + 2, 0, 40, c->stream));   // sums and count restart; the start stamp is rewritten by every launch
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->lf_context_switches = (int64_t)clk[5];
    if (getenv("VP8HIP_DEBUG_CLOCK")) fprintf(stderr, "lf clock: launches %llu anomalies %llu hw_id changed %llu\n", clk[2], clk[4], clk[5]);
    *loop_filter_ms = (double)clk[1] * 1e-5;   // 100 MHz ticks
    *loop_filter_launches = (int64_t)clk[2];
    if (shader_clock_ghz) *shader_clock_ghz = clk[2] > clk[4] ? (double)clk[3] / (double)(clk[2] - clk[4]) * 1e-4 : 0.0;   // (cycles per tick x 1000) x 100 MHz
    return VP8HIP_OK;
}

int vp8hip_profile_search2_clock(vp8hip_ctx *c, int on) {
    if (!c) return VP8HIP_ERR_ARG;
    c->s2_clock_on = on != 0;
    return VP8HIP_OK;
}

// k_search2's launches by the kernel's own clock since the last call: total ms and launches (a batched launch counts once, on
// the batch's first member).  See launch_clock_end (vp8hip_dev.h).
int vp8hip_profile_read_search2_clock(vp8hip_ctx *c, double *ms, int64_t *launches) {
    USE_DEVICE(c);
    if (!c || !ms || !launches) return VP8HIP_ERR_ARG;
    JOIN_LF(c);
    unsigned long long w[5] = {0, 0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(w, s2_clock_words(c), sizeof(w), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(s2_clock_words(c) + 3, 0, 16, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *ms = (double)w[3] * 1e-5;   // 100 MHz ticks
    *launches = (int64_t)w[4];
    return VP8HIP_OK;
}

// Launches (among those of the last vp8hip_profile_read_clock) in which the wave that runs the frame's last row ended on another
// hardware slot than it started on: it was context-switched, i.e. the hardware scheduler is rotating an oversubscribed set of
// queues (more than 24 per process on this part).  0 on a healthy configuration.
int64_t vp8hip_profile_context_switches(const vp8hip_ctx *c) { return c ? c->lf_context_switches : 0; }

int vp8hip_debug_download(vp8hip_ctx *c, int what, int ref, int level, void *dst, size_t bytes) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c || !dst) return VP8HIP_ERR_ARG;
    hipStream_t s = c->stream;
    switch (what) {
        case VP8HIP_DBG_NET1:
        case VP8HIP_DBG_NET2: {
            if (ref < 0 || ref > 2 || bytes != (size_t)c->b8 * 4) return VP8HIP_ERR_ARG;
            HIPCHK(c, hipMemcpyAsync(dst, c->nets.net[ref][what == VP8HIP_DBG_NET1 ? 0 : 1], bytes, hipMemcpyDeviceToHost, s));
            break;
        }
        case VP8HIP_DBG_BDIFF:
            if (ref < 0 || ref > 2 || bytes != (size_t)c->b8 * 4) return VP8HIP_ERR_ARG;
            HIPCHK(c, hipMemcpyAsync(dst, c->nets.bdiff[ref], bytes, hipMemcpyDeviceToHost, s));
            break;
        case VP8HIP_DBG_PYRAMID: {
            if (ref < 0 || ref > 3 || level < 0 || level > 4) return VP8HIP_ERR_ARG;
            if (ref < 3 && c->slot[ref] < 0) return VP8HIP_ERR_STATE;
            const Frame &f = ref == 3 ? c->cur : c->frames[c->slot[ref]].f;
            const Plane &p = f.Y[level];
            if (bytes != (size_t)p.w * p.h) return VP8HIP_ERR_ARG;
            int rc = copy_out(c, dst, p);
            if (rc) return rc;
            break;
        }
        case VP8HIP_DBG_MB_MASK:
        case VP8HIP_DBG_MB_NZ:
            if (bytes != (size_t)c->mbs * 4) return VP8HIP_ERR_ARG;
            HIPCHK(c, hipMemcpyAsync(dst, what == VP8HIP_DBG_MB_MASK ? c->out.mask : c->out.nz, bytes, hipMemcpyDeviceToHost, s));
            break;
        case VP8HIP_DBG_THIRD_CONTEXT:
            if (bytes != (size_t)c->mbs * 25) return VP8HIP_ERR_ARG;
            HIPCHK(c, hipMemcpyAsync(dst, c->ent_third, bytes, hipMemcpyDeviceToHost, s));
            break;
        case VP8HIP_DBG_CURRENT_CHROMA: {
            if (ref < 0 || ref > 1 || c->cur_count == 0) return VP8HIP_ERR_ARG;
            const Plane &p = ref ? c->cur.V : c->cur.U;
            if (bytes != (size_t)p.w * p.h) return VP8HIP_ERR_ARG;
            int rc = copy_out(c, dst, p);
            if (rc) return rc;
            break;
        }
        case 100:  // diagnostic build only (-DLF2_STAMPS): cycle sums written by the loop filter
            if (bytes != 512) return VP8HIP_ERR_ARG;
            HIPCHK(c, hipMemcpyAsync(dst, (const char *)c->d_progress + 4096, 512, hipMemcpyDeviceToHost, s));
            break;
        default:
            return VP8HIP_ERR_ARG;
    }
    HIPCHK(c, hipStreamSynchronize(s));
    return VP8HIP_OK;
}

// test tap (not in the public header): re-run the quarter-pel search of one reference and return, for
// block `block`, 26 x {8 rows x 8 predicted pixels, cost, valid} as 26 x 18 dwords
// test tap (not in the public header): weight_opt of n caller-supplied 4x4 difference blocks
int vp8hip_debug_weight(vp8hip_ctx *c, const int32_t *d, int n, int32_t *out) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c || !d || !out || n <= 0) return VP8HIP_ERR_ARG;
    int32_t *dd = nullptr, *dout = nullptr;
    HIPCHK(c, hipMalloc(&dd, (size_t)n * 64));
    HIPCHK(c, hipMalloc(&dout, (size_t)n * 4));
    HIPCHK(c, hipMemcpyAsync(dd, d, (size_t)n * 64, hipMemcpyHostToDevice, c->stream));
    launch_weight_tap(c->stream, dd, n, dout);
    HIPCHK(c, hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(dd);
    hipFree(dout);
    return VP8HIP_OK;
}

// test hook (not in the public header): while on, the loop filter's inter-band counters are published from a wrong
// base, so every band but the first runs into its bounded wait -> VP8HIP_ERR_TIMEOUT at the next synchronize
int vp8hip_debug_lf_stall(vp8hip_ctx *c, int on) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c) return VP8HIP_ERR_ARG;
    c->lf_stall_test = on ? 1 : 0;
    return VP8HIP_OK;
}

// test hook (not in the public header): MB_SSIM as an inter frame would have left it, so vp8hip_check_ssim can be
// driven from stored inter-frame results (the golden vectors of tests/golden/intra)
int vp8hip_debug_upload_ssim(vp8hip_ctx *c, const float *ssim) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c || !ssim) return VP8HIP_ERR_ARG;
    HIPCHK(c, hipMemcpyAsync(c->out.ssim, ssim, (size_t)c->mbs * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VP8HIP_OK;
}

// test hook (not in the public header): everything vp8hip_encode_header reads, set directly -- lets the tests drive the
// device coder with the stress inputs of tests/bitstream_cases.py and the golden vectors.  NULL = leave as is.
int vp8hip_debug_upload_header_inputs(vp8hip_ctx *c, const int32_t *seg, const int32_t *nz, const int32_t *ref, const int32_t *parts,
                                      const int16_t *vectors, const int32_t *is_inter, const int32_t *modes, const uint32_t *probs,
                                      const uint32_t *denom, const int32_t *sd) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c) return VP8HIP_ERR_ARG;
    hipStream_t s = c->stream;
    const size_t n = c->mbs;
    if (seg) HIPCHK(c, hipMemcpyAsync(c->out.seg, seg, n * 4, hipMemcpyHostToDevice, s));
    if (nz) HIPCHK(c, hipMemcpyAsync(c->out.nz, nz, n * 4, hipMemcpyHostToDevice, s));
    if (ref) HIPCHK(c, hipMemcpyAsync(c->out.ref, ref, n * 4, hipMemcpyHostToDevice, s));
    if (parts) HIPCHK(c, hipMemcpyAsync(c->out.parts, parts, n * 4, hipMemcpyHostToDevice, s));
    if (vectors) HIPCHK(c, hipMemcpyAsync(c->out.vec, vectors, n * 16, hipMemcpyHostToDevice, s));
    if (is_inter) HIPCHK(c, hipMemcpyAsync(c->intra_is_inter, is_inter, n * 4, hipMemcpyHostToDevice, s));
    if (modes) HIPCHK(c, hipMemcpyAsync(c->intra_modes, modes, n * 64, hipMemcpyHostToDevice, s));
    if (probs) HIPCHK(c, hipMemcpyAsync(c->ent_probs, probs, sizeof(uint32_t) * ENT_NCTX, hipMemcpyHostToDevice, s));
    if (denom) HIPCHK(c, hipMemcpyAsync(c->ent_denom0, denom, sizeof(uint32_t) * ENT_NCTX, hipMemcpyHostToDevice, s));
    if (sd) HIPCHK(c, hipMemcpyAsync(c->d_sd, sd, sizeof(SegData), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (probs || denom) c->ent_counted_partitions = 1;
    return VP8HIP_OK;
}

}  // extern "C"
