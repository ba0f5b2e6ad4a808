// api_entropy.hip -- the entropy stage of one context: entropy_encode() (vp8enc.cpp:48-94), encode_header (entropy_host.cpp:709-1256), gather_frame (encIO.h:1-30).
// count_probs / encode_coefficients / encode_header as separate calls, and the whole stage of a frame enqueued at once (vp8hip_encode_frame_begin / _end).
#include "vp8hip_ctx.h"

using namespace vp8;

namespace vp8 {

// scratch of the boolean coder, allocated the first time the stage is used (a context that only runs the
// inter path never pays for it): 64 bools per 4x4 block on average (of at most 304)
void ent_free(vp8hip_ctx *c) {
    EntBuffers &e = c->ent;
    c->ent_arena.release();
    e = EntBuffers{};
    if (c->h_frame) hipHostFree(c->h_frame);   // sized from the scratch: reallocated with it
    hipFree(c->d_frame);
    c->h_frame = nullptr;
    c->d_frame = nullptr;
}

int ent_alloc(vp8hip_ctx *c) {
    if (c->ent.plan) return VP8HIP_OK;   // the last allocation below: set only when all of them succeeded
    if (c->ent.offs) ent_free(c);        // a partial allocation left by an earlier failure
    EntBuffers &e = c->ent;
    const size_t nslots = (size_t)c->mbs * 25;
    e.cap_bools = (uint32_t)(nslots * (size_t)c->ent_bools_per_block);
    e.cap_chunks = e.cap_bools / 256 + 2 * ENT_MAX_PARTITIONS;
    e.cap_words = (uint32_t)(((size_t)e.cap_bools * 7 + 31) / 32 + 8 * ENT_MAX_PARTITIONS);
    DeviceArena &A = c->ent_arena;        // one allocation (DeviceArena, vp8hip_ctx.h); e.plan, set last, says that it succeeded
    EntPlan *plan = nullptr;
    A.want(&e.offs, (nslots + 1) * 4);
    A.want(&e.tile_sum, (nslots / 256 + 8) * 4);   // the frame path sums per 256 slots
    A.want(&e.bools, (size_t)e.cap_bools * 2 + 1024);
    A.want(&e.maps, ent_maps_entries(e.cap_chunks) * 4);
    A.want(&e.start, (size_t)e.cap_chunks * 8);
    A.want(&e.acc, (size_t)e.cap_words * 8);
    A.want(&e.bytes, (size_t)e.cap_words * 4);
    A.want(&e.sizes, ENT_MAX_PARTITIONS * 4);
    A.want(&plan, sizeof(EntPlan));
    HIPCHK(c, A.commit());
    e.plan = plan;
    return VP8HIP_OK;
}

// A frame denser than the scratch was sized for (64 bools per 4x4 block to begin with): double it, up to the 304 bools a
// block can produce at most, so that no frame is ever refused for the device's sake.  false = already at the maximum.
int ent_grow(vp8hip_ctx *c) {
    if (c->ent_bools_per_block >= 304) return VP8HIP_ERR_OVERFLOW;
    hipStreamSynchronize(c->stream);
    ent_free(c);
    c->ent_bools_per_block = c->ent_bools_per_block * 2 > 304 ? 304 : c->ent_bools_per_block * 2;
    const int rc = ent_alloc(c);       // VP8HIP_ERR_HIP (e.g. out of memory) is reported as such, not as an overflow
    if (rc) ent_free(c);
    return rc;
}

int hdr_alloc(vp8hip_ctx *c) {
    if (c->hdr.bools) return VP8HIP_OK;
    EntBuffers &e = c->hdr;
    const size_t n = (size_t)c->mbs;
    e.cap_bools = (uint32_t)(n * 128 + 16384);   // a macroblock header is at most ~125 bools, the frame header < 10 k
    e.cap_chunks = e.cap_bools / 256 + 4;
    e.cap_words = (uint32_t)(((size_t)e.cap_bools * 7 + 31) / 32 + 16);
    DeviceArena &A = c->hdr_arena;        // one allocation; e.bools -- what hdr_alloc looks at -- is set when it has succeeded
    uint16_t *bools = nullptr;
    A.want(&e.offs, (n + 1) * 4);
    A.want(&e.tile_sum, (n / 1024 + 8) * 4);
    A.want(&bools, (size_t)e.cap_bools * 2 + 1024);
    A.want(&e.maps, ent_maps_entries(e.cap_chunks) * 4);
    A.want(&e.start, (size_t)e.cap_chunks * 8);
    A.want(&e.acc, (size_t)e.cap_words * 8);
    A.want(&e.bytes, (size_t)e.cap_words * 4);
    A.want(&e.sizes, ENT_MAX_PARTITIONS * 4);
    A.want(&e.plan, sizeof(EntPlan));
    A.want(&c->hdr_partial, HDR_STAT_WORDS * 4);   // the census of k_hdr_count: zero at rest (k_hdr_frame clears it)
    A.want(&c->hdr_info, 16);
    A.want(&c->hdr_sym, 64);
    HIPCHK(c, A.commit());
    HIPCHK(c, hipMemsetAsync(c->hdr_partial, 0, HDR_STAT_WORDS * 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // once per context: the stage may run on another stream than this one (a batch's, the third)
    e.bools = bools;
    return VP8HIP_OK;
}

// The coder's last kernel writes the finished frame straight into the pinned host buffer (device-visible, a few tens of
// KiB over PCIe) instead of into device memory followed by a copy command: one operation fewer in every frame's chain.
// Same-box A/B (VP8HIP_FRAME_ZEROCOPY=0 brings the copy back): 4 050 vs 3 880 frames/s at 1080p, 1 198 vs 1 128 at 4K.
bool frame_zero_copy() {
    static const bool on = [] { const char *v = getenv("VP8HIP_FRAME_ZEROCOPY"); return !(v && v[0] == '0'); }();
    return on;
}
// Everything of a frame's entropy stage up to the read-back, enqueued; nothing waits.
// buffers + the description of one context's frame for the entropy stage's launchers
int frame_prepare(vp8hip_ctx *c, int P, const vp8hip_header_params *p, FrameEntropy &e, FrameOut &fo) {
    int rc = ent_alloc(c);
    if (rc) return rc;
    if ((rc = hdr_alloc(c))) return rc;
    if (!c->h_frame) {   // the finished frame: device copy + pinned host copy
        c->h_frame_cap = (size_t)c->hdr.cap_words * 4 + (size_t)c->ent.cap_words * 4 + 64;
        HIPCHK(c, hipHostMalloc(&c->h_frame, c->h_frame_cap));
        if (!frame_zero_copy()) HIPCHK(c, hipMalloc(&c->d_frame, c->h_frame_cap));
    }
    e.o = c->out;
    e.flags = c->ent_flags;
    e.third = c->ent_third;
    e.counts = c->ent_counts;
    e.probs = c->ent_probs;
    e.denom0 = c->ent_denom0;
    e.coef = &c->ent;
    e.hdr = &c->hdr;
    e.hdr_partial = c->hdr_partial;
    e.hdr_info = c->hdr_info;
    e.hdr_sym = c->hdr_sym;
    const bool intra_info = p->is_key || p->use_intra_info;
    e.is_inter = (!p->is_key && p->use_intra_info) ? c->intra_is_inter : nullptr;
    e.modes = intra_info ? c->intra_modes : nullptr;
    e.f.is_key = p->is_key ? 1 : 0;
    e.f.is_golden = p->is_golden ? 1 : 0;
    e.f.is_altref = p->is_altref ? 1 : 0;
    e.f.loop_filter_type = p->loop_filter_type;
    e.f.sharpness = p->loop_filter_sharpness;
    e.f.partitions_log2 = P == 8 ? 3 : (P == 4 ? 2 : (P == 2 ? 1 : 0));
    e.d_sd = c->d_sd;
    e.strength = reinterpret_cast<const int32_t *>(c->d_stats + 4);
    e.mbw = c->mbw;
    e.mbh = c->mbh;
    e.P = P;
    fo.frame = frame_zero_copy() ? c->h_frame : c->d_frame;
    fo.head = p->is_key ? 10 : 3;
    fo.capacity = (uint32_t)(c->h_frame_cap - 16);
    return VP8HIP_OK;
}
int frame_enqueue(vp8hip_ctx *c, int P, const vp8hip_header_params *p) {
    FrameEntropy e;
    FrameOut fo;
    const int rc = frame_prepare(c, P, p, e, fo);
    if (rc) return rc;
    // With the loop filter in flight (vp8hip_filter_overlap) the stage runs on a stream of its own from where the filter started:
    // everything it reads was final then -- except the segment data check_SSIM may update INSIDE the filter's launch, so a caller that
    // has not taken the verdict gets the stage behind the filter instead.
    if (c->lf_pending && c->verdict_pending) { const int jr = join_lf(c); if (jr) return jr; }
    if (c->lf_pending && !c->ent_stream && !c->prof_mask) {
        // made with the first frame asked for, not with the overlap mode: an idle stream still takes part in the runtime's stream ->
        // hardware queue assignment (two videos without frames out: 4 000 frames/s, with a third stream each that nothing ran on 2 600)
        static const bool third = [] { const char *v = getenv("VP8HIP_ENT_STREAM"); return !(v && v[0] == '0'); }();
        int least = 0, greatest = 0;
        if (third && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
            hipStreamCreateWithPriority(&c->ent_stream, hipStreamNonBlocking, least) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_ent, FRAME_EVENT_FLAGS) != hipSuccess) {
            hipStreamDestroy(c->ent_stream);
            c->ent_stream = nullptr;
        }
    }
    const bool third = c->lf_pending && c->ent_stream && !c->prof_mask;
    hipStream_t s = third ? c->ent_stream : c->stream;
    // (behind the fork event where one was recorded; otherwise the caller has taken the verdict -- see above -- and the filter's
    // launch, behind which everything the stage reads was final, is under way)
    if (third && !c->fork_by_verdict_at_launch) HIPCHK(c, hipStreamWaitEvent(s, c->ev_fork, 0));
    c->frame_event = nullptr;
    c->frame_gen = c->out_gen;
    static const bool stepwise = [] { const char *v = getenv("VP8HIP_ENT_STEPWISE"); return v && v[0] && v[0] != '0'; }();
    if (stepwise && c->mbs * 25 <= 1024 * 1024) {   // A/B switch (the step-by-step scan stops at 2^20 blocks): the bool strings by the step-by-step kernels (15 launches instead of 5), then the same coder
        const uint8_t *defaults = hdr_default_coeff_probs();
        launch_ent_count(s, c->out, c->ent_flags, c->ent_third, c->ent_counts, c->ent_probs, c->ent_denom0, c->mbw, c->mbh, P, defaults);
        if (!defaults) launch_default_probs(s, c->ent_probs, c->ent_denom0);
        launch_ent_encode(s, c->out, c->ent_third, c->ent_probs, c->ent, c->mbw, c->mbh, P, false);
        launch_hdr_encode(s, c->out, e.is_inter, e.modes, e.f, c->d_sd, e.strength, c->ent_probs, c->ent_denom0, c->hdr, c->hdr_partial,
                          c->hdr_sym, c->hdr_info, c->mbw, c->mbh, false);
    } else {
    {   // count_probs + num_div_denom + the default-probability fallback (vp8enc.cpp:58-76), bools per block and per macroblock header
        Timed t(c, VP8HIP_K_ENT_COUNT);
        launch_fe_count(s, e);
    }
    {   // encode_header's bools (:84) and encode_coefficients' (:77-81)
        Timed t(c, VP8HIP_K_HDR_ENCODE);
        launch_fe_emit(s, e);
    }
    }
    c->ent_counted_partitions = P;
    {   // the boolean coder on both strings; its last kernel is gather_frame (encIO.h:1-30) and writes into the pinned host
        // buffer (or, VP8HIP_FRAME_ZEROCOPY=0, into device memory: then the frame size and the first FRAME_FIRST_COPY bytes travel
        // in one copy and only a larger frame needs a second one)
        Timed t(c, VP8HIP_K_ENT_ENCODE);
        launch_frame_code(s, c->ent, P, c->hdr, fo.head, fo.capacity, fo.frame);
    }
    HIPCHK(c, hipGetLastError());
    if (!frame_zero_copy()) {   // (otherwise the coder's last kernel wrote the frame into the pinned host buffer itself)
        const size_t first = c->h_frame_cap < FRAME_FIRST_COPY ? c->h_frame_cap : FRAME_FIRST_COPY;
        HIPCHK(c, hipMemcpyAsync(c->h_frame, c->d_frame, first, hipMemcpyDeviceToHost, s));
    }
    if (third) {
        HIPCHK(c, hipEventRecord(c->ev_ent, s));
        c->frame_event = c->ev_ent;
        c->ent_pending = true;
    }
    return VP8HIP_OK;
}

}  // namespace vp8

extern "C" {

int vp8hip_count_probs(vp8hip_ctx *c, int num_partitions, uint32_t *new_probs, uint32_t *new_probs_denom) {
    USE_DEVICE(c);
    if (!c || !new_probs || !new_probs_denom) return VP8HIP_ERR_ARG;
    if (num_partitions != 1 && num_partitions != 2 && num_partitions != 4 && num_partitions != 8) return VP8HIP_ERR_ARG;
    {
        Timed t(c, VP8HIP_K_ENT_COUNT);
        launch_ent_count(c->stream, c->out, c->ent_flags, c->ent_third, c->ent_counts, c->ent_probs, c->ent_denom0, c->mbw,
                         c->mbh, num_partitions);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(new_probs, c->ent_probs, sizeof(uint32_t) * ENT_NCTX, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(new_probs_denom, c->ent_denom0, sizeof(uint32_t) * ENT_NCTX, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // the reference's read-backs are blocking (CL_TRUE, vp8enc.cpp:67-68)
    c->ent_counted_partitions = num_partitions;
    return VP8HIP_OK;
}

int vp8hip_encode_coefficients(vp8hip_ctx *c, const uint32_t *coeff_probs, int num_partitions, int partition_step,
                               uint8_t *partitions, int32_t *partition_sizes) {
    USE_DEVICE(c);
    if (!c || !coeff_probs || !partitions || !partition_sizes || partition_step < 4) return VP8HIP_ERR_ARG;
    if (num_partitions != 1 && num_partitions != 2 && num_partitions != 4 && num_partitions != 8) return VP8HIP_ERR_ARG;
    if (c->ent_counted_partitions != num_partitions) return VP8HIP_ERR_STATE;   // needs vp8hip_count_probs first
    int rc = ent_alloc(c);
    if (rc) return rc;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->ent_probs, coeff_probs, sizeof(uint32_t) * ENT_NCTX, hipMemcpyHostToDevice, s));
    EntPlan plan;
    for (;;) {
        {
            Timed t(c, VP8HIP_K_ENT_ENCODE);
            launch_ent_encode(s, c->out, c->ent_third, c->ent_probs, c->ent, c->mbw, c->mbh, num_partitions);
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&plan, c->ent.plan, sizeof(plan), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (!plan.overflow) break;
        if ((rc = ent_grow(c)) != VP8HIP_OK) return rc;   // denser than the scratch: enlarge it and code the frame again
    }
    for (int p = 0; p < num_partitions; ++p)
        if (plan.nbytes[p] > (uint32_t)partition_step) return VP8HIP_ERR_OVERFLOW;
    for (int p = 0; p < num_partitions; ++p) {
        partition_sizes[p] = (int32_t)plan.nbytes[p];
        HIPCHK(c, hipMemcpyAsync(partitions + (size_t)p * partition_step, c->ent.bytes + (size_t)plan.word_base[p] * 4,
                                 plan.nbytes[p], hipMemcpyDeviceToHost, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    return VP8HIP_OK;
}

int vp8hip_encode_header(vp8hip_ctx *c, const vp8hip_header_params *p, uint8_t *out, size_t capacity, size_t *size) {
    USE_DEVICE(c);
    if (!c || !p || !out || !size) return VP8HIP_ERR_ARG;
    if (c->ent_counted_partitions == 0) return VP8HIP_ERR_STATE;    // the coefficient probabilities of this frame: vp8hip_count_probs first
    const size_t head = p->is_key ? 10 : 3;
    if (capacity < head + 8) return VP8HIP_ERR_OVERFLOW;
    int rc = hdr_alloc(c);
    if (rc) return rc;
    hipStream_t s = c->stream;
    HdrFrame f;
    f.is_key = p->is_key ? 1 : 0;
    f.is_golden = p->is_golden ? 1 : 0;
    f.is_altref = p->is_altref ? 1 : 0;
    f.loop_filter_type = p->loop_filter_type;
    f.sharpness = p->loop_filter_sharpness;
    f.partitions_log2 = p->partitions_log2;
    const bool intra_info = p->is_key || p->use_intra_info;
    {
        Timed t(c, VP8HIP_K_HDR_ENCODE);
        launch_hdr_encode(s, c->out, (!p->is_key && p->use_intra_info) ? c->intra_is_inter : nullptr, intra_info ? c->intra_modes : nullptr, f,
                          c->d_sd, reinterpret_cast<const int32_t *>(c->d_stats + 4), c->ent_probs, c->ent_denom0, c->hdr, c->hdr_partial,
                          c->hdr_sym, c->hdr_info, c->mbw, c->mbh);
    }
    HIPCHK(c, hipGetLastError());
    EntPlan plan;
    HIPCHK(c, hipMemcpyAsync(&plan, c->hdr.plan, sizeof(plan), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (plan.overflow || head + plan.nbytes[0] > capacity) return VP8HIP_ERR_OVERFLOW;
    if (plan.nbytes[0] >= (1u << 19)) return VP8HIP_ERR_FORMAT;   // the frame tag has 19 bits for the first partition's size
    HIPCHK(c, hipMemcpyAsync(out + head, c->hdr.bytes, plan.nbytes[0], hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    // frame tag (entropy_host.cpp:1214-1247): key/inter bit, version 0, show_frame, size of the first partition
    const uint32_t tag = (p->is_key ? 0u : 1u) | 0x10u | (plan.nbytes[0] << 5);
    out[0] = (uint8_t)tag;
    out[1] = (uint8_t)(tag >> 8);
    out[2] = (uint8_t)(tag >> 16);
    if (p->is_key) {
        const int w = p->width > 0 ? p->width : c->W, h = p->height > 0 ? p->height : c->H;
        out[3] = 0x9d; out[4] = 0x01; out[5] = 0x2a;
        out[6] = (uint8_t)w; out[7] = (uint8_t)(w >> 8);
        out[8] = (uint8_t)h; out[9] = (uint8_t)(h >> 8);
    }
    *size = head + plan.nbytes[0];
    return VP8HIP_OK;
}

// the entropy stage's scratch and the pinned frame buffer, which are otherwise made when the first frame is asked for
int vp8hip_reserve_frame_path(vp8hip_ctx *c) {
    USE_DEVICE(c);
    if (!c) return VP8HIP_ERR_ARG;
    vp8hip_header_params p{};
    FrameEntropy e;
    FrameOut fo;
    return frame_prepare(c, 1, &p, e, fo);
}

// the same for the densest frame there can be (304 bools per 4x4 block, ~270 MB at 1080p): no frame is ever coded twice, which a
// caller that starts frame n + 1 before it takes frame n's bytes relies on
int vp8hip_reserve_frame_path_dense(vp8hip_ctx *c) {
    USE_DEVICE(c);
    if (!c) return VP8HIP_ERR_ARG;
    if (c->frame_pending) return VP8HIP_ERR_STATE;
    if (c->ent_bools_per_block < 304) {
        JOIN_LF(c);
        hipStreamSynchronize(c->stream);
        if (c->ent_stream) hipStreamSynchronize(c->ent_stream);
        if (c->ent.offs) ent_free(c);
        c->ent_bools_per_block = 304;
    }
    return vp8hip_reserve_frame_path(c);
}

int vp8hip_encode_frame_begin(vp8hip_ctx *c, int num_partitions, const vp8hip_header_params *p) {
    USE_DEVICE(c);
    if (!c || !p) return VP8HIP_ERR_ARG;
    const int P = num_partitions;
    if (P != 1 && P != 2 && P != 4 && P != 8) return VP8HIP_ERR_ARG;
    drop_overflowed_frame(c);                        // a frame given up after VP8HIP_ERR_OVERFLOW is coded again
    if (c->frame_pending) return VP8HIP_ERR_STATE;   // (no size limit here: the frame path's prefix sums take any number of blocks)
    const int rc = frame_enqueue(c, P, p);
    if (rc) return rc;
    c->frame_params = *p;
    c->frame_partitions = P;
    c->frame_pending = true;
    return VP8HIP_OK;
}

int vp8hip_encode_frame_end(vp8hip_ctx *c, uint8_t *out, size_t capacity, size_t *size) {
    USE_DEVICE(c);
    if (!c || !out || !size) return VP8HIP_ERR_ARG;
    if (!c->frame_pending) return VP8HIP_ERR_STATE;
    c->frame_pending = c->frame_overflowed = false;
    const vp8hip_header_params *p = &c->frame_params;
    hipStream_t s = c->stream;
    size_t n;
    for (;;) {
        if (c->frame_event) HIPCHK(c, hipEventSynchronize(c->frame_event));   // the stage ran beside the chain: its end, not the chain's
        else HIPCHK(c, hipStreamSynchronize(s));
        c->frame_event = nullptr;
        n = *reinterpret_cast<const uint32_t *>(c->h_frame);
        if (n) break;
        // denser than the coder's scratch was sized for: enlarge it and code the frame again (at most three times) -- which needs
        // the frame's results, gone if the caller has started the next frame in the meantime (vp8hip_reserve_frame_path_dense
        // sizes the scratch so that this cannot happen)
        if (c->frame_gen != c->out_gen) return VP8HIP_ERR_STATE;
        int rc = ent_grow(c);
        if (rc) return rc;
        rc = frame_enqueue(c, c->frame_partitions, p);
        if (rc) return rc;
    }
    if (n > capacity) {
        c->frame_pending = c->frame_overflowed = true;   // the coded frame stays in h_frame: the caller may come back with a larger
        return VP8HIP_ERR_OVERFLOW;                      // buffer (_end again, or the one-shot vp8hip_encode_frame / vp8drv_get_frame)
    }
    if (reinterpret_cast<const uint32_t *>(c->h_frame)[1] >= (1u << 19)) return VP8HIP_ERR_FORMAT;   // 19-bit size field of the frame tag
    const size_t head = p->is_key ? 10 : 3;
    const size_t first = c->h_frame_cap < FRAME_FIRST_COPY ? c->h_frame_cap : FRAME_FIRST_COPY;
    if (16 + n > first && !frame_zero_copy()) {
        HIPCHK(c, hipMemcpyAsync(c->h_frame + first, c->d_frame + first, 16 + n - first, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    memcpy(out + head, c->h_frame + 16 + head, n - head);
    const uint32_t first_part = reinterpret_cast<const uint32_t *>(c->h_frame)[1];   // size of the first partition, for the frame tag
    const uint32_t tag = (p->is_key ? 0u : 1u) | 0x10u | (first_part << 5);
    out[0] = (uint8_t)tag;
    out[1] = (uint8_t)(tag >> 8);
    out[2] = (uint8_t)(tag >> 16);
    if (p->is_key) {
        const int w = p->width > 0 ? p->width : c->W, h = p->height > 0 ? p->height : c->H;
        out[3] = 0x9d; out[4] = 0x01; out[5] = 0x2a;
        out[6] = (uint8_t)w; out[7] = (uint8_t)(w >> 8);
        out[8] = (uint8_t)h; out[9] = (uint8_t)(h >> 8);
    }
    *size = n;
    return VP8HIP_OK;
}

int vp8hip_encode_frame(vp8hip_ctx *c, int num_partitions, const vp8hip_header_params *p, uint8_t *out, size_t capacity, size_t *size) {
    USE_DEVICE(c);
    if (!c || !p || !out || !size) return VP8HIP_ERR_ARG;
    // the retry after VP8HIP_ERR_OVERFLOW: the frame is coded and waiting, only the delivery is repeated
    const int rc = (c->frame_pending && c->frame_overflowed) ? VP8HIP_OK : vp8hip_encode_frame_begin(c, num_partitions, p);
    return rc ? rc : vp8hip_encode_frame_end(c, out, capacity, size);
}

}  // extern "C"
