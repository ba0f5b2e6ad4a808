// api_inter.hip -- the inter path of one context: prepare_GPU_buffers + inter_transform (inter_part.h:1-384), check_SSIM (vp8enc.cpp:231-263),
// key frames (intra_part.h:1089-1109), filter mask and loop filter (loop_filter.h:25-55, 185-190).
#include "vp8hip_ctx.h"

using namespace vp8;

namespace vp8 {

// A frame that did not fit its caller's buffer (VP8HIP_ERR_OVERFLOW from vp8hip_encode_frame_end) stays pending for a retry
// with a larger one; a caller that goes on to the next frame instead has given it up.
void drop_overflowed_frame(vp8hip_ctx *c) {
    if (c->frame_overflowed) c->frame_pending = c->frame_overflowed = false;
    c->chk_armed = false;   // (a new frame begins: a check armed for the previous reconstruction does not ride with this one's filter)
}

// what inter_begin would refuse, without touching the context (a batch validates every member before it changes any)
int inter_check(const vp8hip_ctx *c, int prev_is_golden, int prev_is_altref, int use_golden, int use_altref) {
    if (c->slot[0] < 0) return VP8HIP_ERR_STATE;
    const int golden = prev_is_golden ? c->slot[0] : c->slot[1], altref = prev_is_altref ? c->slot[0] : c->slot[2];
    if ((use_golden && golden < 0) || (use_altref && altref < 0)) return VP8HIP_ERR_STATE;
    return VP8HIP_OK;   // (a surface for the reconstruction always exists: five surfaces, at most three references)
}

// reference rotation + the reconstruction surface of the new frame: the head of every inter frame
int inter_begin(vp8hip_ctx *c, int prev_is_golden, int prev_is_altref, int use_golden, int use_altref) {
    if (c->slot[0] < 0) return VP8HIP_ERR_STATE;
    c->ent_counted_partitions = 0;
    drop_overflowed_frame(c);
    ++c->out_gen;
    // reference rotation, inter_part.h:35-50,72-83: golden/altref := the frame that is LAST now
    if (prev_is_golden) c->slot[1] = c->slot[0];
    if (prev_is_altref) c->slot[2] = c->slot[0];
    if ((use_golden && c->slot[1] < 0) || (use_altref && c->slot[2] < 0)) return VP8HIP_ERR_STATE;
    if (c->recon < 0 || c->recon == c->slot[0] || c->recon == c->slot[1] || c->recon == c->slot[2]) {
        c->recon = -1;
        c->recon = pick_free_frame(c);
        if (c->recon < 0) return VP8HIP_ERR_STATE;
    }
    return VP8HIP_OK;
}

RefSet ref_set(const vp8hip_ctx *c, int use_last, int use_golden, int use_altref) {
    RefSet refs;
    refs.use[0] = use_last ? 1 : 0;
    refs.use[1] = use_golden ? 1 : 0;
    refs.use[2] = use_altref ? 1 : 0;
    for (int r = 0; r < 3; ++r) refs.ref[r] = c->frames[c->slot[r] >= 0 ? c->slot[r] : c->slot[0]].f;
    return refs;
}

unsigned long long *s2_clock_words(const vp8hip_ctx *c) { return reinterpret_cast<unsigned long long *>(c->d_progress + S2_CLOCK_WORD); }
// what the launches get: the stamping costs 1.1 % of the headline (same-box A/B, 57.1 against 57.8 M MB/s), so it is on only
// while a host asks for it (vp8hip_profile_search2_clock; bench.py: during its warm-up steps)
unsigned long long *s2_clock(const vp8hip_ctx *c) { return c->s2_clock_on ? s2_clock_words(c) : nullptr; }

// hierarchical search, inter_part.h:110-236; ping-pong as bound at init.h:672-854.  One launch per level over the
// references in `which` (the reference runs the three references on three queues, inter_part.h:122-135)
void search_refs(vp8hip_ctx *c, const RefSet &which) {
    hipStream_t s = c->stream;
    const int net_width = c->mbw * 2;
    int src = 0;
    // one video coded frame after frame (filter on its own stream): nothing else fills the chip and every launch is a link of the
    // frame's chain -- the four coarse levels are ONE launch there (k_search1_coarse), the finest follows
    static const int fuse = [] { const char *v = getenv("VP8HIP_S1_COARSE"); return v && v[0] ? atoi(v) : 1; }();   // 0 = a launch per level, 1 = levels 4-1 fused (default), 2 = all five (no faster: the finest level is whole-chip work, profiles/README.md)
    int first = 4;
    if (c->lf_overlap && fuse) {
        Timed t(c, fuse == 2 ? VP8HIP_K_SEARCH1_L0 : VP8HIP_K_SEARCH1_L1);
        launch_search1_coarse(s, c->cur, which, c->nets, net_width, fuse == 2);
        first = fuse == 2 ? -1 : 0;      // (level 0 reads the level-1 net: src index 0)
    }
    for (int l = first; l >= 0; --l) {
        Timed t(c, VP8HIP_K_SEARCH1_L4 + (4 - l));
        // (short waves on the coarse levels of a lone video when the fused launch is switched off: VP8HIP_S1_COARSE=0)
        launch_search1(s, c->cur, which, c->nets, l, src, net_width, c->lf_overlap && l > 0);
        src ^= 1;
    }
    Timed t(c, VP8HIP_K_SEARCH2);
    // (the launch clock only where launches of this context cannot overlap: the one that searches LAST)
    launch_search2(s, c->cur, which, c->nets, which.use[0] ? s2_clock(c) : nullptr);
}

// prepare_GPU_buffers, inter_part.h:1-33 (reset_vectors is folded into k_search1's parent read)
void pyramids(vp8hip_ctx *c) {
    FrameSurf &last = c->frames[c->slot[0]];
    // LAST fresh out of the loop filter: its replicated edges ride in the pyramid launch (they are invalid only together)
    if (!last.pyramid_valid && !c->cur_pyramid_valid) {
        build_pyramid(c, &c->cur, &last.f, last.border_valid ? 0u : 2u);
    } else {
        if (!c->cur_pyramid_valid) build_pyramid(c, &c->cur, nullptr);
        if (!last.pyramid_valid) build_pyramid(c, &last.f, nullptr, last.border_valid ? 0u : 1u);
        else if (!last.border_valid) launch_border(c->stream, last.f);
    }
    last.pyramid_valid = true;
    last.border_valid = true;
    c->cur_pyramid_valid = true;
}

int claim_recon(vp8hip_ctx *c) {
    if (c->recon < 0 || c->recon == c->slot[0] || c->recon == c->slot[1] || c->recon == c->slot[2]) {
        c->recon = -1;
        c->recon = pick_free_frame(c);
        if (c->recon < 0) return VP8HIP_ERR_STATE;
    }
    return VP8HIP_OK;
}

// ---- check_SSIM without the host round trip ---------------------------------------------------------------------------------
void check_item(vp8hip_ctx *c, CheckItem &it, const int32_t refqi[4], int qi_min) {
    it.cur = &c->cur;
    it.recon = &c->frames[c->recon].f;
    it.o = &c->out;
    it.sd = c->d_sd;
    it.modes = c->intra_modes;
    it.is_inter = c->intra_is_inter;
    it.prog = c->intra_prog;
    it.err = c->d_progress + LF_ERR_WORD;
    it.gen = ++c->intra_gen;
    c->ent_counted_partitions = 0;
    c->chk_armed = true;
    for (int k = 0; k < 4; ++k) c->chk_refqi[k] = refqi[k];
    c->chk_qi_min = qi_min;
}
// what the loop filter launch needs to carry an armed check's verdict (on = 0 otherwise)
void lf_check(vp8hip_ctx *c, LfCheck &k) {
    k.on = c->chk_armed ? 1 : 0;
    if (!k.on) return;
    c->chk_armed = false;
    k.qi_min = c->chk_qi_min;
    for (int i = 0; i < 4; ++i) k.refqi[i] = c->chk_refqi[i];
    k.is_inter = c->intra_is_inter;
    k.strength = reinterpret_cast<int32_t *>(c->d_stats + 4);
    k.stats = c->intra_stats;
    k.verdict = c->h_verdict;
    k.seq = ++c->verdict_seq;
    c->verdict_pending = true;
}

// With the reference's default target of -1 no macroblock can lie below it -- a macroblock's SSIM is a product of a factor in (0, 1]
// and one that is > -1 by 2 c2 / (sum of variances + c2), four hundred float steps at the least -- so k_mb never raises the flag and
// the fallback's launch would leave at once: it is not made.  A launch that does nothing still holds its stream for as long as its
// workgroups wait for a place on the full chip: 4 % of the headline (VP8HIP_ALWAYS_LAUNCH_FALLBACK=1 for same-box A/B runs).
bool fallback_possible(float ssim_target) {
    static const bool always = [] { const char *v = getenv("VP8HIP_ALWAYS_LAUNCH_FALLBACK"); return v && v[0] == '1'; }();
    return always || ssim_target > -1.0f;
}

}  // namespace vp8

extern "C" {

int vp8hip_inter_transform(vp8hip_ctx *c, int prev_is_golden, int prev_is_altref, int use_golden, int use_altref) {
    USE_DEVICE(c);
    if (!c) return VP8HIP_ERR_ARG;
    note_queue_oversubscription();
    int rc = inter_begin(c, prev_is_golden, prev_is_altref, use_golden, use_altref);
    if (rc) return rc;
    const RefSet refs = ref_set(c, 1, use_golden, use_altref);
    // While the previous frame's loop filter is still running on its own stream (vp8hip_filter_overlap): only LAST is
    // what it writes.  GOLDEN and ALTREF -- never the frame being filtered: use_golden / use_altref exclude a reference
    // that was refreshed by the previous frame, inter_part.h:103-104 -- are searched first, beside the filter; the filter
    // is joined only then, and LAST follows.  A single video coded frame after frame is bound by the filter's dependency
    // chain (0.36 ms of a 0.60 ms frame at 1080p); this takes 1.8 of the 2.8 reference searches out of the chain.
    const bool split = c->lf_pending && (use_golden || use_altref) && (!use_golden || c->slot[1] != c->slot[0]) &&
                       (!use_altref || c->slot[2] != c->slot[0]);
    bool src_marked = false;
    if (split) {
        if (!c->cur_pyramid_valid) build_pyramid(c, &c->cur, nullptr);
        c->cur_pyramid_valid = true;
        src_marked = hipEventRecord(c->ev_src, c->stream) == hipSuccess;   // (`stream` is the side stream while the filter is pending)
        search_refs(c, ref_set(c, 0, use_golden, use_altref));
    }
    hipStream_t late_side = nullptr;   // the side stream, when its GOLDEN / ALTREF searches are joined in front of k_mb only
    if (c->lf_pending && c->cur_pyramid_valid && c->slot[0] >= 0 && !c->frames[c->slot[0]].pyramid_valid) {
        hipStream_t side = join_lf_swap(c);
        FrameSurf &last = c->frames[c->slot[0]];
        build_pyramid(c, &last.f, nullptr, last.border_valid ? 0u : 1u);     // behind the filter, beside whatever the side stream still runs
        last.pyramid_valid = last.border_valid = true;
        if (src_marked && side_sources_done(c)) late_side = side;
        else {
            const int wr = join_lf_wait(c, side);
            if (wr) return wr;
        }
    } else {
        const int jr = join_lf(c, /*defer_ent=*/true);
        if (jr) return jr;
    }
    pyramids(c);
    search_refs(c, split ? ref_set(c, 1, 0, 0) : refs);
    if (late_side) {
        // ONE barrier packet in front of k_mb: the side stream waits for the previous frame's entropy stage itself (its barrier
        // costs the chain nothing), and the chain for the side stream
        if (c->ent_pending) {
            c->ent_pending = false;
            HIPCHK(c, hipStreamWaitEvent(late_side, c->ev_ent, 0));
        }
        const int wr = join_lf_wait(c, late_side);
        if (wr) return wr;
    } else {
        const int jr = join_ent(c);      // (the previous frame's coefficients, vectors and modes are the stage's until here)
        if (jr) return jr;
    }
    {
        Timed t(c, VP8HIP_K_MB);   // select_reference + pack_8x8_into_16x16 run inside
        launch_mb(c->stream, c->cur, refs, c->nets, c->frames[c->recon].f, c->out, c->d_sd, c->ssim_target, c->mbw, c->mbh, c->conformant != 0);
    }
    c->recon_ready = true;
    HIPCHK(c, hipGetLastError());
    return VP8HIP_OK;
}

// ---- one frame's reference searches on different devices --------------------------------------------
// The reference runs the LAST / GOLDEN / ALTREF searches of a frame on three command queues (inter_part.h:122-135,
// 201-236): they share nothing but the current frame.  Split over devices, each searches the references in its mask,
// the vectors and costs travel (8 bytes per 8x8 block and reference) and the device that finishes the frame needs all of them.
int vp8hip_inter_search(vp8hip_ctx *c, int prev_is_golden, int prev_is_altref, int use_golden, int use_altref, int search_mask) {
    USE_DEVICE(c);
    if (!c) return VP8HIP_ERR_ARG;
    JOIN_LF(c);
    int rc = inter_begin(c, prev_is_golden, prev_is_altref, use_golden, use_altref);
    if (rc) return rc;
    pyramids(c);
    const int m = search_mask & (1 | (use_golden ? 2 : 0) | (use_altref ? 4 : 0));
    if (m) search_refs(c, ref_set(c, m & 1, m & 2, m & 4));
    HIPCHK(c, hipGetLastError());
    return VP8HIP_OK;
}

int vp8hip_inter_finish(vp8hip_ctx *c, int use_golden, int use_altref) {
    USE_DEVICE(c);
    if (!c) return VP8HIP_ERR_ARG;
    if (c->slot[0] < 0 || c->recon < 0) return VP8HIP_ERR_STATE;
    JOIN_LF(c);
    {
        Timed t(c, VP8HIP_K_MB);
        launch_mb(c->stream, c->cur, ref_set(c, 1, use_golden, use_altref), c->nets, c->frames[c->recon].f, c->out, c->d_sd, c->ssim_target,
                  c->mbw, c->mbh, c->conformant != 0);
    }
    c->recon_ready = true;
    HIPCHK(c, hipGetLastError());
    return VP8HIP_OK;
}

int vp8hip_intra_transform(vp8hip_ctx *c) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c) return VP8HIP_ERR_ARG;
    if (c->cur_count == 0) return VP8HIP_ERR_STATE;
    int rc = claim_recon(c);
    if (rc) return rc;
    c->ent_counted_partitions = 0;
    drop_overflowed_frame(c);
    ++c->out_gen;
    {
        Timed t(c, VP8HIP_K_INTRA);
        launch_intra(c->stream, c->cur, c->frames[c->recon].f, c->out, c->d_sd, c->intra_modes, c->intra_is_inter, c->intra_prog,
                     ++c->intra_gen, c->d_progress + LF_ERR_WORD, 0.0f, 1, c->mbw, c->mbh, c->lf_stall_test);
    }
    c->recon_ready = true;
    HIPCHK(c, hipGetLastError());
    return VP8HIP_OK;
}

int vp8hip_check_ssim(vp8hip_ctx *c, int32_t *replaced, float *new_ssim, float *min_ssim) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c) return VP8HIP_ERR_ARG;
    if (!c->recon_ready || c->recon < 0 || c->cur_count == 0) return VP8HIP_ERR_STATE;
    c->ent_counted_partitions = 0;
    {
        Timed t(c, VP8HIP_K_INTRA);
        launch_intra(c->stream, c->cur, c->frames[c->recon].f, c->out, c->d_sd, c->intra_modes, c->intra_is_inter, c->intra_prog,
                     ++c->intra_gen, c->d_progress + LF_ERR_WORD, c->ssim_target, 0, c->mbw, c->mbh, c->lf_stall_test, c->conformant);
    }
    launch_ssim_stats(c->stream, c->out, c->intra_is_inter, c->mbs, c->d_progress + LF_ERR_WORD, c->intra_stats);
    HIPCHK(c, hipGetLastError());
    int32_t st[4];
    HIPCHK(c, hipMemcpyAsync(st, c->intra_stats, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (replaced) *replaced = st[0];
    if (new_ssim) memcpy(new_ssim, &st[1], 4);
    if (min_ssim) memcpy(min_ssim, &st[2], 4);
    if (st[3]) {   // a bounded device-side wait expired (this frame or an earlier, unchecked one)
        HIPCHK(c, hipMemset(c->d_progress + LF_ERR_WORD, 0, 4));
        return VP8HIP_ERR_TIMEOUT;
    }
    return VP8HIP_OK;
}

int vp8hip_check_ssim_async(vp8hip_ctx *c, const int32_t refqi[4], int qi_min) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c || !refqi) return VP8HIP_ERR_ARG;
    if (!c->recon_ready || c->recon < 0 || c->cur_count == 0 || c->verdict_pending || c->chk_armed) return VP8HIP_ERR_STATE;
    CheckItem it;
    check_item(c, it, refqi, qi_min);
    if (fallback_possible(c->ssim_target)) {
        Timed t(c, VP8HIP_K_INTRA);
        launch_check_fallback(c->stream, &it, 1, c->ssim_target, c->mbw, c->mbh, c->conformant);
    }
    HIPCHK(c, hipGetLastError());
    return VP8HIP_OK;
}

int vp8hip_check_ssim_ready(const vp8hip_ctx *c) {   // 1: vp8hip_check_ssim_result would not wait (or there is nothing to wait for)
    if (!c || !c->verdict_pending) return 1;
    return (uint32_t)__atomic_load_n(&c->h_verdict[5], __ATOMIC_ACQUIRE) == c->verdict_seq ? 1 : 0;
}

int vp8hip_check_ssim_result(vp8hip_ctx *c, int32_t *replaced, float *new_ssim, float *min_ssim, int32_t *filter_updated) {
    USE_DEVICE_ONLY(c);
    if (!c) return VP8HIP_ERR_ARG;
    if (!c->verdict_pending) return VP8HIP_ERR_STATE;    // (also: armed, but the loop filter that carries the verdict not yet launched)
    // The verdict workgroup of the loop filter launch writes five words and then the sequence number, at system scope, into
    // host memory the device sees: it is there a few microseconds into that launch, long before the launch ends.
    volatile int32_t *v = c->h_verdict;
    const uint32_t want = c->verdict_seq;
    static const bool nowait = experiment_env("VP8HIP_EXPERIMENT_NOWAIT") != nullptr;   // timing experiment only: what the waiting costs
    for (unsigned spins = 0; !nowait && (uint32_t)__atomic_load_n(&c->h_verdict[5], __ATOMIC_ACQUIRE) != want; ++spins) {
        if ((spins & 0xfff) == 0xfff) {   // every few thousand polls: is the stream still alive?
            const hipError_t q = hipStreamQuery(c->verdict_stream);
            if (q != hipErrorNotReady && (uint32_t)__atomic_load_n(&c->h_verdict[5], __ATOMIC_ACQUIRE) != want) {
                // the stream is idle (or failed) and the word never came: the launch did not run its verdict workgroup
                c->verdict_pending = false;
                if (q != hipSuccess) { c->last_hip_error = (int)q; return VP8HIP_ERR_HIP; }
                return VP8HIP_ERR_TIMEOUT;
            }
        }
        __builtin_ia32_pause();
    }
    c->verdict_pending = false;
    int32_t st[5];
    for (int i = 0; i < 5; ++i) st[i] = v[i];
    if (replaced) *replaced = st[0];
    if (new_ssim) memcpy(new_ssim, &st[1], 4);
    if (min_ssim) memcpy(min_ssim, &st[2], 4);
    if (filter_updated) *filter_updated = st[4];
    if (st[3]) {   // a bounded device-side wait expired (this frame or an earlier, unchecked one)
        HIPCHK(c, hipMemset(c->d_progress + LF_ERR_WORD, 0, 4));
        return VP8HIP_ERR_TIMEOUT;
    }
    return VP8HIP_OK;
}

int vp8hip_download_intra(vp8hip_ctx *c, int32_t *modes, int32_t *is_inter) {
    USE_DEVICE(c);
    if (!c) return VP8HIP_ERR_ARG;
    if (modes) HIPCHK(c, hipMemcpyAsync(modes, c->intra_modes, (size_t)c->mbs * 64, hipMemcpyDeviceToHost, c->stream));
    if (is_inter) HIPCHK(c, hipMemcpyAsync(is_inter, c->intra_is_inter, (size_t)c->mbs * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VP8HIP_OK;
}

int vp8hip_prepare_filter_mask(vp8hip_ctx *c, int32_t *nz_out) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c) return VP8HIP_ERR_ARG;
    c->ent_counted_partitions = 0;
    hipStream_t s = c->stream;
    {
        Timed t(c, VP8HIP_K_FILTER_MASK);
        launch_filter_mask(s, c->out, c->d_sd, c->mbs);
    }
    if (nz_out) {
        HIPCHK(c, hipMemcpyAsync(nz_out, c->out.nz, (size_t)c->mbs * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    return VP8HIP_OK;
}

int vp8hip_loop_filter(vp8hip_ctx *c) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c) return VP8HIP_ERR_ARG;
    if (!c->recon_ready || c->recon < 0) return VP8HIP_ERR_STATE;
    Frame &f = c->frames[c->recon].f;
    LfCheck chk;
    lf_check(c, chk);
    if (c->lf_overlap && !c->prof_mask) {   // (the per-kernel timers bracket launches on the context's stream only)
        hipStream_t chain = c->stream;
        const bool by_verdict = chk.on != 0;      // (see side_stream_ordered)
        if (!by_verdict) HIPCHK(c, hipEventRecord(c->ev_fork, chain));
        launch_loop_filter4(chain, f, c->out, c->d_sd, c->d_progress, c->d_lf_handoff, c->mbw, c->mbh, c->lf_launches++, c->lf_stall_test, &chk);
        c->verdict_stream = chain;
        if (!by_verdict) HIPCHK(c, hipStreamWaitEvent(c->lf_stream, c->ev_fork, 0));   // the side work starts where the filter starts
        c->fork_by_verdict = c->fork_by_verdict_at_launch = by_verdict;
        c->stream = c->lf_stream;
        c->lf_stream = chain;
        c->lf_pending = true;
        c->lf_sd = c->d_sd;
    } else {
        Timed t(c, VP8HIP_K_LOOP_FILTER);
        launch_loop_filter4(c->stream, f, c->out, c->d_sd, c->d_progress, c->d_lf_handoff, c->mbw, c->mbh, c->lf_launches++, c->lf_stall_test, &chk);
        c->verdict_stream = c->stream;
    }
    // the filtered reconstruction is the LAST reference of the next frame (vp8enc.cpp:395-401); its replicated edges are made
    // with its pyramid, in one launch, when that frame begins (pyramids())
    c->frames[c->recon].pyramid_valid = false;
    c->frames[c->recon].border_valid = false;
    c->slot[0] = c->recon;
    c->recon = -1;
    c->recon_ready = false;
    HIPCHK(c, hipGetLastError());
    return VP8HIP_OK;
}

}  // extern "C"
