// vp8hip_api.hip -- context, HBM surfaces and the C ABI (include/vp8hip.h) of the inter-frame path.
//
// What the reference keeps as ~70 cl_mem objects and 60 pre-bound cl_kernel instances
// (init.h:430-593, 595-1271) is one context here: a pool of padded frame surfaces (a reference
// "slot" is an index into the pool, so golden := last is a pointer copy, not the five
// clEnqueueCopyBuffer + three clEnqueueCopyImage of inter_part.h:35-50,72-83), the vector nets,
// the per-macroblock outputs, and one in-order HIP stream.
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>
#include <new>

#include <rccl/rccl.h>

#include "../../include/vp8hip.h"
#include "vp8hip_dev.h"

using namespace vp8;

namespace {

constexpr int NFRAMES = 5;  // LAST, GOLDEN, ALTREF may all differ, + the reconstruction in flight, + 1 spare
constexpr int MAX_EVENTS = 4096;

struct FrameSurf {
    Frame f;
    bool pyramid_valid = false;
    bool border_valid = true;    // false: fresh out of the loop filter, replicated edges still to be made (with its pyramid)
};

}  // namespace

struct vp8hip_batch;
struct vp8hip_ctx {
    int W = 0, H = 0, mbw = 0, mbh = 0, mbs = 0, b8 = 0;
    float ssim_target = -1.0f;
    int device = 0;
    hipStream_t stream = nullptr;       // the stream in use: the context's own, or its batch's (vp8hip_batch_create)
    hipStream_t own_stream = nullptr;   // the one vp8hip_create made
    int last_hip_error = 0;

    uint8_t *pixel_pool = nullptr;  // one allocation for every surface
    FrameSurf frames[NFRAMES];
    Frame cur;
    int slot[3] = {-1, -1, -1};     // pool index of LAST / GOLDEN / ALTREF
    int recon = -1;                 // pool index of the reconstruction being produced
    bool recon_ready = false;       // holds an unfiltered reconstruction
    bool cur_pyramid_valid = false;
    Frame cur_prev;                 // the previous current frame (the two surfaces swap on every upload)
    int cur_count = 0;              // current frames received so far
    uint32_t *d_stats = nullptr;    // reductions of kernels_rc.hip: the block in force (one of d_stats2), travels with d_sd
    uint32_t *d_stats2[2] = {nullptr, nullptr};
    vp8hip_batch *batch = nullptr;  // the batch this context is a member of
    // check_SSIM without the host round trip (vp8hip_check_ssim_async): the verdict lands in host memory the device writes
    int32_t *h_verdict = nullptr;   // {replaced, new_SSIM, min SSIM, time-out flag, filter updated, seq}: polled, no event
    uint32_t verdict_seq = 0;       // the seq the loop filter launch that carries the verdict will write last
    bool verdict_pending = false;   // that launch is enqueued
    hipStream_t verdict_stream = nullptr;   // ... on this stream (with vp8hip_filter_overlap not the one the context is on afterwards)
    bool chk_armed = false;         // vp8hip_check_ssim_async ran: the next loop filter launch carries the verdict
    int32_t chk_refqi[4] = {0, 0, 0, 0};
    int chk_qi_min = 0;
    unsigned intra_gen = 0;         // launches on intra_prog (its counters carry the launch number: nothing to clear)

    NetSet nets{};
    MBOut out{};
    SegData *d_sd = nullptr;        // the segment data in force (one of d_sd2)
    SegData *d_sd2[2] = {nullptr, nullptr};
    const SegData *lf_sd = nullptr; // what the loop filter in flight on lf_stream reads: the next frame's data go to the other buffer
    SegData *h_sd_ring = nullptr;   // pinned staging for vp8hip_set_segments
    unsigned sd_ring_pos = 0;
    int32_t *d_progress = nullptr;
    // a frame's reference searches on several devices (vp8hip_shard_*): this context's communicator
    ncclComm_t shard_comm = nullptr;
    int shard_rank = 0, shard_world = 1;
    void *d_lf_handoff = nullptr;   // loop filter form 4: a band's bottom rows on their way to the next band (tagged granules)
    unsigned lf_launches = 0;       // window index of the loop filter's never-reset band counters
    int src_w = 0, src_h = 0;       // vp8hip_set_source_size: size of the planes handed over as current frames (0 = coded size)
    int conformant = 0;             // vp8hip_conformant_stream (NOT the reference; off by default)
    int lf_stall_test = 0;          // test hook (vp8hip_debug_lf_stall): make the next loop filters / intra wavefronts time out
    void *scratch = nullptr;        // device staging for debug pyramid downloads
    // coefficient entropy stage: per-block flags and third contexts, token counts per partition, probabilities
    uint8_t *ent_flags = nullptr, *ent_third = nullptr;
    uint32_t *ent_counts = nullptr, *ent_probs = nullptr, *ent_denom0 = nullptr;
    int ent_counted_partitions = 0; // partitions of the vp8hip_count_probs whose block contexts are current (0 = stale)
    EntBuffers ent{};               // boolean coder scratch, allocated on first vp8hip_encode_coefficients
    // vp8hip_filter_overlap: the context has a second stream and the loop filter and whatever does not depend on it run side
    // by side (the entropy stage of the same frame, the next frame's pack / parameter scan / GOLDEN + ALTREF searches).  The
    // FILTER stays on the stream the frame was coded on and the context moves over (`stream` and `lf_stream` trade places
    // in vp8hip_loop_filter and back in join_lf): a video's dependency chain -- LAST search, transform, filter, border, LAST
    // search ... -- is then launches of ONE stream, and the two cross-stream hand-offs (12 us each on this part) are on the
    // side work's path, which has 0.25 ms of slack.  Every entry point that needs the filtered frame joins first (join_lf).
    hipStream_t lf_stream = nullptr;   // the stream `stream` is not
    hipEvent_t ev_fork = nullptr, ev_lf = nullptr;
    hipEvent_t ev_src = nullptr;       // behind the current frame's pack / parameter scan / pyramid on the side stream: see side_sources_done()
    // A batch member's parameter scan (vp8hip_batch_auto_segments) waits here for the frame's longest launch, k_search2's, and rides in it
    // (launch_search2_batch); whoever needs the segment data earlier launches it on its own first (flush_scan).
    bool scan_deferred = false;
    ScanRequest scan_req{};
    bool lf_overlap = false, lf_pending = false;
    bool fork_by_verdict = false;      // the pending filter's launch has no fork event in front of it: see side_stream_ordered()
    bool fork_by_verdict_at_launch = false;   // ... as it was launched (fork_by_verdict is cleared once the ordering is established)
    int64_t lf_context_switches = 0;   // see vp8hip_profile_context_switches
    bool s2_clock_on = false;          // k_search2 stamps its launches (vp8hip_profile_search2_clock)
    bool frame_pending = false;     // between vp8hip_encode_frame_begin and _end
    bool frame_overflowed = false;  // ... and _end found the caller's buffer too small: the coded frame waits in h_frame for a retry
    hipEvent_t frame_event = nullptr;   // the end of the pending frame's entropy stage when it ran beside the chain (a batch's second stream, ent_stream)
    // vp8hip_filter_overlap: the entropy stage of a frame on a THIRD stream, beside its loop filter and beside the next frame's side
    // work -- a caller may start the next frame between vp8hip_encode_frame_begin and _end (the chain waits for the stage before
    // anything overwrites what it reads)
    hipStream_t ent_stream = nullptr;
    hipEvent_t ev_ent = nullptr;
    bool ent_pending = false;           // the chain has not yet been told to wait for ev_ent
    unsigned out_gen = 0, frame_gen = 0;   // frames whose results went into `out` so far / when the pending frame's stage was enqueued
    bool counted = false;           // in g_live_contexts
    vp8hip_header_params frame_params{};
    int frame_partitions = 0;
    int ent_bools_per_block = 64;   // what that scratch is sized for; doubled (up to 304, the maximum) when a frame needs more
    // host intra path on the device: sub-block modes, replaced flags, row progress, {replaced, new_SSIM, min SSIM}
    int32_t *intra_modes = nullptr, *intra_is_inter = nullptr, *intra_prog = nullptr, *intra_stats = nullptr;
    // first partition on the device: its own coder scratch, per-workgroup statistics, probability table, {H, skip_prob, replaced}
    EntBuffers hdr{};
    uint32_t *hdr_partial = nullptr, *hdr_info = nullptr;
    uint8_t *hdr_sym = nullptr;
    uint8_t *h_frame = nullptr;     // pinned staging of vp8hip_encode_frame's read-back (pageable targets serialise inside the runtime)
    uint8_t *d_frame = nullptr;     // the frame as gathered on the device: [0] size, [1] first-partition size, bytes from +16
    size_t h_frame_cap = 0;

    uint32_t prof_mask = 0;
    hipEvent_t ev[MAX_EVENTS];
    int ev_kernel[MAX_EVENTS / 2];
    int ev_used = 0;
    int ev_made = 0;                // ev[0 .. ev_made) are taken from the process's pool so far (event_pool_get)
    double prof_ms[VP8HIP_K_COUNT] = {0};
    int64_t prof_n[VP8HIP_K_COUNT] = {0};
};

namespace vp8 {
thread_local LaunchTiming tl_timing;
}

namespace {

#define HIPCHK(c, call)                                  \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) {                          \
            (c)->last_hip_error = (int)e_;               \
            return VP8HIP_ERR_HIP;                       \
        }                                                \
    } while (0)

size_t plane_bytes(int w, int h, int *stride) {
    *stride = (w + 2 * PAD + 63) / 64 * 64;
    return (size_t)(*stride) * (h + 2 * PAD);
}

// carve one plane out of the pool; returns the advanced cursor
uint8_t *carve(uint8_t *cursor, int w, int h, Plane *pl) {
    int stride;
    const size_t bytes = plane_bytes(w, h, &stride);
    pl->p = cursor + (size_t)PAD * stride + PAD;
    pl->stride = stride;
    pl->w = w;
    pl->h = h;
    return cursor + (bytes + 255) / 256 * 256;
}

size_t frame_bytes(int W, int H) {
    size_t n = 0;
    int s;
    for (int l = 0; l < 5; ++l) n += (plane_bytes(W >> l, H >> l, &s) + 255) / 256 * 256;
    n += 2 * ((plane_bytes(W / 2, H / 2, &s) + 255) / 256 * 256);
    return n;
}

uint8_t *carve_frame(uint8_t *cursor, int W, int H, Frame *f) {
    for (int l = 0; l < 5; ++l) cursor = carve(cursor, W >> l, H >> l, &f->Y[l]);
    cursor = carve(cursor, W / 2, H / 2, &f->U);
    cursor = carve(cursor, W / 2, H / 2, &f->V);
    return cursor;
}

// ---- per-kernel event timing --------------------------------------------------------------------
// Stages that are ONE kernel launch get their two events recorded by the dispatch itself (VP8_LAUNCH, vp8hip_dev.h): the
// kernel's own begin and end.  Stages made of several launches (entropy stage, intra) are bracketed with hipEventRecord.
constexpr uint32_t SINGLE_LAUNCH_STAGES = (1u << VP8HIP_K_PACK) | (1u << VP8HIP_K_DOWNSAMPLE) | (1u << VP8HIP_K_SEARCH1_L4) | (1u << VP8HIP_K_SEARCH1_L3) |
                                          (1u << VP8HIP_K_SEARCH1_L2) | (1u << VP8HIP_K_SEARCH1_L1) | (1u << VP8HIP_K_SEARCH1_L0) | (1u << VP8HIP_K_SEARCH2) |
                                          (1u << VP8HIP_K_MB) | (1u << VP8HIP_K_LOOP_FILTER) | (1u << VP8HIP_K_BORDER);
// The timing events come from a pool of the process (per device) and go back to it: they are never destroyed.  Every context used
// to create 4 096 of them and destroy them with itself -- 200 000 per bench leg -- and a process that had TIMED kernels with them
// (hipExtLaunchKernel's start / stop events) ended inside the runtime in hipEventDestroy once in some twenty legs (segmentation
// fault or `double free or corruption`; scripts/stress_headline_flow.py: cycle 19 of 40 with events, none in 60 without).
static std::mutex g_event_mutex;
static std::vector<hipEvent_t> g_event_pool[64];
static hipEvent_t event_pool_get(int device) {
    std::lock_guard<std::mutex> lock(g_event_mutex);
    std::vector<hipEvent_t> &pool = g_event_pool[device & 63];
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
static void event_pool_put(int device, hipEvent_t *ev, int n) {
    std::lock_guard<std::mutex> lock(g_event_mutex);
    std::vector<hipEvent_t> &pool = g_event_pool[device & 63];
    for (int i = 0; i < n; ++i)
        if (ev[i]) pool.push_back(ev[i]);
}

struct Timed {
    vp8hip_ctx *c;
    int slot = -1;
    bool by_dispatch = false;
    Timed(vp8hip_ctx *ctx, int kernel) : c(ctx) {
        if (!(c->prof_mask & (1u << kernel)) || c->ev_used + 2 > MAX_EVENTS) return;
        while (c->ev_made < c->ev_used + 2) {   // taken when first needed: a context that times nothing holds none
            hipEvent_t e = event_pool_get(c->device);
            if (!e) return;
            c->ev[c->ev_made++] = e;
        }
        slot = c->ev_used;
        c->ev_kernel[slot / 2] = kernel;
        c->ev_used += 2;
        by_dispatch = (SINGLE_LAUNCH_STAGES >> kernel) & 1u;
        if (by_dispatch) {
            tl_timing.start = c->ev[slot];
            tl_timing.stop = c->ev[slot + 1];
            tl_timing.launches = 0;
        } else {
            hipEventRecord(c->ev[slot], c->stream);
        }
    }
    ~Timed() {
        if (slot < 0) return;
        if (by_dispatch) {
            if (tl_timing.launches == 0) c->ev_used -= 2;   // nothing was launched (a pyramid level without a block): give the slot back
            tl_timing = LaunchTiming{};
        } else {
            hipEventRecord(c->ev[slot + 1], c->stream);
        }
    }
};

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

int pick_free_frame(const vp8hip_ctx *c) {
    for (int i = 0; i < NFRAMES; ++i) {
        if (i == c->slot[0] || i == c->slot[1] || i == c->slot[2] || i == c->recon) continue;
        return i;
    }
    return -1;
}

// tight host/device planes -> padded surface
int copy_in(vp8hip_ctx *c, const Plane &dst, const void *src, hipMemcpyKind kind) {
    HIPCHK(c, hipMemcpy2DAsync(dst.p, dst.stride, src, dst.w, dst.w, dst.h, kind, c->stream));
    return VP8HIP_OK;
}
int copy_out(vp8hip_ctx *c, void *dst, const Plane &src) {
    HIPCHK(c, hipMemcpy2DAsync(dst, src.w, src.p, src.stride, src.w, src.h, hipMemcpyDeviceToHost, c->stream));
    return VP8HIP_OK;
}

// sw, sh: size of the planes that come in (0 = the coded size): the current frames of a context with a source size
int set_frame_planes(vp8hip_ctx *c, Frame &f, const void *y, const void *u, const void *v, hipMemcpyKind kind, int sw = 0, int sh = 0) {
    Timed t(c, VP8HIP_K_PACK);
    if (kind == hipMemcpyDeviceToDevice) {
        launch_pack(c->stream, f, y, u, v, sw, sh);
        return VP8HIP_OK;
    }
    if (sw > 0) {
        // the source rectangle into the surface, then copy_with_padding in place: the pack kernel with the surface as its own
        // source (samples inside the rectangle are rewritten with themselves, the rest repeats the rectangle's edge)
        HIPCHK(c, hipMemcpy2DAsync(f.Y[0].p, f.Y[0].stride, y, sw, sw, sh, kind, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(f.U.p, f.U.stride, u, sw / 2, sw / 2, sh / 2, kind, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(f.V.p, f.V.stride, v, sw / 2, sw / 2, sh / 2, kind, c->stream));
        launch_pack(c->stream, f, f.Y[0].p, f.U.p, f.V.p, sw, sh, f.Y[0].stride, f.U.stride);
        return VP8HIP_OK;
    }
    int rc;
    if ((rc = copy_in(c, f.Y[0], y, kind))) return rc;
    if ((rc = copy_in(c, f.U, u, kind))) return rc;
    return copy_in(c, f.V, v, kind);
}

void build_pyramid(vp8hip_ctx *c, Frame *a, Frame *b, uint32_t border_mask = 0) {
    // cascade: every level from the rounded previous level (inter_part.h:11-33), one launch
    Timed t(c, VP8HIP_K_DOWNSAMPLE);
    launch_pyramid(c->stream, a, b, border_mask);
}

int make_last(vp8hip_ctx *c, const void *y, const void *u, const void *v, hipMemcpyKind kind) {
    const int idx = pick_free_frame(c);
    if (idx < 0) return VP8HIP_ERR_STATE;
    int rc = set_frame_planes(c, c->frames[idx].f, y, u, v, kind);
    if (rc) return rc;
    {
        Timed t(c, VP8HIP_K_BORDER);
        launch_border(c->stream, c->frames[idx].f);
    }
    c->frames[idx].pyramid_valid = false;
    c->frames[idx].border_valid = true;
    c->slot[0] = idx;
    return VP8HIP_OK;
}

}  // namespace

// Up to MAX_BATCH contexts of one geometry on one device advance one frame together (see "batched contexts" below).
// `prep` is the batch's second stream: what a frame needs done before its searches but what does not depend on the previous
// frame's reconstruction -- taking the new frame in (pack / copy_with_padding), the parameter scan with the segment data, the
// new frame's pyramid -- runs there, beside the previous frame's chain on `stream`, instead of at the head of the chain behind the
// loop filter.  In the kernel trace of 48 chunks in 8 batches those three launches, a few microseconds of work each, lasted
// 0.2-0.45 ms per frame waiting for their turn: a fifth of the summed kernel time.
struct vp8hip_batch {
    int n = 0;
    vp8hip_ctx *c[MAX_BATCH] = {};
    hipStream_t stream = nullptr;
    hipStream_t prep = nullptr;          // nullptr: everything on `stream` (VP8HIP_BATCH_PREP=0)
    bool prep_shared = false;            // prep is the process-wide one (VP8HIP_BATCH_PREP=2), not this batch's to destroy
    hipEvent_t ev_gate = nullptr;        // on `stream`, at the start of a frame call: everything of the earlier frames
    hipEvent_t ev_gate2 = nullptr;       // (the two alternate: a wait never names an event that is recorded again right behind it)
    hipEvent_t ev_prep = nullptr;        // on `prep`: the head-of-frame work enqueued so far
    bool prep_pending = false;           // `stream` has not yet been told to wait for ev_prep
    // The entropy stage of the members' frames beside their loop filter (vp8hip_batch_encode_frame_begin): a second stream, forked
    // from `stream` right before the filter's launch (ev_ent_fork) and joined back behind the stage (ev_ent).
    hipStream_t ent = nullptr;           // nullptr: the stage stays in the chain (VP8HIP_BATCH_ENT_STREAM=0)
    hipEvent_t ev_ent_fork = nullptr, ev_ent = nullptr;
    bool ent_fork_fresh = false;         // nothing was enqueued for a member since ev_ent_fork was recorded
};

// The event behind which a host thread reads a finished frame out of pinned memory the stage's last kernel wrote: recorded with a
// SYSTEM-scope release.  An event made with hipEventDisableTiming alone releases to the device only, and the host then read, once in
// a few hundred frames, the previous frame's size word (the frame tag's first-partition size was the symptom).
static constexpr unsigned FRAME_EVENT_FLAGS = hipEventDisableTiming | hipEventReleaseToSystem;

static std::atomic<int> g_live_contexts{0};   // contexts that launch on a stream of their own (members of a batch share one)

// Contexts overlap only if their streams sit on different hardware queues, and the HIP runtime multiplexes all streams
// of a process onto GPU_MAX_HW_QUEUES queues (default 4), read once at its first call: nothing the library can set.
// A host that runs more contexts on streams of their own than there are queues gets a one-line note (VP8HIP_QUIET=1
// silences it).
static void note_queue_oversubscription() {
    static std::atomic<bool> warned{false};
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    const int queues = q ? atoi(q) : 4, n = g_live_contexts.load();
    if (n > queues && !getenv("VP8HIP_QUIET") && !warned.exchange(true))
        fprintf(stderr, "vp8hip: %d contexts on streams of their own but GPU_MAX_HW_QUEUES=%d hardware queues: the streams will share queues and "
                        "serialise (measured on MI355X with 32 contexts in 8 batches: 37 M MB/s at 4 queues, 55 M at 16).  Export "
                        "GPU_MAX_HW_QUEUES=16 before the process makes its first HIP call -- not more: beyond 24 queues per process the "
                        "hardware scheduler rotates them and context-switches running waves -- and advance the contexts in batches "
                        "(vp8hip_batch_create).\n", n, queues);
}

extern "C" {

// New segment data while the previous frame's loop filter is still in flight on lf_stream (it reads its own frame's data):
// they go to the other buffer.  Consumers on the context's stream are ordered behind the write anyway.
static SegData *sd_for_writing(vp8hip_ctx *c) {
    if (c->lf_pending && c->lf_sd == c->d_sd) c->d_sd = c->d_sd == c->d_sd2[0] ? c->d_sd2[1] : c->d_sd2[0];
    return c->d_sd;
}
// The parameter scan of a new frame (vp8hip_auto_segments) always takes the OTHER pair of blocks -- segment data and the
// strength words beside them -- and makes it the one in force: every consumer gets its pointers when it is enqueued, so the
// previous frame's loop filter and entropy stage keep reading theirs while this frame's scan already runs (on the second
// stream of vp8hip_filter_overlap, or on a batch's head-of-frame stream).
static void next_params(vp8hip_ctx *c) {
    c->d_sd = c->d_sd == c->d_sd2[0] ? c->d_sd2[1] : c->d_sd2[0];
    c->d_stats = c->d_stats == c->d_stats2[0] ? c->d_stats2[1] : c->d_stats2[0];
}

// work enqueued on the context's stream from here on sees the filtered reconstruction
// work enqueued on the context's stream from here on may overwrite what the previous frame's entropy stage (on its own stream) reads
static int join_ent(vp8hip_ctx *c) {
    if (!c->ent_pending) return VP8HIP_OK;
    c->ent_pending = false;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_ent, 0));
    return VP8HIP_OK;
}
// The join in two halves: the context goes back to the stream the filter is on (what is enqueued from then on runs behind the
// FILTER), and later that stream is told to wait for everything that ran beside the filter.  vp8hip_inter_transform puts the new
// LAST's pyramid and replicated edges -- which need the filter's output and nothing of the side work -- between the two: the
// barrier packet of the wait is then evaluated while that kernel runs instead of between the filter and it (11 us per frame).
static hipStream_t join_lf_swap(vp8hip_ctx *c) {
    c->lf_pending = false;
    c->fork_by_verdict = false;
    // The streams trade places first: whatever the calls after this return, `stream` is the one vp8hip_create made again
    // (vp8hip_destroy relies on it)
    hipStream_t side = c->stream;
    c->stream = c->lf_stream;
    c->lf_stream = side;
    return side;
}
static int join_lf_wait(vp8hip_ctx *c, hipStream_t side) {
    HIPCHK(c, hipEventRecord(c->ev_lf, side));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_lf, 0));
    return VP8HIP_OK;
}
// The LAST search needs of the side stream's work only the head: the current frame packed, scanned and downsampled.  The host
// is far ahead of the device there (it enqueues the chain while the filter still has 0.1 ms to run), so it can simply LOOK: once
// the event behind the head has been seen complete, launches enqueued from then on start after it whatever their stream, and the
// chain needs no barrier packet between the new LAST's pyramid and the LAST search (10 us per 1080p frame; the GOLDEN / ALTREF
// searches are waited for in front of k_mb, by the one barrier packet that is there anyway).  Gives up after 20 us.
static bool side_sources_done(vp8hip_ctx *c) {
    for (int spins = 0; spins < 64; ++spins) {
        const hipError_t q = hipEventQuery(c->ev_src);
        if (q == hipSuccess) return true;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return false; }
        for (int i = 0; i < 16; ++i) __builtin_ia32_pause();
    }
    return false;
}
static int join_lf(vp8hip_ctx *c, bool defer_ent = false) {
    if (!c->lf_pending) return defer_ent ? VP8HIP_OK : join_ent(c);
    // back to the stream the filter is on, behind it and behind everything that ran beside it
    hipStream_t side = join_lf_swap(c);
    { const int wr = join_lf_wait(c, side); if (wr) return wr; }
    // ... and behind the previous frame's entropy stage: what follows may overwrite the results it reads.  vp8hip_inter_transform
    // defers that wait to the one kernel of its chain that does (k_mb): the LAST search does not have to stand behind the stage.
    return defer_ent ? VP8HIP_OK : join_ent(c);
}
static void lf_check(vp8hip_ctx *c, LfCheck &k);   // (below, with check_SSIM)

static void flush_scan(vp8hip_ctx *c);
// work enqueued on the batch's stream from here on sees what its head-of-frame stream has been given so far
static void batch_join_prep(vp8hip_batch *b) {
    if (!b) return;
    for (int i = 0; i < b->n; ++i) flush_scan(b->c[i]);   // (a member's parameter scan still waiting for its search launch: ahead of whatever comes now)
    b->ent_fork_fresh = false;   // (every entry point that may enqueue passes here: the entropy stage's early fork point is stale)
    if (!b->prep || !b->prep_pending) return;
    b->prep_pending = false;
    (void)hipEventRecord(b->ev_prep, b->prep);
    (void)hipStreamWaitEvent(b->stream, b->ev_prep, 0);
}
// HIP's current device is per host thread: a context may be driven from a thread other than its creator's, or two contexts
// on two GPUs from one thread -- every entry point that may allocate or use the null stream selects the context's device.
// A member of a batch launches on the batch's stream: whatever it does there comes after the batch's head-of-frame work.
// vp8hip_filter_overlap: the side stream's work must start behind everything the frame's chain enqueued BEFORE the filter (the next
// frame's GOLDEN / ALTREF searches overwrite nets the frame's k_mb reads).  An event recorded in front of the filter's launch says
// so on the device -- and costs the chain 12 us per frame: a marker packet with a completion signal between k_mb and the filter
// (0.375 -> 0.362 ms per 1080p frame without it).  When the filter's launch carries check_SSIM's verdict, the verdict itself is
// the proof: its sequence number arrives in host memory from INSIDE that launch, and a launch starts when everything before it on
// its stream has completed.  So no event is recorded then, and the first entry point that would enqueue on the side stream makes
// sure the number is there (the native frame loop has taken the verdict by then anyway: nothing waits).
static void side_stream_ordered(vp8hip_ctx *c) {
    if (!c->lf_pending || !c->fork_by_verdict) return;
    const uint32_t want = c->verdict_seq;
    for (unsigned spins = 0; (uint32_t)__atomic_load_n(&c->h_verdict[5], __ATOMIC_ACQUIRE) != want; ++spins) {
        if ((spins & 0xfff) == 0xfff && hipStreamQuery(c->lf_stream) != hipErrorNotReady) break;   // the filter's stream is idle: it has run (or failed; the next call says so)
        __builtin_ia32_pause();
    }
    c->fork_by_verdict = false;
}
static void flush_scan(vp8hip_ctx *c) {
    if (!c->scan_deferred) return;
    c->scan_deferred = false;
    const ScanRequest &q = c->scan_req;
    launch_auto_segments(c->stream, c->cur, q.partial, q.stats, q.sd, q.strength_out, q.is_key, q.refqi, q.qi_min);
}
#define USE_DEVICE(c) do { if (c) { (void)hipSetDevice((c)->device); batch_join_prep((c)->batch); side_stream_ordered(c); flush_scan(c); } } while (0)
#define USE_DEVICE_ONLY(c) do { if (c) (void)hipSetDevice((c)->device); } while (0)
#define JOIN_LF(c) do { if (c) { const int jr_ = join_lf(c); if (jr_) return jr_; } } while (0)

int vp8hip_filter_overlap(vp8hip_ctx *c, int on) {
    USE_DEVICE(c);
    if (!c) return VP8HIP_ERR_ARG;
    JOIN_LF(c);
    if (on && !c->lf_stream) {
        // A priority class of its own: the runtime maps streams to a few hardware queues by its own bookkeeping, and the
        // two streams of a context on ONE queue serialise the side work into the chain (0.57 instead of 0.45 ms per 1080p
        // frame; seen after other contexts' streams had come and gone in the same process).  Queues are per priority, and
        // the side work -- which has a quarter of a millisecond of slack -- is the one to yield.
        int least = 0, greatest = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(c, hipStreamCreateWithPriority(&c->lf_stream, hipStreamNonBlocking, least));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_lf, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_src, hipEventDisableTiming));
    }
    c->lf_overlap = on != 0;
    return VP8HIP_OK;
}

int vp8hip_create(vp8hip_ctx **out, int width, int height, float ssim_target, int device_ordinal) {
    if (!out || width < 16 || height < 16 || (width % 16) || (height % 16) || width > 8192 || height > 8192)
        return VP8HIP_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_ordinal < 0 || device_ordinal >= ndev)
        return VP8HIP_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return VP8HIP_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VP8HIP_ERR_ARCH;
    vp8hip_ctx *c = new (std::nothrow) vp8hip_ctx();
    if (!c) return VP8HIP_ERR_ARG;
    c->W = width;
    c->H = height;
    c->mbw = width / 16;
    c->mbh = height / 16;
    c->mbs = c->mbw * c->mbh;
    c->b8 = c->mbs * 4;
    c->ssim_target = ssim_target;
    c->device = device_ordinal;
#define CR(call)                                   \
    do {                                           \
        hipError_t e_ = (call);                    \
        if (e_ != hipSuccess) {                    \
            c->last_hip_error = (int)e_;           \
            vp8hip_destroy(c);                     \
            return VP8HIP_ERR_HIP;                 \
        }                                          \
    } while (0)
    CR(hipSetDevice(device_ordinal));
    CR(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = c->stream;
    const size_t fb = frame_bytes(width, height);
    CR(hipMalloc(&c->pixel_pool, fb * (NFRAMES + 2)));
    CR(hipMemsetAsync(c->pixel_pool, 0, fb * (NFRAMES + 2), c->stream));
    uint8_t *cur = c->pixel_pool;
    for (int i = 0; i < NFRAMES; ++i) cur = carve_frame(cur, width, height, &c->frames[i].f);
    cur = carve_frame(cur, width, height, &c->cur);
    carve_frame(cur, width, height, &c->cur_prev);
    // [0..3] sums, [4] reductor, [5] sharpness, [6] sharpness in force, [8..] partials; two blocks, like the segment data they
    // belong to: a frame's parameters are produced while the previous frame's are still being read
    const size_t stats_words = 8 + rc_partial_words();
    CR(hipMalloc(&c->d_stats2[0], 2 * stats_words * sizeof(uint32_t)));
    CR(hipMemsetAsync(c->d_stats2[0], 0, 2 * stats_words * sizeof(uint32_t), c->stream));   // (holds a completion counter that is zero at rest)
    c->d_stats2[1] = c->d_stats2[0] + stats_words;
    c->d_stats = c->d_stats2[0];
    CR(hipHostMalloc(&c->h_verdict, 16 * sizeof(int32_t), hipHostMallocCoherent));   // fine-grained: the device's stores arrive while its kernel runs
    memset(c->h_verdict, 0, 16 * sizeof(int32_t));
    for (int r = 0; r < 3; ++r) {
        CR(hipMalloc(&c->nets.net[r][0], (size_t)c->b8 * 4));
        CR(hipMalloc(&c->nets.net[r][1], (size_t)c->b8 * 4));
        CR(hipMalloc(&c->nets.bdiff[r], (size_t)c->b8 * 4));
    }
    CR(hipMalloc(&c->out.parts, (size_t)c->mbs * 4));
    CR(hipMalloc(&c->out.ref, (size_t)c->mbs * 4));
    CR(hipMalloc(&c->out.seg, (size_t)c->mbs * 4));
    CR(hipMalloc(&c->out.nz, (size_t)c->mbs * 4));
    CR(hipMalloc(&c->out.mask, (size_t)c->mbs * 4));
    CR(hipMalloc(&c->out.ssim, (size_t)c->mbs * 4));
    CR(hipMalloc(&c->out.vec, (size_t)c->mbs * 16));
    CR(hipMalloc(&c->out.coeffs, (size_t)c->mbs * 800));
    CR(hipMalloc(&c->out.flags, 64));
    CR(hipMemsetAsync(c->out.flags, 0, 64, c->stream));
    CR(hipMalloc(&c->d_sd2[0], 2 * sizeof(SegData)));
    c->d_sd2[1] = c->d_sd2[0] + 1;
    c->d_sd = c->d_sd2[0];
    CR(hipHostMalloc(&c->h_sd_ring, 16 * sizeof(SegData)));
    CR(hipMalloc(&c->d_progress, (size_t)c->mbh * 4 + 8192));   // band counters (+ diagnostic stamps at +4096, error word)
    CR(hipMemsetAsync(c->d_progress, 0, (size_t)c->mbh * 4 + 8192, c->stream));
    CR(hipMalloc(&c->d_lf_handoff, loop_filter4_handoff_bytes(c->mbw, c->mbh)));
    CR(hipMemsetAsync(c->d_lf_handoff, 0, loop_filter4_handoff_bytes(c->mbw, c->mbh), c->stream));
    CR(hipMemsetAsync(c->d_progress + S2_CLOCK_WORD, 0xff, 8, c->stream));   // k_search2's launch clock: "earliest start" is ~0 at rest
    CR(hipMalloc(&c->scratch, (size_t)width * height));
    CR(hipMalloc(&c->ent_flags, (size_t)c->mbs * 25));
    CR(hipMalloc(&c->ent_third, (size_t)c->mbs * 25));
    CR(hipMemsetAsync(c->ent_third, 0, (size_t)c->mbs * 25, c->stream));
    CR(hipMalloc(&c->ent_counts, sizeof(uint32_t) * ENT_NCTX * 2 * c->mbh * 4));   // four partial histograms per macroblock row
    CR(hipMalloc(&c->ent_probs, sizeof(uint32_t) * ENT_NCTX));
    CR(hipMalloc(&c->ent_denom0, sizeof(uint32_t) * ENT_NCTX));
    CR(hipMalloc(&c->intra_modes, (size_t)c->mbs * 64));
    CR(hipMalloc(&c->intra_is_inter, (size_t)c->mbs * 4));
    CR(hipMalloc(&c->intra_prog, (size_t)c->mbh * 4));
    CR(hipMalloc(&c->intra_stats, 32));
    CR(hipMemsetAsync(c->intra_prog, 0, (size_t)c->mbh * 4, c->stream));
    CR(hipMemsetAsync(c->intra_modes, 0, (size_t)c->mbs * 64, c->stream));
    CR(hipMemsetAsync(c->intra_is_inter, 0, (size_t)c->mbs * 4, c->stream));
    CR(hipMemsetAsync(c->out.parts, 0, (size_t)c->mbs * 4, c->stream));
    CR(hipMemsetAsync(c->out.ref, 0, (size_t)c->mbs * 4, c->stream));
    CR(hipMemsetAsync(c->out.seg, 0, (size_t)c->mbs * 4, c->stream));
    CR(hipMemsetAsync(c->out.nz, 0, (size_t)c->mbs * 4, c->stream));
    CR(hipMemsetAsync(c->out.mask, 0, (size_t)c->mbs * 4, c->stream));
    CR(hipMemsetAsync(c->out.ssim, 0, (size_t)c->mbs * 4, c->stream));
    CR(hipMemsetAsync(c->out.vec, 0, (size_t)c->mbs * 16, c->stream));
    CR(hipMemsetAsync(c->out.coeffs, 0, (size_t)c->mbs * 800, c->stream));
    CR(hipMemsetAsync(c->d_sd2[0], 0, 2 * sizeof(SegData), c->stream));
    c->recon = 0;
    CR(hipStreamSynchronize(c->stream));
#undef CR
    *out = c;
    c->counted = true;
    ++g_live_contexts;
    return VP8HIP_OK;
}

int vp8hip_hw_queues(void) {   // what the runtime was told when the process started (the default is 4)
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    return q ? atoi(q) : 4;
}

void vp8hip_destroy(vp8hip_ctx *c) {
    if (!c) return;
    if (c->counted) --g_live_contexts;
    hipSetDevice(c->device);
    if (c->lf_stream) {
        (void)join_lf(c);   // `stream` is the one vp8hip_create made again
        hipStreamSynchronize(c->lf_stream);
        hipStreamDestroy(c->lf_stream);
        hipEventDestroy(c->ev_fork);
        hipEventDestroy(c->ev_lf);
        hipEventDestroy(c->ev_src);
        if (c->ent_stream) {
            hipStreamSynchronize(c->ent_stream);
            hipStreamDestroy(c->ent_stream);
            hipEventDestroy(c->ev_ent);
        }
    }
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->shard_comm) ncclCommDestroy(c->shard_comm);
    event_pool_put(c->device, c->ev, c->ev_made);
    hipFree(c->pixel_pool);
    for (int r = 0; r < 3; ++r) {
        hipFree(c->nets.net[r][0]);
        hipFree(c->nets.net[r][1]);
        hipFree(c->nets.bdiff[r]);
    }
    hipFree(c->out.parts);
    hipFree(c->out.ref);
    hipFree(c->out.seg);
    hipFree(c->out.nz);
    hipFree(c->out.mask);
    hipFree(c->out.ssim);
    hipFree(c->out.vec);
    hipFree(c->out.coeffs);
    hipFree(c->out.flags);
    hipFree(c->d_sd2[0]);
    if (c->h_sd_ring) hipHostFree(c->h_sd_ring);
    if (c->h_frame) hipHostFree(c->h_frame);
    hipFree(c->d_frame);
    hipFree(c->d_progress);
    hipFree(c->d_lf_handoff);
    hipFree(c->d_stats2[0]);
    if (c->h_verdict) hipHostFree(c->h_verdict);
    hipFree(c->scratch);
    hipFree(c->ent_flags);
    hipFree(c->ent_third);
    hipFree(c->ent_counts);
    hipFree(c->ent_probs);
    hipFree(c->ent_denom0);
    hipFree(c->ent.offs);
    hipFree(c->ent.tile_sum);
    hipFree(c->ent.bools);
    hipFree(c->ent.maps);
    hipFree(c->ent.start);
    hipFree(c->ent.acc);
    hipFree(c->ent.bytes);
    hipFree(c->ent.sizes);
    hipFree(c->ent.plan);
    hipFree(c->intra_modes);
    hipFree(c->intra_is_inter);
    hipFree(c->intra_prog);
    hipFree(c->intra_stats);
    hipFree(c->hdr.offs);
    hipFree(c->hdr.tile_sum);
    hipFree(c->hdr.bools);
    hipFree(c->hdr.maps);
    hipFree(c->hdr.start);
    hipFree(c->hdr.acc);
    hipFree(c->hdr.bytes);
    hipFree(c->hdr.sizes);
    hipFree(c->hdr.plan);
    hipFree(c->hdr_partial);
    hipFree(c->hdr_info);
    hipFree(c->hdr_sym);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

// a new current frame goes into the other of the two surfaces: the previous one stays intact for
// vp8hip_chroma_change (the reference keeps last_U/last_V the same way, encIO.h:207-210)
static void next_current(vp8hip_ctx *c) {
    const Frame t = c->cur;
    c->cur = c->cur_prev;
    c->cur_prev = t;
    c->cur_count++;
    c->cur_pyramid_valid = false;
}

int vp8hip_upload_current(vp8hip_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v) {
    USE_DEVICE(c);
    if (!c || !y || !u || !v) return VP8HIP_ERR_ARG;
    next_current(c);
    int rc = set_frame_planes(c, c->cur, y, u, v, hipMemcpyHostToDevice, c->src_w, c->src_h);
    if (rc) return rc;
    // pageable host memory: the call must not return while the copy still reads the host buffer
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VP8HIP_OK;
}

int vp8hip_set_current_device(vp8hip_ctx *c, const void *y, const void *u, const void *v) {
    USE_DEVICE(c);
    if (!c || !y || !u || !v) return VP8HIP_ERR_ARG;
    next_current(c);
    return set_frame_planes(c, c->cur, y, u, v, hipMemcpyDeviceToDevice, c->src_w, c->src_h);
}

int vp8hip_set_source_size(vp8hip_ctx *c, int src_width, int src_height) {
    if (!c) return VP8HIP_ERR_ARG;
    if (src_width == 0 && src_height == 0) { c->src_w = c->src_h = 0; return VP8HIP_OK; }
    if (src_width <= 0 || src_height <= 0 || (src_width & 1) || (src_height & 1) || src_width > c->W || src_height > c->H ||
        c->W - src_width >= 16 || c->H - src_height >= 16)
        return VP8HIP_ERR_ARG;
    const bool same = src_width == c->W && src_height == c->H;
    c->src_w = same ? 0 : src_width;
    c->src_h = same ? 0 : src_height;
    return VP8HIP_OK;
}

int vp8hip_loopfilter_strength(vp8hip_ctx *c, int32_t *reductor, int32_t *sharpness) {
    USE_DEVICE(c);
    if (!c || !reductor || !sharpness) return VP8HIP_ERR_ARG;
    if (c->cur_count == 0) return VP8HIP_ERR_STATE;
    launch_lf_strength(c->stream, c->cur, c->d_stats + 8, c->d_stats);
    HIPCHK(c, hipGetLastError());
    uint32_t st[2];
    HIPCHK(c, hipMemcpyAsync(st, c->d_stats, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // vp8enc.cpp:100-103, 119-123 on the two sums (the reference's int accumulators, modulo 2^32)
    const int n = c->W * c->H, ni = (c->H - 1) * (c->W - 1);
    int avg = (int32_t)st[0];
    avg += n / 2;
    avg /= n;
    *reductor = (avg * 5 / 255) + 3;
    int div = (int32_t)st[1];
    div += ni / 2;
    div /= ni;
    const int sh = div / 8;
    *sharpness = sh > 7 ? 7 : sh;
    return VP8HIP_OK;
}

int vp8hip_chroma_change(vp8hip_ctx *c, int32_t *Udiff, int32_t *Vdiff) {
    USE_DEVICE(c);
    if (!c || !Udiff || !Vdiff) return VP8HIP_ERR_ARG;
    if (c->cur_count == 0) return VP8HIP_ERR_STATE;
    *Udiff = *Vdiff = 0;
    if (c->cur_count < 2) return VP8HIP_OK;
    launch_chroma_sad(c->stream, c->cur, c->cur_prev, c->d_stats + 8, c->d_stats);
    HIPCHK(c, hipGetLastError());
    uint32_t st[2];
    HIPCHK(c, hipMemcpyAsync(st, c->d_stats + 2, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int nc = (c->W / 2) * (c->H / 2);
    *Udiff = (int32_t)st[0] / nc;      // vp8enc.cpp:277, 284
    *Vdiff = (int32_t)st[1] / nc;
    return VP8HIP_OK;
}

int vp8hip_upload_last(vp8hip_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c || !y || !u || !v) return VP8HIP_ERR_ARG;
    int rc = make_last(c, y, u, v, hipMemcpyHostToDevice);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VP8HIP_OK;
}

int vp8hip_set_last_device(vp8hip_ctx *c, const void *y, const void *u, const void *v) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c || !y || !u || !v) return VP8HIP_ERR_ARG;
    return make_last(c, y, u, v, hipMemcpyDeviceToDevice);
}

int vp8hip_set_segments(vp8hip_ctx *c, const int32_t sd[VP8HIP_SD_INTS]) {
    USE_DEVICE(c);
    if (!c || !sd) return VP8HIP_ERR_ARG;
    // staged through a ring of pinned slots so the call neither keeps the caller's pointer nor stalls
    // the stream (176 bytes per frame; 16 slots cover any realistic number of frames in flight)
    SegData *slot = c->h_sd_ring + (c->sd_ring_pos++ & 15);
    memcpy(slot, sd, sizeof(SegData));
    HIPCHK(c, hipMemcpyAsync(sd_for_writing(c), slot, sizeof(SegData), hipMemcpyHostToDevice, c->stream));
    return VP8HIP_OK;
}

int vp8hip_auto_segments(vp8hip_ctx *c, int is_key_frame, const int32_t refqi[4], int qi_min) {
    USE_DEVICE(c);
    if (!c || !refqi) return VP8HIP_ERR_ARG;
    if (c->cur_count == 0) return VP8HIP_ERR_STATE;
    next_params(c);
    launch_auto_segments(c->stream, c->cur, c->d_stats + 8, c->d_stats, c->d_sd, reinterpret_cast<int32_t *>(c->d_stats + 4),
                         is_key_frame ? 1 : 0, refqi, qi_min);
    HIPCHK(c, hipGetLastError());
    return VP8HIP_OK;
}

int vp8hip_get_segments(vp8hip_ctx *c, int32_t sd[VP8HIP_SD_INTS], int32_t *reductor, int32_t *sharpness) {
    USE_DEVICE(c);
    if (!c || !sd) return VP8HIP_ERR_ARG;
    int32_t rs[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(sd, c->d_sd, sizeof(SegData), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(rs, c->d_stats + 4, sizeof(rs), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (reductor) *reductor = rs[0];
    if (sharpness) *sharpness = rs[1];
    return VP8HIP_OK;
}

// A frame that did not fit its caller's buffer (VP8HIP_ERR_OVERFLOW from vp8hip_encode_frame_end) stays pending for a retry
// with a larger one; a caller that goes on to the next frame instead has given it up.
static void drop_overflowed_frame(vp8hip_ctx *c) {
    if (c->frame_overflowed) c->frame_pending = c->frame_overflowed = false;
    c->chk_armed = false;   // (a new frame begins: a check armed for the previous reconstruction does not ride with this one's filter)
}

// what inter_begin would refuse, without touching the context (a batch validates every member before it changes any)
static int inter_check(const vp8hip_ctx *c, int prev_is_golden, int prev_is_altref, int use_golden, int use_altref) {
    if (c->slot[0] < 0) return VP8HIP_ERR_STATE;
    const int golden = prev_is_golden ? c->slot[0] : c->slot[1], altref = prev_is_altref ? c->slot[0] : c->slot[2];
    if ((use_golden && golden < 0) || (use_altref && altref < 0)) return VP8HIP_ERR_STATE;
    return VP8HIP_OK;   // (a surface for the reconstruction always exists: five surfaces, at most three references)
}

// reference rotation + the reconstruction surface of the new frame: the head of every inter frame
static int inter_begin(vp8hip_ctx *c, int prev_is_golden, int prev_is_altref, int use_golden, int use_altref) {
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

static RefSet ref_set(const vp8hip_ctx *c, int use_last, int use_golden, int use_altref) {
    RefSet refs;
    refs.use[0] = use_last ? 1 : 0;
    refs.use[1] = use_golden ? 1 : 0;
    refs.use[2] = use_altref ? 1 : 0;
    for (int r = 0; r < 3; ++r) refs.ref[r] = c->frames[c->slot[r] >= 0 ? c->slot[r] : c->slot[0]].f;
    return refs;
}

static unsigned long long *s2_clock_words(const vp8hip_ctx *c) { return reinterpret_cast<unsigned long long *>(c->d_progress + S2_CLOCK_WORD); }
// what the launches get: the stamping costs 1.1 % of the headline (same-box A/B, 57.1 against 57.8 M MB/s), so it is on only
// while a host asks for it (vp8hip_profile_search2_clock; bench.py: during its warm-up steps)
static unsigned long long *s2_clock(const vp8hip_ctx *c) { return c->s2_clock_on ? s2_clock_words(c) : nullptr; }

// hierarchical search, inter_part.h:110-236; ping-pong as bound at init.h:672-854.  One launch per level over the
// references in `which` (the reference runs the three references on three queues, inter_part.h:122-135)
static void search_refs(vp8hip_ctx *c, const RefSet &which) {
    hipStream_t s = c->stream;
    const int net_width = c->mbw * 2;
    int src = 0;
    for (int l = 4; l >= 0; --l) {
        Timed t(c, VP8HIP_K_SEARCH1_L4 + (4 - l));
        // one video coded frame after frame (filter on its own stream): nothing else fills the chip, short waves pay
        launch_search1(s, c->cur, which, c->nets, l, src, net_width, c->lf_overlap && l > 0);
        src ^= 1;
    }
    Timed t(c, VP8HIP_K_SEARCH2);
    // (the launch clock only where launches of this context cannot overlap: the one that searches LAST)
    launch_search2(s, c->cur, which, c->nets, which.use[0] ? s2_clock(c) : nullptr);
}

// prepare_GPU_buffers, inter_part.h:1-33 (reset_vectors is folded into k_search1's parent read)
static void pyramids(vp8hip_ctx *c) {
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

int vp8hip_export_search(vp8hip_ctx *c, int ref, void *d_vectors, void *d_costs) {
    USE_DEVICE(c);
    if (!c || ref < 0 || ref > 2 || !d_vectors || !d_costs) return VP8HIP_ERR_ARG;
    HIPCHK(c, hipMemcpyAsync(d_vectors, c->nets.net[ref][0], (size_t)c->b8 * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_costs, c->nets.bdiff[ref], (size_t)c->b8 * 4, hipMemcpyDeviceToDevice, c->stream));
    return VP8HIP_OK;
}

int vp8hip_import_search(vp8hip_ctx *c, int ref, const void *d_vectors, const void *d_costs) {
    USE_DEVICE(c);
    if (!c || ref < 0 || ref > 2 || !d_vectors || !d_costs) return VP8HIP_ERR_ARG;
    HIPCHK(c, hipMemcpyAsync(c->nets.net[ref][0], d_vectors, (size_t)c->b8 * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->nets.bdiff[ref], d_costs, (size_t)c->b8 * 4, hipMemcpyDeviceToDevice, c->stream));
    return VP8HIP_OK;
}

int vp8hip_export_last(vp8hip_ctx *c, void *d_y, void *d_u, void *d_v) {
    USE_DEVICE(c);
    if (!c || !d_y || !d_u || !d_v) return VP8HIP_ERR_ARG;
    JOIN_LF(c);
    if (c->slot[0] < 0) return VP8HIP_ERR_STATE;
    const Frame &f = c->frames[c->slot[0]].f;
    const Plane *pl[3] = {&f.Y[0], &f.U, &f.V};
    void *dst[3] = {d_y, d_u, d_v};
    for (int i = 0; i < 3; ++i)
        HIPCHK(c, hipMemcpy2DAsync(dst[i], pl[i]->w, pl[i]->p, pl[i]->stride, pl[i]->w, pl[i]->h, hipMemcpyDeviceToDevice, c->stream));
    return VP8HIP_OK;
}

// ---- the exchanges of a frame split by reference, inside the library: RCCL on the context's stream, no host synchronisation ----
// (the reference runs the three searches of a frame on three command queues that share only the current frame,
// inter_part.h:122-135, 201-236, and reads the results back at :263-266.)
int vp8hip_shard_unique_id(uint8_t id[VP8HIP_SHARD_ID_BYTES]) {
    static_assert(VP8HIP_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as opaque bytes");
    if (!id) return VP8HIP_ERR_ARG;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return VP8HIP_ERR_HIP;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return VP8HIP_OK;
}

int vp8hip_shard_init(vp8hip_ctx *c, const uint8_t id[VP8HIP_SHARD_ID_BYTES], int rank, int world) {
    USE_DEVICE(c);
    if (!c || !id || world < 1 || world > 3 || rank < 0 || rank >= world) return VP8HIP_ERR_ARG;
    if (c->shard_comm || c->batch) return VP8HIP_ERR_STATE;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    if (ncclCommInitRank(&c->shard_comm, world, u, rank) != ncclSuccess) {
        c->shard_comm = nullptr;
        return VP8HIP_ERR_HIP;
    }
    c->shard_rank = rank;
    c->shard_world = world;
    return VP8HIP_OK;
}

int vp8hip_shard_rank(const vp8hip_ctx *c) { return c && c->shard_comm ? c->shard_rank : -1; }
int vp8hip_shard_world(const vp8hip_ctx *c) { return c && c->shard_comm ? c->shard_world : 0; }

// Every searched reference's quarter-pel vector net and cost net (8 bytes per 8x8 block) from the rank that searched it
// (reference r: rank r mod world) to all ranks, IN PLACE in the nets the searches wrote and vp8hip_inter_finish reads: one
// group of broadcasts = one RCCL launch on the context's stream.
int vp8hip_shard_share_search(vp8hip_ctx *c, int used_mask) {
    USE_DEVICE(c);
    if (!c || (used_mask & ~7)) return VP8HIP_ERR_ARG;
    if (!c->shard_comm) return VP8HIP_ERR_STATE;
    JOIN_LF(c);
    bool ok = ncclGroupStart() == ncclSuccess;
    for (int r = 0; r < 3 && ok; ++r) {
        if (!(used_mask & (1 << r))) continue;
        const int root = r % c->shard_world;
        ok = ncclBroadcast(c->nets.net[r][0], c->nets.net[r][0], (size_t)c->b8, ncclInt32, root, c->shard_comm, c->stream) == ncclSuccess &&
             ncclBroadcast(c->nets.bdiff[r], c->nets.bdiff[r], (size_t)c->b8, ncclInt32, root, c->shard_comm, c->stream) == ncclSuccess;
    }
    ok = (ncclGroupEnd() == ncclSuccess) && ok;
    return ok ? VP8HIP_OK : VP8HIP_ERR_HIP;
}

// The filtered reconstruction of rank `root` (its LAST after vp8hip_loop_filter) becomes every rank's LAST: the three padded
// planes straight out of root's frame pool into a free surface of the others' pools (vp8enc.cpp:395-401 is what the
// reference does with it on one device).  Replicated edges and pyramid are made where the next frame begins, as after a
// loop filter of this context's own.
int vp8hip_shard_share_last(vp8hip_ctx *c, int root) {
    USE_DEVICE(c);
    if (!c || root < 0) return VP8HIP_ERR_ARG;
    if (!c->shard_comm || root >= c->shard_world) return VP8HIP_ERR_STATE;
    JOIN_LF(c);
    int idx;
    if (c->shard_rank == root) {
        idx = c->slot[0];
        if (idx < 0) return VP8HIP_ERR_STATE;
    } else {
        idx = pick_free_frame(c);
        if (idx < 0) return VP8HIP_ERR_STATE;
    }
    const Frame &f = c->frames[idx].f;
    const Plane *pl[3] = {&f.Y[0], &f.U, &f.V};
    bool ok = ncclGroupStart() == ncclSuccess;
    for (int i = 0; i < 3 && ok; ++i) {
        uint8_t *base = pl[i]->p - (size_t)PAD * pl[i]->stride - PAD;       // the plane with its margins: one contiguous piece
        const size_t bytes = (size_t)pl[i]->stride * (pl[i]->h + 2 * PAD);
        ok = ncclBroadcast(base, base, bytes, ncclUint8, root, c->shard_comm, c->stream) == ncclSuccess;
    }
    ok = (ncclGroupEnd() == ncclSuccess) && ok;
    if (!ok) return VP8HIP_ERR_HIP;
    if (c->shard_rank != root) {
        c->frames[idx].pyramid_valid = false;
        c->frames[idx].border_valid = false;
        c->slot[0] = idx;
    }
    return VP8HIP_OK;
}

// barrier + maximum over the ranks of one double (a time), on the context's stream; blocks
int vp8hip_shard_max(vp8hip_ctx *c, double *value) {
    USE_DEVICE(c);
    if (!c || !value) return VP8HIP_ERR_ARG;
    if (!c->shard_comm) return VP8HIP_ERR_STATE;
    double *d = reinterpret_cast<double *>(c->scratch);
    HIPCHK(c, hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (ncclAllReduce(d, d, 1, ncclDouble, ncclMax, c->shard_comm, c->stream) != ncclSuccess) return VP8HIP_ERR_HIP;
    HIPCHK(c, hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VP8HIP_OK;
}

// ---- batched contexts ---------------------------------------------------------------------------------------------------
// Up to MAX_BATCH contexts of one geometry on one device advance one frame together: every stage is ONE launch for all of
// them (vp8hip_dev.h, "batched launches").  The members share the batch's stream, so their own entry points (key frames,
// the entropy stage, downloads) stay ordered with the batched stages.

int vp8hip_batch_create(vp8hip_batch **out, vp8hip_ctx *const *ctxs, int n) {
    if (!out || !ctxs || n < 1 || n > MAX_BATCH) return VP8HIP_ERR_ARG;
    *out = nullptr;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->W != ctxs[0]->W || ctxs[i]->H != ctxs[0]->H || ctxs[i]->device != ctxs[0]->device ||
            ctxs[i]->ssim_target != ctxs[0]->ssim_target || ctxs[i]->lf_overlap || ctxs[i]->conformant != ctxs[0]->conformant ||
            ctxs[i]->src_w != ctxs[0]->src_w || ctxs[i]->src_h != ctxs[0]->src_h)
            return VP8HIP_ERR_ARG;
        if (!ctxs[i]->own_stream) return VP8HIP_ERR_STATE;            // already a member of a batch
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return VP8HIP_ERR_ARG;            // the same context twice
    }
    vp8hip_batch *b = new (std::nothrow) vp8hip_batch();
    if (!b) return VP8HIP_ERR_ARG;
    USE_DEVICE(ctxs[0]);
    b->n = n;
    // Idle streams still take part in the runtime's stream -> hardware queue assignment: with 32 contexts' own streams
    // alive, two of eight batch streams could land on one queue and serialise (a slow mode of 0.21 instead of 0.16 ms per
    // frame in one run out of four).  The members' own streams go, the batch gets a new one -- batches made one after the
    // other then sit on consecutive queues -- and vp8hip_batch_destroy gives every member a stream of its own again.
    for (int i = 0; i < n; ++i) {
        hipStreamSynchronize(ctxs[i]->stream);
        if (ctxs[i]->own_stream) hipStreamDestroy(ctxs[i]->own_stream);
        ctxs[i]->own_stream = nullptr;
    }
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) {
        for (int i = 0; i < n; ++i) {
            hipStreamCreateWithFlags(&ctxs[i]->own_stream, hipStreamNonBlocking);
            ctxs[i]->stream = ctxs[i]->own_stream;
        }
        delete b;
        return VP8HIP_ERR_HIP;
    }
    for (int i = 0; i < n; ++i) {
        ctxs[i]->stream = b->stream;
        ctxs[i]->batch = b;
        b->c[i] = ctxs[i];
        if (ctxs[i]->counted) --g_live_contexts;   // the batch's one stream is counted in their place
        ctxs[i]->counted = false;
    }
    ++g_live_contexts;
    // the head-of-frame stream (VP8HIP_BATCH_PREP): 0 (default) = none, the head of the frame stays at the head of the chain;
    // 1 = one per batch, in the lowest priority class; 2 = one for all batches of the process.  Measured on MI355X, 48 chunks
    // in 8 batches, same box: 62.2 M MB/s without, 59.5 with one per batch (59.9 in the default priority class), 60.9 with one
    // for all -- the second set of queues costs more than the shorter chains win, so it is off unless asked for.
    const int prep_mode = vp8hip_batch_prep_mode();
    bool ok = true;
    if (prep_mode) {
        ok = hipEventCreateWithFlags(&b->ev_gate, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b->ev_gate2, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b->ev_prep, hipEventDisableTiming) == hipSuccess;
        int least = 0, greatest = 0;
        ok = ok && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
        static const bool normal_priority = getenv("VP8HIP_BATCH_PREP_PRIO") != nullptr;   // A/B: the head-of-frame stream in the default class
        if (normal_priority) least = 0;
        if (ok && prep_mode == 2) {
            static hipStream_t shared[64] = {};
            hipStream_t &sh = shared[ctxs[0]->device & 63];
            if (!sh) ok = hipStreamCreateWithPriority(&sh, hipStreamNonBlocking, least) == hipSuccess;
            b->prep = sh;
            b->prep_shared = true;
        } else if (ok) {
            ok = hipStreamCreateWithPriority(&b->prep, hipStreamNonBlocking, least) == hipSuccess;
        }
    }
    if (!ok) {
        vp8hip_batch_destroy(b);
        return VP8HIP_ERR_HIP;
    }
    *out = b;
    return VP8HIP_OK;
}

// A batch's second stream for the entropy stage, made when the first frame is asked for as bytes (a batch that only runs the
// inter path never has one: idle streams still take part in the runtime's stream -> hardware queue assignment).
static bool batch_ent_stream(vp8hip_batch *b) {
    static const int mode = [] { const char *v = getenv("VP8HIP_BATCH_ENT_STREAM"); return v && v[0] ? atoi(v) : 0; }();
    if (!b->ev_ent && (hipEventCreateWithFlags(&b->ev_ent_fork, hipEventDisableTiming) != hipSuccess ||
                       hipEventCreateWithFlags(&b->ev_ent, FRAME_EVENT_FLAGS) != hipSuccess))
        return false;
    if (!mode) return false;
    if (b->ent) return true;
    int least = 0, greatest = 0;
    if (mode >= 2 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) {   // A/B: 2 = in the lowest priority class, 3 = in the highest
        if (hipStreamCreateWithPriority(&b->ent, hipStreamNonBlocking, mode == 3 ? greatest : least) != hipSuccess) b->ent = nullptr;
    } else if (hipStreamCreateWithFlags(&b->ent, hipStreamNonBlocking) != hipSuccess) {
        b->ent = nullptr;
    }
    return b->ent != nullptr;
}

void vp8hip_batch_destroy(vp8hip_batch *b) {   // the contexts stay (destroy the batch before its members), each back on its own stream
    if (!b) return;
    hipSetDevice(b->c[0]->device);
    if (b->prep) hipStreamSynchronize(b->prep);
    hipStreamSynchronize(b->stream);
    hipStreamDestroy(b->stream);
    if (b->prep && !b->prep_shared) hipStreamDestroy(b->prep);
    if (b->ev_gate) hipEventDestroy(b->ev_gate);
    if (b->ev_gate2) hipEventDestroy(b->ev_gate2);
    if (b->ev_prep) hipEventDestroy(b->ev_prep);
    if (b->ent) {
        hipStreamSynchronize(b->ent);
        hipStreamDestroy(b->ent);
    }
    if (b->ev_ent_fork) hipEventDestroy(b->ev_ent_fork);
    if (b->ev_ent) hipEventDestroy(b->ev_ent);
    for (int i = 0; i < b->n; ++i) b->c[i]->frame_event = nullptr;
    --g_live_contexts;
    for (int i = 0; i < b->n; ++i) {
        b->c[i]->batch = nullptr;
        if (!b->c[i]->own_stream) hipStreamCreateWithFlags(&b->c[i]->own_stream, hipStreamNonBlocking);
        b->c[i]->stream = b->c[i]->own_stream;
        if (!b->c[i]->counted) ++g_live_contexts;
        b->c[i]->counted = true;
    }
    delete b;
}

int vp8hip_batch_set_current_device(vp8hip_batch *b, const int *active, const void *const *y, const void *const *u, const void *const *v) {
    if (!b || !y || !u || !v) return VP8HIP_ERR_ARG;
    vp8hip_ctx *c0 = b->c[0];
    USE_DEVICE_ONLY(c0);
    const Frame *f[MAX_BATCH];
    const void *py[MAX_BATCH], *pu[MAX_BATCH], *pv[MAX_BATCH];
    int n = 0;
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        if (!y[i] || !u[i] || !v[i]) return VP8HIP_ERR_ARG;
        if (b->c[i]->src_w != c0->src_w || b->c[i]->src_h != c0->src_h) return VP8HIP_ERR_ARG;   // one launch, one source size
        flush_scan(b->c[i]);      // (a parameter scan of the frame that is being replaced, asked for and never used: on that frame, now)
        next_current(b->c[i]);
        f[n] = &b->c[i]->cur;
        py[n] = y[i]; pu[n] = u[i]; pv[n] = v[i];
        ++n;
    }
    if (!n) return VP8HIP_OK;
    hipStream_t ps = b->stream;
    if (b->prep) {
        // A new frame goes into the surface (and its parameters into the blocks) of the frame before the previous one: all
        // of that frame's work -- enqueued on `stream` before the PREVIOUS frame call began, which is where ev_gate was last
        // recorded -- must be over; the previous frame's chain may still be running, and that is the point.
        HIPCHK(c0, hipStreamWaitEvent(b->prep, b->ev_gate, 0));
        HIPCHK(c0, hipEventRecord(b->ev_gate2, b->stream));
        hipEvent_t t_ = b->ev_gate;
        b->ev_gate = b->ev_gate2;
        b->ev_gate2 = t_;
        b->prep_pending = true;
        ps = b->prep;
    }
    Timed t(c0, VP8HIP_K_PACK);
    launch_pack_batch(ps, f, py, pu, pv, n, c0->src_w, c0->src_h);
    HIPCHK(c0, hipGetLastError());
    return VP8HIP_OK;
}

// `active[i] == 0` leaves context i out of the stage (a chunk whose frame is a key frame goes through its own
// vp8hip_intra_transform); active == NULL means all
int vp8hip_batch_auto_segments(vp8hip_batch *b, const int *active, const int *is_key_frame, const int32_t (*refqi)[4], int qi_min) {
    if (!b || !is_key_frame || !refqi) return VP8HIP_ERR_ARG;
    vp8hip_ctx *c0 = b->c[0];
    USE_DEVICE_ONLY(c0);
    const Frame *cur[MAX_BATCH];
    uint32_t *partial[MAX_BATCH], *stats[MAX_BATCH];
    SegData *sd[MAX_BATCH];
    int32_t *strength[MAX_BATCH];
    int key[MAX_BATCH];
    int32_t qi[MAX_BATCH][4];
    int n = 0;
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        vp8hip_ctx *c = b->c[i];
        if (c->cur_count == 0) return VP8HIP_ERR_STATE;
        next_params(c);
        cur[n] = &c->cur;
        partial[n] = c->d_stats + 8;
        stats[n] = c->d_stats;
        sd[n] = c->d_sd;
        strength[n] = reinterpret_cast<int32_t *>(c->d_stats + 4);
        key[n] = is_key_frame[i] ? 1 : 0;
        for (int k = 0; k < 4; ++k) qi[n][k] = refqi[i][k];
        ++n;
    }
    // The scan rides in k_search2's launch of the same frame (vp8hip_batch_inter_transform, next; kernels_s2.hip says why): with the part
    // full a launch of its own holds the batch's stream for half a millisecond where its work is 15 us (VP8HIP_BATCH_SCAN_LAUNCH=1: as it was)
    static const bool own_launch = [] { const char *v = getenv("VP8HIP_BATCH_SCAN_LAUNCH"); return v && v[0] == '1'; }();
    if (n && !b->prep && !own_launch) {
        int k = 0;
        for (int i = 0; i < b->n; ++i) {
            if (active && !active[i]) continue;
            vp8hip_ctx *c = b->c[i];
            c->scan_req = ScanRequest{partial[k], stats[k], sd[k], strength[k], key[k], {qi[k][0], qi[k][1], qi[k][2], qi[k][3]}, qi_min};
            c->scan_deferred = true;
            ++k;
        }
        return VP8HIP_OK;
    }
    if (n && b->prep) b->prep_pending = true;
    if (n) launch_auto_segments_batch(b->prep ? b->prep : b->stream, cur, partial, stats, sd, strength, key, qi, qi_min, n);
    HIPCHK(c0, hipGetLastError());
    return VP8HIP_OK;
}

int vp8hip_batch_inter_transform(vp8hip_batch *b, const int *active, const int *prev_is_golden, const int *prev_is_altref,
                                 const int *use_golden, const int *use_altref) {
    if (!b || !prev_is_golden || !prev_is_altref || !use_golden || !use_altref) return VP8HIP_ERR_ARG;
    vp8hip_ctx *c0 = b->c[0];
    USE_DEVICE_ONLY(c0);
    vp8hip_ctx *m[MAX_BATCH];
    RefSet refs[MAX_BATCH];
    const Frame *cur[MAX_BATCH], *recon[MAX_BATCH], *pyr[2 * MAX_BATCH], *pyr_cur[MAX_BATCH];
    const ScanRequest *scans[MAX_BATCH] = {};   // the members' parameter scans still waiting for a launch to ride in (vp8hip_batch_auto_segments)
    int npyr_cur = 0;
    const NetSet *nets[MAX_BATCH];
    const MBOut *outs[MAX_BATCH];
    const SegData *sds[MAX_BATCH];
    int n = 0, npyr = 0;
    uint32_t pyr_border = 0;
    for (int i = 0; i < b->n; ++i) {   // every member is checked before any member's state changes
        if (active && !active[i]) continue;
        if (b->c[i]->conformant != c0->conformant) return VP8HIP_ERR_ARG;   // one launch, one predictor
        const int rc = inter_check(b->c[i], prev_is_golden[i], prev_is_altref[i], use_golden[i], use_altref[i]);
        if (rc) return rc;
    }
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        vp8hip_ctx *c = b->c[i];
        const int rc = inter_begin(c, prev_is_golden[i], prev_is_altref[i], use_golden[i], use_altref[i]);
        if (rc) return rc;
        FrameSurf &last = c->frames[c->slot[0]];
        if (!c->cur_pyramid_valid) {
            if (b->prep) pyr_cur[npyr_cur++] = &c->cur;   // the new frame's pyramid: head-of-frame work
            else pyr[npyr++] = &c->cur;
        }
        bool border_alone = false;
        if (!last.pyramid_valid) {
            if (!last.border_valid) pyr_border |= 1u << npyr;
            pyr[npyr++] = &last.f;
        } else if (!last.border_valid) {
            border_alone = true;
        }
        if (border_alone) {
            batch_join_prep(b);
            launch_border(b->stream, last.f);
        }
        last.pyramid_valid = true;
        last.border_valid = true;
        c->cur_pyramid_valid = true;
        refs[n] = ref_set(c, 1, use_golden[i], use_altref[i]);
        cur[n] = &c->cur;
        recon[n] = &c->frames[c->recon].f;
        nets[n] = &c->nets;
        outs[n] = &c->out;
        sds[n] = c->d_sd;
        if (c->scan_deferred) {
            scans[n] = &c->scan_req;
            c->scan_deferred = false;      // (this call launches it: in k_search2's launch, or on its own right behind it)
        }
        m[n++] = c;
    }
    if (!n) return VP8HIP_OK;
    hipStream_t s = b->stream;
    if (npyr_cur) {
        b->prep_pending = true;
        launch_pyramid_batch(b->prep, pyr_cur, npyr_cur, 0);
    }
    batch_join_prep(b);   // the chain starts here: LAST's pyramid and replicated edges, the searches, the transform
    if (npyr) {
        Timed t(c0, VP8HIP_K_DOWNSAMPLE);
        launch_pyramid_batch(s, pyr, npyr, pyr_border);
    }
    const int net_width = c0->mbw * 2;
    int src = 0;
    for (int l = 4; l >= 0; --l) {
        Timed t(c0, VP8HIP_K_SEARCH1_L4 + (4 - l));
        launch_search1_batch(s, cur, refs, nets, l, src, net_width, n);
        src ^= 1;
    }
    bool carried;
    {
        Timed t(c0, VP8HIP_K_SEARCH2);
        carried = launch_search2_batch(s, cur, refs, nets, n, s2_clock(c0), scans);
    }
    for (int i = 0; i < n && !carried; ++i)
        if (scans[i]) launch_auto_segments(s, m[i]->cur, scans[i]->partial, scans[i]->stats, scans[i]->sd, scans[i]->strength_out, scans[i]->is_key,
                                           scans[i]->refqi, scans[i]->qi_min);
    {
        Timed t(c0, VP8HIP_K_MB);
        launch_mb_batch(s, cur, refs, nets, recon, outs, sds, c0->ssim_target, c0->mbw, c0->mbh, n, c0->conformant != 0);
    }
    for (int i = 0; i < n; ++i) m[i]->recon_ready = true;
    HIPCHK(c0, hipGetLastError());
    return VP8HIP_OK;
}

int vp8hip_batch_loop_filter(vp8hip_batch *b, const int *active) {
    if (!b) return VP8HIP_ERR_ARG;
    vp8hip_ctx *c0 = b->c[0];
    USE_DEVICE(c0);   // (joins the head-of-frame stream)
    vp8hip_ctx *m[MAX_BATCH];
    const Frame *recon[MAX_BATCH];
    const MBOut *outs[MAX_BATCH];
    SegData *sds[MAX_BATCH];
    int32_t *prog[MAX_BATCH];
    void *hand[MAX_BATCH];
    unsigned launch_no[MAX_BATCH];
    LfCheck chk[MAX_BATCH];
    int n = 0;
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        if (!b->c[i]->recon_ready || b->c[i]->recon < 0) return VP8HIP_ERR_STATE;
    }
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        vp8hip_ctx *c = b->c[i];
        recon[n] = &c->frames[c->recon].f;
        outs[n] = &c->out;
        sds[n] = c->d_sd;
        prog[n] = c->d_progress;
        hand[n] = c->d_lf_handoff;
        launch_no[n] = c->lf_launches++;
        lf_check(c, chk[n]);
        c->verdict_stream = b->stream;
        m[n++] = c;
    }
    if (!n) return VP8HIP_OK;
    // the entropy stage of these frames may start here (everything it reads has been enqueued), beside the filter
    if (b->ent) HIPCHK(c0, hipEventRecord(b->ev_ent_fork, b->stream));
    {
        Timed t(c0, VP8HIP_K_LOOP_FILTER);
        // (form 3 for batches: with the part full a launch's instructions and LDS count, not its latency; VP8HIP_LF_BATCH_FORM=4 for A/B runs)
        static const bool form4 = [] { const char *v = getenv("VP8HIP_LF_BATCH_FORM"); return v && atoi(v) == 4; }();
        if (form4) launch_loop_filter4_batch(b->stream, recon, outs, sds, prog, hand, c0->mbw, c0->mbh, launch_no, n, chk);
        else launch_loop_filter3_batch(b->stream, recon, outs, sds, prog, c0->mbw, c0->mbh, launch_no, n, chk);
    }
    b->ent_fork_fresh = b->ent != nullptr;
    for (int i = 0; i < n; ++i) {   // the filtered reconstruction is the LAST reference of the next frame (vp8enc.cpp:395-401)
        vp8hip_ctx *c = m[i];
        c->frames[c->recon].pyramid_valid = false;
        c->frames[c->recon].border_valid = false;   // its replicated edges are made with its pyramid, in one launch
        c->slot[0] = c->recon;
        c->recon = -1;
        c->recon_ready = false;
    }
    HIPCHK(c0, hipGetLastError());
    return VP8HIP_OK;
}

int vp8hip_download_results(vp8hip_ctx *c, const vp8hip_results *r) {
    USE_DEVICE(c);
    if (!c || !r) return VP8HIP_ERR_ARG;
    if (!c->recon_ready && (r->recon_Y || r->recon_U || r->recon_V)) return VP8HIP_ERR_STATE;
    hipStream_t s = c->stream;
    const size_t n = c->mbs;
    if (r->MB_parts) HIPCHK(c, hipMemcpyAsync(r->MB_parts, c->out.parts, n * 4, hipMemcpyDeviceToHost, s));
    if (r->MB_reference_frame) HIPCHK(c, hipMemcpyAsync(r->MB_reference_frame, c->out.ref, n * 4, hipMemcpyDeviceToHost, s));
    if (r->MB_vectors) HIPCHK(c, hipMemcpyAsync(r->MB_vectors, c->out.vec, n * 16, hipMemcpyDeviceToHost, s));
    if (r->MB_coeffs) HIPCHK(c, hipMemcpyAsync(r->MB_coeffs, c->out.coeffs, n * 800, hipMemcpyDeviceToHost, s));
    if (r->MB_segment_id) HIPCHK(c, hipMemcpyAsync(r->MB_segment_id, c->out.seg, n * 4, hipMemcpyDeviceToHost, s));
    if (r->MB_SSIM) HIPCHK(c, hipMemcpyAsync(r->MB_SSIM, c->out.ssim, n * 4, hipMemcpyDeviceToHost, s));
    const Frame &f = c->frames[c->recon].f;
    int rc;
    if (r->recon_Y && (rc = copy_out(c, r->recon_Y, f.Y[0]))) return rc;
    if (r->recon_U && (rc = copy_out(c, r->recon_U, f.U))) return rc;
    if (r->recon_V && (rc = copy_out(c, r->recon_V, f.V))) return rc;
    HIPCHK(c, hipStreamSynchronize(s));
    return VP8HIP_OK;
}

int vp8hip_upload_mb_data(vp8hip_ctx *c, const int16_t *coeffs, const int32_t *parts, const int32_t *seg) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c) return VP8HIP_ERR_ARG;
    c->ent_counted_partitions = 0;
    hipStream_t s = c->stream;
    const size_t n = c->mbs;
    if (coeffs) HIPCHK(c, hipMemcpyAsync(c->out.coeffs, coeffs, n * 800, hipMemcpyHostToDevice, s));
    if (parts) HIPCHK(c, hipMemcpyAsync(c->out.parts, parts, n * 4, hipMemcpyHostToDevice, s));
    if (seg) HIPCHK(c, hipMemcpyAsync(c->out.seg, seg, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return VP8HIP_OK;
}

int vp8hip_upload_recon(vp8hip_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c || !y || !u || !v) return VP8HIP_ERR_ARG;
    if (c->recon < 0 || c->recon == c->slot[0] || c->recon == c->slot[1] || c->recon == c->slot[2]) {
        c->recon = -1;
        c->recon = pick_free_frame(c);
        if (c->recon < 0) return VP8HIP_ERR_STATE;
    }
    int rc = set_frame_planes(c, c->frames[c->recon].f, y, u, v, hipMemcpyHostToDevice);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->recon_ready = true;
    return VP8HIP_OK;
}

static int claim_recon(vp8hip_ctx *c) {
    if (c->recon < 0 || c->recon == c->slot[0] || c->recon == c->slot[1] || c->recon == c->slot[2]) {
        c->recon = -1;
        c->recon = pick_free_frame(c);
        if (c->recon < 0) return VP8HIP_ERR_STATE;
    }
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

// ---- check_SSIM without the host round trip ---------------------------------------------------------------------------------
static void check_item(vp8hip_ctx *c, CheckItem &it, const int32_t refqi[4], int qi_min) {
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
static void lf_check(vp8hip_ctx *c, LfCheck &k) {
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
static bool fallback_possible(float ssim_target) {
    static const bool always = [] { const char *v = getenv("VP8HIP_ALWAYS_LAUNCH_FALLBACK"); return v && v[0] == '1'; }();
    return always || ssim_target > -1.0f;
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

int vp8hip_batch_check_ssim_async(vp8hip_batch *b, const int *active, const int32_t (*refqi)[4], int qi_min) {
    if (!b || !refqi) return VP8HIP_ERR_ARG;
    vp8hip_ctx *c0 = b->c[0];
    USE_DEVICE(c0);
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        const vp8hip_ctx *c = b->c[i];
        if (!c->recon_ready || c->recon < 0 || c->cur_count == 0 || c->verdict_pending || c->chk_armed) return VP8HIP_ERR_STATE;
    }
    CheckItem it[MAX_BATCH];
    int n = 0;
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        check_item(b->c[i], it[n++], refqi[i], qi_min);
    }
    if (!n) return VP8HIP_OK;
    if (fallback_possible(c0->ssim_target)) {
        Timed t(c0, VP8HIP_K_INTRA);
        launch_check_fallback(b->stream, it, n, c0->ssim_target, c0->mbw, c0->mbh, c0->conformant);
    }
    HIPCHK(c0, hipGetLastError());
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

// scratch of the boolean coder, allocated the first time the stage is used (a context that only runs the
// inter path never pays for it): 64 bools per 4x4 block on average (of at most 304)
static void ent_free(vp8hip_ctx *c) {
    EntBuffers &e = c->ent;
    hipFree(e.offs); hipFree(e.tile_sum); hipFree(e.bools); hipFree(e.maps); hipFree(e.start); hipFree(e.acc); hipFree(e.bytes);
    hipFree(e.sizes); hipFree(e.plan);
    e = EntBuffers{};
    if (c->h_frame) hipHostFree(c->h_frame);   // sized from the scratch: reallocated with it
    hipFree(c->d_frame);
    c->h_frame = nullptr;
    c->d_frame = nullptr;
}

static int ent_alloc(vp8hip_ctx *c) {
    if (c->ent.plan) return VP8HIP_OK;   // the last allocation below: set only when all of them succeeded
    if (c->ent.offs) ent_free(c);        // a partial allocation left by an earlier failure
    EntBuffers &e = c->ent;
    const size_t nslots = (size_t)c->mbs * 25;
    e.cap_bools = (uint32_t)(nslots * (size_t)c->ent_bools_per_block);
    e.cap_chunks = e.cap_bools / 256 + 2 * ENT_MAX_PARTITIONS;
    e.cap_words = (uint32_t)(((size_t)e.cap_bools * 7 + 31) / 32 + 8 * ENT_MAX_PARTITIONS);
    HIPCHK(c, hipMalloc(&e.offs, (nslots + 1) * 4));
    HIPCHK(c, hipMalloc(&e.tile_sum, (nslots / 256 + 8) * 4));   // the frame path sums per 256 slots
    HIPCHK(c, hipMalloc(&e.bools, (size_t)e.cap_bools * 2 + 1024));
    HIPCHK(c, hipMalloc(&e.maps, ent_maps_entries(e.cap_chunks) * 4));
    HIPCHK(c, hipMalloc(&e.start, (size_t)e.cap_chunks * 8));
    HIPCHK(c, hipMalloc(&e.acc, (size_t)e.cap_words * 8));
    HIPCHK(c, hipMalloc(&e.bytes, (size_t)e.cap_words * 4));
    HIPCHK(c, hipMalloc(&e.sizes, ENT_MAX_PARTITIONS * 4));
    HIPCHK(c, hipMalloc(&e.plan, sizeof(EntPlan)));
    return VP8HIP_OK;
}

// A frame denser than the scratch was sized for (64 bools per 4x4 block to begin with): double it, up to the 304 bools a
// block can produce at most, so that no frame is ever refused for the device's sake.  false = already at the maximum.
static int ent_grow(vp8hip_ctx *c) {
    if (c->ent_bools_per_block >= 304) return VP8HIP_ERR_OVERFLOW;
    hipStreamSynchronize(c->stream);
    ent_free(c);
    c->ent_bools_per_block = c->ent_bools_per_block * 2 > 304 ? 304 : c->ent_bools_per_block * 2;
    const int rc = ent_alloc(c);       // VP8HIP_ERR_HIP (e.g. out of memory) is reported as such, not as an overflow
    if (rc) ent_free(c);
    return rc;
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

static int hdr_alloc(vp8hip_ctx *c) {
    if (c->hdr.bools) return VP8HIP_OK;
    EntBuffers &e = c->hdr;
    const size_t n = (size_t)c->mbs;
    e.cap_bools = (uint32_t)(n * 128 + 16384);   // a macroblock header is at most ~125 bools, the frame header < 10 k
    e.cap_chunks = e.cap_bools / 256 + 4;
    e.cap_words = (uint32_t)(((size_t)e.cap_bools * 7 + 31) / 32 + 16);
    HIPCHK(c, hipMalloc(&e.offs, (n + 1) * 4));
    HIPCHK(c, hipMalloc(&e.tile_sum, (n / 1024 + 8) * 4));
    HIPCHK(c, hipMalloc(&e.bools, (size_t)e.cap_bools * 2 + 1024));
    HIPCHK(c, hipMalloc(&e.maps, ent_maps_entries(e.cap_chunks) * 4));
    HIPCHK(c, hipMalloc(&e.start, (size_t)e.cap_chunks * 8));
    HIPCHK(c, hipMalloc(&e.acc, (size_t)e.cap_words * 8));
    HIPCHK(c, hipMalloc(&e.bytes, (size_t)e.cap_words * 4));
    HIPCHK(c, hipMalloc(&e.sizes, ENT_MAX_PARTITIONS * 4));
    HIPCHK(c, hipMalloc(&e.plan, sizeof(EntPlan)));
    HIPCHK(c, hipMalloc(&c->hdr_partial, HDR_STAT_WORDS * 4));   // the census of k_hdr_count: zero at rest (k_hdr_frame clears it)
    HIPCHK(c, hipMemsetAsync(c->hdr_partial, 0, HDR_STAT_WORDS * 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // once per context: the stage may run on another stream than this one (a batch's, the third)
    HIPCHK(c, hipMalloc(&c->hdr_info, 16));
    HIPCHK(c, hipMalloc(&c->hdr_sym, 64));
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

// The coder's last kernel writes the finished frame straight into the pinned host buffer (device-visible, a few tens of
// KiB over PCIe) instead of into device memory followed by a copy command: one operation fewer in every frame's chain.
// Same-box A/B (VP8HIP_FRAME_ZEROCOPY=0 brings the copy back): 4 050 vs 3 880 frames/s at 1080p, 1 198 vs 1 128 at 4K.
static bool frame_zero_copy() {
    static const bool on = [] { const char *v = getenv("VP8HIP_FRAME_ZEROCOPY"); return !(v && v[0] == '0'); }();
    return on;
}
// Everything of a frame's entropy stage up to the read-back, enqueued; nothing waits.
static constexpr size_t FRAME_FIRST_COPY = 192 * 1024;
// buffers + the description of one context's frame for the entropy stage's launchers
static int frame_prepare(vp8hip_ctx *c, int P, const vp8hip_header_params *p, FrameEntropy &e, FrameOut &fo) {
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
static int frame_enqueue(vp8hip_ctx *c, int P, const vp8hip_header_params *p) {
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

// The entropy stage of the frames of a batch's members in the same nine launches (blockIdx.z = member; the coder takes the
// members' bool strings as job pairs).  Every active member is then between _begin and _end: vp8hip_encode_frame_end
// per member reads its frame back (and, should a frame have been denser than the coder's scratch, codes that one again
// on its own).
int vp8hip_batch_encode_frame_begin(vp8hip_batch *b, const int *active, int num_partitions, const vp8hip_header_params *params) {
    if (!b || !params) return VP8HIP_ERR_ARG;
    const int P = num_partitions;
    if (P != 1 && P != 2 && P != 4 && P != 8) return VP8HIP_ERR_ARG;
    vp8hip_ctx *c0 = b->c[0];
    bool early = b->ent_fork_fresh;     // (read before USE_DEVICE, which marks it stale for whatever this call enqueues)
    USE_DEVICE(c0);
    FrameEntropy e[MAX_BATCH];
    FrameOut fo[MAX_BATCH];
    vp8hip_ctx *m[MAX_BATCH];
    int n = 0;
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        if (b->c[i]->frame_pending) return VP8HIP_ERR_STATE;
    }
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        vp8hip_ctx *c = b->c[i];
        const int rc = frame_prepare(c, P, &params[i], e[n], fo[n]);
        if (rc) return rc;
        c->frame_params = params[i];
        c->frame_partitions = P;
        m[n++] = c;
    }
    if (!n) return VP8HIP_OK;
    // Beside the loop filter, on the batch's second stream: the stage reads the frame's coefficients, modes and vectors (final
    // before the filter's launch: ev_ent_fork) and the segment data check_SSIM may have updated INSIDE that launch -- so only a
    // caller that has taken every member's verdict (vp8hip_check_ssim_result: the update is then in memory) gets the early
    // start; otherwise, and whenever anything was enqueued for the batch since the filter, the stage starts behind all of it.
    // The chain waits for the stage before it goes on (the next frame's k_mb overwrites what the stage reads).
    const bool side = !c0->prof_mask && batch_ent_stream(b);
    hipStream_t s = side ? b->ent : b->stream;
    if (side) {
        for (int i = 0; i < n; ++i) early = early && !m[i]->verdict_pending;
        if (!early) HIPCHK(c0, hipEventRecord(b->ev_ent_fork, b->stream));
        HIPCHK(c0, hipStreamWaitEvent(b->ent, b->ev_ent_fork, 0));
    }
    {
        Timed t(c0, VP8HIP_K_ENT_COUNT);
        launch_fe_count_batch(s, e, n);
    }
    {
        Timed t(c0, VP8HIP_K_HDR_ENCODE);
        launch_fe_emit_batch(s, e, n);
    }
    {
        Timed t(c0, VP8HIP_K_ENT_ENCODE);
        launch_frame_code_batch(s, e, fo, n);
    }
    HIPCHK(c0, hipGetLastError());
    for (int i = 0; i < n; ++i) {
        vp8hip_ctx *c = m[i];
        c->ent_counted_partitions = P;
        if (!frame_zero_copy()) {
            const size_t first = c->h_frame_cap < FRAME_FIRST_COPY ? c->h_frame_cap : FRAME_FIRST_COPY;
            HIPCHK(c0, hipMemcpyAsync(c->h_frame, c->d_frame, first, hipMemcpyDeviceToHost, s));
        }
        c->frame_pending = true;
        c->frame_gen = c->out_gen;
        c->frame_event = b->ev_ent;      // the end of the stage, not of whatever the caller enqueues behind it before it takes the bytes
    }
    if (b->ev_ent) HIPCHK(c0, hipEventRecord(b->ev_ent, s));
    if (side) HIPCHK(c0, hipStreamWaitEvent(b->stream, b->ev_ent, 0));
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

// stream idle -> did a bounded device-side wait expire since the last check?  (kernels_lf3.hip, LF_WAIT)
static int check_device_timeout(vp8hip_ctx *c) {
    int32_t flag = 0;
    HIPCHK(c, hipMemcpy(&flag, c->d_progress + LF_ERR_WORD, 4, hipMemcpyDeviceToHost));
    if (!flag) return VP8HIP_OK;
    HIPCHK(c, hipMemset(c->d_progress + LF_ERR_WORD, 0, 4));
    return VP8HIP_ERR_TIMEOUT;
}

int vp8hip_download_last(vp8hip_ctx *c, uint8_t *y, uint8_t *u, uint8_t *v) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c) return VP8HIP_ERR_ARG;
    if (c->slot[0] < 0) return VP8HIP_ERR_STATE;
    const Frame &f = c->frames[c->slot[0]].f;
    int rc;
    if (y && (rc = copy_out(c, y, f.Y[0]))) return rc;
    if (u && (rc = copy_out(c, u, f.U))) return rc;
    if (v && (rc = copy_out(c, v, f.V))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_device_timeout(c);
}

int vp8hip_synchronize(vp8hip_ctx *c) {
    USE_DEVICE(c);
    JOIN_LF(c);
    if (!c) return VP8HIP_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_device_timeout(c);
}

void *vp8hip_stream(vp8hip_ctx *c) { return c ? (void *)c->stream : nullptr; }
int vp8hip_last_hip_error(const vp8hip_ctx *c) { return c ? c->last_hip_error : 0; }

const char *vp8hip_status_string(int status) {
    switch (status) {
        case VP8HIP_OK: return "ok";
        case VP8HIP_ERR_ARG: return "bad argument";
        case VP8HIP_ERR_NO_DEVICE: return "no HIP device";
        case VP8HIP_ERR_HIP: return "HIP runtime error";
        case VP8HIP_ERR_STATE: return "call out of order";
        case VP8HIP_ERR_ARCH: return "device is not gfx950";
        case VP8HIP_ERR_TIMEOUT: return "a bounded device-side wait expired; the frame is invalid";
        case VP8HIP_ERR_OVERFLOW: return "coefficient partitions do not fit the output or the device scratch";
        case VP8HIP_ERR_FORMAT: return "first partition of 512 KB or more: the VP8 frame tag has 19 bits for its size";
        default: return "unknown";
    }
}

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

int vp8hip_abi_version(void) { return VP8HIP_ABI_VERSION; }
// ---- device memory for a caller that has none of its own (include/vp8hip.h) ------------------------------------------------
int vp8hip_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
#define DEVCHK(call) do { if ((call) != hipSuccess) return VP8HIP_ERR_HIP; } while (0)
int vp8hip_device_alloc(int device_ordinal, size_t bytes, void **out) {
    if (!out) return VP8HIP_ERR_ARG;
    *out = nullptr;
    if (device_ordinal < 0 || device_ordinal >= vp8hip_device_count()) return VP8HIP_ERR_NO_DEVICE;
    DEVCHK(hipSetDevice(device_ordinal));
    DEVCHK(hipMalloc(out, bytes ? bytes : 1));
    return VP8HIP_OK;
}
int vp8hip_device_free(int device_ordinal, void *p) {
    if (!p) return VP8HIP_OK;
    DEVCHK(hipSetDevice(device_ordinal));
    DEVCHK(hipFree(p));
    return VP8HIP_OK;
}
int vp8hip_device_upload(int device_ordinal, void *dst, const void *src, size_t bytes) {
    if (!dst || !src) return VP8HIP_ERR_ARG;
    DEVCHK(hipSetDevice(device_ordinal));
    DEVCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return VP8HIP_OK;
}
int vp8hip_device_download(int device_ordinal, void *dst, const void *src, size_t bytes) {
    if (!dst || !src) return VP8HIP_ERR_ARG;
    DEVCHK(hipSetDevice(device_ordinal));
    DEVCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return VP8HIP_OK;
}
int vp8hip_device_synchronize(int device_ordinal) {
    DEVCHK(hipSetDevice(device_ordinal));
    DEVCHK(hipDeviceSynchronize());
    return VP8HIP_OK;
}
int vp8hip_device_mem_info(int device_ordinal, size_t *free_bytes, size_t *total_bytes) {
    if (!free_bytes || !total_bytes) return VP8HIP_ERR_ARG;
    DEVCHK(hipSetDevice(device_ordinal));
    DEVCHK(hipMemGetInfo(free_bytes, total_bytes));
    return VP8HIP_OK;
}
int vp8hip_device_pci_bus_id(int device_ordinal, char *out, int len) {
    if (!out || len < 16) return VP8HIP_ERR_ARG;
    DEVCHK(hipDeviceGetPCIBusId(out, len, device_ordinal));
    return VP8HIP_OK;
}
int vp8hip_runtime_version(void) {
    int v = 0;
    return hipRuntimeGetVersion(&v) == hipSuccess ? v : 0;
}
#undef DEVCHK

int vp8hip_batch_prep_mode(void) {
    static const int prep_mode = [] { const char *v = getenv("VP8HIP_BATCH_PREP"); const int m = v && v[0] ? atoi(v) : 0; return m < 0 || m > 2 ? 0 : m; }();
    return prep_mode;
}
int vp8hip_experiments_compiled_in(void) {
#ifdef VP8HIP_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

int vp8hip_conformant_stream(vp8hip_ctx *c, int on) {
    if (!c) return VP8HIP_ERR_ARG;
    c->conformant = on ? 1 : 0;
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
