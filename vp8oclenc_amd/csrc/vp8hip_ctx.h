// vp8hip_ctx.h -- INTERNAL to libvp8hip.so: the context behind include/vp8hip.h, the batch, and what the api_*.hip units share.
//
// What the reference keeps as ~70 cl_mem objects and 60 pre-bound cl_kernel instances
// (init.h:430-593, 595-1271) is one context here: a pool of padded frame surfaces (a reference
// "slot" is an index into the pool, so golden := last is a pointer copy, not the five
// clEnqueueCopyBuffer + three clEnqueueCopyImage of inter_part.h:35-50,72-83), the vector nets,
// the per-macroblock outputs, and one in-order HIP stream.
//
// Units: api_context.hip (create / destroy, surfaces, parameters, stream ordering, downloads, device memory),
// api_inter.hip (inter path, check_SSIM, key frames, loop filter), api_entropy.hip (coefficient + header entropy stage, frames out),
// api_batch.hip (vp8hip_batch_*), api_shard.hip (export / import, RCCL: vp8hip_shard_*, vp8hip_group_*), api_profile.hip (timers, debug taps).
#ifndef VP8HIP_CTX_H
#define VP8HIP_CTX_H
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>
#include <new>

#include "../../include/vp8hip.h"
#include "vp8hip_dev.h"

struct ncclComm;     // (rccl.h is included by api_shard.hip only; the library resolves RCCL when a host first asks for it)

namespace vp8 {

constexpr int NFRAMES = 5;  // LAST, GOLDEN, ALTREF may all differ, + the reconstruction in flight, + 1 spare
constexpr int MAX_EVENTS = 4096;

struct FrameSurf {
    Frame f;
    bool pyramid_valid = false;
    bool border_valid = true;    // false: fresh out of the loop filter, replicated edges still to be made (with its pyramid)
};

}  // namespace vp8

struct vp8hip_batch;
// Device buffers that are made and freed together come out of ONE allocation: a context has forty of them, and forty hipMalloc + forty
// hipFree -- each a trip through the driver, the frees synchronising -- were 11 ms to make a context and 15 ms to destroy it (48 chunks:
// 1.2 s).  Every buffer starts 256-byte aligned and has a page of slack behind it.
struct DeviceArena {
    uint8_t *base = nullptr;
    size_t bytes = 0;
    struct Want { void **p; size_t bytes; };
    std::vector<Want> wants;
    template <class T> void want(T **p, size_t n) { wants.push_back(Want{reinterpret_cast<void **>(p), n}); }
    static size_t room(size_t n) { return ((n + 255) & ~(size_t)255) + 4096; }
    hipError_t commit() {      // one hipMalloc; the pointers handed to want() are set
        size_t total = 0;
        for (const Want &w : wants) total += room(w.bytes);
        bytes = total ? total : 256;
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&base), bytes);
        if (e != hipSuccess) { base = nullptr; wants.clear(); return e; }
        size_t at = 0;
        for (const Want &w : wants) { *w.p = base + at; at += room(w.bytes); }
        wants.clear();
        return hipSuccess;
    }
    void release() { if (base) (void)hipFree(base); base = nullptr; bytes = 0; }
};

struct vp8hip_ctx {
    int W = 0, H = 0, mbw = 0, mbh = 0, mbs = 0, b8 = 0;
    float ssim_target = -1.0f;
    int device = 0;
    hipStream_t stream = nullptr;       // the stream in use: the context's own, or its batch's (vp8hip_batch_create)
    hipStream_t own_stream = nullptr;   // the one vp8hip_create made
    int last_hip_error = 0;

    uint8_t *pixel_pool = nullptr;  // one allocation for every surface
    vp8::FrameSurf frames[vp8::NFRAMES];
    vp8::Frame cur;
    int slot[3] = {-1, -1, -1};     // pool index of LAST / GOLDEN / ALTREF
    int recon = -1;                 // pool index of the reconstruction being produced
    bool recon_ready = false;       // holds an unfiltered reconstruction
    bool cur_pyramid_valid = false;
    vp8::Frame cur_prev;                 // the previous current frame (the two surfaces swap on every upload)
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

    vp8::NetSet nets{};
    vp8::MBOut out{};
    vp8::SegData *d_sd = nullptr;        // the segment data in force (one of d_sd2)
    vp8::SegData *d_sd2[2] = {nullptr, nullptr};
    const vp8::SegData *lf_sd = nullptr; // what the loop filter in flight on lf_stream reads: the next frame's data go to the other buffer
    vp8::SegData *h_sd_ring = nullptr;   // pinned staging for vp8hip_set_segments
    unsigned sd_ring_pos = 0;
    int32_t *d_progress = nullptr;
    // a frame's reference searches on several devices (vp8hip_shard_*): this context's communicator
    ncclComm *shard_comm = nullptr;
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
    DeviceArena arena;                   // every fixed device buffer of the context but the pixel planes (vp8hip_create)
    DeviceArena ent_arena, hdr_arena;    // the boolean coder's scratch for the coefficient partitions / the first partition
    vp8::EntBuffers ent{};               // boolean coder scratch, allocated on first vp8hip_encode_coefficients
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
    vp8::ScanRequest scan_req{};
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
    // vp8hip_prefetch_current: the NEXT frame's planes (host memory) on their way into one of two staging buffers on a stream of their own,
    // while the current frame is coded; the vp8hip_upload_current that names the same planes packs from there and copies nothing
    hipStream_t h2d_stream = nullptr;
    hipEvent_t ev_h2d = nullptr, ev_stage_read[2] = {nullptr, nullptr};
    hipEvent_t ev_chroma = nullptr;    // behind the fold of vp8hip_chroma_change_async
    bool chroma_pending = false, chroma_none = false;
    bool stage_read_valid[2] = {false, false};
    uint8_t *h2d_stage[2] = {nullptr, nullptr};
    size_t h2d_stage_bytes = 0;
    int h2d_idx = 0;
    const void *h2d_pre[3] = {nullptr, nullptr, nullptr};
    bool h2d_pre_valid = false;
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
    vp8::EntBuffers hdr{};
    uint32_t *hdr_partial = nullptr, *hdr_info = nullptr;
    uint8_t *hdr_sym = nullptr;
    uint8_t *h_frame = nullptr;     // pinned staging of vp8hip_encode_frame's read-back (pageable targets serialise inside the runtime)
    uint8_t *d_frame = nullptr;     // the frame as gathered on the device: [0] size, [1] first-partition size, bytes from +16
    size_t h_frame_cap = 0;

    uint32_t prof_mask = 0;
    hipEvent_t ev[vp8::MAX_EVENTS];
    int ev_kernel[vp8::MAX_EVENTS / 2];
    int ev_used = 0;
    int ev_made = 0;                // ev[0 .. ev_made) are taken from the process's pool so far (event_pool_get)
    double prof_ms[VP8HIP_K_COUNT] = {0};
    int64_t prof_n[VP8HIP_K_COUNT] = {0};
};

// Up to MAX_BATCH contexts of one geometry on one device advance one frame together (see "batched contexts" below).
// `prep` is the batch's second stream: what a frame needs done before its searches but what does not depend on the previous
// frame's reconstruction -- taking the new frame in (pack / copy_with_padding), the parameter scan with the segment data, the
// new frame's pyramid -- runs there, beside the previous frame's chain on `stream`, instead of at the head of the chain behind the
// loop filter.  In the kernel trace of 48 chunks in 8 batches those three launches, a few microseconds of work each, lasted
// 0.2-0.45 ms per frame waiting for their turn: a fifth of the summed kernel time.
struct vp8hip_batch {
    int n = 0;
    vp8hip_ctx *c[vp8::MAX_BATCH] = {};
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
    // vp8hip_batch_upload_current: the members' new frames from HOST memory.  The copies run on a stream of their own (the copy engines,
    // beside whatever the batch's stream still has: the previous frame's loop filter) into one of two staging buffers per member; the
    // launch that packs them waits for the copies, the copies for the pack that last read their buffer.
    hipStream_t copy = nullptr;
    hipEvent_t ev_copied = nullptr, ev_packed[2] = {nullptr, nullptr};
    bool packed_valid[2] = {false, false};
    uint8_t *stage[vp8::MAX_BATCH][2] = {};
    size_t stage_bytes = 0;
    int stage_idx = 0;
    // vp8hip_batch_prefetch_current: the NEXT frame's planes already on their way into the buffer the next upload will pack from
    const void *pre[vp8::MAX_BATCH][3] = {};
    bool pre_valid = false;
};

#define HIPCHK(c, call)                                  \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) {                          \
            (c)->last_hip_error = (int)e_;               \
            return VP8HIP_ERR_HIP;                       \
        }                                                \
    } while (0)

namespace vp8 {

// The event behind which a host thread reads a finished frame out of pinned memory the stage's last kernel wrote: recorded with a
// SYSTEM-scope release.  An event made with hipEventDisableTiming alone releases to the device only, and the host then read, once in
// a few hundred frames, the previous frame's size word (the frame tag's first-partition size was the symptom).
static constexpr unsigned FRAME_EVENT_FLAGS = hipEventDisableTiming | hipEventReleaseToSystem;

// ---- per-kernel event timing --------------------------------------------------------------------
// Stages that are ONE kernel launch get their two events recorded by the dispatch itself (VP8_LAUNCH, vp8hip_dev.h): the
// kernel's own begin and end.  Stages made of several launches (entropy stage, intra) are bracketed with hipEventRecord.
constexpr uint32_t SINGLE_LAUNCH_STAGES = (1u << VP8HIP_K_PACK) | (1u << VP8HIP_K_DOWNSAMPLE) | (1u << VP8HIP_K_SEARCH1_L4) | (1u << VP8HIP_K_SEARCH1_L3) |
                                          (1u << VP8HIP_K_SEARCH1_L2) | (1u << VP8HIP_K_SEARCH1_L1) | (1u << VP8HIP_K_SEARCH1_L0) | (1u << VP8HIP_K_SEARCH2) |
                                          (1u << VP8HIP_K_MB) | (1u << VP8HIP_K_LOOP_FILTER) | (1u << VP8HIP_K_BORDER);
hipEvent_t event_pool_get(int device);
void event_pool_put(int device, hipEvent_t *ev, int n);

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

// ---- api_context.hip: surfaces, parameters, stream ordering ----
extern std::atomic<int> g_live_contexts;   // contexts that launch on a stream of their own (members of a batch share one)
void note_queue_oversubscription();
int pick_free_frame(const vp8hip_ctx *c);
int copy_in(vp8hip_ctx *c, const Plane &dst, const void *src, hipMemcpyKind kind);
int copy_out(vp8hip_ctx *c, void *dst, const Plane &src);
int set_frame_planes(vp8hip_ctx *c, Frame &f, const void *y, const void *u, const void *v, hipMemcpyKind kind, int sw = 0, int sh = 0);
void build_pyramid(vp8hip_ctx *c, Frame *a, Frame *b, uint32_t border_mask = 0);
int make_last(vp8hip_ctx *c, const void *y, const void *u, const void *v, hipMemcpyKind kind);
SegData *sd_for_writing(vp8hip_ctx *c);
void next_params(vp8hip_ctx *c);
void next_current(vp8hip_ctx *c);
int join_ent(vp8hip_ctx *c);
hipStream_t join_lf_swap(vp8hip_ctx *c);
int join_lf_wait(vp8hip_ctx *c, hipStream_t side);
bool side_sources_done(vp8hip_ctx *c);
int join_lf(vp8hip_ctx *c, bool defer_ent = false);
void flush_scan(vp8hip_ctx *c);
void batch_join_prep(vp8hip_batch *b);
void side_stream_ordered(vp8hip_ctx *c);
int check_device_timeout(vp8hip_ctx *c);
// ---- api_inter.hip ----
void drop_overflowed_frame(vp8hip_ctx *c);
int inter_check(const vp8hip_ctx *c, int prev_is_golden, int prev_is_altref, int use_golden, int use_altref);
int inter_begin(vp8hip_ctx *c, int prev_is_golden, int prev_is_altref, int use_golden, int use_altref);
RefSet ref_set(const vp8hip_ctx *c, int use_last, int use_golden, int use_altref);
unsigned long long *s2_clock_words(const vp8hip_ctx *c);
unsigned long long *s2_clock(const vp8hip_ctx *c);
int claim_recon(vp8hip_ctx *c);
void check_item(vp8hip_ctx *c, CheckItem &it, const int32_t refqi[4], int qi_min);
void lf_check(vp8hip_ctx *c, LfCheck &k);
bool fallback_possible(float ssim_target);
// ---- api_entropy.hip ----
constexpr size_t FRAME_FIRST_COPY = 192 * 1024;
bool frame_zero_copy();
int frame_prepare(vp8hip_ctx *c, int P, const vp8hip_header_params *p, FrameEntropy &e, FrameOut &fo);
// ---- api_shard.hip ----
void shard_release(vp8hip_ctx *c);            // the context's communicator, if it has one
int receive_last_surface(const vp8hip_ctx *c);
int adopt_last(vp8hip_ctx *c, int idx);
// ---- api_profile.hip ----
int prof_collect(vp8hip_ctx *c);

}  // namespace vp8

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
#define USE_DEVICE(c) do { if (c) { (void)hipSetDevice((c)->device); batch_join_prep((c)->batch); side_stream_ordered(c); flush_scan(c); } } while (0)
#define USE_DEVICE_ONLY(c) do { if (c) (void)hipSetDevice((c)->device); } while (0)
#define JOIN_LF(c) do { if (c) { const int jr_ = join_lf(c); if (jr_) return jr_; } } while (0)

#endif
