// api_context.hip -- the context: HBM surfaces, per-frame parameters, stream ordering, downloads, device memory for hosts that have none.
// Replaces init_all()'s GPU half (init.h:133-312, 430-582, 595-1166), the uploads of vp8enc.cpp:386-401 and the read-backs of inter_part.h:263-265.
#include <dirent.h>
#include <unistd.h>

#include "vp8hip_ctx.h"

using namespace vp8;

namespace vp8 {

thread_local LaunchTiming tl_timing;

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

// The timing events come from a pool of the process (per device) and go back to it: they are never destroyed.  Every context used
// to create 4 096 of them and destroy them with itself -- 200 000 per bench leg -- and a process that had TIMED kernels with them
// (hipExtLaunchKernel's start / stop events) ended inside the runtime in hipEventDestroy once in some twenty legs (segmentation
// fault or `double free or corruption`; scripts/stress_headline_flow.py: cycle 19 of 40 with events, none in 60 without).
static std::mutex g_event_mutex;
static std::vector<hipEvent_t> g_event_pool[64];
hipEvent_t event_pool_get(int device) {
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
void event_pool_put(int device, hipEvent_t *ev, int n) {
    std::lock_guard<std::mutex> lock(g_event_mutex);
    std::vector<hipEvent_t> &pool = g_event_pool[device & 63];
    for (int i = 0; i < n; ++i)
        if (ev[i]) pool.push_back(ev[i]);
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
int set_frame_planes(vp8hip_ctx *c, Frame &f, const void *y, const void *u, const void *v, hipMemcpyKind kind, int sw, int sh) {
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

void build_pyramid(vp8hip_ctx *c, Frame *a, Frame *b, uint32_t border_mask) {
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

std::atomic<int> g_live_contexts{0};   // contexts that launch on a stream of their own (members of a batch share one)

// Contexts overlap only if their streams sit on different hardware queues, and the HIP runtime multiplexes all streams of a process
// onto GPU_MAX_HW_QUEUES queues (default 4), read once when the runtime initialises at the process's first HIP call.  The reference
// creates the command queues it needs itself (init.h:1162-1165); a drop-in must not depend on its host's environment for that, so the
// library sets the variable when it is LOADED -- unless the host exported a value of its own.  Measured on MI355X with 32 contexts in 8
// batches: 4 queues 37 M MB/s, 8 -> 48, 12 -> 50, 16 -> 55, 20 -> 54, 24 -> 53; beyond 24 queues per process the hardware scheduler
// rotates them and context-switches running waves.
static int g_hw_queues = 4;          // what the runtime was (or will be) told
static bool g_runtime_was_up = false;

// Has this process initialised the GPU runtime already?  Its first act is to open the compute driver's device node.
static bool kfd_is_open() {
    DIR *dir = opendir("/proc/self/fd");      // every descriptor the process holds, however many (not a fixed range of numbers)
    if (!dir) return false;
    bool found = false;
    char link[300], target[128];
    while (const dirent *e = readdir(dir)) {
        if (e->d_name[0] < '0' || e->d_name[0] > '9') continue;
        snprintf(link, sizeof(link), "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(link, target, sizeof(target) - 1);
        if (n <= 0) continue;
        target[n] = 0;
        if (!strcmp(target, "/dev/kfd")) { found = true; break; }
    }
    closedir(dir);
    return found;
}

// (priority 101: ahead of this library's other load-time work -- the registration of its code objects with the runtime)
__attribute__((constructor(101))) static void vp8hip_loaded() {
    g_runtime_was_up = kfd_is_open();
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    // A SIDE EFFECT ON THE HOST PROCESS, documented in include/vp8hip.h (vp8hip_hw_queues): setenv from a library's constructor.  It is safe
    // where the library is loaded as libraries usually are -- at program start or from a host's main thread before it has made threads; a
    // host that dlopen()s the library while other threads may be calling getenv() exports GPU_MAX_HW_QUEUES itself, or sets VP8HIP_NO_ENV=1
    // (the library then touches nothing and reports the value in force).  The variable is inherited by the host's children, like any other.
    if (!q && !g_runtime_was_up && !getenv("VP8HIP_NO_ENV")) {
        setenv("GPU_MAX_HW_QUEUES", "16", 0);
        q = getenv("GPU_MAX_HW_QUEUES");
    }
    g_hw_queues = q && atoi(q) > 0 ? atoi(q) : 4;
    if (g_runtime_was_up && !q && !getenv("VP8HIP_QUIET"))
        fprintf(stderr, "vp8hip: the GPU runtime of this process was initialised before libvp8hip.so was loaded and GPU_MAX_HW_QUEUES was not set: "
                        "its streams share the default 4 hardware queues (37 instead of 55 M MB/s with 32 contexts in flight).  Load the library "
                        "before the first HIP call, or export GPU_MAX_HW_QUEUES=16.\n");
}

// A host that runs more contexts on streams of their own than there are queues gets a one-line note (VP8HIP_QUIET=1 silences it).
void note_queue_oversubscription() {
    static std::atomic<bool> warned{false};
    const int queues = g_hw_queues, n = g_live_contexts.load();
    if (n > queues && !getenv("VP8HIP_QUIET") && !warned.exchange(true))
        fprintf(stderr, "vp8hip: %d contexts on streams of their own but %d hardware queues: the streams will share queues and serialise.  "
                        "Advance the contexts in batches (vp8hip_batch_create)%s.\n", n, queues,
                queues < 16 ? ", and let the library set GPU_MAX_HW_QUEUES (16) by not exporting a smaller value" : "");
}

// New segment data while the previous frame's loop filter is still in flight on lf_stream (it reads its own frame's data):
// they go to the other buffer.  Consumers on the context's stream are ordered behind the write anyway.
SegData *sd_for_writing(vp8hip_ctx *c) {
    if (c->lf_pending && c->lf_sd == c->d_sd) c->d_sd = c->d_sd == c->d_sd2[0] ? c->d_sd2[1] : c->d_sd2[0];
    return c->d_sd;
}
// The parameter scan of a new frame (vp8hip_auto_segments) always takes the OTHER pair of blocks -- segment data and the
// strength words beside them -- and makes it the one in force: every consumer gets its pointers when it is enqueued, so the
// previous frame's loop filter and entropy stage keep reading theirs while this frame's scan already runs (on the second
// stream of vp8hip_filter_overlap, or on a batch's head-of-frame stream).
void next_params(vp8hip_ctx *c) {
    c->d_sd = c->d_sd == c->d_sd2[0] ? c->d_sd2[1] : c->d_sd2[0];
    c->d_stats = c->d_stats == c->d_stats2[0] ? c->d_stats2[1] : c->d_stats2[0];
}

// work enqueued on the context's stream from here on sees the filtered reconstruction
// work enqueued on the context's stream from here on may overwrite what the previous frame's entropy stage (on its own stream) reads
int join_ent(vp8hip_ctx *c) {
    if (!c->ent_pending) return VP8HIP_OK;
    c->ent_pending = false;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_ent, 0));
    return VP8HIP_OK;
}
// The join in two halves: the context goes back to the stream the filter is on (what is enqueued from then on runs behind the
// FILTER), and later that stream is told to wait for everything that ran beside the filter.  vp8hip_inter_transform puts the new
// LAST's pyramid and replicated edges -- which need the filter's output and nothing of the side work -- between the two: the
// barrier packet of the wait is then evaluated while that kernel runs instead of between the filter and it (11 us per frame).
hipStream_t join_lf_swap(vp8hip_ctx *c) {
    c->lf_pending = false;
    c->fork_by_verdict = false;
    // The streams trade places first: whatever the calls after this return, `stream` is the one vp8hip_create made again
    // (vp8hip_destroy relies on it)
    hipStream_t side = c->stream;
    c->stream = c->lf_stream;
    c->lf_stream = side;
    return side;
}
int join_lf_wait(vp8hip_ctx *c, hipStream_t side) {
    HIPCHK(c, hipEventRecord(c->ev_lf, side));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_lf, 0));
    return VP8HIP_OK;
}
// The LAST search needs of the side stream's work only the head: the current frame packed, scanned and downsampled.  The host
// is far ahead of the device there (it enqueues the chain while the filter still has 0.1 ms to run), so it can simply LOOK: once
// the event behind the head has been seen complete, launches enqueued from then on start after it whatever their stream, and the
// chain needs no barrier packet between the new LAST's pyramid and the LAST search (10 us per 1080p frame; the GOLDEN / ALTREF
// searches are waited for in front of k_mb, by the one barrier packet that is there anyway).  Gives up after 20 us.
bool side_sources_done(vp8hip_ctx *c) {
    for (int spins = 0; spins < 64; ++spins) {
        const hipError_t q = hipEventQuery(c->ev_src);
        if (q == hipSuccess) return true;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return false; }
        for (int i = 0; i < 16; ++i) __builtin_ia32_pause();
    }
    return false;
}
int join_lf(vp8hip_ctx *c, bool defer_ent) {
    if (!c->lf_pending) return defer_ent ? VP8HIP_OK : join_ent(c);
    // back to the stream the filter is on, behind it and behind everything that ran beside it
    hipStream_t side = join_lf_swap(c);
    { const int wr = join_lf_wait(c, side); if (wr) return wr; }
    // ... and behind the previous frame's entropy stage: what follows may overwrite the results it reads.  vp8hip_inter_transform
    // defers that wait to the one kernel of its chain that does (k_mb): the LAST search does not have to stand behind the stage.
    return defer_ent ? VP8HIP_OK : join_ent(c);
}
// work enqueued on the batch's stream from here on sees what its head-of-frame stream has been given so far
void batch_join_prep(vp8hip_batch *b) {
    if (!b) return;
    for (int i = 0; i < b->n; ++i) flush_scan(b->c[i]);   // (a member's parameter scan still waiting for its search launch: ahead of whatever comes now)
    b->ent_fork_fresh = false;   // (every entry point that may enqueue passes here: the entropy stage's early fork point is stale)
    if (!b->prep || !b->prep_pending) return;
    b->prep_pending = false;
    (void)hipEventRecord(b->ev_prep, b->prep);
    (void)hipStreamWaitEvent(b->stream, b->ev_prep, 0);
}
void side_stream_ordered(vp8hip_ctx *c) {
    if (!c->lf_pending || !c->fork_by_verdict) return;
    const uint32_t want = c->verdict_seq;
    for (unsigned spins = 0; (uint32_t)__atomic_load_n(&c->h_verdict[5], __ATOMIC_ACQUIRE) != want; ++spins) {
        if ((spins & 0xfff) == 0xfff && hipStreamQuery(c->lf_stream) != hipErrorNotReady) break;   // the filter's stream is idle: it has run (or failed; the next call says so)
        __builtin_ia32_pause();
    }
    c->fork_by_verdict = false;
}
void flush_scan(vp8hip_ctx *c) {
    if (!c->scan_deferred) return;
    c->scan_deferred = false;
    const ScanRequest &q = c->scan_req;
    launch_auto_segments(c->stream, c->cur, q.partial, q.stats, q.sd, q.strength_out, q.is_key, q.refqi, q.qi_min);
}

// a new current frame goes into the other of the two surfaces: the previous one stays intact for
// vp8hip_chroma_change (the reference keeps last_U/last_V the same way, encIO.h:207-210)
void next_current(vp8hip_ctx *c) {
    const Frame t = c->cur;
    c->cur = c->cur_prev;
    c->cur_prev = t;
    c->cur_count++;
    c->cur_pyramid_valid = false;
}

// stream idle -> did a bounded device-side wait expire since the last check?  (kernels_lf3.hip, LF_WAIT)
int check_device_timeout(vp8hip_ctx *c) {
    int32_t flag = 0;
    HIPCHK(c, hipMemcpy(&flag, c->d_progress + LF_ERR_WORD, 4, hipMemcpyDeviceToHost));
    if (!flag) return VP8HIP_OK;
    HIPCHK(c, hipMemset(c->d_progress + LF_ERR_WORD, 0, 4));
    return VP8HIP_ERR_TIMEOUT;
}

}  // namespace vp8

extern "C" {

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
    // every fixed device buffer of the context out of ONE allocation, zeroed by one memset (DeviceArena, vp8hip_ctx.h)
    DeviceArena &A = c->arena;
    A.want(&c->d_stats2[0], 2 * stats_words * sizeof(uint32_t));      // (holds a completion counter that is zero at rest)
    for (int r = 0; r < 3; ++r) {
        A.want(&c->nets.net[r][0], (size_t)c->b8 * 4);
        A.want(&c->nets.net[r][1], (size_t)c->b8 * 4);
        A.want(&c->nets.bdiff[r], (size_t)c->b8 * 4);
    }
    A.want(&c->out.parts, (size_t)c->mbs * 4);
    A.want(&c->out.ref, (size_t)c->mbs * 4);
    A.want(&c->out.seg, (size_t)c->mbs * 4);
    A.want(&c->out.nz, (size_t)c->mbs * 4);
    A.want(&c->out.mask, (size_t)c->mbs * 4);
    A.want(&c->out.ssim, (size_t)c->mbs * 4);
    A.want(&c->out.vec, (size_t)c->mbs * 16);
    A.want(&c->out.coeffs, (size_t)c->mbs * 800);
    A.want(&c->out.flags, 64);
    A.want(&c->d_sd2[0], 2 * sizeof(SegData));
    A.want(&c->d_progress, (size_t)c->mbh * 4 + 8192);   // band counters (+ diagnostic stamps at +4096, error word)
    A.want(&c->d_lf_handoff, loop_filter4_handoff_bytes(c->mbw, c->mbh));
    A.want(&c->scratch, (size_t)width * height);
    A.want(&c->ent_flags, (size_t)c->mbs * 25);
    A.want(&c->ent_third, (size_t)c->mbs * 25);
    A.want(&c->ent_counts, sizeof(uint32_t) * ENT_NCTX * 2 * c->mbh * 4);   // four partial histograms per macroblock row
    A.want(&c->ent_probs, sizeof(uint32_t) * ENT_NCTX);
    A.want(&c->ent_denom0, sizeof(uint32_t) * ENT_NCTX);
    A.want(&c->intra_modes, (size_t)c->mbs * 64);
    A.want(&c->intra_is_inter, (size_t)c->mbs * 4);
    A.want(&c->intra_prog, (size_t)c->mbh * 4);
    A.want(&c->intra_stats, 32);
    CR(A.commit());
    CR(hipMemsetAsync(A.base, 0, A.bytes, c->stream));
    c->d_stats2[1] = c->d_stats2[0] + stats_words;
    c->d_stats = c->d_stats2[0];
    c->d_sd2[1] = c->d_sd2[0] + 1;
    c->d_sd = c->d_sd2[0];
    CR(hipMemsetAsync(c->d_progress + S2_CLOCK_WORD, 0xff, 8, c->stream));   // k_search2's launch clock: "earliest start" is ~0 at rest
    // one page-locked block: the verdict words (fine-grained: the device's stores arrive while its kernel runs), then the segment-data ring
    CR(hipHostMalloc(&c->h_verdict, 256 + 16 * sizeof(SegData), hipHostMallocCoherent));
    memset(c->h_verdict, 0, 16 * sizeof(int32_t));
    c->h_sd_ring = reinterpret_cast<SegData *>(reinterpret_cast<uint8_t *>(c->h_verdict) + 256);
    c->recon = 0;
    CR(hipStreamSynchronize(c->stream));
#undef CR
    *out = c;
    c->counted = true;
    ++g_live_contexts;
    return VP8HIP_OK;
}

int vp8hip_hw_queues(void) { return g_hw_queues; }   // (set when the library was loaded: vp8hip_loaded above)

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
    if (c->h2d_stream) {      // (behind the context's stream: its last pack may have read a staging buffer)
        hipStreamSynchronize(c->h2d_stream);
        hipStreamDestroy(c->h2d_stream);
        hipEventDestroy(c->ev_h2d);
        hipEventDestroy(c->ev_stage_read[0]);
        hipEventDestroy(c->ev_stage_read[1]);
        hipFree(c->h2d_stage[0]);
        hipFree(c->h2d_stage[1]);
    }
    if (c->ev_chroma) hipEventDestroy(c->ev_chroma);
    shard_release(c);
    event_pool_put(c->device, c->ev, c->ev_made);
    hipFree(c->pixel_pool);
    c->arena.release();
    c->ent_arena.release();
    c->hdr_arena.release();
    if (c->h_frame) hipHostFree(c->h_frame);
    hipFree(c->d_frame);
    if (c->h_verdict) hipHostFree(c->h_verdict);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

// The next frame's planes started on their way while the current frame is coded (vp8hip_ctx.h): tight planes of the source size, one
// copy when they lie end to end (an I420 frame as a file reader holds it), on a stream of their own into the staging buffer the pack
// of two frames ago has finished with.  Touches nothing of the frame under way.
int vp8hip_prefetch_current(vp8hip_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v) {
    USE_DEVICE_ONLY(c);
    if (!c || !y || !u || !v) return VP8HIP_ERR_ARG;
    const int sw = c->src_w ? c->src_w : c->W, sh = c->src_h ? c->src_h : c->H;
    const size_t ny = (size_t)sw * sh, nc = (size_t)(sw / 2) * (sh / 2);
    if (!c->h2d_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_h2d, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_stage_read[0], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_stage_read[1], hipEventDisableTiming));
    }
    if (c->h2d_stage_bytes != ny + 2 * nc) {     // first use, or the source size has changed: whatever still reads the old buffers ends first
        HIPCHK(c, hipStreamSynchronize(c->h2d_stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->lf_stream) HIPCHK(c, hipStreamSynchronize(c->lf_stream));
        for (int k = 0; k < 2; ++k) {
            (void)hipFree(c->h2d_stage[k]);
            c->h2d_stage[k] = nullptr;
            HIPCHK(c, hipMalloc(&c->h2d_stage[k], ny + 2 * nc));
        }
        c->h2d_stage_bytes = ny + 2 * nc;
        c->stage_read_valid[0] = c->stage_read_valid[1] = false;
    }
    const int slot = c->h2d_idx ^ 1;
    if (c->stage_read_valid[slot]) HIPCHK(c, hipStreamWaitEvent(c->h2d_stream, c->ev_stage_read[slot], 0));
    uint8_t *d = c->h2d_stage[slot];
    if (u == y + ny && v == u + nc) {
        HIPCHK(c, hipMemcpyAsync(d, y, ny + 2 * nc, hipMemcpyHostToDevice, c->h2d_stream));
    } else {
        HIPCHK(c, hipMemcpyAsync(d, y, ny, hipMemcpyHostToDevice, c->h2d_stream));
        HIPCHK(c, hipMemcpyAsync(d + ny, u, nc, hipMemcpyHostToDevice, c->h2d_stream));
        HIPCHK(c, hipMemcpyAsync(d + ny + nc, v, nc, hipMemcpyHostToDevice, c->h2d_stream));
    }
    HIPCHK(c, hipEventRecord(c->ev_h2d, c->h2d_stream));
    c->h2d_pre[0] = y; c->h2d_pre[1] = u; c->h2d_pre[2] = v;
    c->h2d_pre_valid = true;
    return VP8HIP_OK;
}

int vp8hip_upload_current(vp8hip_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v) {
    USE_DEVICE(c);
    if (!c || !y || !u || !v) return VP8HIP_ERR_ARG;
    const int sw = c->src_w ? c->src_w : c->W, sh = c->src_h ? c->src_h : c->H;
    const size_t ny = (size_t)sw * sh, nc = (size_t)(sw / 2) * (sh / 2);
    // a prefetch counts only for the source size it was made for: the staging buffers hold ny + 2 nc bytes of THAT size and the pack would read
    // them with this one's offsets (vp8hip_set_source_size also drops a pending prefetch; this is the second lock on the same door)
    if (c->h2d_pre_valid && c->h2d_stage_bytes == ny + 2 * nc && c->h2d_pre[0] == y && c->h2d_pre[1] == u && c->h2d_pre[2] == v) {
        // prefetched: the planes are in (or on their way into) the staging buffer; the pack waits for the copy, nothing is copied here
        c->h2d_pre_valid = false;
        const int slot = c->h2d_idx ^= 1;
        next_current(c);
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_h2d, 0));
        const uint8_t *d = c->h2d_stage[slot];
        const int rc = set_frame_planes(c, c->cur, d, d + ny, d + ny + nc, hipMemcpyDeviceToDevice, c->src_w, c->src_h);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->ev_stage_read[slot], c->stream));
        c->stage_read_valid[slot] = true;
        HIPCHK(c, hipEventSynchronize(c->ev_h2d));      // the host's planes are the host's again when this returns (done long ago, normally)
        return VP8HIP_OK;
    }
    c->h2d_pre_valid = false;
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
    if (src_width == 0 && src_height == 0) {
        if (c->src_w || c->src_h) c->h2d_pre_valid = false;
        c->src_w = c->src_h = 0;
        return VP8HIP_OK;
    }
    if (src_width <= 0 || src_height <= 0 || (src_width & 1) || (src_height & 1) || src_width > c->W || src_height > c->H ||
        c->W - src_width >= 16 || c->H - src_height >= 16)
        return VP8HIP_ERR_ARG;
    const bool same = src_width == c->W && src_height == c->H;
    if ((same ? 0 : src_width) != c->src_w || (same ? 0 : src_height) != c->src_h) c->h2d_pre_valid = false;     // planes prefetched at the old size are not this size's frame
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

// The same scan without the host waiting for it: enqueued behind the current frame's pack, its two sums written to page-locked memory by
// the folding kernel itself; vp8hip_chroma_change_result waits for THEM only (an event behind the fold), not for the stream.  A host that
// hands the next frame over early (while the previous frame's loop filter still runs) has the answer by the time scene_change() asks.
int vp8hip_chroma_change_async(vp8hip_ctx *c) {
    USE_DEVICE(c);
    if (!c) return VP8HIP_ERR_ARG;
    if (c->cur_count == 0) return VP8HIP_ERR_STATE;
    c->chroma_pending = true;
    c->chroma_none = c->cur_count < 2;
    if (c->chroma_none) return VP8HIP_OK;
    if (!c->ev_chroma) HIPCHK(c, hipEventCreateWithFlags(&c->ev_chroma, hipEventDisableTiming));
    uint32_t *words = reinterpret_cast<uint32_t *>(c->h_verdict) + 32;      // (the verdict block's second half: coherent page-locked memory)
    launch_chroma_sad(c->stream, c->cur, c->cur_prev, c->d_stats + 8, words - 2);   // (the fold writes words 2 and 3 of what it is given)
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_chroma, c->stream));
    return VP8HIP_OK;
}

int vp8hip_chroma_change_result(vp8hip_ctx *c, int32_t *Udiff, int32_t *Vdiff) {
    if (!c || !Udiff || !Vdiff) return VP8HIP_ERR_ARG;
    if (!c->chroma_pending) return VP8HIP_ERR_STATE;
    c->chroma_pending = false;
    *Udiff = *Vdiff = 0;
    if (c->chroma_none) return VP8HIP_OK;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipEventSynchronize(c->ev_chroma));
    const volatile uint32_t *words = reinterpret_cast<const volatile uint32_t *>(c->h_verdict) + 32;
    const int nc = (c->W / 2) * (c->H / 2);
    *Udiff = (int32_t)words[0] / nc;      // vp8enc.cpp:277, 284
    *Vdiff = (int32_t)words[1] / nc;
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
// page-locked host memory: what makes vp8hip_upload_* / vp8hip_batch_upload_current copies asynchronous (and what finished frames are written into)
int vp8hip_host_alloc(int device_ordinal, size_t bytes, void **out) {
    if (!out) return VP8HIP_ERR_ARG;
    *out = nullptr;
    DEVCHK(hipSetDevice(device_ordinal));
    DEVCHK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return VP8HIP_OK;
}
int vp8hip_host_free(int device_ordinal, void *p) {
    if (!p) return VP8HIP_OK;
    DEVCHK(hipSetDevice(device_ordinal));
    DEVCHK(hipHostFree(p));
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

}  // extern "C"
