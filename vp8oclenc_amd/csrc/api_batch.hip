// api_batch.hip -- batched contexts: up to MAX_BATCH contexts of one geometry advance one frame together, every stage ONE launch (vp8hip_batch_*).
// The stages are those of api_inter.hip / api_entropy.hip; what a host with many GOP chunks in flight uses instead of a stream per chunk.
#include "vp8hip_ctx.h"

using namespace vp8;

namespace vp8 {

// A batch's second stream for the entropy stage, made when the first frame is asked for as bytes (a batch that only runs the
// inter path never has one: idle streams still take part in the runtime's stream -> hardware queue assignment).
bool batch_ent_stream(vp8hip_batch *b) {
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

}  // namespace vp8

extern "C" {

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
    if (b->copy) {
        hipStreamSynchronize(b->copy);
        hipStreamDestroy(b->copy);
        hipEventDestroy(b->ev_copied);
        hipEventDestroy(b->ev_packed[0]);
        hipEventDestroy(b->ev_packed[1]);
        for (int i = 0; i < b->n; ++i) { hipFree(b->stage[i][0]); hipFree(b->stage[i][1]); }
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

// one frame's planes into a staging buffer: ONE copy when they lie end to end in the host's memory (an I420 frame as a file reader or a
// decoder holds it), three otherwise
static int stage_copy(vp8hip_batch *b, uint8_t *d, const void *y, const void *u, const void *v, size_t ny, size_t nc) {
    vp8hip_ctx *c0 = b->c[0];
    const uint8_t *py = static_cast<const uint8_t *>(y);
    if (u == py + ny && v == py + ny + nc) {
        HIPCHK(c0, hipMemcpyAsync(d, y, ny + 2 * nc, hipMemcpyHostToDevice, b->copy));
        return VP8HIP_OK;
    }
    HIPCHK(c0, hipMemcpyAsync(d, y, ny, hipMemcpyHostToDevice, b->copy));
    HIPCHK(c0, hipMemcpyAsync(d + ny, u, nc, hipMemcpyHostToDevice, b->copy));
    HIPCHK(c0, hipMemcpyAsync(d + ny + nc, v, nc, hipMemcpyHostToDevice, b->copy));
    return VP8HIP_OK;
}

// the copy stream, its events and the members' staging buffers (two each), made on first use and again when the source size has changed
static int batch_stage_ready(vp8hip_batch *b) {
    vp8hip_ctx *c0 = b->c[0];
    const int sw = c0->src_w ? c0->src_w : c0->W, sh = c0->src_h ? c0->src_h : c0->H;
    const size_t bytes = (size_t)sw * sh + 2 * (size_t)(sw / 2) * (sh / 2);
    if (!b->copy) {
        HIPCHK(c0, hipStreamCreateWithFlags(&b->copy, hipStreamNonBlocking));
        HIPCHK(c0, hipEventCreateWithFlags(&b->ev_copied, hipEventDisableTiming));
        HIPCHK(c0, hipEventCreateWithFlags(&b->ev_packed[0], hipEventDisableTiming));
        HIPCHK(c0, hipEventCreateWithFlags(&b->ev_packed[1], hipEventDisableTiming));
    }
    if (b->stage_bytes == bytes) return VP8HIP_OK;
    // (first call, or the source size has changed: nothing in flight reads the old buffers after this)
    HIPCHK(c0, hipStreamSynchronize(b->copy));
    HIPCHK(c0, hipStreamSynchronize(b->stream));
    if (b->prep) HIPCHK(c0, hipStreamSynchronize(b->prep));
    for (int i = 0; i < b->n; ++i)
        for (int k = 0; k < 2; ++k) {
            (void)hipFree(b->stage[i][k]);
            b->stage[i][k] = nullptr;
            HIPCHK(c0, hipMalloc(&b->stage[i][k], bytes));
        }
    b->stage_bytes = bytes;
    b->packed_valid[0] = b->packed_valid[1] = false;
    b->pre_valid = false;
    return VP8HIP_OK;
}

// the members' new frames, tight planes in device memory (host == false) or in host memory (copied into the batch's staging buffers first)
static int batch_set_current(vp8hip_batch *b, const int *active, const void *const *y, const void *const *u, const void *const *v, bool host) {
    if (!b || !y || !u || !v) return VP8HIP_ERR_ARG;
    vp8hip_ctx *c0 = b->c[0];
    USE_DEVICE_ONLY(c0);
    const void *sy[MAX_BATCH], *su[MAX_BATCH], *sv[MAX_BATCH];
    int slot = -1;
    // every member's pointers are looked at BEFORE anything changes: an error return must leave the staging slots, the prefetch record and
    // the members' current frames as they were (a retry that flipped stage_idx once more would pack an older frame from the other slot)
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        if (!y[i] || !u[i] || !v[i]) return VP8HIP_ERR_ARG;
        if (b->c[i]->src_w != c0->src_w || b->c[i]->src_h != c0->src_h) return VP8HIP_ERR_ARG;   // one launch, one source size
    }
    if (host) {
        const int rc = batch_stage_ready(b);
        if (rc) return rc;
        const int sw = c0->src_w ? c0->src_w : c0->W, sh = c0->src_h ? c0->src_h : c0->H;
        const size_t ny = (size_t)sw * sh, nc = (size_t)(sw / 2) * (sh / 2);
        slot = b->stage_idx ^= 1;
        bool waited = false;
        for (int i = 0; i < b->n; ++i) {
            if (active && !active[i]) continue;
            uint8_t *d = b->stage[i][slot];
            if (!(b->pre_valid && b->pre[i][0] == y[i] && b->pre[i][1] == u[i] && b->pre[i][2] == v[i])) {   // not prefetched: copied now
                if (!waited && b->packed_valid[slot]) HIPCHK(c0, hipStreamWaitEvent(b->copy, b->ev_packed[slot], 0));   // the pack of two frames ago has read this buffer
                waited = true;
                {
                    const int cr = stage_copy(b, d, y[i], u[i], v[i], ny, nc);
                    if (cr) {       // a failed copy: nothing of this call counts -- not the flip, not the prefetch
                        b->stage_idx ^= 1;
                        b->pre_valid = false;
                        return cr;
                    }
                }
            }
            sy[i] = d; su[i] = d + ny; sv[i] = d + ny + nc;
        }
        b->pre_valid = false;
        HIPCHK(c0, hipEventRecord(b->ev_copied, b->copy));
        y = sy; u = su; v = sv;
    }
    const Frame *f[MAX_BATCH];
    const void *py[MAX_BATCH], *pu[MAX_BATCH], *pv[MAX_BATCH];
    int n = 0;
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
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
    if (host) HIPCHK(c0, hipStreamWaitEvent(ps, b->ev_copied, 0));
    {
        Timed t(c0, VP8HIP_K_PACK);
        launch_pack_batch(ps, f, py, pu, pv, n, c0->src_w, c0->src_h);
    }
    HIPCHK(c0, hipGetLastError());
    if (host) {
        HIPCHK(c0, hipEventRecord(b->ev_packed[slot], ps));
        b->packed_valid[slot] = true;
    }
    return VP8HIP_OK;
}

int vp8hip_batch_set_current_device(vp8hip_batch *b, const int *active, const void *const *y, const void *const *u, const void *const *v) {
    return batch_set_current(b, active, y, u, v, false);
}

// Planes in HOST memory (vp8enc.cpp:386-388, the reference's own hand-over: clEnqueueWriteBuffer from the frame it has read).  The
// copies are asynchronous when the planes are page-locked (vp8hip_host_alloc) and then overlap the batch's previous frame still on the
// device; page-locked or not, the planes must stay as they are until the NEXT vp8hip_batch_upload_current of this batch has returned (it waits for these
// copies first) or the batch's contexts have been synchronised.
// The frame AFTER the one under way, started on its way early: the copies go into the buffers the next vp8hip_batch_upload_current
// will pack from, and that call, given the same planes, finds them there.  A host that knows its next frame (a decoder's ring, a file
// reader one frame ahead) calls this right after it has enqueued the current one: the copies then have a whole frame's time.
int vp8hip_batch_prefetch_current(vp8hip_batch *b, const uint8_t *const *y, const uint8_t *const *u, const uint8_t *const *v) {
    if (!b || !y || !u || !v) return VP8HIP_ERR_ARG;
    vp8hip_ctx *c0 = b->c[0];
    USE_DEVICE_ONLY(c0);
    const int rc = batch_stage_ready(b);
    if (rc) return rc;
    const int sw = c0->src_w ? c0->src_w : c0->W, sh = c0->src_h ? c0->src_h : c0->H;
    const size_t ny = (size_t)sw * sh, nc = (size_t)(sw / 2) * (sh / 2);
    const int slot = b->stage_idx ^ 1;       // what the next upload will flip to
    if (b->packed_valid[slot]) HIPCHK(c0, hipStreamWaitEvent(b->copy, b->ev_packed[slot], 0));
    for (int i = 0; i < b->n; ++i) {
        b->pre[i][0] = b->pre[i][1] = b->pre[i][2] = nullptr;
        if (!y[i] || !u[i] || !v[i]) continue;
        uint8_t *d = b->stage[i][slot];
        { const int cr = stage_copy(b, d, y[i], u[i], v[i], ny, nc); if (cr) return cr; }
        b->pre[i][0] = y[i]; b->pre[i][1] = u[i]; b->pre[i][2] = v[i];
    }
    HIPCHK(c0, hipEventRecord(b->ev_copied, b->copy));
    b->pre_valid = true;
    return VP8HIP_OK;
}

int vp8hip_batch_upload_current(vp8hip_batch *b, const int *active, const uint8_t *const *y, const uint8_t *const *u, const uint8_t *const *v) {
    if (!b) return VP8HIP_ERR_ARG;
    if (b->copy) {
        (void)hipSetDevice(b->c[0]->device);
        HIPCHK(b->c[0], hipEventSynchronize(b->ev_copied));     // the previous call's copies have left the host's planes
    }
    return batch_set_current(b, active, reinterpret_cast<const void *const *>(y), reinterpret_cast<const void *const *>(u),
                             reinterpret_cast<const void *const *>(v), true);
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
        if (rc) {   // (inter_check above has passed: not expected.)  The scans collected so far have not been launched: they wait again
            for (int k = 0; k < n; ++k)
                if (scans[k]) m[k]->scan_deferred = true;
            return rc;
        }
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
    // Fewer launches for the five levels of the hierarchical search?  Measured (profiles/README.md, round 6): NO.  k_search1_coarse_b (a workgroup
    // takes a tile of 4 x 4 level-1 blocks and computes the tile's ancestors itself) is bit-exact and 9 % (levels 4-1 in one launch) to 15 %
    // (all five) SLOWER than a launch per level with 48 chunks in flight: the fused workgroup holds six waves through four barrier-separated
    // phases of which the first three occupy one or two, and its level 1 runs in the short-wave form.  VP8HIP_BATCH_S1_COARSE: 0 = a launch per
    // level (default), 1 = levels 4-1 in one + level 0, 2 = all five in one, 3 = levels 4-2 in one (two-wave workgroups) + levels 1 and 0.
    static const int coarse_mode = [] { const char *v = getenv("VP8HIP_BATCH_S1_COARSE"); return v && v[0] ? atoi(v) : 0; }();
    int first_level = 4, src = 0;
    if (coarse_mode > 0 && c0->mbw >= 2 && c0->mbh >= 2) {
        Timed t(c0, VP8HIP_K_SEARCH1_L2);
        if (launch_search1_coarse_batch(s, cur, refs, nets, net_width, n, coarse_mode == 2, coarse_mode == 3)) {
            first_level = coarse_mode == 2 ? -1 : (coarse_mode == 3 ? 1 : 0);
            src = first_level == 1 ? 1 : 0;      // level 1 reads net 1 (level 2's), level 0 reads net 0 (level 1's)
        }
    }
    for (int l = first_level; l >= 0; --l) {
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

// intra_transform (intra_part.h:1089-1109) for the members whose frame is a key frame, in ONE launch (the members' wavefronts side by side:
// a batch whose chunks all start a GOP used to run them one after the other), then every member's filter mask
// (prepare_filter_mask, loop_filter.h:25-55).  vp8hip_batch_loop_filter for the same members follows.
int vp8hip_batch_intra_transform(vp8hip_batch *b, const int *active) {
    if (!b) return VP8HIP_ERR_ARG;
    vp8hip_ctx *c0 = b->c[0];
    USE_DEVICE(c0);
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        if (b->c[i]->cur_count == 0) return VP8HIP_ERR_STATE;
    }
    CheckItem it[MAX_BATCH];
    vp8hip_ctx *m[MAX_BATCH];
    int n = 0;
    for (int i = 0; i < b->n; ++i) {
        if (active && !active[i]) continue;
        vp8hip_ctx *c = b->c[i];
        const int rc = claim_recon(c);
        if (rc) return rc;
        c->ent_counted_partitions = 0;
        drop_overflowed_frame(c);
        ++c->out_gen;
        CheckItem &k = it[n];
        k.cur = &c->cur;
        k.recon = &c->frames[c->recon].f;
        k.o = &c->out;
        k.sd = c->d_sd;
        k.modes = c->intra_modes;
        k.is_inter = c->intra_is_inter;
        k.prog = c->intra_prog;
        k.err = c->d_progress + LF_ERR_WORD;
        k.gen = ++c->intra_gen;
        m[n++] = c;
    }
    if (!n) return VP8HIP_OK;
    {
        Timed t(c0, VP8HIP_K_INTRA);
        launch_intra_key_batch(b->stream, it, n, c0->mbw, c0->mbh);
    }
    {
        Timed t(c0, VP8HIP_K_FILTER_MASK);
        for (int i = 0; i < n; ++i) launch_filter_mask(b->stream, m[i]->out, m[i]->d_sd, m[i]->mbs);
    }
    for (int i = 0; i < n; ++i) m[i]->recon_ready = true;
    HIPCHK(c0, hipGetLastError());
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

}  // extern "C"
