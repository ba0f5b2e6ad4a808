// vp8_driver.cpp -- the reference's frame loop (main(), src/vp8enc.cpp:351-488), native host code over the C ABI
// (include/vp8hip_driver.h).  No GPU code here: it only sequences vp8hip_* calls with the parameters the host
// mirror (vp8_host.cpp) or the device (vp8hip_auto_segments) produces.
#include <stdlib.h>

#include <new>
#include <chrono>
#include <string.h>
#include <thread>
#include <vector>

#include "../../include/vp8hip_bitstream.h"
#include "../../include/vp8hip_driver.h"
#include "../../include/vp8hip_host.h"

struct vp8drv {
    vp8hip_ctx *hip = nullptr;
    vp8drv_config cfg{};
    int W = 0, H = 0, mbs = 0;
    vp8host_gop gop{};
    vp8host_scene_state scene{};     // scene_change()'s hold-over and frames.last_key_detect
    int32_t lastqi[4]{}, altrefqi[4]{};
    int qi_min = 0;
    vp8drv_stats st{};
    // what vp8drv_get_frame needs to know about the frame just coded
    bool have_frame = false, last_key = false, last_altref = false, checked = false;
    int sharpness = VP8HIP_SHARPNESS_ON_DEVICE;   // video.loop_filter_sharpness in force, or: still on the device (vp8hip_get_segments).  May be NEGATIVE (vp8hip.h)
    int replaced = 0;
    bool verdict_pending = false;    // check_SSIM's verdict on the frame just coded is still on its way (resolve())
    const uint8_t *staged = nullptr; // vp8drv_stage_frame_host: these planes are the context's current frame already (and, with scene_detect, their scan is under way)
    bool staged_scan = false;
    // read-back buffers of vp8drv_get_frame
    std::vector<int32_t> seg, nz, ref, parts, is_inter, modes;
    std::vector<int16_t> vectors;
    std::vector<uint8_t> partitions;
};

extern "C" {

void vp8drv_default_config(vp8drv_config *c) {
    if (!c) return;
    c->gop_size = 150;
    c->altref_range = 5;
    c->qi_min = 0;
    c->qi_max = 48;
    c->ssim_target = -1.0f;
    c->device_params = 1;
    c->check_ssim = 1;
    c->num_partitions = 1;
    c->display_width = 0;
    c->display_height = 0;
    c->host_bitstream = 0;
    c->overlap_filter = 0;
    c->ref_mask = 3;
    c->conformant_stream = 0;
    c->scene_detect = 0;
    c->src_width = c->src_height = 0;
}

int vp8drv_create(vp8drv **out, int width, int height, int device_ordinal, const vp8drv_config *cfg) {
    if (!out || !cfg) return VP8HIP_ERR_ARG;
    *out = nullptr;
    vp8drv *d = new (std::nothrow) vp8drv();
    if (!d) return VP8HIP_ERR_ARG;
    d->cfg = *cfg;
    const int rc = vp8hip_create(&d->hip, width, height, cfg->ssim_target, device_ordinal);
    if (rc != VP8HIP_OK) {
        delete d;
        return rc;
    }
    if (cfg->overlap_filter) vp8hip_filter_overlap(d->hip, 1);
    if (cfg->conformant_stream) vp8hip_conformant_stream(d->hip, 1);
    if (cfg->src_width || cfg->src_height) {
        const int rc = cfg->device_params ? vp8hip_set_source_size(d->hip, cfg->src_width, cfg->src_height) : VP8HIP_ERR_ARG;
        if (rc != VP8HIP_OK) {
            vp8hip_destroy(d->hip);
            delete d;
            return rc;
        }
        if (!d->cfg.display_width) d->cfg.display_width = cfg->src_width;       // video.dst_width/height = the source's
        if (!d->cfg.display_height) d->cfg.display_height = cfg->src_height;
    }
    d->W = width;
    d->H = height;
    d->mbs = (width / 16) * (height / 16);
    vp8host_gop_init(&d->gop, cfg->gop_size, cfg->altref_range);
    vp8host_quantizer_ladders(cfg->qi_min, cfg->qi_max, d->lastqi, d->altrefqi);
    d->qi_min = cfg->qi_min < cfg->qi_max ? cfg->qi_min : cfg->qi_max;
    *out = d;
    return VP8HIP_OK;
}

void vp8drv_destroy(vp8drv *d) {
    if (!d) return;
    vp8hip_destroy(d->hip);
    delete d;
}

vp8hip_ctx *vp8drv_context(vp8drv *d) { return d ? d->hip : nullptr; }

void vp8drv_get_stats(const vp8drv *d, vp8drv_stats *s) {
    if (d && s) *s = d->st;
}

}  // extern "C"

namespace {

#define DRV_CHK(call)                     \
    do {                                  \
        const int rc_ = (call);           \
        if (rc_ != VP8HIP_OK) return rc_; \
    } while (0)

// segment data of the current frame: on the device, or from the host mirror on the caller's luma plane
int segments(vp8drv *d, const uint8_t *host_y, bool key, const int32_t *refqi) {
    d->sharpness = VP8HIP_SHARPNESS_ON_DEVICE;
    if (d->cfg.device_params || !host_y) return vp8hip_auto_segments(d->hip, key ? 1 : 0, refqi, d->qi_min);
    int32_t red = 0, sharp = 0, sd[VP8HIP_SD_INTS];
    vp8host_loopfilter_strength(host_y, d->W, d->H, &red, &sharp);
    vp8host_prepare_segments_data(key ? 1 : 0, refqi, d->qi_min, red, sharp, 0, 0, sd);
    d->sharpness = sharp;
    return vp8hip_set_segments(d->hip, sd);
}

// prepare_segments_data() + intra_transform() (vp8enc.cpp:379-383, :411-414, :446-450) and the common tail
// (:472-473): filter mask, loop filter.  The filtered key frame is LAST = GOLDEN = ALTREF (intra_part.h:1091-1098).
int key_frame(vp8drv *d, const uint8_t *host_y) {
    DRV_CHK(segments(d, host_y, true, d->altrefqi));
    DRV_CHK(vp8hip_intra_transform(d->hip));
    d->scene.last_key_detect = d->gop.frame_number;       // intra_part.h:1093, for every key frame whoever asked for it
    vp8host_gop_key_coded(&d->gop);
    DRV_CHK(vp8hip_prepare_filter_mask(d->hip, nullptr));
    DRV_CHK(vp8hip_loop_filter(d->hip));
    vp8host_gop_frame_done(&d->gop);
    d->st.key_frames++;
    d->st.frame_number = d->gop.frame_number;
    d->have_frame = true;
    d->last_key = d->last_altref = true;
    d->checked = false;
    d->replaced = 0;
    return 1;
}

// the inter frame just coded is final: the counters of main()'s loop tail
void inter_frame_done(vp8drv *d) {
    vp8host_gop_frame_done(&d->gop);
    d->st.inter_frames++;
    d->st.frame_number = d->gop.frame_number;
}

// The verdict of the asynchronous check_SSIM on the frame just coded (vp8enc.cpp:442-453): statistics, and -- when more than a
// sixth of the macroblocks were replaced or the frame's SSIM is below the target -- the frame again as a key frame, on the
// current frame the context still holds.  (The inter version's loop filter has run by then: its output is simply never used.)
// Returns 1 if the frame ended as a key frame, 0 if not, < 0 = vp8hip_status.
int resolve(vp8drv *d) {
    if (!d->verdict_pending) return 0;
    d->verdict_pending = false;
    int32_t replaced = 0, updated = 0;
    float new_ssim = 0.0f, min1 = 2.0f;
    DRV_CHK(vp8hip_check_ssim_result(d->hip, &replaced, &new_ssim, &min1, &updated));
    d->st.last_replaced = replaced;
    d->st.last_new_ssim = new_ssim;
    d->st.last_min_ssim = min1;
    // e_data.is_inter_mb / mode travel to the header coder only when something was replaced: with nothing below the target the
    // fallback's launch left at once and did not even initialise them (all macroblocks inter: the same bits without them)
    d->checked = replaced > 0;
    d->replaced = replaced;
    if (updated) d->sharpness = 7;      // prepare_segments_data(1, 7), :260-261, happened on the device
    if (replaced > d->mbs / 6 || new_ssim < d->cfg.ssim_target) {
        d->st.redone_as_key++;
        return key_frame(d, nullptr);
    }
    inter_frame_done(d);
    return 0;
}

// the loop body once the current frame is on the device; host_y: the caller's luma plane or nullptr
int frame_body(vp8drv *d, const uint8_t *host_y, bool key) {
    const bool scan_under_way = d->staged_scan;
    d->staged_scan = false;
    if (!key && d->cfg.scene_detect) {     // vp8enc.cpp:408-416: only frames that would be inter frames are looked at
        int32_t Udiff = 0, Vdiff = 0;
        if (scan_under_way) DRV_CHK(vp8hip_chroma_change_result(d->hip, &Udiff, &Vdiff));     // (started when the frame was handed over early)
        else DRV_CHK(vp8hip_chroma_change(d->hip, &Udiff, &Vdiff));
        if (vp8host_scene_change(&d->scene, Udiff, Vdiff, d->gop.frame_number)) {
            d->st.scene_changes++;
            key = true;
        }
    }
    if (key) return key_frame(d, host_y);
    // vp8enc.cpp:390, 419: loop-filter strength of the current frame -> segment data
    const int32_t *refqi = d->gop.current_is_altref ? d->altrefqi : d->lastqi;   // vp8enc.cpp:149-151
    DRV_CHK(segments(d, host_y, false, refqi));
    int32_t use_golden = 0, use_altref = 0;
    vp8host_gop_inter_flags(&d->gop, &use_golden, &use_altref);                  // inter_part.h:103-104
    use_golden &= d->cfg.ref_mask & 1;
    use_altref &= (d->cfg.ref_mask >> 1) & 1;
    DRV_CHK(vp8hip_inter_transform(d->hip, d->gop.prev_is_golden, d->gop.prev_is_altref, use_golden, use_altref));
    d->st.last_use_golden = use_golden;
    d->st.last_use_altref = use_altref;
    d->st.refs_searched += 1 + use_golden + use_altref;
    d->st.last_prev_is_golden = d->gop.prev_is_golden;
    d->st.last_prev_is_altref = d->gop.prev_is_altref;
    d->st.last_was_altref = d->gop.current_is_altref;
    d->checked = false;
    d->replaced = 0;
    if (d->cfg.check_ssim && (d->cfg.device_params || !host_y)) {
        // check_SSIM (vp8enc.cpp:231-263) with nobody waiting for it: fallback, statistics and the filter update on the device,
        // the loop filter right behind them.  What the host has to decide -- "redo as key frame", :443-453 -- it decides when it
        // next needs to know: at the start of the next call or in vp8drv_get_frame (resolve()).
        DRV_CHK(vp8hip_check_ssim_async(d->hip, refqi, d->qi_min));
        DRV_CHK(vp8hip_loop_filter(d->hip));
        d->verdict_pending = true;
        d->have_frame = true;
        d->last_key = false;
        d->last_altref = d->st.last_was_altref != 0;
        return 0;
    }
    if (d->cfg.check_ssim) {
        // the same with the host in the middle (host parameter mirror: the reference's own sequence of calls)
        int32_t replaced = 0;
        float new_ssim = 0.0f, min1 = 2.0f;
        DRV_CHK(vp8hip_check_ssim(d->hip, &replaced, &new_ssim, &min1));
        d->st.last_replaced = replaced;
        d->st.last_new_ssim = new_ssim;
        d->st.last_min_ssim = min1;
        d->checked = true;
        d->replaced = replaced;
        if (min1 > 0.95f) {
            // even the worst macroblock is good: weaken the loop filter (reductor * 2, sharpness 7; :155-159, :260-261)
            int32_t red = 0, sharp = 0, sd[VP8HIP_SD_INTS];
            if (d->cfg.device_params || !host_y) DRV_CHK(vp8hip_get_segments(d->hip, sd, &red, &sharp));
            else vp8host_loopfilter_strength(host_y, d->W, d->H, &red, &sharp);
            vp8host_prepare_segments_data(0, refqi, d->qi_min, red, sharp, 1, 7, sd);
            DRV_CHK(vp8hip_set_segments(d->hip, sd));
            d->sharpness = 7;
        }
        if (replaced > d->mbs / 6 || new_ssim < d->cfg.ssim_target) {   // vp8enc.cpp:443-453: redo as a key frame
            d->st.redone_as_key++;
            return key_frame(d, host_y);
        }
        // the fallback changed coefficients: mask and non-zero counts again (loop_filter.h:25-55)
        if (replaced > 0) DRV_CHK(vp8hip_prepare_filter_mask(d->hip, nullptr));
    }
    // otherwise prepare_filter_mask was produced by vp8hip_inter_transform for its own coefficients;
    // do_loop_filter (loop_filter.h:185-190) follows directly
    DRV_CHK(vp8hip_loop_filter(d->hip));
    inter_frame_done(d);
    d->have_frame = true;
    d->last_key = false;
    d->last_altref = d->st.last_was_altref != 0;
    return 0;
}

// entropy_encode() + gather_frame() (vp8enc.cpp:48-94, 476-481; encIO.h:1-30) for the frame just coded:
// coefficient statistics and partitions on the device, first partition on the host
vp8hip_header_params header_params(const vp8drv *d) {
    vp8hip_header_params hp{};
    hp.is_key = d->last_key;
    hp.is_golden = d->last_key;                 // current_is_golden_frame = current_is_key_frame (vp8enc.cpp:369)
    hp.is_altref = d->last_altref;
    hp.loop_filter_type = 0;                    // init.h:1583
    hp.loop_filter_sharpness = d->sharpness;    // or VP8HIP_SHARPNESS_ON_DEVICE
    hp.width = d->cfg.display_width;
    hp.height = d->cfg.display_height;
    hp.use_intra_info = d->checked;
    return hp;
}

int get_frame(vp8drv *d, uint8_t *out, size_t capacity, size_t *size) {
    const int P = d->cfg.num_partitions;
    const size_t n = (size_t)d->mbs;
    if (!d->cfg.host_bitstream) {   // the whole entropy stage on the device, one read-back
        const vp8hip_header_params hp = header_params(d);
        return vp8hip_encode_frame(d->hip, P, &hp, out, capacity, size);
    }
    uint32_t probs[VP8BS_NUM_COEFF_PROBS], denom[VP8BS_NUM_COEFF_PROBS];
    d->nz.resize(n);
    DRV_CHK(vp8hip_prepare_filter_mask(d->hip, d->nz.data()));                   // counts of the final coefficients (vp8enc.cpp:472)
    DRV_CHK(vp8hip_count_probs(d->hip, P, probs, denom));                        // :58-68
    vp8bs_default_probs(probs, denom);                                           // :69-76
    // four times video.partition_step (init.h:409,1190): a partition that outgrows the reference's buffer overruns it
    // there; found by the randomised parity run with noise at a low quantizer on a frame of two macroblock rows
    const size_t step = n * 3200 / (size_t)P + 4096;
    d->partitions.resize(step * (size_t)P);
    int32_t sizes[8] = {0};
    DRV_CHK(vp8hip_encode_coefficients(d->hip, probs, P, (int)step, d->partitions.data(), sizes));   // :77-81
    // what encode_header reads from the frame's results (entropy_host.cpp:221-223, 808, 986-987, 1086)
    d->seg.resize(n);
    d->ref.resize(n);
    d->parts.resize(n);
    d->vectors.resize(n * 8);
    vp8hip_results r{};
    r.MB_segment_id = d->seg.data();
    if (!d->last_key) {
        r.MB_reference_frame = d->ref.data();
        r.MB_parts = d->parts.data();
        r.MB_vectors = d->vectors.data();
    }
    DRV_CHK(vp8hip_download_results(d->hip, &r));
    const bool intra_info = d->last_key || d->checked;
    if (intra_info) {
        d->modes.resize(n * 16);
        d->is_inter.resize(n);
        DRV_CHK(vp8hip_download_intra(d->hip, d->modes.data(), d->last_key ? nullptr : d->is_inter.data()));
    }
    int32_t sd[VP8HIP_SD_INTS], red = 0, sharp = 0;
    DRV_CHK(vp8hip_get_segments(d->hip, sd, &red, &sharp));
    if (d->sharpness != VP8HIP_SHARPNESS_ON_DEVICE) sharp = d->sharpness;
    vp8bs_frame f{};
    f.width = d->cfg.display_width > 0 ? d->cfg.display_width : d->W;
    f.height = d->cfg.display_height > 0 ? d->cfg.display_height : d->H;
    f.mb_width = d->W / 16;
    f.mb_height = d->H / 16;
    f.is_key = d->last_key;
    f.is_golden = d->last_key;                  // current_is_golden_frame = current_is_key_frame (vp8enc.cpp:369)
    f.is_altref = d->last_altref;
    f.loop_filter_type = 0;                     // init.h:1583
    f.loop_filter_sharpness = sharp;
    f.partitions_log2 = P == 8 ? 3 : (P == 4 ? 2 : (P == 2 ? 1 : 0));
    f.skip_prob = vp8host_skip_prob(d->nz.data(), d->mbs);
    f.replaced = d->last_key ? 0 : d->replaced;
    f.segments = sd;
    f.MB_segment_id = d->seg.data();
    f.MB_non_zero_coeffs = d->nz.data();
    f.MB_reference_frame = d->last_key ? nullptr : d->ref.data();
    f.MB_parts = d->last_key ? nullptr : d->parts.data();
    f.MB_vectors = d->last_key ? nullptr : d->vectors.data();
    f.is_inter_mb = (!d->last_key && d->checked) ? d->is_inter.data() : nullptr;
    f.modes = intra_info ? d->modes.data() : nullptr;
    f.new_probs = probs;
    f.new_probs_denom = denom;
    const size_t head = vp8bs_encode_header(&f, out, capacity, nullptr);          // :84
    if (head == (size_t)-1) return VP8HIP_ERR_FORMAT;   // 19-bit size field of the frame tag
    if (!head) return VP8HIP_ERR_OVERFLOW;
    const size_t total = vp8bs_gather_frame(out, head, capacity, P, d->partitions.data(), step, sizes);
    if (!total) return VP8HIP_ERR_OVERFLOW;
    *size = total;
    return VP8HIP_OK;
}

}  // namespace

extern "C" {

int vp8drv_encode_frame_device(vp8drv *d, const void *y, const void *u, const void *v, int force_key) {
    if (!d || !y || !u || !v) return VP8HIP_ERR_ARG;
    { const int rc = resolve(d); if (rc < 0) return rc; }                        // the previous frame's check_SSIM verdict, if still open
    vp8host_gop_next(&d->gop);
    DRV_CHK(vp8hip_set_current_device(d->hip, y, u, v));                          // vp8enc.cpp:386-388
    return frame_body(d, nullptr, d->gop.current_is_key || force_key);
}

int vp8drv_encode_frame_host(vp8drv *d, const uint8_t *y, const uint8_t *u, const uint8_t *v, int force_key) {
    if (!d || !y || !u || !v) return VP8HIP_ERR_ARG;
    { const int rc = resolve(d); if (rc < 0) return rc; }
    vp8host_gop_next(&d->gop);
    const bool staged = d->staged == y;
    d->staged = nullptr;
    if (!staged) {
        d->staged_scan = false;
        DRV_CHK(vp8hip_upload_current(d->hip, y, u, v));
    }
    return frame_body(d, y, d->gop.current_is_key || force_key);
}

// The NEXT frame handed to the device early -- while the frame just coded is in its loop filter -- so that the next
// vp8drv_encode_frame_host, given the same planes, finds the frame there and (with cfg.scene_detect) scene_change()'s two sums on their
// way or back: its side work (parameter scan, pyramid, GOLDEN / ALTREF searches) is then enqueued early enough to run beside the filter.
// Takes the open verdict first (the frame it belongs to must not lose its place as the context's current frame before a "redo as key
// frame").  Call it after vp8drv_get_frame_begin / vp8drv_resolve of the frame just coded; the planes stay unchanged until the
// vp8drv_encode_frame_host that names them has returned.
int vp8drv_stage_frame_host(vp8drv *d, const uint8_t *y, const uint8_t *u, const uint8_t *v) {
    if (!d || !y || !u || !v) return VP8HIP_ERR_ARG;
    { const int rc = resolve(d); if (rc < 0) return rc; }
    DRV_CHK(vp8hip_upload_current(d->hip, y, u, v));
    d->staged = y;
    d->staged_scan = false;
    if (d->cfg.scene_detect) {
        DRV_CHK(vp8hip_chroma_change_async(d->hip));
        d->staged_scan = true;
    }
    return VP8HIP_OK;
}

int vp8drv_prefetch_frame_host(vp8drv *d, const uint8_t *y, const uint8_t *u, const uint8_t *v) {
    if (!d || !y || !u || !v) return VP8HIP_ERR_ARG;
    return vp8hip_prefetch_current(d->hip, y, u, v);
}

// ---- several GOP chunks one frame at a time, every stage one launch for all of them (vp8hip_batch_*) -----------------------
struct vp8drv_batch {
    int n = 0;
    vp8drv *d[VP8HIP_MAX_BATCH] = {};
    vp8hip_batch *hb = nullptr;
};

int vp8drv_batch_create(vp8drv_batch **out, vp8drv *const *drv, int n) {
    if (!out || !drv || n < 1 || n > VP8HIP_MAX_BATCH) return VP8HIP_ERR_ARG;
    *out = nullptr;
    vp8hip_ctx *ctx[VP8HIP_MAX_BATCH];
    for (int i = 0; i < n; ++i) {
        // the batched loop is the device-parameter loop (what bench.py and a file-to-file transcode run); one launch serves all
        // members, so what travels as ONE kernel argument must agree: quantizer range, check_SSIM on or off, partitions
        if (!drv[i] || !drv[i]->cfg.device_params || drv[i]->cfg.overlap_filter || drv[i]->cfg.scene_detect) return VP8HIP_ERR_ARG;
        const vp8drv_config &a = drv[i]->cfg, &z = drv[0]->cfg;
        if (a.qi_min != z.qi_min || a.qi_max != z.qi_max || a.num_partitions != z.num_partitions || (a.check_ssim != 0) != (z.check_ssim != 0))
            return VP8HIP_ERR_ARG;
        ctx[i] = drv[i]->hip;
    }
    vp8drv_batch *b = new (std::nothrow) vp8drv_batch();
    if (!b) return VP8HIP_ERR_ARG;
    const int rc = vp8hip_batch_create(&b->hb, ctx, n);
    if (rc != VP8HIP_OK) {
        delete b;
        return rc;
    }
    b->n = n;
    for (int i = 0; i < n; ++i) b->d[i] = drv[i];
    *out = b;
    return VP8HIP_OK;
}

void vp8drv_batch_destroy(vp8drv_batch *b) {
    if (!b) return;
    vp8hip_batch_destroy(b->hb);
    delete b;
}

static int batch_encode_frame(vp8drv_batch *b, const int *members, const void *const *y, const void *const *u, const void *const *v,
                              const int *force_key, int *was_key, bool host) {
    if (!b || !y || !u || !v) return VP8HIP_ERR_ARG;
    int key[VP8HIP_MAX_BATCH], active[VP8HIP_MAX_BATCH], zero[VP8HIP_MAX_BATCH] = {};
    int pg[VP8HIP_MAX_BATCH], pa[VP8HIP_MAX_BATCH], ug[VP8HIP_MAX_BATCH], ua[VP8HIP_MAX_BATCH];
    int32_t refqi[VP8HIP_MAX_BATCH][4];
    for (int i = 0; i < b->n; ++i) {   // the members' open check_SSIM verdicts: a frame sent back is recoded as a key frame now
        const int rc = resolve(b->d[i]);
        if (rc < 0) return rc;
    }
    for (int i = 0; i < b->n; ++i) {
        vp8drv *d = b->d[i];
        key[i] = active[i] = 0;
        if (was_key) was_key[i] = 0;
        if (members && !members[i]) continue;      // this member sits the call out
        vp8host_gop_next(&d->gop);
        key[i] = d->gop.current_is_key || (force_key && force_key[i]);
        active[i] = !key[i];
        if (was_key) was_key[i] = key[i];
    }
    if (host) DRV_CHK(vp8hip_batch_upload_current(b->hb, members, reinterpret_cast<const uint8_t *const *>(y), reinterpret_cast<const uint8_t *const *>(u),
                                                  reinterpret_cast<const uint8_t *const *>(v)));
    else DRV_CHK(vp8hip_batch_set_current_device(b->hb, members, y, u, v));          // vp8enc.cpp:386-388, all members in one launch
    int n_inter = 0, n_key = 0;
    for (int i = 0; i < b->n; ++i) n_key += key[i];
    static const bool batch_keys = [] { const char *e = getenv("VP8DRV_BATCH_KEYS"); return !(e && e[0] == '0'); }();   // =0: one after the other, as before (A/B runs)
    if (!batch_keys) n_key = n_key ? 1 : 0;
    if (n_key > 1) {
        // Several members start a GOP in this call (every chunk of a file coded GOPs side by side does, together): their key frames --
        // each a raster-order wavefront a millisecond and a half long -- in ONE launch instead of one after the other on the shared
        // stream, with the segment data, filter masks and loop filters of key_frame() batched the same way
        int32_t kq[VP8HIP_MAX_BATCH][4];
        int ones[VP8HIP_MAX_BATCH];
        for (int i = 0; i < b->n; ++i) {
            ones[i] = 1;
            for (int k = 0; k < 4; ++k) kq[i][k] = b->d[i]->altrefqi[k];
        }
        DRV_CHK(vp8hip_batch_auto_segments(b->hb, key, ones, kq, b->d[0]->qi_min));   // prepare_segments_data(), vp8enc.cpp:379-383
        DRV_CHK(vp8hip_batch_intra_transform(b->hb, key));                            // intra_transform() + prepare_filter_mask
        DRV_CHK(vp8hip_batch_loop_filter(b->hb, key));
        for (int i = 0; i < b->n; ++i) {
            if (!key[i]) continue;
            vp8drv *d = b->d[i];
            d->scene.last_key_detect = d->gop.frame_number;       // intra_part.h:1093
            vp8host_gop_key_coded(&d->gop);
            vp8host_gop_frame_done(&d->gop);
            d->st.key_frames++;
            d->st.frame_number = d->gop.frame_number;
            d->have_frame = true;
            d->last_key = d->last_altref = true;
            d->checked = false;
            d->replaced = 0;
            d->sharpness = VP8HIP_SHARPNESS_ON_DEVICE;
        }
    }
    for (int i = 0; i < b->n; ++i) {
        vp8drv *d = b->d[i];
        if (members && !members[i]) continue;
        if (key[i]) {   // a single key frame: the member's ordinary path, on the shared stream
            if (n_key > 1) continue;
            const int rc = key_frame(d, nullptr);
            if (rc < 0) return rc;
            continue;
        }
        ++n_inter;
        const int32_t *q = d->gop.current_is_altref ? d->altrefqi : d->lastqi;   // vp8enc.cpp:149-151
        for (int k = 0; k < 4; ++k) refqi[i][k] = q[k];
        int32_t g = 0, a = 0;
        vp8host_gop_inter_flags(&d->gop, &g, &a);                               // inter_part.h:103-104
        ug[i] = g & (d->cfg.ref_mask & 1);
        ua[i] = a & ((d->cfg.ref_mask >> 1) & 1);
        pg[i] = d->gop.prev_is_golden;
        pa[i] = d->gop.prev_is_altref;
        d->sharpness = VP8HIP_SHARPNESS_ON_DEVICE;
    }
    if (!n_inter) return VP8HIP_OK;
    DRV_CHK(vp8hip_batch_auto_segments(b->hb, active, zero, refqi, b->d[0]->qi_min));   // vp8enc.cpp:390, 419
    DRV_CHK(vp8hip_batch_inter_transform(b->hb, active, pg, pa, ug, ua));
    const bool check = b->d[0]->cfg.check_ssim != 0;
    if (check) DRV_CHK(vp8hip_batch_check_ssim_async(b->hb, active, refqi, b->d[0]->qi_min));   // vp8enc.cpp:442, nobody waiting
    DRV_CHK(vp8hip_batch_loop_filter(b->hb, active));
    for (int i = 0; i < b->n; ++i) {
        if (!active[i]) continue;
        vp8drv *d = b->d[i];
        d->st.last_use_golden = ug[i];
        d->st.last_use_altref = ua[i];
        d->st.refs_searched += 1 + ug[i] + ua[i];
        d->st.last_prev_is_golden = pg[i];
        d->st.last_prev_is_altref = pa[i];
        d->st.last_was_altref = d->gop.current_is_altref;
        d->checked = false;
        d->replaced = 0;
        if (check) d->verdict_pending = true;   // counted when the verdict is in (resolve())
        else inter_frame_done(d);
        d->have_frame = true;
        d->last_key = false;
        d->last_altref = d->st.last_was_altref != 0;
    }
    return VP8HIP_OK;
}

int vp8drv_batch_encode_frame_device(vp8drv_batch *b, const int *members, const void *const *y, const void *const *u, const void *const *v,
                                     const int *force_key, int *was_key) {
    return batch_encode_frame(b, members, y, u, v, force_key, was_key, false);
}
int vp8drv_batch_encode_frame_host(vp8drv_batch *b, const int *members, const void *const *y, const void *const *u, const void *const *v,
                                   const int *force_key, int *was_key) {
    return batch_encode_frame(b, members, y, u, v, force_key, was_key, true);
}

int vp8drv_batch_prefetch_frame_host(vp8drv_batch *b, const uint8_t *const *y, const uint8_t *const *u, const uint8_t *const *v) {
    if (!b) return VP8HIP_ERR_ARG;
    return vp8hip_batch_prefetch_current(b->hb, y, u, v);
}

int vp8drv_resolve(vp8drv *d) {
    if (!d) return VP8HIP_ERR_ARG;
    const int rc = resolve(d);
    return rc < 0 ? rc : (d->have_frame && d->last_key ? 1 : 0);
}

int vp8drv_ready(const vp8drv *d) { return !d || !d->verdict_pending || vp8hip_check_ssim_ready(d->hip); }
int vp8drv_batch_ready(const vp8drv_batch *b) {
    if (!b) return 1;
    for (int i = 0; i < b->n; ++i)
        if (!vp8drv_ready(b->d[i])) return 0;
    return 1;
}

int vp8drv_batches_encode_frame_device(vp8drv_batch *const *batches, int nbatches, const void *const *const *y, const void *const *const *u,
                                       const void *const *const *v, int *const *was_key) {
    if (!batches || nbatches < 1 || nbatches > 64 || !y || !u || !v) return VP8HIP_ERR_ARG;
    // In the order of the array, each batch waiting for its own verdicts.  (Serving whichever batch is ready first was tried
    // and is slower -- 55-58.8 against 58.8-59.5 M MB/s on the same box: a fixed order keeps the batches evenly staggered, and it
    // is the staggering that lets one batch's latency-bound loop filter run beside the others' searches.)
#ifdef VP8HIP_EXPERIMENTS
    static const bool ready_first = getenv("VP8DRV_EXPERIMENT_READY_FIRST") != nullptr;
#else
    constexpr bool ready_first = false;
#endif
    if (!ready_first) {
        for (int k = 0; k < nbatches; ++k)
            DRV_CHK(vp8drv_batch_encode_frame_device(batches[k], nullptr, y[k], u[k], v[k], nullptr, was_key ? was_key[k] : nullptr));
        return VP8HIP_OK;
    }
    bool done[64] = {};
    for (int left = nbatches; left > 0;) {
        bool progress = false;
        for (int k = 0; k < nbatches; ++k) {
            if (done[k] || !vp8drv_batch_ready(batches[k])) continue;
            DRV_CHK(vp8drv_batch_encode_frame_device(batches[k], nullptr, y[k], u[k], v[k], nullptr, was_key ? was_key[k] : nullptr));
            done[k] = true;
            --left;
            progress = true;
        }
        if (!progress) __builtin_ia32_pause();
    }
    return VP8HIP_OK;
}

uint64_t vp8drv_frame_check(uint64_t h, const uint8_t *frame, size_t size) {
    // position-dependent: every word is folded into a running value that is rotated and multiplied before the next one, so words
    // that trade places (a misplaced chunk of the parallel coder's output) or changes that cancel in a sum do not go unnoticed
    uint64_t f = h * 0x9E3779B97F4A7C15ull + (uint64_t)size;
    size_t i = 0;
    for (; i + 8 <= size; i += 8) {
        uint64_t w;
        memcpy(&w, frame + i, 8);
        f = (((f << 5) | (f >> 59)) ^ w) * 0x100000001B3ull;
    }
    if (i < size) {
        uint64_t w = 0;
        memcpy(&w, frame + i, size - i);
        f = (((f << 5) | (f >> 59)) ^ w) * 0x100000001B3ull;
    }
    return f;
}

static int batches_encode_frames(vp8drv_batch *const *batches, int nbatches, int nframes, const void *const (*frames)[3], int nd,
                                 const int *const *start, int *const *keys_out, uint64_t *const *bytes_out, uint64_t *const *check_out, bool host) {
    if (!batches || nbatches < 1 || nbatches > 64 || nframes < 0 || !frames || nd < 1 || !start || (check_out && !bytes_out)) return VP8HIP_ERR_ARG;
    for (int k = 0; k < nbatches; ++k)
        if (!batches[k] || batches[k]->n < 1 || !batches[k]->d[0] || !start[k]) return VP8HIP_ERR_ARG;
    std::vector<std::thread> th;
    std::vector<int> rc((size_t)nbatches, VP8HIP_OK);
    // Thread k starts k * 200 us after thread 0.  Batches that start together from an idle device stay in lockstep -- every verdict
    // arrives at the same moment, every next frame is enqueued at the same moment -- and then all the latency-bound loop filters run
    // at once with nothing wide beside them: 55.3-59.0 M MB/s from run to run on one box; with the starts a fraction of a frame
    // apart the batches stay staggered: 60.2-60.6 (800 us apart: 60.1-60.4).  VP8DRV_STAGGER_US overrides (experiments).
    static const int stagger_us = [] { const char *v = getenv("VP8DRV_STAGGER_US"); return v ? atoi(v) : 200; }();
    for (int k = 0; k < nbatches; ++k)
        th.emplace_back([&, k] {
            if (stagger_us > 0) std::this_thread::sleep_for(std::chrono::microseconds((long)k * stagger_us));
            vp8drv_batch *b = batches[k];
            const void *y[VP8HIP_MAX_BATCH], *u[VP8HIP_MAX_BATCH], *v[VP8HIP_MAX_BATCH];
            int key[VP8HIP_MAX_BATCH] = {};
            std::vector<uint8_t> frame;
            if (bytes_out) frame.resize((size_t)b->d[0]->mbs * 1900 + (1u << 20));
            auto encode = [&](int t) {
                for (int i = 0; i < b->n; ++i) {
                    const void *const *f = frames[(start[k][i] + t) % nd];
                    y[i] = f[0]; u[i] = f[1]; v[i] = f[2];
                }
                rc[k] = batch_encode_frame(b, nullptr, y, u, v, nullptr, key, host);
                if (rc[k] == VP8HIP_OK && keys_out && keys_out[k])   // (a failed call may not have filled key[])
                    for (int i = 0; i < b->n; ++i) keys_out[k][i] += key[i];
                if (host && rc[k] == VP8HIP_OK && t + 1 < nframes) {   // the next frame's planes on their way while this one is coded
                    const uint8_t *py[VP8HIP_MAX_BATCH], *pu[VP8HIP_MAX_BATCH], *pv[VP8HIP_MAX_BATCH];
                    for (int i = 0; i < b->n; ++i) {
                        const void *const *f = frames[(start[k][i] + t + 1) % nd];
                        py[i] = static_cast<const uint8_t *>(f[0]); pu[i] = static_cast<const uint8_t *>(f[1]); pv[i] = static_cast<const uint8_t *>(f[2]);
                    }
                    rc[k] = vp8hip_batch_prefetch_current(b->hb, py, pu, pv);
                }
            };
            // A bounded device-side wait that expires inside a frame's loop filter shows in the NEXT frame's verdict at the earliest (the
            // verdict workgroup samples the error word when the filter's launch starts), and nobody comes after the last frame: the call
            // ends with every member's last verdict taken and its stream synchronised, so that a time-out anywhere in the run is THIS
            // call's return value (VP8HIP_ERR_TIMEOUT), not a later call's surprise.
            auto finish = [&] {
                for (int i = 0; i < b->n && rc[k] == VP8HIP_OK; ++i) {
                    const int r = vp8drv_resolve(b->d[i]);
                    rc[k] = r < 0 ? r : vp8hip_synchronize(b->d[i]->hip);
                }
            };
            if (!bytes_out) {
                for (int t = 0; t < nframes && rc[k] == VP8HIP_OK; ++t) encode(t);
                finish();
                return;
            }
            // The frames as bytes: one set of launches for the batch's entropy stage, then every member's read-back -- with frame t + 1
            // ENQUEUED before frame t's bytes are waited for: the stage is a link of the batch's chain (same stream), so frame t + 1's
            // kernels queue up behind it and the stream never runs dry while this thread sleeps on the stage's event, copies six
            // frames and comes back (in the old order -- encode, begin, end, encode -- it did, once per frame and batch).  A frame
            // denser than the coder's scratch could not be coded again once the next frame has overwritten its coefficients, so the
            // scratch is sized for the densest frame there can be (a no-op when the caller has done it: vp8hip_reserve_frame_path_dense).
            for (int i = 0; i < b->n && rc[k] == VP8HIP_OK; ++i) rc[k] = vp8hip_reserve_frame_path_dense(b->d[i]->hip);
            if (nframes > 0 && rc[k] == VP8HIP_OK) encode(0);
            for (int t = 0; t < nframes && rc[k] == VP8HIP_OK; ++t) {
                rc[k] = vp8drv_batch_get_frame_begin(b, nullptr);                       // frame t's type is final here (its verdict is in)
                if (rc[k] == VP8HIP_OK && t + 1 < nframes) encode(t + 1);
                for (int i = 0; i < b->n && rc[k] == VP8HIP_OK; ++i) {
                    size_t size = 0;
                    rc[k] = vp8drv_get_frame_end(b->d[i], frame.data(), frame.size(), &size);
                    if (bytes_out[k]) bytes_out[k][i] += size;
                    if (check_out && check_out[k] && rc[k] == VP8HIP_OK) check_out[k][i] = vp8drv_frame_check(check_out[k][i], frame.data(), size);
                }
            }
            finish();
        });
    for (auto &t : th) t.join();
    for (int k = 0; k < nbatches; ++k)
        if (rc[k] != VP8HIP_OK) return rc[k];
    return VP8HIP_OK;
}

int vp8drv_batches_encode_frames_device(vp8drv_batch *const *batches, int nbatches, int nframes, const void *const (*frames)[3], int nd,
                                        const int *const *start, int *const *keys_out, uint64_t *const *bytes_out, uint64_t *const *check_out) {
    return batches_encode_frames(batches, nbatches, nframes, frames, nd, start, keys_out, bytes_out, check_out, false);
}
int vp8drv_batches_encode_frames_host(vp8drv_batch *const *batches, int nbatches, int nframes, const void *const (*frames)[3], int nd,
                                      const int *const *start, int *const *keys_out, uint64_t *const *bytes_out, uint64_t *const *check_out) {
    return batches_encode_frames(batches, nbatches, nframes, frames, nd, start, keys_out, bytes_out, check_out, true);
}

int vp8drv_get_frame(vp8drv *d, uint8_t *out, size_t capacity, size_t *size) {
    if (!d || !out || !size) return VP8HIP_ERR_ARG;
    if (!d->have_frame) return VP8HIP_ERR_STATE;
    const int P = d->cfg.num_partitions;
    if (P != 1 && P != 2 && P != 4 && P != 8) return VP8HIP_ERR_ARG;
    { const int rc = resolve(d); if (rc < 0) return rc; }    // the frame's type and intra information must be final
    return get_frame(d, out, capacity, size);
}

int vp8drv_get_frame_begin(vp8drv *d) {
    if (!d) return VP8HIP_ERR_ARG;
    if (!d->have_frame || d->cfg.host_bitstream) return VP8HIP_ERR_STATE;
    const int P = d->cfg.num_partitions;
    if (P != 1 && P != 2 && P != 4 && P != 8) return VP8HIP_ERR_ARG;
    { const int rc = resolve(d); if (rc < 0) return rc; }
    const vp8hip_header_params hp = header_params(d);
    return vp8hip_encode_frame_begin(d->hip, P, &hp);
}

// vp8drv_get_frame_begin for the members of a batch in one set of launches; every member then takes its frame with
// vp8drv_get_frame_end
int vp8drv_batch_get_frame_begin(vp8drv_batch *b, const int *members) {
    if (!b) return VP8HIP_ERR_ARG;
    vp8hip_header_params hp[VP8HIP_MAX_BATCH];
    const int P = b->d[0]->cfg.num_partitions;
    if (P != 1 && P != 2 && P != 4 && P != 8) return VP8HIP_ERR_ARG;
    for (int i = 0; i < b->n; ++i) {
        if (members && !members[i]) continue;
        vp8drv *d = b->d[i];
        if (d->cfg.num_partitions != P) return VP8HIP_ERR_ARG;
        if (!d->have_frame || d->cfg.host_bitstream) return VP8HIP_ERR_STATE;
        const int rc = resolve(d);
        if (rc < 0) return rc;
        hp[i] = header_params(d);
    }
    return vp8hip_batch_encode_frame_begin(b->hb, members, P, hp);
}

int vp8drv_get_frame_end(vp8drv *d, uint8_t *out, size_t capacity, size_t *size) {
    if (!d || !out || !size) return VP8HIP_ERR_ARG;
    return vp8hip_encode_frame_end(d->hip, out, capacity, size);
}

int vp8drv_encode_video_device(vp8drv *d, int nframes, const void *const (*frames)[3], int nd, int start, uint8_t *out, size_t capacity,
                               uint32_t *sizes, int *keys) {
    if (!d || nframes < 0 || !frames || nd < 1 || start < 0 || (out && !sizes)) return VP8HIP_ERR_ARG;
    size_t used = 0;
    int nkeys = 0;
    if (!out) {      // no frames out: a frame's verdict is taken by the next call (resolve() at its head), the last one's here
        for (int t = 0; t < nframes; ++t) {
            const void *const *f = frames[(start + t) % nd];
            const int rc = vp8drv_encode_frame_device(d, f[0], f[1], f[2], 0);
            if (rc < 0) return rc;
        }
        const int rc = vp8drv_resolve(d);
        if (rc < 0) return rc;
        if (keys) *keys = -1;     // (not counted here: vp8drv_get_stats has key_frames and redone_as_key)
        // The last frame's filter is still running, and a bounded device-side wait that expires INSIDE it shows only in a later
        // verdict or here (the verdict workgroup samples the error word when the filter's launch starts): nobody comes after the last frame
        return vp8hip_synchronize(d->hip);
    }
    // Frame t + 1 is started before frame t's bytes are taken: a frame denser than the coder's scratch could not be coded again by
    // then (vp8hip_encode_frame_end: VP8HIP_ERR_STATE), so the scratch is sized for the densest frame there can be -- a no-op if the
    // caller has done it.  ~270 MB per context at 1080p, four times that at 4K; allocated behind a synchronisation, so callers that
    // time this call reserve beforehand (vp8hip_reserve_frame_path_dense).
    if (nframes > 0) DRV_CHK(vp8hip_reserve_frame_path_dense(d->hip));
    for (int t = 0; t < nframes; ++t) {
        const void *const *f = frames[(start + t) % nd];
        int rc = vp8drv_encode_frame_device(d, f[0], f[1], f[2], 0);          // frame t under way ...
        if (rc < 0) return rc;
        if (t > 0) {                                                          // ... now frame t - 1's bytes
            size_t n = 0;
            rc = vp8drv_get_frame_end(d, out + used, capacity - used, &n);
            if (rc < 0) return rc;
            sizes[t - 1] = (uint32_t)n;
            used += n;
        }
        rc = vp8drv_get_frame_begin(d);                                       // frame t's type is final (the verdict is in): its stage
        if (rc < 0) return rc;
        rc = vp8drv_resolve(d);
        if (rc < 0) return rc;
        nkeys += rc > 0;
    }
    if (nframes > 0) {
        size_t n = 0;
        const int rc = vp8drv_get_frame_end(d, out + used, capacity - used, &n);
        if (rc < 0) return rc;
        sizes[nframes - 1] = (uint32_t)n;
    }
    if (keys) *keys = nkeys;
    return VP8HIP_OK;
}

}  // extern "C"
