// vp8_driver.cpp -- the reference's frame loop (main(), src/vp8enc.cpp:351-488), native host code over the C ABI
// (include/vp8hip_driver.h).  No GPU code here: it only sequences vp8hip_* calls with the parameters the host
// mirror (vp8_host.cpp) or the device (vp8hip_auto_segments) produces.
#include <new>

#include "../../include/vp8hip_driver.h"
#include "../../include/vp8hip_host.h"

struct vp8drv {
    vp8hip_ctx *hip = nullptr;
    vp8drv_config cfg{};
    int W = 0, H = 0, mbs = 0;
    vp8host_gop gop{};
    int32_t lastqi[4]{}, altrefqi[4]{};
    int qi_min = 0;
    vp8drv_stats st{};
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
    c->check_ssim = 0;
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
    if (d->cfg.device_params || !host_y) return vp8hip_auto_segments(d->hip, key ? 1 : 0, refqi, d->qi_min);
    int32_t red = 0, sharp = 0, sd[VP8HIP_SD_INTS];
    vp8host_loopfilter_strength(host_y, d->W, d->H, &red, &sharp);
    vp8host_prepare_segments_data(key ? 1 : 0, refqi, d->qi_min, red, sharp, 0, 0, sd);
    return vp8hip_set_segments(d->hip, sd);
}

// prepare_segments_data() + intra_transform() (vp8enc.cpp:379-383, :411-414, :446-450) and the common tail
// (:472-473): filter mask, loop filter.  The filtered key frame is LAST = GOLDEN = ALTREF (intra_part.h:1091-1098).
int key_frame(vp8drv *d, const uint8_t *host_y) {
    DRV_CHK(segments(d, host_y, true, d->altrefqi));
    DRV_CHK(vp8hip_intra_transform(d->hip));
    vp8host_gop_key_coded(&d->gop);
    DRV_CHK(vp8hip_prepare_filter_mask(d->hip, nullptr));
    DRV_CHK(vp8hip_loop_filter(d->hip));
    vp8host_gop_frame_done(&d->gop);
    d->st.key_frames++;
    d->st.frame_number = d->gop.frame_number;
    return 1;
}

// the loop body once the current frame is on the device; host_y: the caller's luma plane or nullptr
int frame_body(vp8drv *d, const uint8_t *host_y, bool key) {
    if (key) return key_frame(d, host_y);
    // vp8enc.cpp:390, 419: loop-filter strength of the current frame -> segment data
    const int32_t *refqi = d->gop.current_is_altref ? d->altrefqi : d->lastqi;   // vp8enc.cpp:149-151
    DRV_CHK(segments(d, host_y, false, refqi));
    int32_t use_golden = 0, use_altref = 0;
    vp8host_gop_inter_flags(&d->gop, &use_golden, &use_altref);                  // inter_part.h:103-104
    DRV_CHK(vp8hip_inter_transform(d->hip, d->gop.prev_is_golden, d->gop.prev_is_altref, use_golden, use_altref));
    d->st.last_use_golden = use_golden;
    d->st.last_use_altref = use_altref;
    d->st.last_prev_is_golden = d->gop.prev_is_golden;
    d->st.last_prev_is_altref = d->gop.prev_is_altref;
    d->st.last_was_altref = d->gop.current_is_altref;
    if (d->cfg.check_ssim) {
        // check_SSIM, vp8enc.cpp:231-263, on the device: intra fallback of the macroblocks below the target
        int32_t replaced = 0;
        float new_ssim = 0.0f, min1 = 2.0f;
        DRV_CHK(vp8hip_check_ssim(d->hip, &replaced, &new_ssim, &min1));
        d->st.last_replaced = replaced;
        d->st.last_new_ssim = new_ssim;
        d->st.last_min_ssim = min1;
        if (min1 > 0.95f) {
            // even the worst macroblock is good: weaken the loop filter (reductor * 2, sharpness 7; :155-159, :260-261)
            int32_t red = 0, sharp = 0, sd[VP8HIP_SD_INTS];
            if (d->cfg.device_params || !host_y) DRV_CHK(vp8hip_get_segments(d->hip, sd, &red, &sharp));
            else vp8host_loopfilter_strength(host_y, d->W, d->H, &red, &sharp);
            vp8host_prepare_segments_data(0, refqi, d->qi_min, red, sharp, 1, 7, sd);
            DRV_CHK(vp8hip_set_segments(d->hip, sd));
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
    vp8host_gop_frame_done(&d->gop);
    d->st.inter_frames++;
    d->st.frame_number = d->gop.frame_number;
    return 0;
}

}  // namespace

extern "C" {

int vp8drv_encode_frame_device(vp8drv *d, const void *y, const void *u, const void *v, int force_key) {
    if (!d || !y || !u || !v) return VP8HIP_ERR_ARG;
    vp8host_gop_next(&d->gop);
    DRV_CHK(vp8hip_set_current_device(d->hip, y, u, v));                          // vp8enc.cpp:386-388
    return frame_body(d, nullptr, d->gop.current_is_key || force_key);
}

int vp8drv_encode_frame_host(vp8drv *d, const uint8_t *y, const uint8_t *u, const uint8_t *v, int force_key) {
    if (!d || !y || !u || !v) return VP8HIP_ERR_ARG;
    vp8host_gop_next(&d->gop);
    DRV_CHK(vp8hip_upload_current(d->hip, y, u, v));
    return frame_body(d, y, d->gop.current_is_key || force_key);
}

}  // extern "C"
