// vp8_driver.cpp -- the reference's frame loop (main(), src/vp8enc.cpp:351-488) for the inter-frame path, native
// host code over the C ABI (include/vp8hip_driver.h).  No GPU code here: it only sequences vp8hip_* calls with
// the parameters the host mirror (vp8_host.cpp) or the device (vp8hip_auto_segments) produces.
#include <new>
#include <vector>

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
    std::vector<float> ssim;   // check_SSIM read-back
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
    if (cfg->check_ssim) d->ssim.resize(d->mbs);
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

// the loop body after the current frame is on the device; host_y: the caller's luma plane or nullptr
int frame_body(vp8drv *d, const uint8_t *host_y, bool key, const void *y, const void *u, const void *v, bool on_device) {
#define DRV_CHK(call)               \
    do {                            \
        const int rc_ = (call);     \
        if (rc_ != VP8HIP_OK) return rc_; \
    } while (0)
    if (key) {
        // stand-in for intra_transform() (intra_part.h:1089-1128): the source planes become the reconstruction,
        // i.e. LAST (= GOLDEN = ALTREF after a key frame, intra_part.h:1091-1098)
        vp8host_gop_key_coded(&d->gop);
        if (on_device) DRV_CHK(vp8hip_set_last_device(d->hip, y, u, v));
        else DRV_CHK(vp8hip_upload_last(d->hip, (const uint8_t *)y, (const uint8_t *)u, (const uint8_t *)v));
        vp8host_gop_frame_done(&d->gop);
        d->st.key_frames++;
        d->st.frame_number = d->gop.frame_number;
        return 1;
    }
    // vp8enc.cpp:386-388
    if (on_device) DRV_CHK(vp8hip_set_current_device(d->hip, y, u, v));
    else DRV_CHK(vp8hip_upload_current(d->hip, (const uint8_t *)y, (const uint8_t *)u, (const uint8_t *)v));
    // vp8enc.cpp:390, 419: loop-filter strength of the current frame -> segment data
    const int32_t *refqi = d->gop.current_is_altref ? d->altrefqi : d->lastqi;   // vp8enc.cpp:149-151
    int32_t red = 0, sharp = 0, sd[VP8HIP_SD_INTS];
    const bool dev_params = d->cfg.device_params || !host_y;
    if (dev_params) {
        DRV_CHK(vp8hip_auto_segments(d->hip, 0, refqi, d->qi_min));
    } else {
        vp8host_loopfilter_strength(host_y, d->W, d->H, &red, &sharp);
        vp8host_prepare_segments_data(0, refqi, d->qi_min, red, sharp, 0, 0, sd);
        DRV_CHK(vp8hip_set_segments(d->hip, sd));
    }
    int32_t use_golden = 0, use_altref = 0;
    vp8host_gop_inter_flags(&d->gop, &use_golden, &use_altref);                  // inter_part.h:103-104
    DRV_CHK(vp8hip_inter_transform(d->hip, d->gop.prev_is_golden, d->gop.prev_is_altref, use_golden, use_altref));
    d->st.last_use_golden = use_golden;
    d->st.last_use_altref = use_altref;
    d->st.last_prev_is_golden = d->gop.prev_is_golden;
    d->st.last_prev_is_altref = d->gop.prev_is_altref;
    d->st.last_was_altref = d->gop.current_is_altref;
    if (d->cfg.check_ssim) {
        // check_SSIM, vp8enc.cpp:231-263: only its filter-parameter update belongs to this path -- if even the
        // worst macroblock is above 0.95 the loop filter is weakened (reductor * 2, sharpness 7, :155-159, :260-261)
        vp8hip_results r{};
        r.MB_SSIM = d->ssim.data();
        DRV_CHK(vp8hip_download_results(d->hip, &r));
        float min1 = 2.0f;
        for (int i = 0; i < d->mbs; ++i) min1 = d->ssim[i] < min1 ? d->ssim[i] : min1;
        if (min1 > 0.95f) {
            if (dev_params) DRV_CHK(vp8hip_get_segments(d->hip, sd, &red, &sharp));
            vp8host_prepare_segments_data(0, refqi, d->qi_min, red, sharp, 1, 7, sd);
            DRV_CHK(vp8hip_set_segments(d->hip, sd));
        }
    }
    // prepare_filter_mask (loop_filter.h:25-55) was produced by vp8hip_inter_transform for its own coefficients;
    // the host did not touch them here, so do_loop_filter (loop_filter.h:185-190) follows directly
    DRV_CHK(vp8hip_loop_filter(d->hip));
    vp8host_gop_frame_done(&d->gop);
    d->st.inter_frames++;
    d->st.frame_number = d->gop.frame_number;
    return 0;
#undef DRV_CHK
}

}  // namespace

extern "C" {

int vp8drv_encode_frame_device(vp8drv *d, const void *y, const void *u, const void *v, int force_key) {
    if (!d || !y || !u || !v) return VP8HIP_ERR_ARG;
    vp8host_gop_next(&d->gop);
    return frame_body(d, nullptr, d->gop.current_is_key || force_key, y, u, v, true);
}

int vp8drv_encode_frame_host(vp8drv *d, const uint8_t *y, const uint8_t *u, const uint8_t *v, int force_key) {
    if (!d || !y || !u || !v) return VP8HIP_ERR_ARG;
    vp8host_gop_next(&d->gop);
    return frame_body(d, y, d->gop.current_is_key || force_key, y, u, v, false);
}

}  // extern "C"
