/*
 * ref_host_driver.cpp -- calls into the reference's own HOST code for the intra path (src/intra_part.h:
 * predict_and_transform_mb, test_inter_on_intra; src/vp8enc.cpp: check_SSIM), compiled for x86 by
 * oracle/build_ref.sh into oracle/_ref/libvp8refhost.so.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference keeps these functions `static` inside one translation unit (vp8enc.cpp includes every header),
 * so this file includes that translation unit from where it lies (-I $REF/src) with its main() renamed, fills the
 * encoder's global state (`video`, `frames`) from plain arrays and calls the functions.  No OpenCL call is
 * reached: GOP_size is left at 1, which makes intra_transform/prepare_segments_data skip their uploads
 * (src/intra_part.h:1112, src/vp8enc.cpp:221).  Single-threaded (global state).
 */
#define main vp8enc_reference_main
#include "vp8enc.cpp"
#undef main

#include <stdint.h>
#include <string.h>
#include <vector>

namespace {
std::vector<macroblock_extra_data> g_edata;

void bind(int width, int height, const uint8_t *cy, const uint8_t *cu, const uint8_t *cv, uint8_t *ry, uint8_t *ru,
          uint8_t *rv, int16_t *coeffs, int32_t *parts, int32_t *seg, float *ssim, const int32_t *sd) {
    video.wrk_width = width;
    video.wrk_height = height;
    video.mb_width = width / 16;
    video.mb_height = height / 16;
    video.mb_count = video.mb_width * video.mb_height;
    video.wrk_frame_size_luma = width * height;
    video.wrk_frame_size_chroma = width * height / 4;
    video.GOP_size = 1; /* no uploads */
    video.print_info = 0;
    frames.current_Y = const_cast<uint8_t *>(cy);
    frames.current_U = const_cast<uint8_t *>(cu);
    frames.current_V = const_cast<uint8_t *>(cv);
    frames.reconstructed_Y = ry;
    frames.reconstructed_U = ru;
    frames.reconstructed_V = rv;
    frames.MB = reinterpret_cast<macroblock_coeffs_t *>(coeffs);
    frames.MB_parts = parts;
    frames.MB_segment_id = seg;
    frames.MB_SSIM = ssim;
    g_edata.assign(video.mb_count, macroblock_extra_data());
    frames.e_data = g_edata.data();
    /* quantizer steps exactly as prepare_segments_data derives them (src/vp8enc.cpp:164-187) from the 4x11 ints */
    for (int i = 0; i < 4; ++i) {
        memcpy(&frames.segments_data[i], sd + 11 * i, sizeof(segment_data));
        const segment_data &s0 = frames.segments_data[0];
        const int base = frames.segments_data[i].y_ac_i;
        auto cl = [](int q) { return q > 127 ? 127 : (q < 0 ? 0 : q); };
        frames.y_ac_q[i] = vp8_ac_qlookup[base];
        frames.y_dc_q[i] = vp8_dc_qlookup[cl(base + s0.y_dc_idelta)];
        frames.uv_dc_q[i] = vp8_dc_qlookup[cl(base + s0.uv_dc_idelta)];
        frames.uv_ac_q[i] = vp8_ac_qlookup[cl(base + s0.uv_ac_idelta)];
        if (frames.uv_dc_q[i] > 132) frames.uv_dc_q[i] = 132;
    }
}
}  // namespace

extern "C" {

/* intra_transform's loop, src/intra_part.h:1103-1109.  modes: [mbs][16] */
void ref_intra_transform(int width, int height, const uint8_t *cy, const uint8_t *cu, const uint8_t *cv, const int32_t *sd,
                         uint8_t *ry, uint8_t *ru, uint8_t *rv, int16_t *coeffs, int32_t *parts, int32_t *seg,
                         int32_t *modes) {
    bind(width, height, cy, cu, cv, ry, ru, rv, coeffs, parts, seg, nullptr, sd);
    for (int mb = 0; mb < video.mb_count; ++mb) predict_and_transform_mb(mb);
    for (int mb = 0; mb < video.mb_count; ++mb)
        for (int b = 0; b < 16; ++b) modes[mb * 16 + b] = frames.e_data[mb].mode[b];
}

/* check_SSIM, src/vp8enc.cpp:231-263, after the frame loop set is_inter_mb = 1 (:437-438).
 * in/out: recon planes, coeffs, parts, seg, ssim.  out: is_inter[mbs], modes[mbs][16] (0 where never tested),
 * stats = {replaced, new_SSIM, min1 > 0.95}.  Block 24 of a replaced macroblock is whatever the reference's
 * uninitialised stack held (src/intra_part.h:858,1066): callers must not compare it. */
void ref_check_ssim(int width, int height, float ssim_target, const uint8_t *cy, const uint8_t *cu, const uint8_t *cv,
                    const int32_t *sd, uint8_t *ry, uint8_t *ru, uint8_t *rv, int16_t *coeffs, int32_t *parts,
                    int32_t *seg, float *ssim, int32_t *is_inter, int32_t *modes, int32_t *replaced, float *new_ssim,
                    int32_t *filter_updated) {
    bind(width, height, cy, cu, cv, ry, ru, rv, coeffs, parts, seg, ssim, sd);
    video.SSIM_target = ssim_target;
    for (int mb = 0; mb < video.mb_count; ++mb) frames.e_data[mb].is_inter_mb = 1;
    /* check_SSIM ends in prepare_segments_data(1, 7) when the worst macroblock is above 0.95; that call only
     * needs the frame-type flags and the quantizer ladder, which do not influence what is compared here */
    frames.current_is_key_frame = 0;
    frames.current_is_altref_frame = 0;
    video.qi_min = 0;
    for (int i = 0; i < 4; ++i) video.lastqi[i] = video.altrefqi[i] = frames.segments_data[i].y_ac_i;
    const int sharp_before = video.loop_filter_sharpness = -12345;
    check_SSIM();
    *filter_updated = video.loop_filter_sharpness != sharp_before;
    *replaced = frames.replaced;
    *new_ssim = frames.new_SSIM;
    for (int mb = 0; mb < video.mb_count; ++mb) {
        is_inter[mb] = frames.e_data[mb].is_inter_mb;
        for (int b = 0; b < 16; ++b) modes[mb * 16 + b] = frames.e_data[mb].mode[b];
    }
}

/* count_SSIM_16x16, src/intra_part.h:744-853 */
float ref_count_ssim_16x16(const uint8_t *y1, const uint8_t *u1, const uint8_t *v1, int w1, const uint8_t *y2,
                           const uint8_t *u2, const uint8_t *v2, int w2) {
    return count_SSIM_16x16(y1, u1, v1, w1, y2, u2, v2, w2);
}

/* pick_luma_predictor, src/intra_part.h:252-515.  top: 8 values, left: 4 */
int ref_pick_luma_predictor(const uint8_t *orig, uint8_t *pred, int16_t *resid, const int16_t *top, const int16_t *left,
                            int top_left) {
    return pick_luma_predictor(orig, pred, resid, top, left, (cl_short)top_left);
}
}

/* copy_with_padding, src/encIO.h:141-196: the reference's own function on the caller's planes (source planes tight at
 * src_w x src_h, destination planes tight at w x h, pre-filled by the caller so that what the function never writes shows) */
extern "C" void ref_copy_with_padding(const uint8_t *sy, const uint8_t *su, const uint8_t *sv, int src_w, int src_h, uint8_t *dy,
                                      uint8_t *du, uint8_t *dv, int w, int h) {
    video.src_width = src_w;
    video.src_height = src_h;
    video.wrk_width = w;
    video.wrk_height = h;
    frames.tmp_Y = const_cast<uint8_t *>(sy);
    frames.tmp_U = const_cast<uint8_t *>(su);
    frames.tmp_V = const_cast<uint8_t *>(sv);
    frames.current_Y = dy;
    frames.current_U = du;
    frames.current_V = dv;
    copy_with_padding();
}

/* ---- the host producers of the path's parameters, the reference's own functions (src/vp8enc.cpp, src/init.h) -------------------- */
extern "C" {

/* get_loopfilter_strength, src/vp8enc.cpp:96-127 */
void ref_loopfilter_strength(const uint8_t *y, int w, int h, int32_t *reductor, int32_t *sharpness) {
    video.wrk_width = w;
    video.wrk_height = h;
    video.wrk_frame_size_luma = w * h;
    frames.current_Y = const_cast<uint8_t *>(y);
    int red = 0;
    cl_int sh = 0;
    get_loopfilter_strength(&red, &sh);
    *reductor = red;
    *sharpness = sh;
}

/* prepare_segments_data, src/vp8enc.cpp:129-229 (GOP_size 1: it returns before its uploads, :221).  sd_out: 4 x 11 ints */
void ref_prepare_segments_data(const uint8_t *y, int w, int h, int is_key, int is_altref, const int32_t *lastqi, const int32_t *altrefqi,
                               int qi_min, int update_filter, int shrpnss, int32_t *sd_out, int32_t *sharpness_out) {
    video.wrk_width = w;
    video.wrk_height = h;
    video.wrk_frame_size_luma = w * h;
    video.GOP_size = 1;
    video.qi_min = qi_min;
    frames.current_Y = const_cast<uint8_t *>(y);
    frames.current_is_key_frame = is_key;
    frames.current_is_altref_frame = is_altref;
    for (int i = 0; i < 4; ++i) { video.lastqi[i] = lastqi[i]; video.altrefqi[i] = altrefqi[i]; }
    memset(frames.segments_data, 0, sizeof(frames.segments_data));
    prepare_segments_data(update_filter, shrpnss);
    memcpy(sd_out, frames.segments_data, 4 * sizeof(segment_data));
    *sharpness_out = video.loop_filter_sharpness;
}

/* scene_change, src/vp8enc.cpp:265-311 (its hold-over is a function-static: one sequence per process).  Returns its verdict;
 * last_key_detect goes in and comes out (intra_transform sets it when the key frame is coded: the caller's job here) */
int ref_scene_change(const uint8_t *cur_u, const uint8_t *cur_v, const uint8_t *last_u, const uint8_t *last_v, int n_chroma, int frame_number,
                     int32_t *last_key_detect) {
    video.wrk_frame_size_chroma = n_chroma;
    frames.current_U = const_cast<uint8_t *>(cur_u);
    frames.current_V = const_cast<uint8_t *>(cur_v);
    frames.last_U = const_cast<uint8_t *>(last_u);
    frames.last_V = const_cast<uint8_t *>(last_v);
    frames.frame_number = frame_number;
    frames.last_key_detect = *last_key_detect;
    const int r = scene_change();
    *last_key_detect = frames.last_key_detect;
    return r;
}

/* ParseArgs, src/init.h:1295-1608, on an argument vector of the caller ("-i" and "-o" must be in it): the defaults and the two
 * quantizer ladders it derives.  out = {qi_min, qi_max, GOP_size, altref_range, number_of_partitions, lastqi[4], altrefqi[4]} */
int ref_parse_args(int argc, char **argv, int32_t *out, float *ssim_target) {
    const int rc = ParseArgs(argc, argv);
    out[0] = video.qi_min; out[1] = video.qi_max; out[2] = video.GOP_size; out[3] = video.altref_range; out[4] = (int32_t)video.number_of_partitions;
    for (int i = 0; i < 4; ++i) { out[5 + i] = video.lastqi[i]; out[9 + i] = video.altrefqi[i]; }
    *ssim_target = video.SSIM_target;
    return rc;
}
}

/* write_output_header / write_output_file, src/encIO.h:32-139: an .ivf of the caller's frames written by the reference's own
 * functions the way main() uses them (header first, every frame, the header again with the final count; init.h:104-105 for the
 * time base) */
extern "C" int ref_write_ivf(const char *path, int w, int h, int framerate, int nframes, const uint8_t *const *frame, const int32_t *size) {
    output_file.handle = fopen(path, "wb");
    if (!output_file.handle) return -1;
    video.dst_width = w;
    video.dst_height = h;
    video.framerate = framerate;
    video.timestep = 1;
    video.timescale = 1;
    frames.frame_number = -1;
    write_output_header();
    for (int t = 0; t < nframes; ++t) {
        frames.frame_number = t;
        frames.encoded_frame = const_cast<uint8_t *>(frame[t]);
        frames.encoded_frame_size = size[t];
        write_output_file();
    }
    frames.frame_number = nframes;          // main() has counted the last frame too by the time it rewrites the header (vp8enc.cpp:487-489)
    write_output_header();
    fclose(output_file.handle);
    output_file.handle = nullptr;
    return 0;
}

/* OpenYUV420FileAndParseHeader, src/init.h:1610-1737, on a file of the caller (it also opens the output file: `scratch` is
 * a path it may create).  Returns the function's result; on success the sizes, the frame rate and where the file position
 * stands (= the first frame's samples). */
extern "C" int ref_parse_y4m_header(const char *path, const char *scratch, int32_t *w, int32_t *h, int32_t *fps, int64_t *offset) {
    input_file.path = const_cast<char *>(path);
    output_file.path = const_cast<char *>(scratch);
    video.framerate = 0;
    const int rc = OpenYUV420FileAndParseHeader();
    *w = video.src_width;
    *h = video.src_height;
    *fps = video.framerate;
    *offset = input_file.handle ? (int64_t)ftell(input_file.handle) : -1;
    if (input_file.handle) fclose(input_file.handle);
    if (output_file.handle) fclose(output_file.handle);
    input_file.handle = output_file.handle = nullptr;
    return rc;
}

/* ---- frame header / first partition (src/entropy_host.cpp), container (src/encIO.h) ------------------------------- */
extern "C" {

/* encode_header (src/entropy_host.cpp:709-1256) on plain arrays.  Returns frames.encoded_frame_size.
 * vectors: [MBs][4] (x, y) shorts; modes [MBs][16]; probs/denom [4][8][3][11]; flags = {key, golden, altref} */
int ref_encode_header(int width, int height, int dst_width, int dst_height, const int32_t *flags, const int32_t *sd,
                      int loop_filter_type, int sharpness, int partitions_ind, const int32_t *seg, const int32_t *nz,
                      const int32_t *ref_frame, const int32_t *parts, const int16_t *vectors, const int32_t *is_inter,
                      const int32_t *modes, const uint32_t *probs, const uint32_t *denom, int skip_prob, int replaced,
                      uint8_t *out) {
    video.wrk_width = width;
    video.wrk_height = height;
    video.mb_width = width / 16;
    video.mb_height = height / 16;
    video.mb_count = video.mb_width * video.mb_height;
    video.dst_width = dst_width;
    video.dst_height = dst_height;
    video.loop_filter_type = loop_filter_type;
    video.loop_filter_sharpness = sharpness;
    video.number_of_partitions_ind = partitions_ind;
    frames.current_is_key_frame = flags[0];
    frames.current_is_golden_frame = flags[1];
    frames.current_is_altref_frame = flags[2];
    memcpy(frames.segments_data, sd, sizeof frames.segments_data);
    frames.MB_segment_id = const_cast<int32_t *>(seg);
    frames.MB_non_zero_coeffs = const_cast<int32_t *>(nz);
    frames.MB_reference_frame = const_cast<int32_t *>(ref_frame);
    frames.MB_parts = const_cast<int32_t *>(parts);
    frames.MB_vectors = reinterpret_cast<macroblock_vectors_t *>(const_cast<int16_t *>(vectors));
    g_edata.assign(video.mb_count, macroblock_extra_data());
    for (int mb = 0; mb < video.mb_count; ++mb) {
        g_edata[mb].is_inter_mb = is_inter ? is_inter[mb] : 1;
        for (int b = 0; b < 16; ++b) g_edata[mb].mode[b] = modes ? modes[mb * 16 + b] : 0;
    }
    frames.e_data = g_edata.data();
    memcpy(frames.new_probs, probs, sizeof frames.new_probs);
    memcpy(frames.new_probs_denom, denom, sizeof frames.new_probs_denom);
    frames.skip_prob = skip_prob;
    frames.replaced = replaced;
    encode_header(out);
    return (int)frames.encoded_frame_size;
}
}
