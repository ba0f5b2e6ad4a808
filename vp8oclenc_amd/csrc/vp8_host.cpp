// vp8_host.cpp -- host-side mirror of the reference's parameter producers and frame sequencing
// (include/vp8hip_host.h).  Plain C++, no HIP calls: usable (and tested) without a GPU.
#include "../../include/vp8hip_host.h"

namespace {

// src/vp8enc.h:17-39 (same tables as the kernels use)
const int dc_q[128] = {
    4,   5,   6,   7,   8,   9,   10,  10,  11,  12,  13,  14,  15,  16,  17,  17,  18,  19,  20,  20,  21,  21,
    22,  22,  23,  23,  24,  25,  25,  26,  27,  28,  29,  30,  31,  32,  33,  34,  35,  36,  37,  37,  38,  39,
    40,  41,  42,  43,  44,  45,  46,  46,  47,  48,  49,  50,  51,  52,  53,  54,  55,  56,  57,  58,  59,  60,
    61,  62,  63,  64,  65,  66,  67,  68,  69,  70,  71,  72,  73,  74,  75,  76,  76,  77,  78,  79,  80,  81,
    82,  83,  84,  85,  86,  87,  88,  89,  91,  93,  95,  96,  98,  100, 101, 102, 104, 106, 108, 110, 112, 114,
    116, 118, 122, 124, 126, 128, 130, 132, 134, 136, 138, 140, 143, 145, 148, 151, 154, 157};

inline int clamp_qi(int q) { return q > 127 ? 127 : (q < 0 ? 0 : q); }

}  // namespace

extern "C" {

void vp8host_quantizer_ladders(int qi_min, int qi_max, int32_t lastqi[4], int32_t altrefqi[4]) {
    if (qi_max < qi_min) {  // init.h:1585-1592
        const int t = qi_max;
        qi_max = qi_min;
        qi_min = t;
    }
    lastqi[0] = (qi_max + qi_min * 3 + 2) / 4;  // init.h:1593-1596
    lastqi[1] = (qi_max + qi_min + 1) / 2;
    lastqi[2] = (qi_max * 3 + qi_min + 2) / 4;
    lastqi[3] = qi_max;
    altrefqi[0] = lastqi[0] / 4;  // init.h:1598-1603
    altrefqi[1] = lastqi[1] / 3;
    altrefqi[2] = lastqi[2] / 3;
    altrefqi[3] = lastqi[3] / 2;
    if (altrefqi[0] < qi_min) altrefqi[0] = qi_min;
}

void vp8host_loopfilter_strength(const uint8_t *y, int width, int height, int32_t *reductor, int32_t *sharpness) {
    // vp8enc.cpp:96-127.  The reference's accumulators are `int`; the second one overflows on large noisy
    // frames.  They are kept modulo 2^32 here -- what that overflow does on every compiler the reference was
    // built with, and order-independent, so the device reduction (kernels_rc.hip) can match it.
    const int n = width * height;
    uint32_t sum = 0;
    for (int i = 0; i < n; ++i) sum += y[i];
    int avg = (int32_t)sum;
    avg += n / 2;
    avg /= n;
    *reductor = (avg * 5 / 255) + 3;
    uint32_t acc = 0;
    for (int i = 1; i < height - 1; ++i)
        for (int j = 1; j < width - 1; ++j) {
            const int p = i * width + j;
            int a = y[p - width - 1] + y[p - width] + y[p - width + 1] + y[p - 1] + y[p + 1] + y[p + width - 1] +
                    y[p + width] + y[p + width + 1];
            a /= 8;
            acc += (uint32_t)((y[p] - a) * (y[p] - a));
        }
    int div = (int32_t)acc;
    div += (height - 1) * (width - 1) / 2;
    div /= (height - 1) * (width - 1);
    int sh = div / 8;
    *sharpness = sh > 7 ? 7 : sh;
}

// OpenYUV420FileAndParseHeader, init.h:1610-1737, as a walk over a buffer: `next` stands for its fread of one char
int vp8host_y4m_parse_header(const uint8_t *data, size_t size, int32_t *width, int32_t *height, int32_t *framerate, size_t *first_frame_offset) {
    if (!data || !width || !height || !framerate || !first_frame_offset) return -1;
    static const char magic[] = "YUV4MPEG2 ";
    size_t j = 0;
    int ch = 0;
    auto next = [&]() -> bool {
        if (j >= size) return false;
        ch = data[j++];
        return true;
    };
    // The reference accumulates whatever bytes follow a tag (ch - 0x30, no digit check) in an int; here the same arithmetic
    // modulo 2^32 (what its overflow does, without the undefined behaviour), and a size outside the format's 14 bits is
    // refused at the end -- the one deviation: the reference would go on with it.
    auto digit = [&](int &acc) -> bool {
        acc = (int)((uint32_t)acc * 10u + (uint32_t)(ch - 0x30));
        return true;
    };
    int w = 0, h = 0, fps = 0;
    for (int i = 0; i < 10; ++i) {
        if (!next() || ch != magic[i]) return -1;
    }
    for (int i = 0; i < 3; ++i) {                    // three tags, whichever of W / H / F come first (:1634-1690)
        while (ch != 'W' && ch != 'H' && ch != 'F')
            if (!next()) return -1;
        if (ch == 'W') {
            for (;;) {
                if (!next()) return -1;
                if (ch == 0x20) break;
                if (!digit(w)) return -1;
            }
        } else if (ch == 'H') {
            for (;;) {
                if (!next()) return -1;
                if (ch == 0x20) break;
                if (!digit(h)) return -1;
            }
        } else {
            int num = 0, denom = 0;
            for (;;) {
                if (!next()) return -1;
                if (ch == ':') break;
                if (!digit(num)) return -1;
            }
            for (;;) {
                if (!next()) return -1;
                if (ch == 0x20) break;
                if (!digit(denom)) return -1;
            }
            if (denom == 0) return -1;               // the reference divides by it (:1688)
            fps = (num + denom / 2) / denom;
        }
    }
    if (w + h == 0) return -1;
    if (w < 1 || h < 1 || w > 16383 || h > 16383) return -1;   // RFC 6386 section 9.1: 14 bits each
    for (;;) {                                       // the first "FRAME" followed by a line feed (:1696-1728)
        while (ch != 'F')
            if (!next()) return -1;
        if (!next()) return -1;
        if (ch != 'R') continue;
        if (!next()) return -1;
        if (ch != 'A') continue;
        if (!next()) return -1;
        if (ch != 'M') continue;
        if (!next()) return -1;
        if (ch != 'E') continue;
        if (!next()) return -1;
        if (ch != 0x0A) return -1;
        break;
    }
    *width = w;
    *height = h;
    *framerate = fps;
    *first_frame_offset = j;
    return 0;
}

int vp8host_y4m_frame_marker_ok(const uint8_t m[6]) { return m && m[0] == 'F' && m[4] == 'E'; }   // encIO.h:245

int vp8host_scene_change(vp8host_scene_state *st, int Udiff, int Vdiff, int frame_number) {
    // vp8enc.cpp:285-310
    const int detect = (Udiff > 7) || (Vdiff > 7) || (Udiff + Vdiff > 10);
    const bool recent = (frame_number - st->last_key_detect) < 4;   // "workaround to exclude serial intra_frames"
    if (detect && recent) {
        st->last_key_detect = frame_number;
        st->holdover = 1;
        return 0;
    }
    if (detect) return 1;            // last_key_detect is set when the key frame is coded
    if (st->holdover && recent) return 0;
    if (st->holdover) {
        st->holdover = 0;
        return 1;
    }
    return 0;
}

void vp8host_prepare_segments_data(int is_key_frame, const int32_t refqi[4], int qi_min, int reductor, int sharpness,
                                   int update_filter, int shrpnss, int32_t sd[44]) {
    for (int i = 0; i < 44; ++i) sd[i] = 0;
    // segment 0 carries the deltas shared by all segments, vp8enc.cpp:133-148
    sd[1] = 15;                      // y_dc_idelta
    sd[2] = 0;                       // y2_dc_idelta
    sd[3] = 0;                       // y2_ac_idelta
    sd[4] = is_key_frame ? 0 : -15;  // uv_dc_idelta
    sd[5] = is_key_frame ? 0 : -15;  // uv_ac_idelta
    if (update_filter) {             // vp8enc.cpp:155-159
        reductor *= 2;
        sharpness = shrpnss;
    }
    for (int i = 0; i < 4; ++i) {
        int32_t *s = sd + 11 * i;
        s[0] = is_key_frame ? qi_min : refqi[i];  // y_ac_i, :164
        const int y_dc_q = dc_q[clamp_qi(s[0] + sd[1])];
        int lvl = y_dc_q / reductor;  // :187-189
        lvl = lvl > 63 ? 63 : (lvl < 0 ? 0 : lvl);
        s[6] = lvl;
        int il = lvl;  // :192-199
        if (sharpness) {
            il >>= sharpness > 4 ? 2 : 1;
            if (il > 9 - sharpness) il = 9 - sharpness;
        }
        if (!il) il = 1;
        s[9] = il;
        s[7] = ((lvl + 2) * 2) + il;  // mbedge_limit
        s[8] = (lvl * 2) + il;        // sub_bedge_limit
        s[10] = 0;                    // hev_threshold, :204-220
        if (is_key_frame) {
            if (lvl >= 40) s[10] = 2;
            else if (lvl >= 15) s[10] = 1;
        } else {
            if (lvl >= 40) s[10] = 3;
            else if (lvl >= 20) s[10] = 2;
            else if (lvl >= 15) s[10] = 1;
        }
    }
}

int vp8host_skip_prob(const int32_t *nz, int mb_count) {
    int p = 0;
    for (int i = 0; i < mb_count; ++i)
        if (nz[i] > 0) ++p;
    p *= 256;
    p /= mb_count;
    p = p > 254 ? 254 : p;
    return p < 2 ? 2 : p;
}

void vp8host_gop_init(vp8host_gop *g, int gop_size, int altref_range) {
    *g = vp8host_gop{};
    g->gop_size = gop_size;
    g->altref_range = altref_range;
    g->frames_until_key = 1;     // vp8enc.cpp:340-344
    g->frames_until_altref = 2;
    g->frame_number = 0;
    g->golden_frame_number = -1;
    g->altref_frame_number = -1;
}

void vp8host_gop_next(vp8host_gop *g) {  // vp8enc.cpp:364-374
    g->prev_is_key = g->current_is_key;
    g->prev_is_golden = g->current_is_golden;
    g->prev_is_altref = g->current_is_altref;
    --g->frames_until_key;
    --g->frames_until_altref;
    g->current_is_key = g->frames_until_key < 1;
    g->current_is_golden = g->current_is_key;
    g->current_is_altref = (g->frames_until_altref < 1) || g->current_is_key;
    g->frames_until_altref = ((g->frames_until_altref < 1) || g->current_is_key) ? g->altref_range : g->frames_until_altref;
    g->golden_frame_number = g->current_is_golden ? g->frame_number : g->golden_frame_number;
    g->altref_frame_number = g->current_is_altref ? g->frame_number : g->altref_frame_number;
}

void vp8host_gop_key_coded(vp8host_gop *g) {  // intra_part.h:1091-1098
    g->current_is_key = 1;
    g->frames_until_key = g->gop_size;
    g->frames_until_altref = g->altref_range;
    g->current_is_golden = 1;
    g->current_is_altref = 1;
    g->golden_frame_number = g->frame_number;
    g->altref_frame_number = g->frame_number;
}

void vp8host_gop_inter_flags(const vp8host_gop *g, int32_t *use_golden, int32_t *use_altref) {  // inter_part.h:103-104
    *use_golden = !g->prev_is_golden;
    *use_altref = (!g->prev_is_altref) && (g->altref_frame_number != g->golden_frame_number);
}

void vp8host_gop_frame_done(vp8host_gop *g) { ++g->frame_number; }

}  // extern "C"
