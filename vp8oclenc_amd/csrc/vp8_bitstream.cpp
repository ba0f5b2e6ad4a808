// vp8_bitstream.cpp -- first partition (frame header, per-macroblock modes and motion vectors), frame assembly and
// the IVF container: the host half of the reference's entropy stage (include/vp8hip_bitstream.h), byte for byte.
//
// Restated from the bitstream format (RFC 6386 sections 7, 9, 10, 11, 13, 16, 17, 19) the way the reference uses
// it (src/entropy_host.cpp): every inter macroblock is ZERO/NEAREST/NEAR/NEW or SPLITMV with the "quarters" split,
// every intra macroblock is B_PRED + TM_PRED, segment map and quantizers are sent with every inter frame,
// motion-vector probabilities are re-estimated per frame, coefficient probabilities are sent for every context that
// occurred.  Structure is this file's own: one near-vector search shared by the counting and the coding pass,
// one component coder for both vector components, table-driven trees.
#include <string.h>

#include <vector>

#include "../../include/vp8hip_bitstream.h"

namespace {

#include "vp8_rfc6386_tables.inc"

// ---- boolean entropy encoder, RFC 6386 section 7.3 (src/entropy_host.cpp:20-110) -------------------------------------
class BoolWriter {
public:
    explicit BoolWriter(uint8_t *start, size_t capacity) : first_(start), out_(start), end_(start + capacity) {}
    void put(int prob, int bit) {
        const uint32_t split = 1 + (((range_ - 1) * (uint32_t)prob) >> 8);
        if (bit) {
            bottom_ += split;
            range_ -= split;
        } else {
            range_ = split;
        }
        while (range_ < 128) {
            range_ <<= 1;
            if (bottom_ & (1u << 31)) carry();
            bottom_ <<= 1;
            if (!--bit_count_) {
                emit((uint8_t)(bottom_ >> 24));
                bottom_ &= (1u << 24) - 1;
                bit_count_ = 8;
            }
        }
    }
    void flag(int b) { put(128, b ? 1 : 0); }
    void literal(int v, int bits) {
        for (int m = 1 << (bits - 1); m; m >>= 1) flag((v & m) != 0);
    }
    void quantizer_delta(int d) {   // 4-bit magnitude + sign, present flag first (section 9.6)
        if (!d) {
            flag(0);
            return;
        }
        flag(1);
        literal(d < 0 ? -d : d, 4);
        flag(d < 0);
    }
    // `size` decisions of the tree `t`, most significant bit of `bits` first (write_symbol, :112-123)
    void tree(const int8_t *t, const uint8_t *p, int bits, int size) {
        int i = 0;
        do {
            const int b = (bits >> --size) & 1;
            put(p[i >> 1], b);
            i = t[i + b];
        } while (size);
    }
    void finish() {   // flush_bool_encoder, :93-110: always four more bytes
        int c = bit_count_;
        uint32_t v = bottom_;
        if (v & (1u << (32 - c))) carry();
        v <<= c & 7;
        for (c >>= 3; --c >= 0;) v <<= 8;
        for (c = 4; --c >= 0; v <<= 8) emit((uint8_t)(v >> 24));
    }
    size_t count() const { return (size_t)(out_ - first_); }
    bool overflow() const { return overflow_; }

private:
    void carry() {
        uint8_t *q = out_;
        while (q > first_ && *--q == 255) *q = 0;
        if (q >= first_) ++*q;
    }
    void emit(uint8_t b) {
        if (out_ < end_) *out_++ = b;
        else overflow_ = true;
    }
    uint8_t *first_, *out_, *end_;
    uint32_t range_ = 255, bottom_ = 0;
    int bit_count_ = 24;
    bool overflow_ = false;
};

// ---- trees (leaves are never followed by tree(): only the interior links matter) ------------------------------------
const int8_t T_SEGMENT[6] = {2, 4, 0, 0, 0, 0};                                     // section 9.3
const int8_t T_KF_YMODE[8] = {0, 2, 4, 6, 0, 0, 0, 0};                              // B_PRED = "0"
const int8_t T_YMODE[8] = {0, 2, 4, 6, 0, 0, 0, 0};                                 // B_PRED = "111"
const int8_t T_UVMODE[6] = {0, 2, 0, 4, 0, 0};                                      // TM_PRED = "111"
const int8_t T_BMODE[18] = {0, 2, 0, 4, 0, 6, 8, 12, 0, 10, 0, 0, 0, 14, 0, 16, 0, 0};   // section 11.2
const int8_t T_MV_REF[8] = {0, 2, 0, 4, 0, 6, 0, 0};                                // zero 0, nearest 10, near 110, new 1110, split 1111
const int8_t T_SPLIT[6] = {0, 2, 0, 4, 0, 0};                                       // quarters = "10"
const int8_t T_SUBMV[6] = {0, 2, 0, 4, 0, 0};                                       // left 0, above 10, zero 110, new 111
const int8_t T_SMALL_MV[14] = {2, 8, 4, 6, 0, 0, 0, 0, 10, 12, 0, 0, 0, 0};         // section 17.1, 3 bits

const uint8_t P_KF_YMODE[4] = {145, 156, 163, 128};
const uint8_t P_YMODE[4] = {112, 86, 140, 37};
const uint8_t P_KF_UVMODE[3] = {142, 114, 183};
const uint8_t P_UVMODE[3] = {162, 101, 204};
const uint8_t P_ZERO[4] = {0, 0, 0, 0};   // what the reference sends when more than 7 macroblocks were replaced: B_PRED / TM_PRED certain (:1009-1019)
const uint8_t P_BMODE[9] = {120, 90, 79, 133, 87, 85, 80, 111, 151};
const uint8_t P_SPLIT[3] = {110, 111, 150};
const uint8_t P_SUBMV[5][3] = {{147, 136, 18}, {106, 145, 1}, {179, 121, 1}, {223, 1, 34}, {208, 1, 1}};
const uint8_t P_MODE_CONTEXTS[6][4] = {{7, 1, 1, 143}, {14, 18, 14, 107}, {135, 64, 57, 68}, {60, 56, 128, 65}, {159, 134, 128, 34}, {234, 188, 128, 28}};
// B_PRED sub-block modes as tree paths (section 11.2), indexed by the mode numbers of e_data.mode (DC, TM, VE, HE, LD, RD, VR, VL, HD, HU)
const int BMODE_BITS[10] = {0, 2, 6, 28, 30, 58, 59, 62, 126, 127};
const int BMODE_SIZE[10] = {1, 2, 3, 5, 5, 6, 6, 6, 7, 7};

// motion-vector component probabilities: is_short, sign, 7 short-tree nodes, 10 long bits (section 17.2)
enum { MV_IS_SHORT = 0, MV_SIGN = 1, MV_SHORT = 2, MV_BITS = 9, MV_LONG_WIDTH = 10, MV_PROBS = 19 };

struct Mv {
    int16_t x, y;
    bool operator==(const Mv &o) const { return x == o.x && y == o.y; }
    bool operator!=(const Mv &o) const { return !(*this == o); }
    bool zero() const { return x == 0 && y == 0; }
};

struct MvStats { uint32_t num[2][MV_PROBS], den[2][MV_PROBS]; };

// one component: v in quarter pixels; comp 0 = row (y), 1 = column (x).  With a writer it is coded (write_mv,
// :125-207), with statistics its decisions are tallied (count_mv, :445-540): num counts the zero decisions.
void mv_component(int v, int comp, const uint8_t (*probs)[MV_PROBS], BoolWriter *w, MvStats *st) {
    const int a = v < 0 ? -v : v;
    auto decide = [&](int idx, int bit) {
        if (w) w->put(probs[comp][idx], bit);
        if (st) {
            st->num[comp][idx] += 1 - bit;
            st->den[comp][idx] += 1;
        }
    };
    if (a <= 7) {
        decide(MV_IS_SHORT, 0);
        int i = 0;
        for (int size = 3; size;) {
            const int b = (a >> --size) & 1;
            decide(MV_SHORT + (i >> 1), b);
            i = T_SMALL_MV[i + b];
        }
        if (a != 0) decide(MV_SIGN, v < 0);
    } else {
        decide(MV_IS_SHORT, 1);
        for (int i = 0; i < 3; ++i) decide(MV_BITS + i, (a >> i) & 1);
        for (int i = MV_LONG_WIDTH - 1; i > 3; --i) decide(MV_BITS + i, (a >> i) & 1);
        if (a & 0xFFF0) decide(MV_BITS + 3, (a >> 3) & 1);   // bit 3 is implied when nothing above it is set
        decide(MV_SIGN, v < 0);
    }
}

struct FrameView {
    const vp8bs_frame *f;
    int mbw, mbs;
    Mv vec(int mb, int k) const { return Mv{f->MB_vectors[(mb * 4 + k) * 2], f->MB_vectors[(mb * 4 + k) * 2 + 1]}; }
    bool inter(int mb) const { return f->is_inter_mb ? f->is_inter_mb[mb] != 0 : true; }
};

// find_near_mvs as the reference restates it (:232-320): census of the above, left and above-left macroblocks.
// A neighbour counts if it is an inter macroblock inside the frame; its vector is its fourth (bottom-right) one.
struct Near {
    Mv best, nearest, near;
    uint8_t p[4];   // probabilities of the mv_ref tree for this macroblock
};
Near near_mvs(const FrameView &v, int mb) {
    const int row = mb / v.mbw, col = mb % v.mbw;
    const int nb[3] = {row > 0 ? mb - v.mbw : -1, col > 0 ? mb - 1 : -1, row > 0 && col > 0 ? mb - v.mbw - 1 : -1};
    const int weight[3] = {2, 2, 1};
    Mv list[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    int cnt[4] = {0, 0, 0, 0};
    int k = 0;   // index of the last distinct vector found
    int split = 0;
    for (int n = 0; n < 3; ++n) {
        if (nb[n] < 0 || !v.inter(nb[n])) continue;
        const Mv m = v.vec(nb[n], 3);
        split += (v.f->MB_parts[nb[n]] != 0) * weight[n];
        if (m.zero()) {
            cnt[n == 0 ? k : 0] += weight[n];   // the first neighbour adds to the current slot (slot 0 then), the others to slot 0
            continue;
        }
        if (n == 0 || m != list[k]) {
            ++k;
            list[k] = m;
        }
        cnt[k] += weight[n];
    }
    cnt[1] += cnt[3] & (list[k] == list[1]);   // three distinct vectors: merge above-left into nearest if equal
    cnt[3] = split;
    if (cnt[2] > cnt[1]) {
        const int t = cnt[1]; cnt[1] = cnt[2]; cnt[2] = t;
        const Mv m = list[1]; list[1] = list[2]; list[2] = m;
    }
    Near r;
    r.best = cnt[1] >= cnt[0] ? list[1] : list[0];
    r.nearest = list[1];
    r.near = list[2];
    for (int i = 0; i < 4; ++i) r.p[i] = P_MODE_CONTEXTS[cnt[i]][i];
    return r;
}

// mode and vectors of one inter macroblock (bool_encode_inter_mb_modes_and_mvs :209-443 with a writer,
// count_mv_probs :542-707 with statistics only)
void inter_mb(const FrameView &v, int mb, const uint8_t (*mvp)[MV_PROBS], BoolWriter *w, MvStats *st) {
    const Near nr = near_mvs(v, mb);
    auto new_mv = [&](Mv m) {
        mv_component(m.y - nr.best.y, 0, mvp, w, st);
        mv_component(m.x - nr.best.x, 1, mvp, w, st);
    };
    if (v.f->MB_parts[mb] == 1) {   // SPLITMV, four 8x8 quarters
        if (w) {
            w->tree(T_MV_REF, nr.p, 15, 4);
            w->tree(T_SPLIT, P_SPLIT, 2, 2);
        }
        const bool left_ok = mb % v.mbw > 0 && v.inter(mb - 1), above_ok = mb >= v.mbw && v.inter(mb - v.mbw);
        for (int b = 0; b < 4; ++b) {
            const Mv zero{0, 0};
            const Mv left = (b & 1) ? v.vec(mb, b - 1) : (left_ok ? v.vec(mb - 1, b + 1) : zero);
            const Mv above = (b >> 1) ? v.vec(mb, b - 2) : (above_ok ? v.vec(mb - v.mbw, b + 2) : zero);
            const Mv me = v.vec(mb, b);
            const bool lez = left.zero(), aez = above.zero(), lea = left == above;
            const int ctx = lea ? (lez ? 4 : 3) : (aez ? 2 : (lez ? 1 : 0));
            if (me == left) {
                if (w) w->tree(T_SUBMV, P_SUBMV[ctx], 0, 1);
            } else if (me == above) {
                if (w) w->tree(T_SUBMV, P_SUBMV[ctx], 2, 2);
            } else if (me.zero()) {
                if (w) w->tree(T_SUBMV, P_SUBMV[ctx], 6, 3);
            } else {
                if (w) w->tree(T_SUBMV, P_SUBMV[ctx], 7, 3);
                new_mv(me);
            }
        }
    } else {   // one vector for the macroblock
        const Mv me = v.vec(mb, 3);
        if (me.zero()) {
            if (w) w->tree(T_MV_REF, nr.p, 0, 1);
        } else if (me == nr.nearest) {
            if (w) w->tree(T_MV_REF, nr.p, 2, 2);
        } else if (me == nr.near) {
            if (w) w->tree(T_MV_REF, nr.p, 6, 3);
        } else {
            if (w) w->tree(T_MV_REF, nr.p, 14, 4);
            new_mv(me);
        }
    }
}

int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace

extern "C" {

void vp8bs_default_probs(uint32_t *probs, const uint32_t *denom) {
    if (!probs || !denom) return;
    const uint8_t *d = &k_default_coeff_probs[0][0][0][0];
    for (int i = 0; i < VP8BS_NUM_COEFF_PROBS; ++i)
        if (denom[i] < 2) probs[i] = d[i];
}

size_t vp8bs_encode_header(const vp8bs_frame *f, uint8_t *out, size_t capacity, uint8_t *out_mv_probs) {
    if (!f || !out || !f->segments || !f->MB_segment_id || !f->MB_non_zero_coeffs || !f->new_probs || !f->new_probs_denom) return 0;
    const bool key = f->is_key != 0;
    if (!key && (!f->MB_reference_frame || !f->MB_parts || !f->MB_vectors)) return 0;
    const size_t head = key ? 10 : 3;
    if (capacity < head + 8) return 0;
    const FrameView v{f, f->mb_width, f->mb_width * f->mb_height};
    const int mbs = v.mbs;
    const int32_t *sd = f->segments;
    enum { SD = 11, Y_AC_I = 0, Y_DC_D = 1, Y2_DC_D = 2, Y2_AC_D = 3, UV_DC_D = 4, UV_AC_D = 5, LF_LEVEL = 6 };
    BoolWriter w(out + head, capacity - head);

    if (key) {   // colour space, clamping type
        w.flag(0);
        w.flag(0);
    }
    const bool segmentation = !key;
    uint8_t seg_prob[3] = {128, 128, 128};
    w.flag(segmentation);
    if (segmentation) {   // update_segmentation(): map and absolute quantizer / filter level of all four segments, every frame
        w.flag(1);
        w.flag(1);
        w.flag(1);
        for (int i = 0; i < 4; ++i) { w.flag(1); w.literal(sd[i * SD + Y_AC_I], 7); w.flag(0); }
        for (int i = 0; i < 4; ++i) { w.flag(1); w.literal(sd[i * SD + LF_LEVEL], 6); w.flag(0); }
        int c[4] = {0, 0, 0, 0};
        for (int i = 0; i < mbs; ++i) ++c[f->MB_segment_id[i] & 3];
        int d01 = c[0] + c[1], d23 = c[2] + c[3];
        seg_prob[0] = (uint8_t)(d01 * 255 / mbs);
        d01 += d01 == 0;
        d23 += d23 == 0;
        seg_prob[1] = (uint8_t)(c[0] * 255 / d01);
        seg_prob[2] = (uint8_t)(c[2] * 255 / d23);
        for (int i = 0; i < 3; ++i) { w.flag(1); w.literal(seg_prob[i], 8); }
    }
    w.flag(f->loop_filter_type);
    w.literal(sd[LF_LEVEL], 6);
    w.literal(f->loop_filter_sharpness, 3);
    w.flag(0);                                  // no loop-filter adjustments
    w.literal(f->partitions_log2, 2);
    w.literal(sd[Y_AC_I], 7);                   // quant_indices(): segment 0's index and the shared deltas
    w.quantizer_delta(sd[Y_DC_D]);
    w.quantizer_delta(sd[Y2_DC_D]);
    w.quantizer_delta(sd[Y2_AC_D]);
    w.quantizer_delta(sd[UV_DC_D]);
    w.quantizer_delta(sd[UV_AC_D]);
    if (key) {
        w.flag(0);                              // refresh_entropy_probs
    } else {
        w.flag(f->is_golden);                   // refresh_golden_frame, refresh_alternate_frame
        w.flag(f->is_altref);
        if (!f->is_golden) w.literal(0, 2);     // no buffer copies
        if (!f->is_altref) w.literal(0, 2);
        w.flag(0);                              // sign biases
        w.flag(0);
        w.flag(0);                              // refresh_entropy_probs
        w.flag(1);                              // refresh_last
    }
    {   // token_prob_update(): every context that occurred gets its probability
        const uint8_t *upd = &k_coeff_update_probs[0][0][0][0];
        for (int i = 0; i < VP8BS_NUM_COEFF_PROBS; ++i) {
            if (f->new_probs_denom[i] < 2) {
                w.put(upd[i], 0);
            } else {
                w.put(upd[i], 1);
                w.literal((int)f->new_probs[i], 8);
            }
        }
    }
    w.flag(1);                                  // mb_no_skip_coeff
    w.literal(f->skip_prob, 8);

    int prob_intra = 0, prob_last = 0, prob_gf = 0;
    const uint8_t *ymode_p = P_YMODE, *uvmode_p = P_UVMODE;
    uint8_t mvp[2][MV_PROBS];
    memset(mvp, 0, sizeof mvp);
    if (!key) {
        prob_intra = f->replaced * 255 / mbs;
        if (f->replaced > 0 && prob_intra < 2) prob_intra = 2;
        if (f->replaced < mbs && prob_intra > 254) prob_intra = 254;
        w.literal(prob_intra, 8);
        int last = 0, gf = 0;
        for (int i = 0; i < mbs; ++i) {
            last += f->MB_reference_frame[i] == 0;
            gf += f->MB_reference_frame[i] == 1;
        }
        prob_gf = clampi(gf * 256 / (mbs - last + 1), 1, 255);
        prob_last = clampi(last * 256 / mbs, 1, 255);
        w.literal(prob_last, 8);
        w.literal(prob_gf, 8);
        if (f->replaced > 7) {                  // intra mode probabilities: make B_PRED / TM_PRED certain
            w.flag(1);
            ymode_p = P_ZERO;
            for (int i = 0; i < 4; ++i) w.literal(0, 8);
            w.flag(1);
            uvmode_p = P_ZERO;
            for (int i = 0; i < 3; ++i) w.literal(0, 8);
        } else {
            w.flag(0);
            w.flag(0);
        }
        // mv_prob_update(): probabilities of this frame's vector differences
        MvStats st;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < MV_PROBS; ++j) { st.num[i][j] = 0; st.den[i][j] = 1; }
        for (int mb = 0; mb < mbs; ++mb)
            if (v.inter(mb)) inter_mb(v, mb, nullptr, nullptr, &st);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < MV_PROBS; ++j) {
                w.put(k_mv_update_probs[i][j], 1);
                int p = (uint8_t)((st.num[i][j] << 8) / st.den[i][j]);
                p &= ~1;                        // 7 bits are sent
                p = clampi(p, 2, 254);
                mvp[i][j] = (uint8_t)p;
                w.literal(p >> 1, 7);
            }
    }
    if (out_mv_probs) memcpy(out_mv_probs, mvp, sizeof mvp);

    // ---- macroblock headers -------------------------------------------------------------------------------------
    for (int mb = 0; mb < mbs; ++mb) {
        if (segmentation) w.tree(T_SEGMENT, seg_prob, f->MB_segment_id[mb], 2);
        w.put(f->skip_prob, f->MB_non_zero_coeffs[mb] == 0);
        const bool inter = !key && v.inter(mb);
        if (!key) w.put(prob_intra, inter);
        if (inter) {
            const int ref = f->MB_reference_frame[mb];
            w.put(prob_last, ref != 0);
            if (ref != 0) w.put(prob_gf, ref == 2);
            inter_mb(v, mb, mvp, &w, nullptr);
            continue;
        }
        const int32_t *modes = f->modes ? f->modes + 16 * mb : nullptr;
        if (key) {
            w.tree(T_KF_YMODE, P_KF_YMODE, 0, 1);   // B_PRED
            for (int b = 0; b < 16; ++b) {
                // contexts: the sub-block above and the one to the left, B_DC_PRED outside the frame (section 11.3)
                int above = 0, left = 0;
                if (b >= 4) above = modes[b - 4];
                else if (mb >= v.mbw) above = f->modes[16 * (mb - v.mbw) + b + 12];
                if (b & 3) left = modes[b - 1];
                else if (mb % v.mbw) left = f->modes[16 * (mb - 1) + b + 3];
                const int m = modes[b];
                w.tree(T_BMODE, k_kf_bmode_probs[above][left], BMODE_BITS[m], BMODE_SIZE[m]);
            }
            w.tree(T_UVMODE, P_KF_UVMODE, 7, 3);    // TM_PRED
        } else {
            w.tree(T_YMODE, ymode_p, 7, 3);         // B_PRED
            for (int b = 0; b < 16; ++b) {
                const int m = modes ? modes[b] : 0;
                w.tree(T_BMODE, P_BMODE, BMODE_BITS[m], BMODE_SIZE[m]);
            }
            w.tree(T_UVMODE, uvmode_p, 7, 3);       // TM_PRED
        }
    }
    w.finish();
    if (w.overflow()) return 0;
    const size_t first_part = w.count();

    // frame tag: key/inter bit, version 0, show_frame, 19 bits of first-partition size (section 9.1)
    const uint32_t tag = (key ? 0u : 1u) | 0x10u | ((uint32_t)first_part << 5);
    out[0] = (uint8_t)tag;
    out[1] = (uint8_t)(tag >> 8);
    out[2] = (uint8_t)(tag >> 16);
    if (key) {
        out[3] = 0x9d; out[4] = 0x01; out[5] = 0x2a;
        out[6] = (uint8_t)f->width;  out[7] = (uint8_t)(f->width >> 8);     // no upscaling
        out[8] = (uint8_t)f->height; out[9] = (uint8_t)(f->height >> 8);
    }
    return head + first_part;
}

size_t vp8bs_gather_frame(uint8_t *frame, size_t header_size, size_t capacity, int num_partitions, const uint8_t *partitions,
                          size_t partition_step, const int32_t *sizes) {
    if (!frame || !partitions || !sizes || num_partitions < 1 || num_partitions > 8) return 0;
    size_t need = header_size + 3 * (size_t)(num_partitions - 1);
    for (int p = 0; p < num_partitions; ++p) need += (size_t)sizes[p];
    if (need > capacity) return 0;
    size_t n = header_size;
    for (int p = 0; p < num_partitions - 1; ++p) {   // the last partition's size is implied
        frame[n++] = (uint8_t)sizes[p];
        frame[n++] = (uint8_t)(sizes[p] >> 8);
        frame[n++] = (uint8_t)(sizes[p] >> 16);
    }
    for (int p = 0; p < num_partitions; ++p) {
        memcpy(frame + n, partitions + (size_t)p * partition_step, (size_t)sizes[p]);
        n += (size_t)sizes[p];
    }
    return n;
}

static void le(uint8_t *p, uint64_t v, int bytes) {
    for (int i = 0; i < bytes; ++i) p[i] = (uint8_t)(v >> (8 * i));
}

size_t vp8bs_ivf_file_header(uint8_t out[32], int width, int height, uint32_t framerate, uint32_t timescale, uint32_t frame_count) {
    memcpy(out, "DKIF", 4);
    le(out + 4, 0, 2);            // version
    le(out + 6, 32, 2);           // header length
    memcpy(out + 8, "VP80", 4);
    le(out + 12, (uint32_t)width, 2);
    le(out + 14, (uint32_t)height, 2);
    le(out + 16, framerate, 4);
    le(out + 20, timescale, 4);
    le(out + 24, frame_count, 4);
    le(out + 28, 0, 4);
    return 32;
}

size_t vp8bs_ivf_frame_header(uint8_t out[12], uint32_t frame_size, uint64_t timestamp) {
    le(out, frame_size, 4);
    le(out + 4, timestamp, 8);
    return 12;
}

}  // extern "C"
