// vp8_bitstream.cpp -- first partition (frame header, per-macroblock modes and motion vectors), frame assembly and
// the IVF container: the host half of the reference's entropy stage (include/vp8hip_bitstream.h), byte for byte.
//
// Restated from the bitstream format (RFC 6386 sections 7, 9, 10, 11, 13, 16, 17, 19) the way the reference uses
// it (src/entropy_host.cpp): every inter macroblock is ZERO/NEAREST/NEAR/NEW or SPLITMV with the "quarters" split,
// every intra macroblock is B_PRED + TM_PRED, segment map and quantizers are sent with every inter frame,
// motion-vector probabilities are re-estimated per frame, coefficient probabilities are sent for every context that
// occurred.  The per-macroblock part is vp8_mbhdr.h, shared with the device coder (kernels_hdr.hip); this file adds
// the frame-level fields, the serial boolean writer and the container.
#include <string.h>

#include "../../include/vp8hip_bitstream.h"
#include "vp8_mbhdr.h"

namespace {

#define VP8_RFC_TABLE static const
#include "vp8_rfc6386_tables.inc"

using namespace vp8hdr;

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
        // Renormalisation, all shifts of this decision at once (the format describes it bit by bit).  With k shifts
        // left until the next byte is complete, `bottom_` stays below 2^(33-k): its bit 31 -- the carry into the
        // bytes already written -- can only be set at the last shift before a byte is emitted, and a decision
        // shifts at most 7 times, so it completes at most one byte.
        const int s = __builtin_clz(range_) - 24;
        if (!s) return;
        range_ <<= s;
        if (s < bit_count_) {
            bottom_ <<= s;
            bit_count_ -= s;
            return;
        }
        const int pre = bit_count_;
        if (bottom_ & (1u << (32 - pre))) carry();
        bottom_ <<= pre;
        emit((uint8_t)(bottom_ >> 24));
        bottom_ &= (1u << 24) - 1;
        bottom_ <<= s - pre;
        bit_count_ = 8 - (s - pre);
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

// the frame's results as vp8_mbhdr.h wants to see them
struct View {
    const vp8bs_frame *f;
    int mbw() const { return f->mb_width; }
    Mv vec(int mb, int k) const { return Mv{f->MB_vectors[(mb * 4 + k) * 2], f->MB_vectors[(mb * 4 + k) * 2 + 1]}; }
    bool inter(int mb) const { return f->is_inter_mb ? f->is_inter_mb[mb] != 0 : true; }
    int parts(int mb) const { return f->MB_parts[mb]; }
    int ref(int mb) const { return f->MB_reference_frame[mb]; }
    int seg(int mb) const { return f->MB_segment_id[mb]; }
    int nz(int mb) const { return f->MB_non_zero_coeffs[mb]; }
    int mode(int mb, int b) const { return f->modes ? f->modes[16 * mb + b] : 0; }
};

// counting pass: the statistics behind mv_prob_update() (count_mv_probs, :542-707); num counts the zero decisions
struct StatSink {
    uint32_t num[2][MV_PROBS], den[2][MV_PROBS];
    void put(int, int) {}
    void mv_stat(int comp, int idx, int bit) {
        num[comp][idx] += 1 - bit;
        den[comp][idx] += 1;
    }
};
// coding pass: resolve the symbolic probabilities and drive the writer
struct WriteSink {
    BoolWriter &w;
    const uint8_t *sym;   // [SYM_COUNT]
    void put(int p, int bit) { w.put(p >= HDR_SYM ? sym[p - HDR_SYM] : p, bit); }
    void mv_stat(int, int, int) {}
};

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
    if (key && !f->modes) return 0;
    const size_t head = key ? 10 : 3;
    if (capacity < head + 8) return 0;
    const View v{f};
    const int mbs = f->mb_width * f->mb_height;
    const int32_t *sd = f->segments;
    enum { SD = 11, Y_AC_I = 0, Y_DC_D = 1, Y2_DC_D = 2, Y2_AC_D = 3, UV_DC_D = 4, UV_AC_D = 5, LF_LEVEL = 6 };
    BoolWriter w(out + head, capacity - head);
    uint8_t sym[SYM_COUNT];   // the frame's probability table (what vp8_mbhdr.h refers to symbolically)
    memset(sym, 128, sizeof sym);

    if (key) {   // colour space, clamping type
        w.flag(0);
        w.flag(0);
    }
    const bool segmentation = !key;
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
        sym[SYM_SEG + 0] = (uint8_t)(d01 * 255 / mbs);
        d01 += d01 == 0;
        d23 += d23 == 0;
        sym[SYM_SEG + 1] = (uint8_t)(c[0] * 255 / d01);
        sym[SYM_SEG + 2] = (uint8_t)(c[2] * 255 / d23);
        for (int i = 0; i < 3; ++i) { w.flag(1); w.literal(sym[SYM_SEG + i], 8); }
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
    sym[SYM_SKIP] = (uint8_t)f->skip_prob;

    static const uint8_t P_YMODE[4] = {112, 86, 140, 37}, P_UVMODE[3] = {162, 101, 204};
    memcpy(sym + SYM_YMODE, P_YMODE, 4);
    memcpy(sym + SYM_UVMODE, P_UVMODE, 3);
    if (!key) {
        int prob_intra = f->replaced * 255 / mbs;
        if (f->replaced > 0 && prob_intra < 2) prob_intra = 2;
        if (f->replaced < mbs && prob_intra > 254) prob_intra = 254;
        w.literal(prob_intra, 8);
        int last = 0, gf = 0;
        for (int i = 0; i < mbs; ++i) {
            last += f->MB_reference_frame[i] == 0;
            gf += f->MB_reference_frame[i] == 1;
        }
        const int prob_gf = clampi(gf * 256 / (mbs - last + 1), 1, 255);
        const int prob_last = clampi(last * 256 / mbs, 1, 255);
        w.literal(prob_last, 8);
        w.literal(prob_gf, 8);
        sym[SYM_INTRA] = (uint8_t)prob_intra;
        sym[SYM_LAST] = (uint8_t)prob_last;
        sym[SYM_GF] = (uint8_t)prob_gf;
        if (f->replaced > 7) {                  // intra mode probabilities: make B_PRED / TM_PRED certain (:1009-1019)
            w.flag(1);
            memset(sym + SYM_YMODE, 0, 4);
            for (int i = 0; i < 4; ++i) w.literal(0, 8);
            w.flag(1);
            memset(sym + SYM_UVMODE, 0, 3);
            for (int i = 0; i < 3; ++i) w.literal(0, 8);
        } else {
            w.flag(0);
            w.flag(0);
        }
        // mv_prob_update(): probabilities of this frame's vector differences
        StatSink st;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < MV_PROBS; ++j) { st.num[i][j] = 0; st.den[i][j] = 1; }
        for (int mb = 0; mb < mbs; ++mb)
            if (v.inter(mb)) inter_mb(v, mb, st);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < MV_PROBS; ++j) {
                w.put(k_mv_update_probs[i][j], 1);
                int p = (uint8_t)((st.num[i][j] << 8) / st.den[i][j]);
                p &= ~1;                        // 7 bits are sent
                p = clampi(p, 2, 254);
                sym[SYM_MV + i * MV_PROBS + j] = (uint8_t)p;
                w.literal(p >> 1, 7);
            }
    }
    if (out_mv_probs) memcpy(out_mv_probs, sym + SYM_MV, 2 * MV_PROBS);

    WriteSink ws{w, sym};
    for (int mb = 0; mb < mbs; ++mb) mb_header(v, mb, key, k_kf_bmode_probs, ws);
    w.finish();
    if (w.overflow()) return 0;
    const size_t first_part = w.count();
    if (first_part >= ((size_t)1 << 19)) return (size_t)-1;   // does not fit the frame tag's 19-bit size field: not a decodable frame

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
