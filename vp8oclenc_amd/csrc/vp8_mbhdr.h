// vp8_mbhdr.h -- the macroblock-header part of the first partition (segment id, skip flag, inter/intra, reference
// frame, motion-vector mode and vectors, or intra modes) as one template over a data view and a sink, compiled
// both for the host (vp8_bitstream.cpp: counts statistics / drives the boolean writer) and for the device
// (kernels_hdr.hip: counts bools / emits (probability, bit) pairs for the parallel boolean coder).
//
// Follows the reference's use of the format (src/entropy_host.cpp): bool_encode_inter_mb_modes_and_mvs :209-443,
// count_mv_probs :542-707, write_mv :125-207, count_mv :445-540 and the per-macroblock loop of encode_header
// :1063-1212.  A sink gets put(p, bit) for every decision, where p is a literal probability (0..255) or
// HDR_SYM + an index into the frame's probability table (the ones the frame header transmits), and
// mv_stat(component, index, bit) for every motion-vector decision (the statistics behind the next mv_prob_update).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) && defined(VP8_MBHDR_DEVICE)
#define VP8_HD __host__ __device__ __forceinline__
#define VP8_TABLE __device__ __constant__ const
#else
#define VP8_HD inline
#define VP8_TABLE static const
#endif

namespace vp8hdr {

// symbolic probabilities: resolved from the frame's table
enum { HDR_SYM = 0x100, SYM_SEG = 0 /*3*/, SYM_SKIP = 3, SYM_INTRA = 4, SYM_LAST = 5, SYM_GF = 6, SYM_YMODE = 7 /*4*/,
       SYM_UVMODE = 11 /*3*/, SYM_MV = 14 /*2 x 19*/, SYM_COUNT = 14 + 38 };
// motion-vector component probabilities: is_short, sign, 7 short-tree nodes, 10 long bits (RFC 6386 section 17.2)
enum { MV_IS_SHORT = 0, MV_SIGN = 1, MV_SHORT = 2, MV_BITS = 9, MV_LONG_WIDTH = 10, MV_PROBS = 19 };

// ---- trees: interior links only (a path never continues from a leaf) ---------------------------------------------------
VP8_TABLE int8_t T_SEGMENT[6] = {2, 4, 0, 0, 0, 0};                                     // section 9.3
VP8_TABLE int8_t T_KF_YMODE[8] = {0, 2, 4, 6, 0, 0, 0, 0};                              // B_PRED = "0"
VP8_TABLE int8_t T_YMODE[8] = {0, 2, 4, 6, 0, 0, 0, 0};                                 // B_PRED = "111"
VP8_TABLE int8_t T_UVMODE[6] = {0, 2, 0, 4, 0, 0};                                      // TM_PRED = "111"
VP8_TABLE int8_t T_BMODE[18] = {0, 2, 0, 4, 0, 6, 8, 12, 0, 10, 0, 0, 0, 14, 0, 16, 0, 0};   // section 11.2
VP8_TABLE int8_t T_MV_REF[8] = {0, 2, 0, 4, 0, 6, 0, 0};                                // zero 0, nearest 10, near 110, new 1110, split 1111
VP8_TABLE int8_t T_SPLIT[6] = {0, 2, 0, 4, 0, 0};                                       // quarters = "10"
VP8_TABLE int8_t T_SUBMV[6] = {0, 2, 0, 4, 0, 0};                                       // left 0, above 10, zero 110, new 111
VP8_TABLE int8_t T_SMALL_MV[14] = {2, 8, 4, 6, 0, 0, 0, 0, 10, 12, 0, 0, 0, 0};         // section 17.1, 3 bits

VP8_TABLE uint8_t P_KF_YMODE[4] = {145, 156, 163, 128};
VP8_TABLE uint8_t P_KF_UVMODE[3] = {142, 114, 183};
VP8_TABLE uint8_t P_BMODE[9] = {120, 90, 79, 133, 87, 85, 80, 111, 151};
VP8_TABLE uint8_t P_SPLIT[3] = {110, 111, 150};
VP8_TABLE uint8_t P_SUBMV[5][3] = {{147, 136, 18}, {106, 145, 1}, {179, 121, 1}, {223, 1, 34}, {208, 1, 1}};
VP8_TABLE uint8_t P_MODE_CONTEXTS[6][4] = {{7, 1, 1, 143}, {14, 18, 14, 107}, {135, 64, 57, 68}, {60, 56, 128, 65}, {159, 134, 128, 34}, {234, 188, 128, 28}};
// B_PRED sub-block modes as tree paths (section 11.2), by the mode numbers of e_data.mode (DC, TM, VE, HE, LD, RD, VR, VL, HD, HU)
VP8_TABLE uint8_t BMODE_BITS[10] = {0, 2, 6, 28, 30, 58, 59, 62, 126, 127};
VP8_TABLE uint8_t BMODE_SIZE[10] = {1, 2, 3, 5, 5, 6, 6, 6, 7, 7};

struct Mv {
    int16_t x, y;
    VP8_HD bool operator==(const Mv &o) const { return x == o.x && y == o.y; }
    VP8_HD bool operator!=(const Mv &o) const { return x != o.x || y != o.y; }
    VP8_HD bool zero() const { return x == 0 && y == 0; }
};

// `size` decisions of the tree `t`, most significant bit of `bits` first (write_symbol, :112-123); p = literal probabilities
template <class Sink>
VP8_HD void tree_lit(Sink &s, const int8_t *t, const uint8_t *p, int bits, int size) {
    int i = 0;
    do {
        const int b = (bits >> --size) & 1;
        s.put(p[i >> 1], b);
        i = t[i + b];
    } while (size);
}
// same with symbolic probabilities HDR_SYM + base + node
template <class Sink>
VP8_HD void tree_sym(Sink &s, const int8_t *t, int base, int bits, int size) {
    int i = 0;
    do {
        const int b = (bits >> --size) & 1;
        s.put(HDR_SYM + base + (i >> 1), b);
        i = t[i + b];
    } while (size);
}

// one vector component in quarter pixels; comp 0 = row (y), 1 = column (x)  (write_mv :125-207, count_mv :445-540)
template <class Sink>
VP8_HD void mv_component(Sink &s, int v, int comp) {
    const int a = v < 0 ? -v : v;
    const int base = HDR_SYM + SYM_MV + comp * MV_PROBS;
    if (a <= 7) {
        s.put(base + MV_IS_SHORT, 0); s.mv_stat(comp, MV_IS_SHORT, 0);
        int i = 0;
        for (int size = 3; size;) {
            const int b = (a >> --size) & 1;
            s.put(base + MV_SHORT + (i >> 1), b); s.mv_stat(comp, MV_SHORT + (i >> 1), b);
            i = T_SMALL_MV[i + b];
        }
        if (a != 0) { s.put(base + MV_SIGN, v < 0); s.mv_stat(comp, MV_SIGN, v < 0); }
    } else {
        s.put(base + MV_IS_SHORT, 1); s.mv_stat(comp, MV_IS_SHORT, 1);
        for (int i = 0; i < 3; ++i) { s.put(base + MV_BITS + i, (a >> i) & 1); s.mv_stat(comp, MV_BITS + i, (a >> i) & 1); }
        for (int i = MV_LONG_WIDTH - 1; i > 3; --i) { s.put(base + MV_BITS + i, (a >> i) & 1); s.mv_stat(comp, MV_BITS + i, (a >> i) & 1); }
        if (a & 0xFFF0) { s.put(base + MV_BITS + 3, (a >> 3) & 1); s.mv_stat(comp, MV_BITS + 3, (a >> 3) & 1); }   // bit 3 is implied when nothing above it is set
        s.put(base + MV_SIGN, v < 0); s.mv_stat(comp, MV_SIGN, v < 0);
    }
}

// find_near_mvs as the reference restates it (:232-320): census of the above, left and above-left macroblocks.
// A neighbour counts if it is an inter macroblock inside the frame; its vector is its fourth (bottom-right) one.
struct Near {
    Mv best, nearest, near;
    uint8_t p[4];   // probabilities of the mv_ref tree for this macroblock
};
template <class View>
VP8_HD Near near_mvs(const View &v, int mb) {
    const int mbw = v.mbw();
    const int row = mb / mbw, col = mb % mbw;
    // (the list of up to three distinct vectors and the four counts live in NAMED variables picked by selects: as arrays indexed by
    // the running k they were 32 bytes of scratch memory per lane on the device, a dozen round trips per macroblock)
    Mv l0{0, 0}, l1{0, 0}, l2{0, 0}, l3{0, 0};
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int k = 0;   // index of the last distinct vector found
    int split = 0;
#define VP8_NEAR_STEP(N, NBEXPR, WEIGHT)                                                            \
    {                                                                                              \
        const int nbn = (NBEXPR);                                                                  \
        if (nbn >= 0 && v.inter(nbn)) {                                                            \
            const Mv m = v.vec(nbn, 3);                                                            \
            split += (v.parts(nbn) != 0) * (WEIGHT);                                               \
            const Mv lk = k == 0 ? l0 : (k == 1 ? l1 : (k == 2 ? l2 : l3));                        \
            int slot;                                                                              \
            if (m.zero()) {                                                                        \
                slot = (N) == 0 ? k : 0;   /* the first neighbour adds to the current slot (slot 0 then), the others to slot 0 */ \
            } else {                                                                               \
                if ((N) == 0 || m != lk) {                                                         \
                    ++k;                                                                           \
                    if (k == 1) l1 = m; else if (k == 2) l2 = m; else l3 = m;                      \
                }                                                                                  \
                slot = k;                                                                          \
            }                                                                                      \
            c0 += slot == 0 ? (WEIGHT) : 0;                                                        \
            c1 += slot == 1 ? (WEIGHT) : 0;                                                        \
            c2 += slot == 2 ? (WEIGHT) : 0;                                                        \
            c3 += slot == 3 ? (WEIGHT) : 0;                                                        \
        }                                                                                          \
    }
    VP8_NEAR_STEP(0, row > 0 ? mb - mbw : -1, 2)
    VP8_NEAR_STEP(1, col > 0 ? mb - 1 : -1, 2)
    VP8_NEAR_STEP(2, (row > 0 && col > 0) ? mb - mbw - 1 : -1, 1)
#undef VP8_NEAR_STEP
    const Mv lk = k == 0 ? l0 : (k == 1 ? l1 : (k == 2 ? l2 : l3));
    c1 += c3 & (lk == l1 ? 1 : 0);   // three distinct vectors: merge above-left into nearest if equal
    c3 = split;
    if (c2 > c1) {
        const int t = c1; c1 = c2; c2 = t;
        const Mv m = l1; l1 = l2; l2 = m;
    }
    Near r;
    r.best = c1 >= c0 ? l1 : l0;
    r.nearest = l1;
    r.near = l2;
    r.p[0] = P_MODE_CONTEXTS[c0][0];
    r.p[1] = P_MODE_CONTEXTS[c1][1];
    r.p[2] = P_MODE_CONTEXTS[c2][2];
    r.p[3] = P_MODE_CONTEXTS[c3][3];
    return r;
}

// mode and vectors of one inter macroblock
template <class View, class Sink>
VP8_HD void inter_mb(const View &v, int mb, Sink &s) {
    const Near nr = near_mvs(v, mb);
    const int mbw = v.mbw();
    if (v.parts(mb) == 1) {   // SPLITMV, four 8x8 quarters
        tree_lit(s, T_MV_REF, nr.p, 15, 4);
        tree_lit(s, T_SPLIT, P_SPLIT, 2, 2);
        const bool left_ok = mb % mbw > 0 && v.inter(mb - 1), above_ok = mb >= mbw && v.inter(mb - mbw);
        for (int b = 0; b < 4; ++b) {
            const Mv zero{0, 0};
            const Mv left = (b & 1) ? v.vec(mb, b - 1) : (left_ok ? v.vec(mb - 1, b + 1) : zero);
            const Mv above = (b >> 1) ? v.vec(mb, b - 2) : (above_ok ? v.vec(mb - mbw, b + 2) : zero);
            const Mv me = v.vec(mb, b);
            const bool lez = left.zero(), aez = above.zero(), lea = left == above;
            const int ctx = lea ? (lez ? 4 : 3) : (aez ? 2 : (lez ? 1 : 0));
            if (me == left) tree_lit(s, T_SUBMV, P_SUBMV[ctx], 0, 1);
            else if (me == above) tree_lit(s, T_SUBMV, P_SUBMV[ctx], 2, 2);
            else if (me.zero()) tree_lit(s, T_SUBMV, P_SUBMV[ctx], 6, 3);
            else {
                tree_lit(s, T_SUBMV, P_SUBMV[ctx], 7, 3);
                mv_component(s, me.y - nr.best.y, 0);
                mv_component(s, me.x - nr.best.x, 1);
            }
        }
    } else {   // one vector for the macroblock
        const Mv me = v.vec(mb, 3);
        if (me.zero()) tree_lit(s, T_MV_REF, nr.p, 0, 1);
        else if (me == nr.nearest) tree_lit(s, T_MV_REF, nr.p, 2, 2);
        else if (me == nr.near) tree_lit(s, T_MV_REF, nr.p, 6, 3);
        else {
            tree_lit(s, T_MV_REF, nr.p, 14, 4);
            mv_component(s, me.y - nr.best.y, 0);
            mv_component(s, me.x - nr.best.x, 1);
        }
    }
}

// macroblock_header() of one macroblock (encode_header :1063-1212).  kf_bmode: [10][10][9] (RFC 6386 section 11.5)
template <class View, class Sink>
VP8_HD void mb_header(const View &v, int mb, bool key, const uint8_t (*kf_bmode)[10][9], Sink &s) {
    if (!key) tree_sym(s, T_SEGMENT, SYM_SEG, v.seg(mb), 2);   // segmentation is on for inter frames only (:783)
    s.put(HDR_SYM + SYM_SKIP, v.nz(mb) == 0);
    const bool inter = !key && v.inter(mb);
    if (!key) s.put(HDR_SYM + SYM_INTRA, inter);
    if (inter) {
        const int ref = v.ref(mb);
        s.put(HDR_SYM + SYM_LAST, ref != 0);
        if (ref != 0) s.put(HDR_SYM + SYM_GF, ref == 2);
        inter_mb(v, mb, s);
        return;
    }
    const int mbw = v.mbw();
    if (key) {
        tree_lit(s, T_KF_YMODE, P_KF_YMODE, 0, 1);   // B_PRED
        for (int b = 0; b < 16; ++b) {
            // contexts: the sub-block above and the one to the left, B_DC_PRED outside the frame (section 11.3)
            int above = 0, left = 0;
            if (b >= 4) above = v.mode(mb, b - 4);
            else if (mb >= mbw) above = v.mode(mb - mbw, b + 12);
            if (b & 3) left = v.mode(mb, b - 1);
            else if (mb % mbw) left = v.mode(mb - 1, b + 3);
            const int m = v.mode(mb, b);
            tree_lit(s, T_BMODE, kf_bmode[above][left], BMODE_BITS[m], BMODE_SIZE[m]);
        }
        tree_lit(s, T_UVMODE, P_KF_UVMODE, 7, 3);    // TM_PRED
    } else {
        tree_sym(s, T_YMODE, SYM_YMODE, 7, 3);       // B_PRED
        for (int b = 0; b < 16; ++b) {
            const int m = v.mode(mb, b);
            tree_lit(s, T_BMODE, P_BMODE, BMODE_BITS[m], BMODE_SIZE[m]);
        }
        tree_sym(s, T_UVMODE, SYM_UVMODE, 7, 3);     // TM_PRED
    }
}

}  // namespace vp8hdr
