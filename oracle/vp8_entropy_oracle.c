/*
 * vp8_entropy_oracle.c -- CPU restatement of the reference's coefficient entropy stage
 * (count_probs, num_div_denom, encode_coefficients: src/CPU_kernels.cl:347-778; host order
 * src/vp8enc.cpp:48-94).
 *
 * TEST INFRASTRUCTURE ONLY (see vp8_oracle.h): the parity checker for the device token/boolean
 * coder, never linked into the product.  Pinned bit-for-bit against the reference's own kernels
 * compiled by oracle/build_ref.sh (tests/test_entropy_oracle.py) and the golden vectors they
 * produced (tests/golden/entropy_*.npz).
 *
 * The coefficient token alphabet, its tree, the extra-bit probabilities and the boolean coder are
 * those of the VP8 bitstream (RFC 6386 sections 7, 13); the restatement is table-driven where the
 * reference spells the cases out.
 */
#include <stdint.h>
#include <string.h>

#include "vp8_oracle.h"

/* ---- token alphabet -------------------------------------------------------------------------- */
enum { T_ZERO, T_ONE, T_TWO, T_THREE, T_FOUR, T_CAT1, T_CAT2, T_CAT3, T_CAT4, T_CAT5, T_CAT6, T_EOB, T_COUNT };

/* path of a token through the coefficient tree (src/CPU_kernels.cl:181-193) as (node, bit) pairs,
 * node = tree index / 2 = the probability slot; every path starts at node 0 */
typedef struct { int len; uint8_t node[7], bit[7]; } token_path;
static const token_path k_path[T_COUNT] = {
    /* ZERO  "10"      */ {2, {0, 1}, {1, 0}},
    /* ONE   "110"     */ {3, {0, 1, 2}, {1, 1, 0}},
    /* TWO   "11100"   */ {5, {0, 1, 2, 3, 4}, {1, 1, 1, 0, 0}},
    /* THREE "111010"  */ {6, {0, 1, 2, 3, 4, 5}, {1, 1, 1, 0, 1, 0}},
    /* FOUR  "111011"  */ {6, {0, 1, 2, 3, 4, 5}, {1, 1, 1, 0, 1, 1}},
    /* CAT1  "111100"  */ {6, {0, 1, 2, 3, 6, 7}, {1, 1, 1, 1, 0, 0}},
    /* CAT2  "111101"  */ {6, {0, 1, 2, 3, 6, 7}, {1, 1, 1, 1, 0, 1}},
    /* CAT3  "1111100" */ {7, {0, 1, 2, 3, 6, 8, 9}, {1, 1, 1, 1, 1, 0, 0}},
    /* CAT4  "1111101" */ {7, {0, 1, 2, 3, 6, 8, 9}, {1, 1, 1, 1, 1, 0, 1}},
    /* CAT5  "1111110" */ {7, {0, 1, 2, 3, 6, 8, 10}, {1, 1, 1, 1, 1, 1, 0}},
    /* CAT6  "1111111" */ {7, {0, 1, 2, 3, 6, 8, 10}, {1, 1, 1, 1, 1, 1, 1}},
    /* EOB   "0"       */ {1, {0}, {0}},
};
/* categories: smallest magnitude, number of extra bits, their probabilities (:194-199) */
static const int k_cat_base[6] = {5, 7, 11, 19, 35, 67};
static const int k_cat_bits[6] = {1, 2, 3, 4, 5, 11};
static const uint8_t k_cat_prob[6][11] = {
    {159}, {165, 145}, {173, 148, 140}, {176, 155, 140, 135}, {180, 157, 141, 134, 130},
    {254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129}};
static const int k_band[16] = {0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7};   /* :200 */

static int classify(int mag) {          /* tokenize_block, :263-345 */
    if (mag <= 4) return mag;            /* T_ZERO .. T_FOUR */
    if (mag <= 6) return T_CAT1;
    if (mag <= 10) return T_CAT2;
    if (mag <= 18) return T_CAT3;
    if (mag <= 34) return T_CAT4;
    if (mag <= 66) return T_CAT5;
    return T_CAT6;
}

/* tokens of one block: tok[i] for i = first..15; returns the index after the last token (the EOB
 * position + 1, or 16 when the last coefficient is non-zero).  A zero is EOB exactly when nothing
 * non-zero follows it. */
static int tokenize(const int16_t *c, int first, int tok[16]) {
    int last = -1;
    for (int i = 15; i >= first; --i)
        if (c[i] != 0) { last = i; break; }
    for (int i = first; i <= last; ++i) tok[i] = classify(c[i] < 0 ? -c[i] : c[i]);
    if (last < 15) {
        const int e = last < first ? first : last + 1;
        tok[e] = T_EOB;
        return e + 1;
    }
    return 16;
}

/* index of coeff_probs[part][ctx1][ctx2][ctx3][node], :506-507 */
static int pidx(int part, int ctx1, int ctx2, int ctx3, int node) {
    return ((((part << 5) + (ctx1 << 3)) + ctx2) * 3 + ctx3) * 11 + node;
}

/* ---- neighbour context ("third_context"), :560-760 ---------------------------------------------- */
static int block_nonzero(const int16_t *MB, int mb, int b, int first) {
    const int16_t *c = MB + ((size_t)mb * 25 + b) * 16;
    for (int i = first; i < 16; ++i)
        if (c[i]) return 1;
    return 0;
}

static int neighbour_context(const int16_t *MB, const int32_t *parts, int mb, int b, int mb_row, int mb_col, int mbw) {
    int ctx = 0;
    if (b == 24) {
        /* nearest macroblock above / to the left (same row) that has a Y2 block, :575-604 */
        int p;
        if (mb_row > 0) {
            for (p = mb - mbw; p >= 0 && parts[p] != VP8O_16x16; p -= mbw) {}
            if (p >= 0) ctx += block_nonzero(MB, p, 24, 0);
        }
        if (mb_col > 0) {
            for (p = mb - 1; p >= mb_row * mbw && parts[p] != VP8O_16x16; --p) {}
            if (p >= mb_row * mbw) ctx += block_nonzero(MB, p, 24, 0);
        }
        return ctx;
    }
    /* plane geometry: blocks per row of the plane inside a macroblock, first block index */
    const int w = b < 16 ? 4 : 2, base = b < 16 ? 0 : (b < 20 ? 16 : 20);
    const int bx = (b - base) % w, by = (b - base) / w;
    int nmb, nb;
    /* above */
    nmb = -1;
    if (by > 0) { nmb = mb; nb = b - w; }
    else if (mb_row > 0) { nmb = mb - mbw; nb = b + w * (w - 1); }
    if (nmb >= 0) ctx += block_nonzero(MB, nmb, nb, (b < 16 && parts[nmb] == VP8O_16x16) ? 1 : 0);
    /* left */
    nmb = -1;
    if (bx > 0) { nmb = mb; nb = b - 1; }
    else if (mb_col > 0) { nmb = mb - 1; nb = b + (w - 1); }
    if (nmb >= 0) ctx += block_nonzero(MB, nmb, nb, (b < 16 && parts[nmb] == VP8O_16x16) ? 1 : 0);
    return ctx;
}

/* order of the blocks of one macroblock and their plane context (ctx1), :371-403 */
static int block_order(int has_y2, int k, int *ctx1) {
    if (has_y2) {
        if (k == 0) { *ctx1 = 1; return 24; }
        --k;
    }
    if (k < 16) { *ctx1 = has_y2 ? 0 : 3; return k; }
    *ctx1 = 2;
    return k;
}

/* ---- count_probs, :536-762 ------------------------------------------------------------------------- */
void vp8o_count_probs(const int16_t *MB, const int32_t *MB_non_zero_coeffs, const int32_t *MB_parts,
                      uint32_t *coeff_probs, uint32_t *coeff_probs_denom, uint8_t *third_context, int mb_height,
                      int mb_width, int num_partitions) {
    for (int part = 0; part < num_partitions; ++part) {
        for (int i = 0; i < 4 * 8 * 3 * 11; ++i) {
            coeff_probs[pidx(part, 0, 0, 0, 0) + i] = 0;
            coeff_probs_denom[pidx(part, 0, 0, 0, 0) + i] = 1;
        }
        for (int mb_row = part; mb_row < mb_height; mb_row += num_partitions)
            for (int mb_col = 0; mb_col < mb_width; ++mb_col) {
                const int mb = mb_row * mb_width + mb_col;
                if (MB_non_zero_coeffs[mb] == 0) continue;
                const int has_y2 = MB_parts[mb] == VP8O_16x16;
                for (int k = 0; k < 24 + has_y2; ++k) {
                    int ctx1;
                    const int b = block_order(has_y2, k, &ctx1);
                    int ctx3 = neighbour_context(MB, MB_parts, mb, b, mb_row, mb_col, mb_width);
                    third_context[mb * 25 + b] = (uint8_t)ctx3;
                    const int first = ctx1 == 0 ? 1 : 0;
                    int tok[16];
                    const int end = tokenize(MB + ((size_t)mb * 25 + b) * 16, first, tok);
                    int after_zero = 0;
                    /* Reference quirk: unlike encode_block (:238), count_probs_in_block (:478-534) does not
                     * stop at the end-of-block token -- every remaining position is tokenised as EOB too and is
                     * counted, in context 2 (the "neither zero nor one" context an EOB leaves behind). */
                    for (int i = first; i < 16; ++i) {
                        const int t = i < end ? tok[i] : T_EOB;
                        const token_path *p = &k_path[t];
                        for (int s = after_zero; s < p->len; ++s) {      /* after a ZERO the first branch is implied */
                            const int at = pidx(part, ctx1, k_band[i], ctx3, p->node[s]);
                            coeff_probs[at] += 1 - p->bit[s];
                            coeff_probs_denom[at] += 1;
                        }
                        after_zero = t == T_ZERO;
                        ctx3 = t == T_ZERO ? 0 : (t == T_ONE ? 1 : 2);
                    }
                }
            }
    }
}

/* ---- num_div_denom, :764-778: probability of a zero bit per context, summed over the partitions ---- */
void vp8o_num_div_denom(uint32_t *coeff_probs, const uint32_t *coeff_probs_denom, int num_partitions) {
    for (int i = 0; i < 4 * 8 * 3 * 11; ++i) {
        uint32_t num = 0, den = 0;
        for (int p = 0; p < num_partitions; ++p) {
            num += coeff_probs[p * 1056 + i];
            den += coeff_probs_denom[p * 1056 + i];
        }
        num = (num << 8) / den;
        coeff_probs[i] = num > 255 ? 255 : (num == 0 ? 1 : num);
    }
}

/* ---- boolean coder, :63-146 (RFC 6386 section 7.3) ---------------------------------------------------- */
typedef struct { uint8_t *out; uint32_t range, bottom; int bit_count; uint32_t count; } boolenc;

static void carry(uint8_t *q) {
    while (*--q == 255) *q = 0;
    ++*q;
}
static void put_bool(boolenc *e, int prob, int bit) {
    const uint32_t split = 1 + (((e->range - 1) * (uint32_t)prob) >> 8);
    if (bit) { e->bottom += split; e->range -= split; }
    else e->range = split;
    while (e->range < 128) {
        e->range <<= 1;
        if (e->bottom & 0x80000000u) carry(e->out);
        e->bottom <<= 1;
        if (--e->bit_count == 0) {
            *e->out++ = (uint8_t)(e->bottom >> 24);
            e->count++;
            e->bottom &= 0xffffffu;
            e->bit_count = 8;
        }
    }
}
static void flush(boolenc *e) {
    int c = e->bit_count;
    uint32_t v = e->bottom;
    if (v & (1u << (32 - c))) carry(e->out);
    v <<= c & 7;
    for (c >>= 3; c > 0; --c) v <<= 8;
    for (c = 0; c < 4; ++c) {
        *e->out++ = (uint8_t)(v >> 24);
        e->count++;
        v <<= 8;
    }
}

/* ---- encode_coefficients, :347-414 --------------------------------------------------------------------- */
void vp8o_encode_coefficients(const int16_t *MB, const int32_t *MB_non_zero_coeffs, const int32_t *MB_parts,
                              uint8_t *output, int32_t *partition_sizes, const uint8_t *third_context,
                              const uint32_t *coeff_probs, int mb_height, int mb_width, int num_partitions,
                              int partition_step) {
    for (int part = 0; part < num_partitions; ++part) {
        boolenc e = {output + (size_t)partition_step * part, 255, 0, 24, 0};
        for (int mb_row = part; mb_row < mb_height; mb_row += num_partitions)
            for (int mb_col = 0; mb_col < mb_width; ++mb_col) {
                const int mb = mb_row * mb_width + mb_col;
                if (MB_non_zero_coeffs[mb] == 0) continue;
                const int has_y2 = MB_parts[mb] == VP8O_16x16;
                for (int k = 0; k < 24 + has_y2; ++k) {
                    int ctx1;
                    const int b = block_order(has_y2, k, &ctx1);
                    const int16_t *c = MB + ((size_t)mb * 25 + b) * 16;
                    int ctx3 = third_context[mb * 25 + b];
                    const int first = ctx1 == 0 ? 1 : 0;
                    int tok[16];
                    const int end = tokenize(c, first, tok);
                    int after_zero = 0;
                    for (int i = first; i < end; ++i) {
                        const token_path *p = &k_path[tok[i]];
                        for (int s = after_zero; s < p->len; ++s)
                            put_bool(&e, (uint8_t)coeff_probs[pidx(0, ctx1, k_band[i], ctx3, p->node[s])], p->bit[s]);
                        if (tok[i] == T_EOB) break;
                        if (tok[i] >= T_CAT1) {
                            const int cat = tok[i] - T_CAT1;
                            const int extra = (c[i] < 0 ? -c[i] : c[i]) - k_cat_base[cat];
                            for (int j = 0; j < k_cat_bits[cat]; ++j)
                                put_bool(&e, k_cat_prob[cat][j], (extra >> (k_cat_bits[cat] - 1 - j)) & 1);
                        }
                        if (tok[i] != T_ZERO) put_bool(&e, 128, c[i] < 0);
                        after_zero = tok[i] == T_ZERO;
                        ctx3 = tok[i] == T_ZERO ? 0 : (tok[i] == T_ONE ? 1 : 2);
                    }
                }
            }
        flush(&e);
        partition_sizes[part] = (int32_t)e.count;
    }
}
