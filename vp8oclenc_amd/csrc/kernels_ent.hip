// kernels_ent.hip -- coefficient entropy stage on gfx950 (the reference runs it on a CPU OpenCL device, one
// work-item per partition: count_probs, num_div_denom, encode_coefficients, src/CPU_kernels.cl:347-778,
// orchestrated at src/vp8enc.cpp:48-94).
//
// Part 1 (this section): token statistics.  count_probs is a histogram: every 4x4 block is tokenised on its
// own -- the only coupling between blocks is the "third context" of its first token (how many of the blocks
// above / to the left carry a non-zero coefficient), which is a pure function of the coefficient buffer.
//   k_ent_flags   one thread per block: does it have a non-zero coefficient at all / past its DC?
//   k_ent_count   one thread per block of every coded macroblock: third context from the neighbours' flags,
//                 then the token walk of count_probs_in_block (including the reference's quirk of counting an
//                 EOB at every position after the first one, :478-534) into an LDS histogram per workgroup
//                 (one workgroup per macroblock row), written out as that row's histogram;
//   k_ent_probs   num_div_denom: sums the partitions, divides, clamps to 1..255.
#include "vp8hip_dev.h"

namespace vp8 {

namespace ent {

constexpr int NCTX = 4 * 8 * 3 * 11;   // [ctx1][ctx2][ctx3][tree node], :506

// token ids and their paths through the coefficient tree (src/CPU_kernels.cl:181-193): node = tree index / 2 is
// the probability slot, bit the branch taken.  Packed 4 bits per node, LSB first; every path starts at node 0.
enum { T_ZERO, T_ONE, T_TWO, T_THREE, T_FOUR, T_CAT1, T_CAT2, T_CAT3, T_CAT4, T_CAT5, T_CAT6, T_EOB };
__device__ __constant__ const uint32_t k_path_nodes[12] = {
    0x10, 0x210, 0x43210, 0x543210, 0x543210, 0x763210, 0x763210, 0x9863210, 0x9863210, 0xa863210, 0xa863210, 0x0};
__device__ __constant__ const uint8_t k_path_bits[12] = {   // branch bits, LSB = first node
    0x01, 0x03, 0x07, 0x17, 0x37, 0x0f, 0x2f, 0x1f, 0x5f, 0x3f, 0x7f, 0x00};
__device__ __constant__ const uint8_t k_path_len[12] = {2, 3, 5, 6, 6, 6, 6, 7, 7, 7, 7, 1};
__device__ __constant__ const uint8_t k_band[16] = {0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7};   // :200
// extra-bit categories (src/CPU_kernels.cl:194-199)
__device__ __constant__ const int k_cat_base[6] = {5, 7, 11, 19, 35, 67};
__device__ __constant__ const int k_cat_bits[6] = {1, 2, 3, 4, 5, 11};
__device__ __constant__ const uint8_t k_cat_prob[6][11] = {
    {159}, {165, 145}, {173, 148, 140}, {176, 155, 140, 135}, {180, 157, 141, 134, 130},
    {254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129}};

// The token tables in LDS, one 64-bit entry per token: path nodes (bits 0-27), branch bits (32-39), path length (40-43), extra
// bits (44-47) and base (48-54) of the categories.  As arrays in constant memory every lookup was a vector load from the cache
// hierarchy on the walk's serial chain -- k_fe_emit alone held 154 of them behind 205 waits; a workgroup copies them once.
struct WalkTab {
    unsigned long long tok[12];
    uint8_t catp[6][12];
};
__device__ __forceinline__ void walk_tab_fill(WalkTab &T) {   // by the first 84 threads of a workgroup; the caller synchronises
    const int t = threadIdx.x;
    if (t < 12) {
        unsigned long long e = (unsigned long long)k_path_nodes[t] | ((unsigned long long)k_path_bits[t] << 32) | ((unsigned long long)k_path_len[t] << 40);
        if (t >= T_CAT1 && t < T_EOB) e |= ((unsigned long long)k_cat_bits[t - T_CAT1] << 44) | ((unsigned long long)k_cat_base[t - T_CAT1] << 48);
        T.tok[t] = e;
    } else if (t < 12 + 72) {
        const int i = t - 12, c = i / 12, j = i % 12;
        T.catp[c][j] = j < 11 ? k_cat_prob[c][j] : 0;
    }
}

__device__ __forceinline__ int classify(int mag) {   // tokenize_block, :263-345
    return mag <= 4 ? mag : (mag <= 6 ? T_CAT1 : (mag <= 10 ? T_CAT2 : (mag <= 18 ? T_CAT3 : (mag <= 34 ? T_CAT4 : (mag <= 66 ? T_CAT5 : T_CAT6)))));
}

// block order inside a macroblock as coded (:371-403): [24 if 16x16], 0..15, 16..23; plane context ctx1
__device__ __forceinline__ int plane_ctx(int b, bool has_y2) { return b == 24 ? 1 : (b < 16 ? (has_y2 ? 0 : 3) : 2); }

// (the *_body functions take the workgroup index as an argument: the frame path runs several of them in one launch,
// kernels_entropy_stage.hip)
__device__ __forceinline__ void flags_body(int vb, const int16_t *coeffs, uint8_t *flags, int nblocks) {
    const int i = vb * 256 + threadIdx.x;
    if (i >= nblocks) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(coeffs + (size_t)i * 16);
    const uint4 a = p[0], b = p[1];
    const uint32_t rest = (a.x & 0xffff0000u) | a.y | a.z | a.w | b.x | b.y | b.z | b.w;
    flags[i] = (uint8_t)(((a.x & 0xffffu) | rest ? 1 : 0) | (rest ? 2 : 0));   // bit0: any coefficient, bit1: any past the first
}
__global__ __launch_bounds__(256) void k_ent_flags(const int16_t *coeffs, uint8_t *flags, int nblocks) { flags_body(blockIdx.x, coeffs, flags, nblocks); }

// third context of block b of macroblock mb (count_probs, :560-760)
__device__ __forceinline__ int third_context(const uint8_t *flags, const int32_t *parts, int mb, int b, int mb_row, int mb_col,
                                             int mbw) {
    int ctx = 0;
    if (b == 24) {   // nearest macroblock above / to the left in the row that has a Y2 block, :575-604
        if (mb_row > 0) {
            int p = mb - mbw;
            while (p >= 0 && parts[p] != 0) p -= mbw;
            if (p >= 0) ctx += flags[p * 25 + 24] & 1;
        }
        if (mb_col > 0) {
            int p = mb - 1;
            while (p >= mb_row * mbw && parts[p] != 0) --p;
            if (p >= mb_row * mbw) ctx += flags[p * 25 + 24] & 1;
        }
        return ctx;
    }
    const int w = b < 16 ? 4 : 2, base = b < 16 ? 0 : (b < 20 ? 16 : 20);
    const int bx = (b - base) % w, by = (b - base) / w;
    int nmb = -1, nb = 0;
    if (by > 0) { nmb = mb; nb = b - w; }
    else if (mb_row > 0) { nmb = mb - mbw; nb = b + w * (w - 1); }
    if (nmb >= 0) {
        const int f = flags[nmb * 25 + nb];
        ctx += (b < 16 && parts[nmb] == 0) ? (f >> 1) & 1 : f & 1;   // a 16x16 neighbour's luma DC slot is not a coefficient
    }
    nmb = -1;
    if (bx > 0) { nmb = mb; nb = b - 1; }
    else if (mb_col > 0) { nmb = mb - 1; nb = b + (w - 1); }
    if (nmb >= 0) {
        const int f = flags[nmb * 25 + nb];
        ctx += (b < 16 && parts[nmb] == 0) ? (f >> 1) & 1 : f & 1;
    }
    return ctx;
}

// counts[mb_row * CNT_SPLIT + q][NCTX][2] = {zero branches taken, branches seen} of one quarter of a macroblock
// row (the reference's num and denom - 1 are the sums over the rows of a partition).  Histogram in LDS, written
// out with plain stores and summed by k_ent_probs: no atomics in HBM.
constexpr int CNT_SPLIT = 4;   // workgroups per macroblock row
struct CountItem {
    const int16_t *coeffs;
    const int32_t *nzc, *parts;
    const uint8_t *flags;
    uint8_t *third_ctx;
    uint32_t *counts;
    int mbw;
};
__device__ __forceinline__ void count_body(int mb_row, int quarter, const int16_t *coeffs, const int32_t *nzc, const int32_t *parts,
                                           const uint8_t *flags, uint8_t *third_ctx, uint32_t *counts, int mbw) {
    __shared__ uint32_t s_h[NCTX * 2];
    __shared__ WalkTab s_tab;
    walk_tab_fill(s_tab);
    for (int i = threadIdx.x; i < NCTX * 2; i += 256) s_h[i] = 0;
    __syncthreads();
    const int per = (mbw * 25 + CNT_SPLIT - 1) / CNT_SPLIT, i0 = quarter * per;
    const int i1 = i0 + per < mbw * 25 ? i0 + per : mbw * 25;
    for (int item = i0 + threadIdx.x; item < i1; item += 256) {
        const int mb_col = item / 25, b = item % 25;
        const int mb = mb_row * mbw + mb_col;
        const bool has_y2 = parts[mb] == 0;
        if (nzc[mb] != 0 && (b < 24 || has_y2)) {
            const int ctx1 = plane_ctx(b, has_y2);
            int ctx3 = third_context(flags, parts, mb, b, mb_row, mb_col, mbw);
            third_ctx[mb * 25 + b] = (uint8_t)ctx3;
            const uint4 *p = reinterpret_cast<const uint4 *>(coeffs + ((size_t)mb * 25 + b) * 16);
            const uint4 q0 = p[0], q1 = p[1];
            const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const int first = ctx1 == 0 ? 1 : 0;
            int last = -1;   // last non-zero position >= first
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = (int16_t)(w[i >> 1] >> (16 * (i & 1)));
                if (c != 0 && i >= first) last = i;
            }
            bool after_zero = false;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i < first) continue;
                const int c = (int16_t)(w[i >> 1] >> (16 * (i & 1)));
                const int t = i > last ? T_EOB : classify(c < 0 ? -c : c);   // every position past the end is an EOB, counted too
                const unsigned long long e = s_tab.tok[t];
                const uint32_t nodes = (uint32_t)e;
                const int bits = (int)(e >> 32) & 255, len = (int)(e >> 40) & 15;
                const int base = ((ctx1 * 8 + k_band[i]) * 3 + ctx3) * 11;
                for (int s = after_zero ? 1 : 0; s < len; ++s) {   // after a ZERO the first branch is implied
                    const int at = (base + ((nodes >> (4 * s)) & 15)) * 2;
                    if (!((bits >> s) & 1)) atomicAdd(&s_h[at], 1u);
                    atomicAdd(&s_h[at + 1], 1u);
                }
                after_zero = t == T_ZERO;
                ctx3 = t == T_ZERO ? 0 : (t == T_ONE ? 1 : 2);
            }
        }
    }
    __syncthreads();
    uint32_t *dst = counts + (size_t)(mb_row * CNT_SPLIT + quarter) * NCTX * 2;
    for (int i = threadIdx.x; i < NCTX * 2; i += 256) dst[i] = s_h[i];
}
__global__ __launch_bounds__(256) void k_ent_count(const int16_t *coeffs, const int32_t *nzc, const int32_t *parts,
                                                   const uint8_t *flags, uint8_t *third_ctx, uint32_t *counts, int mbw) {
    count_body(blockIdx.x, blockIdx.y, coeffs, nzc, parts, flags, third_ctx, counts, mbw);
}
__global__ __launch_bounds__(256) void k_ent_count_b(BatchOf<CountItem> b) {   // blockIdx.z = member of the batch
    const CountItem &a = b.item[blockIdx.z];
    count_body(blockIdx.x, blockIdx.y, a.coeffs, a.nzc, a.parts, a.flags, a.third_ctx, a.counts, a.mbw);
}

// num_div_denom (:764-778) + the denominators of partition 0 that the host inspects (vp8enc.cpp:69-76):
// every partition's denominator starts at 1 (:552)
// 16 contexts per workgroup, 16 lanes per context over the partial histograms
// defaults (may be null): the format's default probabilities, taken for contexts that never occurred (vp8enc.cpp:69-76)
__device__ __forceinline__ void probs_body(int vb, const uint32_t *counts, uint32_t *probs, uint32_t *denom0, int mbh,
                                           int num_partitions, const uint8_t *defaults) {
    const int lane = threadIdx.x & 15, i = vb * 16 + (threadIdx.x >> 4);   // NCTX is a multiple of 16
    uint32_t num = 0, den = 0, den0 = 0;
    for (int h = lane; h < mbh * CNT_SPLIT; h += 16) {
        const uint2 c = *reinterpret_cast<const uint2 *>(counts + ((size_t)h * NCTX + i) * 2);
        num += c.x;
        den += c.y;
        if ((h / CNT_SPLIT) % num_partitions == 0) den0 += c.y;
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
        num += (uint32_t)__shfl_xor((int)num, m, 16);
        den += (uint32_t)__shfl_xor((int)den, m, 16);
        den0 += (uint32_t)__shfl_xor((int)den0, m, 16);
    }
    if (lane == 0) {
        den += (uint32_t)num_partitions;   // every partition's denominator starts at 1 (:552)
        num = (num << 8) / den;
        probs[i] = (defaults && den0 + 1u < 2u) ? (uint32_t)defaults[i] : (num > 255u ? 255u : (num == 0u ? 1u : num));
        denom0[i] = den0 + 1u;
    }
}
__global__ __launch_bounds__(256) void k_ent_probs(const uint32_t *counts, uint32_t *probs, uint32_t *denom0, int mbh,
                                                   int num_partitions, const uint8_t *defaults) {
    probs_body(blockIdx.x, counts, probs, denom0, mbh, num_partitions, defaults);
}

}  // namespace ent

void launch_ent_count(hipStream_t s, const MBOut &o, uint8_t *flags, uint8_t *third_ctx, uint32_t *counts, uint32_t *probs,
                      uint32_t *denom0, int mbw, int mbh, int num_partitions, const uint8_t *defaults) {
    const int nblocks = mbw * mbh * 25;
    hipLaunchKernelGGL(ent::k_ent_flags, dim3((nblocks + 255) / 256), dim3(256), 0, s, o.coeffs, flags, nblocks);
    hipLaunchKernelGGL(ent::k_ent_count, dim3(mbh, ent::CNT_SPLIT), dim3(256), 0, s, o.coeffs, o.nz, o.parts, flags, third_ctx,
                       counts, mbw);
    hipLaunchKernelGGL(ent::k_ent_probs, dim3(ent::NCTX / 16), dim3(256), 0, s, counts, probs, denom0, mbh, num_partitions, defaults);
}

// ====================================================================================================
// Part 2: encode_coefficients (src/CPU_kernels.cl:347-414) -- the boolean coder, in parallel.
//
// The reference codes one partition per work-item, serially.  The coder's state after each bool is
// (range, bottom, bit_count): `range` (128..255 after renormalisation) decides every split, `bottom` only
// accumulates.  Written as exact arithmetic the output of a partition is the big number
//        sum_i  add_i * 2^(8*nbytes - 8 - W_i)        (add_i = split if bool i is 1, else 0; W_i = shifts before i)
// in nbytes = emitted + 4 bytes (the reference's flush writes the last 32 bits), carries included.  So:
//   1. k_ent_boolcount / scan / k_ent_emit: every block turns its tokens into (probability, bit) pairs at its
//      offset in the partition's bool string (block order = coding order);
//   2. k_ent_maps: the string is cut into chunks of CHUNK bools; 128 lanes run a chunk from each of the 128
//      possible start ranges and record (end range, shifts) -- the chunk as a function of its start state;
//   3. k_ent_walk: one lane per partition composes those functions in order (one LDS lookup per chunk) and so
//      learns every chunk's true start range and bit position;
//   4. k_ent_encode: one lane per chunk replays it from its true state and adds each split at its bit position
//      into 64-bit accumulators, one per 32 bits of output (neighbouring chunks meet in atomics);
//   5. k_ent_finish: one wave per partition resolves the carries (carry-lookahead over 64 words per step) and
//      writes the bytes.
// Every step is checked against the serial reference coder through the byte-exact partitions.
// ====================================================================================================
namespace ent {

constexpr int CHUNK = ENT_CHUNK;  // bools per chunk
constexpr int SCAN_TILE = 1024;   // slots per scan tile (256 threads x 4)



struct Geom {
    int mbw, mbh, P;
    uint32_t slot_base[ENT_MAX_PARTITIONS + 1];   // first slot of each partition (25 slots per macroblock, coding order)
    uint32_t cap_bools, cap_chunks, cap_words;
};
using Plan = EntPlan;   // written on the device, read back by the host after the launch sequence (vp8hip_dev.h)

// slot of (macroblock, k): k = 0 is block 24, 1..16 are Y 0..15, 17..24 are blocks 16..23
__device__ __forceinline__ uint32_t slot_of(const Geom &g, int mb_row, int mb_col, int k) {
    return g.slot_base[mb_row % g.P] + (uint32_t)(((mb_row / g.P) * g.mbw + mb_col) * 25 + k);
}
__device__ __forceinline__ int block_of_k(int k) { return k == 0 ? 24 : k - 1; }

// the bools of one block in coding order (encode_block, :202-261): ctx(index into coeff_probs, bit) for tree
// branches, lit(probability, bit) for extra bits and signs
struct ConstTab {   // the step-by-step kernels: straight from constant memory
    __device__ __forceinline__ unsigned long long tok(int t) const {
        unsigned long long e = (unsigned long long)k_path_nodes[t] | ((unsigned long long)k_path_bits[t] << 32) | ((unsigned long long)k_path_len[t] << 40);
        if (t >= T_CAT1 && t < T_EOB) e |= ((unsigned long long)k_cat_bits[t - T_CAT1] << 44) | ((unsigned long long)k_cat_base[t - T_CAT1] << 48);
        return e;
    }
    __device__ __forceinline__ int catp(int c, int j) const { return k_cat_prob[c][j]; }
};
struct LdsTab {     // the frame path: the workgroup's copy
    const WalkTab &T;
    __device__ __forceinline__ unsigned long long tok(int t) const { return T.tok[t]; }
    __device__ __forceinline__ int catp(int c, int j) const { return T.catp[c][j]; }
};
template <class Sink, class Tab = ConstTab>
__device__ __forceinline__ void walk_block(const int16_t *blk, int ctx1, int ctx3, Sink &sink, const Tab tab = Tab()) {
    const uint4 *p = reinterpret_cast<const uint4 *>(blk);
    const uint4 q0 = p[0], q1 = p[1];
    const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    const int first = ctx1 == 0 ? 1 : 0;
    int last = -1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = (int16_t)(w[i >> 1] >> (16 * (i & 1)));
        if (c != 0 && i >= first) last = i;
    }
    bool after_zero = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {   // unrolled: w[] stays in registers
        if (i < first) continue;
        const int c = (int16_t)(w[i >> 1] >> (16 * (i & 1)));
        const int mag = c < 0 ? -c : c;
        const int t = i > last ? T_EOB : classify(mag);
        const unsigned long long e = tab.tok(t);
        const uint32_t nodes = (uint32_t)e;
        const int bits = (int)(e >> 32) & 255, len = (int)(e >> 40) & 15;
        const int base = ((ctx1 * 8 + k_band[i]) * 3 + ctx3) * 11;
        for (int s = after_zero ? 1 : 0; s < len; ++s) sink.ctx(base + ((nodes >> (4 * s)) & 15), (bits >> s) & 1);
        if (t == T_EOB) break;
        if (t >= T_CAT1) {
            const int cat = t - T_CAT1, nb = (int)(e >> 44) & 15, extra = mag - ((int)(e >> 48) & 127);
            for (int j = 0; j < nb; ++j) sink.lit(tab.catp(cat, j), (extra >> (nb - 1 - j)) & 1);
        }
        if (t != T_ZERO) sink.lit(128, c < 0);
        after_zero = t == T_ZERO;
        ctx3 = t == T_ZERO ? 0 : (t == T_ONE ? 1 : 2);
    }
}

struct CountSink {
    uint32_t n = 0;
    __device__ __forceinline__ void ctx(int, int) { ++n; }
    __device__ __forceinline__ void lit(int, int) { ++n; }
};
struct EmitSink {
    uint16_t *out;
    const uint32_t *probs;
    __device__ __forceinline__ void ctx(int idx, int bit) { *out++ = (uint16_t)((probs[idx] & 255u) | (bit << 8)); }
    __device__ __forceinline__ void lit(int prob, int bit) { *out++ = (uint16_t)(prob | (bit << 8)); }
};
struct EmitSinkLds {   // the frame path: the 1 056 probabilities as bytes in LDS
    uint16_t *out;
    const uint8_t *probs8;
    __device__ __forceinline__ void ctx(int idx, int bit) { *out++ = (uint16_t)(probs8[idx] | (bit << 8)); }
    __device__ __forceinline__ void lit(int prob, int bit) { *out++ = (uint16_t)(prob | (bit << 8)); }
};

// is slot k of macroblock mb coded at all?
__device__ __forceinline__ bool slot_live(const int32_t *nzc, const int32_t *parts, int mb, int k, bool &has_y2) {
    has_y2 = parts[mb] == 0;
    return nzc[mb] != 0 && (k > 0 || has_y2);
}

__global__ __launch_bounds__(256) void k_ent_boolcount(const int16_t *coeffs, const int32_t *nzc, const int32_t *parts, Geom g,
                                                       uint32_t *cnt) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.mbw * g.mbh * 25) return;
    const int mb = t / 25, k = t % 25, b = block_of_k(k);
    bool has_y2;
    CountSink sink;
    if (slot_live(nzc, parts, mb, k, has_y2)) walk_block(coeffs + ((size_t)mb * 25 + b) * 16, plane_ctx(b, has_y2), 0, sink);
    cnt[slot_of(g, mb / g.mbw, mb % g.mbw, k)] = sink.n;
}

// ---- exclusive scan of cnt[0..n) in place, total in cnt[n]: tile sums, scan of the sums, per-tile scan -------
__global__ __launch_bounds__(256) void k_scan_tiles(const uint32_t *v, uint32_t *tile_sum, int n) {
    __shared__ uint32_t s[256];
    const int i0 = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
    uint32_t a = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) a += i0 + j < n ? v[i0 + j] : 0u;
    s[threadIdx.x] = a;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) s[threadIdx.x] += s[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = s[0];
}
__global__ __launch_bounds__(1024) void k_scan_top(uint32_t *tile_sum, int ntiles) {   // exclusive, total at [ntiles]; any ntiles
    __shared__ uint32_t s[1024];
    const int t = threadIdx.x, per = (ntiles + 1023) / 1024, i0 = t * per;   // a thread owns `per` consecutive tile sums
    uint32_t own = 0;
    for (int j = 0; j < per; ++j) own += i0 + j < ntiles ? tile_sum[i0 + j] : 0u;
    s[t] = own;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t add = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] += add;
        __syncthreads();
    }
    uint32_t run = s[t] - own;
    for (int j = 0; j < per && i0 + j < ntiles; ++j) {
        const uint32_t x = tile_sum[i0 + j];
        tile_sum[i0 + j] = run;
        run += x;
    }
    if (t == 1023) tile_sum[ntiles] = s[t];
}
__global__ __launch_bounds__(256) void k_scan_apply(uint32_t *v, const uint32_t *tile_sum, int n, int ntiles) {
    __shared__ uint32_t s[256];
    const int t = threadIdx.x, i0 = blockIdx.x * SCAN_TILE + t * 4;
    uint32_t x[4], a = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = i0 + j < n ? v[i0 + j] : 0u; a += x[j]; }
    s[t] = a;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t add = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] += add;
        __syncthreads();
    }
    uint32_t run = tile_sum[blockIdx.x] + s[t] - a;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i0 + j < n) v[i0 + j] = run;
        run += x[j];
    }
    if (blockIdx.x == 0 && t == 0) v[n] = tile_sum[ntiles];
}

// per-partition layout of the bool string, its chunks and its output words
__global__ void k_ent_plan(const uint32_t *offs, Geom g, Plan *plan) {
    if (threadIdx.x != 0) return;
    uint32_t cb = 0, wb = 0;
    for (int p = 0; p < g.P; ++p) {
        const uint32_t b0 = offs[g.slot_base[p]], b1 = offs[g.slot_base[p + 1]];
        plan->bool_base[p] = b0;
        plan->nbools[p] = b1 - b0;
        plan->chunk_base[p] = cb;
        plan->word_base[p] = wb;
        cb += (b1 - b0 + CHUNK - 1) / CHUNK;
        wb += ((b1 - b0) * 7 + 31) / 32 + 4;      // a bool shifts at most 7 bits out; + the flush
    }
    plan->bool_base[g.P] = offs[g.slot_base[g.P]];
    plan->chunk_base[g.P] = cb;
    plan->word_base[g.P] = wb;
    plan->total_chunks = cb;
    plan->overflow = (plan->bool_base[g.P] > g.cap_bools || cb > g.cap_chunks || wb > g.cap_words) ? 1u : 0u;
    if (plan->overflow) {   // the later kernels then see empty partitions and touch nothing outside the scratch
        plan->total_chunks = 0;
        for (int p = 0; p <= g.P; ++p) plan->bool_base[p] = plan->chunk_base[p] = plan->word_base[p] = 0;
        for (int p = 0; p < g.P; ++p) plan->nbools[p] = 0;
    }
}

__global__ __launch_bounds__(256) void k_ent_emit(const int16_t *coeffs, const int32_t *nzc, const int32_t *parts,
                                                  const uint8_t *third_ctx, const uint32_t *probs, const uint32_t *offs,
                                                  const Plan *plan, Geom g, uint16_t *bools, unsigned long long *acc) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (uint32_t i = (uint32_t)t, n = plan->word_base[g.P]; i < n; i += gridDim.x * 256) acc[i] = 0ull;   // the coder's accumulators (was a launch of its own)
    if (t >= g.mbw * g.mbh * 25 || plan->overflow) return;
    const int mb = t / 25, k = t % 25, b = block_of_k(k);
    bool has_y2;
    if (!slot_live(nzc, parts, mb, k, has_y2)) return;
    EmitSink sink{bools + offs[slot_of(g, mb / g.mbw, mb % g.mbw, k)], probs};
    walk_block(coeffs + ((size_t)mb * 25 + b) * 16, plane_ctx(b, has_y2), third_ctx[mb * 25 + b], sink);
}

// ---- the same three steps for the frame path, in SLOT order (slot = position in coding order through the partitions), so
// that the prefix sums need no pass of their own: counts + per-workgroup sums, one workgroup scans the sums and lays
// out the plan, the emit kernel scans its own 256 counts -----------------------------------------------------------
__device__ __forceinline__ bool slot_to_block(const Geom &g, uint32_t s, int &mb, int &k) {
    if (s >= g.slot_base[g.P]) return false;
    int p = 0;
    while (p + 1 < g.P && s >= g.slot_base[p + 1]) ++p;
    const uint32_t local = s - g.slot_base[p], per_row = (uint32_t)g.mbw * 25u;
    const uint32_t r = local / per_row, rem = local % per_row;
    mb = (int)((r * (uint32_t)g.P + (uint32_t)p) * (uint32_t)g.mbw + rem / 25u);
    k = (int)(rem % 25u);
    return true;
}
// exclusive scan over the 256 threads of a workgroup (s_w: 4 words of LDS); total = sum over the workgroup
__device__ __forceinline__ uint32_t wg_scan256(uint32_t v, uint32_t *s_w, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();   // s_w may still be read from an earlier call
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += s_w[w];
    total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return base + x - v;
}
__device__ __forceinline__ void boolcount_slots_body(int vb, const int16_t *coeffs, const int32_t *nzc, const int32_t *parts, const Geom &g,
                                                     uint32_t *cnt, uint32_t *tile_sum) {
    __shared__ uint32_t s_w[4];
    __shared__ WalkTab s_tab;
    walk_tab_fill(s_tab);
    __syncthreads();
    const uint32_t s = (uint32_t)vb * 256u + threadIdx.x;
    int mb, k;
    CountSink sink;
    if (slot_to_block(g, s, mb, k)) {
        bool has_y2;
        const int b = block_of_k(k);
        if (slot_live(nzc, parts, mb, k, has_y2)) walk_block(coeffs + ((size_t)mb * 25 + b) * 16, plane_ctx(b, has_y2), 0, sink, LdsTab{s_tab});
        cnt[s] = sink.n;
    }
    uint32_t total;
    wg_scan256(sink.n, s_w, total);
    if (threadIdx.x == 0) tile_sum[vb] = total;
}
// one workgroup: exclusive scan of the ntiles workgroup sums in place, then the plan
__device__ __forceinline__ void scan_plan_body(const uint32_t *cnt, uint32_t *tile_sum, int ntiles, const Geom &g, Plan *plan) {
    __shared__ uint32_t s_w[4], s_off[ENT_MAX_PARTITIONS + 1];
    const int t = threadIdx.x, per = (ntiles + 255) / 256, i0 = t * per;
    uint32_t a = 0;
    for (int j = 0; j < per; ++j) a += i0 + j < ntiles ? tile_sum[i0 + j] : 0u;
    uint32_t total;
    uint32_t run = wg_scan256(a, s_w, total);
    for (int j = 0; j < per && i0 + j < ntiles; ++j) {
        const uint32_t x = tile_sum[i0 + j];
        tile_sum[i0 + j] = run;
        run += x;
    }
    __syncthreads();
    if (t <= g.P) {   // bools before the first slot of partition t
        const uint32_t sb = g.slot_base[t], tile = sb >> 8;
        uint32_t o = total;
        if ((int)tile < ntiles) {
            o = tile_sum[tile];
            for (uint32_t i = tile << 8; i < sb; ++i) o += cnt[i];
        }
        s_off[t] = o;
    }
    __syncthreads();
    if (t != 0) return;
    uint32_t cb = 0, wb = 0;
    for (int p = 0; p < g.P; ++p) {
        const uint32_t b0 = s_off[p], b1 = s_off[p + 1];
        plan->bool_base[p] = b0;
        plan->nbools[p] = b1 - b0;
        plan->chunk_base[p] = cb;
        plan->word_base[p] = wb;
        cb += (b1 - b0 + CHUNK - 1) / CHUNK;
        wb += ((b1 - b0) * 7 + 31) / 32 + 4;      // a bool shifts at most 7 bits out; + the flush
    }
    plan->bool_base[g.P] = s_off[g.P];
    plan->chunk_base[g.P] = cb;
    plan->word_base[g.P] = wb;
    plan->total_chunks = cb;
    plan->overflow = (s_off[g.P] > g.cap_bools || cb > g.cap_chunks || wb > g.cap_words) ? 1u : 0u;
    if (plan->overflow) {   // the later kernels then see empty partitions and touch nothing outside the scratch
        plan->total_chunks = 0;
        for (int p = 0; p <= g.P; ++p) plan->bool_base[p] = plan->chunk_base[p] = plan->word_base[p] = 0;
        for (int p = 0; p < g.P; ++p) plan->nbools[p] = 0;
    }
}
__device__ __forceinline__ void emit_slots_body(int vb, int nvb, const int16_t *coeffs, const int32_t *nzc, const int32_t *parts,
                                                const uint8_t *third_ctx, const uint32_t *probs, const uint32_t *cnt,
                                                const uint32_t *tile_pre, const Plan *plan, const Geom &g, uint16_t *bools,
                                                unsigned long long *acc) {
    __shared__ uint32_t s_w[4];
    __shared__ WalkTab s_tab;
    __shared__ uint8_t s_probs[NCTX];
    for (uint32_t i = (uint32_t)vb * 256u + threadIdx.x, n = plan->word_base[g.P]; i < n; i += (uint32_t)nvb * 256u) acc[i] = 0ull;   // the coder's accumulators
    if (plan->overflow) return;
    walk_tab_fill(s_tab);
    for (int i = threadIdx.x; i < NCTX; i += 256) s_probs[i] = (uint8_t)probs[i];   // (read below behind wg_scan256's barriers)
    const uint32_t s = (uint32_t)vb * 256u + threadIdx.x;
    int mb = 0, k = 0;
    const bool valid = slot_to_block(g, s, mb, k);
    const uint32_t n = valid ? cnt[s] : 0u;
    uint32_t total;
    const uint32_t off = tile_pre[vb] + wg_scan256(n, s_w, total);
    if (!n) return;   // dead slot (a live one has at least its end-of-block)
    const int b = block_of_k(k);
    const bool has_y2 = parts[mb] == 0;
    EmitSinkLds sink{bools + off, s_probs};
    walk_block(coeffs + ((size_t)mb * 25 + b) * 16, plane_ctx(b, has_y2), third_ctx[mb * 25 + b], sink, LdsTab{s_tab});
}

// chunk -> its partition and its slice of the bool string
__device__ __forceinline__ void chunk_slice(const Plan *plan, int P, uint32_t chunk, int &p, uint32_t &b0, int &n) {
    p = 0;
    while (p + 1 < P && chunk >= plan->chunk_base[p + 1]) ++p;
    const uint32_t local = chunk - plan->chunk_base[p];
    b0 = plan->bool_base[p] + local * CHUNK;
    const uint32_t end = plan->bool_base[p] + plan->nbools[p];
    n = (int)(end - b0 < (uint32_t)CHUNK ? end - b0 : (uint32_t)CHUNK);
}

// one step of the coder's range recursion (write_bool, :82-105): returns the shift count
__device__ __forceinline__ int range_step(uint32_t &r, uint32_t prob, uint32_t bit, uint32_t &split) {
    // 1 + ((r - 1) * prob >> 8) = (r * prob + 256 - prob) >> 8: one 24-bit multiply-add (both factors below 256) and a shift
    split = (__umul24(r, prob) + (256u - prob)) >> 8;
    r = bit ? r - split : split;
    const int z = __builtin_clz(r);   // r >= 1: split >= 1 and split < r
    r = (r << z) >> 24;               // renormalised to 128..255
    return z - 24;
}
// the same with the leading-zero count handed back raw: a caller that only sums shifts subtracts 24 per bool at the end
__device__ __forceinline__ int range_step_z(uint32_t &r, uint32_t prob, uint32_t bias /* 256 - prob */, uint32_t bit) {
    const uint32_t split = (__umul24(r, prob) + bias) >> 8;
    r = bit ? r - split : split;
    const int z = __builtin_clz(r);
    r = (r << z) >> 24;
    return z;
}

// One or two coder jobs per launch (the coefficient partitions and the first partition of a frame go through the coder
// together: blockIdx.y picks the job).
struct CodeJob {
    const uint16_t *bools;
    Plan *plan;
    uint32_t *maps, *smaps;   // per chunk / per super-chunk: [128 start ranges] -> end range | shifts << 8
    uint2 *start;
    unsigned long long *acc;
    uint8_t *bytes;
    int32_t *sizes;
    int P;
};
// Frames of a batch (vp8hip_batch_encode_frame_begin) are job pairs 2m, 2m + 1 of the same launches.
struct CodeJobs {
    CodeJob j[2 * MAX_BATCH];
    uint8_t *frame[MAX_BATCH];                       // [m] not null: j[2m] = coefficient partitions, j[2m + 1] = first partition, and
    uint32_t head[MAX_BATCH], capacity[MAX_BATCH];   // k_ent_finish lays the finished frame out at frame[m] + 16 (gather_frame, src/encIO.h:1-30)
};

constexpr int SUP = 8;   // chunks per super-chunk (never across a partition boundary)

__device__ __forceinline__ uint32_t supers_of(const Plan *plan, int p) {
    return (plan->chunk_base[p + 1] - plan->chunk_base[p] + SUP - 1) / SUP;
}
// super-chunk (counted through all partitions of the job) -> partition, first chunk, number of chunks
__device__ __forceinline__ bool super_slice(const Plan *plan, int P, uint32_t sup, int &p, uint32_t &c0, int &nc) {
    uint32_t sb = 0;
    for (p = 0; p < P; ++p) {
        const uint32_t nch = plan->chunk_base[p + 1] - plan->chunk_base[p], ns = (nch + SUP - 1) / SUP;
        if (sup < sb + ns) {
            const uint32_t l = (sup - sb) * SUP;
            c0 = plan->chunk_base[p] + l;
            nc = (int)(nch - l < (uint32_t)SUP ? nch - l : (uint32_t)SUP);
            return true;
        }
        sb += ns;
    }
    return false;
}

// A workgroup takes a super-chunk.  Running a chunk from each of the 128 possible start ranges was three fifths of the whole
// entropy stage's instructions -- and most of it redundant: ranges that start apart fall together (a bool that comes out the
// unlikely way leaves a handful of values), on coded frames 16 distinct ranges are left after 32 bools, 11 after 64, but hardly
// ever ONE before the chunk ends.  So:
//   1. one wave per chunk runs the first MAPS_HEAD bools from all 128 start ranges (two chains per lane, the bools wave-uniform in
//      scalar registers: seven vector instructions per bool and chain);
//   2. the ranges reached are counted and numbered (a 128-bit mark per chunk, ranks by popcount);
//   3. the REST of every chunk is run once per distinct range, one (chunk, range) task per lane, the tasks of the eight chunks
//      packed into as few waves as they need (typically two or three instead of eight; the bools per lane from LDS);
//   4. a start range's entry = its head's shifts + the tail of the range its head reached; then 128 lanes compose the eight chunk
//      maps into the super-chunk's map, which is what the serial walk steps through.
// 3 584 -> about 1 300 vector instructions per chunk (the kernel alone on a 1080p frame: 45.8 -> 19.1 us, 3.4 M wave instructions);
// the maps are the same numbers (byte-exact partitions, tests/test_gpu_entropy.py).
constexpr int MAPS_THREADS = SUP * 64;
constexpr int MAPS_HEAD = 32;   // (frames out on one box, scripts/ab_build_bitstream.sh: 8: 51.4-52.1, 16 / 32 / 64: 52.0-52.8, 256 = no tail: 50.6-50.8 M MB/s)
__global__ __launch_bounds__(MAPS_THREADS) void k_ent_maps(CodeJobs jobs) {
    const CodeJob &J = jobs.j[blockIdx.y];
    __shared__ __attribute__((aligned(16))) uint16_t s_b[SUP][CHUNK];
    __shared__ uint32_t s_m[SUP][128];
    __shared__ uint32_t s_tail[SUP][128];    // [chunk][rank of the range after the head] -> end range | tail shifts << 8
    __shared__ uint8_t s_list[SUP][128];     // [chunk][rank] -> that range
    __shared__ unsigned long long s_mark[SUP][2];
    __shared__ int s_d[SUP + 1], s_n[SUP];   // distinct ranges per chunk (prefix sums), bools per chunk
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    __shared__ Plan s_plan;   // consulted several times per super-chunk: from LDS, not through chains of dependent global loads
    for (int i = tid; i < (int)(sizeof(Plan) / 4); i += MAPS_THREADS) reinterpret_cast<uint32_t *>(&s_plan)[i] = reinterpret_cast<const uint32_t *>(J.plan)[i];
    __syncthreads();
    const Plan *plan = &s_plan;
    for (uint32_t sup = blockIdx.x;; sup += gridDim.x) {
        int p, nc;
        uint32_t c0;
        if (!super_slice(plan, J.P, sup, p, c0, nc)) break;
        int n = 0;
        __syncthreads();
        if (w < nc) {
            const uint32_t b0 = plan->bool_base[p] + (c0 + w - plan->chunk_base[p]) * CHUNK, end = plan->bool_base[p] + plan->nbools[p];
            n = (int)(end - b0 < (uint32_t)CHUNK ? end - b0 : (uint32_t)CHUNK);
            for (int i = l; i < n; i += 64) s_b[w][i] = J.bools[b0 + i];
        }
        if (l == 0) s_n[w] = n;
        __syncthreads();
        // ---- 1: the head from every start range -------------------------------------------------------------------------
        uint32_t r0 = 128u + l, r1 = 192u + l, Z0 = 0, Z1 = 0;
        const int head = n < MAPS_HEAD ? n : MAPS_HEAD;
        if (w < nc) {
            // every lane steps through the SAME bools: probability, its bias and the bit are wave-uniform and live in scalar registers
            const uint4 *row = reinterpret_cast<const uint4 *>(&s_b[w][0]);   // eight bools per LDS read
            int i = 0;
            for (; i + 8 <= head; i += 8) {
                const uint4 q = row[i >> 3];
                const uint32_t e8[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)q.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)q.y),
                                        (uint32_t)__builtin_amdgcn_readfirstlane((int)q.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)q.w)};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t e = (e8[k >> 1] >> (16 * (k & 1))) & 0xffffu, pr = e & 255u, bias = 256u - pr, bit = e >> 8;
                    Z0 += range_step_z(r0, pr, bias, bit);
                    Z1 += range_step_z(r1, pr, bias, bit);
                }
            }
            for (; i < head; ++i) {
                const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_b[w][i]), pr = e & 255u, bias = 256u - pr, bit = e >> 8;
                Z0 += range_step_z(r0, pr, bias, bit);
                Z1 += range_step_z(r1, pr, bias, bit);
            }
        }
        // ---- 2: the ranges the heads reached, numbered ---------------------------------------------------------------------
        if (l < 2) s_mark[w][l] = 0ull;
        __syncthreads();
        if (w < nc) {
            atomicOr(&s_mark[w][(r0 - 128u) >> 6], 1ull << ((r0 - 128u) & 63u));
            atomicOr(&s_mark[w][(r1 - 128u) >> 6], 1ull << ((r1 - 128u) & 63u));
        }
        __syncthreads();
        const unsigned long long m0 = s_mark[w][0], m1 = s_mark[w][1];
        const int d0 = __popcll(m0), dw = w < nc ? d0 + __popcll(m1) : 0;
        if (w < nc) {   // lane l owns the values 128 + l and 192 + l: if reached, they go into the list at their rank
            const unsigned long long below = (1ull << l) - 1ull;
            if ((m0 >> l) & 1ull) s_list[w][__popcll(m0 & below)] = (uint8_t)(128 + l);
            if ((m1 >> l) & 1ull) s_list[w][d0 + __popcll(m1 & below)] = (uint8_t)(192 + l);
        }
        if (l == 0) s_d[w + 1] = dw;
        __syncthreads();
        if (tid == 0) {
            s_d[0] = 0;
            for (int k = 0; k < SUP; ++k) s_d[k + 1] += s_d[k];
        }
        __syncthreads();
        // ---- 3: the rest of each chunk once per distinct range, (chunk, range) tasks packed over the workgroup --------------------
        const int T = s_d[SUP];
        for (int t = tid; t < T; t += MAPS_THREADS) {
            int cw = 0;
#pragma unroll
            for (int k = 1; k < SUP; ++k) cw += t >= s_d[k] ? 1 : 0;
            const int j = t - s_d[cw], nn = s_n[cw];
            uint32_t r = s_list[cw][j], Z = 0;
            const int h = nn < MAPS_HEAD ? nn : MAPS_HEAD;
            const uint16_t *bw = &s_b[cw][0];
            int i = h;
            for (; i + 8 <= nn; i += 8) {      // (h is a multiple of 8 unless the chunk ends inside the head: then there is no rest)
                const uint4 q = *reinterpret_cast<const uint4 *>(bw + i);
                const uint32_t e8[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t e = (e8[k >> 1] >> (16 * (k & 1))) & 0xffffu, pr = e & 255u;
                    Z += range_step_z(r, pr, 256u - pr, e >> 8);
                }
            }
            for (; i < nn; ++i) {
                const uint32_t e = bw[i], pr = e & 255u;
                Z += range_step_z(r, pr, 256u - pr, e >> 8);
            }
            s_tail[cw][j] = r | ((Z - 24u * (uint32_t)(nn - h)) << 8);
        }
        __syncthreads();
        // ---- 4: head + tail per start range; the super-chunk's map ---------------------------------------------------------------
        if (w < nc) {
            const uint32_t x0 = r0 - 128u, x1 = r1 - 128u;
            const int k0 = x0 < 64u ? __popcll(m0 & ((1ull << x0) - 1ull)) : d0 + __popcll(m1 & ((1ull << (x0 - 64u)) - 1ull));
            const int k1 = x1 < 64u ? __popcll(m0 & ((1ull << x1) - 1ull)) : d0 + __popcll(m1 & ((1ull << (x1 - 64u)) - 1ull));
            const uint32_t t0 = s_tail[w][k0], t1 = s_tail[w][k1];
            const uint32_t S0 = Z0 - 24u * (uint32_t)head + (t0 >> 8), S1 = Z1 - 24u * (uint32_t)head + (t1 >> 8);
            const uint32_t e0 = (t0 & 255u) | (S0 << 8), e1 = (t1 & 255u) | (S1 << 8);
            J.maps[(size_t)(c0 + w) * 128 + l] = e0;
            J.maps[(size_t)(c0 + w) * 128 + 64 + l] = e1;
            s_m[w][l] = e0;
            s_m[w][64 + l] = e1;
        }
        __syncthreads();
        if (tid < 128) {
            uint32_t r = 128u + tid, S = 0;
            for (int k = 0; k < nc; ++k) {
                const uint32_t e = s_m[k][r - 128u];
                r = e & 255u;
                S += e >> 8;
            }
            J.smaps[(size_t)sup * 128 + tid] = r | (S << 8);   // S < 8 * 256 * 7
        }
    }
}

// One workgroup per partition.  Lane 0 steps through the super-chunk maps (one LDS lookup each) and leaves every
// super-chunk's true start state; then one lane per super-chunk steps through its chunks.
constexpr int WALK_TILE = 64;   // super-chunk maps staged in LDS per step (32 KB)
__global__ __launch_bounds__(256) void k_ent_walk(CodeJobs jobs) {
    const CodeJob &J = jobs.j[blockIdx.y];
    const int p = blockIdx.x;
    if (p >= J.P) return;
    __shared__ uint32_t s_m[WALK_TILE * 128];
    Plan *plan = J.plan;
    uint32_t sb = 0;
    for (int q = 0; q < p; ++q) sb += supers_of(plan, q);
    const uint32_t cb = plan->chunk_base[p], nch = plan->chunk_base[p + 1] - cb, ns = (nch + SUP - 1) / SUP;
    uint32_t r = 255, W = 0;
    for (uint32_t t0 = 0; t0 < ns; t0 += WALK_TILE) {
        const uint32_t tn = ns - t0 < (uint32_t)WALK_TILE ? ns - t0 : (uint32_t)WALK_TILE;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < tn * 128; i += 256) s_m[i] = J.smaps[(size_t)(sb + t0) * 128 + i];
        __syncthreads();
        if (threadIdx.x == 0)
            for (uint32_t k = 0; k < tn; ++k) {
                J.start[cb + (t0 + k) * SUP] = make_uint2(r, W);
                const uint32_t e = s_m[k * 128 + (r - 128u)];
                r = e & 255u;
                W += e >> 8;
            }
    }
    if (threadIdx.x == 0) {
        plan->w_end[p] = W;
        plan->nbytes[p] = (W >= 24 ? (W - 24) / 8 + 1 : 0) + 4;   // bytes emitted while coding + the flush (:130-146)
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ns; i += 256) {
        const uint32_t c = cb + i * SUP, nc = nch - i * SUP < (uint32_t)SUP ? nch - i * SUP : (uint32_t)SUP;
        const uint2 st = J.start[c];
        uint32_t rr = st.x, WW = st.y;
        for (uint32_t k = 1; k < nc; ++k) {
            const uint32_t e = J.maps[(size_t)(c + k - 1) * 128 + (rr - 128u)];
            rr = e & 255u;
            WW += e >> 8;
            J.start[c + k] = make_uint2(rr, WW);
        }
    }
}

// Eight lanes per chunk: lane j first replays bools [0, 32j) of its chunk for the range and bit position only
// (cheap), then codes bools [32j, 32j+32) into the accumulators.  Bools come straight from HBM/L2, sixteen loads
// issued back to back per batch (a version that staged chunks through LDS row by row serialised on load latency:
// 141 us per 1080p frame).
constexpr int ENC_SUB = 8, ENC_SUBLEN = CHUNK / ENC_SUB, ENC_CHUNKS = 256 / ENC_SUB;   // lanes per chunk, bools each codes, chunks per workgroup of 256
__global__ __launch_bounds__(256) void k_ent_encode(CodeJobs jobs) {
    const CodeJob &J = jobs.j[blockIdx.y];
    __shared__ Plan s_plan;   // the plan is consulted per chunk: from LDS, not through a chain of dependent global loads
    for (int i = threadIdx.x; i < (int)(sizeof(Plan) / 4); i += 256) reinterpret_cast<uint32_t *>(&s_plan)[i] = reinterpret_cast<const uint32_t *>(J.plan)[i];
    __syncthreads();
    const Plan *plan = &s_plan;
    const int P = J.P;
    const uint16_t *bools = J.bools;
    const uint2 *start = J.start;
    unsigned long long *acc = J.acc;
    const int j = threadIdx.x & (ENC_SUB - 1);
    for (uint32_t c0 = blockIdx.x * ENC_CHUNKS; c0 < plan->total_chunks; c0 += gridDim.x * ENC_CHUNKS) {
        const uint32_t chunk = c0 + (threadIdx.x / ENC_SUB);
        if (chunk >= plan->total_chunks) continue;
        int p, n;
        uint32_t b0;
        chunk_slice(plan, P, chunk, p, b0, n);
        unsigned long long *out = acc + plan->word_base[p];
        const uint2 st = start[chunk];
        uint32_t r = st.x, W = st.y, split;
        const int lo = j * ENC_SUBLEN < n ? j * ENC_SUBLEN : n, hi = lo + ENC_SUBLEN < n ? lo + ENC_SUBLEN : n;
        const uint16_t *src = bools + b0;
        // sixteen bools per batch, the NEXT batch's loads issued before this batch is stepped through (the lane's bools are
        // contiguous through both phases; the buffer has slack behind its last bool): a lane waits for memory once, not
        // once per batch -- the kernel is a handful of workgroups and nothing else hides that latency
        uint32_t e[16], nx[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) e[k] = src[k];
        int i0 = 0;
        for (; i0 < lo; i0 += 16) {                     // lo is a multiple of 16 (or n: then the guard below ends it)
#pragma unroll
            for (int k = 0; k < 16; ++k) nx[k] = src[i0 + 16 + k];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (i0 + k < lo) W += range_step(r, e[k] & 255u, e[k] >> 8, split);
#pragma unroll
            for (int k = 0; k < 16; ++k) e[k] = nx[k];
        }
        uint32_t widx = W >> 5;
        unsigned long long cur = 0, nxt = 0;   // sums for output words widx and widx+1
        for (; i0 < hi; i0 += 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) nx[k] = src[i0 + 16 + k];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (i0 + k >= hi) continue;
                const uint32_t bit = e[k] >> 8;
                const int o = (int)(W & 31u);
                const int s = range_step(r, e[k] & 255u, bit, split);
                if (bit) {   // split occupies stream bits W .. W+7 (bit 0 = most significant bit of the first byte)
                    if (o <= 24) cur += (unsigned long long)split << (24 - o);
                    else { cur += split >> (o - 24); nxt += ((unsigned long long)split << (56 - o)) & 0xffffffffull; }
                }
                W += s;
                if ((W >> 5) != widx) {
                    if (cur) atomicAdd(&out[widx], cur);
                    cur = nxt; nxt = 0; ++widx;
                }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) e[k] = nx[k];
        }
        if (cur) atomicAdd(&out[widx], cur);
        if (nxt) atomicAdd(&out[widx + 1], nxt);
    }
}

// carries + bytes.  One workgroup of sixteen waves per partition, from the least significant word up, 1024 words per
// step: every word adds the high half of the word below it; what is left is a ripple of single carries, resolved by
// carry-lookahead inside each wave (the ballot trick), across the sixteen waves, and from step to step.
constexpr int FIN_WAVES = 16;
__global__ __launch_bounds__(64 * FIN_WAVES) void k_ent_finish(CodeJobs jobs) {
    const CodeJob &J = jobs.j[blockIdx.y];
    const int p = blockIdx.x;
    if (p >= J.P) return;
    __shared__ uint32_t s_gp[2][2][FIN_WAVES];
    const Plan *plan = J.plan;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t nb = plan->nbytes[p], nw = (nb + 3) / 4;
    const unsigned long long *in = J.acc + plan->word_base[p];
    uint8_t *out = J.bytes + (size_t)plan->word_base[p] * 4;
    const int m = blockIdx.y >> 1, side = blockIdx.y & 1;
    uint8_t *const frame = jobs.frame[m];
    if (frame) {
        // gather_frame on the fly: `head` bytes left for the host, first partition, the 3-byte sizes of all coefficient
        // partitions but the last, the partitions.  frame[0] = frame size (0 = a coder overflowed its scratch or the
        // frame does not fit), frame[1] = size of the first partition; the sizes are known since k_ent_walk.
        const Plan *pc = jobs.j[2 * m].plan, *ph = jobs.j[2 * m + 1].plan;
        const int P = jobs.j[2 * m].P;
        const uint32_t head = jobs.head[m];
        const uint32_t table = head + ph->nbytes[0];
        uint32_t o = table + 3u * (uint32_t)(P - 1), mine = head;
        for (int q = 0; q < P; ++q) {
            if (side == 0 && q == p) mine = o;
            o += pc->nbytes[q];
        }
        const bool ok = !(pc->overflow || ph->overflow || o > jobs.capacity[m]);
        if (side == 1 && threadIdx.x == 0) {
            reinterpret_cast<uint32_t *>(frame)[0] = ok ? o : 0u;
            reinterpret_cast<uint32_t *>(frame)[1] = ph->nbytes[0];
        }
        if (!ok) return;
        if (side == 1 && threadIdx.x < 3u * (uint32_t)(P - 1))
            frame[16 + table + threadIdx.x] = (uint8_t)(pc->nbytes[threadIdx.x / 3] >> (8 * (threadIdx.x % 3)));
        out = frame + 16 + mine;
    }
    uint32_t C = 0;   // carry into the least significant word of the step
    int par = 0;
    for (int hi_end = (int)nw; hi_end > 0; hi_end -= 64 * FIN_WAVES, par ^= 1) {
        const int j = hi_end - 1 - (wv * 64 + lane);            // wave 0, lane 0 = least significant word of the step
        const unsigned long long v = j >= 0 ? in[j] : 0ull;
        unsigned long long from_below = __shfl_up(v >> 32, 1);
        if (lane == 0) from_below = (j >= 0 && j + 1 < (int)nw) ? in[j + 1] >> 32 : 0ull;
        const unsigned long long t = (v & 0xffffffffull) + from_below;   // < 2^33: the high parts are tiny
        const uint32_t r = (uint32_t)t;
        const unsigned long long G = __ballot((t >> 32) != 0), Pm = __ballot(r == 0xffffffffu);
        // ripple c[i+1] = g[i] | (p[i] & c[i]) for all 64 lanes at once: the carries of the addition (G|P) + G + carry-in
        const unsigned long long A = G | Pm, S0 = A + G;
        if (lane == 0) {
            s_gp[par][0][wv] = S0 < A ? 1u : 0u;           // the wave hands a carry up whatever comes in
            s_gp[par][1][wv] = S0 == ~0ull ? 1u : 0u;      // ... only if one comes in
        }
        __syncthreads();
        uint32_t Gw = 0, Pw = 0;
#pragma unroll
        for (int k = 0; k < FIN_WAVES; ++k) {
            Gw |= s_gp[par][0][k] << k;
            Pw |= s_gp[par][1][k] << k;
        }
        const uint32_t Sw = (Gw | Pw) + Gw + C, into = Sw ^ Pw;   // bit k = carry into wave k, bit FIN_WAVES = out of the step
        const unsigned long long S = S0 + ((into >> wv) & 1u);
        const unsigned long long carries = S ^ Pm;                // bit i = carry into lane i
        const uint32_t word = r + (uint32_t)((carries >> lane) & 1ull);
        C = (into >> FIN_WAVES) & 1u;
        if (j >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((uint32_t)(4 * j + k) < nb) out[4 * j + k] = (uint8_t)(word >> (24 - 8 * k));
        }
    }
    if (threadIdx.x == 0) J.sizes[p] = (int32_t)nb;
}

// exclusive scan of up to 64 k values by one workgroup (the macroblock-header counts): one launch instead of three
__global__ __launch_bounds__(1024) void k_scan_small(uint32_t *v, int n) {
    __shared__ uint32_t s[1024];
    const int t = threadIdx.x, per = (n + 1023) / 1024, i0 = t * per;
    uint32_t a = 0;
    for (int j = 0; j < per; ++j) a += i0 + j < n ? v[i0 + j] : 0u;
    s[t] = a;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t add = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] += add;
        __syncthreads();
    }
    uint32_t run = s[t] - a;
    for (int j = 0; j < per; ++j) {
        if (i0 + j >= n) break;
        const uint32_t x = v[i0 + j];
        v[i0 + j] = run;
        run += x;
    }
    if (t == 1023) v[n] = s[1023];
}

}  // namespace ent

// exclusive prefix sum of v[0..n) in place, total in v[n]
void launch_scan_exclusive(hipStream_t s, uint32_t *v, uint32_t *tile_sum, int n) {
    if (n <= 65536) {
        hipLaunchKernelGGL(ent::k_scan_small, dim3(1), dim3(1024), 0, s, v, n);
        return;
    }
    const int ntiles = (n + ent::SCAN_TILE - 1) / ent::SCAN_TILE;
    hipLaunchKernelGGL(ent::k_scan_tiles, dim3(ntiles), dim3(256), 0, s, v, tile_sum, n);
    hipLaunchKernelGGL(ent::k_scan_top, dim3(1), dim3(1024), 0, s, tile_sum, ntiles);
    hipLaunchKernelGGL(ent::k_scan_apply, dim3(ntiles), dim3(256), 0, s, v, tile_sum, n, ntiles);
}

// the boolean coder proper on bool strings that are laid out as eb.plan says (steps 2-5 above), the emit kernel having
// cleared the accumulators; one job (P partitions of a) or two (+ Pb partitions of b) in the same four launches
static ent::CodeJob code_job(const EntBuffers &eb, int P) {
    ent::CodeJob j;
    j.bools = eb.bools;
    j.plan = eb.plan;
    j.maps = eb.maps;
    j.smaps = eb.maps + (size_t)eb.cap_chunks * 128;
    j.start = reinterpret_cast<uint2 *>(eb.start);
    j.acc = reinterpret_cast<unsigned long long *>(eb.acc);
    j.bytes = eb.bytes;
    j.sizes = eb.sizes;
    j.P = P;
    return j;
}
// VP8HIP_EXPERIMENT_SKIP_ENT: bit i leaves out launch i of the frame's entropy stage (the bytes are then garbage: what a launch
// costs with the part full, measured by the throughput without it -- scripts/ab_bitstream.sh)
unsigned ent_skip_mask() {
    static const unsigned m = [] { const char *v = experiment_env("VP8HIP_EXPERIMENT_SKIP_ENT"); return v ? (unsigned)strtoul(v, nullptr, 0) : 0u; }();
    return m;
}
static void bool_code(hipStream_t s, const ent::CodeJobs &jobs, int njobs, unsigned skip = 0) {
    int maxP = 1;
    for (int i = 0; i < njobs; ++i) maxP = jobs.j[i].P > maxP ? jobs.j[i].P : maxP;
    if (!(skip & 32)) hipLaunchKernelGGL(ent::k_ent_maps, dim3(512, njobs), dim3(ent::MAPS_THREADS), 0, s, jobs);
    if (!(skip & 64)) hipLaunchKernelGGL(ent::k_ent_walk, dim3(maxP, njobs), dim3(256), 0, s, jobs);
    if (!(skip & 128)) hipLaunchKernelGGL(ent::k_ent_encode, dim3(512, njobs), dim3(256), 0, s, jobs);
    if (!(skip & 256)) hipLaunchKernelGGL(ent::k_ent_finish, dim3(maxP, njobs), dim3(64 * ent::FIN_WAVES), 0, s, jobs);
}
void launch_bool_code(hipStream_t s, const EntBuffers &eb, int P) {
    ent::CodeJobs jobs{};
    jobs.j[0] = jobs.j[1] = code_job(eb, P);
    bool_code(s, jobs, 1);
}
void launch_frame_code(hipStream_t s, const EntBuffers &coef, int P, const EntBuffers &hdr, uint32_t head, uint32_t capacity, uint8_t *frame) {
    ent::CodeJobs jobs{};
    jobs.j[0] = code_job(coef, P);
    jobs.j[1] = code_job(hdr, 1);
    jobs.frame[0] = frame;
    jobs.head[0] = head;
    jobs.capacity[0] = capacity;
    bool_code(s, jobs, 2);
}
void launch_frame_code_batch(hipStream_t s, const FrameEntropy *e, const FrameOut *fo, int n) {
    ent::CodeJobs jobs{};
    for (int m = 0; m < n; ++m) {
        jobs.j[2 * m] = code_job(*e[m].coef, e[m].P);
        jobs.j[2 * m + 1] = code_job(*e[m].hdr, 1);
        jobs.frame[m] = fo[m].frame;
        jobs.head[m] = fo[m].head;
        jobs.capacity[m] = fo[m].capacity;
    }
    bool_code(s, jobs, 2 * n, ent_skip_mask());   // (the switch acts on batched launches only: the first frames run one by one and fill every buffer)
}

static ent::Geom make_geom(const EntBuffers &eb, int mbw, int mbh, int P) {
    ent::Geom g;
    g.mbw = mbw;
    g.mbh = mbh;
    g.P = P;
    uint32_t sb = 0;
    for (int p = 0; p <= ENT_MAX_PARTITIONS; ++p) {
        g.slot_base[p] = sb;
        if (p < P) sb += (uint32_t)((mbh - p + P - 1) / P) * mbw * 25;
    }
    g.cap_bools = eb.cap_bools;
    g.cap_chunks = eb.cap_chunks;
    g.cap_words = eb.cap_words;
    return g;
}

void launch_ent_encode(hipStream_t s, const MBOut &o, const uint8_t *third_ctx, const uint32_t *probs, const EntBuffers &eb,
                       int mbw, int mbh, int P, bool code) {
    const ent::Geom g = make_geom(eb, mbw, mbh, P);
    const int nslots = mbw * mbh * 25;
    EntPlan *plan = eb.plan;
    hipLaunchKernelGGL(ent::k_ent_boolcount, dim3((nslots + 255) / 256), dim3(256), 0, s, o.coeffs, o.nz, o.parts, g, eb.offs);
    launch_scan_exclusive(s, eb.offs, eb.tile_sum, nslots);
    hipLaunchKernelGGL(ent::k_ent_plan, dim3(1), dim3(64), 0, s, eb.offs, g, plan);
    hipLaunchKernelGGL(ent::k_ent_emit, dim3((nslots + 255) / 256), dim3(256), 0, s, o.coeffs, o.nz, o.parts, third_ctx, probs,
                       eb.offs, plan, g, eb.bools, reinterpret_cast<unsigned long long *>(eb.acc));
    if (code) launch_bool_code(s, eb, P);
}

}  // namespace vp8
