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
//                 (a workgroup stays inside one macroblock row = one partition), flushed with atomics;
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

__device__ __forceinline__ int classify(int mag) {   // tokenize_block, :263-345
    return mag <= 4 ? mag : (mag <= 6 ? T_CAT1 : (mag <= 10 ? T_CAT2 : (mag <= 18 ? T_CAT3 : (mag <= 34 ? T_CAT4 : (mag <= 66 ? T_CAT5 : T_CAT6)))));
}

// block order inside a macroblock as coded (:371-403): [24 if 16x16], 0..15, 16..23; plane context ctx1
__device__ __forceinline__ int plane_ctx(int b, bool has_y2) { return b == 24 ? 1 : (b < 16 ? (has_y2 ? 0 : 3) : 2); }

__global__ __launch_bounds__(256) void k_ent_flags(const int16_t *coeffs, uint8_t *flags, int nblocks) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nblocks) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(coeffs + (size_t)i * 16);
    const uint4 a = p[0], b = p[1];
    const uint32_t rest = (a.x & 0xffff0000u) | a.y | a.z | a.w | b.x | b.y | b.z | b.w;
    flags[i] = (uint8_t)(((a.x & 0xffffu) | rest ? 1 : 0) | (rest ? 2 : 0));   // bit0: any coefficient, bit1: any past the first
}

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

constexpr int CNT_MBS = 10;   // macroblocks per workgroup of k_ent_count (250 of 256 threads)

// counts[part][NCTX][2] = {zero branches taken, branches seen} (the reference's num / denom - 1)
__global__ __launch_bounds__(256) void k_ent_count(const int16_t *coeffs, const int32_t *nzc, const int32_t *parts,
                                                   const uint8_t *flags, uint8_t *third_ctx, uint32_t *counts, int mbw,
                                                   int num_partitions) {
    __shared__ uint32_t s_h[NCTX * 2];
    for (int i = threadIdx.x; i < NCTX * 2; i += 256) s_h[i] = 0;
    __syncthreads();
    const int mb_row = blockIdx.y;
    const int mb_col = blockIdx.x * CNT_MBS + threadIdx.x / 25, b = threadIdx.x % 25;
    if (threadIdx.x < CNT_MBS * 25 && mb_col < mbw) {
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
                const uint32_t nodes = k_path_nodes[t];
                const int bits = k_path_bits[t], len = k_path_len[t];
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
    uint32_t *dst = counts + (size_t)(mb_row % num_partitions) * NCTX * 2;
    for (int i = threadIdx.x; i < NCTX * 2; i += 256)
        if (s_h[i]) atomicAdd(&dst[i], s_h[i]);
}

// num_div_denom (:764-778) + the denominators of partition 0 that the host inspects (vp8enc.cpp:69-76):
// every partition's denominator starts at 1 (:552)
__global__ __launch_bounds__(256) void k_ent_probs(const uint32_t *counts, uint32_t *probs, uint32_t *denom0, int num_partitions) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NCTX) return;
    uint32_t num = 0, den = 0;
    for (int p = 0; p < num_partitions; ++p) {
        num += counts[((size_t)p * NCTX + i) * 2];
        den += counts[((size_t)p * NCTX + i) * 2 + 1] + 1u;
    }
    num = (num << 8) / den;
    probs[i] = num > 255u ? 255u : (num == 0u ? 1u : num);
    denom0[i] = counts[(size_t)i * 2 + 1] + 1u;
}

}  // namespace ent

void launch_ent_count(hipStream_t s, const MBOut &o, uint8_t *flags, uint8_t *third_ctx, uint32_t *counts, uint32_t *probs,
                      uint32_t *denom0, int mbw, int mbh, int num_partitions) {
    const int nblocks = mbw * mbh * 25;
    hipMemsetAsync(counts, 0, sizeof(uint32_t) * ent::NCTX * 2 * num_partitions, s);
    hipLaunchKernelGGL(ent::k_ent_flags, dim3((nblocks + 255) / 256), dim3(256), 0, s, o.coeffs, flags, nblocks);
    hipLaunchKernelGGL(ent::k_ent_count, dim3((mbw + ent::CNT_MBS - 1) / ent::CNT_MBS, mbh), dim3(256), 0, s, o.coeffs, o.nz,
                       o.parts, flags, third_ctx, counts, mbw, num_partitions);
    hipLaunchKernelGGL(ent::k_ent_probs, dim3((ent::NCTX + 255) / 256), dim3(256), 0, s, counts, probs, denom0, num_partitions);
}

}  // namespace vp8
