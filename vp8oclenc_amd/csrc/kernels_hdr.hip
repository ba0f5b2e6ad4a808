// kernels_hdr.hip -- the first partition (frame header + macroblock headers: encode_header, src/entropy_host.cpp:709-1256)
// produced on the device from the frame's results that already live there.
//
// The reference codes it on one host thread, bool by bool (2-3 ms per 1080p frame: five times the whole device
// path).  The boolean coder itself is already parallel here (kernels_ent.hip, part 2: chunk maps, walk, accumulate,
// carry-lookahead); what this file adds is the bool STRING of the first partition:
//   k_hdr_count   one thread per macroblock runs the macroblock-header template (vp8_mbhdr.h, the same source the
//                 host coder compiles) with a counting sink: bools per macroblock, the motion-vector statistics
//                 behind mv_prob_update() and the frame's census (segments, references, skipped, replaced) as
//                 per-workgroup partial sums -- no same-address HBM atomics;
//   k_hdr_frame   one workgroup turns the counts into offsets (exclusive scan), folds the partials into the frame's
//                 probability table, writes the frame-level
//                 bools (the 1056 coefficient-probability updates in parallel, the rest by one lane) and the plan;
//   k_hdr_emit    one thread per macroblock again, now emitting (probability, bit) pairs at its offset;
// followed by the generic coder on that string.  Byte-exact against the host coder and the reference.
#define VP8_MBHDR_DEVICE 1
#include "vp8hip_dev.h"
#include "vp8_mbhdr.h"

namespace vp8 {
namespace hdr {

#define VP8_RFC_TABLE __device__ __constant__ const
#include "vp8_rfc6386_tables.inc"

using namespace vp8hdr;

// Lanes per macroblock of the two walks (k_hdr_count, k_hdr_emit): every macroblock takes its own path through the header template, so
// the lanes of a wavefront serialise.  ONE video wants the walk short: one macroblock per SIXTEEN lanes, a wavefront walks 4 paths
// instead of 64 and the frame spreads over sixteen times as many wavefronts (104 us with one per lane, 47 us with one per four at
// 1080p, alone on the part).  A BATCH of frames with the part full wants it cheap: what counts there is the issue slots a launch takes
// from the other batches' kernels, and a wavefront that walks 64 paths issues their common instructions once -- a twelfth of the
// wavefront instructions (profiles/README.md, round 5).
constexpr int HDR_LPM_SHIFT_ONE_VIDEO = 4;
__host__ __device__ constexpr int hdr_mb_per_wg(int lpm_shift) { return 256 >> lpm_shift; }
constexpr int NSTAT = 2 * MV_PROBS * 2 + 8;   // mv num/den + {seg0..3, coded (nz != 0), ref last, ref golden, replaced}
enum { ST_SEG = 76, ST_CODED = 80, ST_LAST = 81, ST_GF = 82, ST_REPLACED = 83 };

struct View {
    const int32_t *seg_, *nz_, *ref_, *parts_, *is_inter_, *modes_;
    const int16_t *vec_;
    int mbw_;
    __device__ __forceinline__ int mbw() const { return mbw_; }
    __device__ __forceinline__ Mv vec(int mb, int k) const {
        const uint32_t w = reinterpret_cast<const uint32_t *>(vec_)[mb * 4 + k];
        return Mv{(int16_t)(w & 0xffffu), (int16_t)(w >> 16)};
    }
    __device__ __forceinline__ bool inter(int mb) const { return is_inter_ ? is_inter_[mb] != 0 : true; }
    __device__ __forceinline__ int parts(int mb) const { return parts_[mb]; }
    __device__ __forceinline__ int ref(int mb) const { return ref_[mb]; }
    __device__ __forceinline__ int seg(int mb) const { return seg_[mb]; }
    __device__ __forceinline__ int nz(int mb) const { return nz_[mb]; }
    __device__ __forceinline__ int mode(int mb, int b) const { return modes_ ? modes_[16 * mb + b] : 0; }
};

struct CountSink {
    uint32_t n = 0;
    uint32_t *stat;   // workgroup's LDS tallies
    __device__ __forceinline__ void put(int, int) { ++n; }
    __device__ __forceinline__ void mv_stat(int comp, int idx, int bit) {
        atomicAdd(&stat[(comp * MV_PROBS + idx) * 2], 1u - (uint32_t)bit);
        atomicAdd(&stat[(comp * MV_PROBS + idx) * 2 + 1], 1u);
    }
};
struct EmitSink {
    uint16_t *out;
    const uint8_t *sym;
    __device__ __forceinline__ void put(int p, int bit) { *out++ = (uint16_t)((p >= HDR_SYM ? sym[p - HDR_SYM] : p) | (bit << 8)); }
    __device__ __forceinline__ void mv_stat(int, int, int) {}
};

struct Params {
    View v;
    int mbs, key;
    int is_golden, is_altref, loop_filter_type, sharpness, partitions_log2;   // sharpness == INT32_MIN (VP8HIP_SHARPNESS_ON_DEVICE): take it from `strength`
    const SegData *sd;
    const int32_t *strength;      // {reductor, sharpness, sharpness in force} of vp8hip_auto_segments / the check_SSIM verdict
    const uint32_t *probs, *denom0;
    uint32_t cap_bools, cap_chunks, cap_words;
    int lpm_shift;                // lanes per macroblock of the two macroblock walks = 1 << lpm_shift (hdr_lanes_shift)
};

__device__ __forceinline__ void hdr_count_body(int vb, const Params &a, uint32_t *cnt, uint32_t *partial) {
    __shared__ uint32_t s_stat[NSTAT];
    for (int i = threadIdx.x; i < NSTAT; i += 256) s_stat[i] = 0;
    __syncthreads();
    const int mb = vb * hdr_mb_per_wg(a.lpm_shift) + ((int)threadIdx.x >> a.lpm_shift);
    if ((threadIdx.x & ((1u << a.lpm_shift) - 1u)) == 0 && mb < a.mbs) {
        CountSink s;
        s.stat = s_stat;
        mb_header(a.v, mb, a.key != 0, k_kf_bmode_probs, s);
        cnt[mb] = s.n;
        atomicAdd(&s_stat[ST_SEG + (a.v.seg(mb) & 3)], 1u);
        if (a.v.nz(mb) != 0) atomicAdd(&s_stat[ST_CODED], 1u);
        if (!a.key) {
            const int r = a.v.ref(mb);
            if (r == 0) atomicAdd(&s_stat[ST_LAST], 1u);
            if (r == 1) atomicAdd(&s_stat[ST_GF], 1u);
            if (!a.v.inter(mb)) atomicAdd(&s_stat[ST_REPLACED], 1u);
        }
    }
    __syncthreads();
    // the frame's totals: one row of NSTAT words, zero at rest (k_hdr_frame clears it after reading)
    for (int i = threadIdx.x; i < NSTAT; i += 256)
        if (s_stat[i]) atomicAdd(&partial[i], s_stat[i]);
}
__global__ __launch_bounds__(256) void k_hdr_count(Params a, uint32_t *cnt, uint32_t *partial) { hdr_count_body(blockIdx.x, a, cnt, partial); }

// frame-level bool writer of one lane
struct Lane {
    uint16_t *out;
    uint32_t n = 0;
    __device__ __forceinline__ void put(int p, int bit) { out[n++] = (uint16_t)(p | ((bit ? 1 : 0) << 8)); }
    __device__ __forceinline__ void flag(int b) { put(128, b); }
    __device__ __forceinline__ void literal(int v, int bits) {
        for (int m = 1 << (bits - 1); m; m >>= 1) flag((v & m) != 0);
    }
    __device__ __forceinline__ void qdelta(int d) {
        if (!d) { flag(0); return; }
        flag(1);
        literal(d < 0 ? -d : d, 4);
        flag(d < 0);
    }
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// frame header (encode_header :735-1061).  One workgroup.  cnt: bools per macroblock in, their exclusive prefix sums out.
__device__ __forceinline__ void hdr_frame_body(const Params &a, uint32_t *partial, uint32_t *cnt, uint16_t *bools,
                                               uint8_t *sym_out, EntPlan *plan, uint32_t *info) {
    __shared__ uint32_t s_tot[NSTAT];
    __shared__ int32_t s_sd[4 * SD_INTS];   // the segment data, read many times by the one lane that writes the frame-level bools
    __shared__ uint32_t s_scan[256];
    __shared__ uint32_t s_n1, s_total;
    __shared__ uint8_t s_sym[64];
    const int t = threadIdx.x;
    {   // exclusive scan of the per-macroblock bool counts, in place (at most 127 values per thread at 4K): the offsets
        // k_hdr_emit writes at; done here rather than by the three-launch tile scan
        const int per = (a.mbs + 255) / 256, i0 = t * per;
        uint32_t acc = 0;
        for (int j = 0; j < per; ++j) acc += i0 + j < a.mbs ? cnt[i0 + j] : 0u;
        s_scan[t] = acc;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const uint32_t add = t >= d ? s_scan[t - d] : 0u;
            __syncthreads();
            s_scan[t] += add;
            __syncthreads();
        }
        uint32_t run = s_scan[t] - acc;
        for (int j = 0; j < per && i0 + j < a.mbs; ++j) {
            const uint32_t x = cnt[i0 + j];
            cnt[i0 + j] = run;
            run += x;
        }
        if (t == 255) s_total = s_scan[255];
        __syncthreads();
    }
    for (int i = t; i < NSTAT; i += 256) {
        s_tot[i] = partial[i];
        partial[i] = 0;   // zero at rest, for the next frame's k_hdr_count
    }
    for (int i = t; i < 4 * SD_INTS; i += 256) s_sd[i] = a.sd->v[i];
    __syncthreads();
    const int mbs = a.mbs;
    const bool key = a.key != 0;
    const int32_t *sd = s_sd;
    const int sharpness = a.sharpness != INT32_MIN ? a.sharpness : a.strength[2];   // (a negative sharpness is a value: the reference's overflowed accumulator)
    const int replaced = key ? 0 : (int)s_tot[ST_REPLACED];
    if (t == 0) {
        // ---- the probability table of this frame (what the macroblock headers refer to symbolically) ----
        for (int i = 0; i < 64; ++i) s_sym[i] = 128;
        const int c0 = s_tot[ST_SEG], c1 = s_tot[ST_SEG + 1], c2 = s_tot[ST_SEG + 2], c3 = s_tot[ST_SEG + 3];
        int d01 = c0 + c1, d23 = c2 + c3;
        s_sym[SYM_SEG + 0] = (uint8_t)(d01 * 255 / mbs);
        d01 += d01 == 0;
        d23 += d23 == 0;
        s_sym[SYM_SEG + 1] = (uint8_t)(c0 * 255 / d01);
        s_sym[SYM_SEG + 2] = (uint8_t)(c2 * 255 / d23);
        s_sym[SYM_SKIP] = (uint8_t)clampi((int)s_tot[ST_CODED] * 256 / mbs, 2, 254);   // frames.skip_prob, loop_filter.h:37-44
        int prob_intra = replaced * 255 / mbs;
        if (replaced > 0 && prob_intra < 2) prob_intra = 2;
        if (replaced < mbs && prob_intra > 254) prob_intra = 254;
        const int last = s_tot[ST_LAST], gf = s_tot[ST_GF];
        s_sym[SYM_INTRA] = (uint8_t)prob_intra;
        s_sym[SYM_GF] = (uint8_t)clampi(gf * 256 / (mbs - last + 1), 1, 255);
        s_sym[SYM_LAST] = (uint8_t)clampi(last * 256 / mbs, 1, 255);
        const uint8_t ym[4] = {112, 86, 140, 37}, uvm[3] = {162, 101, 204};
        for (int i = 0; i < 4; ++i) s_sym[SYM_YMODE + i] = replaced > 7 ? 0 : ym[i];
        for (int i = 0; i < 3; ++i) s_sym[SYM_UVMODE + i] = replaced > 7 ? 0 : uvm[i];
        for (int i = 0; i < 2 * MV_PROBS; ++i) {
            const uint32_t num = s_tot[2 * i], den = s_tot[2 * i + 1] + 1u;   // the reference starts its denominators at 1 (:1036)
            int p = (int)(uint8_t)((num << 8) / den);
            p &= ~1;
            s_sym[SYM_MV + i] = (uint8_t)clampi(p, 2, 254);
        }
        // ---- bools before the coefficient-probability updates ----
        enum { SD = SD_INTS };
        Lane w{bools};
        if (key) { w.flag(0); w.flag(0); }
        w.flag(!key);
        if (!key) {
            w.flag(1); w.flag(1); w.flag(1);
            for (int i = 0; i < 4; ++i) { w.flag(1); w.literal(sd[i * SD + SD_Y_AC_I], 7); w.flag(0); }
            for (int i = 0; i < 4; ++i) { w.flag(1); w.literal(sd[i * SD + SD_LOOP_FILTER_LEVEL], 6); w.flag(0); }
            for (int i = 0; i < 3; ++i) { w.flag(1); w.literal(s_sym[SYM_SEG + i], 8); }
        }
        w.flag(a.loop_filter_type);
        w.literal(sd[SD_LOOP_FILTER_LEVEL], 6);
        w.literal(sharpness, 3);
        w.flag(0);
        w.literal(a.partitions_log2, 2);
        w.literal(sd[SD_Y_AC_I], 7);
        w.qdelta(sd[SD_Y_DC_IDELTA]);
        w.qdelta(sd[SD_Y2_DC_IDELTA]);
        w.qdelta(sd[SD_Y2_AC_IDELTA]);
        w.qdelta(sd[SD_UV_DC_IDELTA]);
        w.qdelta(sd[SD_UV_AC_IDELTA]);
        if (key) {
            w.flag(0);
        } else {
            w.flag(a.is_golden);
            w.flag(a.is_altref);
            if (!a.is_golden) w.literal(0, 2);
            if (!a.is_altref) w.literal(0, 2);
            w.flag(0); w.flag(0); w.flag(0); w.flag(1);
        }
        s_n1 = w.n;
    }
    __syncthreads();
    // ---- token_prob_update(): 1056 contexts, 1 bool (never seen) or 9 (flag + 8-bit probability) each ----
    constexpr int PER = 5;   // 256 x 5 >= 1056
    uint32_t mine = 0;
    for (int k = 0; k < PER; ++k) {
        const int i = t * PER + k;
        if (i < ENT_NCTX) mine += a.denom0[i] < 2 ? 1u : 9u;
    }
    s_scan[t] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t add = t >= d ? s_scan[t - d] : 0u;
        __syncthreads();
        s_scan[t] += add;
        __syncthreads();
    }
    {
        Lane w{bools + s_n1 + s_scan[t] - mine};
        const uint8_t *upd = &k_coeff_update_probs[0][0][0][0];
        for (int k = 0; k < PER; ++k) {
            const int i = t * PER + k;
            if (i >= ENT_NCTX) break;
            if (a.denom0[i] < 2) {
                w.put(upd[i], 0);
            } else {
                w.put(upd[i], 1);
                w.literal((int)(a.probs[i] & 255u), 8);
            }
        }
    }
    __syncthreads();
    if (t == 0) {
        const uint32_t n2 = s_scan[255];
        Lane w{bools + s_n1 + n2};
        w.flag(1);
        w.literal(s_sym[SYM_SKIP], 8);
        if (!key) {
            w.literal(s_sym[SYM_INTRA], 8);
            w.literal(s_sym[SYM_LAST], 8);
            w.literal(s_sym[SYM_GF], 8);
            if (replaced > 7) {
                w.flag(1);
                for (int i = 0; i < 4; ++i) w.literal(0, 8);
                w.flag(1);
                for (int i = 0; i < 3; ++i) w.literal(0, 8);
            } else {
                w.flag(0);
                w.flag(0);
            }
            for (int i = 0; i < 2 * MV_PROBS; ++i) {
                w.put((&k_mv_update_probs[0][0])[i], 1);
                w.literal(s_sym[SYM_MV + i] >> 1, 7);
            }
        }
        const uint32_t H = s_n1 + n2 + w.n, total = H + s_total;
        const uint32_t chunks = (total + ENT_CHUNK - 1) / ENT_CHUNK, words = (total * 7 + 31) / 32 + 4;
        const bool over = total > a.cap_bools || chunks > a.cap_chunks || words > a.cap_words;
        for (int p = 0; p <= ENT_MAX_PARTITIONS; ++p) plan->bool_base[p] = plan->chunk_base[p] = plan->word_base[p] = 0;
        for (int p = 0; p < ENT_MAX_PARTITIONS; ++p) plan->nbools[p] = plan->w_end[p] = plan->nbytes[p] = 0;
        plan->overflow = over ? 1u : 0u;
        plan->total_chunks = over ? 0u : chunks;
        if (!over) {
            plan->nbools[0] = total;
            plan->bool_base[1] = total;
            plan->chunk_base[1] = chunks;
            plan->word_base[1] = words;
        }
        info[0] = H;
        info[1] = s_sym[SYM_SKIP];
        info[2] = (uint32_t)replaced;
    }
    __syncthreads();
    if (t < 64) sym_out[t] = s_sym[t];
}
__global__ __launch_bounds__(256) void k_hdr_frame(Params a, uint32_t *partial, uint32_t *cnt, uint16_t *bools,
                                                   uint8_t *sym_out, EntPlan *plan, uint32_t *info) {
    hdr_frame_body(a, partial, cnt, bools, sym_out, plan, info);
}
struct FrameItem {
    Params a;
    uint32_t *partial, *cnt;
    uint16_t *bools;
    uint8_t *sym_out;
    EntPlan *plan;
    uint32_t *info;
};
__global__ __launch_bounds__(256) void k_hdr_frame_b(BatchOf<FrameItem> b) {   // one workgroup per member of the batch
    const FrameItem &f = b.item[blockIdx.x];
    hdr_frame_body(f.a, f.partial, f.cnt, f.bools, f.sym_out, f.plan, f.info);
}

__device__ __forceinline__ void hdr_emit_body(int vb, int nvb, const Params &a, const uint32_t *offs, const uint8_t *sym, const EntPlan *plan,
                                              const uint32_t *info, uint16_t *bools, unsigned long long *acc) {
    for (uint32_t i = vb * 256 + threadIdx.x, n = plan->word_base[1]; i < n; i += nvb * 256) acc[i] = 0ull;   // the coder's accumulators
    __shared__ uint8_t s_sym[64];
    if (threadIdx.x < 64) s_sym[threadIdx.x] = sym[threadIdx.x];
    __syncthreads();
    const int mb = vb * hdr_mb_per_wg(a.lpm_shift) + ((int)threadIdx.x >> a.lpm_shift);
    if ((threadIdx.x & ((1u << a.lpm_shift) - 1u)) != 0 || mb >= a.mbs || plan->overflow) return;
    EmitSink s{bools + info[0] + offs[mb], s_sym};
    mb_header(a.v, mb, a.key != 0, k_kf_bmode_probs, s);
}
__global__ __launch_bounds__(256) void k_hdr_emit(Params a, const uint32_t *offs, const uint8_t *sym, const EntPlan *plan, const uint32_t *info,
                                                  uint16_t *bools, unsigned long long *acc) {
    hdr_emit_body(blockIdx.x, gridDim.x, a, offs, sym, plan, info, bools, acc);
}

// vp8enc.cpp:69-76 on the device: contexts that never occurred take the default probability
__global__ __launch_bounds__(256) void k_default_probs(uint32_t *probs, const uint32_t *denom0) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ENT_NCTX && denom0[i] < 2) probs[i] = (&k_default_coeff_probs[0][0][0][0])[i];
}

}  // namespace hdr

const uint8_t *hdr_default_coeff_probs() {
    static const uint8_t *p = nullptr;
    if (!p) {
        void *q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(hdr::k_default_coeff_probs)) == hipSuccess) p = static_cast<const uint8_t *>(q);
    }
    return p;
}

void launch_default_probs(hipStream_t s, uint32_t *probs, const uint32_t *denom0) {
    hipLaunchKernelGGL(hdr::k_default_probs, dim3((ENT_NCTX + 255) / 256), dim3(256), 0, s, probs, denom0);
}

static hdr::Params make_hdr_params_impl(const MBOut &o, const int32_t *is_inter, const int32_t *modes, const HdrFrame &f, const SegData *d_sd,
                                        const int32_t *strength, const uint32_t *probs, const uint32_t *denom0, const EntBuffers &eb, int mbw, int mbh);
static inline hdr::Params make_hdr_params(const MBOut &o, const int32_t *is_inter, const int32_t *modes, const HdrFrame &f, const SegData *d_sd,
                                          const int32_t *strength, const uint32_t *probs, const uint32_t *denom0, const EntBuffers &eb, int mbw, int mbh) {
    return make_hdr_params_impl(o, is_inter, modes, f, d_sd, strength, probs, denom0, eb, mbw, mbh);
}

void launch_hdr_encode(hipStream_t s, const MBOut &o, const int32_t *is_inter, const int32_t *modes, const HdrFrame &f, const SegData *d_sd,
                       const int32_t *strength, const uint32_t *probs, const uint32_t *denom0, const EntBuffers &eb, uint32_t *partial,
                       uint8_t *sym, uint32_t *info, int mbw, int mbh, bool code) {
    const hdr::Params a = make_hdr_params(o, is_inter, modes, f, d_sd, strength, probs, denom0, eb, mbw, mbh);
    const int nwg = (a.mbs + hdr::hdr_mb_per_wg(a.lpm_shift) - 1) / hdr::hdr_mb_per_wg(a.lpm_shift);
    hipLaunchKernelGGL(hdr::k_hdr_count, dim3(nwg), dim3(256), 0, s, a, eb.offs, partial);
    hipLaunchKernelGGL(hdr::k_hdr_frame, dim3(1), dim3(256), 0, s, a, partial, eb.offs, eb.bools, sym, eb.plan, info);
    hipLaunchKernelGGL(hdr::k_hdr_emit, dim3(nwg), dim3(256), 0, s, a, eb.offs, sym, eb.plan, info, eb.bools,
                       reinterpret_cast<unsigned long long *>(eb.acc));
    if (code) launch_bool_code(s, eb, 1);
}

static hdr::Params make_hdr_params_impl(const MBOut &o, const int32_t *is_inter, const int32_t *modes, const HdrFrame &f, const SegData *d_sd,
                                        const int32_t *strength, const uint32_t *probs, const uint32_t *denom0, const EntBuffers &eb, int mbw, int mbh) {
    hdr::Params a;
    a.v.seg_ = o.seg; a.v.nz_ = o.nz; a.v.ref_ = o.ref; a.v.parts_ = o.parts; a.v.is_inter_ = is_inter; a.v.modes_ = modes;
    a.v.vec_ = o.vec;
    a.v.mbw_ = mbw;
    a.mbs = mbw * mbh;
    a.key = f.is_key;
    a.is_golden = f.is_golden;
    a.is_altref = f.is_altref;
    a.loop_filter_type = f.loop_filter_type;
    a.sharpness = f.sharpness;
    a.partitions_log2 = f.partitions_log2;
    a.sd = d_sd;
    a.strength = strength;
    a.probs = probs;
    a.denom0 = denom0;
    a.cap_bools = eb.cap_bools;
    a.cap_chunks = eb.cap_chunks;
    a.cap_words = eb.cap_words;
    a.lpm_shift = hdr::HDR_LPM_SHIFT_ONE_VIDEO;
    return a;
}

}  // namespace vp8
