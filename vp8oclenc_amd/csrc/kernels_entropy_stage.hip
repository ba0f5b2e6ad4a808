// kernels_entropy_stage.hip -- the whole entropy stage of a frame (entropy_encode + gather_frame, src/vp8enc.cpp:48-94,
// src/encIO.h:1-30) in NINE launches instead of the twenty-one its steps take one by one.
//
// On this part a frame's entropy stage is bound by the number of kernels it takes, not by their work: every launch is a
// packet for the command processor and a link in a dependency chain ("Complete frames out").  So steps
// that do not depend on each other share a launch (the workgroup index picks the step), the prefix sums ride along in
// the kernels on either side of them, and the last kernel of the coder writes the finished frame:
//   1  k_fe_first   bools per macroblock header + census | bools per coefficient block + per-workgroup sums | block flags
//   2  k_ent_count  token histogram (needs every flag)
//   3  k_fe_mid     coefficient probabilities | scan of the workgroup sums + layout of the partitions
//   4  k_hdr_frame  frame header bools (needs the probabilities) + layout of the first partition
//   5  k_fe_emit    macroblock header bools | coefficient bools, each workgroup scanning its own 256 counts
//   6..9            the boolean coder on both strings at once (kernels_ent.hip), its last kernel writing the frame.
// Both halves live in this one translation unit for that; each still has its step-by-step launchers (vp8hip_count_probs,
// vp8hip_encode_coefficients, vp8hip_encode_header) on kernels of their own, which the parity tests compare with.
#include "kernels_ent.hip"
#include "kernels_hdr.hip"

namespace vp8 {
namespace fe {

struct FirstArgs {
    const int16_t *coeffs;
    const int32_t *nzc, *parts;
    uint8_t *flags;
    int nblocks;
    ent::Geom g;
    uint32_t *cnt, *tile_sum;
    hdr::Params h;
    uint32_t *hcnt, *hpartial;
    int nb_hdr, nb_slots;
};
__device__ __forceinline__ void fe_first_body(const FirstArgs &a, int vb) {   // the longest-running step first
    if (vb < a.nb_hdr) { hdr::hdr_count_body(vb, a.h, a.hcnt, a.hpartial); return; }
    vb -= a.nb_hdr;
    if (vb < a.nb_slots) { ent::boolcount_slots_body(vb, a.coeffs, a.nzc, a.parts, a.g, a.cnt, a.tile_sum); return; }
    ent::flags_body(vb - a.nb_slots, a.coeffs, a.flags, a.nblocks);
}
__global__ __launch_bounds__(256) void k_fe_first(FirstArgs a) { fe_first_body(a, blockIdx.x); }
__global__ __launch_bounds__(256) void k_fe_first_b(BatchOf<FirstArgs> b) { fe_first_body(b.item[blockIdx.z], blockIdx.x); }

struct MidArgs {
    const uint32_t *counts;
    uint32_t *probs, *denom0;
    int mbh, P;
    const uint8_t *defaults;
    const uint32_t *cnt;
    uint32_t *tile_sum;
    int ntiles;
    ent::Geom g;
    EntPlan *plan;
};
__device__ __forceinline__ void fe_mid_body(const MidArgs &a, int vb) {
    if (vb == 0) ent::scan_plan_body(a.cnt, a.tile_sum, a.ntiles, a.g, a.plan);
    else ent::probs_body(vb - 1, a.counts, a.probs, a.denom0, a.mbh, a.P, a.defaults);
}
__global__ __launch_bounds__(256) void k_fe_mid(MidArgs a) { fe_mid_body(a, blockIdx.x); }
__global__ __launch_bounds__(256) void k_fe_mid_b(BatchOf<MidArgs> b) { fe_mid_body(b.item[blockIdx.z], blockIdx.x); }

struct EmitArgs {
    const int16_t *coeffs;
    const int32_t *nzc, *parts;
    const uint8_t *third_ctx;
    const uint32_t *probs, *cnt, *tile_pre;
    const EntPlan *plan;
    ent::Geom g;
    uint16_t *bools;
    unsigned long long *acc;
    hdr::Params h;
    const uint32_t *hoffs, *hinfo;
    const uint8_t *hsym;
    const EntPlan *hplan;
    uint16_t *hbools;
    unsigned long long *hacc;
    int nb_hdr, nb_slots;
};
__device__ __forceinline__ void fe_emit_body(const EmitArgs &a, int vb) {
    if (vb < a.nb_hdr) hdr::hdr_emit_body(vb, a.nb_hdr, a.h, a.hoffs, a.hsym, a.hplan, a.hinfo, a.hbools, a.hacc);
    else ent::emit_slots_body(vb - a.nb_hdr, a.nb_slots, a.coeffs, a.nzc, a.parts, a.third_ctx, a.probs, a.cnt, a.tile_pre, a.plan, a.g,
                              a.bools, a.acc);
}
__global__ __launch_bounds__(256) void k_fe_emit(EmitArgs a) { fe_emit_body(a, blockIdx.x); }
__global__ __launch_bounds__(256) void k_fe_emit_b(BatchOf<EmitArgs> b) { fe_emit_body(b.item[blockIdx.z], blockIdx.x); }
static_assert(sizeof(BatchOf<FirstArgs>) <= 4096 && sizeof(BatchOf<EmitArgs>) <= 4096 && sizeof(BatchOf<hdr::FrameItem>) <= 4096 &&
              sizeof(ent::CodeJobs) <= 4096, "kernel arguments travel in the dispatch packet's 4 KiB segment");

}  // namespace fe

// lanes per macroblock of the header walks in a BATCH's launches (kernels_hdr.hip says why it differs from one video's): VP8HIP_HDR_BATCH_LPM = 1, 2, 4, 8, 16
static int hdr_batch_lpm_shift() {
    static const int sh = [] {
        const char *v = getenv("VP8HIP_HDR_BATCH_LPM");
        const int lpm = v && v[0] ? atoi(v) : 4;      // (same box, frames-out leg, M MB/s: 16 lanes 54.1-55.3, 8 56.3-56.8, 4 57.2-57.4, 2 55.6-57.7, 1 54.9-56.7)
        int s = 0;
        while ((1 << s) < lpm && s < 4) ++s;
        return s;
    }();
    return sh;
}

static fe::FirstArgs first_args(const FrameEntropy &e, int lpm_shift = hdr::HDR_LPM_SHIFT_ONE_VIDEO) {
    const EntBuffers &c = *e.coef, &h = *e.hdr;
    const int mbs = e.mbw * e.mbh, nblocks = mbs * 25;
    fe::FirstArgs a;
    a.coeffs = e.o.coeffs; a.nzc = e.o.nz; a.parts = e.o.parts;
    a.flags = e.flags;
    a.nblocks = nblocks;
    a.g = make_geom(c, e.mbw, e.mbh, e.P);
    a.cnt = c.offs; a.tile_sum = c.tile_sum;
    a.h = make_hdr_params(e.o, e.is_inter, e.modes, e.f, e.d_sd, e.strength, e.probs, e.denom0, h, e.mbw, e.mbh);
    a.hcnt = h.offs; a.hpartial = e.hdr_partial;
    a.h.lpm_shift = lpm_shift;
    a.nb_hdr = (mbs + hdr::hdr_mb_per_wg(lpm_shift) - 1) / hdr::hdr_mb_per_wg(lpm_shift);
    a.nb_slots = (nblocks + 255) / 256;
    return a;
}
static int first_grid(const fe::FirstArgs &a) { return a.nb_hdr + a.nb_slots + (a.nblocks + 255) / 256; }
static fe::MidArgs mid_args(const FrameEntropy &e) {
    const EntBuffers &c = *e.coef;
    fe::MidArgs m;
    m.counts = e.counts; m.probs = e.probs; m.denom0 = e.denom0;
    m.mbh = e.mbh; m.P = e.P;
    m.defaults = hdr_default_coeff_probs();   // the fallback for contexts that never occurred (vp8enc.cpp:69-76) rides along
    m.cnt = c.offs; m.tile_sum = c.tile_sum;
    m.ntiles = (e.mbw * e.mbh * 25 + 255) / 256;
    m.g = make_geom(c, e.mbw, e.mbh, e.P);
    m.plan = c.plan;
    return m;
}
static hdr::FrameItem hdr_frame_item(const FrameEntropy &e) {
    const EntBuffers &h = *e.hdr;
    hdr::FrameItem f;
    f.a = make_hdr_params(e.o, e.is_inter, e.modes, e.f, e.d_sd, e.strength, e.probs, e.denom0, h, e.mbw, e.mbh);
    f.partial = e.hdr_partial; f.cnt = h.offs; f.bools = h.bools; f.sym_out = e.hdr_sym; f.plan = h.plan; f.info = e.hdr_info;
    return f;
}
static fe::EmitArgs emit_args(const FrameEntropy &e, int lpm_shift = hdr::HDR_LPM_SHIFT_ONE_VIDEO) {
    const EntBuffers &c = *e.coef, &h = *e.hdr;
    const int mbs = e.mbw * e.mbh, nblocks = mbs * 25;
    fe::EmitArgs a;
    a.coeffs = e.o.coeffs; a.nzc = e.o.nz; a.parts = e.o.parts;
    a.third_ctx = e.third;
    a.probs = e.probs; a.cnt = c.offs; a.tile_pre = c.tile_sum;
    a.plan = c.plan;
    a.g = make_geom(c, e.mbw, e.mbh, e.P);
    a.bools = c.bools;
    a.acc = reinterpret_cast<unsigned long long *>(c.acc);
    a.h = make_hdr_params(e.o, e.is_inter, e.modes, e.f, e.d_sd, e.strength, e.probs, e.denom0, h, e.mbw, e.mbh);
    a.hoffs = h.offs; a.hinfo = e.hdr_info;
    a.hsym = e.hdr_sym;
    a.hplan = h.plan;
    a.hbools = h.bools;
    a.hacc = reinterpret_cast<unsigned long long *>(h.acc);
    a.h.lpm_shift = lpm_shift;
    a.nb_hdr = (mbs + hdr::hdr_mb_per_wg(lpm_shift) - 1) / hdr::hdr_mb_per_wg(lpm_shift);
    a.nb_slots = (nblocks + 255) / 256;
    return a;
}

// steps 1-3: everything up to the probabilities and the layout of the coefficient partitions
void launch_fe_count(hipStream_t s, const FrameEntropy &e) {
    const fe::FirstArgs a = first_args(e);
    hipLaunchKernelGGL(fe::k_fe_first, dim3(first_grid(a)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(ent::k_ent_count, dim3(e.mbh, ent::CNT_SPLIT), dim3(256), 0, s, e.o.coeffs, e.o.nz, e.o.parts, e.flags, e.third,
                       e.counts, e.mbw);
    const fe::MidArgs m = mid_args(e);
    hipLaunchKernelGGL(fe::k_fe_mid, dim3(1 + ent::NCTX / 16), dim3(256), 0, s, m);
    if (!m.defaults) launch_default_probs(s, e.probs, e.denom0);
}

// steps 4-5: the two bool strings
void launch_fe_emit(hipStream_t s, const FrameEntropy &e) {
    const hdr::FrameItem f = hdr_frame_item(e);
    hipLaunchKernelGGL(hdr::k_hdr_frame, dim3(1), dim3(256), 0, s, f.a, f.partial, f.cnt, f.bools, f.sym_out, f.plan, f.info);
    const fe::EmitArgs a = emit_args(e);
    hipLaunchKernelGGL(fe::k_fe_emit, dim3(a.nb_hdr + a.nb_slots), dim3(256), 0, s, a);
}

// the same for the frames of n contexts of one geometry (mbw, mbh and P of e[0] hold for all)
void launch_fe_count_batch(hipStream_t s, const FrameEntropy *e, int n) {
    BatchOf<fe::FirstArgs> a;
    BatchOf<ent::CountItem> c;
    BatchOf<fe::MidArgs> m;
    a.n = c.n = m.n = n;
    for (int i = 0; i < n; ++i) {
        a.item[i] = first_args(e[i], hdr_batch_lpm_shift());
        c.item[i] = ent::CountItem{e[i].o.coeffs, e[i].o.nz, e[i].o.parts, e[i].flags, e[i].third, e[i].counts, e[i].mbw};
        m.item[i] = mid_args(e[i]);
    }
    const unsigned skip = ent_skip_mask();
    if (!(skip & 1)) hipLaunchKernelGGL(fe::k_fe_first_b, dim3(first_grid(a.item[0]), 1, n), dim3(256), 0, s, a);
    if (!(skip & 2)) hipLaunchKernelGGL(ent::k_ent_count_b, dim3(e[0].mbh, ent::CNT_SPLIT, n), dim3(256), 0, s, c);
    if (!(skip & 4)) hipLaunchKernelGGL(fe::k_fe_mid_b, dim3(1 + ent::NCTX / 16, 1, n), dim3(256), 0, s, m);
    if (!m.item[0].defaults)
        for (int i = 0; i < n; ++i) launch_default_probs(s, e[i].probs, e[i].denom0);
}
void launch_fe_emit_batch(hipStream_t s, const FrameEntropy *e, int n) {
    BatchOf<hdr::FrameItem> f;
    BatchOf<fe::EmitArgs> a;
    f.n = a.n = n;
    for (int i = 0; i < n; ++i) {
        f.item[i] = hdr_frame_item(e[i]);
        a.item[i] = emit_args(e[i], hdr_batch_lpm_shift());
    }
    const unsigned skip = ent_skip_mask();
    if (!(skip & 8)) hipLaunchKernelGGL(hdr::k_hdr_frame_b, dim3(n), dim3(256), 0, s, f);
    if (!(skip & 16)) hipLaunchKernelGGL(fe::k_fe_emit_b, dim3(a.item[0].nb_hdr + a.item[0].nb_slots, 1, n), dim3(256), 0, s, a);
}

}  // namespace vp8
