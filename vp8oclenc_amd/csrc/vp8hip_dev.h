// vp8hip_dev.h -- shared declarations of the gfx950 inter-frame kernels (internal, not the ABI).
//
// Surfaces: every plane lives in HBM with a PAD-pixel margin on all four sides and a row
// stride that is a multiple of 64 B, so pixel (0,0) is 32-byte aligned, rows of an 8x8 block
// are one aligned 8-byte load, and a search window that hangs over the frame edge is an
// ordinary load (the margin of a reference plane holds the replicated edge = the reference's
// CLK_ADDRESS_CLAMP_TO_EDGE sampler, src/GPU_kernels.cl:562).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace vp8 {

// Per-kernel timing (vp8hip_profile_enable): the launchers of the single-kernel stages go through VP8_LAUNCH, which hands
// the dispatch a start and a stop event when the API layer has set them for this host thread.  Events recorded BY the
// dispatch carry the kernel's own begin / end timestamps (what rocprofv3's kernel trace shows); events recorded around it
// with hipEventRecord also count the time the packet waits for the queue to be scheduled, which with sixteen streams on the
// part is as long again as the kernel (loop filter: 0.70 ms bracketed vs 0.39 ms in the trace).
struct LaunchTiming { hipEvent_t start = nullptr, stop = nullptr; int launches = 0; };
extern thread_local LaunchTiming tl_timing;
#define VP8_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                                               \
    do {                                                                                                                                  \
        ++::vp8::tl_timing.launches;                                                                                                      \
        hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ::vp8::tl_timing.start, ::vp8::tl_timing.stop, 0, __VA_ARGS__);         \
    } while (0)

constexpr int PAD = 32;        // allocated margin (pixels) around every plane
constexpr int EXT = 8;         // replicated-edge width actually filled (max reach of any filter: 3)
constexpr int LF_ERR_WORD = 1536;   // int index inside the loop filter's progress buffer of its time-out flag
constexpr int S2_CLOCK_WORD = LF_ERR_WORD + 32;   // ... and of k_search2's launch clock (five 64-bit words, see launch_clock_end)
constexpr int CLOCK_SAMPLE = 64;    // every 64th workgroup of a launch stamps the clock
constexpr int SD_INTS = 11;    // ints per segment_data, src/vp8enc.h:80-92
enum { SD_Y_AC_I = 0, SD_Y_DC_IDELTA, SD_Y2_DC_IDELTA, SD_Y2_AC_IDELTA, SD_UV_DC_IDELTA, SD_UV_AC_IDELTA,
       SD_LOOP_FILTER_LEVEL, SD_MBEDGE_LIMIT, SD_SUB_BEDGE_LIMIT, SD_INTERIOR_LIMIT, SD_HEV_THRESHOLD };

struct Plane {
    uint8_t *p;   // address of pixel (0,0)
    int stride;   // bytes per row
    int w, h;
};

struct Frame {
    Plane Y[5];   // Y[l] = luma downsampled by 2^l
    Plane U, V;
};

struct RefSet {            // what one search/predict launch needs to know about the references
    Frame ref[3];          // LAST, GOLDEN, ALTREF
    int use[3];
};

// per-reference motion state: two short2 nets (ping-pong, src/init.h:672-854) and the block costs
struct NetSet {
    int16_t *net[3][2];
    int32_t *bdiff[3];
};

struct MBOut {
    int32_t *parts, *ref, *seg, *nz, *mask;
    int16_t *vec;      // [MBs][4][2]
    int16_t *coeffs;   // [MBs][25][16]
    float *ssim;
    int32_t *flags;    // [0]: set by k_mb when a macroblock's SSIM is below the target, i.e. check_SSIM's fallback has work to do;
                       //      cleared by the loop filter launch that carries the verdict (zero at rest)
};

struct SegData { int32_t v[4 * SD_INTS]; };

// ---- batched launches -------------------------------------------------------------------------------------------------
// Measured on MI355X (scripts/ubench/concurrency.hip, and the kernel trace of sixteen GOP chunks on sixteen streams): the
// part runs FOUR TO FIVE kernels at once however many streams offer work, and with more than ~8 active streams the
// hardware queues are time-sliced.  Sixteen chunks each launching its own narrow kernels therefore leave the chip with
// one wide kernel at a time (loop filters of 9 workgroups hold two of the slots on average).  So the same kernels also
// exist in a batched form: ONE launch does a stage for up to MAX_BATCH contexts of equal geometry (blockIdx.z = context),
// the argument blocks of the single form travel as an array in the kernel arguments, and a few streams carry what sixteen did.
constexpr int MAX_BATCH = 8;
template <typename A> struct BatchOf { int n; A item[MAX_BATCH]; };

// A launch's duration by the kernel's own clock (s_memrealtime, 100 MHz), for kernels of many workgroups: every CLOCK_SAMPLE-th
// workgroup (in dispatch order, the first one included) stamps w = {earliest start, latest end, sampled workgroups done,
// sum over launches of (end - start), launches}; the last sampled workgroup to finish closes the launch.  HIP events around a
// launch also count the time its packet waits for its hardware queue; this does not, and it is what a rocprofv3 kernel trace
// shows.  w[0] is ~0 at rest.  (Stamping every workgroup would put 150 000 atomics per launch on one cache line.)
#if defined(__HIPCC__)
__device__ __forceinline__ bool launch_clock_sampled() {
    const unsigned wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    return wg % CLOCK_SAMPLE == 0;
}
__device__ __forceinline__ void launch_clock_begin(unsigned long long *w) {
    // A SCALAR condition (the first wave of a sampled workgroup; the compiler folds the wave's atomics into one): with the per-lane
    // "threadIdx.x == 0" here, hipcc carried the workgroup indices through the branch in VGPRs and k_search2 picked its reference's
    // planes and nets with vector loads and 64-bit vector address arithmetic -- 49 of its 771 vector instructions.
    if (w && launch_clock_sampled() && __builtin_amdgcn_readfirstlane((int)threadIdx.x) == 0)
        atomicMin(&w[0], (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
__device__ __forceinline__ void launch_clock_end(unsigned long long *w) {   // every thread of the workgroup gets here
    if (!w || !launch_clock_sampled()) return;
    __syncthreads();
    if (threadIdx.x != 0) return;
    atomicMax(&w[1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    __threadfence();
    const unsigned total = gridDim.x * gridDim.y * gridDim.z, sampled = (total + CLOCK_SAMPLE - 1) / CLOCK_SAMPLE;
    if (atomicAdd(&w[2], 1ull) + 1 == sampled) {
        __threadfence();
        const unsigned long long t1 = atomicExch(&w[1], 0ull), t0 = atomicExch(&w[0], ~0ull);
        atomicExch(&w[2], 0ull);
        atomicAdd(&w[3], t1 - t0);
        atomicAdd(&w[4], 1ull);
    }
}
#endif
// Switches that LEAVE WORK OUT of a launch or a wait (the results are then garbage) exist for same-box timing runs only: they
// are compiled in with -DVP8HIP_EXPERIMENTS (scripts/ab_*.sh build that way) and are constants otherwise, so no environment
// can take work out of a run of the shipped library.  vp8hip_experiments_compiled_in() tells a caller which build it has.
#ifdef VP8HIP_EXPERIMENTS
inline bool experiment_skip(const char *what) { const char *v = getenv("VP8HIP_EXPERIMENT_SKIP"); return v && strstr(v, what) != nullptr; }
inline const char *experiment_env(const char *name) { return getenv(name); }
#else
inline bool experiment_skip(const char *) { return false; }
inline const char *experiment_env(const char *) { return nullptr; }
#endif
// workgroups of a persistent launch (0 = launch the full grid); VP8HIP_PERSIST overrides (same-box A/B runs)
int persistent_workgroups();

// ---- launchers (kernels_*.hip) ---------------------------------------------------------------
void launch_border(hipStream_t s, const Frame &f);
// all four levels of one or two frames; bit z of border_mask: also the replicated edges of surface z (launch_border's work, same launch)
void launch_pyramid(hipStream_t s, const Frame *a, const Frame *b, uint32_t border_mask = 0);
// sw, sh: luma size of the source planes (0 = the coded size); ssy, ssc: their row strides (0 = tight)
void launch_pack(hipStream_t s, const Frame &f, const void *y, const void *u, const void *v, int sw = 0, int sh = 0, int ssy = 0, int ssc = 0);
// batched forms: n <= MAX_BATCH contexts (pyramid: nframes <= 2 * MAX_BATCH surfaces)
void launch_pyramid_batch(hipStream_t s, const Frame *const *f, int nframes, uint32_t border_mask = 0);
// launch_auto_segments' work (same arguments) for a launch that lets it ride along (launch_search2_batch)
struct ScanRequest { uint32_t *partial, *stats; SegData *sd; int32_t *strength_out; int is_key; int32_t refqi[4]; int qi_min; };
void launch_pack_batch(hipStream_t s, const Frame *const *f, const void *const *y, const void *const *u, const void *const *v, int n,
                       int sw = 0, int sh = 0);
bool launch_search1_coarse_batch(hipStream_t s, const Frame *const *cur, const RefSet *refs, const NetSet *const *nets, int net_width, int n, bool finest, bool top_only = false);
void launch_search1_batch(hipStream_t s, const Frame *const *cur, const RefSet *refs, const NetSet *const *nets, int level, int src_idx,
                          int net_width, int n);
// scan[i] (may be nullptr, as may scan): member i's new frame gets its loop-filter strength scan + segment data in this launch.
// Returns false if the scans were NOT carried (nothing to search, or the persistent form): the caller launches them on their own.
bool launch_search2_batch(hipStream_t s, const Frame *const *cur, const RefSet *refs, const NetSet *const *nets, int n,
                          unsigned long long *clk = nullptr, const ScanRequest *const *scan = nullptr);   // clk: launch clock words (launch_clock_end) or nullptr
void launch_mb_batch(hipStream_t s, const Frame *const *cur, const RefSet *refs, const NetSet *const *nets, const Frame *const *recon,
                     const MBOut *const *o, const SegData *const *d_sd, float ssim_target, int mbw, int mbh, int n,
                     bool conformant = false);
struct LfCheck;
void launch_loop_filter3_batch(hipStream_t s, const Frame *const *recon, const MBOut *const *o, SegData *const *d_sd,
                               int32_t *const *progress, int mbw, int mbh, const unsigned *launch_no, int n, const LfCheck *chk = nullptr);
void launch_loop_filter4_batch(hipStream_t s, const Frame *const *recon, const MBOut *const *o, SegData *const *d_sd,
                               int32_t *const *progress, void *const *handoff, int mbw, int mbh, const unsigned *launch_no, int n,
                               const LfCheck *chk = nullptr);
void launch_auto_segments_batch(hipStream_t s, const Frame *const *cur, uint32_t *const *partial, uint32_t *const *stats, SegData *const *sd,
                                int32_t *const *strength_out, const int *is_key, const int32_t (*refqi)[4], int qi_min, int n);
// levels 4, 3, 2, 1 (and, finest, 0) in ONE launch (a single video: kernels_me.hip, k_search1_coarse); leaves the level-1 / level-0 nets where the per-level launches leave them
void launch_search1_coarse(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, int net_width, bool finest);
void launch_search1(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, int level,
                    int src_idx, int net_width, bool latency = false);   // latency: the short-wave mapping whatever the size
void launch_search2(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, unsigned long long *clk = nullptr);
void launch_weight_tap(hipStream_t s, const int32_t *d, int n, int32_t *out);
void launch_mb(hipStream_t s, const Frame &cur, const RefSet &refs, const NetSet &nets, const Frame &recon,
               const MBOut &o, const SegData *d_sd, float ssim_target, int mbw, int mbh, bool conformant = false);
void launch_filter_mask(hipStream_t s, const MBOut &o, const SegData *d_sd, int mbs);
// The loop filter exists in two forms with the same results.  Form 3: byte tiles and strips in 46 KB of LDS,
// the worker wave does everything itself -- what batches of GOP chunks launch (with the part full what counts is the
// instructions and the LDS a launch takes from the other kernels, not its latency).  Form 4: a band-tall plane of dwords in
// 112 KB of LDS, the workers only filter, porter waves feed and drain -- the short dependency chain of ONE video.
void launch_loop_filter3(hipStream_t s, const Frame &recon, const MBOut &o, SegData *d_sd, int32_t *progress,
                         int mbw, int mbh, unsigned launch_no, int stall_test = 0, const LfCheck *chk = nullptr);
void launch_loop_filter4(hipStream_t s, const Frame &recon, const MBOut &o, SegData *d_sd, int32_t *progress, void *handoff,
                         int mbw, int mbh, unsigned launch_no, int stall_test = 0, const LfCheck *chk = nullptr);
size_t loop_filter4_handoff_bytes(int mbw, int mbh);   // the HBM buffer form 4 hands a band's bottom rows to the next band through

// per-frame parameter scans on the device copy of the current frame (kernels_rc.hip); stats = 4 uint32
size_t rc_partial_words();   // uint32 words of per-workgroup partial sums the three launchers below need
void launch_lf_strength(hipStream_t s, const Frame &cur, uint32_t *partial, uint32_t *stats);    // [0] sum Y, [1] sum of squared deviations
void launch_chroma_sad(hipStream_t s, const Frame &cur, const Frame &prev, uint32_t *partial, uint32_t *stats);   // [2] sum |dU|, [3] sum |dV|
void launch_auto_segments(hipStream_t s, const Frame &cur, uint32_t *partial, uint32_t *stats, SegData *sd, int32_t *strength_out,
                          int is_key, const int32_t refqi[4], int qi_min);   // strength + prepare_segments_data, all on the device

// coefficient entropy stage (kernels_ent.hip): flags + third context + token histogram + probabilities
constexpr int ENT_NCTX = 4 * 8 * 3 * 11;
constexpr int ENT_MAX_PARTITIONS = 8;
#if !defined(VP8HIP_ENT_CHUNK)
#define VP8HIP_ENT_CHUNK 256
#endif
constexpr int ENT_CHUNK = VP8HIP_ENT_CHUNK;   // bools per chunk of the parallel boolean coder (kernels_ent.hip)
void launch_ent_count(hipStream_t s, const MBOut &o, uint8_t *flags, uint8_t *third_ctx, uint32_t *counts, uint32_t *probs,
                      uint32_t *denom0, int mbw, int mbh, int num_partitions, const uint8_t *defaults = nullptr);   // defaults: see hdr_default_coeff_probs
// layout of one frame's bool strings / chunks / output words per partition; written by the device, read by the host
struct EntPlan {
    uint32_t bool_base[ENT_MAX_PARTITIONS + 1], chunk_base[ENT_MAX_PARTITIONS + 1], word_base[ENT_MAX_PARTITIONS + 1];
    uint32_t nbools[ENT_MAX_PARTITIONS], w_end[ENT_MAX_PARTITIONS], nbytes[ENT_MAX_PARTITIONS];
    uint32_t total_chunks, overflow;
};
struct EntBuffers {          // device scratch of the boolean coder (allocated on first use)
    uint32_t *offs, *tile_sum;      // bools per block slot -> exclusive offsets; scan tile sums
    uint16_t *bools;                // (probability | bit << 8) per bool, partition after partition
    uint32_t *maps;                 // [chunk][128 start ranges] -> end range | shifts << 8
    void *start;                    // uint2 [chunk]: true start range and bit position
    void *acc;                      // uint64 per 32 output bits
    uint8_t *bytes;                 // the partitions, at 4 * word_base[p]
    int32_t *sizes;
    EntPlan *plan;
    uint32_t cap_bools, cap_chunks, cap_words;
};
void launch_ent_encode(hipStream_t s, const MBOut &o, const uint8_t *third_ctx, const uint32_t *probs, const EntBuffers &eb,
                       int mbw, int mbh, int P, bool code = true);   // code = false: bool strings only, the coder is launched by the caller

void launch_scan_exclusive(hipStream_t s, uint32_t *v, uint32_t *tile_sum, int n);   // in place, total in v[n]
// the coder on bool strings laid out per eb.plan (accumulators cleared by the emit kernel); eb.maps holds
// cap_chunks chunk maps followed by the super-chunk maps (ent_maps_entries)
void launch_bool_code(hipStream_t s, const EntBuffers &eb, int P);
// the coder on the coefficient partitions and the first partition at once, its last kernel laying the finished frame
// out (gather_frame): frame[0] = frame size (0 = overflow), frame[1] = first-partition size, frame bytes from frame + 16
void launch_frame_code(hipStream_t s, const EntBuffers &coef, int P, const EntBuffers &hdr, uint32_t head, uint32_t capacity, uint8_t *frame);
inline size_t ent_maps_entries(uint32_t cap_chunks) { return ((size_t)cap_chunks + cap_chunks / 8 + 2 * ENT_MAX_PARTITIONS) * 128; }

// first partition on the device (kernels_hdr.hip): encode_header, src/entropy_host.cpp:709-1256
struct HdrFrame { int is_key, is_golden, is_altref, loop_filter_type, sharpness /* INT32_MIN: the device's */, partitions_log2; };
void launch_default_probs(hipStream_t s, uint32_t *probs, const uint32_t *denom0);   // vp8enc.cpp:69-76
const uint8_t *hdr_default_coeff_probs();   // device address of the default coefficient probabilities [4][8][3][11] (RFC 6386 13.5)
constexpr int HDR_STAT_WORDS = 84;   // per-workgroup partial sums of k_hdr_count
void launch_hdr_encode(hipStream_t s, const MBOut &o, const int32_t *is_inter, const int32_t *modes, const HdrFrame &f, const SegData *d_sd,
                       const int32_t *strength, const uint32_t *probs, const uint32_t *denom0, const EntBuffers &eb, uint32_t *partial,
                       uint8_t *sym, uint32_t *info, int mbw, int mbh, bool code = true);

// the frame path of the entropy stage (kernels_entropy_stage.hip): the same steps sharing launches
struct FrameEntropy {
    MBOut o;
    uint8_t *flags, *third;
    uint32_t *counts, *probs, *denom0;
    const EntBuffers *coef, *hdr;    // coef->offs receives bools per slot, coef->tile_sum the sums per 256 slots
    uint32_t *hdr_partial, *hdr_info;
    uint8_t *hdr_sym;
    const int32_t *is_inter, *modes;
    HdrFrame f;
    const SegData *d_sd;
    const int32_t *strength;
    int mbw, mbh, P;
};
void launch_fe_count(hipStream_t s, const FrameEntropy &e);   // counts, probabilities, layout of the coefficient partitions
void launch_fe_emit(hipStream_t s, const FrameEntropy &e);    // frame header + both bool strings; then launch_frame_code
// the frames of up to MAX_BATCH contexts of one geometry in the same nine launches (blockIdx.z = member; the coder takes
// the job pairs 2m, 2m + 1)
struct FrameOut { uint8_t *frame; uint32_t head, capacity; };
void launch_fe_count_batch(hipStream_t s, const FrameEntropy *e, int n);
void launch_fe_emit_batch(hipStream_t s, const FrameEntropy *e, int n);
void launch_frame_code_batch(hipStream_t s, const FrameEntropy *e, const FrameOut *fo, int n);

// host intra path on the device (kernels_intra.hip): key frames (key = 1) and check_SSIM's intra fallback (key = 0).
// prog: mbh ints (row progress), zeroed by the launcher; err: time-out flag; stats out: {replaced, new_SSIM, min SSIM, time-out flag}
// gen: the launch number on this progress buffer (the rows' counters carry it, so nothing has to be cleared between launches)
void launch_intra(hipStream_t s, const Frame &cur, const Frame &recon, const MBOut &o, const SegData *d_sd, int32_t *modes,
                  int32_t *is_inter, int32_t *prog, unsigned gen, int32_t *err, float target, int key, int mbw, int mbh, int stall_test = 0,
                  int modes_of_kept = 0);   // modes_of_kept: vp8hip_conformant_stream
void launch_ssim_stats(hipStream_t s, const MBOut &o, const int32_t *is_inter, int mbs, const int32_t *err, int32_t *out);   // out[3] = *err
// check_SSIM() (src/vp8enc.cpp:231-263) for up to MAX_BATCH contexts without a host round trip and without a launch that is
// not there anyway.  The intra fallback is a launch of its own whose workgroups leave at once unless k_mb has flagged a
// macroblock below the target (o.flags[0]); it also renews nz / mask of the macroblocks it replaces.  The statistics and what
// the host does with them ride in the loop filter's launch (LfCheck): every band takes the minimum SSIM itself and, above
// 0.95, filters with the segment data prepare_segments_data(1, 7) gives; one extra workgroup sums the frame's SSIM in the
// reference's order, writes the updated segment data back for the entropy stage (strength[2], the sharpness the frame header
// carries, becomes 7), clears the flag and hands {replaced, new_SSIM, min SSIM, time-out flag, filter updated, seq} to the
// host through memory the host polls -- the verdict is there long before the filter has finished.
struct CheckItem {
    const Frame *cur, *recon;
    const MBOut *o;
    const SegData *sd;
    int32_t *modes, *is_inter, *prog, *err;
    unsigned gen;
};
struct LfCheck {
    int on, qi_min;
    int32_t refqi[4];
    const int32_t *is_inter;
    int32_t *strength;       // {reductor, sharpness, sharpness in force}
    int32_t *stats;          // device copy of the verdict (vp8hip_check_ssim's read-back buffer)
    int32_t *verdict;        // host memory the device writes: the five words, then seq
    uint32_t seq;
};
void launch_check_fallback(hipStream_t s, const CheckItem *items, int n, float target, int mbw, int mbh, int modes_of_kept);
void launch_intra_key_batch(hipStream_t s, const CheckItem *items, int n, int mbw, int mbh);   // the key frames of n members of a batch, one launch

// ---- device helpers ---------------------------------------------------------------------------
#if defined(__HIPCC__)
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int byte_of(uint32_t w, int k) { return (int)((w >> (8 * k)) & 0xffu); }

// The workgroups of a launch go to the part's eight XCDs round-robin by their linear index, and every XCD has an L2 of its own.  A
// kernel whose workgroup x works on tile x therefore spreads every neighbourhood of the frame over all eight L2s: the reference rows
// two vertically adjacent search windows share, the 128-byte line two horizontally adjacent tiles share, are fetched once per XCD
// that meets them (k_search2: 3.5 x its algorithmic bytes, profiles/pmc_traffic.json of round 3).  With this map the workgroups an
// XCD is given (x = k mod 8) work on ONE contiguous run of tiles -- a band of the frame -- so a line is fetched by the one L2 whose
// band it lies in (and by a neighbour's at a band's edge).  A bijection of [0, n) for any n.
__device__ __forceinline__ int xcd_band(int x, int n) {
#if defined(VP8HIP_NO_XCD_BANDS)
    (void)n;
    return x;
#else
    const int k = x & 7, chunk = n >> 3, rem = n & 7;
    return k * chunk + (k < rem ? k : rem) + (x >> 3);
#endif
}

// VP8 quantiser index -> step tables (RFC 6386 14.1; the reference keeps them at GPU_kernels.cl:58-80)
static __device__ __constant__ const int k_dc_q[128] = {
    4,   5,   6,   7,   8,   9,   10,  10,  11,  12,  13,  14,  15,  16,  17,  17,  18,  19,  20,  20,  21,  21,
    22,  22,  23,  23,  24,  25,  25,  26,  27,  28,  29,  30,  31,  32,  33,  34,  35,  36,  37,  37,  38,  39,
    40,  41,  42,  43,  44,  45,  46,  46,  47,  48,  49,  50,  51,  52,  53,  54,  55,  56,  57,  58,  59,  60,
    61,  62,  63,  64,  65,  66,  67,  68,  69,  70,  71,  72,  73,  74,  75,  76,  76,  77,  78,  79,  80,  81,
    82,  83,  84,  85,  86,  87,  88,  89,  91,  93,  95,  96,  98,  100, 101, 102, 104, 106, 108, 110, 112, 114,
    116, 118, 122, 124, 126, 128, 130, 132, 134, 136, 138, 140, 143, 145, 148, 151, 154, 157};
static __device__ __constant__ const int k_ac_q[128] = {
    4,   5,   6,   7,   8,   9,   10,  11,  12,  13,  14,  15,  16,  17,  18,  19,  20,  21,  22,  23,  24,  25,
    26,  27,  28,  29,  30,  31,  32,  33,  34,  35,  36,  37,  38,  39,  40,  41,  42,  43,  44,  45,  46,  47,
    48,  49,  50,  51,  52,  53,  54,  55,  56,  57,  58,  60,  62,  64,  66,  68,  70,  72,  74,  76,  78,  80,
    82,  84,  86,  88,  90,  92,  94,  96,  98,  100, 102, 104, 106, 108, 110, 112, 114, 116, 119, 122, 125, 128,
    131, 134, 137, 140, 143, 146, 149, 152, 155, 158, 161, 164, 167, 170, 173, 177, 181, 185, 189, 193, 197, 201,
    205, 209, 213, 217, 221, 225, 229, 234, 239, 245, 249, 254, 259, 264, 269, 274, 279, 284};
__device__ __forceinline__ int qi(int v) { return iclamp(v, 0, 127); }

// prepare_segments_data(), src/vp8enc.cpp:129-221, from the strength pair of get_loopfilter_strength (:96-127).
// update_filter: check_SSIM's call prepare_segments_data(1, 7) (:155-159, :260-261): reductor doubled, sharpness 7.
// Returns video.loop_filter_sharpness as it stands afterwards (what the frame header carries).  One thread.
__device__ __forceinline__ int fill_segment_data(SegData *sd, int is_key, const int refqi[4], int qi_min, int reductor, int sharpness,
                                                 bool update_filter) {
    if (update_filter) { reductor *= 2; sharpness = 7; }
    int32_t *v = sd->v;
    for (int i = 0; i < 4 * SD_INTS; ++i) v[i] = 0;
    v[SD_Y_DC_IDELTA] = 15;                         // segment 0 carries the deltas, :133-148
    v[SD_UV_DC_IDELTA] = is_key ? 0 : -15;
    v[SD_UV_AC_IDELTA] = is_key ? 0 : -15;
    for (int i = 0; i < 4; ++i) {
        int32_t *s = v + SD_INTS * i;
        s[SD_Y_AC_I] = is_key ? qi_min : refqi[i];  // :164
        const int y_dc_q = k_dc_q[qi(s[SD_Y_AC_I] + v[SD_Y_DC_IDELTA])];
        int lvl = y_dc_q / reductor;                // :187-189
        lvl = lvl > 63 ? 63 : (lvl < 0 ? 0 : lvl);
        s[SD_LOOP_FILTER_LEVEL] = lvl;
        int il = lvl;                               // :192-199
        if (sharpness) {
            il >>= sharpness > 4 ? 2 : 1;
            if (il > 9 - sharpness) il = 9 - sharpness;
        }
        if (!il) il = 1;
        s[SD_INTERIOR_LIMIT] = il;
        s[SD_MBEDGE_LIMIT] = ((lvl + 2) * 2) + il;
        s[SD_SUB_BEDGE_LIMIT] = (lvl * 2) + il;
        int hev = 0;                                // :204-220
        if (is_key) hev = lvl >= 40 ? 2 : (lvl >= 15 ? 1 : 0);
        else hev = lvl >= 40 ? 3 : (lvl >= 20 ? 2 : (lvl >= 15 ? 1 : 0));
        s[SD_HEV_THRESHOLD] = hev;
    }
    return sharpness;
}

__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) {  // byte-aligned dword load
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint2 ld_u64(const uint8_t *p) {     // byte-aligned 8-byte load
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// The block-match metric (src/GPU_kernels.cl:85-190, weight_opt): sum of |coefficients| of a 4x4
// forward transform of the difference block, DC/4.  Column pass keeps the reference's quirk (its b1
// is overwritten; rows 1 and 3 use the raw r2).  d = 4 rows x 4 columns, row-major.
// The two rotations x*2217 + y*5352 + k and y*2217 - x*5352 + k are one v_dot2_i32_i16 each on the
// packed pair (x,y): every operand fits int16 (|x|,|y| <= 16320 in the row pass, <= 4080 in the
// column pass for 8-bit pixel differences).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk16(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }
__device__ __forceinline__ int dot2(uint32_t xy, uint32_t k, int c) {
    // clamp (int32 saturation, never reached here) selects the three-address VOP3P form that takes the
    // rounding constant straight from an SGPR; without it hipcc emits v_dot2c + a v_mov per use
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, xy), __builtin_bit_cast(s16x2, k), c, true);
}
constexpr uint32_t K_ROT_A = 2217u | (5352u << 16);              // (x, y) . ( 2217, 5352)
constexpr uint32_t K_ROT_B = (uint32_t)(-5352 & 0xffff) | (2217u << 16);  // (x, y) . (-5352, 2217)

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
// sum of |lo| + |hi| of a packed pair of int16, added to acc: one xor + one v_sad_u16
__device__ __forceinline__ uint32_t abs_acc(s16x2 v, uint32_t acc) {
    return __builtin_amdgcn_sad_u16(as_u32(v) ^ 0x80008000u, 0x80008000u, acc);
}

// ---- the form the search kernels use --------------------------------------------------------------------------
// Issue cost on gfx950 (scripts/ubench/valu_cost.hip, profiles/valu_cost.json): only plain VOP1/VOP2 add / sub / logic /
// right shifts issue in ~2.3 SIMD cycles; every SDWA, VOP3 and VOP3P instruction (perm, sad, dot2, dot4, v_pk_*,
// even v_lshlrev_b32) costs ~4.2.  So the metric is organised to need the FEWEST instructions, whatever their kind:
//   * a 4x4 block arrives as four COLUMN dwords (byte r = row r), pixels biased by -128 (signed bytes);
//   * the column pass is linear in the pixels except for two rounding shifts, and the current block is the same for
//     every candidate: its share is computed once per column (weight_pre_column: four dot4) and each candidate only adds
//     the prediction's share with ONE v_dot4_i32_i8 per quantity, the accumulator input carrying the current block's:
//         R0 = 8(r0+r1-r2+r3), R2 = 8(r0-r1+r2+r3), X = 16 r2, Y = 32 (r0-r3)        (r = current - prediction)
//   * the two rotations take (X, Y) as one packed pair: scaled by 16 so that ">> 12" becomes "the high half":
//         R1 = (2217 r2 + 5352*8 (r0-r3) + 14500) >> 12 = (2217 X + 21408 Y + 16*14500) >> 16,
//         R3 = (2217*8 (r0-r3) - 5352 r2 + 7500) >> 12 = (8868 Y - 5352 X + 16*7500) >> 16
//     (|X| <= 4080, |Y| <= 8160, sums < 2^28: exact);
//   * row pass on PAIRS of rows in packed 16-bit (every intermediate fits int16 for 8-bit pixel differences:
//     |R| <= 8160, |a1|,|b1| <= 16320, |a1 +- b1 + 7| <= 32647), |a|+|b| of a pair = xor + one v_sad_u16.
// 9 instructions per column instead of 4 byte subtractions + 13: the device metric is pinned on 200k random and on the
// extreme difference blocks by test_block_match_metric_device_vs_oracle through k_weight_tap.
constexpr uint32_t pk8s(int a, int b, int c, int d) {
    return (uint32_t)(a & 255) | ((uint32_t)(b & 255) << 8) | ((uint32_t)(c & 255) << 16) | ((uint32_t)(d & 255) << 24);
}
constexpr uint32_t K_W_R0 = pk8s(8, 8, -8, 8), K_W_R2 = pk8s(8, -8, 8, 8), K_W_X = pk8s(0, 0, 16, 0), K_W_Y = pk8s(32, 0, 0, -32);
constexpr uint32_t K_W_R0N = pk8s(-8, -8, 8, -8), K_W_R2N = pk8s(-8, 8, -8, -8), K_W_XN = pk8s(0, 0, -16, 0), K_W_YN = pk8s(-32, 0, 0, 32);
constexpr uint32_t K_ROT16_A = 2217u | (21408u << 16);                       // (X, Y) . (2217, 21408)
constexpr uint32_t K_ROT16_B = (uint32_t)(-5352 & 0xffff) | (8868u << 16);   // (X, Y) . (-5352, 8868)
// clamp (int32 saturation, never reached: |sums| < 2^15) selects the three-address VOP3P form; without it hipcc emits the
// two-address v_dot4c plus a v_mov per use to keep the accumulator input alive
__device__ __forceinline__ int dot4s(uint32_t a, uint32_t k, int c) { return __builtin_amdgcn_sdot4((int)a, (int)k, c, true); }

// the current block's share of one column (ccol = its four rows as biased bytes): {R0, R2, X, Y} parts
__device__ __forceinline__ void weight_pre_column(uint32_t ccol, int pre[4]) {
    pre[0] = dot4s(ccol, K_W_R0, 0);
    pre[1] = dot4s(ccol, K_W_R2, 0);
    pre[2] = dot4s(ccol, K_W_X, 0);
    pre[3] = dot4s(ccol, K_W_Y, 0);
}

// Row pass of the metric on the column pass's output: A[c] = (R0[c], R1[c]) = rows 0,1 of column c, B[c] = rows 2,3.
// |a| + |b| of a packed pair is one v_sad_u16 against a constant once the pair carries a bias that makes it
// unsigned; the biases ride on work that is done anyway: +0x8000 joins the rounding 7 of a1 (then ">> 4" is a LOGICAL
// shift and the pair comes out with +2048: floor((x + 32768) / 16) = floor(x / 16) + 2048), and +2^28 joins the rounding
// constants of the rotations (the high half comes out with +4096; |sums| < 2^28, so nothing saturates).
__device__ __forceinline__ int weight_rows(const s16x2 A[4], const s16x2 B[4]) {
    uint32_t acc = 0;
    int o00 = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const s16x2 *X = h == 0 ? A : B;
        const s16x2 a1 = X[0] + X[3], d1 = X[0] - X[3], b1 = X[1] + X[2], c1 = X[1] - X[2];
        const u16x2 a7 = __builtin_bit_cast(u16x2, a1) + u16x2{0x8007, 0x8007};
        const u16x2 o0 = (a7 + __builtin_bit_cast(u16x2, b1)) >> u16x2{4, 4};
        const u16x2 o2 = (a7 - __builtin_bit_cast(u16x2, b1)) >> u16x2{4, 4};
        const uint32_t xy_lo = __builtin_amdgcn_perm(as_u32(d1), as_u32(c1), 0x05040100u);   // (c1, d1) of the first row
        const uint32_t xy_hi = __builtin_amdgcn_perm(as_u32(d1), as_u32(c1), 0x07060302u);   // ... of the second row
        const int t1l = dot2(xy_lo, K_ROT_A, 12000 + (1 << 28)), t1h = dot2(xy_hi, K_ROT_A, 12000 + (1 << 28));
        const int t3l = dot2(xy_lo, K_ROT_B, 51000 + (1 << 28)), t3h = dot2(xy_hi, K_ROT_B, 51000 + (1 << 28));
        // (x >> 16) of both rows = the high halves, packed (+4096 each)
        const uint32_t o1 = __builtin_amdgcn_perm((uint32_t)t1h, (uint32_t)t1l, 0x07060302u);
        const uint32_t o3 = __builtin_amdgcn_perm((uint32_t)t3h, (uint32_t)t3l, 0x07060302u);
        // |o1 + (d1 != 0)| per half: the +1 rides in the subtrahend of the absolute difference.
        // d1 != 0, per half: min(d1 as unsigned, 1), one v_pk_min_u16 (was sub / or / shift / and on the pair).  The 1s pass
        // through an empty asm: min(x, 1) with a constant hipcc can see becomes x != 0, which it scalarises into two v_cmp and
        // two v_cndmask per pair.
        uint32_t ones = 0x00010001u;
        asm("" : "+s"(ones));
        const uint32_t nz = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, d1), __builtin_bit_cast(u16x2, ones)));
        acc = __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, o0), 0x08000800u, acc);
        acc = __builtin_amdgcn_sad_u16(o1, 0x10001000u - nz, acc);
        acc = __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, o2), 0x08000800u, acc);
        acc = __builtin_amdgcn_sad_u16(o3, 0x10001000u, acc);
        if (h == 0) o00 = (int)(__builtin_bit_cast(uint32_t, o0) & 0xffffu);   // the DC, still carrying its +2048
    }
    const int a00 = (int)__builtin_amdgcn_sad_u16((uint32_t)o00, 2048u, 0u);   // |DC| (the high halves are both zero)
    return (int)acc - (a00 - (a00 >> 2));   // DC counts a quarter (DC_UNSIGNIFICANCE, :83,:183)
}

// weight of (current - prediction) for one 4x4 block: pre = weight_pre_column of the four current columns,
// p = the four prediction columns as biased bytes
__device__ __forceinline__ int weight_cols_pre(const int pre[16], const uint32_t p[4]) {
    s16x2 A[4], B[4];   // A[c] = (R0[c], R1[c]) = rows 0,1 of the column-pass output, B[c] = rows 2,3
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int R0 = dot4s(p[c], K_W_R0N, pre[4 * c + 0]);
        const int R2 = dot4s(p[c], K_W_R2N, pre[4 * c + 1]);
        const int X = dot4s(p[c], K_W_XN, pre[4 * c + 2]);
        const int Y = dot4s(p[c], K_W_YN, pre[4 * c + 3]);
        const uint32_t xy = pk16(X, Y);
        const int t1 = dot2(xy, K_ROT16_A, 16 * 14500);
        const int t3 = dot2(xy, K_ROT16_B, 16 * 7500);
        A[c] = as_s16x2(__builtin_amdgcn_perm((uint32_t)t1, (uint32_t)R0, 0x07060100u));   // (low half of R0, high half of t1)
        B[c] = as_s16x2(__builtin_amdgcn_perm((uint32_t)t3, (uint32_t)R2, 0x07060100u));
    }
    return weight_rows(A, B);
}

// (Tried for the whole-pel search, where a reference column is the prediction column of up to four candidates: the four linear
// quantities of a column made once, as packed pairs (R0, R2) and (X, Y), and a (current, prediction) pair as two packed
// subtractions + the rotations + two permutes.  7 % fewer instructions in k_search1, but 75-79 registers instead of 61 -- six waves per
// SIMD instead of eight -- and the same or a lower rate with the part full: 68.1-68.5 and 67.4-67.7 against 68.4-68.7 M MB/s.)

// 4x4 byte transpose: rows r[0..3] (byte k = column k) -> columns (byte r = row r); eight v_perm
__device__ __forceinline__ void transpose4x4(const uint32_t r[4], uint32_t c[4]) {
    const uint32_t t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u);   // r0.b0 r1.b0 r0.b1 r1.b1
    const uint32_t t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);   // r0.b2 r1.b2 r0.b3 r1.b3
    const uint32_t t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u);
    const uint32_t t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
    c[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    c[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    c[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    c[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// VP8 six-tap filters by 1/8-pel phase, src/GPU_kernels.cl:563-572
static __device__ __constant__ const int8_t k_sixtap[8][8] = {
    {0, 0, 127, 0, 0, 0, 0, 0}, /* phase 0 is the identity: handled explicitly (tap 128 does not fit) */
    {0, -6, 123, 12, -1, 0, 0, 0},   {2, -11, 108, 36, -8, 1, 0, 0}, {0, -9, 93, 50, -6, 0, 0, 0},
    {3, -16, 77, 77, -16, 3, 0, 0},  {0, -6, 50, 93, -9, 0, 0, 0},   {1, -8, 36, 108, -11, 2, 0, 0},
    {0, -1, 12, 123, -6, 0, 0, 0},
};
__device__ __forceinline__ void load_taps(int phase, int f[6]) {
#pragma unroll
    for (int t = 0; t < 6; ++t) f[t] = k_sixtap[phase][t];
    if (phase == 0) f[2] = 128;
}
// sat8(s >> 7), written as clamp-then-shift.  The shift-then-saturate form is pattern-matched by
// hipcc (ROCm 7.2) into v_ashr_pk_u8_i32, whose result the compiler then ORs with further bytes as
// if bits 31:16 were zero; on gfx950 they keep the previous contents of the destination VGPR
// (0xffff after a negative sum), which corrupted two neighbouring pixels.  Found by the parity tests.
__device__ __forceinline__ int sat8_shr7(int s) { return iclamp(s, 0, 32767) >> 7; }
// (sum + 64)/128 with C truncation toward zero
__device__ __forceinline__ int div128(int s) { return (s + ((s >> 31) & 127)) >> 7; }   // s / 128 toward zero
#endif

}  // namespace vp8
