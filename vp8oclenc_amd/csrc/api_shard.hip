// api_shard.hip -- one frame's reference searches on several devices and the process group of a GOP-sharded run: RCCL inside the library.
// The reference runs the LAST / GOLDEN / ALTREF searches of a frame on three command queues (inter_part.h:122-135, 201-236), reads the
// results back at :263-266, and hands the filtered reconstruction to the next frame at vp8enc.cpp:395-401.
//
// RCCL is resolved when a host first asks for it (dlopen): a single-GPU host never needs librccl.so to load this library, and a process
// that already has a copy mapped gets that one.
#include <ctype.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include <rccl/rccl.h>

#include "vp8hip_ctx.h"

using namespace vp8;

namespace {

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    bool ok = false;
};

// $VP8HIP_RCCL_LIBRARY; then the RCCL that belongs to the HIP runtime this process runs on (next to its libamdhip64: a process that
// also holds another copy -- PyTorch bundles one, built for ITS runtime -- must not get that one by soname); then $ROCM_PATH/lib,
// /opt/rocm/lib; then whatever the process's search path says
const Rccl *rccl() {
    static const Rccl table = [] {
        Rccl r;
        void *h = nullptr;
        std::string tried[8];
        int n = 0;
        if (const char *e = getenv("VP8HIP_RCCL_LIBRARY")) tried[n++] = e;
        Dl_info info;
        if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir = info.dli_fname;
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) tried[n++] = dir.substr(0, slash) + "/librccl.so.1";
        }
        if (const char *e = getenv("ROCM_PATH")) tried[n++] = std::string(e) + "/lib/librccl.so.1";
        tried[n++] = "/opt/rocm/lib/librccl.so.1";
        tried[n++] = "librccl.so.1";
        tried[n++] = "librccl.so";
        for (int i = 0; i < n && !h; ++i) h = dlopen(tried[i].c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            fprintf(stderr, "vp8hip: librccl.so.1 not found (%s): the vp8hip_shard_* / vp8hip_group_* entry points need RCCL; set VP8HIP_RCCL_LIBRARY\n", dlerror());
            return r;
        }
#define RCCL_SYM(name) r.name = reinterpret_cast<decltype(r.name)>(dlsym(h, "nccl" #name))
        RCCL_SYM(GetUniqueId); RCCL_SYM(CommInitRank); RCCL_SYM(CommDestroy); RCCL_SYM(CommCount); RCCL_SYM(GroupStart); RCCL_SYM(GroupEnd);
        RCCL_SYM(Broadcast); RCCL_SYM(AllReduce); RCCL_SYM(AllGather); RCCL_SYM(Send); RCCL_SYM(Recv);
#undef RCCL_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.GroupStart && r.GroupEnd && r.Broadcast && r.AllReduce &&
               r.AllGather && r.Send && r.Recv;
        return r;
    }();
    return table.ok ? &table : nullptr;
}

// Where one reference's search results live -- what vp8hip_inter_finish reads: the landing place of vp8hip_import_search and of
// the broadcasts of vp8hip_shard_share_search alike
struct SearchNets { int32_t *vectors, *costs; size_t words; };
SearchNets search_nets(vp8hip_ctx *c, int ref) {
    return SearchNets{reinterpret_cast<int32_t *>(c->nets.net[ref][0]), reinterpret_cast<int32_t *>(c->nets.bdiff[ref]), (size_t)c->b8};
}

}  // namespace

namespace vp8 {

void shard_release(vp8hip_ctx *c) {
    if (!c->shard_comm) return;
    if (const Rccl *r = rccl()) r->CommDestroy(c->shard_comm);
    c->shard_comm = nullptr;
}

// The surface a LAST that comes from elsewhere is received into (any surface no reference and no reconstruction lives in) ...
int receive_last_surface(const vp8hip_ctx *c) { return pick_free_frame(c); }
// ... and what makes it this context's LAST once the planes are there (or enqueued in front of whatever reads them): replicated edges
// and pyramid are made where the next frame begins, as after a loop filter of this context's own (vp8enc.cpp:395-401).  Shared by
// vp8hip_import_last and the receiving ranks of vp8hip_shard_share_last.
int adopt_last(vp8hip_ctx *c, int idx) {
    if (idx < 0 || idx >= NFRAMES) return VP8HIP_ERR_STATE;
    c->frames[idx].pyramid_valid = false;
    c->frames[idx].border_valid = false;
    c->slot[0] = idx;
    return VP8HIP_OK;
}

}  // namespace vp8

extern "C" {

int vp8hip_export_search(vp8hip_ctx *c, int ref, void *d_vectors, void *d_costs) {
    USE_DEVICE(c);
    if (!c || ref < 0 || ref > 2 || !d_vectors || !d_costs) return VP8HIP_ERR_ARG;
    const SearchNets n = search_nets(c, ref);
    HIPCHK(c, hipMemcpyAsync(d_vectors, n.vectors, n.words * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_costs, n.costs, n.words * 4, hipMemcpyDeviceToDevice, c->stream));
    return VP8HIP_OK;
}

int vp8hip_import_search(vp8hip_ctx *c, int ref, const void *d_vectors, const void *d_costs) {
    USE_DEVICE(c);
    if (!c || ref < 0 || ref > 2 || !d_vectors || !d_costs) return VP8HIP_ERR_ARG;
    const SearchNets n = search_nets(c, ref);
    HIPCHK(c, hipMemcpyAsync(n.vectors, d_vectors, n.words * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(n.costs, d_costs, n.words * 4, hipMemcpyDeviceToDevice, c->stream));
    return VP8HIP_OK;
}

int vp8hip_export_last(vp8hip_ctx *c, void *d_y, void *d_u, void *d_v) {
    USE_DEVICE(c);
    if (!c || !d_y || !d_u || !d_v) return VP8HIP_ERR_ARG;
    JOIN_LF(c);
    if (c->slot[0] < 0) return VP8HIP_ERR_STATE;
    const Frame &f = c->frames[c->slot[0]].f;
    const Plane *pl[3] = {&f.Y[0], &f.U, &f.V};
    void *dst[3] = {d_y, d_u, d_v};
    for (int i = 0; i < 3; ++i)
        HIPCHK(c, hipMemcpy2DAsync(dst[i], pl[i]->w, pl[i]->p, pl[i]->stride, pl[i]->w, pl[i]->h, hipMemcpyDeviceToDevice, c->stream));
    return VP8HIP_OK;
}

// The other end of vp8hip_export_last: tight planes in this device's memory become LAST the way a receiving rank of
// vp8hip_shard_share_last gets it -- into a free surface of the pool, adopted without touching GOLDEN / ALTREF, edges and pyramid
// made when the next frame begins.  (vp8hip_set_last_device makes the edges at once and is what a host uses to START a sequence.)
int vp8hip_import_last(vp8hip_ctx *c, const void *d_y, const void *d_u, const void *d_v) {
    USE_DEVICE(c);
    if (!c || !d_y || !d_u || !d_v) return VP8HIP_ERR_ARG;
    JOIN_LF(c);
    const int idx = receive_last_surface(c);
    if (idx < 0) return VP8HIP_ERR_STATE;
    const Frame &f = c->frames[idx].f;
    const Plane *pl[3] = {&f.Y[0], &f.U, &f.V};
    const void *src[3] = {d_y, d_u, d_v};
    for (int i = 0; i < 3; ++i)
        HIPCHK(c, hipMemcpy2DAsync(pl[i]->p, pl[i]->stride, src[i], pl[i]->w, pl[i]->w, pl[i]->h, hipMemcpyDeviceToDevice, c->stream));
    return adopt_last(c, idx);
}

// ---- the exchanges of a frame split by reference, inside the library: RCCL on the context's stream, no host synchronisation ----
int vp8hip_shard_unique_id(uint8_t id[VP8HIP_SHARD_ID_BYTES]) {
    static_assert(VP8HIP_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as opaque bytes");
    if (!id) return VP8HIP_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return VP8HIP_ERR_HIP;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return VP8HIP_ERR_HIP;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return VP8HIP_OK;
}

int vp8hip_shard_init(vp8hip_ctx *c, const uint8_t id[VP8HIP_SHARD_ID_BYTES], int rank, int world) {
    USE_DEVICE(c);
    if (!c || !id || world < 1 || world > 3 || rank < 0 || rank >= world) return VP8HIP_ERR_ARG;
    if (c->shard_comm || c->batch) return VP8HIP_ERR_STATE;
    const Rccl *r = rccl();
    if (!r) return VP8HIP_ERR_HIP;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    if (r->CommInitRank(&c->shard_comm, world, u, rank) != ncclSuccess) {
        c->shard_comm = nullptr;
        return VP8HIP_ERR_HIP;
    }
    c->shard_rank = rank;
    c->shard_world = world;
    return VP8HIP_OK;
}

int vp8hip_shard_rank(const vp8hip_ctx *c) { return c && c->shard_comm ? c->shard_rank : -1; }
int vp8hip_shard_world(const vp8hip_ctx *c) { return c && c->shard_comm ? c->shard_world : 0; }

// Every searched reference's quarter-pel vector net and cost net (8 bytes per 8x8 block) from the rank that searched it
// (reference r: rank r mod world) to all ranks, IN PLACE in the nets the searches wrote and vp8hip_inter_finish reads: one
// group of broadcasts = one RCCL launch on the context's stream.
int vp8hip_shard_share_search(vp8hip_ctx *c, int used_mask) {
    USE_DEVICE(c);
    if (!c || (used_mask & ~7)) return VP8HIP_ERR_ARG;
    const Rccl *rc = c->shard_comm ? rccl() : nullptr;
    if (!rc) return VP8HIP_ERR_STATE;
    JOIN_LF(c);
    bool ok = rc->GroupStart() == ncclSuccess;
    for (int r = 0; r < 3 && ok; ++r) {
        if (!(used_mask & (1 << r))) continue;
        const int root = r % c->shard_world;
        const SearchNets n = search_nets(c, r);      // (a receiving rank's landing place is vp8hip_import_search's)
        ok = rc->Broadcast(n.vectors, n.vectors, n.words, ncclInt32, root, c->shard_comm, c->stream) == ncclSuccess &&
             rc->Broadcast(n.costs, n.costs, n.words, ncclInt32, root, c->shard_comm, c->stream) == ncclSuccess;
    }
    ok = (rc->GroupEnd() == ncclSuccess) && ok;
    return ok ? VP8HIP_OK : VP8HIP_ERR_HIP;
}

// The filtered reconstruction of rank `root` (its LAST after vp8hip_loop_filter) becomes every rank's LAST: the three padded
// planes straight out of root's frame pool into a free surface of the others' pools (vp8enc.cpp:395-401 is what the
// reference does with it on one device).
int vp8hip_shard_share_last(vp8hip_ctx *c, int root) {
    USE_DEVICE(c);
    if (!c || root < 0) return VP8HIP_ERR_ARG;
    const Rccl *rc = c->shard_comm ? rccl() : nullptr;
    if (!rc || root >= c->shard_world) return VP8HIP_ERR_STATE;
    JOIN_LF(c);
    const bool receiver = c->shard_rank != root;
    const int idx = receiver ? receive_last_surface(c) : c->slot[0];
    if (idx < 0) return VP8HIP_ERR_STATE;
    const Frame &f = c->frames[idx].f;
    const Plane *pl[3] = {&f.Y[0], &f.U, &f.V};
    bool ok = rc->GroupStart() == ncclSuccess;
    for (int i = 0; i < 3 && ok; ++i) {
        uint8_t *base = pl[i]->p - (size_t)PAD * pl[i]->stride - PAD;       // the plane with its margins: one contiguous piece
        const size_t bytes = (size_t)pl[i]->stride * (pl[i]->h + 2 * PAD);
        ok = rc->Broadcast(base, base, bytes, ncclUint8, root, c->shard_comm, c->stream) == ncclSuccess;
    }
    ok = (rc->GroupEnd() == ncclSuccess) && ok;
    if (!ok) return VP8HIP_ERR_HIP;
    return receiver ? adopt_last(c, idx) : VP8HIP_OK;
}

// barrier + maximum over the ranks of one double (a time), on the context's stream; blocks
int vp8hip_shard_max(vp8hip_ctx *c, double *value) {
    USE_DEVICE(c);
    if (!c || !value) return VP8HIP_ERR_ARG;
    const Rccl *rc = c->shard_comm ? rccl() : nullptr;
    if (!rc) return VP8HIP_ERR_STATE;
    double *d = reinterpret_cast<double *>(c->scratch);
    HIPCHK(c, hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (rc->AllReduce(d, d, 1, ncclDouble, ncclMax, c->shard_comm, c->stream) != ncclSuccess) return VP8HIP_ERR_HIP;
    HIPCHK(c, hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VP8HIP_OK;
}

}  // extern "C"

// ---- the process group of a GOP-sharded run -------------------------------------------------------------------------------------
// GOP chunks are independent (intra_part.h:1091-1098): the data path of a sharded run has no collective.  What the ranks of one node
// still need from each other -- starting together, the slowest rank's time, the finished frames in one place (the reference's single
// output file, encIO.h:1-30 + vp8enc.cpp:476-481) -- is here, so that a host needs no second GPU framework for it.
struct vp8hip_group {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint8_t *d_buf = nullptr;      // staging in device memory (RCCL moves device memory), grown on demand
    size_t d_cap = 0;
    uint8_t *h_pin = nullptr;      // pinned bounce buffer for small records
    int last_hip_error = 0;
    std::string id_file;           // rank 0: the rendezvous file to remove
};

namespace {

constexpr size_t GROUP_PIN_BYTES = 1 << 16;

int group_reserve(vp8hip_group *g, size_t bytes) {
    if (bytes <= g->d_cap) return VP8HIP_OK;
    if (g->d_buf) (void)hipFree(g->d_buf);
    g->d_buf = nullptr;
    g->d_cap = 0;
    const size_t cap = (bytes + (1u << 20)) & ~(size_t)((1u << 20) - 1);
    HIPCHK(g, hipMalloc(&g->d_buf, cap));
    g->d_cap = cap;
    return VP8HIP_OK;
}

// Where the id file of a run lives: $VP8HIP_RENDEZVOUS_DIR (the caller's responsibility), else $XDG_RUNTIME_DIR, else /tmp/vp8hip-<uid> --
// in every case a directory that must belong to this user and be closed to everybody else (0700): on a shared node nobody else can then
// plant a file or a symbolic link where the ranks will read or rank 0 will write.  Empty string = no such directory can be had.
std::string rendezvous_dir() {
    const uid_t me = getuid();
    auto private_dir = [&](const std::string &d) {
        struct stat st;
        return lstat(d.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == me && (st.st_mode & 077) == 0;
    };
    if (const char *e = getenv("VP8HIP_RENDEZVOUS_DIR"))
        if (e[0]) return e;          // (a test's tmp_path, a job's scratch: named explicitly, taken as it is)
    if (const char *e = getenv("XDG_RUNTIME_DIR"))
        if (e[0] && private_dir(e)) return e;
    const std::string d = "/tmp/vp8hip-" + std::to_string((unsigned)me);
    if (mkdir(d.c_str(), 0700) != 0 && errno != EEXIST) return "";
    return private_dir(d) ? d : "";     // (somebody else's directory, a link, or one open to others: refused)
}

std::string rendezvous_path(const char *key) {
    std::string p = rendezvous_dir();
    if (p.empty()) return p;
    p += "/vp8hip-rdzv-";
    p += std::to_string((unsigned)getuid());
    p += "-";
    for (const char *k = key; *k; ++k) p += (isalnum((unsigned char)*k) || *k == '-' || *k == '_' || *k == '.') ? *k : '_';
    return p;
}

}  // namespace

extern "C" {

// The 128 bytes of vp8hip_shard_unique_id from rank 0 to the other ranks of ONE node through a file: rank 0 removes whatever a crashed run
// of the same key left behind, makes the id and writes <dir>/vp8hip-rdzv-<uid>-<key> atomically (a new file opened O_EXCL | O_NOFOLLOW,
// then rename), the others poll for it (timeout_s) and take only a regular file of THIS user that is not older than timeout_s before their
// own start (the ranks of a run start within that of each other; anything older is a leftover).  `key` names the run: every rank of a run
// passes the same string, and no two runs alive on the node at once may share it (bench.py: launcher pid + MASTER_PORT).  The directory:
// rendezvous_dir() above -- private to the user.  Rank 0 removes the file in vp8hip_group_create once every rank has joined.
int vp8hip_group_rendezvous(const char *key, int rank, double timeout_s, uint8_t id[VP8HIP_SHARD_ID_BYTES]) {
    if (!key || !key[0] || rank < 0 || !id) return VP8HIP_ERR_ARG;
    const std::string path = rendezvous_path(key);
    if (path.empty()) return VP8HIP_ERR_STATE;
    if (rank == 0) {
        const int rc = vp8hip_shard_unique_id(id);
        if (rc) return rc;
        unlink(path.c_str());                                    // a stale id of an earlier run with this key must never be read
        const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
        unlink(tmp.c_str());
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
        if (fd < 0) return VP8HIP_ERR_STATE;
        const bool ok = write(fd, id, VP8HIP_SHARD_ID_BYTES) == VP8HIP_SHARD_ID_BYTES;
        close(fd);
        if (!ok || rename(tmp.c_str(), path.c_str()) != 0) {
            unlink(tmp.c_str());
            return VP8HIP_ERR_STATE;
        }
        return VP8HIP_OK;
    }
    timespec t0, wall0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    clock_gettime(CLOCK_REALTIME, &wall0);
    const double oldest = (double)wall0.tv_sec + 1e-9 * (double)wall0.tv_nsec - (timeout_s > 1.0 ? timeout_s : 1.0);
    const uid_t me = getuid();
    for (;;) {
        const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
        if (fd >= 0) {
            struct stat st;
            const bool mine = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == me &&
                              (double)st.st_mtim.tv_sec + 1e-9 * (double)st.st_mtim.tv_nsec >= oldest;
            const ssize_t n = mine ? read(fd, id, VP8HIP_SHARD_ID_BYTES) : -1;
            close(fd);
            if (n == VP8HIP_SHARD_ID_BYTES) return VP8HIP_OK;
        }
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        if ((double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec) > timeout_s) return VP8HIP_ERR_TIMEOUT;
        usleep(2000);
    }
}

int vp8hip_group_create(vp8hip_group **out, int device_ordinal, const uint8_t id[VP8HIP_SHARD_ID_BYTES], int rank, int world, const char *key) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return VP8HIP_ERR_ARG;
    *out = nullptr;
    const Rccl *r = rccl();
    if (!r) return VP8HIP_ERR_HIP;
    if (device_ordinal < 0 || device_ordinal >= vp8hip_device_count()) return VP8HIP_ERR_NO_DEVICE;
    vp8hip_group *g = new (std::nothrow) vp8hip_group();
    if (!g) return VP8HIP_ERR_ARG;
    g->device = device_ordinal;
    g->rank = rank;
    g->world = world;
    if (key && key[0] && rank == 0) g->id_file = rendezvous_path(key);
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    bool ok = hipSetDevice(device_ordinal) == hipSuccess && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess &&
              hipHostMalloc(&g->h_pin, GROUP_PIN_BYTES) == hipSuccess;
    bool timed_out = false;
    if (ok) {
        // ncclCommInitRank is a collective that returns when EVERY rank has joined -- and never, if one of them died or read a wrong id.
        // It runs on a thread of its own and this call waits for it $VP8HIP_GROUP_TIMEOUT_S seconds (default 300; 0 = for ever): past that
        // the call fails with VP8HIP_ERR_TIMEOUT and the host can end the run (the thread and the half-made communicator are abandoned --
        // RCCL offers no way to take a blocking init back; a process that gets this error is expected to exit).
        struct Init { std::mutex m; std::condition_variable cv; bool done = false; ncclResult_t res = ncclSuccess; ncclComm_t comm = nullptr; };
        std::shared_ptr<Init> st = std::make_shared<Init>();
        std::thread([st, r, u, world, rank, device_ordinal] {
            ncclComm_t c = nullptr;
            ncclResult_t res = hipSetDevice(device_ordinal) == hipSuccess ? r->CommInitRank(&c, world, u, rank) : ncclUnhandledCudaError;
            std::lock_guard<std::mutex> l(st->m);
            st->res = res;
            st->comm = c;
            st->done = true;
            st->cv.notify_all();
        }).detach();
        const char *e = getenv("VP8HIP_GROUP_TIMEOUT_S");
        const double limit = e && e[0] ? atof(e) : 300.0;
        std::unique_lock<std::mutex> l(st->m);
        if (limit > 0) timed_out = !st->cv.wait_for(l, std::chrono::duration<double>(limit), [&] { return st->done; });
        else st->cv.wait(l, [&] { return st->done; });
        if (timed_out || st->res != ncclSuccess) ok = false;
        else g->comm = st->comm;
    }
    if (!g->id_file.empty()) unlink(g->id_file.c_str());                     // (every rank has read it by then -- or never will)
    if (!ok) {
        if (timed_out) fprintf(stderr, "vp8hip_group_create: rank %d of %d waited for the other ranks in ncclCommInitRank and gave up (VP8HIP_GROUP_TIMEOUT_S)\n", rank, world);
        vp8hip_group_destroy(g);
        return timed_out ? VP8HIP_ERR_TIMEOUT : VP8HIP_ERR_HIP;
    }
    *out = g;
    return VP8HIP_OK;
}

void vp8hip_group_destroy(vp8hip_group *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->comm)
        if (const Rccl *r = rccl()) r->CommDestroy(g->comm);
    if (g->d_buf) (void)hipFree(g->d_buf);
    if (g->h_pin) (void)hipHostFree(g->h_pin);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

int vp8hip_group_rank(const vp8hip_group *g) { return g ? g->rank : -1; }
int vp8hip_group_world(const vp8hip_group *g) { return g ? g->world : 0; }
// ranks RCCL itself counts in the communicator (ncclCommCount): what a log line shows to prove that N ranks met
int vp8hip_group_count(const vp8hip_group *g) {
    int n = 0;
    const Rccl *r = g && g->comm ? rccl() : nullptr;
    return r && r->CommCount(g->comm, &n) == ncclSuccess ? n : 0;
}

// `bytes` (<= 4 KB) from every rank, in rank order, into all[world * bytes] on every rank: blocks.  A barrier as a side effect.
int vp8hip_group_all_gather(vp8hip_group *g, const void *mine, size_t bytes, void *all) {
    if (!g || !mine || !all || bytes == 0 || bytes > 4096 || bytes * (size_t)g->world > GROUP_PIN_BYTES / 2) return VP8HIP_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return VP8HIP_ERR_HIP;
    HIPCHK(g, hipSetDevice(g->device));
    const size_t total = bytes * (size_t)g->world;
    int rc = group_reserve(g, total + bytes);
    if (rc) return rc;
    memcpy(g->h_pin, mine, bytes);
    HIPCHK(g, hipMemcpyAsync(g->d_buf + total, g->h_pin, bytes, hipMemcpyHostToDevice, g->stream));
    if (r->AllGather(g->d_buf + total, g->d_buf, bytes, ncclUint8, g->comm, g->stream) != ncclSuccess) return VP8HIP_ERR_HIP;
    HIPCHK(g, hipMemcpyAsync(g->h_pin + GROUP_PIN_BYTES / 2, g->d_buf, total, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    memcpy(all, g->h_pin + GROUP_PIN_BYTES / 2, total);
    return VP8HIP_OK;
}

int vp8hip_group_barrier(vp8hip_group *g) {
    if (!g) return VP8HIP_ERR_ARG;
    uint8_t mine = 1, all[1024];
    if (g->world > (int)sizeof(all)) return VP8HIP_ERR_ARG;
    return vp8hip_group_all_gather(g, &mine, 1, all);
}

// barrier + maximum over the ranks of one double (the slowest rank's time); blocks
int vp8hip_group_max(vp8hip_group *g, double *value) {
    if (!g || !value) return VP8HIP_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return VP8HIP_ERR_HIP;
    HIPCHK(g, hipSetDevice(g->device));
    int rc = group_reserve(g, 64);
    if (rc) return rc;
    memcpy(g->h_pin, value, sizeof(double));
    double *d = reinterpret_cast<double *>(g->d_buf);
    HIPCHK(g, hipMemcpyAsync(d, g->h_pin, sizeof(double), hipMemcpyHostToDevice, g->stream));
    if (r->AllReduce(d, d, 1, ncclDouble, ncclMax, g->comm, g->stream) != ncclSuccess) return VP8HIP_ERR_HIP;
    HIPCHK(g, hipMemcpyAsync(g->h_pin, d, sizeof(double), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    memcpy(value, g->h_pin, sizeof(double));
    return VP8HIP_OK;
}

// `bytes` of host memory from rank `root` to every rank's buf; blocks
int vp8hip_group_broadcast(vp8hip_group *g, int root, void *buf, size_t bytes) {
    if (!g || !buf || root < 0 || root >= g->world) return VP8HIP_ERR_ARG;
    if (!bytes) return VP8HIP_OK;
    const Rccl *r = rccl();
    if (!r) return VP8HIP_ERR_HIP;
    HIPCHK(g, hipSetDevice(g->device));
    int rc = group_reserve(g, bytes);
    if (rc) return rc;
    if (g->rank == root) HIPCHK(g, hipMemcpyAsync(g->d_buf, buf, bytes, hipMemcpyHostToDevice, g->stream));
    if (r->Broadcast(g->d_buf, g->d_buf, bytes, ncclUint8, root, g->comm, g->stream) != ncclSuccess) return VP8HIP_ERR_HIP;
    if (g->rank != root) HIPCHK(g, hipMemcpyAsync(buf, g->d_buf, bytes, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    return VP8HIP_OK;
}

// Every rank's `bytes` of host memory (counts[r] on rank r: the same array on every rank, from vp8hip_group_all_gather) laid end to
// end in rank order into dst on `root` (dst is not used elsewhere): the finished frames of a GOP-sharded run on their way to the one
// writer.  ONE code path at every world size: every rank, root included, sends with ncclSend; root receives from every rank, itself
// included, with ncclRecv, in one group.  Blocks.
int vp8hip_group_gather_bytes(vp8hip_group *g, int root, const void *src, size_t bytes, void *dst, const uint64_t *counts) {
    if (!g || !counts || root < 0 || root >= g->world || (bytes && !src) || counts[g->rank] != bytes) return VP8HIP_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return VP8HIP_ERR_HIP;
    HIPCHK(g, hipSetDevice(g->device));
    size_t total = 0;
    for (int k = 0; k < g->world; ++k) total += counts[k];
    if (g->rank == root && total && !dst) return VP8HIP_ERR_ARG;
    const size_t send_at = g->rank == root ? total : 0;      // root's own bytes are staged behind the place everything lands in
    int rc = group_reserve(g, send_at + bytes + 16);
    if (rc) return rc;
    if (bytes) HIPCHK(g, hipMemcpyAsync(g->d_buf + send_at, src, bytes, hipMemcpyHostToDevice, g->stream));
    bool ok = r->GroupStart() == ncclSuccess;
    if (ok && bytes) ok = r->Send(g->d_buf + send_at, bytes, ncclUint8, root, g->comm, g->stream) == ncclSuccess;
    if (g->rank == root) {
        size_t at = 0;
        for (int k = 0; k < g->world && ok; ++k) {
            if (counts[k]) ok = r->Recv(g->d_buf + at, counts[k], ncclUint8, k, g->comm, g->stream) == ncclSuccess;
            at += counts[k];
        }
    }
    ok = (r->GroupEnd() == ncclSuccess) && ok;
    if (!ok) return VP8HIP_ERR_HIP;
    if (g->rank == root && total) HIPCHK(g, hipMemcpyAsync(dst, g->d_buf, total, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    return VP8HIP_OK;
}

int vp8hip_group_last_hip_error(const vp8hip_group *g) { return g ? g->last_hip_error : 0; }

}  // extern "C"
