// standin_rccl.cpp -- TEST INFRASTRUCTURE: the eleven RCCL entry points libvp8hip.so resolves at run time (csrc/api_shard.hip), over a
// shared-memory segment between processes of ONE node, so that the library's multi-rank code -- vp8hip_shard_share_search / _share_last
// with a rank that is NOT the root, vp8hip_group_gather_bytes with real peers, bench.py --gpus 2 -- can be EXECUTED on a box with one
// GPU.  RCCL itself refuses two ranks on one device ("Duplicate GPU detected"); everything above the transport is the product's code.
//   g++ -shared -fPIC -O2 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tests/standin_rccl/standin_rccl.cpp -o standin_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -lpthread
//   VP8HIP_RCCL_LIBRARY=.../standin_rccl.so   (the library's own switch: include/vp8hip.h)
// Not a communication library: every operation synchronises its stream, moves the bytes through host memory with blocking copies and
// meets the other ranks at a barrier in the segment.  The order of operations is the callers' (every rank issues the same collectives
// in the same order, as RCCL demands); operations between ncclGroupStart and ncclGroupEnd are carried out at ncclGroupEnd, all sends
// before all receives.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t BOX_BYTES = 128u << 20;           // a rank's outbox: one group's sends, or the payload of a collective
constexpr int MAX_ENTRIES = 64;

struct Entry { int dst; size_t offset, bytes; };
struct Control {
    std::atomic<int> arrived;                      // ranks that have mapped the segment
    std::atomic<int> bar_count;
    std::atomic<int> bar_sense;
    std::atomic<int> failed;
    int n_entries[MAX_RANKS];
    Entry entries[MAX_RANKS][MAX_ENTRIES];
    double slot[MAX_RANKS][64];                    // small records (AllReduce, AllGather of a few bytes)
};

struct Comm {
    int rank, world;
    Control *ctl;
    uint8_t *boxes;                                // world * BOX_BYTES
    size_t map_bytes;
    char name[80];
    int sense = 0;
};

struct Op { int kind; const void *send; void *recv; size_t bytes; int peer; ncclRedOp_t op; ncclDataType_t type; Comm *comm; hipStream_t stream; };
thread_local int tl_group_depth = 0;
thread_local std::vector<Op> tl_ops;

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

bool barrier(Comm *c) {
    Control *k = c->ctl;
    c->sense ^= 1;
    if (k->bar_count.fetch_add(1) + 1 == c->world) {
        k->bar_count.store(0);
        k->bar_sense.store(c->sense);
        return true;
    }
    timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (k->bar_sense.load() != c->sense) {
        if (k->failed.load()) return false;
        clock_gettime(CLOCK_MONOTONIC, &t);
        if (t.tv_sec - t0.tv_sec > 120) { k->failed.store(1); return false; }   // a rank died: nobody waits for ever
        usleep(20);
    }
    return true;
}

uint8_t *box(Comm *c, int r) { return c->boxes + (size_t)r * BOX_BYTES; }

ncclResult_t run(const Op &o) {
    Comm *c = o.comm;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;   // what was enqueued in front of the operation is done
    switch (o.kind) {
        case 0: {   // broadcast
            if (o.bytes > BOX_BYTES) return ncclInvalidArgument;
            if (c->rank == o.peer && hipMemcpy(box(c, o.peer), o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            if (!barrier(c)) return ncclSystemError;
            if (c->rank != o.peer || o.recv != o.send)
                if (hipMemcpy(o.recv, box(c, o.peer), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
            return barrier(c) ? ncclSuccess : ncclSystemError;
        }
        case 1: {   // all-reduce of doubles, max (all the library asks for)
            if (o.type != ncclDouble || o.op != ncclMax || o.bytes > sizeof(c->ctl->slot[0])) return ncclInvalidArgument;
            if (hipMemcpy(c->ctl->slot[c->rank], o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            if (!barrier(c)) return ncclSystemError;
            double out[64];
            const int n = (int)(o.bytes / sizeof(double));
            for (int i = 0; i < n; ++i) {
                out[i] = c->ctl->slot[0][i];
                for (int r = 1; r < c->world; ++r) out[i] = c->ctl->slot[r][i] > out[i] ? c->ctl->slot[r][i] : out[i];
            }
            if (hipMemcpy(o.recv, out, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
            return barrier(c) ? ncclSuccess : ncclSystemError;
        }
        case 2: {   // all-gather
            if (o.bytes > BOX_BYTES) return ncclInvalidArgument;
            if (hipMemcpy(box(c, c->rank), o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            if (!barrier(c)) return ncclSystemError;
            for (int r = 0; r < c->world; ++r)
                if (hipMemcpy((uint8_t *)o.recv + (size_t)r * o.bytes, box(c, r), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
            return barrier(c) ? ncclSuccess : ncclSystemError;
        }
    }
    return ncclInvalidArgument;
}

// the sends and receives of one group: every send into the sender's outbox, a barrier, every receive out of the sender's outbox
ncclResult_t run_p2p(const std::vector<Op> &ops) {
    if (ops.empty()) return ncclSuccess;
    Comm *c = ops[0].comm;
    size_t at = 0;
    int n = 0;
    for (const Op &o : ops) {
        if (o.kind != 3) continue;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (at + o.bytes > BOX_BYTES || n >= MAX_ENTRIES) return ncclInvalidArgument;
        if (hipMemcpy(box(c, c->rank) + at, o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        c->ctl->entries[c->rank][n++] = Entry{o.peer, at, o.bytes};
        at += o.bytes;
    }
    c->ctl->n_entries[c->rank] = n;
    if (!barrier(c)) return ncclSystemError;
    int taken[MAX_RANKS] = {};
    for (const Op &o : ops) {
        if (o.kind != 4) continue;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        const int src = o.peer;
        int seen = 0, k = -1;
        for (int i = 0; i < c->ctl->n_entries[src]; ++i)         // the (taken[src] + 1)-th send of `src` to this rank
            if (c->ctl->entries[src][i].dst == c->rank && seen++ == taken[src]) { k = i; break; }
        if (k < 0 || c->ctl->entries[src][k].bytes != o.bytes) return ncclInvalidArgument;
        ++taken[src];
        if (hipMemcpy(o.recv, box(c, src) + c->ctl->entries[src][k].offset, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return barrier(c) ? ncclSuccess : ncclSystemError;
}

ncclResult_t submit(const Op &o) {
    if (tl_group_depth > 0) { tl_ops.push_back(o); return ncclSuccess; }
    if (o.kind >= 3) { std::vector<Op> one{o}; return run_p2p(one); }
    return run(o);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof(*id));
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof(id->internal), "/vp8standin-%d-%ld-%ld", (int)getpid(), (long)t.tv_sec, (long)t.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank;
    c->world = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->map_bytes = ((sizeof(Control) + 4095) & ~(size_t)4095) + (size_t)nranks * BOX_BYTES;
    int fd = -1;
    for (int tries = 0; tries < 60000 && fd < 0; ++tries) {      // rank 0 makes the segment, the others wait for it
        fd = rank == 0 ? shm_open(c->name, O_CREAT | O_RDWR | O_EXCL, 0600) : shm_open(c->name, O_RDWR, 0600);
        if (fd < 0) { if (rank == 0) break; usleep(1000); }
    }
    if (fd < 0) { delete c; return ncclSystemError; }
    if (rank == 0 && ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); delete c; return ncclSystemError; }
    struct stat st;
    for (int tries = 0; tries < 60000; ++tries) {                 // (a freshly made segment has no size yet)
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes) break;
        usleep(1000);
    }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->ctl = static_cast<Control *>(p);          // (zero pages: every counter starts at 0)
    c->boxes = static_cast<uint8_t *>(p) + ((sizeof(Control) + 4095) & ~(size_t)4095);
    c->ctl->arrived.fetch_add(1);
    timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (c->ctl->arrived.load() < nranks) {
        clock_gettime(CLOCK_MONOTONIC, &t);
        if (t.tv_sec - t0.tv_sec > 180) { munmap(p, c->map_bytes); delete c; return ncclSystemError; }
        usleep(200);
    }
    if (rank == 0) shm_unlink(c->name);          // every rank has it mapped: the name can go
    *out = reinterpret_cast<ncclComm_t>(c);
    fprintf(stderr, "STAND-IN RCCL (tests/standin_rccl): rank %d of %d\n", rank, nranks);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    munmap(c->ctl, c->map_bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = reinterpret_cast<const Comm *>(comm)->world;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++tl_group_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (tl_group_depth <= 0) return ncclInvalidUsage;
    if (--tl_group_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(tl_ops);
    ncclResult_t rc = ncclSuccess;
    std::vector<Op> p2p;
    for (const Op &o : ops) {
        if (o.kind >= 3) { p2p.push_back(o); continue; }
        if (rc == ncclSuccess) rc = run(o);
    }
    if (rc == ncclSuccess && !p2p.empty()) rc = run_p2p(p2p);
    return rc;
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) {
    return submit(Op{0, sendbuff, recvbuff, count * type_size(datatype), root, ncclSum, datatype, reinterpret_cast<Comm *>(comm), stream});
}
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    return submit(Op{1, sendbuff, recvbuff, count * type_size(datatype), 0, op, datatype, reinterpret_cast<Comm *>(comm), stream});
}
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    return submit(Op{2, sendbuff, recvbuff, sendcount * type_size(datatype), 0, ncclSum, datatype, reinterpret_cast<Comm *>(comm), stream});
}
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return submit(Op{3, sendbuff, nullptr, count * type_size(datatype), peer, ncclSum, datatype, reinterpret_cast<Comm *>(comm), stream});
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return submit(Op{4, nullptr, recvbuff, count * type_size(datatype), peer, ncclSum, datatype, reinterpret_cast<Comm *>(comm), stream});
}

}  // extern "C"
