// fake_rccl.cpp — a stand-in for librccl.so in TESTS ONLY.
//
// RCCL does not admit two ranks on one GPU, and the test boxes have one GPU, so the
// library's multi-rank code (roots, index ranges, padded all-gather, grouped all-reduces,
// calls from ranks that own no particle) could otherwise only ever run in a world of one.
// This file implements the ten RCCL entry points the library resolves (PCL_RCCL_LIBRARY
// names the file to load) with host staging, so that several ranks can share a device:
//   * ncclCommInitAll: all ranks in one process; the calls of one ncclGroupStart/End pair
//     are queued and carried out at ncclGroupEnd, when every rank's call is known;
//   * ncclCommInitRank: one rank per process; the ranks meet in a POSIX shared-memory
//     segment named after the unique id and every call blocks until all ranks made it.
// Every call first waits for the caller's stream, then moves the bytes through host memory
// with synchronous copies: stricter ordering than RCCL's, the same results.  What this does
// NOT test is RCCL itself and the stream ordering of its kernels.
//
//   hipcc -shared -fPIC -O1 -o tests/fake_rccl/libfake_rccl.so tests/fake_rccl/fake_rccl.cpp
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <random>
#include <string>
#include <vector>

namespace {

enum Kind { kBroadcast, kAllReduce, kAllGather };

struct Op {
    Kind kind;
    const void* send;
    void* recv;
    size_t count;
    ncclDataType_t type;
    ncclRedOp_t red;
    int root;
    hipStream_t stream;
};

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

template <typename T>
void reduce_typed(T* acc, const T* in, size_t n, ncclRedOp_t op)
{
    for (size_t i = 0; i < n; ++i) {
        if (op == ncclSum) acc[i] = acc[i] + in[i];
        else if (op == ncclMax) acc[i] = in[i] > acc[i] ? in[i] : acc[i];
        else if (op == ncclMin) acc[i] = in[i] < acc[i] ? in[i] : acc[i];
    }
}

bool reduce_into(void* acc, const void* in, size_t n, ncclDataType_t t, ncclRedOp_t op)
{
    if (op != ncclSum && op != ncclMax && op != ncclMin) return false;
    switch (t) {
    case ncclFloat32: reduce_typed((float*)acc, (const float*)in, n, op); return true;
    case ncclFloat64: reduce_typed((double*)acc, (const double*)in, n, op); return true;
    case ncclUint64: reduce_typed((unsigned long long*)acc, (const unsigned long long*)in, n, op); return true;
    case ncclInt64: reduce_typed((long long*)acc, (const long long*)in, n, op); return true;
    case ncclUint32: reduce_typed((unsigned*)acc, (const unsigned*)in, n, op); return true;
    case ncclInt32: reduce_typed((int*)acc, (const int*)in, n, op); return true;
    default: return false;
    }
}

// ---- ranks in separate processes: a shared-memory segment ------------------------
constexpr size_t kSlotBytes = size_t(16) << 20;   // per rank and call; the tests move a few hundred KiB

struct ShmHeader {
    std::atomic<int> ready;
    std::atomic<int> arrived;
    std::atomic<int> generation;
    int world;
};

struct Team {
    int world = 1;
    // in-process teams
    std::vector<std::deque<Op>> pending;
    int alive = 0;
    // cross-process teams
    ShmHeader* shm = nullptr;
    size_t shm_bytes = 0;
    std::string shm_name;
};

}  // namespace

struct ncclComm {
    Team* team;
    int rank;
};

namespace {

thread_local int g_group_depth = 0;
thread_local std::vector<Team*> g_touched;   // in-process teams with queued calls

char* slot_of(Team* t, int rank) { return reinterpret_cast<char*>(t->shm) + 4096 + (size_t)rank * kSlotBytes; }

bool shm_barrier(Team* t)
{
    ShmHeader* h = t->shm;
    const int gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == t->world) {
        h->arrived.store(0);
        h->generation.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->generation.load() == gen) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;   // a rank died: do not hang the test
    }
    return true;
}

ncclResult_t run_across_processes(ncclComm* c, const Op& op)
{
    Team* t = c->team;
    const size_t esz = type_bytes(op.type), bytes = op.count * esz;
    if (esz == 0 || bytes > kSlotBytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    const bool contributes = op.kind != kBroadcast || c->rank == op.root;
    if (contributes && bytes && hipMemcpy(slot_of(t, c->rank), op.send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    if (!shm_barrier(t)) return ncclSystemError;
    ncclResult_t rc = ncclSuccess;
    if (op.kind == kBroadcast) {
        if (bytes && hipMemcpy(op.recv, slot_of(t, op.root), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    } else if (op.kind == kAllGather) {
        for (int r = 0; r < t->world && rc == ncclSuccess; ++r)
            if (bytes && hipMemcpy((char*)op.recv + (size_t)r * bytes, slot_of(t, r), bytes, hipMemcpyHostToDevice) != hipSuccess)
                rc = ncclUnhandledCudaError;
    } else {
        std::vector<char> acc(slot_of(t, 0), slot_of(t, 0) + bytes);
        for (int r = 1; r < t->world; ++r)
            if (!reduce_into(acc.data(), slot_of(t, r), op.count, op.type, op.red)) rc = ncclInvalidArgument;
        if (rc == ncclSuccess && bytes && hipMemcpy(op.recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
            rc = ncclUnhandledCudaError;
    }
    if (!shm_barrier(t)) return ncclSystemError;   // the slots may be overwritten from here on
    return rc;
}

// Carries out the k-th queued call of every rank of an in-process team.
ncclResult_t run_in_process(Team* t)
{
    for (;;) {
        bool any = false, all = true;
        for (auto& q : t->pending) { any = any || !q.empty(); all = all && !q.empty(); }
        if (!any) return ncclSuccess;
        if (!all) return ncclInvalidUsage;   // a rank is missing from the group call
        std::vector<Op> ops;
        for (auto& q : t->pending) { ops.push_back(q.front()); q.pop_front(); }
        const Op& o0 = ops[0];
        const size_t esz = type_bytes(o0.type), bytes = o0.count * esz;
        if (esz == 0) return ncclInvalidArgument;
        for (const Op& o : ops) {
            if (o.kind != o0.kind || o.count != o0.count || o.type != o0.type || o.root != o0.root || o.red != o0.red)
                return ncclInvalidUsage;
            if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        }
        std::vector<std::vector<char>> host((size_t)t->world, std::vector<char>(bytes));
        for (int r = 0; r < t->world; ++r)
            if ((o0.kind != kBroadcast || r == o0.root) && bytes &&
                hipMemcpy(host[r].data(), ops[r].send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
                return ncclUnhandledCudaError;
        if (o0.kind == kAllReduce)
            for (int r = 1; r < t->world; ++r)
                if (!reduce_into(host[0].data(), host[r].data(), o0.count, o0.type, o0.red)) return ncclInvalidArgument;
        for (int r = 0; r < t->world; ++r) {
            if (!bytes) continue;
            if (o0.kind == kAllGather) {
                for (int s = 0; s < t->world; ++s)
                    if (hipMemcpy((char*)ops[r].recv + (size_t)s * bytes, host[s].data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
                        return ncclUnhandledCudaError;
            } else {
                const std::vector<char>& src = host[o0.kind == kBroadcast ? o0.root : 0];
                if (hipMemcpy(ops[r].recv, src.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
            }
        }
    }
}

ncclResult_t submit(ncclComm* c, const Op& op)
{
    if (!c || !c->team) return ncclInvalidArgument;
    if (op.kind == kBroadcast && (op.root < 0 || op.root >= c->team->world)) return ncclInvalidArgument;
    Team* t = c->team;
    if (t->shm) return run_across_processes(c, op);
    t->pending[(size_t)c->rank].push_back(op);
    if (g_group_depth > 0) {
        bool known = false;
        for (Team* k : g_touched) known = known || k == t;
        if (!known) g_touched.push_back(t);
        return ncclSuccess;
    }
    return run_in_process(t);   // outside a group: legal only for a team of one, which run_in_process checks
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    std::random_device rd;
    std::snprintf(id->internal, sizeof id->internal, "/pcl_fake_rccl_%d_%08x%08x", (int)getpid(), rd(), rd());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Team* t = new Team();
    t->world = nranks;
    t->shm_name.assign(id.internal, strnlen(id.internal, sizeof id.internal));
    t->shm_bytes = 4096 + (size_t)nranks * kSlotBytes;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(t->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)t->shm_bytes) != 0) { delete t; return ncclSystemError; }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while ((fd = shm_open(t->shm_name.c_str(), O_RDWR, 0600)) < 0) {
            usleep(2000);
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { delete t; return ncclSystemError; }
        }
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < t->shm_bytes) usleep(2000);   // rank 0 is still sizing it
    }
    void* p = mmap(nullptr, t->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete t; return ncclSystemError; }
    t->shm = static_cast<ShmHeader*>(p);
    if (rank == 0) {
        t->shm->arrived.store(0);
        t->shm->generation.store(0);
        t->shm->world = nranks;
        t->shm->ready.store(1);
    } else {
        while (t->shm->ready.load() != 1) usleep(1000);
        if (t->shm->world != nranks) { munmap(p, t->shm_bytes); delete t; return ncclInvalidArgument; }
    }
    *comm = new ncclComm{t, rank};
    if (!shm_barrier(t)) return ncclSystemError;
    if (rank == 0) shm_unlink(t->shm_name.c_str());   // every rank has it mapped: the name can go
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
    (void)devlist;   // ranks may share a device here; that is the point
    if (!comms || ndev < 1) return ncclInvalidArgument;
    Team* t = new Team();
    t->world = ndev;
    t->pending.resize((size_t)ndev);
    t->alive = ndev;
    for (int i = 0; i < ndev; ++i) comms[i] = new ncclComm{t, i};
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->team->world;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclSuccess;
    Team* t = comm->team;
    if (t->shm) {
        munmap(t->shm, t->shm_bytes);
        delete t;
    } else if (--t->alive == 0) {
        delete t;
    }
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm,
                           hipStream_t stream)
{
    return submit(comm, Op{kBroadcast, send, recv, count, type, ncclSum, root, stream});
}

// For tests that count what a sequence of library calls puts on the fabric: all-reduce calls this
// process has made so far, by element type (index = ncclDataType_t, up to 15).
static unsigned long long g_allreduce_calls[16];
unsigned long long fake_rccl_allreduce_calls(int type) { return type >= 0 && type < 16 ? g_allreduce_calls[type] : 0; }

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    if ((int)type >= 0 && (int)type < 16) ++g_allreduce_calls[(int)type];
    return submit(comm, Op{kAllReduce, send, recv, count, type, op, 0, stream});
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t type, ncclComm_t comm,
                           hipStream_t stream)
{
    return submit(comm, Op{kAllGather, send, recv, sendcount, type, ncclSum, 0, stream});
}

ncclResult_t ncclGroupStart()
{
    ++g_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (Team* t : g_touched) {
        const ncclResult_t r = run_in_process(t);
        if (rc == ncclSuccess) rc = r;
    }
    g_touched.clear();
    return rc;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "stand-in: a HIP call failed";
    case ncclSystemError: return "stand-in: shared memory or a rank that never arrived";
    case ncclInvalidArgument: return "stand-in: invalid argument";
    case ncclInvalidUsage: return "stand-in: the ranks' calls do not match";
    default: return "stand-in: error";
    }
}

}  // extern "C"
