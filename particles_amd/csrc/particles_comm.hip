// particles_comm.hip — the update over several GPUs, RCCL inside the library.
//
// Particles shard by contiguous index range, one handle (pcl_compute) per GPU;
// the field and the parameters are replicated.  What travels over xGMI:
//   * the field: one ncclBroadcast straight into every handle's field buffer
//     at construction and after a whole-field replace, the dirty range only
//     after a patch, 20 bytes per cell for an indexed patch, and the 64-byte
//     cursor record alone for a brush event (every replica then edits itself);
//   * per step, only when the pair term is active: one ncclAllGather of 16
//     bytes per particle (position and type), after which every range is
//     updated against all records in global index order;
//   * per step, only when step diagnostics are on: one ncclGroupStart/End pair
//     of all-reduces over the 48-byte record (sum of count and kinetic energy
//     in f64, max of the six f32 bounds).
// Every collective is enqueued on the handle's own stream, so it is ordered
// with the kernels before and after it without host synchronisation.  One
// exception, one process per rank only: the first step after attaching, after
// a call that replaced the particle set, and after a change of what a step
// exchanges (pair term active or not, step diagnostics on or off) starts with
// one all-reduce that is read back on the host (agree): the length the ranges
// are padded to in the all-gather, and a check that the ranks are about to
// issue the same collectives -- if not, every rank returns an error instead
// of waiting for the others' missing calls.  Ranks must therefore replace
// their particle sets and change these settings between the same two steps.
//
// Two ways to form the communicator: one process (or thread) per GPU attaches
// its handle with a shared unique id (pcl_comm_attach: ncclCommInitRank), or
// one process drives all GPUs of the node through a group (pcl_group_create:
// ncclCommInitAll, every collective inside ncclGroupStart/End).  No
// counterpart in the reference, which runs on one device
// (src/compute.rs:221-227); the call sites a host would change are
// src/main.rs:232-236 (construction) and src/main.rs:371-390 (per frame).
//
// RCCL is resolved at first use with dlopen, so the library loads (and the
// single-GPU path runs) on machines without it; PCL_RCCL_LIBRARY names the
// file to load when the default search (an RCCL already in the process, then
// librccl.so.1, then librccl.so) is not wanted.
#include "handle_access.hpp"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

static_assert(PCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace {

using pcl_internal::fail;

// ---- RCCL entry points, resolved at first use ---------------------------------
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    Rccl& r = g_rccl;
    const char* named = std::getenv("PCL_RCCL_LIBRARY");
    if (named && *named) {
        r.lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { r.error = std::string("PCL_RCCL_LIBRARY=") + named + ": " + dlerror(); return; }
    } else {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)   // a copy the process already holds (a framework's) wins
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { r.error = std::string("RCCL not found (librccl.so.1): ") + dlerror(); return; }
    }
    bool ok = true;
    auto sym = [&](const char* name) -> void* {
        void* p = dlsym(r.lib, name);
        if (!p) { ok = false; r.error = std::string("RCCL lacks ") + name; }
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(r.lib); r.lib = nullptr; }
}

int need_rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.lib) return fail(PCL_ERR_COMM, g_rccl.error);
    return PCL_OK;
}

#define PCL_NCCL_TRY(expr)                                                                        \
    do {                                                                                          \
        ncclResult_t pcl_r_ = (expr);                                                             \
        if (pcl_r_ != ncclSuccess)                                                                \
            return fail(PCL_ERR_COMM, std::string(#expr) + ": " + g_rccl.GetErrorString(pcl_r_)); \
    } while (0)

#define PCL_HIP_TRY(expr)                                                                   \
    do {                                                                                    \
        hipError_t pcl_e_ = (expr);                                                         \
        if (pcl_e_ != hipSuccess)                                                           \
            return fail(PCL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(pcl_e_)); \
    } while (0)

#define PCL_REQUIRE(cond, msg)                           \
    do {                                                 \
        if (!(cond)) return fail(PCL_ERR_INVALID, msg);  \
    } while (0)

// ---- per-handle communicator state ---------------------------------------------
constexpr int kRecordRing = 8;

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    void* d_record = nullptr;          // 64-byte cursor record as broadcast
    // The root's records wait in pinned memory for their copy to the device, so that a brush event
    // never makes the host wait for the stream (the caller's record is only valid during the call):
    // a ring of kRecordRing slots, each taken again once the copy made from it that many events ago has run.
    pcl_brush_input* h_records = nullptr;
    hipEvent_t ev_record_copied[kRecordRing] = {};
    int record_next = 0;
    char* d_stage = nullptr;           // indexed patches: cells then indices
    size_t stage_bytes = 0;
    // The root's cells and indices wait in pinned memory for their copy to the device (the caller's arrays
    // are only valid during the call): two buffers taken in turn, each again once its last copy has run.
    char* h_stage[2] = {nullptr, nullptr};
    size_t h_stage_bytes[2] = {0, 0};
    hipEvent_t ev_stage_copied[2] = {nullptr, nullptr};
    int stage_next = 0;
    float* d_pair_send = nullptr;      // this range's 16-byte (x, y, z, type) records, padded
    float* d_pair_recv = nullptr;      // all ranges' records
    size_t pair_cap = 0;               // records per range in those buffers
    unsigned long long* d_agree = nullptr;   // all-reduce(max) of {local count, flags}: see agree()
    unsigned long long* h_agree = nullptr;   // pinned
    bool agreed = false;                     // what the ranks last agreed on, and for which local state
    uint64_t agreed_epoch = 0;
    bool agreed_pairs = false, agreed_diag = false;
    size_t agreed_cap = 0;
    pcl_diagnostics* d_global = nullptr;     // the all-reduced record
    // The per-step all-reduce of the diagnostics runs beside the next step, on a stream of its own:
    // nothing on the handle's stream needs its result.  Steps write their record alternately into two
    // slots (pcl_internal::set_diag_target), the side stream waits for the step, all-reduces its slot
    // into d_global and marks the slot free again; a step waits for the all-reduce that last read the
    // slot it is about to write, which was issued two steps earlier.
    hipStream_t side = nullptr;
    hipEvent_t ev_ready[2] = {nullptr, nullptr};   // handle's stream: the slot holds the step's record
    hipEvent_t ev_done[2] = {nullptr, nullptr};    // side stream: the all-reduce of the slot has finished
    pcl_diagnostics* d_slot = nullptr;             // two records
    int next_slot = 0;                             // the slot the next step with diagnostics writes
    int armed_slot = -1;                           // the slot the handle currently writes its record into
    int pending_slot = -1;                         // slot of the latest all-reduce issued on the side stream
};

Comm* comm_of(pcl_compute* h) { return h ? static_cast<Comm*>(pcl_internal::comm_of(h)) : nullptr; }

void release_comm(void* p)
{
    Comm* c = static_cast<Comm*>(p);
    if (!c) return;
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    for (int k = 0; k < 2; ++k) {
        if (c->ev_ready[k]) (void)hipEventDestroy(c->ev_ready[k]);
        if (c->ev_done[k]) (void)hipEventDestroy(c->ev_done[k]);
    }
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->d_slot) (void)hipFree(c->d_slot);
    if (c->d_record) (void)hipFree(c->d_record);
    if (c->h_records) (void)hipHostFree(c->h_records);
    for (hipEvent_t e : c->ev_record_copied)
        if (e) (void)hipEventDestroy(e);
    if (c->d_stage) (void)hipFree(c->d_stage);
    for (int k = 0; k < 2; ++k) {
        if (c->h_stage[k]) (void)hipHostFree(c->h_stage[k]);
        if (c->ev_stage_copied[k]) (void)hipEventDestroy(c->ev_stage_copied[k]);
    }
    if (c->d_pair_send) (void)hipFree(c->d_pair_send);
    if (c->d_pair_recv) (void)hipFree(c->d_pair_recv);
    if (c->d_agree) (void)hipFree(c->d_agree);
    if (c->h_agree) (void)hipHostFree(c->h_agree);
    if (c->d_global) (void)hipFree(c->d_global);
    delete c;
}

// What a call that waits for the handle's stream waits for first: the side stream's all-reduces.
int sync_comm(void* p)
{
    Comm* c = static_cast<Comm*>(p);
    if (c && c->side && c->pending_slot >= 0) PCL_HIP_TRY(hipStreamSynchronize(c->side));
    return PCL_OK;
}

// Work enqueued on the handle's stream from here on runs after the latest all-reduce of the side stream.
int order_after_side(pcl_compute* h, Comm* c)
{
    if (c->pending_slot >= 0) PCL_HIP_TRY(hipStreamWaitEvent(pcl_internal::stream_of(h), c->ev_done[c->pending_slot], 0));
    return PCL_OK;
}

// Adopts `comm` (already initialised for h's device) as h's communicator.
int adopt(pcl_compute* h, ncclComm_t comm, int rank, int world)
{
    Comm* c = new (std::nothrow) Comm();
    if (!c) return fail(PCL_ERR_INVALID, "out of host memory");
    c->comm = comm;
    c->rank = rank;
    c->world = world;
    pcl_internal::set_comm(h, c, release_comm, sync_comm);   // from here on pcl_destroy releases it
    if (int rc = pcl_internal::make_current(h)) return rc;
    PCL_HIP_TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        PCL_HIP_TRY(hipEventCreateWithFlags(&c->ev_ready[k], hipEventDisableTiming));
        PCL_HIP_TRY(hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming));
    }
    PCL_HIP_TRY(hipMalloc((void**)&c->d_slot, 2 * sizeof(pcl_diagnostics)));
    PCL_HIP_TRY(hipMalloc(&c->d_record, sizeof(pcl_brush_input)));
    PCL_HIP_TRY(hipHostMalloc((void**)&c->h_records, kRecordRing * sizeof(pcl_brush_input), hipHostMallocDefault));
    for (int k = 0; k < kRecordRing; ++k) PCL_HIP_TRY(hipEventCreateWithFlags(&c->ev_record_copied[k], hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) PCL_HIP_TRY(hipEventCreateWithFlags(&c->ev_stage_copied[k], hipEventDisableTiming));
    PCL_HIP_TRY(hipMalloc((void**)&c->d_agree, 8 * sizeof(unsigned long long)));
    PCL_HIP_TRY(hipHostMalloc((void**)&c->h_agree, 8 * sizeof(unsigned long long), hipHostMallocDefault));
    PCL_HIP_TRY(hipMalloc((void**)&c->d_global, sizeof(pcl_diagnostics)));
    PCL_HIP_TRY(hipMemset(c->d_global, 0, sizeof(pcl_diagnostics)));
    return PCL_OK;
}

// The handles one call works on: one (a rank of its own) or all of a group's.
// Several communicators driven from one thread must be inside a group call.
struct Span {
    pcl_compute* const* h;
    int n;
};

// One ncclGroupStart/End pair around the collectives a call issues (needed as soon as one thread drives
// more than one communicator).  The group is closed on every way out of the scope: left open by an early
// return, it would swallow every later collective of the thread (queued, never launched), and the next
// synchronisation would hang.  The first error stays the call's result.
class GroupScope {
    bool open_ = false;

public:
    GroupScope() = default;
    GroupScope(const GroupScope&) = delete;
    GroupScope& operator=(const GroupScope&) = delete;
    ~GroupScope() { if (open_) (void)g_rccl.GroupEnd(); }
    int begin(bool needed)
    {
        if (needed) {
            PCL_NCCL_TRY(g_rccl.GroupStart());
            open_ = true;
        }
        return PCL_OK;
    }
    int end()
    {
        if (open_) {
            open_ = false;
            PCL_NCCL_TRY(g_rccl.GroupEnd());
        }
        return PCL_OK;
    }
};

int check_span(const Span& s)
{
    if (int rc = need_rccl()) return rc;
    PCL_REQUIRE(s.n >= 1, "no handle");
    for (int i = 0; i < s.n; ++i)
        PCL_REQUIRE(s.h[i] && comm_of(s.h[i]), "the handle has no communicator: pcl_comm_attach or pcl_group_create first");
    return PCL_OK;
}

// Handle of the span that is rank `root` (null when another process holds it).
pcl_compute* root_of(const Span& s, int root)
{
    for (int i = 0; i < s.n; ++i)
        if (comm_of(s.h[i])->rank == root) return s.h[i];
    return nullptr;
}

// Broadcast of `count` cells into every handle's field at `first_cell`; the root's cells come
// from host memory (`host_cells`, only looked at where the root is in this span).
int bcast_cells(const Span& s, const float (*host_cells)[4], size_t first_cell, size_t count, int root)
{
    if (count == 0) return PCL_OK;
    if (pcl_compute* r = root_of(s, root)) {
        PCL_REQUIRE(host_cells != nullptr, "the root rank passed no cells");
        if (int rc = pcl_patch_force_grid(r, first_cell, host_cells, count)) return rc;   // host -> root's field (synchronises)
    }
    GroupScope group;
    if (int rc = group.begin(s.n > 1)) return rc;
    for (int i = 0; i < s.n; ++i) {
        pcl_compute* h = s.h[i];
        if (int rc = pcl_internal::make_current(h)) return rc;
        float* at = static_cast<float*>(pcl_internal::field_of(h)) + 4 * first_cell;
        PCL_NCCL_TRY(g_rccl.Broadcast(at, at, count * 4, ncclFloat, root, comm_of(h)->comm, pcl_internal::stream_of(h)));
    }
    return group.end();
}

int update_force_grid(const Span& s, const float (*grid)[4], size_t n_cells, int root)
{
    if (int rc = check_span(s)) return rc;
    PCL_REQUIRE(n_cells >= 1, "the force grid is empty");
    PCL_REQUIRE(root >= 0 && root < comm_of(s.h[0])->world, "root rank out of range");
    for (int i = 0; i < s.n; ++i) {
        void* dev = nullptr;
        if (int rc = pcl_internal::field_for_write(s.h[i], n_cells, &dev)) return rc;
    }
    return bcast_cells(s, grid, 0, n_cells, root);
}

int patch_force_grid(const Span& s, size_t first_cell, const float (*cells)[4], size_t count, int root)
{
    if (int rc = check_span(s)) return rc;
    for (int i = 0; i < s.n; ++i)
        if (first_cell > pcl_internal::cells_of(s.h[i]) || count > pcl_internal::cells_of(s.h[i]) - first_cell)
            return fail(PCL_ERR_OUT_OF_RANGE, "patch_force_grid: range outside the field");
    return bcast_cells(s, cells, first_cell, count, root);
}

int reserve_stage(pcl_compute* h, Comm* c, size_t bytes)
{
    if (bytes <= c->stage_bytes) return PCL_OK;
    PCL_HIP_TRY(hipStreamSynchronize(pcl_internal::stream_of(h)));
    if (c->d_stage) PCL_HIP_TRY(hipFree(c->d_stage));
    c->d_stage = nullptr;
    c->stage_bytes = 0;
    PCL_HIP_TRY(hipMalloc((void**)&c->d_stage, bytes));
    c->stage_bytes = bytes;
    return PCL_OK;
}

int patch_force_grid_indexed(const Span& s, const uint32_t* idx, const float (*cells)[4], size_t count, int root)
{
    if (int rc = check_span(s)) return rc;
    if (count == 0) return PCL_OK;
    const size_t bytes = count * 20;   // 16 bytes of cell, then 4 of index
    for (int i = 0; i < s.n; ++i) {
        if (int rc = pcl_internal::make_current(s.h[i])) return rc;
        if (int rc = reserve_stage(s.h[i], comm_of(s.h[i]), bytes)) return rc;
    }
    if (pcl_compute* r = root_of(s, root)) {
        PCL_REQUIRE(idx != nullptr && cells != nullptr, "the root rank passed no cells");
        for (size_t k = 0; k < count; ++k)
            if (idx[k] >= pcl_internal::cells_of(r))
                return fail(PCL_ERR_OUT_OF_RANGE, "patch_force_grid_indexed: cell index outside the field");
        Comm* c = comm_of(r);
        if (int rc = pcl_internal::make_current(r)) return rc;
        const int k = c->stage_next;
        c->stage_next ^= 1;
        PCL_HIP_TRY(hipEventSynchronize(c->ev_stage_copied[k]));   // two patches old (never recorded: returns at once)
        if (c->h_stage_bytes[k] < bytes) {
            if (c->h_stage[k]) PCL_HIP_TRY(hipHostFree(c->h_stage[k]));
            c->h_stage[k] = nullptr;
            c->h_stage_bytes[k] = 0;
            PCL_HIP_TRY(hipHostMalloc((void**)&c->h_stage[k], bytes, hipHostMallocDefault));
            c->h_stage_bytes[k] = bytes;
        }
        std::memcpy(c->h_stage[k], cells, count * 16);
        std::memcpy(c->h_stage[k] + count * 16, idx, count * 4);
        PCL_HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage[k], bytes, hipMemcpyHostToDevice, pcl_internal::stream_of(r)));
        PCL_HIP_TRY(hipEventRecord(c->ev_stage_copied[k], pcl_internal::stream_of(r)));
    }
    {
        GroupScope group;
        if (int rc = group.begin(s.n > 1)) return rc;
        for (int i = 0; i < s.n; ++i) {
            pcl_compute* h = s.h[i];
            Comm* c = comm_of(h);
            if (int rc = pcl_internal::make_current(h)) return rc;
            PCL_NCCL_TRY(g_rccl.Broadcast(c->d_stage, c->d_stage, bytes, ncclChar, root, c->comm, pcl_internal::stream_of(h)));
        }
        if (int rc = group.end()) return rc;
    }
    for (int i = 0; i < s.n; ++i) {
        Comm* c = comm_of(s.h[i]);
        if (int rc = pcl_patch_force_grid_indexed_device(s.h[i], c->d_stage + count * 16, c->d_stage, count)) return rc;
    }
    return PCL_OK;
}

// The root's 64-byte cursor record on every rank's device, then the brush event there.
int brush_event(const Span& s, const pcl_brush_input* in, int root, bool down)
{
    if (int rc = check_span(s)) return rc;
    if (pcl_compute* r = root_of(s, root)) {
        PCL_REQUIRE(in != nullptr, "the root rank passed no cursor record");
        PCL_REQUIRE(in->mode >= PCL_EDIT_CENTERED && in->mode <= PCL_EDIT_ROTATE, "unknown edit mode");
        PCL_REQUIRE(in->falloff >= PCL_FALLOFF_ABRUPT && in->falloff <= PCL_FALLOFF_INVERSE_DISTANCE, "unknown falloff");
        if (int rc = pcl_internal::make_current(r)) return rc;
        Comm* c = comm_of(r);
        const int k = c->record_next;
        c->record_next = (k + 1) % kRecordRing;
        PCL_HIP_TRY(hipEventSynchronize(c->ev_record_copied[k]));   // eight events old (never recorded: returns at once)
        c->h_records[k] = *in;
        PCL_HIP_TRY(hipMemcpyAsync(c->d_record, &c->h_records[k], sizeof(pcl_brush_input), hipMemcpyHostToDevice,
                                   pcl_internal::stream_of(r)));
        PCL_HIP_TRY(hipEventRecord(c->ev_record_copied[k], pcl_internal::stream_of(r)));
    }
    {
        GroupScope group;
        if (int rc = group.begin(s.n > 1)) return rc;
        for (int i = 0; i < s.n; ++i) {
            pcl_compute* h = s.h[i];
            Comm* c = comm_of(h);
            if (int rc = pcl_internal::make_current(h)) return rc;
            PCL_NCCL_TRY(g_rccl.Broadcast(c->d_record, c->d_record, sizeof(pcl_brush_input), ncclChar, root, c->comm,
                                          pcl_internal::stream_of(h)));
        }
        if (int rc = group.end()) return rc;
    }
    for (int i = 0; i < s.n; ++i) {
        const int rc = down ? pcl_brush_mouse_down_device(s.h[i], comm_of(s.h[i])->d_record)
                            : pcl_brush_mouse_moved_device(s.h[i], comm_of(s.h[i])->d_record);
        if (rc) return rc;
    }
    return PCL_OK;
}

// Before a step whose record will be all-reduced beside the next step: the handle writes that record
// into the slot that is free, once the all-reduce that last read the slot (two steps ago) has finished.
int arm_diagnostics_slot(pcl_compute* h, Comm* c)
{
    const int k = c->next_slot;
    c->next_slot ^= 1;
    if (int rc = pcl_internal::make_current(h)) return rc;
    PCL_HIP_TRY(hipStreamWaitEvent(pcl_internal::stream_of(h), c->ev_done[k], 0));   // never recorded: no wait
    pcl_internal::set_diag_target(h, c->d_slot + k);
    c->armed_slot = k;
    return PCL_OK;
}

// Sum {count, kinetic energy} (f64) and max {max xyz, -min xyz} (f32) of every handle's record
// into its d_global: one group call, two all-reduces per handle.  `beside` (the per-step all-reduce):
// on the side stream, from the slot armed for the step, while the handle's stream goes on with the
// next step; otherwise on the handle's stream, after whatever the side stream still has to write
// into d_global.
int allreduce_diagnostics(const Span& s, bool beside)
{
    std::vector<const void*> local(s.n, nullptr);
    for (int i = 0; i < s.n; ++i)
        if (int rc = pcl_internal::diagnostics_on_device(s.h[i], &local[i])) return rc;
    std::vector<int> slot(s.n, -1);
    for (int i = 0; i < s.n; ++i) {
        pcl_compute* h = s.h[i];
        Comm* c = comm_of(h);
        if (int rc = pcl_internal::make_current(h)) return rc;
        hipStream_t own = pcl_internal::stream_of(h);
        if (!beside) {
            if (int rc = order_after_side(h, c)) return rc;
            continue;
        }
        const int k = slot[i] = c->armed_slot;
        PCL_REQUIRE(k >= 0 && local[i] == c->d_slot + k, "the step's record is not in the slot armed for it");
        PCL_HIP_TRY(hipEventRecord(c->ev_ready[k], own));
        PCL_HIP_TRY(hipStreamWaitEvent(c->side, c->ev_ready[k], 0));
    }
    {
        GroupScope group;   // always: two all-reduces per handle travel as one call
        if (int rc = group.begin(true)) return rc;
        for (int i = 0; i < s.n; ++i) {
            pcl_compute* h = s.h[i];
            Comm* c = comm_of(h);
            if (int rc = pcl_internal::make_current(h)) return rc;
            const char* in = beside ? reinterpret_cast<const char*>(c->d_slot + slot[i]) : static_cast<const char*>(local[i]);
            char* out = reinterpret_cast<char*>(c->d_global);
            hipStream_t stream = beside ? c->side : pcl_internal::stream_of(h);
            PCL_NCCL_TRY(g_rccl.AllReduce(in, out, 2, ncclDouble, ncclSum, c->comm, stream));
            PCL_NCCL_TRY(g_rccl.AllReduce(in + 16, out + 16, 6, ncclFloat, ncclMax, c->comm, stream));
        }
        if (int rc = group.end()) return rc;
    }
    if (beside)
        for (int i = 0; i < s.n; ++i) {
            pcl_compute* h = s.h[i];
            Comm* c = comm_of(h);
            if (int rc = pcl_internal::make_current(h)) return rc;
            PCL_HIP_TRY(hipEventRecord(c->ev_done[slot[i]], c->side));
            c->pending_slot = slot[i];
        }
    return PCL_OK;
}

// What the ranks agree on before they step together: the largest local particle count (every range
// is padded to it in the all-gather) and whether a step carries the pair exchange and the
// diagnostics all-reduce.  A span that holds every rank (one process driving the whole group) knows
// all of it without asking.  One process per rank asks once, with one all-reduce (max) and one host
// synchronisation, and keeps the answer until its own handle's particle set is replaced or its own
// flags change: ranks make the same calls between the same two steps, so they all ask again at the
// same step.  Flags found to differ between ranks when they ask come back as an error on every rank
// instead of a hang on mismatched collectives (a rank that changes its flags alone, without the others
// reaching a point where they ask, is not caught: the others do not know that there is a question).
constexpr int kAgreeWords = 5;   // local count, pairs, !pairs, diagnostics, !diagnostics

uint64_t key_epoch(const pcl_compute* h) { return pcl_internal::particles_epoch(h); }

bool agreement_current(const Span& s, bool pairs, bool diag)
{
    for (int i = 0; i < s.n; ++i) {
        const Comm* c = comm_of(s.h[i]);
        if (!c->agreed || c->agreed_epoch != key_epoch(s.h[i]) || c->agreed_pairs != pairs || c->agreed_diag != diag)
            return false;
    }
    return true;
}

int agree(const Span& s, bool pairs, bool diag)
{
    auto settle = [&](size_t cap) {
        for (int i = 0; i < s.n; ++i) {
            Comm* c = comm_of(s.h[i]);
            c->agreed = true;
            c->agreed_epoch = key_epoch(s.h[i]);
            c->agreed_pairs = pairs;
            c->agreed_diag = diag;
            c->agreed_cap = cap;
        }
    };
    if (s.n == comm_of(s.h[0])->world) {
        size_t most = 0;
        for (int i = 0; i < s.n; ++i) most = std::max(most, pcl_num_particles(s.h[i]));
        settle(most);
        return PCL_OK;
    }
    for (int i = 0; i < s.n; ++i) {
        pcl_compute* h = s.h[i];
        Comm* c = comm_of(h);
        c->agreed = false;
        if (int rc = pcl_internal::make_current(h)) return rc;
        c->h_agree[0] = (unsigned long long)pcl_num_particles(h);
        c->h_agree[1] = pairs ? 1 : 0;
        c->h_agree[2] = pairs ? 0 : 1;
        c->h_agree[3] = diag ? 1 : 0;
        c->h_agree[4] = diag ? 0 : 1;
        PCL_HIP_TRY(hipMemcpyAsync(c->d_agree, c->h_agree, kAgreeWords * sizeof(unsigned long long), hipMemcpyHostToDevice,
                                   pcl_internal::stream_of(h)));
    }
    {
        GroupScope group;
        if (int rc = group.begin(s.n > 1)) return rc;
        for (int i = 0; i < s.n; ++i) {
            pcl_compute* h = s.h[i];
            Comm* c = comm_of(h);
            if (int rc = pcl_internal::make_current(h)) return rc;
            PCL_NCCL_TRY(g_rccl.AllReduce(c->d_agree, c->d_agree, kAgreeWords, ncclUint64, ncclMax, c->comm,
                                          pcl_internal::stream_of(h)));
        }
        if (int rc = group.end()) return rc;
    }
    for (int i = 0; i < s.n; ++i) {
        pcl_compute* h = s.h[i];
        Comm* c = comm_of(h);
        if (int rc = pcl_internal::make_current(h)) return rc;
        PCL_HIP_TRY(hipMemcpyAsync(c->h_agree, c->d_agree, kAgreeWords * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                   pcl_internal::stream_of(h)));
        PCL_HIP_TRY(hipStreamSynchronize(pcl_internal::stream_of(h)));
    }
    const unsigned long long* a = comm_of(s.h[0])->h_agree;
    if (a[1] && a[2])
        return fail(PCL_ERR_INVALID, "the ranks hold different parameters: the pair term is active on some and inert on others");
    if (a[3] && a[4])
        return fail(PCL_ERR_INVALID, "step diagnostics are enabled on some ranks and not on others");
    settle((size_t)a[0]);
    return PCL_OK;
}

// One update of every range.  Pair term inert: no data-path collective at all.  Active: pack,
// all-gather, update against all records in global index order (src/compute.wgsl:115-144 sums
// over j = 0 .. N-1; the sharded sums then equal the one-device sums bit for bit).
int step(const Span& s)
{
    if (int rc = check_span(s)) return rc;
    const bool pairs = pcl_internal::pair_term_active(s.h[0]);
    const bool diag = pcl_internal::step_diagnostics_enabled(s.h[0]);
    for (int i = 1; i < s.n; ++i) {
        PCL_REQUIRE(pcl_internal::pair_term_active(s.h[i]) == pairs, "the handles of a group hold different parameters");
        PCL_REQUIRE(pcl_internal::step_diagnostics_enabled(s.h[i]) == diag,
                    "step diagnostics are enabled on some handles of a group and not on others");
    }
    if (!agreement_current(s, pairs, diag))
        if (int rc = agree(s, pairs, diag)) return rc;
    if (diag)
        for (int i = 0; i < s.n; ++i)
            if (int rc = arm_diagnostics_slot(s.h[i], comm_of(s.h[i]))) return rc;
    if (!pairs) {
        for (int i = 0; i < s.n; ++i)
            if (int rc = pcl_step(s.h[i])) return rc;
    } else {
        const size_t cap = comm_of(s.h[0])->agreed_cap;
        if (cap == 0) {
            for (int i = 0; i < s.n; ++i)
                if (int rc = pcl_step(s.h[i])) return rc;
        } else {
            const size_t world = (size_t)comm_of(s.h[0])->world;
            for (int i = 0; i < s.n; ++i) {
                pcl_compute* h = s.h[i];
                Comm* c = comm_of(h);
                if (int rc = pcl_internal::make_current(h)) return rc;
                if (c->pair_cap < cap) {
                    PCL_HIP_TRY(hipStreamSynchronize(pcl_internal::stream_of(h)));
                    if (c->d_pair_send) PCL_HIP_TRY(hipFree(c->d_pair_send));
                    if (c->d_pair_recv) PCL_HIP_TRY(hipFree(c->d_pair_recv));
                    c->d_pair_send = c->d_pair_recv = nullptr;
                    c->pair_cap = 0;
                    PCL_HIP_TRY(hipMalloc((void**)&c->d_pair_send, cap * 16));
                    PCL_HIP_TRY(hipMalloc((void**)&c->d_pair_recv, cap * world * 16));
                    c->pair_cap = cap;
                }
                if (int rc = pcl_pack_pair_sources_device(h, c->d_pair_send, cap)) return rc;
            }
            {
                GroupScope group;
                if (int rc = group.begin(s.n > 1)) return rc;
                for (int i = 0; i < s.n; ++i) {
                    pcl_compute* h = s.h[i];
                    Comm* c = comm_of(h);
                    if (int rc = pcl_internal::make_current(h)) return rc;
                    PCL_NCCL_TRY(g_rccl.AllGather(c->d_pair_send, c->d_pair_recv, cap * 4, ncclFloat, c->comm,
                                                  pcl_internal::stream_of(h)));
                }
                if (int rc = group.end()) return rc;
            }
            for (int i = 0; i < s.n; ++i)
                if (int rc = pcl_step_pairs_gathered(s.h[i], comm_of(s.h[i])->d_pair_recv, cap * world)) return rc;
        }
    }
    if (diag) return allreduce_diagnostics(s, true);
    return PCL_OK;
}

int diagnostics_read(const Span& s, pcl_diagnostics* out)
{
    if (int rc = check_span(s)) return rc;
    PCL_REQUIRE(out != nullptr, "out is null");
    if (int rc = allreduce_diagnostics(s, false)) return rc;
    pcl_compute* h = s.h[0];
    if (int rc = pcl_internal::make_current(h)) return rc;
    PCL_HIP_TRY(hipMemcpyAsync(out, comm_of(h)->d_global, sizeof(pcl_diagnostics), hipMemcpyDeviceToHost,
                               pcl_internal::stream_of(h)));
    for (int i = 0; i < s.n; ++i)
        if (int rc = pcl_synchronize(s.h[i])) return rc;
    return PCL_OK;
}

}  // namespace

struct pcl_group {
    std::vector<pcl_compute*> h;
    std::vector<size_t> first;   // first[i] .. first[i + 1]: the global indices handle i holds
    size_t n_total = 0;
    Span span() const { return Span{h.data(), (int)h.size()}; }
};

namespace {

void split_ranges(pcl_group* g, size_t n_total)
{
    const size_t G = g->h.size();
    g->n_total = n_total;
    g->first.assign(G + 1, 0);
    // g * N / G without overflow of the product
    for (size_t i = 0; i <= G; ++i)
        g->first[i] = (size_t)(((unsigned __int128)i * n_total) / G);
}

}  // namespace

extern "C" {

int pcl_comm_unique_id(void* out_id)
{
    if (int rc = need_rccl()) return rc;
    PCL_REQUIRE(out_id != nullptr, "pcl_comm_unique_id: out_id is null");
    ncclUniqueId id;
    PCL_NCCL_TRY(g_rccl.GetUniqueId(&id));
    std::memcpy(out_id, &id, sizeof id);
    return PCL_OK;
}

int pcl_comm_attach(pcl_compute* h, const void* unique_id, int rank, int world)
{
    if (int rc = need_rccl()) return rc;
    PCL_REQUIRE(h != nullptr && unique_id != nullptr, "pcl_comm_attach: null argument");
    PCL_REQUIRE(world >= 1 && rank >= 0 && rank < world, "pcl_comm_attach: rank outside [0, world)");
    PCL_REQUIRE(comm_of(h) == nullptr, "pcl_comm_attach: the handle already has a communicator");
    if (int rc = pcl_internal::make_current(h)) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    PCL_NCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank));
    return adopt(h, comm, rank, world);
}

int pcl_comm_detach(pcl_compute* h)
{
    PCL_REQUIRE(h != nullptr, "null handle");
    if (!comm_of(h)) return PCL_OK;
    if (int rc = pcl_synchronize(h)) return rc;
    pcl_internal::set_diag_target(h, nullptr);   // the handle's own record again: the slots go with the communicator
    release_comm(comm_of(h));
    pcl_internal::set_comm(h, nullptr, nullptr);
    return PCL_OK;
}

int pcl_comm_rank(const pcl_compute* h)
{
    const Comm* c = h ? static_cast<const Comm*>(pcl_internal::comm_of(h)) : nullptr;
    return c ? c->rank : -1;
}

int pcl_comm_world(const pcl_compute* h)
{
    const Comm* c = h ? static_cast<const Comm*>(pcl_internal::comm_of(h)) : nullptr;
    return c ? c->world : 0;
}

int pcl_comm_count(const pcl_compute* h)
{
    const Comm* c = h ? static_cast<const Comm*>(pcl_internal::comm_of(h)) : nullptr;
    if (!c || !c->comm || !g_rccl.CommCount) return 0;
    int count = 0;
    if (g_rccl.CommCount(c->comm, &count) != ncclSuccess) return -1;
    return count;
}

int pcl_comm_update_force_grid(pcl_compute* h, const float (*force_grid)[4], size_t n_cells, int root)
{
    return update_force_grid(Span{&h, 1}, force_grid, n_cells, root);
}

int pcl_comm_patch_force_grid(pcl_compute* h, size_t first_cell, const float (*cells)[4], size_t count, int root)
{
    return patch_force_grid(Span{&h, 1}, first_cell, cells, count, root);
}

int pcl_comm_patch_force_grid_indexed(pcl_compute* h, const uint32_t* cell_indices, const float (*cells)[4],
                                      size_t count, int root)
{
    return patch_force_grid_indexed(Span{&h, 1}, cell_indices, cells, count, root);
}

int pcl_comm_brush_mouse_down(pcl_compute* h, const pcl_brush_input* in, int root)
{
    return brush_event(Span{&h, 1}, in, root, true);
}

int pcl_comm_brush_mouse_moved(pcl_compute* h, const pcl_brush_input* in, int root)
{
    return brush_event(Span{&h, 1}, in, root, false);
}

int pcl_comm_step(pcl_compute* h) { return step(Span{&h, 1}); }

int pcl_comm_diagnostics_read(pcl_compute* h, pcl_diagnostics* out) { return diagnostics_read(Span{&h, 1}, out); }

int pcl_comm_diagnostics_device(pcl_compute* h, const void** dev_record)
{
    PCL_REQUIRE(h != nullptr && dev_record != nullptr && comm_of(h) != nullptr, "pcl_comm_diagnostics_device: no communicator");
    // what the caller enqueues on the handle's stream from here on sees the latest step's all-reduce finished
    if (int rc = pcl_internal::make_current(h)) return rc;
    if (int rc = order_after_side(h, comm_of(h))) return rc;
    *dev_record = comm_of(h)->d_global;
    return PCL_OK;
}

// ---- one process, all GPUs of the node -----------------------------------------

int pcl_group_destroy(pcl_group* g)
{
    if (!g) return PCL_OK;
    for (pcl_compute* h : g->h) (void)pcl_destroy(h);
    delete g;
    return PCL_OK;
}

int pcl_group_create(const int* devices, int n_devices, const pcl_particle* particles, size_t n,
                     const float (*force_grid)[4], size_t n_cells, pcl_group** out)
{
    PCL_REQUIRE(out != nullptr, "pcl_group_create: out is null");
    *out = nullptr;
    if (int rc = need_rccl()) return rc;
    PCL_REQUIRE(devices != nullptr && n_devices >= 1, "pcl_group_create: no devices");
    PCL_REQUIRE(n == 0 || particles != nullptr, "pcl_group_create: particles is null");
    PCL_REQUIRE(force_grid != nullptr && n_cells >= 1, "pcl_group_create: the force grid is empty");
    pcl_group* g = new (std::nothrow) pcl_group();
    if (!g) return fail(PCL_ERR_INVALID, "out of host memory");
    auto cleanup = [&](int rc) { pcl_group_destroy(g); return rc; };
    g->h.assign((size_t)n_devices, nullptr);
    split_ranges(g, n);
    // every handle starts with a one-cell field; the real one arrives by broadcast from rank 0
    const float one_cell[1][4] = {{0.0f, 0.0f, 0.0f, 1.0f}};
    for (int i = 0; i < n_devices; ++i) {
        const size_t first = g->first[i], cnt = g->first[i + 1] - first;
        if (int rc = pcl_create(cnt ? particles + first : nullptr, cnt, one_cell, 1, devices[i], &g->h[i])) return cleanup(rc);
    }
    std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
    ncclResult_t r = g_rccl.CommInitAll(comms.data(), n_devices, devices);
    if (r != ncclSuccess)
        return cleanup(fail(PCL_ERR_COMM, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)));
    for (int i = 0; i < n_devices; ++i)
        if (int rc = adopt(g->h[i], comms[i], i, n_devices)) {
            for (int k = i + 1; k < n_devices; ++k) (void)g_rccl.CommDestroy(comms[k]);   // not adopted yet
            return cleanup(rc);
        }
    if (int rc = update_force_grid(g->span(), force_grid, n_cells, 0)) return cleanup(rc);
    *out = g;
    return PCL_OK;
}

int pcl_group_size(const pcl_group* g) { return g ? (int)g->h.size() : 0; }

pcl_compute* pcl_group_handle(pcl_group* g, int i)
{
    return (g && i >= 0 && (size_t)i < g->h.size()) ? g->h[i] : nullptr;
}

size_t pcl_group_num_particles(const pcl_group* g) { return g ? g->n_total : 0; }

int pcl_group_upload_particles(pcl_group* g, const pcl_particle* particles, size_t n)
{
    PCL_REQUIRE(g != nullptr, "null group");
    PCL_REQUIRE(n == 0 || particles != nullptr, "pcl_group_upload_particles: particles is null");
    for (size_t i = 0; i < n; ++i)   // reject before any handle changes
        PCL_REQUIRE(particles[i].ty < PCL_NUM_TYPES, "pcl_group_upload_particles: a particle's type is >= 5");
    split_ranges(g, n);
    for (size_t i = 0; i < g->h.size(); ++i) {
        const size_t first = g->first[i], cnt = g->first[i + 1] - first;
        if (int rc = pcl_upload_particles(g->h[i], cnt ? particles + first : nullptr, cnt)) return rc;
    }
    return PCL_OK;
}

int pcl_group_generate_particles(pcl_group* g, size_t n_total, uint64_t seed, float pos_extent, float vel_extent)
{
    PCL_REQUIRE(g != nullptr, "null group");
    split_ranges(g, n_total);
    for (size_t i = 0; i < g->h.size(); ++i)
        if (int rc = pcl_generate_particles(g->h[i], g->first[i + 1] - g->first[i], seed, g->first[i], pos_extent, vel_extent))
            return rc;
    return PCL_OK;
}

int pcl_group_update_force_grid(pcl_group* g, const float (*force_grid)[4], size_t n_cells)
{
    PCL_REQUIRE(g != nullptr, "null group");
    return update_force_grid(g->span(), force_grid, n_cells, 0);
}

int pcl_group_patch_force_grid(pcl_group* g, size_t first_cell, const float (*cells)[4], size_t count)
{
    PCL_REQUIRE(g != nullptr, "null group");
    return patch_force_grid(g->span(), first_cell, cells, count, 0);
}

int pcl_group_patch_force_grid_indexed(pcl_group* g, const uint32_t* cell_indices, const float (*cells)[4], size_t count)
{
    PCL_REQUIRE(g != nullptr, "null group");
    return patch_force_grid_indexed(g->span(), cell_indices, cells, count, 0);
}

int pcl_group_update_sim_params(pcl_group* g, const pcl_sim_params* params)
{
    PCL_REQUIRE(g != nullptr, "null group");
    for (pcl_compute* h : g->h)   // replicated by value: no collective
        if (int rc = pcl_update_sim_params(h, params)) return rc;
    return PCL_OK;
}

int pcl_group_enable_step_diagnostics(pcl_group* g, int enable)
{
    PCL_REQUIRE(g != nullptr, "null group");
    for (pcl_compute* h : g->h)
        if (int rc = pcl_enable_step_diagnostics(h, enable)) return rc;
    return PCL_OK;
}

int pcl_group_step(pcl_group* g)
{
    PCL_REQUIRE(g != nullptr, "null group");
    return step(g->span());
}

int pcl_group_brush_mouse_down(pcl_group* g, const pcl_brush_input* in)
{
    PCL_REQUIRE(g != nullptr, "null group");
    return brush_event(g->span(), in, 0, true);
}

int pcl_group_brush_mouse_moved(pcl_group* g, const pcl_brush_input* in)
{
    PCL_REQUIRE(g != nullptr, "null group");
    return brush_event(g->span(), in, 0, false);
}

int pcl_group_brush_mouse_up(pcl_group* g)
{
    PCL_REQUIRE(g != nullptr, "null group");
    for (pcl_compute* h : g->h)
        if (int rc = pcl_brush_mouse_up(h)) return rc;
    return PCL_OK;
}

int pcl_group_diagnostics_read(pcl_group* g, pcl_diagnostics* out)
{
    PCL_REQUIRE(g != nullptr, "null group");
    return diagnostics_read(g->span(), out);
}

int pcl_group_read_particles(pcl_group* g, pcl_particle* out, size_t first, size_t count)
{
    PCL_REQUIRE(g != nullptr, "null group");
    if (first > g->n_total || count > g->n_total - first)
        return fail(PCL_ERR_OUT_OF_RANGE, "pcl_group_read_particles: range outside [0, n)");
    PCL_REQUIRE(count == 0 || out != nullptr, "pcl_group_read_particles: out is null");
    for (size_t i = 0; i < g->h.size(); ++i) {
        const size_t lo = std::max(first, g->first[i]), hi = std::min(first + count, g->first[i + 1]);
        if (lo < hi)
            if (int rc = pcl_read_particles(g->h[i], out + (lo - first), lo - g->first[i], hi - lo)) return rc;
    }
    return PCL_OK;
}

int pcl_group_read_force_grid(pcl_group* g, int replica, float (*out)[4], size_t first_cell, size_t count)
{
    PCL_REQUIRE(g != nullptr && replica >= 0 && (size_t)replica < g->h.size(), "pcl_group_read_force_grid: no such replica");
    return pcl_read_force_grid(g->h[(size_t)replica], out, first_cell, count);
}

int pcl_group_synchronize(pcl_group* g)
{
    PCL_REQUIRE(g != nullptr, "null group");
    for (pcl_compute* h : g->h)
        if (int rc = pcl_synchronize(h)) return rc;
    return PCL_OK;
}

}  // extern "C"
