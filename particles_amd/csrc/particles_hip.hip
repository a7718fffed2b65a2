// particles_hip.hip — C ABI of the gfx950 particle-update library
// (declarations and reference citations: include/particles_hip.h).
//
// One handle = one GPU.  Device memory per handle:
//   state[2][6]  ping-pong fp32 arrays x,y,z,vx,vy,vz   (2 * 24 B per particle)
//   ty           one byte per particle, never rewritten  (1 B per particle)
//   field        float4 per cell
//   ty[2], orig[2], slot_of   only once the particles are kept in field-tile
//                order (reorder_kernels.hpp): second type array, caller index of
//                every slot (ping-pong) and its inverse for caller-order reads
// The step reads state[cur] and writes state[cur ^ 1], then flips `cur`
// (reference ping-pong: src/compute.rs:147-180,221-227).
#include "step_kernels.hpp"
#include "reorder_kernels.hpp"
#include "brush_kernels.hpp"

#include "particles_hip.h"
#include "handle_access.hpp"
#include <hip/hip_ext.h>   // hipExtLaunchKernelGGL: a launch that stamps two events with its own begin and end

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <unordered_set>
#include <vector>

static_assert(sizeof(pcl_particle) == 48, "Particle is 48 bytes (src/main.rs:76-83)");
static_assert(offsetof(pcl_particle, vel) == 16 && offsetof(pcl_particle, ty) == 32, "Particle offsets");
static_assert(sizeof(pcl_poly7) == 32, "Poly7 is 32 bytes (src/poly7.rs:6-10)");
static_assert(sizeof(pcl_mass_wrap) == 16, "MassWrap is 16 bytes (src/main.rs:196-201)");
static_assert(sizeof(pcl_sim_params) == 912, "SimParams is 912 bytes (src/sim_params.rs:6-17)");
static_assert(offsetof(pcl_sim_params, particle_type_masses) == 800, "masses @800");
static_assert(offsetof(pcl_sim_params, force_grid_dimensions) == 880, "dims @880");
static_assert(offsetof(pcl_sim_params, delta_t) == 892, "delta_t @892");
static_assert(offsetof(pcl_sim_params, max_velocity) == 896, "max_velocity @896");
static_assert(offsetof(pcl_sim_params, bounding_volume_radius) == 900, "radius @900");
static_assert(offsetof(pcl_sim_params, cut_off_distance) == 904, "cut_off @904");
static_assert(offsetof(pcl_sim_params, distance_exponent) == 908, "exponent @908");
static_assert(sizeof(pcl_diagnostics) == 48 && sizeof(pcl::DiagOut) == 48, "diagnostics is 48 bytes");
static_assert(sizeof(pcl::ParamsBlock) == sizeof(pcl_sim_params), "device view of SimParams");
static_assert(sizeof(pcl::ParticleAoS) == sizeof(pcl_particle), "device view of Particle");

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define PCL_HIP_TRY(expr)                                                              \
    do {                                                                               \
        hipError_t pcl_e_ = (expr);                                                    \
        if (pcl_e_ != hipSuccess)                                                      \
            return fail(PCL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(pcl_e_)); \
    } while (0)

#define PCL_REQUIRE(cond, msg)                       \
    do {                                             \
        if (!(cond)) return fail(PCL_ERR_INVALID, msg); \
    } while (0)

constexpr size_t kStagingRecords = size_t(1) << 22;  // AoS staging chunk: 4 Mi records = 192 MiB
constexpr uint32_t kLdsFieldMaxBytes = 64u * 1024u;  // fields up to this size are staged in LDS
// Steps between refreshes of the tile order.  Fixed intervals on the headline workload (2^28 particles, 1024^2 field;
// whole step as a fraction of 8 TB/s): 10: 70.0 %, 12: 70.4, 16: 71.1, 20: 71.3, 24: 71.7, 32: 71.7, 40: 71.5, 48: 70.9,
// 64: 69.2 (tools/intervalsweep.sh).  With few waves in flight the plain step hardly notices the age of the order
// (2.186 ms at 10, 2.199 at 32), so the interval is set by the scatter step, which grows with it (3.6 ms at 10, 4.3 at 24,
// 4.7 at 32, 6.2 at 64: a tile's particles spread over more bins).
constexpr int kAutoReorderInterval = 24;
constexpr size_t kPairWideMinParticles = size_t(1) << 18;   // below: the sort over many cells costs more than the sweep saves
constexpr size_t kHistEntries = (size_t)pcl::kMaxReorderBins * pcl::kMaxCursorGroups;   // counters (and cursors) in d_hist
constexpr uint32_t kPairMaxCellsPerAxis = 64;          // cut-off cells per axis (beyond 20: counters in global memory)
constexpr int kAutoReorderMaxInterval = 96;           // automatic pacing never waits longer than this
constexpr size_t kAutoReorderMinFieldBytes = 16u * 1024u;        // up to here the vector L1 holds the field well enough
// The LDS-staged field needs a persistent grid, and persistent grids stream more slowly than one
// tile per block once the set is large (2^28 particles, 64 KiB field: 2.60 ms against 2.38 ms per
// step in tile order; 2^24: 0.173 against 0.167; 2^22: 0.0502 against 0.0506).
constexpr size_t kLdsFieldMaxParticles = size_t(1) << 24;
constexpr size_t kAutoReorderMinParticles = size_t(1) << 20;
constexpr size_t kPairBinnedMinParticles = size_t(1) << 15;  // automatic switch to the cell-binned pair kernel
// Most steps one launch of step_stream_fused_kernel performs (the kernel is VALU-bound well before).
constexpr uint32_t kMaxFusedSteps = 16;
constexpr size_t kPairSmallSet = size_t(1) << 18;     // up to here the pair kernel runs one wave per block
constexpr size_t kArrayStagger = 4096;                // see reserve_particles
constexpr uint32_t kFoldBlocks = 64;                 // first-level blocks of the diagnostics fold

}  // namespace

// ---- scratch of pcl_diagnostics_device_noalloc ------------------------------
// One block partial per block of the reduction, in a file-scope device array:
// every GPU the library runs on holds its own copy (4096 x 48 = 196,608 bytes)
// from the moment the code object is loaded until the process ends.  In
// host code the array's name is only a stand-in for the symbol; its address on
// a device is what hipGetSymbolAddress returns with that device current, taken
// once per device in pcl_create and kept in g_scratch_ptr.  Every call on a GPU
// shares that GPU's array: nothing in it needs to start at zero (each launch
// overwrites the entries it later reads), but two calls that overlap on one
// GPU (different handles on different streams) race.
constexpr uint32_t kNoAllocBlocks = 4096;
static __device__ __align__(256) pcl::DiagPartial scratch[kNoAllocBlocks];
namespace {
constexpr int kMaxDevices = 64;
pcl::DiagPartial* g_scratch_ptr[kMaxDevices] = {nullptr};
std::mutex g_scratch_mutex;
}  // namespace

struct pcl_compute {
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;

    size_t n = 0;
    uint64_t particles_epoch = 0;      // counts the calls that replaced the particle set
    size_t capacity = 0;
    char* slab = nullptr;              // one allocation holding every particle array
    float* state[2][6] = {{nullptr}};
    uint8_t* ty_buf[2] = {nullptr, nullptr};   // [1] is allocated with the reorder buffers
    int ty_cur = 0;
    int cur = 0;
    uint64_t steps = 0;

    // field-tile order (reorder_kernels.hpp)
    int reorder_interval = -1;         // -1 automatic, 0 never, k > 0 every k steps
    bool sorted = false;               // slots are in tile order; orig[orig_cur] holds caller indices
    uint32_t* orig[2] = {nullptr, nullptr};
    int orig_cur = 0;
    uint32_t* slot_of = nullptr;       // inverse of orig, built on demand
    bool slot_of_valid = false;
    uint32_t* d_hist = nullptr;        // kMaxReorderBins * kMaxCursorGroups counters, as many cursors, as many range ends
    size_t reorder_capacity = 0;
    uint64_t steps_since_sort = 0;
    bool hist_valid = false;           // d_hist counts the tile keys of state[cur] (left by a counting step)
    pcl::TileMap hist_tiles;           // the key function that histogram was taken with
    uint32_t hist_groups = 4;          // and its cursor groups
    uint64_t reorders = 0;
    // sort by cut-off cell for the binned pair term (stable radix sort of (cell, caller index))
    uint32_t* d_cellsort = nullptr;    // key in, key out, index in, index out: 4 x capacity
    size_t cellsort_capacity = 0;
    void* d_sort_temp = nullptr;       // the radix sort's temporary storage
    size_t sort_temp_bytes = 0;
    uint32_t* d_bin_end = nullptr;     // one past the last slot of every cell
    size_t bin_end_capacity = 0;
    // How many cut-off cells held particles at the last two cell sorts (device counters, pinned copies, one event
    // each): the sort of step k fills slot k % 2, and step k picks its sweep from what the same slot received at
    // step k - 2, after waiting for that copy -- which has long arrived, so the wait costs nothing and the choice
    // does not depend on how far the host runs ahead of the GPU.
    uint32_t* d_occupied = nullptr;    // 2 counters
    uint32_t* h_occupied = nullptr;    // pinned, 2
    hipEvent_t ev_occupied[2] = {nullptr, nullptr};
    bool occupied_pending[2] = {false, false};
    uint64_t cell_sorts = 0;
    uint32_t cells_occupied = 0;       // the latest count that was read (0: none yet)
    bool last_sweep_shared = false;    // which sweep the last cell-binned step ran
    // conditions only a running kernel can detect (kErr* bits), reported by the next synchronising call
    uint32_t* d_error = nullptr;
    uint32_t* h_error = nullptr;       // pinned landing place of the asynchronous copy
    // automatic pacing of the refreshes (reorder_interval == -1)
    int auto_interval = 0;             // current interval; 0 until first used
    uint32_t* d_runs = nullptr;        // bins touched per block, summed by the counting step
    uint32_t* h_runs = nullptr;        // pinned host copy
    hipEvent_t ev_runs = nullptr;      // recorded after the copy into h_runs
    bool runs_pending = false;
    uint64_t runs_tiles = 0;           // blocks of the counting step that produced *h_runs

    pcl::f32x4* field = nullptr;
    size_t n_cells = 0;
    // second copy of the field packed to 12 bytes per cell, what the streaming kernels fetch from
    // (fetch_cell); built before a streaming step when stale, kept equal by brush and indexed patches
    float* field3 = nullptr;
    size_t field3_capacity = 0;
    bool field3_valid = false;
    // device word the streaming kernels read: non-zero when some field component lies outside the window
    // of their packed path (step_kernels.hpp: cell_in_window); recomputed over the whole field together
    // with the packed copy, raised by the kernels that rewrite single cells
    uint32_t* d_field_flag = nullptr;
    bool field_checked = false;
    // field brush (src/cursor.rs): selection made by mouse_down, used by mouse_moved
    bool brush_down = false;
    pcl::BrushState* d_brush_state = nullptr;
    uint32_t* d_brush_index = nullptr;
    pcl::f32x4* d_brush_vector = nullptr;
    size_t brush_capacity = 0;
    size_t field_capacity = 0;

    pcl_sim_params params;
    pcl_sim_params* d_params = nullptr;
    bool params_dirty = true;

    bool force_pairs = false;
    int pair_mode = 0;                 // PCL_PAIRS_AUTO / _ALL / _BINNED
    uint32_t fuse_request = 0;                 // pcl_step_n: how many steps the caller still wants
    uint32_t last_advanced = 1;                // steps the last launch performed
    bool gathered_step = false;                // the step in progress takes its partners from ext_sources
    const pcl::f32x4* ext_sources = nullptr;   // caller's device memory, valid for that step only
    size_t n_ext_sources = 0;
    bool step_diag = false;
    bool diag_current = false;  // d_diag describes state[cur]
    int last_kernel = PCL_KERNEL_NONE;

    pcl::DiagPartial* d_partials = nullptr;
    size_t partials_capacity = 0;
    pcl::DiagOut* d_diag = nullptr;       // where reductions write the record: d_diag_own, or memory of the multi-GPU layer
    pcl::DiagOut* d_diag_own = nullptr;

    uint32_t grid_cap = 0;  // 0: default
    uint32_t variant = 0;
    // residency of the last plain streaming launch: blocks per CU the launch shape asked for (0: no cap)
    // and what the runtime's occupancy calculation grants that kernel with the LDS it was launched with
    uint32_t residency_intended = 0;
    uint32_t residency_granted = 0;

    // state of the multi-GPU layer (particles_comm.hip), released in pcl_destroy
    void* comm = nullptr;
    void (*comm_release)(void*) = nullptr;
    int (*comm_sync)(void*) = nullptr;

    // per-step launch timing (pcl_step_timing_*): one event pair per step
    std::vector<hipEvent_t> timing_events;
    std::vector<int> timing_kinds;
    size_t timing_used = 0;
    bool timing_on = false;
    // A timed step starts with its start event NOT yet on the stream (kTimingDeferred): a step that turns out to
    // be one plain kernel hands both events to that launch (hipExtLaunchKernelGGL: the kernel's own begin and end
    // stamps, no packets of their own on the stream); every other kind records the start event before its first
    // launch (timing_start_now) and do_step records the stop event behind its last.
    int timing_state = 0;
};

namespace {

int bind(const pcl_compute* h)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    PCL_HIP_TRY(hipSetDevice(h->device));
    return PCL_OK;
}

// Waits for the handle's stream and reports what its kernels flagged in the
// device error word since the last report (the word is cleared).
int sync_stream(pcl_compute* h)
{
    if (h->comm && h->comm_sync)
        if (int rc = h->comm_sync(h->comm)) return rc;
    if (!h->d_error) {
        PCL_HIP_TRY(hipStreamSynchronize(h->stream));
        return PCL_OK;
    }
    *h->h_error = 0u;
    PCL_HIP_TRY(hipMemcpyAsync(h->h_error, h->d_error, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    PCL_HIP_TRY(hipStreamSynchronize(h->stream));
    const uint32_t bits = *h->h_error;
    if (bits == 0u) return PCL_OK;
    PCL_HIP_TRY(hipMemsetAsync(h->d_error, 0, sizeof(uint32_t), h->stream));
    std::string msg = "a kernel of this handle reported:";
    if (bits & pcl::kErrScatterOverflow)
        msg += " the refresh of the particle order reserved more slots for a field tile than its histogram counted"
               " (particle state may be corrupt: upload the particles again);";
    if (bits & pcl::kErrBrushRecord)
        msg += " a cursor record read from device memory had an unknown mode or falloff (that brush event edited nothing);";
    return fail(PCL_ERR_DEVICE, msg);
}

void free_particles(pcl_compute* h)
{
    if (h->slab) (void)hipFree(h->slab);
    h->slab = nullptr;
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 6; ++a) h->state[b][a] = nullptr;
    h->ty_buf[0] = nullptr;
    h->ty_cur = 0;
    h->capacity = 0;
    if (h->ty_buf[1]) (void)hipFree(h->ty_buf[1]);
    h->ty_buf[1] = nullptr;
    for (int b = 0; b < 2; ++b) {
        if (h->orig[b]) (void)hipFree(h->orig[b]);
        h->orig[b] = nullptr;
    }
    if (h->slot_of) (void)hipFree(h->slot_of);
    h->slot_of = nullptr;
    h->reorder_capacity = 0;
    h->sorted = false;
    h->slot_of_valid = false;
}

uint8_t* cur_ty(const pcl_compute* h) { return h->ty_buf[h->ty_cur]; }

// The 13 particle arrays live in one allocation.  Array k starts at
// k * (round_up(4n, 2 MiB) + 2 MiB + 4 KiB): the 4 KiB stagger keeps the 13
// concurrent streams of a wave off the same HBM channel (all-equal low address
// bits cost up to 15 % of the streaming rate).
int reserve_particles(pcl_compute* h, size_t n)
{
    if (n <= h->capacity) return PCL_OK;
    // All work queued on the old arrays must finish before they are freed.
    PCL_HIP_TRY(hipStreamSynchronize(h->stream));
    free_particles(h);
    constexpr size_t kTwoMiB = size_t(2) << 20;
    const size_t stride = (n * sizeof(float) + kTwoMiB - 1) / kTwoMiB * kTwoMiB + kTwoMiB + kArrayStagger;
    PCL_HIP_TRY(hipMalloc((void**)&h->slab, 13 * stride));
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 6; ++a) h->state[b][a] = reinterpret_cast<float*>(h->slab + (size_t)(b * 6 + a) * stride);
    h->ty_buf[0] = reinterpret_cast<uint8_t*>(h->slab + 12 * stride);
    h->ty_cur = 0;
    h->capacity = n;
    return PCL_OK;
}

int reserve_field(pcl_compute* h, size_t n_cells)
{
    if (n_cells <= h->field_capacity) return PCL_OK;
    PCL_HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->field) (void)hipFree(h->field);
    h->field = nullptr;
    h->field_capacity = 0;
    h->field3_valid = false;
    h->field_checked = false;
    PCL_HIP_TRY(hipMalloc((void**)&h->field, n_cells * sizeof(pcl::f32x4)));
    h->field_capacity = n_cells;
    return PCL_OK;
}

// The packed copy pays once the field no longer sits in the vector L1 under the particle streams
// (tuning bit 0x400000 turns it off).
bool use_field3(const pcl_compute* h)
{
    return h->n_cells * sizeof(pcl::f32x4) > kAutoReorderMinFieldBytes && (h->variant & 0x400000u) == 0;
}

// What a streaming launch needs beside the field itself: the packed copy (*out; null when not used)
int timing_start_now(pcl_compute* h);   // below: a timed step's start event goes on the stream before its first launch

// and the window flag, both brought up to date with the field in one pass when either is stale.
int current_field3(pcl_compute* h, const float** out)
{
    *out = nullptr;
    const bool packed = use_field3(h);
    if (packed && h->field3_capacity < h->n_cells) {
        PCL_HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->field3) (void)hipFree(h->field3);
        h->field3 = nullptr;
        h->field3_capacity = 0;
        h->field3_valid = false;
        PCL_HIP_TRY(hipMalloc((void**)&h->field3, h->field_capacity * 3 * sizeof(float)));
        h->field3_capacity = h->field_capacity;
    }
    if (!h->field_checked || (packed && !h->field3_valid)) {
        if (int rc = timing_start_now(h)) return rc;
        PCL_HIP_TRY(hipMemsetAsync(h->d_field_flag, 0, sizeof(uint32_t), h->stream));
        const size_t blocks = (h->n_cells + pcl::kBlockThreads - 1) / pcl::kBlockThreads;
        hipLaunchKernelGGL(pcl::prepare_field_kernel, dim3((uint32_t)std::min<size_t>(blocks, 4096)), dim3(pcl::kBlockThreads), 0,
                           h->stream, (const pcl::f32x4*)h->field, 0ull, (unsigned long long)h->n_cells,
                           packed ? h->field3 : (float*)nullptr, h->d_field_flag);
        PCL_HIP_TRY(hipGetLastError());
        h->field_checked = true;
        if (packed) h->field3_valid = true;
    }
    if (packed) *out = h->field3;
    return PCL_OK;
}

// For the kernels that rewrite single cells: the packed copy and the flag to keep up to date, or
// null while a full pass (current_field3) is pending anyway.
float* live_field3(pcl_compute* h)
{
    if (!use_field3(h)) h->field3_valid = false;   // not kept while unused: a later use rebuilds it
    return h->field3_valid ? h->field3 : nullptr;
}
uint32_t* live_field_flag(const pcl_compute* h) { return h->field_checked ? h->d_field_flag : nullptr; }

int reserve_partials(pcl_compute* h, size_t blocks)
{
    if (blocks <= h->partials_capacity) return PCL_OK;
    PCL_HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->d_partials) (void)hipFree(h->d_partials);
    h->d_partials = nullptr;
    h->partials_capacity = 0;
    const size_t cap = std::max<size_t>(blocks, 4096);
    PCL_HIP_TRY(hipMalloc((void**)&h->d_partials, (cap + kFoldBlocks) * sizeof(pcl::DiagPartial)));
    h->partials_capacity = cap;
    return PCL_OK;
}

pcl::StateSoA soa_of(const pcl_compute* h, int which)
{
    pcl::StateSoA s;
    for (int a = 0; a < 6; ++a) s.a[a] = h->state[which][a];
    return s;
}

// The launch-uniform operands of src/compute.wgsl, each evaluated once with
// the same fp32 operation the shader applies per invocation.
pcl::StepUniforms uniforms_of(const pcl_compute* h)
{
    const pcl_sim_params& p = h->params;
    pcl::StepUniforms U;
    U.decay = expf(-p.delta_t);                                 // wgsl:151
    U.dt = p.delta_t;
    U.max_velocity = p.max_velocity;
    U.radius = p.bounding_volume_radius;
    U.two_radius = 2.0f * p.bounding_volume_radius;             // wgsl:60,63
    const float side = p.bounding_volume_radius * 2.0f;         // wgsl:147
    U.slh = side * 0.5f;                                        // wgsl:83
    U.neg_slh = -U.slh;
    U.span_a = U.slh - U.neg_slh;                               // wgsl:75
    for (int c = 0; c < 3; ++c) U.span_b[c] = (float)p.force_grid_dimensions[c] - 0.0f;  // wgsl:77
    U.dim_y = p.force_grid_dimensions[1];
    U.dim_z = p.force_grid_dimensions[2];
    U.last_cell = (uint32_t)(h->n_cells - 1);
    U.degenerate = (U.neg_slh == U.slh) ? 1u : 0u;              // wgsl:69
    for (int k = 0; k < 5; ++k) U.mass[k] = p.particle_type_masses[k].mass;
    // The packed path divides by these through their reciprocals (step_kernels.hpp: div_refined), which is
    // exact while no intermediate leaves the normal range: divisors within [2^-40, 2^40] in magnitude.
    auto in_window = [](float v) { const float a = std::fabs(v); return a >= 0x1p-40f && a <= 0x1p40f; };
    bool ok = in_window(U.span_a) && in_window(U.slh) && (h->variant & 0x4u) == 0;   // tuning bit 4: scalar path only
    U.rcp_span_a = ok ? (float)(1.0 / (double)U.span_a) : 0.0f;
    for (int k = 0; k < 5; ++k) {
        ok = ok && in_window(U.mass[k]);
        U.rcp_mass[k] = in_window(U.mass[k]) ? (float)(1.0 / (double)U.mass[k]) : 0.0f;
    }
    U.fast_div = ok ? 1u : 0u;
    return U;
}

// Smallest non-negative float s (by bit pattern, +inf included) for which
// pred(sqrtf(s)) holds, pred being monotone (false ... false true ... true);
// NaN when it never holds.  sqrtf is correctly rounded on the host and in the
// kernel, so the threshold transfers exactly.
template <typename Pred>
float smallest_square_with(Pred pred)
{
    auto from_bits = [](uint32_t b) { float f; std::memcpy(&f, &b, sizeof f); return f; };
    uint32_t lo = 0u, hi = 0x7F800000u;   // +0 .. +inf
    if (!pred(std::sqrt(from_bits(hi)))) return std::nanf("");
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (pred(std::sqrt(from_bits(mid)))) hi = mid; else lo = mid + 1;
    }
    return from_bits(lo);
}

// The pair loop adds exactly +0 to the acceleration when no distance can pass
// both cut-offs, or when every polynomial coefficient is +-0 AND every factor
// that multiplies the zero stays finite (src/compute.wgsl:126-131,139:
// dN * 0 * mass * pow(len, exponent)): the five masses are finite and
// pow(len, exponent) cannot overflow for any len in [0.001, cut_off] (0 * inf
// would be NaN).  The overflow test is done on logarithms in double precision
// with a wide margin (pow <= 1e30), so that every host language evaluates it
// identically.  Finite particle positions are assumed.
bool pair_term_inert(const pcl_sim_params& p)
{
    if (p.cut_off_distance < 0.001f) return true;
    for (int k = 0; k < PCL_NUM_POLYS; ++k)
        for (int c = 0; c < 8; ++c)
            if (p.attraction_force[k].coeffs[c] != 0.0f) return false;
    for (int k = 0; k < PCL_NUM_TYPES; ++k)
        if (!std::isfinite(p.particle_type_masses[k].mass)) return false;
    const double e = (double)p.distance_exponent;
    if (e == 0.0) return true;
    if (!std::isfinite(e)) return false;
    const double hi = std::isfinite(p.cut_off_distance) ? (double)p.cut_off_distance : 3.4028234663852886e38;
    const double reach = std::max(std::fabs(std::log(0.001)), std::fabs(std::log(hi)));
    return std::fabs(e) * reach <= 69.0;   // ln(1e30) = 69.08
}

size_t stream_tiles(size_t n)
{
    const size_t groups = n / 4;
    return std::max<size_t>(1, (groups + pcl::kStreamThreads - 1) / pcl::kStreamThreads);
}

// Small field, enough tiles per persistent block to amortise the copy, and a set small
// enough for the persistent grid not to cost more than it saves: stage the field in LDS
// (variant bit 1 disables it).
bool use_lds_field(const pcl_compute* h)
{
    const size_t field_bytes = h->n_cells * sizeof(pcl::f32x4);
    const size_t persistent = (size_t)h->num_cus * 2;
    return field_bytes <= kLdsFieldMaxBytes && (h->variant & 2u) == 0 && stream_tiles(h->n) >= persistent * 4 &&
           h->n < kLdsFieldMaxParticles;
}

// One tile per block, or (LDS-staged field, or an explicit cap) a persistent
// grid that strides over the tiles.
uint32_t stream_grid(const pcl_compute* h)
{
    size_t blocks = stream_tiles(h->n);
    if (h->grid_cap) blocks = std::min<size_t>(blocks, h->grid_cap);
    else if (use_lds_field(h)) blocks = std::min<size_t>(blocks, (size_t)h->num_cus * 2);
    return (uint32_t)std::min<size_t>(blocks, 0x7FFFFFFFu);
}

uint32_t linear_grid(const pcl_compute* h, size_t count)
{
    const size_t blocks = (count + pcl::kBlockThreads - 1) / pcl::kBlockThreads;
    return (uint32_t)std::max<size_t>(1, std::min<size_t>(blocks, (size_t)h->num_cus * 16u));
}

// Two-level fold of the first n_partials entries of d_partials (the second
// level lives in the 64 entries past partials_capacity).
int launch_finalize(pcl_compute* h, uint32_t n_partials, pcl::DiagOut* dev_out)
{
    const pcl::DiagPartial* in = h->d_partials;
    if (n_partials > 4096u) {
        pcl::DiagPartial* level2 = h->d_partials + h->partials_capacity;
        const uint32_t chunk = (n_partials + kFoldBlocks - 1) / kFoldBlocks;
        const uint32_t blocks = (n_partials + chunk - 1) / chunk;
        hipLaunchKernelGGL(pcl::diag_fold_kernel, dim3(blocks), dim3(pcl::kBlockThreads), 0, h->stream,
                           in, n_partials, chunk, level2, (pcl::DiagOut*)nullptr);
        PCL_HIP_TRY(hipGetLastError());
        in = level2;
        n_partials = blocks;
    }
    hipLaunchKernelGGL(pcl::diag_fold_kernel, dim3(1), dim3(pcl::kBlockThreads), 0, h->stream,
                       in, n_partials, n_partials, (pcl::DiagPartial*)nullptr, dev_out);
    PCL_HIP_TRY(hipGetLastError());
    return PCL_OK;
}

// The memory system of an MI355X delivers less when more waves wait on it than it takes to
// keep it busy: a copy of the step's twelve streams runs at 5.8 TB/s with every wave slot
// filled and at 6.4 TB/s with 8 waves per CU (tools/copyshape.hip).  The plain step kernel,
// whose waves also wait for a field fetch, is fastest at 16-20 waves per CU, fewer than its
// registers allow (24-28), from about 5 * 10^7 particles on (below that more waves are as fast or faster).
// The number of resident blocks is capped by asking for dynamic LDS the kernel never touches:
// `cap` blocks fit into a CU's 160 KiB, cap + 1 do not (cap <= 12).
size_t occupancy_pad(uint32_t cap, size_t used)
{
    if (cap == 0 || cap > 12) return used;
    const size_t per_block = (160u * 1024u) / cap - 1024u;
    return per_block > used ? per_block : used;
}

// More than 64 KiB of dynamic LDS (caps of one or two blocks per CU) has to be allowed per kernel
// and per device, once; remembered by the kernel's address (every instantiation is its own kernel).
int allow_large_lds_impl(const void* kernel, int device, size_t bytes)
{
    if (bytes <= 64u * 1024u) return PCL_OK;
    static std::mutex mutex;
    static std::vector<std::pair<const void*, int>> done;
    std::lock_guard<std::mutex> lock(mutex);
    for (const auto& d : done)
        if (d.first == kernel && d.second == device) return PCL_OK;
    PCL_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    done.emplace_back(kernel, device);
    return PCL_OK;
}

template <typename K>
int allow_large_lds(K kernel, int device, size_t bytes)
{
    return allow_large_lds_impl(reinterpret_cast<const void*>(kernel), device, bytes);
}

// The cap is only as good as the runtime's LDS accounting (allocation granularity, bytes the runtime
// reserves for itself): ask the occupancy calculation what it grants `kernel` in blocks of `block`
// threads with `lds_bytes` of dynamic LDS, once per (kernel, device, LDS size).
int granted_blocks_per_cu(const void* kernel, int device, int block, size_t lds_bytes, uint32_t* granted)
{
    struct Entry { const void* kernel; int device; size_t lds; uint32_t granted; };
    static std::mutex mutex;
    static std::vector<Entry> known;
    std::lock_guard<std::mutex> lock(mutex);
    for (const Entry& e : known)
        if (e.kernel == kernel && e.device == device && e.lds == lds_bytes) { *granted = e.granted; return PCL_OK; }
    int blocks = 0;
    PCL_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, block, lds_bytes));
    known.push_back(Entry{kernel, device, lds_bytes, (uint32_t)std::max(blocks, 0)});
    *granted = known.back().granted;
    return PCL_OK;
}

enum { kTimingOff = 0, kTimingDeferred = 1, kTimingStarted = 2, kTimingAttached = 3 };

// The start event of a timed step goes on the stream now, unless it is there already.
int timing_start_now(pcl_compute* h)
{
    if (h->timing_state != kTimingDeferred) return PCL_OK;
    PCL_HIP_TRY(hipEventRecord(h->timing_events[2 * h->timing_used], h->stream));
    h->timing_state = kTimingStarted;
    return PCL_OK;
}

// Which tiles the blocks of the plain streaming step take (StreamArgs::xcd_chunk; 0: block b takes tile b).
// Blocks are dealt round-robin over the 8 XCDs, so in tile order every XCD's L2 fetches every field tile that
// has particles.  That is noise while a field tile holds many blocks' worth of particles, and a second stream
// when it holds few: 7 redundant fetches of 12 bytes per cell against 49 bytes per particle.  Dealing runs of 128
// tiles so that each XCD takes 16 consecutive ones removes the redundancy, but costs the streams themselves
// something at large sets (2^28 particles, 1024^2 field: 2.35-2.39 against 2.20 ms, for any run length from 2
// up) and gains nothing at small ones (up to 2 * 10^7 particles: within 1 %), so it is used in the window where
// it was measured to pay: 3 * 2^23 .. 2^26 particles with at most 40 of them per field cell (2^25 particles:
// 1024^2 field 0.284 against 0.301 ms, 2048^2 field 0.305 against 0.359 ms; 2^26 on 2048^2: 0.588 against
// 0.622 ms).  Tuning bit 0x100 forces it on, 0x400 off.
uint32_t xcd_chunk_of(const pcl_compute* h, size_t tiles)
{
    if (tiles < 256 || (h->variant & 0x400u)) return 0u;
    if (h->variant & 0x100u) return 16u;
    const bool window = h->n >= ((size_t)3 << 23) && h->n <= ((size_t)1 << 26);
    return (window && h->sorted && h->n <= 40 * h->n_cells) ? 16u : 0u;
}

// Plain streaming step (field through the caches, no histogram) in blocks of BLOCK threads,
// one tile of 4 * BLOCK particles per block, at most `cap` blocks per CU (0: whatever fits).
// A cap that the runtime does not grant exactly (fewer blocks: registers or wave slots bind first;
// more: the LDS accounting has changed) is not used: the launch falls back to no cap, the handle
// records what was asked and what was granted (pcl_launch_residency) and the mismatch is reported
// once per process on stderr and in pcl_last_error.
template <bool DIAG, bool NT, int BLOCK>
int launch_plain_shaped(pcl_compute* h, const pcl::StreamArgs& a, const pcl::StepUniforms& U, const pcl::TileMap& T, uint32_t cap,
                        uint32_t& grid)
{
    const size_t groups = h->n / 4;
    const size_t tiles = std::max<size_t>(1, (groups + BLOCK - 1) / BLOCK);
    PCL_REQUIRE(tiles <= 0x7FFFFFFFu, "too many particles for one launch");
    pcl::StreamArgs args = a;
    size_t blocks = tiles;
    if (const uint32_t k = xcd_chunk_of(h, tiles)) {
        args.xcd_chunk = k;
        blocks = (tiles + 8 * k - 1) / (8 * k) * (8 * k);
    }
    if (DIAG)
        if (int rc = reserve_partials(h, (uint32_t)blocks)) return rc;
    args.partials = h->d_partials;
    const void* kernel = reinterpret_cast<const void*>(pcl::step_stream_kernel<DIAG, NT, false, false, BLOCK>);
    size_t pad = occupancy_pad(cap, 0);
    if (int rc = allow_large_lds_impl(kernel, h->device, pad)) return rc;
    uint32_t granted = 0;
    if (int rc = granted_blocks_per_cu(kernel, h->device, BLOCK, pad, &granted)) return rc;
    h->residency_intended = cap;
    h->residency_granted = granted;
    if (cap != 0 && granted != cap) {
        static std::atomic<bool> reported{false};
        g_last_error = "the residency cap of the streaming step asked for " + std::to_string(cap) + " blocks of " +
                       std::to_string(BLOCK) + " threads per CU with " + std::to_string(pad) +
                       " bytes of dynamic LDS, the runtime grants " + std::to_string(granted) + ": launching without the cap";
        if (!reported.exchange(true)) std::fprintf(stderr, "particles_hip: %s\n", g_last_error.c_str());
        pad = 0;
    }
    if (!DIAG && h->timing_state == kTimingDeferred) {
        // the whole step is this one launch: its own begin and end are the step's time
        hipExtLaunchKernelGGL((pcl::step_stream_kernel<DIAG, NT, false, false, BLOCK>), dim3((uint32_t)blocks), dim3(BLOCK),
                              pad, h->stream, h->timing_events[2 * h->timing_used], h->timing_events[2 * h->timing_used + 1], 0,
                              args, U, T);
        h->timing_state = kTimingAttached;
    } else {
        if (int rc = timing_start_now(h)) return rc;
        hipLaunchKernelGGL((pcl::step_stream_kernel<DIAG, NT, false, false, BLOCK>), dim3((uint32_t)blocks), dim3(BLOCK),
                           pad, h->stream, args, U, T);
    }
    PCL_HIP_TRY(hipGetLastError());
    grid = (uint32_t)blocks;   // what the fold of the partials has to cover
    return PCL_OK;
}

// Block size and residency cap of the plain streaming step (no LDS field, no histogram) for this handle.
struct PlainShape {
    uint32_t block;   // threads per block
    uint32_t cap;     // resident blocks per CU, 0 = as many as fit
};

// Tuning bits 24-27 = resident blocks per CU (15: no cap), bits 28-30 = block size (512 >> value);
// all zero = automatic.  A caller-set grid cap (persistent blocks) keeps the one launch shape.
//
// Automatic: the update's arithmetic on the packed path (step_kernels.hpp: integrate2) is short enough for
// the kernel to be fastest with few waves in flight; measured with 2^28 particles on a 1024^2 field:
// 256 threads x 3 per CU (12 waves) 2.20 ms, x 2: 2.21, x 4: 2.34, x 5: 2.32; 128 x 4: 2.23; 512 x 2: 2.36;
// 512 x 3 (what the registers allow): 2.36.  The same order from 2^20 particles up on that field (2^24: 0.147
// against 0.169 ms uncapped) and from 2^23 up on a 256^2 field (2^22: 0.0424 capped against 0.0401).
// With fused diagnostics (80 registers, a block-level fold at the end) the block count per particle matters:
// on a field beyond the L2's reach (1024^2) 512 threads x 3 per CU (2^27 particles: 1.157 ms against 1.200 at
// 256 x 4, 1.26 uncapped), on a 256^2 field 256 x 4 (2^26: 0.557 against 0.568 ms).  On the scalar path (operands outside the packed
// path's window) the arithmetic is twice as long and five blocks were the optimum (2.28 against 2.34 ms).
PlainShape plain_shape(const pcl_compute* h, bool diag, bool packed)
{
    const uint32_t cap_bits = (h->variant >> 24) & 0xFu, bsel = (h->variant >> 28) & 7u;
    if (cap_bits || bsel) return PlainShape{512u >> std::min(bsel, 3u), cap_bits > 12u ? 0u : cap_bits};
    const size_t field_bytes = h->n_cells * sizeof(pcl::f32x4);
    const bool large = h->n >= ((size_t)1 << 23) || (h->n >= ((size_t)1 << 20) && field_bytes > ((size_t)4 << 20));
    if (!large) return PlainShape{(uint32_t)pcl::kStreamThreads, 0u};
    if (!packed) return (diag || h->n < ((size_t)3 << 24)) ? PlainShape{(uint32_t)pcl::kStreamThreads, 0u} : PlainShape{256u, 5u};
    if (diag) return field_bytes > ((size_t)4 << 20) ? PlainShape{512u, 3u} : PlainShape{256u, 4u};
    return PlainShape{256u, 3u};
}

// Resident blocks per CU of the counting step (512 threads; 0: whatever fits).  2^28 particles: 2.46 ms at
// three (2.46 at two) against 2.65 ms without the cap, although the registers allow three either way;
// with fused diagnostics the cap costs (2^27: 1.44 against 1.33 ms).  Tuning bits 20-21 override.
uint32_t count_cap(const pcl_compute* h, bool diag, bool packed)
{
    const uint32_t bits = (h->variant >> 20) & 3u;
    if (bits) return bits;
    return (packed && !diag && h->n >= ((size_t)1 << 22)) ? 3u : 0u;
}

template <bool DIAG, bool NT>
int launch_stream(pcl_compute* h, const pcl::StreamArgs& a, const pcl::StepUniforms& U, const pcl::TileMap& T,
                  uint32_t& grid, bool lds, bool count)
{
    if (!lds && !count && !h->grid_cap) {
        const PlainShape shape = plain_shape(h, DIAG, U.fast_div != 0u);
        if (shape.cap || shape.block != (uint32_t)pcl::kStreamThreads) {
            switch (shape.block) {
            case 512: return launch_plain_shaped<DIAG, NT, 512>(h, a, U, T, shape.cap, grid);
            case 256: return launch_plain_shaped<DIAG, NT, 256>(h, a, U, T, shape.cap, grid);
            case 128: return launch_plain_shaped<DIAG, NT, 128>(h, a, U, T, shape.cap, grid);
            default: return launch_plain_shaped<DIAG, NT, 64>(h, a, U, T, shape.cap, grid);
            }
        }
    }
    if (int rc = timing_start_now(h)) return rc;
    if (lds) {
        const size_t bytes = (size_t)a.lds_cells * sizeof(pcl::f32x4);
        hipLaunchKernelGGL((pcl::step_stream_kernel<DIAG, NT, true, false>), dim3(grid),
                           dim3(pcl::kStreamThreads), bytes, h->stream, a, U, T);
    } else if (count) {
        const size_t bytes = occupancy_pad(count_cap(h, DIAG, U.fast_div != 0u), (size_t)T.n_bins * sizeof(uint32_t));
        if (int rc = allow_large_lds(pcl::step_stream_kernel<DIAG, NT, false, true>, h->device, bytes)) return rc;
        hipLaunchKernelGGL((pcl::step_stream_kernel<DIAG, NT, false, true>), dim3(grid),
                           dim3(pcl::kStreamThreads), bytes, h->stream, a, U, T);
    } else {
        hipLaunchKernelGGL((pcl::step_stream_kernel<DIAG, NT, false, false>), dim3(grid),
                           dim3(pcl::kStreamThreads), 0, h->stream, a, U, T);
    }
    PCL_HIP_TRY(hipGetLastError());
    return PCL_OK;
}

// ---- field-tile order (reorder_kernels.hpp) -------------------------------

uint32_t floor_log2(uint32_t v)
{
    uint32_t l = 0;
    while ((v >> (l + 1)) != 0u) ++l;
    return l;
}

// Tiles of at most 2^log2_tile_cells cells (256: 4 KiB of field), z extent first because z is the
// fastest-varying cell coordinate; tiles grow until there are at most
// kMaxReorderBins of them.
pcl::TileMap make_tile_map(const pcl_sim_params& p, int interval, uint32_t log2_tile_cells = 8, float aim = 0.5f)
{
    const uint32_t d[3] = {std::max(1u, p.force_grid_dimensions[0]), std::max(1u, p.force_grid_dimensions[1]),
                           std::max(1u, p.force_grid_dimensions[2])};
    uint32_t sh[3];
    const uint32_t lt = log2_tile_cells, half = (lt + 1u) / 2u;
    sh[2] = std::min(half, floor_log2(d[2]));
    sh[1] = std::min(std::min(half, lt - sh[2]), floor_log2(d[1]));
    sh[0] = std::min(lt - sh[2] - sh[1], floor_log2(d[0]));
    auto tiles = [&](int a) { return (uint64_t)((d[a] + (1u << sh[a]) - 1u) >> sh[a]); };
    while (tiles(0) * tiles(1) * tiles(2) > pcl::kMaxReorderBins) {
        int grow = 0;
        for (int a = 1; a < 3; ++a)
            if (tiles(a) > tiles(grow)) grow = a;
        ++sh[grow];
    }
    pcl::TileMap T;
    T.sx = sh[0]; T.sy = sh[1]; T.sz = sh[2];
    T.nty = (uint32_t)tiles(1);
    T.ntz = (uint32_t)tiles(2);
    T.n_bins = (uint32_t)(tiles(0) * tiles(1) * tiles(2));
    for (int a = 0; a < 3; ++a) T.max_ix[a] = d[a] - 1u;
    const float side = p.bounding_volume_radius * 2.0f;
    for (int a = 0; a < 3; ++a) T.inv_cell[a] = (side > 0.0f && std::isfinite(side)) ? (float)d[a] / side : 0.0f;
    // aim the order at the middle of the refresh interval
    T.lookahead = interval > 1 ? p.delta_t * (aim * (float)interval) : 0.0f;
    T.reach = 0;
    T.cell_len = 0.0f;
    return T;
}

// Cursor groups of a refresh (reorder_kernels.hpp: kMaxCursorGroups): 8 above 65536 particles per bin
// (more than 32 scatter blocks reserving in every bin), 4 below.
uint32_t cursor_groups(const pcl_compute* h, const pcl::TileMap& T)
{
    return h->n / std::max<uint32_t>(T.n_bins, 1u) > 65536u ? pcl::kMaxCursorGroups : 4u;
}

// log2 of the cells per sort tile (tuning: variant bits 4..7 override it).
// The sort tile trades the two costs of the order against each other: small tiles keep a block's field cells
// few, large tiles give the refresh long runs per bin (fewer reservations, fewer partial lines).  Since the
// particles of a bin are sub-ordered by where they came from (see do_step_impl: the first sort in two passes),
// a block's cells stay few in larger tiles too, and the plain step hardly notices the tile size (2^28
// particles on 1024^2 cells: 2.211 ms with 256-cell tiles, 2.213 with 512, 2.22 with 1024, 2.235 with 2048),
// while the scatter step does (4.28 / 3.97 / 3.83 / 3.65 ms).  The fewer particles a cell holds the shorter
// the runs and the more a larger tile pays: whole step at 2^28 (256 per cell) +0.4 % with 512 cells, at 2^27
// +0.7 %, 2^26 +1.3 %, 2^25 +2.2 %, 2^24 (16 per cell) +2.1 % with 1024 cells.  With fused diagnostics (512-thread
// blocks, whose field footprint is twice as large) one size smaller.  Never fewer than 256 tiles: their
// cursors are what the blocks' reservations spread over.
uint32_t tile_log2(const pcl_compute* h)
{
    const uint32_t v = (h->variant >> 4) & 0xFu;
    if (v) return v;
    const size_t per_cell = h->n / std::max<size_t>(h->n_cells, 1);
    uint32_t want = per_cell <= 128 ? 10u : 9u;
    if (h->step_diag) want -= 1u;
    while (want > 8u && (h->n_cells >> want) < 256u) --want;
    return want;
}

// Which point of the refresh interval the order is aimed at, as a fraction of
// the interval (tuning: variant bits 12..15 give it in tenths).
float tile_aim(const pcl_compute* h)
{
    const uint32_t v = (h->variant >> 12) & 0xFu;
    return v ? 0.1f * (float)v : 0.5f;
}

// How often the tile order is refreshed: the caller's setting, or by default
// every kAutoReorderInterval steps once the field is too large to be staged in
// LDS and the set is large enough for the fetches to matter (measured gain:
// 1.25x with a 1 MiB field, 2.1x with a 16 MiB field, uniformly random positions).
int effective_reorder_interval(const pcl_compute* h)
{
    if (h->n >= 0xFFFFFFFFull || h->n < 4096) return 0;   // caller indices are 32-bit
    if (use_lds_field(h)) return 0;                       // the whole field sits in LDS: order is irrelevant
    if (h->reorder_interval >= 0) return h->reorder_interval;
    const size_t field_bytes = h->n_cells * sizeof(pcl::f32x4);
    if (!(field_bytes > kAutoReorderMinFieldBytes && h->n >= kAutoReorderMinParticles)) return 0;
    return h->auto_interval > 0 ? h->auto_interval : kAutoReorderInterval;
}

int reserve_reorder(pcl_compute* h)
{
    if (h->reorder_capacity >= h->capacity && h->d_hist) return PCL_OK;
    PCL_HIP_TRY(hipStreamSynchronize(h->stream));
    if (!h->d_hist) PCL_HIP_TRY(hipMalloc((void**)&h->d_hist, 3 * kHistEntries * sizeof(uint32_t)));
    if (h->ty_buf[1]) (void)hipFree(h->ty_buf[1]);
    h->ty_buf[1] = nullptr;
    for (int b = 0; b < 2; ++b) {
        if (h->orig[b]) (void)hipFree(h->orig[b]);
        h->orig[b] = nullptr;
    }
    h->reorder_capacity = 0;
    PCL_HIP_TRY(hipMalloc((void**)&h->ty_buf[1], h->capacity));
    for (int b = 0; b < 2; ++b) PCL_HIP_TRY(hipMalloc((void**)&h->orig[b], h->capacity * sizeof(uint32_t)));
    h->reorder_capacity = h->capacity;
    return PCL_OK;
}

// Histogram of the tile keys of the current positions into d_hist (separate
// pass; a preceding counting step leaves the same histogram for free).
int reorder_count(pcl_compute* h, const pcl::StepUniforms& U, const pcl::TileMap& T, uint32_t groups)
{
    uint32_t* hist = h->d_hist;
    PCL_HIP_TRY(hipMemsetAsync(hist, 0, (size_t)T.n_bins * groups * sizeof(uint32_t), h->stream));
    const size_t tiles = stream_tiles(h->n);   // one tile per block: the counter group is the tile's residue
    PCL_REQUIRE(tiles <= 0x7FFFFFFFu, "too many particles to reorder");
    hipLaunchKernelGGL(pcl::reorder_count_kernel, dim3((uint32_t)tiles), dim3(pcl::kStreamThreads),
                       (size_t)T.n_bins * sizeof(uint32_t), h->stream, soa_of(h, h->cur), (unsigned long long)h->n,
                       U, T, hist, groups);
    PCL_HIP_TRY(hipGetLastError());
    return PCL_OK;
}

// Source and destination arrays of a scatter from state[cur] into state[cur ^ 1].
void reorder_args(const pcl_compute* h, pcl::ReorderArgs* a)
{
    const int src = h->cur, dst = h->cur ^ 1;
    for (int k = 0; k < 6; ++k) { a->src[k] = h->state[src][k]; a->dst[k] = h->state[dst][k]; }
    a->src_ty = h->ty_buf[h->ty_cur];
    a->dst_ty = h->ty_buf[h->ty_cur ^ 1];
    a->src_orig = h->sorted ? h->orig[h->orig_cur] : nullptr;
    a->dst_orig = h->orig[h->sorted ? (h->orig_cur ^ 1) : 0];
    a->cursor = nullptr;
    a->cursor_end = nullptr;
    a->error = h->d_error;
    a->groups = 1;
    a->n_tiles = 0;
    a->n = h->n;
}

// Defined in pair_sort.hip: stable LSD radix sort of (key, value) pairs on bits [0, end_bit) of the
// key into keys_out / values_out; the input buffers are overwritten; with temp == nullptr only the
// temporary storage size is returned.
}  // namespace
namespace pcl {
hipError_t radix_sort_pairs_u32(void* temp, size_t* temp_bytes, uint32_t* keys_in, uint32_t* keys_out,
                                uint32_t* values_in, uint32_t* values_out, size_t n, unsigned end_bit,
                                hipStream_t stream);
}
namespace {

// Reproducible sort of the particles by the cells of `C` from state[cur] into
// state[cur ^ 1] (see cell_key_kernel); leaves bin_end[b] = one past the last
// slot of cell b and commits the new order.
int restore_caller_order(pcl_compute* h);
void reorder_commit(pcl_compute* h);
void reorder_args(const pcl_compute* h, pcl::ReorderArgs* a);

int cell_sort(pcl_compute* h, const pcl::StepUniforms& U, const pcl::TileMap& C, const uint32_t** bin_end_out)
{
    if (int rc = restore_caller_order(h)) return rc;
    if (int rc = reserve_reorder(h)) return rc;
    PCL_REQUIRE(h->n <= 0x7FFFFFFFull, "too many particles for the binned pair term");
    if (h->cellsort_capacity < h->capacity) {
        PCL_HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->d_cellsort) PCL_HIP_TRY(hipFree(h->d_cellsort));
        h->d_cellsort = nullptr;
        h->cellsort_capacity = 0;
        PCL_HIP_TRY(hipMalloc((void**)&h->d_cellsort, 4 * h->capacity * sizeof(uint32_t)));
        h->cellsort_capacity = h->capacity;
    }
    if (h->bin_end_capacity < C.n_bins) {
        PCL_HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->d_bin_end) PCL_HIP_TRY(hipFree(h->d_bin_end));
        h->d_bin_end = nullptr;
        h->bin_end_capacity = 0;
        PCL_HIP_TRY(hipMalloc((void**)&h->d_bin_end, (size_t)C.n_bins * sizeof(uint32_t)));
        h->bin_end_capacity = C.n_bins;
    }
    uint32_t* key_in = h->d_cellsort;
    uint32_t* key_out = key_in + h->cellsort_capacity;
    uint32_t* idx_in = key_out + h->cellsort_capacity;
    uint32_t* idx_out = idx_in + h->cellsort_capacity;
    unsigned end_bit = 1;
    while (end_bit < 32u && (1ull << end_bit) < (unsigned long long)C.n_bins) ++end_bit;
    size_t need = 0;
    PCL_HIP_TRY(pcl::radix_sort_pairs_u32(nullptr, &need, key_in, key_out, idx_in, idx_out, h->n, end_bit, h->stream));
    if (need > h->sort_temp_bytes) {
        PCL_HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->d_sort_temp) PCL_HIP_TRY(hipFree(h->d_sort_temp));
        h->d_sort_temp = nullptr;
        h->sort_temp_bytes = 0;
        PCL_HIP_TRY(hipMalloc(&h->d_sort_temp, need));
        h->sort_temp_bytes = need;
    }
    hipLaunchKernelGGL(pcl::cell_key_kernel, dim3(linear_grid(h, h->n)), dim3(pcl::kBlockThreads), 0, h->stream,
                       soa_of(h, h->cur), (unsigned long long)h->n, U, C, key_in, idx_in);
    PCL_HIP_TRY(hipGetLastError());
    size_t temp_bytes = h->sort_temp_bytes;
    PCL_HIP_TRY(pcl::radix_sort_pairs_u32(h->d_sort_temp, &temp_bytes, key_in, key_out, idx_in, idx_out, h->n, end_bit,
                                          h->stream));
    const uint32_t end_blocks = (C.n_bins + pcl::kBlockThreads - 1) / pcl::kBlockThreads;
    if (!h->d_occupied) {
        PCL_HIP_TRY(hipMalloc((void**)&h->d_occupied, 2 * sizeof(uint32_t)));
        PCL_HIP_TRY(hipHostMalloc((void**)&h->h_occupied, 2 * sizeof(uint32_t), hipHostMallocDefault));
        for (int k = 0; k < 2; ++k) PCL_HIP_TRY(hipEventCreateWithFlags(&h->ev_occupied[k], hipEventDisableTiming));
    }
    const int slot = (int)(h->cell_sorts & 1u);
    if (h->occupied_pending[slot]) {   // what the sort before the last one counted
        PCL_HIP_TRY(hipEventSynchronize(h->ev_occupied[slot]));
        h->cells_occupied = h->h_occupied[slot];
        h->occupied_pending[slot] = false;
    }
    PCL_HIP_TRY(hipMemsetAsync(h->d_occupied + slot, 0, sizeof(uint32_t), h->stream));
    hipLaunchKernelGGL(pcl::cell_end_kernel, dim3(end_blocks), dim3(pcl::kBlockThreads), 0, h->stream,
                       (const uint32_t*)key_out, (unsigned long long)h->n, C.n_bins, h->d_bin_end, h->d_occupied + slot);
    PCL_HIP_TRY(hipGetLastError());
    PCL_HIP_TRY(hipMemcpyAsync(h->h_occupied + slot, h->d_occupied + slot, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    PCL_HIP_TRY(hipEventRecord(h->ev_occupied[slot], h->stream));
    h->occupied_pending[slot] = true;
    h->cell_sorts += 1;
    pcl::ReorderArgs ra;
    reorder_args(h, &ra);
    hipLaunchKernelGGL(pcl::cell_gather_kernel, dim3(linear_grid(h, h->n)), dim3(pcl::kBlockThreads), 0, h->stream,
                       ra, (const uint32_t*)idx_out);
    PCL_HIP_TRY(hipGetLastError());
    reorder_commit(h);
    h->steps_since_sort = 0;
    *bin_end_out = h->d_bin_end;
    return PCL_OK;
}

// Exclusive scan of d_hist into the cursors and the argument block of a
// scatter from state[cur] into state[cur ^ 1].
int reorder_prepare(pcl_compute* h, const pcl::TileMap& T, uint32_t groups, pcl::ReorderArgs* a, uint32_t* grid)
{
    uint32_t* hist = h->d_hist;
    uint32_t* cursor = h->d_hist + kHistEntries;
    uint32_t* cursor_end = h->d_hist + 2 * kHistEntries;
    hipLaunchKernelGGL(pcl::reorder_scan_kernel, dim3(1), dim3(1024), 0, h->stream, (const uint32_t*)hist, T.n_bins, groups,
                       cursor, cursor_end);
    PCL_HIP_TRY(hipGetLastError());
    reorder_args(h, a);
    a->cursor = cursor;
    a->cursor_end = cursor_end;
    a->groups = groups;
    const size_t full_tiles = (h->n / 4 + pcl::kStreamThreads - 1) / pcl::kStreamThreads;
    a->n_tiles = (uint32_t)std::min<size_t>(full_tiles, 0x7FFFFFFFu);
    const size_t g = full_tiles + ((h->n % 4) ? 1 : 0);   // one extra block carries the ragged tail
    PCL_REQUIRE(g >= 1 && g <= 0x7FFFFFFFu, "too many particles to reorder");
    *grid = (uint32_t)g;
    return PCL_OK;
}

// Bookkeeping after a scatter pass wrote state[cur ^ 1].
void reorder_commit(pcl_compute* h)
{
    h->orig_cur = h->sorted ? (h->orig_cur ^ 1) : 0;
    h->cur ^= 1;
    h->ty_cur ^= 1;
    h->sorted = true;
    h->slot_of_valid = false;
    h->hist_valid = false;
    h->reorders += 1;
}

// One counting-sort pass state[cur] -> state[cur ^ 1] by field tile of the
// current positions (no time step).
// `aim`: which point of the refresh interval the keys look at (see make_tile_map); negative = backwards.
int do_reorder(pcl_compute* h, float aim)
{
    if (h->n == 0) return PCL_OK;
    if (int rc = reserve_reorder(h)) return rc;
    const pcl::TileMap T = make_tile_map(h->params, effective_reorder_interval(h), tile_log2(h), aim);
    const pcl::StepUniforms U = uniforms_of(h);
    const uint32_t groups = cursor_groups(h, T);
    if (int rc = reorder_count(h, U, T, groups)) return rc;
    pcl::ReorderArgs a;
    uint32_t grid = 0;
    if (int rc = reorder_prepare(h, T, groups, &a, &grid)) return rc;
    hipLaunchKernelGGL(pcl::reorder_scatter_kernel, dim3(grid), dim3(pcl::kStreamThreads),
                       pcl::scatter_lds_bytes(T.n_bins), h->stream, a, U, T);
    PCL_HIP_TRY(hipGetLastError());
    reorder_commit(h);
    h->steps_since_sort = 0;
    return PCL_OK;
}

// Back to caller order (the pair kernel sums over the other particles in
// caller order, and uploads address slots by caller index).
int restore_caller_order(pcl_compute* h)
{
    if (!h->sorted) return PCL_OK;
    if (h->n) {
        const int dst = h->cur ^ 1;
        hipLaunchKernelGGL(pcl::restore_order_kernel, dim3(linear_grid(h, h->n)), dim3(pcl::kBlockThreads), 0,
                           h->stream, soa_of(h, h->cur), (const uint8_t*)h->ty_buf[h->ty_cur],
                           (const uint32_t*)h->orig[h->orig_cur], (unsigned long long)h->n, soa_of(h, dst),
                           h->ty_buf[h->ty_cur ^ 1]);
        PCL_HIP_TRY(hipGetLastError());
        h->cur = dst;
        h->ty_cur ^= 1;
    }
    h->sorted = false;
    h->slot_of_valid = false;
    h->hist_valid = false;
    return PCL_OK;
}

// Caller index -> slot map for reads; *out is null while slots are in caller order.
int caller_order_map(pcl_compute* h, const uint32_t** out)
{
    *out = nullptr;
    if (!h->sorted || h->n == 0) return PCL_OK;
    if (!h->slot_of) PCL_HIP_TRY(hipMalloc((void**)&h->slot_of, h->capacity * sizeof(uint32_t)));
    if (!h->slot_of_valid) {
        hipLaunchKernelGGL(pcl::invert_permutation_kernel, dim3(linear_grid(h, h->n)), dim3(pcl::kBlockThreads), 0,
                           h->stream, (const uint32_t*)h->orig[h->orig_cur], (unsigned long long)h->n, h->slot_of);
        PCL_HIP_TRY(hipGetLastError());
        h->slot_of_valid = true;
    }
    *out = h->slot_of;
    return PCL_OK;
}

// Uniform grid of at most 20^3 cells whose edge is at least 1.0001 x the
// cut-off, or half of that with a sweep of two cells each way (the margin absorbs
// the rounding of the reciprocal-multiply key), as a TileMap with unit tiles: the
// sort key is (cx * n + cy) * n + cz.
pcl::TileMap make_pair_cell_map(const pcl_sim_params& p, bool allow_reach2 = true, uint32_t max_axis = 20)
{
    const double side = 2.0 * (double)p.bounding_volume_radius;
    const double edge = (double)p.cut_off_distance * 1.0001;
    uint32_t n = 1, reach = 1;
    if (side > 0.0 && edge > 0.0 && std::isfinite(side) && std::isfinite(edge)) {
        // cells of at least the cut-off swept one cell each way, or of at least half the cut-off
        // swept two each way: whichever sweeps the smaller volume, (2 * reach + 1) / n per axis
        const double cap = (double)max_axis;
        const double n1 = std::min(cap, std::max(1.0, std::floor(side / edge)));
        const double n2 = std::min(cap, std::max(1.0, std::floor(2.0 * side / edge)));
        if (allow_reach2 && 5.0 / n2 < 3.0 / n1) { n = (uint32_t)n2; reach = 2; }
        else { n = (uint32_t)n1; reach = 1; }
    }
    pcl::TileMap C;
    C.sx = C.sy = C.sz = 0;
    C.nty = C.ntz = n;
    C.n_bins = n * n * n;
    for (int a = 0; a < 3; ++a) {
        C.max_ix[a] = n - 1u;
        C.inv_cell[a] = (side > 0.0 && std::isfinite(side)) ? (float)((double)n / side) : 0.0f;
    }
    C.lookahead = 0.0f;
    C.reach = reach;
    C.cell_len = (side > 0.0 && std::isfinite(side)) ? (float)(side / (double)n) : 0.0f;
    return C;
}

// The cut-off cells the binned pair step of this handle sorts by
// (tuning bits: 0x10000 one-cell reach only, 0x20000 at most 20 cells per axis).
pcl::TileMap pair_cell_map_of(const pcl_compute* h)
{
    return make_pair_cell_map(h->params, (h->variant & 0x10000u) == 0,
                              ((h->variant & 0x20000u) || h->n < kPairWideMinParticles) ? 20u : kPairMaxCellsPerAxis);
}

bool use_binned_pairs(const pcl_compute* h)
{
    if (h->pair_mode == PCL_PAIRS_ALL) return false;
    if (h->n >= 0xFFFFFFFFull || h->n < 2) return false;
    const pcl_sim_params& p = h->params;
    const bool usable = p.cut_off_distance > 0.0f && std::isfinite(p.cut_off_distance) &&
                        p.bounding_volume_radius > 0.0f && std::isfinite(p.bounding_volume_radius);
    if (!usable) return false;
    if (h->pair_mode == PCL_PAIRS_BINNED) return true;
    // automatic: only where the all-pairs sweep is the slower one
    return h->n >= kPairBinnedMinParticles && make_pair_cell_map(p).n_bins >= 27;
}

// Automatic pacing: a counting step leaves the number of (block, bin) pairs its
// output touches.  About one or two per block means the order has not decayed
// yet and the next refresh can wait twice as long; many mean it should come
// sooner again.  The value arrives asynchronously (pinned copy + event) and is
// applied to the cycles after the one that measured it.  The order never
// affects results, only how well the field fetches hit the caches.
void poll_refresh_pacing(pcl_compute* h)
{
    if (!h->runs_pending || hipEventQuery(h->ev_runs) != hipSuccess) return;
    h->runs_pending = false;
    if (h->reorder_interval >= 0 || h->runs_tiles == 0) return;
    uint64_t total = 0;
    for (uint32_t k = 0; k < pcl::kRunSlots; ++k) total += h->h_runs[k];
    const double per_block = (double)total / (double)h->runs_tiles;
    int iv = h->auto_interval > 0 ? h->auto_interval : kAutoReorderInterval;
    if (per_block <= 3.0) iv = std::min(iv * 2, kAutoReorderMaxInterval);
    else if (per_block >= 8.0) iv = std::max(iv / 2, kAutoReorderInterval);
    h->auto_interval = iv;
}

int do_step_impl(pcl_compute* h, int* kind)
{
    if (int rc = bind(h)) return rc;
    const bool inert = pair_term_inert(h->params);
    const bool gathered = h->gathered_step;
    const bool use_pairs = h->force_pairs || !inert || gathered;
    h->last_kernel = use_pairs ? PCL_KERNEL_PAIRS : PCL_KERNEL_STREAM;
    *kind = use_pairs ? PCL_STEP_PAIRS : PCL_STEP_STREAM;
    h->last_advanced = 1;
    if (h->n == 0) {
        h->cur ^= 1;
        h->steps += 1;
        h->diag_current = false;
        return PCL_OK;
    }
    PCL_REQUIRE(h->n_cells >= 1, "the force grid is empty");
    const pcl::StepUniforms U = uniforms_of(h);
    uint32_t grid = 0;
    uint32_t advanced = 1;   // steps this launch performs

    bool flipped = false;   // the launch already flipped the buffers
    if (use_pairs)
        if (int rc = timing_start_now(h)) return rc;
    if (!use_pairs) {
        poll_refresh_pacing(h);
        const int interval = effective_reorder_interval(h);
        const bool sort_now = interval > 0 && (!h->sorted || h->steps_since_sort >= (uint64_t)interval);
        // A pending histogram fixes the key function (delta_t, hence the lookahead, may change per frame).
        pcl::TileMap T = (sort_now && h->hist_valid) ? h->hist_tiles : make_tile_map(h->params, interval, tile_log2(h), tile_aim(h));
        const uint32_t groups = (sort_now && h->hist_valid) ? h->hist_groups : cursor_groups(h, T);
        // tuning bit 0x40000 (tests only): scatter with other keys than the pending histogram counted,
        // which the scatter kernel must detect and report through the error word
        if (sort_now && h->hist_valid && (h->variant & 0x40000u)) T.lookahead += 1.0f;
        if (sort_now) {
            if (int rc = timing_start_now(h)) return rc;
            // Particles in caller order (fresh upload) are sorted twice: first by the tile they were in half an
            // interval AGO, then, below, like every refresh, by the tile they will be in half an interval from
            // now.  A refresh appends each source tile's particles to their new tile in source order, so from
            // the second refresh on the particles of one tile are sub-ordered by where they came from, i.e. by
            // velocity, and neighbours in memory stay neighbours in the field for the whole interval; the extra
            // pass gives the first interval after an upload that same quality (tuning bit 0x200 skips it).
            if (!h->sorted && interval > 1 && (h->variant & 0x200u) == 0)
                if (int rc = do_reorder(h, -tile_aim(h))) return rc;
            // Update and scatter into tile order in one pass.
            if (int rc = reserve_reorder(h)) return rc;
            if (!h->hist_valid)
                if (int rc = reorder_count(h, U, T, groups)) return rc;
            pcl::StepScatterArgs sa;
            if (int rc = reorder_prepare(h, T, groups, &sa.r, &grid)) return rc;
            if (h->step_diag)
                if (int rc = reserve_partials(h, grid)) return rc;
            sa.field = h->field;
            const float* packed = nullptr;
            if (int rc = current_field3(h, &packed)) return rc;
            sa.field3 = packed;
            sa.field_flag = h->d_field_flag;
            sa.partials = h->d_partials;
            const size_t lds_bytes = pcl::scatter_lds_bytes(T.n_bins);
            if (h->step_diag)
                hipLaunchKernelGGL((pcl::step_scatter_kernel<true>), dim3(grid), dim3(pcl::kStreamThreads), lds_bytes,
                                   h->stream, sa, U, T);
            else
                hipLaunchKernelGGL((pcl::step_scatter_kernel<false>), dim3(grid), dim3(pcl::kStreamThreads), lds_bytes,
                                   h->stream, sa, U, T);
            PCL_HIP_TRY(hipGetLastError());
            reorder_commit(h);
            h->steps_since_sort = 0;   // incremented below: the new order is one step old
            flipped = true;
            *kind = PCL_STEP_STREAM_SCATTER;
        } else {
            // The step before a sort also histograms its output positions.
            const bool count = interval > 1 && h->sorted && h->steps_since_sort + 1 >= (uint64_t)interval &&
                               (h->variant & 8u) == 0;
            grid = stream_grid(h);
            if (count) {
                if (int rc = timing_start_now(h)) return rc;
                if (int rc = reserve_reorder(h)) return rc;
                // One tile per block here too: blocks dispatched together then work on
                // neighbouring tiles, which is what keeps the field fetches cache-resident
                // (a persistent grid striding over the array measured 15 % slower).
                PCL_HIP_TRY(hipMemsetAsync(h->d_hist, 0, (size_t)T.n_bins * groups * sizeof(uint32_t), h->stream));
                // one tile per block whatever the grid cap says: the counter group is the tile's residue
                const size_t tiles = stream_tiles(h->n);
                PCL_REQUIRE(tiles <= 0x7FFFFFFFu, "too many particles for one launch");
                grid = (uint32_t)tiles;
                if (!h->d_runs) {
                    PCL_HIP_TRY(hipMalloc((void**)&h->d_runs, pcl::kRunSlots * sizeof(uint32_t)));
                    PCL_HIP_TRY(hipHostMalloc((void**)&h->h_runs, pcl::kRunSlots * sizeof(uint32_t), hipHostMallocDefault));
                    PCL_HIP_TRY(hipEventCreateWithFlags(&h->ev_runs, hipEventDisableTiming));
                }
                PCL_HIP_TRY(hipMemsetAsync(h->d_runs, 0, pcl::kRunSlots * sizeof(uint32_t), h->stream));
            }
            if (h->step_diag)
                if (int rc = reserve_partials(h, grid)) return rc;
            // In place: every lane reads its particles before it writes them and no lane reads
            // another's, so the update may overwrite its input (tuning bit 0x80000; measured: 0.3 %
            // faster than the ping-pong, within the noise, so the ping-pong stays the default).
            const bool in_place = (h->variant & 0x80000u) != 0;
            const int src = h->cur, dst = in_place ? h->cur : (h->cur ^ 1);
            if (in_place) flipped = true;   // state[cur] stays the current buffer
            pcl::StreamArgs a;
            for (int k = 0; k < 6; ++k) { a.src[k] = h->state[src][k]; a.dst[k] = h->state[dst][k]; }
            a.ty = cur_ty(h);
            a.field = h->field;
            a.n = h->n;
            a.partials = h->d_partials;
            a.hist = h->d_hist;
            a.hist_groups = groups;
            const float* packed = nullptr;
            if (int rc = current_field3(h, &packed)) return rc;
            a.field3 = packed;
            a.field_flag = h->d_field_flag;
            a.xcd_chunk = 0u;
            a.runs = (count && !h->runs_pending) ? h->d_runs : nullptr;
            const bool lds = use_lds_field(h);
            a.lds_cells = lds ? (uint32_t)h->n_cells : 0u;
            // pcl_step_n: as many of the requested steps as fit before the next counting or
            // scatter step go into one launch.
            if (h->fuse_request > 1 && !count && !lds) {
                uint64_t room = h->fuse_request;
                if (interval > 0 && h->sorted) {
                    const uint64_t plain_left = (uint64_t)interval - h->steps_since_sort - ((h->variant & 8u) ? 0u : 1u);
                    room = std::min<uint64_t>(room, interval > 1 ? plain_left : 1u);
                }
                advanced = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(room, 1u), kMaxFusedSteps);
            }
            if (advanced > 1) {
                if (int rc = timing_start_now(h)) return rc;
                const size_t tiles = stream_tiles(h->n);   // the fused kernel takes exactly one tile per block
                PCL_REQUIRE(tiles <= 0x7FFFFFFFu, "too many particles for one launch");
                grid = (uint32_t)tiles;
                if (h->step_diag)
                    if (int rc = reserve_partials(h, grid)) return rc;
                if (h->step_diag)
                    hipLaunchKernelGGL((pcl::step_stream_fused_kernel<true>), dim3(grid), dim3(pcl::kStreamThreads), 0,
                                       h->stream, a, U, advanced);
                else
                    hipLaunchKernelGGL((pcl::step_stream_fused_kernel<false>), dim3(grid), dim3(pcl::kStreamThreads), 0,
                                       h->stream, a, U, advanced);
                PCL_HIP_TRY(hipGetLastError());
                h->hist_valid = false;
                *kind = PCL_STEP_STREAM_FUSED;
            } else {
            const bool nt = (h->variant & 1u) == 0;
            int rc;
            if (h->step_diag) rc = nt ? launch_stream<true, true>(h, a, U, T, grid, lds, count) : launch_stream<true, false>(h, a, U, T, grid, lds, count);
            else              rc = nt ? launch_stream<false, true>(h, a, U, T, grid, lds, count) : launch_stream<false, false>(h, a, U, T, grid, lds, count);
            if (rc) return rc;
            h->hist_valid = count;
            if (count) { h->hist_tiles = T; h->hist_groups = groups; }
            if (count) *kind = PCL_STEP_STREAM_COUNT;
            if (count && a.runs) {
                PCL_HIP_TRY(hipMemcpyAsync(h->h_runs, h->d_runs, pcl::kRunSlots * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
                PCL_HIP_TRY(hipEventRecord(h->ev_runs, h->stream));
                h->runs_pending = true;
                h->runs_tiles = grid;
            }
            }
        }
    } else if (!gathered && use_binned_pairs(h)) {
        // Cell-binned pair term: sort by cut-off cell, then sweep the 27 neighbour cells.
        h->hist_valid = false;
        if (h->params_dirty) {
            PCL_HIP_TRY(hipMemcpyAsync(h->d_params, &h->params, sizeof(pcl_sim_params),
                                       hipMemcpyHostToDevice, h->stream));
            PCL_HIP_TRY(hipStreamSynchronize(h->stream));
            h->params_dirty = false;
        }
        const pcl::TileMap C = pair_cell_map_of(h);
        const uint32_t* bin_end = nullptr;   // bin_end[b]: one past the last slot of cell b
        if (int rc = cell_sort(h, U, C, &bin_end)) return rc;
        // Two sweeps with the same results (step_kernels.hpp): one particle per lane with its own candidate ranges,
        // or one wave per 64 slots with the candidates shared through LDS.  The shared sweep wins where cells are
        // crowded -- the reference's start-up cluster, 50 000 particles in a 2-unit cube of the 20-unit box:
        // 3.4 against 9.0 ms at a cut-off of 0.3, 7.4 against 8.4 ms at 1.0 -- and loses 10-25 % where a cell
        // holds tens to a few hundred particles (uniform sets).  Picked by the particles per OCCUPIED cell as
        // counted two sorts ago (512 and more: shared); tuning bits 0x800000 / 0x80000000 force the per-lane / the
        // shared sweep.  Until the first count is in: per lane.
        bool per_lane = h->cells_occupied == 0 || h->n / h->cells_occupied < 512;
        if (h->variant & 0x800000u) per_lane = true;
        if (h->variant & 0x80000000u) per_lane = false;
        h->last_sweep_shared = !per_lane;
        const size_t per_block = per_lane ? (size_t)pcl::kBlockThreads : 64;
        const size_t blocks = (h->n + per_block - 1) / per_block;
        PCL_REQUIRE(blocks <= 0x7FFFFFFFu, "too many particles for the pair kernel");
        grid = (uint32_t)blocks;
        if (h->step_diag)
            if (int rc = reserve_partials(h, grid)) return rc;
        const int src = h->cur, dst = h->cur ^ 1;
        pcl::PairArgs a;
        for (int k = 0; k < 6; ++k) { a.src[k] = h->state[src][k]; a.dst[k] = h->state[dst][k]; }
        a.ty = cur_ty(h);
        a.field = h->field;
        a.params = reinterpret_cast<const pcl::ParamsBlock*>(h->d_params);
        a.n = h->n;
        a.partials = h->d_partials;
        a.s_near = smallest_square_with([](float len) { return !(len < 0.001f); });
        const float cut = h->params.cut_off_distance;
        a.s_far = smallest_square_with([cut](float len) { return len > cut; });
        if (std::isnan(a.s_far)) a.s_far = INFINITY;   // no distance is too far: see no_candidate() in step_kernels.hpp
        if (!per_lane) {
            if (h->step_diag)
                hipLaunchKernelGGL((pcl::step_pairs_binned_wave_kernel<true>), dim3(grid), dim3(64), 0, h->stream, a, U, C, bin_end);
            else
                hipLaunchKernelGGL((pcl::step_pairs_binned_wave_kernel<false>), dim3(grid), dim3(64), 0, h->stream, a, U, C, bin_end);
        } else if (h->step_diag)
            hipLaunchKernelGGL((pcl::step_pairs_binned_kernel<true>), dim3(grid), dim3(pcl::kBlockThreads), 0,
                               h->stream, a, U, C, bin_end);
        else
            hipLaunchKernelGGL((pcl::step_pairs_binned_kernel<false>), dim3(grid), dim3(pcl::kBlockThreads), 0,
                               h->stream, a, U, C, bin_end);
        PCL_HIP_TRY(hipGetLastError());
        *kind = PCL_STEP_PAIRS_BINNED;
    } else {
        if (int rc = restore_caller_order(h)) return rc;   // the pair sums run in caller order
        h->hist_valid = false;
        if (h->params_dirty) {
            PCL_HIP_TRY(hipMemcpyAsync(h->d_params, &h->params, sizeof(pcl_sim_params),
                                       hipMemcpyHostToDevice, h->stream));
            // The source is a member of the handle; wait so later edits cannot race the copy.
            PCL_HIP_TRY(hipStreamSynchronize(h->stream));
            h->params_dirty = false;
        }
        // One wave per block while the set is small (more CUs busy), four otherwise.
        const bool small = h->n <= kPairSmallSet;
        const size_t pb = small ? 64 : pcl::kBlockThreads;
        const size_t blocks = (h->n + pb - 1) / pb;
        PCL_REQUIRE(blocks <= 0x7FFFFFFFu, "too many particles for the pair kernel");
        grid = (uint32_t)blocks;
        if (h->step_diag)
            if (int rc = reserve_partials(h, grid)) return rc;
        const int src = h->cur, dst = h->cur ^ 1;
        pcl::PairArgs a;
        for (int k = 0; k < 6; ++k) { a.src[k] = h->state[src][k]; a.dst[k] = h->state[dst][k]; }
        a.ty = cur_ty(h);
        a.field = h->field;
        a.params = reinterpret_cast<const pcl::ParamsBlock*>(h->d_params);
        a.n = h->n;
        a.partials = h->d_partials;
        a.s_near = smallest_square_with([](float len) { return !(len < 0.001f); });
        const float cut = h->params.cut_off_distance;
        a.s_far = smallest_square_with([cut](float len) { return len > cut; });
        if (std::isnan(a.s_far)) a.s_far = INFINITY;   // no distance is too far: see no_candidate() in step_kernels.hpp
        a.sources = h->ext_sources;
        a.n_sources = h->n_ext_sources;
        if (gathered) {
            *kind = PCL_STEP_PAIRS_GATHERED;
            if (small) {
                if (h->step_diag)
                    hipLaunchKernelGGL((pcl::step_pairs_kernel<true, 64, true>), dim3(grid), dim3(64), 0, h->stream, a, U);
                else
                    hipLaunchKernelGGL((pcl::step_pairs_kernel<false, 64, true>), dim3(grid), dim3(64), 0, h->stream, a, U);
            } else {
                if (h->step_diag)
                    hipLaunchKernelGGL((pcl::step_pairs_kernel<true, pcl::kBlockThreads, true>), dim3(grid),
                                       dim3(pcl::kBlockThreads), 0, h->stream, a, U);
                else
                    hipLaunchKernelGGL((pcl::step_pairs_kernel<false, pcl::kBlockThreads, true>), dim3(grid),
                                       dim3(pcl::kBlockThreads), 0, h->stream, a, U);
            }
        } else if (small) {
            if (h->step_diag)
                hipLaunchKernelGGL((pcl::step_pairs_kernel<true, 64>), dim3(grid), dim3(64), 0, h->stream, a, U);
            else
                hipLaunchKernelGGL((pcl::step_pairs_kernel<false, 64>), dim3(grid), dim3(64), 0, h->stream, a, U);
        } else {
            if (h->step_diag)
                hipLaunchKernelGGL((pcl::step_pairs_kernel<true, pcl::kBlockThreads>), dim3(grid),
                                   dim3(pcl::kBlockThreads), 0, h->stream, a, U);
            else
                hipLaunchKernelGGL((pcl::step_pairs_kernel<false, pcl::kBlockThreads>), dim3(grid),
                                   dim3(pcl::kBlockThreads), 0, h->stream, a, U);
        }
        PCL_HIP_TRY(hipGetLastError());
    }

    if (!flipped) h->cur ^= 1;
    h->steps += advanced;
    h->steps_since_sort += advanced;
    h->last_advanced = advanced;
    if (h->step_diag) {
        if (int rc = launch_finalize(h, grid, h->d_diag)) return rc;
        h->diag_current = true;
    } else {
        h->diag_current = false;
    }
    return PCL_OK;
}

// One step; with step timing on, bracketed by an event pair on the stream.
int do_step(pcl_compute* h)
{
    int kind = PCL_STEP_NONE;
    const bool timed = h && h->timing_on && h->timing_used < h->timing_kinds.size();
    if (timed) {
        if (int rc = bind(h)) return rc;
        h->timing_state = kTimingDeferred;
        // tuning bit 0x800: an event record on either side of every step, whatever its kind
        if (h->variant & 0x800u)
            if (int rc = timing_start_now(h)) return rc;
    }
    const int rc = do_step_impl(h, &kind);
    if (timed) {
        const int state = h->timing_state;
        h->timing_state = kTimingOff;
        if (state == kTimingDeferred)   // nothing was launched (no particles, or an error on the way)
            PCL_HIP_TRY(hipEventRecord(h->timing_events[2 * h->timing_used], h->stream));
        if (state != kTimingAttached) PCL_HIP_TRY(hipEventRecord(h->timing_events[2 * h->timing_used + 1], h->stream));
        h->timing_kinds[h->timing_used] = rc == PCL_OK ? kind : PCL_STEP_NONE;
        h->timing_used += 1;
    }
    return rc;
}

// Reduce diagnostics of state[cur] into h->d_diag unless already current.
int ensure_diag(pcl_compute* h)
{
    if (h->diag_current) return PCL_OK;
    const uint32_t grid = h->n ? linear_grid(h, h->n) : 1u;
    if (int rc = reserve_partials(h, grid)) return rc;
    hipLaunchKernelGGL(pcl::diag_reduce_kernel, dim3(grid), dim3(pcl::kBlockThreads), 0, h->stream,
                       soa_of(h, h->cur), cur_ty(h), (unsigned long long)h->n, uniforms_of(h), h->d_partials);
    PCL_HIP_TRY(hipGetLastError());
    if (int rc = launch_finalize(h, grid, h->d_diag)) return rc;
    h->diag_current = true;
    return PCL_OK;
}

int check_range(size_t first, size_t count, size_t n, const char* what)
{
    if (first > n || count > n - first)
        return fail(PCL_ERR_OUT_OF_RANGE, std::string(what) + ": range outside [0, n)");
    return PCL_OK;
}

int upload_field(pcl_compute* h, size_t first_cell, const void* src, size_t count, bool from_device)
{
    if (count == 0) return PCL_OK;
    h->field3_valid = false;
    h->field_checked = false;
    PCL_HIP_TRY(hipMemcpyAsync(h->field + first_cell, src, count * sizeof(pcl::f32x4),
                               from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                               h->stream));
    if (!from_device) PCL_HIP_TRY(hipStreamSynchronize(h->stream));  // caller's buffer is only valid during the call
    return PCL_OK;
}

// Geometry of the field as the host-side Grid sees it: dimensions from the
// parameter block, bounds (-r, 2r) as in src/sim_params.rs:33-45.
int field_geometry(const pcl_compute* h, pcl::FieldGeometry* G)
{
    const uint32_t* d = h->params.force_grid_dimensions;
    PCL_REQUIRE((uint64_t)d[0] * d[1] * d[2] == (uint64_t)h->n_cells,
                "force_grid_dimensions of the parameters do not match the uploaded field");
    PCL_REQUIRE(h->n_cells <= 0xFFFFFFFFull, "field too large for the brush");
    G->nx = d[0]; G->ny = d[1]; G->nz = d[2];
    G->left = -h->params.bounding_volume_radius;
    G->dir = 2.0f * h->params.bounding_volume_radius;
    return PCL_OK;
}

int reserve_brush(pcl_compute* h)
{
    if (!h->d_brush_state) PCL_HIP_TRY(hipMalloc((void**)&h->d_brush_state, sizeof(pcl::BrushState)));
    if (h->brush_capacity >= h->n_cells) return PCL_OK;
    PCL_HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->d_brush_index) PCL_HIP_TRY(hipFree(h->d_brush_index));
    if (h->d_brush_vector) PCL_HIP_TRY(hipFree(h->d_brush_vector));
    h->d_brush_index = nullptr;
    h->d_brush_vector = nullptr;
    h->brush_capacity = 0;
    PCL_HIP_TRY(hipMalloc((void**)&h->d_brush_index, h->n_cells * sizeof(uint32_t)));
    PCL_HIP_TRY(hipMalloc((void**)&h->d_brush_vector, h->n_cells * sizeof(pcl::f32x4)));
    h->brush_capacity = h->n_cells;
    return PCL_OK;
}

pcl::BrushInput brush_input_of(const pcl_brush_input* in)
{
    static_assert(sizeof(pcl_brush_input) == sizeof(pcl::BrushInput), "brush record layout");
    pcl::BrushInput r;
    std::memset(&r, 0, sizeof r);
    if (in) std::memcpy(&r, in, sizeof r);
    return r;
}

int brush_mouse_down(pcl_compute* h, const pcl_brush_input* host_in, const void* dev_in)
{
    if (int rc = bind(h)) return rc;
    pcl::FieldGeometry G;
    if (int rc = field_geometry(h, &G)) return rc;
    if (int rc = reserve_brush(h)) return rc;
    PCL_HIP_TRY(hipMemsetAsync(h->d_brush_state, 0, sizeof(pcl::BrushState), h->stream));
    hipLaunchKernelGGL(pcl::brush_select_kernel, dim3(linear_grid(h, h->n_cells)), dim3(pcl::kBlockThreads), 0,
                       h->stream, brush_input_of(host_in), static_cast<const pcl::BrushInput*>(dev_in), G,
                       (uint32_t)h->n_cells, (const pcl::f32x4*)h->field, h->d_brush_state, h->d_brush_index,
                       h->d_brush_vector, h->d_error);
    PCL_HIP_TRY(hipGetLastError());
    h->brush_down = true;
    return PCL_OK;
}

int brush_mouse_moved(pcl_compute* h, const pcl_brush_input* host_in, const void* dev_in)
{
    if (!h->brush_down) return PCL_OK;   // src/cursor.rs:82: nothing happens without a mouse-down
    if (int rc = bind(h)) return rc;
    pcl::FieldGeometry G;
    if (int rc = field_geometry(h, &G)) return rc;
    PCL_REQUIRE(h->brush_capacity >= h->n_cells, "the field was replaced by a larger one since mouse_down");
    // the selection size lives on the device: a fixed grid strides over it
    const uint32_t grid = std::min<uint32_t>(linear_grid(h, h->n_cells), 1024u);
    hipLaunchKernelGGL(pcl::brush_apply_kernel, dim3(grid), dim3(pcl::kBlockThreads), 0, h->stream,
                       brush_input_of(host_in), static_cast<const pcl::BrushInput*>(dev_in), G, h->field,
                       live_field3(h), live_field_flag(h), (const pcl::BrushState*)h->d_brush_state,
                       (const uint32_t*)h->d_brush_index, (const pcl::f32x4*)h->d_brush_vector, h->d_error);
    PCL_HIP_TRY(hipGetLastError());
    return PCL_OK;
}

// The listed cells into the 16-byte field, then (a second launch: the first must have finished) the
// packed copy of those cells from the field as it ended up.
hipError_t launch_indexed_patch(pcl_compute* h, const uint32_t* d_index, const pcl::f32x4* d_cells, size_t count)
{
    hipLaunchKernelGGL(pcl::patch_field_kernel, dim3(linear_grid(h, count)), dim3(pcl::kBlockThreads), 0, h->stream,
                       h->field, live_field_flag(h), (unsigned long long)h->n_cells, d_index, d_cells,
                       (unsigned long long)count);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    if (float* packed = live_field3(h)) {
        hipLaunchKernelGGL(pcl::repack_cells_kernel, dim3(linear_grid(h, count)), dim3(pcl::kBlockThreads), 0, h->stream,
                           (const pcl::f32x4*)h->field, packed, (unsigned long long)h->n_cells, d_index,
                           (unsigned long long)count);
        return hipGetLastError();
    }
    return hipSuccess;
}

}  // namespace

namespace pcl_internal {

int fail(int code, const std::string& message) { return ::fail(code, message); }
int make_current(const pcl_compute* h) { return bind(h); }
int device_of(const pcl_compute* h) { return h->device; }
hipStream_t stream_of(const pcl_compute* h) { return h->stream; }
void* field_of(pcl_compute* h)
{
    h->field3_valid = false;   // handed out for writing (collectives deliver cells straight into it)
    h->field_checked = false;
    return h->field;
}
size_t cells_of(const pcl_compute* h) { return h->n_cells; }
bool pair_term_active(const pcl_compute* h) { return !pair_term_inert(h->params); }
bool step_diagnostics_enabled(const pcl_compute* h) { return h->step_diag; }
uint64_t particles_epoch(const pcl_compute* h) { return h->particles_epoch; }
void set_comm(pcl_compute* h, void* comm, void (*release)(void*), int (*sync)(void*))
{
    h->comm = comm;
    h->comm_release = release;
    h->comm_sync = sync;
}
void* comm_of(const pcl_compute* h) { return h->comm; }

int field_for_write(pcl_compute* h, size_t n_cells, void** dev_field)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(n_cells >= 1 && n_cells <= 0xFFFFFFFFull, "the force grid must have 1 .. 2^32 - 1 cells");
    if (int rc = reserve_field(h, n_cells)) return rc;
    if (n_cells != h->n_cells) h->brush_down = false;   // a selection indexes the field it was made on
    h->n_cells = n_cells;
    h->field3_valid = false;   // the caller is about to write the cells
    h->field_checked = false;
    *dev_field = h->field;
    return PCL_OK;
}

void set_diag_target(pcl_compute* h, void* dev_record)
{
    pcl::DiagOut* target = dev_record ? static_cast<pcl::DiagOut*>(dev_record) : h->d_diag_own;
    if (target == h->d_diag) return;
    h->d_diag = target;
    h->diag_current = false;   // the record of the current state, if any, is in the other place
}

int diagnostics_on_device(pcl_compute* h, const void** dev_record)
{
    if (int rc = bind(h)) return rc;
    if (int rc = ensure_diag(h)) return rc;
    *dev_record = h->d_diag;
    return PCL_OK;
}

}  // namespace pcl_internal

extern "C" {

const char* pcl_last_error(void) { return g_last_error.c_str(); }

int pcl_device_count(void)
{
    int n = 0;
    PCL_HIP_TRY(hipGetDeviceCount(&n));
    return n;
}

void pcl_sim_params_default(pcl_sim_params* out)
{
    if (!out) return;
    std::memset(out, 0, sizeof *out);
    for (int k = 0; k < PCL_NUM_TYPES; ++k) out->particle_type_masses[k].mass = 1.0f;
    out->force_grid_dimensions[0] = out->force_grid_dimensions[1] = out->force_grid_dimensions[2] = 10;
    out->delta_t = 0.0f;
    out->max_velocity = 100.0f;
    out->bounding_volume_radius = 10.0f;
    out->cut_off_distance = 1.0f;
    out->distance_exponent = 0.0f;
}

int pcl_destroy(pcl_compute* h)
{
    if (!h) return PCL_OK;
    (void)hipSetDevice(h->device);
    if (h->comm && h->comm_release) h->comm_release(h->comm);   // while the stream still exists
    h->comm = nullptr;
    h->comm_sync = nullptr;
    (void)hipStreamSynchronize(h->stream);   // a null handle names the device's null stream
    if (h->own_stream && h->own_stream != h->stream) (void)hipStreamSynchronize(h->own_stream);
    free_particles(h);
    if (h->field) (void)hipFree(h->field);
    if (h->field3) (void)hipFree(h->field3);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_partials) (void)hipFree(h->d_partials);
    if (h->d_diag_own) (void)hipFree(h->d_diag_own);
    if (h->d_hist) (void)hipFree(h->d_hist);
    if (h->d_runs) (void)hipFree(h->d_runs);
    if (h->d_cellsort) (void)hipFree(h->d_cellsort);
    if (h->d_sort_temp) (void)hipFree(h->d_sort_temp);
    if (h->d_bin_end) (void)hipFree(h->d_bin_end);
    if (h->d_occupied) (void)hipFree(h->d_occupied);
    if (h->h_occupied) (void)hipHostFree(h->h_occupied);
    for (hipEvent_t e : h->ev_occupied)
        if (e) (void)hipEventDestroy(e);
    if (h->d_error) (void)hipFree(h->d_error);
    if (h->d_field_flag) (void)hipFree(h->d_field_flag);
    if (h->h_error) (void)hipHostFree(h->h_error);
    if (h->h_runs) (void)hipHostFree(h->h_runs);
    if (h->ev_runs) (void)hipEventDestroy(h->ev_runs);
    if (h->d_brush_state) (void)hipFree(h->d_brush_state);
    if (h->d_brush_index) (void)hipFree(h->d_brush_index);
    if (h->d_brush_vector) (void)hipFree(h->d_brush_vector);
    for (hipEvent_t e : h->timing_events) (void)hipEventDestroy(e);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return PCL_OK;
}

int pcl_create(const pcl_particle* particles, size_t n, const float (*force_grid)[4], size_t n_cells,
               int device, pcl_compute** out)
{
    PCL_REQUIRE(out != nullptr, "pcl_create: out is null");
    *out = nullptr;
    PCL_REQUIRE(n == 0 || particles != nullptr, "pcl_create: particles is null");
    PCL_REQUIRE(force_grid != nullptr && n_cells >= 1, "pcl_create: the force grid is empty");
    PCL_REQUIRE(n_cells <= 0xFFFFFFFFull, "pcl_create: the force grid has more than 2^32 - 1 cells");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PCL_ERR_NO_DEVICE, "pcl_create: no HIP device is visible");
    PCL_REQUIRE(device >= 0 && device < count, "pcl_create: device ordinal out of range");
    PCL_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    PCL_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PCL_ERR_NO_DEVICE, std::string("pcl_create: device is ") + prop.gcnArchName +
                                           ", this library is built for gfx950 only");

    if (device < kMaxDevices) {
        // this device's copy of the no-allocation scratch array (the device is current)
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            pcl::DiagPartial* ptr = nullptr;
            PCL_HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
            g_scratch_ptr[device] = ptr;
        }
    }

    pcl_compute* h = new (std::nothrow) pcl_compute();
    if (!h) return fail(PCL_ERR_INVALID, "pcl_create: out of host memory");
    h->device = device;
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    pcl_sim_params_default(&h->params);  // src/compute.rs:33: parameters start as SimParams::new()

    auto cleanup = [&](int rc) { pcl_destroy(h); return rc; };
#define PCL_CREATE_TRY(expr)                                                                    \
    do {                                                                                        \
        hipError_t pcl_e_ = (expr);                                                             \
        if (pcl_e_ != hipSuccess)                                                               \
            return cleanup(fail(PCL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(pcl_e_))); \
    } while (0)
    PCL_CREATE_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    PCL_CREATE_TRY(hipEventCreate(&h->ev_start));
    PCL_CREATE_TRY(hipEventCreate(&h->ev_stop));
    PCL_CREATE_TRY(hipMalloc((void**)&h->d_params, sizeof(pcl_sim_params)));
    PCL_CREATE_TRY(hipMalloc((void**)&h->d_diag_own, sizeof(pcl::DiagOut)));
    h->d_diag = h->d_diag_own;
    PCL_CREATE_TRY(hipMalloc((void**)&h->d_field_flag, sizeof(uint32_t)));
    PCL_CREATE_TRY(hipMalloc((void**)&h->d_error, sizeof(uint32_t)));
    PCL_CREATE_TRY(hipMemset(h->d_error, 0, sizeof(uint32_t)));
    PCL_CREATE_TRY(hipHostMalloc((void**)&h->h_error, sizeof(uint32_t), hipHostMallocDefault));
#undef PCL_CREATE_TRY
    if (int rc = reserve_partials(h, 1)) return cleanup(rc);
    if (int rc = pcl_update_force_grid(h, force_grid, n_cells)) return cleanup(rc);
    if (int rc = pcl_upload_particles(h, particles, n)) return cleanup(rc);
    *out = h;
    return PCL_OK;
}

int pcl_upload_particles(pcl_compute* h, const pcl_particle* particles, size_t n)
{
    if (int rc = bind(h)) return rc;
    // a rejected call leaves the handle as it was
    PCL_REQUIRE(n == 0 || particles != nullptr, "pcl_upload_particles: particles is null");
    for (size_t i = 0; i < n; ++i)
        if (particles[i].ty >= PCL_NUM_TYPES)
            return fail(PCL_ERR_INVALID, "pcl_upload_particles: particle " + std::to_string(i) +
                                             " has type " + std::to_string(particles[i].ty) + " (must be < 5)");
    if (int rc = reserve_particles(h, n)) return rc;
    h->sorted = false;          // a fresh set is in caller order
    h->slot_of_valid = false;
    h->hist_valid = false;
    h->n = n;
    ++h->particles_epoch;
    h->diag_current = false;
    if (n == 0) return PCL_OK;

    const size_t chunk = std::min(n, kStagingRecords);
    pcl::ParticleAoS* staging = nullptr;
    PCL_HIP_TRY(hipMalloc((void**)&staging, chunk * sizeof(pcl::ParticleAoS)));
    int rc = PCL_OK;
    for (size_t first = 0; first < n && rc == PCL_OK; first += chunk) {
        const size_t cnt = std::min(chunk, n - first);
        hipError_t e = hipMemcpyAsync(staging, particles + first, cnt * sizeof(pcl_particle),
                                      hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pcl::unpack_aos_kernel, dim3(linear_grid(h, cnt)), dim3(pcl::kBlockThreads),
                               0, h->stream, staging, (unsigned long long)cnt, soa_of(h, h->cur), cur_ty(h),
                               (unsigned long long)first);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(PCL_ERR_HIP, std::string("pcl_upload_particles: ") + hipGetErrorString(e));
    }
    (void)hipFree(staging);
    return rc;
}

int pcl_upload_particles_soa(pcl_compute* h, size_t n, const float* x, const float* y, const float* z,
                             const float* vx, const float* vy, const float* vz, const uint8_t* ty)
{
    if (int rc = bind(h)) return rc;
    const float* src[6] = {x, y, z, vx, vy, vz};
    if (n) {
        for (int a = 0; a < 6; ++a) PCL_REQUIRE(src[a] != nullptr, "pcl_upload_particles_soa: null array");
        PCL_REQUIRE(ty != nullptr, "pcl_upload_particles_soa: ty is null");
        for (size_t i = 0; i < n; ++i)
            if (ty[i] >= PCL_NUM_TYPES)
                return fail(PCL_ERR_INVALID, "pcl_upload_particles_soa: particle " + std::to_string(i) +
                                                 " has type " + std::to_string(ty[i]) + " (must be < 5)");
    }
    if (int rc = reserve_particles(h, n)) return rc;
    h->sorted = false;          // a fresh set is in caller order
    h->slot_of_valid = false;
    h->hist_valid = false;
    h->n = n;
    ++h->particles_epoch;
    h->diag_current = false;
    if (n == 0) return PCL_OK;
    for (int a = 0; a < 6; ++a)
        PCL_HIP_TRY(hipMemcpyAsync(h->state[h->cur][a], src[a], n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    PCL_HIP_TRY(hipMemcpyAsync(cur_ty(h), ty, n, hipMemcpyHostToDevice, h->stream));
    PCL_HIP_TRY(hipStreamSynchronize(h->stream));
    return PCL_OK;
}

int pcl_generate_particles(pcl_compute* h, size_t n, uint64_t seed, uint64_t first_global_index,
                           float pos_extent, float vel_extent)
{
    if (int rc = bind(h)) return rc;
    h->sorted = false;          // a fresh set is in caller order
    h->slot_of_valid = false;
    h->hist_valid = false;
    if (int rc = reserve_particles(h, n)) return rc;
    h->n = n;
    ++h->particles_epoch;
    h->diag_current = false;
    if (n == 0) return PCL_OK;
    hipLaunchKernelGGL(pcl::generate_kernel, dim3(linear_grid(h, n)), dim3(pcl::kBlockThreads), 0, h->stream,
                       soa_of(h, h->cur), cur_ty(h), (unsigned long long)n, (unsigned long long)seed,
                       (unsigned long long)first_global_index, pos_extent, vel_extent);
    PCL_HIP_TRY(hipGetLastError());
    return PCL_OK;
}

int pcl_update_force_grid(pcl_compute* h, const float (*force_grid)[4], size_t n_cells)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(force_grid != nullptr && n_cells >= 1, "pcl_update_force_grid: the force grid is empty");
    PCL_REQUIRE(n_cells <= 0xFFFFFFFFull, "pcl_update_force_grid: more than 2^32 - 1 cells");
    if (int rc = reserve_field(h, n_cells)) return rc;
    if (n_cells != h->n_cells) h->brush_down = false;   // a selection indexes the field it was made on
    h->n_cells = n_cells;
    return upload_field(h, 0, force_grid, n_cells, false);
}

int pcl_update_force_grid_device(pcl_compute* h, const void* dev_force_grid, size_t n_cells)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(dev_force_grid != nullptr && n_cells >= 1, "pcl_update_force_grid_device: the force grid is empty");
    PCL_REQUIRE(n_cells <= 0xFFFFFFFFull, "pcl_update_force_grid_device: more than 2^32 - 1 cells");
    if (int rc = reserve_field(h, n_cells)) return rc;
    if (n_cells != h->n_cells) h->brush_down = false;   // a selection indexes the field it was made on
    h->n_cells = n_cells;
    return upload_field(h, 0, dev_force_grid, n_cells, true);
}

int pcl_patch_force_grid(pcl_compute* h, size_t first_cell, const float (*cells)[4], size_t count)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(count == 0 || cells != nullptr, "pcl_patch_force_grid: cells is null");
    if (int rc = check_range(first_cell, count, h->n_cells, "pcl_patch_force_grid")) return rc;
    return upload_field(h, first_cell, cells, count, false);
}

int pcl_patch_force_grid_device(pcl_compute* h, size_t first_cell, const void* dev_cells, size_t count)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(count == 0 || dev_cells != nullptr, "pcl_patch_force_grid_device: dev_cells is null");
    if (int rc = check_range(first_cell, count, h->n_cells, "pcl_patch_force_grid_device")) return rc;
    return upload_field(h, first_cell, dev_cells, count, true);
}

int pcl_patch_force_grid_indexed(pcl_compute* h, const uint32_t* cell_indices, const float (*cells)[4], size_t count)
{
    if (int rc = bind(h)) return rc;
    if (count == 0) return PCL_OK;
    PCL_REQUIRE(cell_indices != nullptr && cells != nullptr, "pcl_patch_force_grid_indexed: null argument");
    for (size_t i = 0; i < count; ++i)
        if (cell_indices[i] >= h->n_cells)
            return fail(PCL_ERR_OUT_OF_RANGE, "pcl_patch_force_grid_indexed: cell index " +
                                                  std::to_string(cell_indices[i]) + " is outside the field");
    // The reference's loop writes the listed cells one after the other, so of several occurrences of an
    // index the last one stays: keep only that one (the kernel's order of writes is not defined).
    std::vector<uint32_t> keep_index;
    std::vector<float> keep_cells;
    {
        std::unordered_set<uint32_t> seen;
        seen.reserve(count * 2);
        bool duplicates = false;
        for (size_t i = 0; i < count && !duplicates; ++i) duplicates = !seen.insert(cell_indices[i]).second;
        if (duplicates) {
            seen.clear();
            std::vector<size_t> last;   // positions that survive, descending
            for (size_t i = count; i-- > 0;)
                if (seen.insert(cell_indices[i]).second) last.push_back(i);
            keep_index.reserve(last.size());
            keep_cells.reserve(last.size() * 4);
            for (size_t k = last.size(); k-- > 0;) {
                keep_index.push_back(cell_indices[last[k]]);
                keep_cells.insert(keep_cells.end(), cells[last[k]], cells[last[k]] + 4);
            }
            cell_indices = keep_index.data();
            cells = reinterpret_cast<const float (*)[4]>(keep_cells.data());
            count = keep_index.size();
        }
    }
    char* staging = nullptr;
    PCL_HIP_TRY(hipMalloc((void**)&staging, count * (sizeof(pcl::f32x4) + sizeof(uint32_t))));
    pcl::f32x4* d_cells = reinterpret_cast<pcl::f32x4*>(staging);
    uint32_t* d_index = reinterpret_cast<uint32_t*>(staging + count * sizeof(pcl::f32x4));
    hipError_t e = hipMemcpyAsync(d_cells, cells, count * sizeof(pcl::f32x4), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_index, cell_indices, count * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = launch_indexed_patch(h, d_index, d_cells, count);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(staging);
    if (e != hipSuccess) return fail(PCL_ERR_HIP, std::string("pcl_patch_force_grid_indexed: ") + hipGetErrorString(e));
    return PCL_OK;
}

int pcl_patch_force_grid_indexed_device(pcl_compute* h, const void* dev_cell_indices, const void* dev_cells, size_t count)
{
    if (int rc = bind(h)) return rc;
    if (count == 0) return PCL_OK;
    PCL_REQUIRE(dev_cell_indices != nullptr && dev_cells != nullptr, "pcl_patch_force_grid_indexed_device: null argument");
    PCL_HIP_TRY(launch_indexed_patch(h, reinterpret_cast<const uint32_t*>(dev_cell_indices),
                                     reinterpret_cast<const pcl::f32x4*>(dev_cells), count));
    return PCL_OK;
}

void pcl_brush_input_default(pcl_brush_input* out)
{
    if (!out) return;
    std::memset(out, 0, sizeof *out);
    out->cam_right[0] = 1.0f;   // Cursor::new: rot = identity, src/cursor.rs:20-34
    out->cam_dir[2] = 1.0f;
    out->mode = PCL_EDIT_SHIFT;   // EditMode::default, src/cursor.rs:206-216
    out->relative = 1;
    out->falloff = PCL_FALLOFF_ABRUPT;
    out->falloff_dist = 1.0f;
    out->strength = 1.0f;
}

static int check_brush_input(const pcl_brush_input* in)
{
    PCL_REQUIRE(in, "null brush record");
    PCL_REQUIRE(in->mode >= PCL_EDIT_CENTERED && in->mode <= PCL_EDIT_ROTATE, "unknown edit mode");
    PCL_REQUIRE(in->falloff >= PCL_FALLOFF_ABRUPT && in->falloff <= PCL_FALLOFF_INVERSE_DISTANCE, "unknown falloff");
    return PCL_OK;
}

int pcl_brush_mouse_down(pcl_compute* h, const pcl_brush_input* in)
{
    PCL_REQUIRE(h, "null handle");
    if (int rc = check_brush_input(in)) return rc;
    return brush_mouse_down(h, in, nullptr);
}

int pcl_brush_mouse_down_device(pcl_compute* h, const void* dev_in)
{
    PCL_REQUIRE(h && dev_in, "null handle or record");
    return brush_mouse_down(h, nullptr, dev_in);
}

int pcl_brush_mouse_moved(pcl_compute* h, const pcl_brush_input* in)
{
    PCL_REQUIRE(h, "null handle");
    if (int rc = check_brush_input(in)) return rc;
    return brush_mouse_moved(h, in, nullptr);
}

int pcl_brush_mouse_moved_device(pcl_compute* h, const void* dev_in)
{
    PCL_REQUIRE(h && dev_in, "null handle or record");
    return brush_mouse_moved(h, nullptr, dev_in);
}

int pcl_brush_mouse_up(pcl_compute* h)
{
    PCL_REQUIRE(h, "null handle");
    h->brush_down = false;   // src/cursor.rs:118-121
    return PCL_OK;
}

int pcl_brush_selection(pcl_compute* h, uint32_t* out_indices, size_t capacity, size_t* count)
{
    PCL_REQUIRE(h && count, "null handle or count");
    *count = 0;
    if (!h->brush_down) return PCL_OK;
    if (int rc = bind(h)) return rc;
    if (int rc = sync_stream(h)) return rc;
    pcl::BrushState st;
    PCL_HIP_TRY(hipMemcpy(&st, h->d_brush_state, sizeof st, hipMemcpyDeviceToHost));
    *count = st.count;
    const size_t take = std::min<size_t>(st.count, capacity);
    if (take) {
        PCL_REQUIRE(out_indices, "null output");
        std::vector<uint32_t> all(st.count);
        PCL_HIP_TRY(hipMemcpy(all.data(), h->d_brush_index, all.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        std::sort(all.begin(), all.end());   // the device appends in arbitrary order
        std::memcpy(out_indices, all.data(), take * sizeof(uint32_t));
    }
    return PCL_OK;
}

size_t pcl_num_cells(const pcl_compute* h) { return h ? h->n_cells : 0; }

int pcl_read_force_grid(pcl_compute* h, float (*out)[4], size_t first_cell, size_t count)
{
    PCL_REQUIRE(h, "null handle");
    if (int rc = check_range(first_cell, count, h->n_cells, "pcl_read_force_grid")) return rc;
    if (count == 0) return PCL_OK;
    PCL_REQUIRE(out, "null output");
    if (int rc = bind(h)) return rc;
    PCL_HIP_TRY(hipMemcpyAsync(out, h->field + first_cell, count * sizeof(pcl::f32x4), hipMemcpyDeviceToHost, h->stream));
    return sync_stream(h);
}

int pcl_update_sim_params(pcl_compute* h, const pcl_sim_params* params)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    PCL_REQUIRE(params != nullptr, "pcl_update_sim_params: params is null");
    // the field dimensions and the radius define the tile keys of a pending histogram
    if (std::memcmp(h->params.force_grid_dimensions, params->force_grid_dimensions, sizeof params->force_grid_dimensions) != 0 ||
        std::memcmp(&h->params.bounding_volume_radius, &params->bounding_volume_radius, sizeof(float)) != 0)
        h->hist_valid = false;
    h->params = *params;
    h->params_dirty = true;
    h->diag_current = false;  // the masses weight the kinetic energy
    return PCL_OK;
}

int pcl_step(pcl_compute* h) { return do_step(h); }

int pcl_step_n(pcl_compute* h, uint64_t n_steps)
{
    PCL_REQUIRE(h, "null handle");
    int rc = PCL_OK;
    while (n_steps > 0 && rc == PCL_OK) {
        h->fuse_request = (uint32_t)std::min<uint64_t>(n_steps, 0xFFFFFFFFull);
        rc = do_step(h);
        n_steps -= std::min<uint64_t>(n_steps, h->last_advanced);
    }
    h->fuse_request = 0;
    return rc;
}

int pcl_step_dt(pcl_compute* h, float delta_t)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    h->params.delta_t = delta_t;
    h->params_dirty = true;
    return do_step(h);
}

int pcl_pack_pair_sources_device(pcl_compute* h, void* dev_out, size_t capacity_records)
{
    PCL_REQUIRE(h, "null handle");
    PCL_REQUIRE(capacity_records >= h->n, "capacity_records is smaller than the particle count");
    if (capacity_records == 0) return PCL_OK;
    PCL_REQUIRE(dev_out, "null output pointer");
    PCL_REQUIRE(reinterpret_cast<uintptr_t>(dev_out) % 16 == 0, "output pointer must be 16-byte aligned");
    if (int rc = bind(h)) return rc;
    const uint32_t* slot_of = nullptr;
    if (int rc = caller_order_map(h, &slot_of)) return rc;
    hipLaunchKernelGGL(pcl::pack_pair_sources_kernel, dim3(linear_grid(h, capacity_records)), dim3(pcl::kBlockThreads),
                       0, h->stream, soa_of(h, h->cur), cur_ty(h), slot_of, (unsigned long long)h->n,
                       (unsigned long long)(capacity_records - h->n), static_cast<pcl::f32x4*>(dev_out));
    PCL_HIP_TRY(hipGetLastError());
    return PCL_OK;
}

int pcl_step_pairs_gathered(pcl_compute* h, const void* dev_sources, size_t n_sources)
{
    PCL_REQUIRE(h, "null handle");
    PCL_REQUIRE(dev_sources || n_sources == 0, "null sources pointer");
    PCL_REQUIRE(reinterpret_cast<uintptr_t>(dev_sources) % 16 == 0, "sources pointer must be 16-byte aligned");
    h->gathered_step = true;
    h->ext_sources = static_cast<const pcl::f32x4*>(dev_sources);
    h->n_ext_sources = n_sources;
    const int rc = do_step(h);
    h->gathered_step = false;
    h->ext_sources = nullptr;
    h->n_ext_sources = 0;
    return rc;
}

int pcl_force_pair_kernel(pcl_compute* h, int enable)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    h->force_pairs = enable != 0;
    return PCL_OK;
}

int pcl_set_pair_mode(pcl_compute* h, int mode)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    PCL_REQUIRE(mode == PCL_PAIRS_AUTO || mode == PCL_PAIRS_ALL || mode == PCL_PAIRS_BINNED,
                "pcl_set_pair_mode: unknown mode");
    h->pair_mode = mode;
    return PCL_OK;
}

int pcl_last_step_kernel(const pcl_compute* h) { return h ? h->last_kernel : PCL_KERNEL_NONE; }

int pcl_enable_step_diagnostics(pcl_compute* h, int enable)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    h->step_diag = enable != 0;
    return PCL_OK;
}

size_t pcl_num_particles(const pcl_compute* h) { return h ? h->n : 0; }

uint64_t pcl_step_count(const pcl_compute* h) { return h ? h->steps : 0; }

int pcl_read_particles(pcl_compute* h, pcl_particle* out, size_t first, size_t count)
{
    if (int rc = bind(h)) return rc;
    if (int rc = check_range(first, count, h->n, "pcl_read_particles")) return rc;
    if (count == 0) return PCL_OK;
    PCL_REQUIRE(out != nullptr, "pcl_read_particles: out is null");
    const uint32_t* slot_of = nullptr;
    if (int rc = caller_order_map(h, &slot_of)) return rc;
    const size_t chunk = std::min(count, kStagingRecords);
    pcl::ParticleAoS* staging = nullptr;
    PCL_HIP_TRY(hipMalloc((void**)&staging, chunk * sizeof(pcl::ParticleAoS)));
    int rc = PCL_OK;
    for (size_t done = 0; done < count && rc == PCL_OK; done += chunk) {
        const size_t cnt = std::min(chunk, count - done);
        hipLaunchKernelGGL(pcl::pack_aos_kernel, dim3(linear_grid(h, cnt)), dim3(pcl::kBlockThreads), 0,
                           h->stream, soa_of(h, h->cur), (const uint8_t*)cur_ty(h), slot_of,
                           (unsigned long long)(first + done), (unsigned long long)cnt, staging);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(out + done, staging, cnt * sizeof(pcl_particle), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(PCL_ERR_HIP, std::string("pcl_read_particles: ") + hipGetErrorString(e));
    }
    (void)hipFree(staging);
    if (rc == PCL_OK) rc = sync_stream(h);
    return rc;
}

int pcl_read_soa(pcl_compute* h, size_t first, size_t count, float* x, float* y, float* z, float* vx,
                 float* vy, float* vz, uint8_t* ty)
{
    if (int rc = bind(h)) return rc;
    if (int rc = check_range(first, count, h->n, "pcl_read_soa")) return rc;
    if (count == 0) return PCL_OK;
    float* dst[6] = {x, y, z, vx, vy, vz};
    const uint32_t* slot_of = nullptr;
    if (int rc = caller_order_map(h, &slot_of)) return rc;
    if (!slot_of) {
        for (int a = 0; a < 6; ++a)
            if (dst[a])
                PCL_HIP_TRY(hipMemcpyAsync(dst[a], h->state[h->cur][a] + first, count * sizeof(float),
                                           hipMemcpyDeviceToHost, h->stream));
        if (ty) PCL_HIP_TRY(hipMemcpyAsync(ty, cur_ty(h) + first, count, hipMemcpyDeviceToHost, h->stream));
        return sync_stream(h);
    }
    // Tile order: gather the range into caller order on the device first.
    char* tmp = nullptr;
    PCL_HIP_TRY(hipMalloc((void**)&tmp, count * (6 * sizeof(float) + 1)));
    pcl::StateSoA out;
    for (int a = 0; a < 6; ++a) out.a[a] = reinterpret_cast<float*>(tmp) + (size_t)a * count;
    uint8_t* out_ty = reinterpret_cast<uint8_t*>(tmp) + 6 * sizeof(float) * count;
    hipLaunchKernelGGL(pcl::gather_soa_kernel, dim3(linear_grid(h, count)), dim3(pcl::kBlockThreads), 0, h->stream,
                       soa_of(h, h->cur), (const uint8_t*)cur_ty(h), slot_of, (unsigned long long)first,
                       (unsigned long long)count, out, out_ty);
    hipError_t e = hipGetLastError();
    for (int a = 0; a < 6 && e == hipSuccess; ++a)
        if (dst[a]) e = hipMemcpyAsync(dst[a], out.a[a], count * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (ty && e == hipSuccess) e = hipMemcpyAsync(ty, out_ty, count, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(PCL_ERR_HIP, std::string("pcl_read_soa: ") + hipGetErrorString(e));
    return sync_stream(h);
}

int pcl_pack_particles_device(pcl_compute* h, void* dev_out, size_t first, size_t count)
{
    if (int rc = bind(h)) return rc;
    if (int rc = check_range(first, count, h->n, "pcl_pack_particles_device")) return rc;
    if (count == 0) return PCL_OK;
    PCL_REQUIRE(dev_out != nullptr, "pcl_pack_particles_device: dev_out is null");
    if (h->sorted && count >= h->n / 8) {
        // Tile order and a large range: walk the slots and write every record at its caller index.
        hipLaunchKernelGGL(pcl::pack_aos_by_slot_kernel, dim3(linear_grid(h, h->n)), dim3(pcl::kBlockThreads), 0,
                           h->stream, soa_of(h, h->cur), (const uint8_t*)cur_ty(h),
                           (const uint32_t*)h->orig[h->orig_cur], (unsigned long long)h->n, (unsigned long long)first,
                           (unsigned long long)count, reinterpret_cast<pcl::ParticleAoS*>(dev_out));
        PCL_HIP_TRY(hipGetLastError());
        return PCL_OK;
    }
    const uint32_t* slot_of = nullptr;
    if (int rc = caller_order_map(h, &slot_of)) return rc;
    hipLaunchKernelGGL(pcl::pack_aos_kernel, dim3(linear_grid(h, count)), dim3(pcl::kBlockThreads), 0,
                       h->stream, soa_of(h, h->cur), (const uint8_t*)cur_ty(h), slot_of, (unsigned long long)first,
                       (unsigned long long)count, reinterpret_cast<pcl::ParticleAoS*>(dev_out));
    PCL_HIP_TRY(hipGetLastError());
    return PCL_OK;
}

int pcl_pack_particles_device_unordered(pcl_compute* h, void* dev_out)
{
    if (int rc = bind(h)) return rc;
    if (h->n == 0) return PCL_OK;
    PCL_REQUIRE(dev_out != nullptr, "pcl_pack_particles_device_unordered: dev_out is null");
    hipLaunchKernelGGL(pcl::pack_aos_kernel, dim3(linear_grid(h, h->n)), dim3(pcl::kBlockThreads), 0,
                       h->stream, soa_of(h, h->cur), (const uint8_t*)cur_ty(h), (const uint32_t*)nullptr,
                       0ull, (unsigned long long)h->n, reinterpret_cast<pcl::ParticleAoS*>(dev_out));
    PCL_HIP_TRY(hipGetLastError());
    return PCL_OK;
}

int pcl_diagnostics_read(pcl_compute* h, pcl_diagnostics* out)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(out != nullptr, "pcl_diagnostics_read: out is null");
    if (int rc = ensure_diag(h)) return rc;
    PCL_HIP_TRY(hipMemcpyAsync(out, h->d_diag, sizeof(pcl_diagnostics), hipMemcpyDeviceToHost, h->stream));
    return sync_stream(h);
}

int pcl_diagnostics_device(pcl_compute* h, void* dev_out)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(dev_out != nullptr, "pcl_diagnostics_device: dev_out is null");
    if (int rc = ensure_diag(h)) return rc;
    PCL_HIP_TRY(hipMemcpyAsync(dev_out, h->d_diag, sizeof(pcl_diagnostics), hipMemcpyDeviceToDevice, h->stream));
    return PCL_OK;
}

int pcl_diagnostics_device_noalloc(pcl_compute* h, void* dev_out)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(dev_out != nullptr, "pcl_diagnostics_device_noalloc: dev_out is null");
    PCL_REQUIRE(reinterpret_cast<uintptr_t>(dev_out) % 8 == 0, "pcl_diagnostics_device_noalloc: dev_out must be 8-byte aligned");
    PCL_REQUIRE(h->n <= PCL_NOALLOC_MAX_PARTICLES,
                "pcl_diagnostics_device_noalloc: more than PCL_NOALLOC_MAX_PARTICLES (2^32) particles; use pcl_diagnostics_device");
    pcl::DiagPartial* ptr = h->device < kMaxDevices ? g_scratch_ptr[h->device] : nullptr;
    PCL_REQUIRE(ptr != nullptr, "pcl_diagnostics_device_noalloc: no scratch array on this device (device ordinal >= 64)");
    // the same two launches as ensure_diag, with the block partials in the device's scratch array
    // and the record written straight to the caller's memory
    const uint32_t grid = h->n ? std::min<uint32_t>(linear_grid(h, h->n), kNoAllocBlocks) : 1u;
    hipLaunchKernelGGL(pcl::diag_reduce_kernel, dim3(grid), dim3(pcl::kBlockThreads), 0, h->stream,
                       soa_of(h, h->cur), cur_ty(h), (unsigned long long)h->n, uniforms_of(h), ptr);
    PCL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(pcl::diag_fold_kernel, dim3(1), dim3(pcl::kBlockThreads), 0, h->stream,
                       (const pcl::DiagPartial*)ptr, grid, grid, (pcl::DiagPartial*)nullptr,
                       static_cast<pcl::DiagOut*>(dev_out));
    PCL_HIP_TRY(hipGetLastError());
    return PCL_OK;
}

int pcl_debug_cell_indices(pcl_compute* h, uint32_t* out, size_t first, size_t count)
{
    if (int rc = bind(h)) return rc;
    if (int rc = check_range(first, count, h->n, "pcl_debug_cell_indices")) return rc;
    if (count == 0) return PCL_OK;
    PCL_REQUIRE(out != nullptr, "pcl_debug_cell_indices: out is null");
    PCL_REQUIRE(h->n_cells >= 1, "the force grid is empty");
    const uint32_t* slot_of = nullptr;
    if (int rc = caller_order_map(h, &slot_of)) return rc;
    uint32_t* d = nullptr;
    PCL_HIP_TRY(hipMalloc((void**)&d, count * sizeof(uint32_t)));
    hipLaunchKernelGGL(pcl::cell_index_kernel, dim3(linear_grid(h, count)), dim3(pcl::kBlockThreads), 0,
                       h->stream, soa_of(h, h->cur), slot_of, (unsigned long long)first, (unsigned long long)count,
                       uniforms_of(h), d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, count * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PCL_ERR_HIP, std::string("pcl_debug_cell_indices: ") + hipGetErrorString(e));
    return sync_stream(h);
}

int pcl_debug_pair_cells(pcl_compute* h, uint32_t* cell_of_particle, uint32_t* bin_end, size_t bin_capacity,
                         uint32_t* cells_per_axis, uint32_t* reach)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(cells_per_axis && reach, "pcl_debug_pair_cells: null output");
    PCL_REQUIRE(h->n >= 1 && h->n <= 0x7FFFFFFFull, "pcl_debug_pair_cells: 1 .. 2^31 - 1 particles");
    const pcl::TileMap C = pair_cell_map_of(h);
    *cells_per_axis = C.nty;
    *reach = C.reach;
    PCL_REQUIRE(!bin_end || bin_capacity >= C.n_bins, "pcl_debug_pair_cells: bin_end is too small");
    // exactly the sort the binned pair step runs before its sweep (the particles end up in cell order)
    const uint32_t* d_bin_end = nullptr;
    if (int rc = cell_sort(h, uniforms_of(h), C, &d_bin_end)) return rc;
    if (bin_end)
        PCL_HIP_TRY(hipMemcpyAsync(bin_end, d_bin_end, (size_t)C.n_bins * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    std::vector<uint32_t> key(h->n), caller(h->n);
    const uint32_t* key_out = h->d_cellsort + h->cellsort_capacity;          // sorted keys, slot by slot
    const uint32_t* idx_out = h->d_cellsort + 3 * h->cellsort_capacity;      // caller index of every slot
    PCL_HIP_TRY(hipMemcpyAsync(key.data(), key_out, h->n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    PCL_HIP_TRY(hipMemcpyAsync(caller.data(), idx_out, h->n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    if (int rc = sync_stream(h)) return rc;
    if (cell_of_particle)
        for (size_t slot = 0; slot < h->n; ++slot) {
            PCL_REQUIRE(caller[slot] < h->n, "pcl_debug_pair_cells: the sort returned an index outside the set");
            cell_of_particle[caller[slot]] = key[slot];
        }
    return PCL_OK;
}

int pcl_debug_state_checksum(pcl_compute* h, uint64_t first_global_index, uint64_t* sum, uint64_t* xor_of_all)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(sum != nullptr && xor_of_all != nullptr, "pcl_debug_state_checksum: null output");
    *sum = *xor_of_all = 0;
    if (h->n == 0) return sync_stream(h);
    unsigned long long* d = nullptr;
    PCL_HIP_TRY(hipMalloc((void**)&d, 2 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pcl::state_checksum_kernel, dim3(linear_grid(h, h->n)), dim3(pcl::kBlockThreads), 0, h->stream,
                           soa_of(h, h->cur), (const uint8_t*)cur_ty(h),
                           h->sorted ? (const uint32_t*)h->orig[h->orig_cur] : (const uint32_t*)nullptr,
                           (unsigned long long)h->n, (unsigned long long)first_global_index, d);
        e = hipGetLastError();
    }
    unsigned long long host[2] = {0ull, 0ull};
    if (e == hipSuccess) e = hipMemcpyAsync(host, d, sizeof host, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PCL_ERR_HIP, std::string("pcl_debug_state_checksum: ") + hipGetErrorString(e));
    *sum = host[0];
    *xor_of_all = host[1];
    return sync_stream(h);
}

int pcl_debug_slot_order(pcl_compute* h, uint32_t* caller_index_of_slot, size_t capacity)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(caller_index_of_slot != nullptr && capacity >= h->n, "pcl_debug_slot_order: the output is too small");
    PCL_REQUIRE(h->n <= 0xFFFFFFFFull, "pcl_debug_slot_order: more than 2^32 particles");
    if (h->n == 0) return PCL_OK;
    if (h->sorted) {
        PCL_HIP_TRY(hipMemcpyAsync(caller_index_of_slot, h->orig[h->orig_cur], h->n * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                   h->stream));
        return sync_stream(h);
    }
    if (int rc = sync_stream(h)) return rc;
    for (size_t i = 0; i < h->n; ++i) caller_index_of_slot[i] = (uint32_t)i;   // still in caller order
    return PCL_OK;
}

int pcl_set_stream(pcl_compute* h, void* hip_stream)
{
    if (int rc = bind(h)) return rc;
    hipStream_t next = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : h->own_stream;
    if (next != h->stream) PCL_HIP_TRY(hipStreamSynchronize(h->stream));  // keep work ordered across the switch
    h->stream = next;
    return PCL_OK;
}

int pcl_use_null_stream(pcl_compute* h)
{
    if (int rc = bind(h)) return rc;
    if (h->stream != nullptr) PCL_HIP_TRY(hipStreamSynchronize(h->stream));
    h->stream = nullptr;   // the device's null stream
    return PCL_OK;
}

int pcl_synchronize(pcl_compute* h)
{
    if (int rc = bind(h)) return rc;
    return sync_stream(h);
}

int pcl_timer_start(pcl_compute* h)
{
    if (int rc = bind(h)) return rc;
    PCL_HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    return PCL_OK;
}

int pcl_timer_stop(pcl_compute* h, float* elapsed_ms)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(elapsed_ms != nullptr, "pcl_timer_stop: elapsed_ms is null");
    PCL_HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    PCL_HIP_TRY(hipEventSynchronize(h->ev_stop));
    PCL_HIP_TRY(hipEventElapsedTime(elapsed_ms, h->ev_start, h->ev_stop));
    return PCL_OK;
}

int pcl_step_timing_begin(pcl_compute* h, size_t max_steps)
{
    if (int rc = bind(h)) return rc;
    while (h->timing_events.size() < 2 * max_steps) {
        hipEvent_t e = nullptr;
        // timestamps only: no system-scope fence (cache write-back) with every record
        PCL_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
        h->timing_events.push_back(e);
    }
    h->timing_kinds.assign(max_steps, PCL_STEP_NONE);
    h->timing_used = 0;
    h->timing_on = max_steps > 0;
    return PCL_OK;
}

int pcl_step_timing_read(pcl_compute* h, float* elapsed_ms, int32_t* kinds, size_t capacity, size_t* count)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(count != nullptr, "pcl_step_timing_read: count is null");
    h->timing_on = false;
    const size_t n = std::min(capacity, h->timing_used);
    PCL_REQUIRE(n == 0 || (elapsed_ms != nullptr && kinds != nullptr), "pcl_step_timing_read: null output");
    if (h->timing_used) PCL_HIP_TRY(hipEventSynchronize(h->timing_events[2 * h->timing_used - 1]));
    for (size_t i = 0; i < n; ++i) {
        PCL_HIP_TRY(hipEventElapsedTime(&elapsed_ms[i], h->timing_events[2 * i], h->timing_events[2 * i + 1]));
        kinds[i] = h->timing_kinds[i];
    }
    *count = n;
    return PCL_OK;
}

int pcl_launch_geometry(const pcl_compute* h, uint32_t* grid_blocks, uint32_t* block_threads,
                        uint32_t* particles_per_lane)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    uint32_t blocks = stream_grid(h), threads = pcl::kStreamThreads;
    if (!use_lds_field(h) && !h->grid_cap) {   // the plain step's own shape (the counting and scatter steps keep 512 threads)
        const PlainShape shape = plain_shape(h, h->step_diag, uniforms_of(h).fast_div != 0u);
        threads = shape.block;
        blocks = (uint32_t)std::min<size_t>(std::max<size_t>(1, (h->n / 4 + threads - 1) / threads), 0x7FFFFFFFu);
    }
    if (grid_blocks) *grid_blocks = blocks;
    if (block_threads) *block_threads = threads;
    if (particles_per_lane) *particles_per_lane = pcl::kParticlesPerLane;
    return PCL_OK;
}

int pcl_pair_sweep_info(const pcl_compute* h, uint32_t* shared_sweep, uint32_t* cells_occupied)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    if (shared_sweep) *shared_sweep = h->last_sweep_shared ? 1u : 0u;
    if (cells_occupied) *cells_occupied = h->cells_occupied;
    return PCL_OK;
}

int pcl_device_pci_bus_id(const pcl_compute* h, char* out, size_t capacity)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    PCL_REQUIRE(out != nullptr && capacity >= 16, "pcl_device_pci_bus_id: output buffer of at least 16 bytes");
    PCL_HIP_TRY(hipDeviceGetPCIBusId(out, (int)std::min<size_t>(capacity, 64), h->device));
    return PCL_OK;
}

int pcl_launch_residency(const pcl_compute* h, uint32_t* intended_blocks_per_cu, uint32_t* granted_blocks_per_cu)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    if (intended_blocks_per_cu) *intended_blocks_per_cu = h->residency_intended;
    if (granted_blocks_per_cu) *granted_blocks_per_cu = h->residency_granted;
    return PCL_OK;
}

int pcl_set_reorder_interval(pcl_compute* h, int interval)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    PCL_REQUIRE(interval >= -1, "pcl_set_reorder_interval: interval must be -1 (automatic), 0 (never) or positive");
    h->reorder_interval = interval;
    h->hist_valid = false;
    return PCL_OK;
}

int pcl_reorder_interval_in_use(const pcl_compute* h) { return h ? effective_reorder_interval(h) : 0; }

int pcl_reorder_now(pcl_compute* h)
{
    if (int rc = bind(h)) return rc;
    PCL_REQUIRE(h->n < 0xFFFFFFFFull, "pcl_reorder_now: more than 2^32 - 2 particles");
    return do_reorder(h, tile_aim(h));
}

uint64_t pcl_reorder_count(const pcl_compute* h) { return h ? h->reorders : 0; }

#ifdef PCL_PHASE_CLOCKS
int pcl_debug_experiment(pcl_compute* h, int mode)
{
    if (int rc = bind(h)) return rc;
    PCL_HIP_TRY(hipStreamSynchronize(h->stream));
    PCL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(pcl::experiment_mode), &mode, sizeof mode));
    return PCL_OK;
}
// development builds only: the phase clocks of the last scatter launch (256 sampled blocks x 24 stamps)
int pcl_debug_phase_clocks(pcl_compute* h, unsigned long long* out)
{
    if (int rc = bind(h)) return rc;
    PCL_HIP_TRY(hipStreamSynchronize(h->stream));
    PCL_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(pcl::phase_clocks), sizeof(unsigned long long) * 256 * 24));
    return PCL_OK;
}
#endif

int pcl_set_tuning(pcl_compute* h, uint32_t grid_blocks, uint32_t variant)
{
    if (!h) return fail(PCL_ERR_INVALID, "null handle");
    h->grid_cap = grid_blocks;
    h->variant = variant;
    return PCL_OK;
}

}  // extern "C"
