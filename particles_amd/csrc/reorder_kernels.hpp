// reorder_kernels.hpp — keeps the particle arrays in field-tile order.
//
// The update of one particle depends only on its own state (pair term inert),
// so the order of the particles in memory is free.  The only non-streaming
// access of the step is one 16-byte field fetch per particle; with particles
// in arbitrary order and a field larger than an XCD's 4 MiB L2 most of those
// fetches leave the L2, and each costs as much fabric traffic as a particle's
// whole stream.  Storing the particles sorted by the field tile they sit in
// makes concurrently running workgroups touch a few adjacent tiles, so the
// fetches hit L1/L2 again.  Particles drift a few cells per step, so the order
// is refreshed every few steps by one counting-sort pass:
//   count   : histogram of tile keys                      (reads the state)
//   scan    : exclusive prefix over the bins               (one block)
//   scatter : every tile of 2048 particles reserves, per bin it touches, a
//             contiguous segment with one atomic add, and writes its
//             particles there (state, type and the caller's index).
// Results per particle are unchanged bit for bit; `orig` remembers the
// caller's index of every slot so that read-back returns caller order.
// No counterpart in the reference (src/compute.wgsl processes particles in
// buffer order and never reorders them).
#pragma once

#include "step_kernels.hpp"

namespace pcl {

// Development only (-DPCL_PHASE_CLOCKS, never in the shipped library): thread 0 of every 1024th block of
// a scatter kernel records the 100 MHz wall clock at its phase boundaries.
#ifdef PCL_PHASE_CLOCKS
// timing experiments with wrong results: 1 = dword stores to tile-local slots, 2 = 16-byte stores to tile-local
// slots, 3 = no state stores
__device__ int experiment_mode;
__device__ unsigned long long phase_clocks[256][24];
#define PCL_PHASE(i)                                                                                   \
    do {                                                                                               \
        if (threadIdx.x == 0 && (blockIdx.x & 1023u) == 512u && (blockIdx.x >> 10) < 256u)              \
            phase_clocks[blockIdx.x >> 10][i] = wall_clock64();                                        \
    } while (0)
#else
#define PCL_PHASE(i) do { } while (0)
#endif

constexpr uint32_t kMaxReorderBins = 8192;
// Every bin's counter and cursor exist several times, one per residue of the tile index (a "cursor
// group"): the tiles that run at the same time fall into the same few bins, and their reservations on
// one address serialise.  More groups cut every bin's range into more pieces, which shortens the
// contiguous runs the scatter writes and the order's locality.  Measured on the scatter step of 2^28
// particles with 256-cell sort tiles (65536 particles per bin): 1 group 4.58 ms, 2: 4.29, 4: 3.96,
// 8: 3.96, 16: 4.25, 64: 5.14; with the 512-cell tiles the library uses there now (131072 per bin):
// 2: 4.43, 4: 4.05, 8: 3.65, 16: 3.92; at 2^25 particles (1024-cell tiles, 32768 per bin): 4: 0.520,
// 8: 0.515, 16: 0.54 ms with the whole step best at 4.  The host picks 4 or 8 by particles per bin.
// Counter of bin b for tile t: group t % groups (storage: counter_index); the scan runs in
// bin-major order, so every bin still owns one contiguous range of slots.
constexpr uint32_t kMaxCursorGroups = 8;

// Where the counter (and cursor) of (bin, group) is stored: group-major, which puts the
// groups of one bin into different cache lines.
__device__ __forceinline__ uint32_t counter_index(uint32_t bin, uint32_t group, uint32_t n_bins)
{
    return group * n_bins + bin;
}

// Histogram of tile keys: one tile of kStreamThreads * 4 particles per block, LDS
// counters, one global atomic add per non-empty bin into the counter of the
// tile's group (see kMaxCursorGroups; groups = 1 gives one counter per bin).
__global__ __launch_bounds__(kStreamThreads) void reorder_count_kernel(
    StateSoA s, unsigned long long n, StepUniforms U, TileMap T, uint32_t* __restrict__ hist, uint32_t groups)
{
    extern __shared__ uint32_t lds_hist[];
    for (uint32_t b = threadIdx.x; b < T.n_bins; b += kStreamThreads) lds_hist[b] = 0u;
    __syncthreads();
    const unsigned long long n_groups = n / 4ull;
    const unsigned long long g = (unsigned long long)blockIdx.x * kStreamThreads + threadIdx.x;
    if (g < n_groups) {
        f32x4 V[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) V[c] = load4<false>(s.a[c], g);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            lds_rank_add(lds_hist, tile_key(U, T, V[0][k], V[1][k], V[2][k], V[3][k], V[4][k], V[5][k]), true);
    }
    // the ragged tail is counted with tile 0, as the scatter kernels place it
    const unsigned long long tail = n_groups * 4ull + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 3 && tail < n)
        atomicAdd(&lds_hist[tile_key(U, T, s.a[0][tail], s.a[1][tail], s.a[2][tail], s.a[3][tail], s.a[4][tail], s.a[5][tail])], 1u);
    __syncthreads();
    const uint32_t group = blockIdx.x % groups;
    for (uint32_t b = threadIdx.x; b < T.n_bins; b += kStreamThreads) {
        const uint32_t c = lds_hist[b];
        if (c) atomicAdd(&hist[counter_index(b, group, T.n_bins)], c);
    }
}

// Exclusive prefix of the counters into the cursors; one block.  Logical order is
// bin-major, group-minor (k = bin * groups + group) so that every bin's slots are
// contiguous; counter and cursor of (bin, group) are STORED at group * n_bins + bin,
// which puts the groups of one bin into different cache lines.
// `cursor_end` (may be null) receives the end of every counter's range, which the scatter
// kernels check their reservations against.
__global__ __launch_bounds__(1024) void reorder_scan_kernel(const uint32_t* __restrict__ hist, uint32_t n_bins,
                                                            uint32_t groups, uint32_t* __restrict__ cursor,
                                                            uint32_t* __restrict__ cursor_end)
{
    __shared__ uint32_t part[1024];
    const uint32_t total = n_bins * groups;
    const uint32_t per = (total + 1023u) / 1024u;
    const uint32_t begin = threadIdx.x * per;
    auto physical = [&](uint32_t k) { return counter_index(k / groups, k % groups, n_bins); };
    uint32_t sum = 0u;
    for (uint32_t i = 0; i < per; ++i)
        if (begin + i < total) sum += hist[physical(begin + i)];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {   // Hillis-Steele inclusive scan
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
    for (uint32_t i = 0; i < per; ++i)
        if (begin + i < total) {
            const uint32_t at = physical(begin + i);
            cursor[at] = run;
            run += hist[at];
            if (cursor_end) cursor_end[at] = run;
        }
}

struct ReorderArgs {
    const float* __restrict__ src[6];
    float* __restrict__ dst[6];
    const uint8_t* __restrict__ src_ty;
    uint8_t* __restrict__ dst_ty;
    const uint32_t* __restrict__ src_orig;  // nullptr: slots are still in caller order
    uint32_t* __restrict__ dst_orig;
    uint32_t* __restrict__ cursor;          // next free slot per (bin, group), at counter_index(bin, group, n_bins)
    const uint32_t* __restrict__ cursor_end; // end of that counter's range (what the histogram counted)
    uint32_t* error;                        // device error word
    uint32_t groups;
    uint32_t n_tiles;                       // full tiles; block n_tiles (if any) carries the ragged tail, in group 0
    unsigned long long n;
};

// LDS plan of the scatter kernels (dynamic shared memory), B = number of bins:
//   u32 slot[B]      per-bin particle count of this tile, then (bin's reserved
//                    global base) - (bin's first local position)
//   u16 lstart[B]    first local position of the bin inside the tile
//   u32 dest[2048]   global slot of every local position
//   u32 stage[2][2048], u8 stage_ty[2048]   one array at a time, regrouped
// The tile's particles are regrouped by bin in LDS so that consecutive lanes
// write consecutive slots: a tile whose particles fall into k bins issues k
// contiguous runs per array instead of 2048 scattered elements.
__host__ __device__ constexpr size_t scatter_lds_bytes(uint32_t n_bins)
{
    return (size_t)n_bins * 6u + (size_t)kStreamTile * 4u * 3u + (size_t)kStreamTile;
}

struct ScatterLds {
    uint32_t* slot;
    uint16_t* lstart;
    uint32_t* dest;
    uint32_t* stage[2];
    uint8_t* stage_ty;
    __device__ __forceinline__ ScatterLds(unsigned char* base, uint32_t n_bins)
    {
        slot = reinterpret_cast<uint32_t*>(base);
        dest = slot + n_bins;
        stage[0] = dest + kStreamTile;
        stage[1] = stage[0] + kStreamTile;
        lstart = reinterpret_cast<uint16_t*>(stage[1] + kStreamTile);
        stage_ty = reinterpret_cast<uint8_t*>(lstart + n_bins);
    }
};

// Planning a tile.  After every thread has added its particles to slot[] (lds_rank_add_listed) the per-bin
// counts are turned into local start positions (a sweep over all bins in ascending order, which is the
// order the tile's runs take inside the tile), every non-empty bin's run of global slots is reserved with
// one atomic add, and slot[b] is left holding the offset that maps a local position of bin b to its global
// slot.  The reservations are kept off the sweep: a thread that reserved the non-empty bins among its own
// consecutive bins one after the other would wait for every atomic before the next (a tile whose particles
// fall into a column of adjacent bins paid that many round trips in sequence: 8 us of a block's 23).
// Instead the bins a tile touches are listed while the particles are ranked (the lane that finds a counter
// at zero appends the bin), lane t of the block reserves listed bin t, so that all of a tile's reservations
// are in flight at once, and the results are consumed after the sweep.
// A reservation that runs past the end of its counter's range means the histogram and the keys disagree
// (they never do while both come from tile_key with one TileMap): particles would overwrite another bin's
// slots.  That sets kErrScatterOverflow in the error word.
struct ListedBins {
    uint32_t* list;      // bins of this tile that hold at least one particle, in the order they were first hit
    uint32_t* n_list;    // their number
};

// counters[key] += 1, returning this particle's rank; the first particle of a bin lists the bin.
__device__ __forceinline__ uint32_t lds_rank_add_listed(uint32_t* counters, const ListedBins& B, uint32_t key, bool active)
{
    if (!active) return 0u;
    const uint32_t r = atomicAdd(&counters[key], 1u);
    if (r == 0u) B.list[atomicAdd(B.n_list, 1u)] = key;
    return r;
}

// One reservation in flight per thread: listed bin threadIdx.x (if any).
struct Reservation {
    uint32_t bin;
    uint32_t count;
    uint32_t base;
    uint32_t end;
};

// To be called once every thread's ranks are in (after a barrier): issues this thread's reservation.
__device__ __forceinline__ Reservation reserve_listed(const ScatterLds& L, const ListedBins& B, uint32_t n_bins,
                                                      uint32_t* __restrict__ cursor, const uint32_t* __restrict__ cursor_end,
                                                      uint32_t group)
{
    Reservation r{0u, 0u, 0u, 0u};
    if (threadIdx.x < *B.n_list) {
        r.bin = B.list[threadIdx.x];
        r.count = L.slot[r.bin];
        const uint32_t at = counter_index(r.bin, group, n_bins);
        r.end = cursor_end[at];
        r.base = atomicAdd(&cursor[at], r.count);
    }
    return r;
}

// The sweep of scatter_plan without its reservations, then the reservations' results: leaves
// lstart[] and, in slot[b], (reserved base of bin b) - (first local position of bin b).  Ends with a barrier.
__device__ __forceinline__ void scatter_plan_listed(const ScatterLds& L, const ListedBins& B, const Reservation& mine,
                                                    uint32_t n_bins, uint32_t* __restrict__ cursor,
                                                    const uint32_t* __restrict__ cursor_end, uint32_t* error, uint32_t group)
{
    __shared__ uint32_t wave_total[kStreamThreads / 64];
    const uint32_t per = (n_bins + kStreamThreads - 1) / kStreamThreads;
    const uint32_t b0 = threadIdx.x * per;
    uint32_t sum = 0u;
    for (uint32_t i = 0; i < per; ++i)
        if (b0 + i < n_bins) sum += L.slot[b0 + i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wave_total[wave] = incl;
    lds_barrier();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wave_total[w];
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t b = b0 + i;
        if (b < n_bins) {
            L.lstart[b] = (uint16_t)run;
            run += L.slot[b];
        }
    }
    lds_barrier();   // every lstart is written, every count has been read
    if (mine.count) {
        if (mine.base + mine.count > mine.end) atomicOr(error, kErrScatterOverflow);
        L.slot[mine.bin] = mine.base - (uint32_t)L.lstart[mine.bin];
    }
    // a tile that touches more bins than the block has threads (unordered input): the rest, one after the other
    const uint32_t n_list = *B.n_list;
    for (uint32_t k = threadIdx.x + kStreamThreads; k < n_list; k += kStreamThreads) {
        const uint32_t b = B.list[k];
        const uint32_t c = L.slot[b];
        const uint32_t at = counter_index(b, group, n_bins);
        const uint32_t end = cursor_end[at];
        const uint32_t base = atomicAdd(&cursor[at], c);
        if (base + c > end) atomicOr(error, kErrScatterOverflow);
        L.slot[b] = base - (uint32_t)L.lstart[b];
    }
    lds_barrier();
}

// Regroup one per-particle value through LDS and write it out: every thread
// deposits its `live` values at their local positions, then lane t writes the
// values of local positions t, t+512, ... to their global slots.
template <typename T>
__device__ __forceinline__ void scatter_array(uint32_t* stage, const uint32_t lpos[4], const T val[4], int live,
                                              const uint32_t out_slot[4], uint32_t n_local, T* __restrict__ dst)
{
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < live) stage[lpos[k]] = __builtin_bit_cast(uint32_t, val[k]);
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t l = threadIdx.x + (uint32_t)i * kStreamThreads;
        if (l < n_local) dst[out_slot[i]] = __builtin_bit_cast(T, stage[l]);
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void scatter_load(const ReorderArgs& a, unsigned long long g, unsigned long long tail,
                                             bool full, bool is_tail, f32x4 V[6], uint32_t& t4, uint32_t orig[4])
{
    if (full) {
        // this block's tile through per-tile buffer resources (see tile_rsrc)
        const unsigned long long n_groups = a.n / 4ull;
        const unsigned long long g0 = (unsigned long long)blockIdx.x * kStreamThreads;   // g = g0 + threadIdx.x
        const uint32_t groups = (uint32_t)(n_groups - g0 < (unsigned long long)kStreamThreads ? n_groups - g0 : kStreamThreads);
        const uint32_t lane16 = threadIdx.x * 16u;
#pragma unroll
        for (int c = 0; c < 6; ++c) V[c] = tile_load4(tile_rsrc(a.src[c] + g0 * 4ull, groups * 16u), lane16);
        t4 = __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(a.src_ty + g0 * 4ull, groups * 4u), (int)(threadIdx.x * 4u), 0, 2);
        if (a.src_orig) {
            const f32x4 o = tile_load4(tile_rsrc(a.src_orig + g0 * 4ull, groups * 16u), lane16);
            orig[0] = __float_as_uint(o.x); orig[1] = __float_as_uint(o.y);
            orig[2] = __float_as_uint(o.z); orig[3] = __float_as_uint(o.w);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) orig[k] = (uint32_t)(g * 4ull + k);
        }
    } else if (is_tail) {
#pragma unroll
        for (int c = 0; c < 6; ++c) V[c][0] = a.src[c][tail];
        t4 = a.src_ty[tail];
        orig[0] = a.src_orig ? a.src_orig[tail] : (uint32_t)tail;
    } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) V[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// Cursor group of this block's tile (the appended tail block counts with group 0).
__device__ __forceinline__ uint32_t scatter_group(const ReorderArgs& a)
{
    return blockIdx.x < a.n_tiles ? blockIdx.x % a.groups : 0u;
}

// Writes state, type and caller index of the thread's `live` particles of a planned tile
// (scatter_plan_listed) to their slots, one array at a time through LDS.
__device__ __forceinline__ void scatter_write(const ReorderArgs& a, const ScatterLds& L,
                                              const f32x4 V[6], uint32_t t4, const uint32_t orig[4],
                                              const uint32_t key[4], const uint32_t rank[4], int live)
{
    __shared__ uint32_t n_local_sh;
    PCL_PHASE(5);
    uint32_t lpos[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lpos[k] = 0u;
        if (k < live) {
            lpos[k] = (uint32_t)L.lstart[key[k]] + rank[k];
            L.dest[lpos[k]] = L.slot[key[k]] + lpos[k];
        }
    }
    // number of particles in this tile (2048 except in the last blocks)
    const unsigned long long first = (unsigned long long)blockIdx.x * kStreamTile;
    const unsigned long long n_groups = a.n / 4ull;
    if (threadIdx.x == 0) {
        unsigned long long cnt = 0;
        if (first < n_groups * 4ull) cnt = (n_groups * 4ull - first < (unsigned long long)kStreamTile) ? n_groups * 4ull - first : kStreamTile;
        else cnt = a.n - n_groups * 4ull;   // the appended tail block
        n_local_sh = (uint32_t)cnt;
    }
    lds_barrier();
    const uint32_t n_local = n_local_sh;
    uint32_t out_slot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t l = threadIdx.x + (uint32_t)i * kStreamThreads;
        // a slot >= n means histogram and keys disagree (scatter_plan has flagged it); never write out of bounds
        out_slot[i] = l < n_local ? L.dest[l] : 0u;
        if (out_slot[i] >= a.n) out_slot[i] = (uint32_t)(a.n - 1);
    }
    PCL_PHASE(6);
#ifdef PCL_PHASE_CLOCKS
    const int mode = experiment_mode;
    if (mode == 1)
        for (int i = 0; i < 4; ++i) out_slot[i] = (uint32_t)first + threadIdx.x + (uint32_t)i * kStreamThreads;
    if (mode == 2 || mode == 3) {
        for (int c = 0; c < 7; ++c) {
            uint32_t* stage = L.stage[c & 1];
            for (int k = 0; k < 4; ++k)
                if (k < live) stage[lpos[k]] = c < 6 ? __float_as_uint(V[c][k]) : orig[k];
            lds_barrier();
            if (mode == 2 && 4u * threadIdx.x < n_local) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(stage + 4u * threadIdx.x);
                uint32_t* dst = c < 6 ? reinterpret_cast<uint32_t*>(a.dst[c]) : a.dst_orig;
                *reinterpret_cast<u32x4*>(dst + first + 4u * threadIdx.x) = v;
            }
        }
    } else {
#endif
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const float val[4] = {V[c][0], V[c][1], V[c][2], V[c][3]};
        scatter_array<float>(L.stage[c & 1], lpos, val, live, out_slot, n_local, a.dst[c]);
        PCL_PHASE(7 + c);
    }
    scatter_array<uint32_t>(L.stage[0], lpos, orig, live, out_slot, n_local, a.dst_orig);
#ifdef PCL_PHASE_CLOCKS
    }
#endif
    PCL_PHASE(13);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < live) L.stage_ty[lpos[k]] = (uint8_t)((t4 >> (8 * k)) & 0xFFu);
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t l = threadIdx.x + (uint32_t)i * kStreamThreads;
        if (l < n_local) a.dst_ty[out_slot[i]] = L.stage_ty[l];
    }
}

// One tile of 2048 consecutive source particles per block (plus one block for
// the ragged tail when n is not a multiple of 4).
__global__ __launch_bounds__(kStreamThreads) void reorder_scatter_kernel(ReorderArgs a, StepUniforms U,
                                                                         TileMap T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const ScatterLds L(lds_raw, T.n_bins);
    __shared__ uint32_t n_list_sh;
    // the list lives where the first staging buffer will be: that is not written before the plan is complete
    const ListedBins B{L.stage[0], &n_list_sh};

    const unsigned long long n_groups = a.n / 4ull;
    const unsigned long long g = (unsigned long long)blockIdx.x * kStreamThreads + threadIdx.x;
    const bool full = g < n_groups;
    // Up to 3 particles past the last full group: the host appends one block
    // that holds no full group, and its lanes 0..2 carry them.
    const unsigned long long tail = n_groups * 4ull + threadIdx.x;
    const bool is_tail = !full && blockIdx.x == gridDim.x - 1 && threadIdx.x < 3 && tail < a.n;
    const int live = full ? 4 : (is_tail ? 1 : 0);

    f32x4 V[6];
    uint32_t t4 = 0u, orig[4] = {0u, 0u, 0u, 0u};
    scatter_load(a, g, tail, full, is_tail, V, t4, orig);
    // the loads are in flight while the bin counters are cleared
    for (uint32_t b = threadIdx.x; b < T.n_bins; b += kStreamThreads) L.slot[b] = 0u;
    if (threadIdx.x == 0) n_list_sh = 0u;
    lds_barrier();
    uint32_t key[4], rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        key[k] = k < live ? tile_key(U, T, V[0][k], V[1][k], V[2][k], V[3][k], V[4][k], V[5][k]) : 0u;
        rank[k] = lds_rank_add_listed(L.slot, B, key[k], k < live);
    }
    lds_barrier();
    const uint32_t group = scatter_group(a);
    const Reservation mine = reserve_listed(L, B, T.n_bins, a.cursor, a.cursor_end, group);
    scatter_plan_listed(L, B, mine, T.n_bins, a.cursor, a.cursor_end, a.error, group);
    scatter_write(a, L, V, t4, orig, key, rank, live);
}

// The update and the scatter in one pass: every particle is advanced one step
// (exactly as step_stream_kernel does) and its new state is written to the slot
// its bin assigns, the bin being the field tile of the position it had BEFORE
// the step (one step stale, which is what the histogram in `cursor` counted).
// Saves the extra read and write of a separate reorder pass.
struct StepScatterArgs {
    ReorderArgs r;
    const f32x4* __restrict__ field;
    const float* __restrict__ field3;   // the field packed to 12 bytes per cell, or null (fetch_cell)
    const uint32_t* __restrict__ field_flag;   // see StreamArgs
    DiagPartial* partials;
};

// Order of a block's memory operations: memory returns in the order of issue, so the field fetches go out
// first (before the ranks), then the reservations of the plan, and the update waits for its fetches only
// (reservations first: the update stalls on them, 3.98 against 3.81 ms for 2^28 particles).
// PACKED: the update on the packed path (step_kernels.hpp: integrate2), for blocks whose lanes all hold
// four particles; a separate instantiation so that neither path's registers weigh on the other.
template <bool DIAG, bool PACKED>
__device__ __forceinline__ void step_scatter_block(const StepScatterArgs& sa, const StepUniforms& U, const TileMap& T,
                                                   unsigned char* lds_raw, uint32_t* n_list, float2* wave_mass_tab)
{
    const ReorderArgs& a = sa.r;
    const ScatterLds L(lds_raw, T.n_bins);
    const ListedBins B{L.stage[0], n_list};

    const unsigned long long n_groups = a.n / 4ull;
    const unsigned long long g = (unsigned long long)blockIdx.x * kStreamThreads + threadIdx.x;
    const bool full = g < n_groups;
    const unsigned long long tail = n_groups * 4ull + threadIdx.x;
    const bool is_tail = !full && blockIdx.x == gridDim.x - 1 && threadIdx.x < 3 && tail < a.n;
    const int live = PACKED ? 4 : (full ? 4 : (is_tail ? 1 : 0));

    PCL_PHASE(0);
    f32x4 V[6];
    uint32_t t4 = 0u, orig[4] = {0u, 0u, 0u, 0u};
    scatter_load(a, g, tail, PACKED || full, is_tail, V, t4, orig);
    // the loads are in flight while the bin counters are cleared
    for (uint32_t b = threadIdx.x; b < T.n_bins; b += kStreamThreads) L.slot[b] = 0u;
    if (threadIdx.x == 0) *n_list = 0u;
    MassTable M;
    if (PACKED) M.init(wave_mass_tab, U);
    lds_barrier();
    PCL_PHASE(1);
    f32x4 F[4];
    {
        uint32_t ci[4] = {0u, 0u, 0u, 0u};
        if (PACKED) {
            cell_index2(U, V[0].xy, V[1].xy, V[2].xy, ci);
            cell_index2(U, V[0].zw, V[1].zw, V[2].zw, ci + 2);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) ci[k] = cell_index(U, V[0][k], V[1][k], V[2][k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            F[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < live) F[k] = fetch_cell(sa.field, sa.field3, ci[k]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // the fetches are issued before anything below (see step4_packed)
    uint32_t key[4], rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        key[k] = k < live ? tile_key(U, T, V[0][k], V[1][k], V[2][k], V[3][k], V[4][k], V[5][k]) : 0u;
        rank[k] = lds_rank_add_listed(L.slot, B, key[k], k < live);
    }
    PCL_PHASE(2);
    lds_barrier();   // every rank is in: the counters hold the tile's totals
    const uint32_t group = scatter_group(a);
    const Reservation mine = reserve_listed(L, B, T.n_bins, a.cursor, a.cursor_end, group);
    PCL_PHASE(3);
    DiagAcc acc;
    if (DIAG) acc.init();
    if (PACKED) {
        float2 mt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) mt[k] = M.of((t4 >> (8 * k)) & 0xFFu);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = 2 * h, q = 2 * h + 1;
            f32x2 x = {V[0][p], V[0][q]}, y = {V[1][p], V[1][q]}, z = {V[2][p], V[2][q]};
            f32x2 vx = {V[3][p], V[3][q]}, vy = {V[4][p], V[4][q]}, vz = {V[5][p], V[5][q]};
            integrate2(U, f32x2{F[p].x, F[q].x}, f32x2{F[p].y, F[q].y}, f32x2{F[p].z, F[q].z},
                       f32x2{mt[p].x, mt[q].x}, f32x2{mt[p].y, mt[q].y}, x, y, z, vx, vy, vz);
            V[0][p] = x.x; V[0][q] = x.y; V[1][p] = y.x; V[1][q] = y.y; V[2][p] = z.x; V[2][q] = z.y;
            V[3][p] = vx.x; V[3][q] = vx.y; V[4][p] = vy.x; V[4][q] = vy.y; V[5][p] = vz.x; V[5][q] = vz.y;
            if (DIAG) {
                acc.add(mt[p].x, x.x, y.x, z.x, vx.x, vy.x, vz.x);
                acc.add(mt[q].x, x.y, y.y, z.y, vx.y, vy.y, vz.y);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < live) {
                const float m = mass_of(U, (t4 >> (8 * k)) & 0xFFu);
                float x = V[0][k], y = V[1][k], z = V[2][k], vx = V[3][k], vy = V[4][k], vz = V[5][k];
                integrate(U, F[k].x, F[k].y, F[k].z, m, 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
                V[0][k] = x; V[1][k] = y; V[2][k] = z; V[3][k] = vx; V[4][k] = vy; V[5][k] = vz;
                if (DIAG) acc.add(m, x, y, z, vx, vy, vz);
            }
        }
    }
    PCL_PHASE(4);
    scatter_plan_listed(L, B, mine, T.n_bins, a.cursor, a.cursor_end, a.error, group);
    scatter_write(a, L, V, t4, orig, key, rank, live);
    PCL_PHASE(16);
    if (DIAG) block_reduce_store<kStreamThreads>(acc, sa.partials + blockIdx.x);
}

// Two resident blocks per CU (up to 96 registers, no spills): held to the 80 registers that a third block
// needs, both instantiations spill and the step takes 4.06 instead of 3.81 ms (2^28 particles).
template <bool DIAG>
__global__ __launch_bounds__(kStreamThreads, 5) void step_scatter_kernel(StepScatterArgs sa, StepUniforms U, TileMap T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ uint32_t n_list_sh;
    __shared__ float2 mass_tab[kStreamThreads / 64][8];
    // blocks that lie wholly inside the full groups may take the packed path; the ragged end goes the scalar way
    const bool whole = ((unsigned long long)blockIdx.x + 1ull) * kStreamThreads <= sa.r.n / 4ull;
    if (whole && packed_path_ok(U, sa.field_flag))
        step_scatter_block<DIAG, true>(sa, U, T, lds_raw, &n_list_sh, mass_tab[threadIdx.x >> 6]);
    else
        step_scatter_block<DIAG, false>(sa, U, T, lds_raw, &n_list_sh, mass_tab[threadIdx.x >> 6]);
}

// Sort by cut-off cell for the binned pair term.  The order of a particle's
// partners decides the order its force terms are added in, so this sort is
// made reproducible: the particles are first put back into caller order, then
// (cell key, caller index) pairs go through a stable radix sort, which leaves
// every cell's particles in ascending caller index whatever order the memory
// was in before.  Keys use tile_key without lookahead.
__global__ __launch_bounds__(kBlockThreads) void cell_key_kernel(
    StateSoA s, unsigned long long n, StepUniforms U, TileMap C, uint32_t* __restrict__ key,
    uint32_t* __restrict__ index)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x; i < n; i += stride) {
        key[i] = tile_key(U, C, s.a[0][i], s.a[1][i], s.a[2][i], 0.0f, 0.0f, 0.0f);
        index[i] = (uint32_t)i;
    }
}

// bin_end[b] = number of sorted keys <= b (one past the last slot of cell b).
// occupied[slot] (may be null) is raised by the number of cells that hold at least one particle:
// what the host sizes up the cells' population with when it picks the sweep (see step_pairs_binned_wave_kernel).
__global__ __launch_bounds__(kBlockThreads) void cell_end_kernel(
    const uint32_t* __restrict__ sorted_key, unsigned long long n, uint32_t n_bins, uint32_t* __restrict__ bin_end,
    uint32_t* __restrict__ occupied)
{
    __shared__ uint32_t end_of[kBlockThreads];
    const uint32_t b = blockIdx.x * kBlockThreads + threadIdx.x;
    auto keys_up_to = [&](uint32_t bin) -> uint32_t {   // first position whose key is > bin
        unsigned long long lo = 0, hi = n;
        while (lo < hi) {
            const unsigned long long mid = lo + (hi - lo) / 2;
            if (sorted_key[mid] <= bin) lo = mid + 1; else hi = mid;
        }
        return (uint32_t)lo;
    };
    const uint32_t end = b < n_bins ? keys_up_to(b) : 0u;
    if (b < n_bins) bin_end[b] = end;
    if (!occupied) return;
    end_of[threadIdx.x] = end;
    __syncthreads();
    uint32_t before = 0u;   // the end of the cell below
    if (threadIdx.x) before = end_of[threadIdx.x - 1];
    else if (b && b < n_bins) before = keys_up_to(b - 1u);
    const unsigned long long holds = __builtin_amdgcn_ballot_w64(b < n_bins && end > before);
    if ((threadIdx.x & 63u) == 0u && holds) atomicAdd(occupied, (uint32_t)__popcll(holds));
}

// dst[slot] = src[from[slot]]: `from` holds the caller index of every sorted slot
// (the source is in caller order), which is also what dst_orig receives.
__global__ __launch_bounds__(kBlockThreads) void cell_gather_kernel(ReorderArgs a, const uint32_t* __restrict__ from)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x; i < a.n; i += stride) {
        const uint32_t j = from[i];
#pragma unroll
        for (int c = 0; c < 6; ++c) a.dst[c][i] = a.src[c][j];
        a.dst_ty[i] = a.src_ty[j];
        a.dst_orig[i] = j;
    }
}

// slot_of[orig[i]] = i: where each caller index lives.
__global__ __launch_bounds__(kBlockThreads) void invert_permutation_kernel(
    const uint32_t* __restrict__ orig, unsigned long long n, uint32_t* __restrict__ slot_of)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x; i < n; i += stride)
        slot_of[orig[i]] = (uint32_t)i;
}

// Back to caller order: dst[orig[i]] = src[i].
__global__ __launch_bounds__(kBlockThreads) void restore_order_kernel(
    StateSoA src, const uint8_t* __restrict__ src_ty, const uint32_t* __restrict__ orig,
    unsigned long long n, StateSoA dst, uint8_t* __restrict__ dst_ty)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x; i < n; i += stride) {
        const uint32_t d = orig[i];
#pragma unroll
        for (int c = 0; c < 6; ++c) dst.a[c][d] = src.a[c][i];
        dst_ty[d] = src_ty[i];
    }
}

// Caller-order gather of a range: out arrays indexed by (caller index - first).
__global__ __launch_bounds__(kBlockThreads) void gather_soa_kernel(
    StateSoA s, const uint8_t* __restrict__ ty, const uint32_t* __restrict__ slot_of,
    unsigned long long first, unsigned long long count, StateSoA out, uint8_t* __restrict__ out_ty)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x; i < count; i += stride) {
        const uint32_t slot = slot_of[first + i];
#pragma unroll
        for (int c = 0; c < 6; ++c)
            if (out.a[c]) out.a[c][i] = s.a[c][slot];
        if (out_ty) out_ty[i] = ty[slot];
    }
}

}  // namespace pcl
