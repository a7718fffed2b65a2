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

constexpr uint32_t kMaxReorderBins = 8192;
// Every bin's counter and cursor exist kCursorGroups times, one per residue of the tile index:
// the tiles that run at the same time fall into the same few bins, and their reservations on
// one address serialise.  Measured on the scatter step of 2^28 particles: 1 group 4.58 ms,
// 2: 4.29, 4: 3.96, 8: 3.96, 16: 4.25, 64: 5.14 (more groups cut every bin's range into more
// pieces, which shortens the contiguous runs the scatter writes and the order's locality).
// Counter of bin b for tile t: group t % groups (storage: counter_index); the scan runs in
// bin-major order, so every bin still owns one contiguous range of slots.
constexpr uint32_t kCursorGroups = 4;

// Where the counter (and cursor) of (bin, group) is stored: group-major, which puts the
// groups of one bin into different cache lines.
__device__ __forceinline__ uint32_t counter_index(uint32_t bin, uint32_t group, uint32_t n_bins)
{
    return group * n_bins + bin;
}

// Histogram of tile keys: one tile of kStreamThreads * 4 particles per block, LDS
// counters, one global atomic add per non-empty bin into the counter of the
// tile's group (see kCursorGroups; groups = 1 gives one counter per bin).
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

// After every thread has added its particles to slot[] (lds_rank_add): turn the
// per-bin counts into local start positions, reserve each non-empty bin's run of
// global slots with one atomic add, and leave in slot[b] the offset that maps a
// local position of bin b to its global slot.  Ends with a barrier.
// A reservation that runs past the end of its counter's range means the histogram and the keys
// disagree (they never do while both come from tile_key with one TileMap): particles would
// overwrite another bin's slots.  That sets kErrScatterOverflow in the error word.
__device__ __forceinline__ void scatter_plan(const ScatterLds& L, uint32_t n_bins, uint32_t* __restrict__ cursor,
                                             const uint32_t* __restrict__ cursor_end, uint32_t* error, uint32_t group)
{
    __shared__ uint32_t wave_total[kStreamThreads / 64];
    const uint32_t per = (n_bins + kStreamThreads - 1) / kStreamThreads;
    const uint32_t b0 = threadIdx.x * per;
    uint32_t sum = 0u;
    for (uint32_t i = 0; i < per; ++i)
        if (b0 + i < n_bins) sum += L.slot[b0 + i];
    // block-wide exclusive scan of `sum`
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
            const uint32_t c = L.slot[b];
            L.lstart[b] = (uint16_t)run;
            if (c) {
                const uint32_t at = counter_index(b, group, n_bins);
                const uint32_t end = cursor_end[at];
                const uint32_t base = atomicAdd(&cursor[at], c);
                if (base + c > end) atomicOr(error, kErrScatterOverflow);
                L.slot[b] = base - run;
            }
            run += c;
        }
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

// Plans the tile (scatter_plan) and writes state, type and caller index of the
// thread's `live` particles to their slots, one array at a time through LDS.
__device__ __forceinline__ void scatter_write(const ReorderArgs& a, const ScatterLds& L, uint32_t n_bins,
                                              const f32x4 V[6], uint32_t t4, const uint32_t orig[4],
                                              const uint32_t key[4], const uint32_t rank[4], int live)
{
    __shared__ uint32_t n_local_sh;
    scatter_plan(L, n_bins, a.cursor, a.cursor_end, a.error, blockIdx.x < a.n_tiles ? blockIdx.x % a.groups : 0u);
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
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const float val[4] = {V[c][0], V[c][1], V[c][2], V[c][3]};
        scatter_array<float>(L.stage[c & 1], lpos, val, live, out_slot, n_local, a.dst[c]);
    }
    scatter_array<uint32_t>(L.stage[0], lpos, orig, live, out_slot, n_local, a.dst_orig);
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
    lds_barrier();
    uint32_t key[4], rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // every lane takes part: lds_rank_add is a wave-wide operation
        key[k] = k < live ? tile_key(U, T, V[0][k], V[1][k], V[2][k], V[3][k], V[4][k], V[5][k]) : 0u;
        rank[k] = lds_rank_add(L.slot, key[k], k < live);
    }
    lds_barrier();
    scatter_write(a, L, T.n_bins, V, t4, orig, key, rank, live);
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
    DiagPartial* partials;
};

// With fused diagnostics the kernel needs 84 registers, four more than three resident blocks of 512 threads
// allow (6 waves per SIMD = 80 registers); held to 80 it keeps the third block: two blocks per CU cost 20 %.
template <bool DIAG>
__global__ __launch_bounds__(kStreamThreads, DIAG ? 6 : 1) void step_scatter_kernel(StepScatterArgs sa, StepUniforms U, TileMap T)
{
    const ReorderArgs& a = sa.r;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const ScatterLds L(lds_raw, T.n_bins);

    const unsigned long long n_groups = a.n / 4ull;
    const unsigned long long g = (unsigned long long)blockIdx.x * kStreamThreads + threadIdx.x;
    const bool full = g < n_groups;
    const unsigned long long tail = n_groups * 4ull + threadIdx.x;
    const bool is_tail = !full && blockIdx.x == gridDim.x - 1 && threadIdx.x < 3 && tail < a.n;
    const int live = full ? 4 : (is_tail ? 1 : 0);

    DiagAcc acc;
    if (DIAG) acc.init();

    f32x4 V[6];
    uint32_t t4 = 0u, orig[4] = {0u, 0u, 0u, 0u};
    scatter_load(a, g, tail, full, is_tail, V, t4, orig);
    // the loads are in flight while the bin counters are cleared
    for (uint32_t b = threadIdx.x; b < T.n_bins; b += kStreamThreads) L.slot[b] = 0u;
    lds_barrier();
    uint32_t key[4], rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // every lane takes part: lds_rank_add is a wave-wide operation
        key[k] = k < live ? tile_key(U, T, V[0][k], V[1][k], V[2][k], V[3][k], V[4][k], V[5][k]) : 0u;
        rank[k] = lds_rank_add(L.slot, key[k], k < live);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < live) {
            const f32x4 f = fetch_cell(sa.field, sa.field3, cell_index(U, V[0][k], V[1][k], V[2][k]));
            const float m = mass_of(U, (t4 >> (8 * k)) & 0xFFu);
            float x = V[0][k], y = V[1][k], z = V[2][k], vx = V[3][k], vy = V[4][k], vz = V[5][k];
            integrate(U, f.x, f.y, f.z, m, 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
            V[0][k] = x; V[1][k] = y; V[2][k] = z; V[3][k] = vx; V[4][k] = vy; V[5][k] = vz;
            if (DIAG) acc.add(m, x, y, z, vx, vy, vz);
        }
    }
    lds_barrier();
    scatter_write(a, L, T.n_bins, V, t4, orig, key, rank, live);
    if (DIAG) block_reduce_store<kStreamThreads>(acc, sa.partials + blockIdx.x);
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
__global__ __launch_bounds__(kBlockThreads) void cell_end_kernel(
    const uint32_t* __restrict__ sorted_key, unsigned long long n, uint32_t n_bins, uint32_t* __restrict__ bin_end)
{
    const uint32_t b = blockIdx.x * kBlockThreads + threadIdx.x;
    if (b >= n_bins) return;
    unsigned long long lo = 0, hi = n;   // first position whose key is > b
    while (lo < hi) {
        const unsigned long long mid = lo + (hi - lo) / 2;
        if (sorted_key[mid] <= b) lo = mid + 1; else hi = mid;
    }
    bin_end[b] = (uint32_t)lo;
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
