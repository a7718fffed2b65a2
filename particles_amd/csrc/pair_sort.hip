// pair_sort.hip — stable least-significant-digit radix sort of (key, value) pairs of 32-bit words,
// used to order particles by cut-off cell for the binned pair term (reorder_kernels.hpp:
// cell_key_kernel makes the keys, the values are the caller indices).  Stability is what the pair
// term needs from it: inside a cell the particles stay in ascending caller index, so the order in
// which a particle's partners are summed does not depend on the order the memory was in before.
//
// One pass per 8 bits of key, three launches per pass:
//   radix_hist_kernel    : every block counts the digits of its contiguous range of elements
//                          into table[digit * n_blocks + block]
//   radix_scan_*_kernel  : exclusive prefix over the table in that (digit-major) order, which is
//                          the order the output takes: all of digit 0 block by block, then digit 1 ...
//   radix_scatter_kernel : every block walks its range in chunks of 2048 elements; an element's slot is
//                          the block's running base of its digit + the number of elements of that
//                          digit before it in the chunk.  That number comes from a wave-wide match
//                          of the digit (8 ballots) and per-(round, wave, digit) counts in LDS that
//                          one thread per digit turns into offsets in element order.
// No counterpart in the reference (src/compute.wgsl:115-144 visits all pairs in buffer order).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcl {

namespace {

constexpr int kSortThreads = 256;                         // 4 waves
constexpr int kSortRounds = 8;                            // elements per thread and chunk
constexpr int kSortChunk = kSortThreads * kSortRounds;    // 2048
constexpr int kRadix = 256;                               // 8-bit digits
constexpr size_t kMaxSortBlocks = 65536;                  // bounds the table at 64 MiB
constexpr uint32_t kScanThreads = 1024;
constexpr uint32_t kScanPerThread = 16;
constexpr uint32_t kScanSegment = kScanThreads * kScanPerThread;   // table entries per scan block

__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const uint32_t* __restrict__ keys, unsigned long long n,
                                                                  unsigned long long per_block, unsigned shift,
                                                                  uint32_t n_blocks, uint32_t* __restrict__ table)
{
    __shared__ uint32_t hist[kRadix];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned long long begin = (unsigned long long)blockIdx.x * per_block;
    const unsigned long long end = begin + per_block < n ? begin + per_block : n;
    for (unsigned long long i = begin + threadIdx.x; i < end; i += kSortThreads)
        atomicAdd(&hist[(keys[i] >> shift) & (kRadix - 1u)], 1u);
    __syncthreads();
    table[(size_t)threadIdx.x * n_blocks + blockIdx.x] = hist[threadIdx.x];
}

// Inclusive scan of one value per thread over a block of kScanThreads, returning the exclusive
// prefix of this thread and, in `total`, the block's sum.  Ends with a barrier-protected state:
// may be called again right away.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_sum /* 16 */, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    __syncthreads();   // the previous call's readers are done with wave_sum
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t before = 0u, all = 0u;
#pragma unroll
    for (int w = 0; w < (int)(kScanThreads / 64); ++w) {
        const uint32_t s = wave_sum[w];
        if (w < wave) before += s;
        all += s;
    }
    total = all;
    return before + incl - v;
}

// Sum of every segment of kScanSegment table entries.
__global__ __launch_bounds__(kScanThreads) void radix_scan_sums_kernel(const uint32_t* __restrict__ table, uint32_t total_entries,
                                                                       uint32_t* __restrict__ segment_sum)
{
    __shared__ uint32_t wave_sum[kScanThreads / 64];
    const uint32_t base = blockIdx.x * kScanSegment;
    uint32_t v = 0u;
#pragma unroll
    for (uint32_t j = 0; j < kScanPerThread; ++j) {
        const uint32_t i = base + j * kScanThreads + threadIdx.x;
        if (i < total_entries) v += table[i];
    }
    uint32_t total;
    (void)block_exclusive_scan(v, wave_sum, total);
    if (threadIdx.x == 0) segment_sum[blockIdx.x] = total;
}

// Exclusive scan of the segment sums in place; one block (at most kScanThreads segments).
__global__ __launch_bounds__(kScanThreads) void radix_scan_segments_kernel(uint32_t* __restrict__ segment_sum, uint32_t n_segments)
{
    __shared__ uint32_t wave_sum[kScanThreads / 64];
    const uint32_t v = threadIdx.x < n_segments ? segment_sum[threadIdx.x] : 0u;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(v, wave_sum, total);
    if (threadIdx.x < n_segments) segment_sum[threadIdx.x] = excl;
}

// Exclusive scan inside every segment, in entry order, on top of the segment's own prefix.
__global__ __launch_bounds__(kScanThreads) void radix_scan_apply_kernel(uint32_t* __restrict__ table, uint32_t total_entries,
                                                                        const uint32_t* __restrict__ segment_prefix)
{
    __shared__ uint32_t wave_sum[kScanThreads / 64];
    const uint32_t base = blockIdx.x * kScanSegment;
    uint32_t run = segment_prefix[blockIdx.x];
    for (uint32_t j = 0; j < kScanPerThread; ++j) {
        const uint32_t i = base + j * kScanThreads + threadIdx.x;
        const uint32_t v = i < total_entries ? table[i] : 0u;
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(v, wave_sum, total);
        if (i < total_entries) table[i] = run + excl;
        run += total;
    }
}

__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ values_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ values_out, unsigned long long n, unsigned long long per_block, unsigned shift, uint32_t n_blocks,
    const uint32_t* __restrict__ table)
{
    __shared__ uint32_t base[kRadix];                                          // next output slot of every digit
    __shared__ uint16_t count[kSortRounds][kSortThreads / 64][kRadix];         // then: offset inside the chunk
    __shared__ uint32_t chunk_total[kRadix];
    base[threadIdx.x] = table[(size_t)threadIdx.x * n_blocks + blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long begin = (unsigned long long)blockIdx.x * per_block;
    const unsigned long long end = begin + per_block < n ? begin + per_block : n;
    for (unsigned long long chunk = begin; chunk < end; chunk += kSortChunk) {
        uint32_t* words = reinterpret_cast<uint32_t*>(&count[0][0][0]);
        for (uint32_t i = threadIdx.x; i < sizeof(count) / 4u; i += kSortThreads) words[i] = 0u;
        __syncthreads();   // also: base[] of the previous chunk is final
        uint32_t key[kSortRounds], value[kSortRounds], rank[kSortRounds];
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) {
            const unsigned long long i = chunk + (unsigned long long)r * kSortThreads + threadIdx.x;
            const bool active = i < end;
            key[r] = active ? keys_in[i] : 0u;
            value[r] = active ? values_in[i] : 0u;
            const uint32_t digit = (key[r] >> shift) & (kRadix - 1u);
            // the active lanes of this wave that hold the same digit
            unsigned long long same = __builtin_amdgcn_ballot_w64(active);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (digit >> b) & 1u;
                const unsigned long long with_bit = __builtin_amdgcn_ballot_w64(active && bit);
                same &= bit ? with_bit : ~with_bit;
            }
            rank[r] = (uint32_t)__popcll(same & below);
            if (active && rank[r] == 0u) count[r][wave][digit] = (uint16_t)__popcll(same);
        }
        __syncthreads();
        {   // thread d: the counts of digit d in element order (round, then wave) become offsets
            uint32_t run = 0u;
#pragma unroll
            for (int r = 0; r < kSortRounds; ++r)
#pragma unroll
                for (int w = 0; w < kSortThreads / 64; ++w) {
                    const uint32_t c = count[r][w][threadIdx.x];
                    count[r][w][threadIdx.x] = (uint16_t)run;
                    run += c;
                }
            chunk_total[threadIdx.x] = run;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) {
            const unsigned long long i = chunk + (unsigned long long)r * kSortThreads + threadIdx.x;
            if (i < end) {
                const uint32_t digit = (key[r] >> shift) & (kRadix - 1u);
                const uint32_t slot = base[digit] + count[r][wave][digit] + rank[r];
                keys_out[slot] = key[r];
                values_out[slot] = value[r];
            }
        }
        __syncthreads();   // every slot of this chunk is computed
        base[threadIdx.x] += chunk_total[threadIdx.x];
    }
}

struct SortPlan {
    unsigned long long per_block;
    uint32_t n_blocks;
    uint32_t table_entries;
    uint32_t n_segments;
    size_t temp_bytes;
};

SortPlan plan_of(size_t n)
{
    SortPlan p;
    size_t per = (n + kMaxSortBlocks - 1) / kMaxSortBlocks;
    per = (per + kSortChunk - 1) / kSortChunk * kSortChunk;
    if (per < (size_t)kSortChunk) per = kSortChunk;
    p.per_block = per;
    p.n_blocks = (uint32_t)((n + per - 1) / per);
    if (p.n_blocks == 0) p.n_blocks = 1;
    p.table_entries = (uint32_t)kRadix * p.n_blocks;                       // at most 2^24
    p.n_segments = (p.table_entries + kScanSegment - 1) / kScanSegment;    // at most 1024 = kScanThreads
    p.temp_bytes = ((size_t)p.table_entries + p.n_segments) * sizeof(uint32_t);
    return p;
}

}  // namespace

// Sorts n (key, value) pairs by bits [0, end_bit) of the key, stably, into keys_out / values_out.
// keys_in / values_in are overwritten (they serve as the second buffer of the passes).  With
// temp == nullptr only *temp_bytes (the size of the temporary storage) is set.
hipError_t radix_sort_pairs_u32(void* temp, size_t* temp_bytes, uint32_t* keys_in, uint32_t* keys_out, uint32_t* values_in,
                                uint32_t* values_out, size_t n, unsigned end_bit, hipStream_t stream)
{
    const SortPlan p = plan_of(n);
    if (!temp) {
        *temp_bytes = p.temp_bytes;
        return hipSuccess;
    }
    if (*temp_bytes < p.temp_bytes) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    uint32_t* table = static_cast<uint32_t*>(temp);
    uint32_t* segment_sum = table + p.table_entries;
    const unsigned passes = end_bit == 0 ? 1u : (end_bit + 7u) / 8u;
    uint32_t *ks = keys_in, *vs = values_in, *kd = keys_out, *vd = values_out;
    for (unsigned pass = 0; pass < passes; ++pass) {
        const unsigned shift = 8u * pass;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(p.n_blocks), dim3(kSortThreads), 0, stream, (const uint32_t*)ks,
                           (unsigned long long)n, p.per_block, shift, p.n_blocks, table);
        hipLaunchKernelGGL(radix_scan_sums_kernel, dim3(p.n_segments), dim3(kScanThreads), 0, stream, (const uint32_t*)table,
                           p.table_entries, segment_sum);
        hipLaunchKernelGGL(radix_scan_segments_kernel, dim3(1), dim3(kScanThreads), 0, stream, segment_sum, p.n_segments);
        hipLaunchKernelGGL(radix_scan_apply_kernel, dim3(p.n_segments), dim3(kScanThreads), 0, stream, table, p.table_entries,
                           (const uint32_t*)segment_sum);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(p.n_blocks), dim3(kSortThreads), 0, stream, (const uint32_t*)ks,
                           (const uint32_t*)vs, kd, vd, (unsigned long long)n, p.per_block, shift, p.n_blocks,
                           (const uint32_t*)table);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        uint32_t* t = ks; ks = kd; kd = t;
        t = vs; vs = vd; vd = t;
    }
    if (ks != keys_out) {   // an even number of passes ends in the input buffers
        if (hipError_t e = hipMemcpyAsync(keys_out, ks, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream); e != hipSuccess) return e;
        if (hipError_t e = hipMemcpyAsync(values_out, vs, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream); e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace pcl
