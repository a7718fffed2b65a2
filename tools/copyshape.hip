// copyshape.hip — what shape of copy the memory system of an MI355X rewards.
//
// Six input arrays and six output arrays of n floats each (the step kernel's
// stream without its arithmetic), copied by kernels that differ only in how
// the 16-byte accesses are spread over lanes, waves and blocks.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/copyshape tools/copyshape.hip
//   tools/copyshape [log2 n = 28] [repeats = 10] [stagger bytes = 4096] [anything: capped shapes only]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Arrays {
    const f32x4* src[6];
    f32x4* dst[6];
    unsigned long long n4;   // 16-byte groups per array
};

__device__ __forceinline__ f32x4 ld(const f32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st(f32x4* p, f32x4 v) { __builtin_nontemporal_store(v, p); }

// every lane: one group of each of the six arrays (the step kernel's shape)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void deep_kernel(Arrays a)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n4) return;
    f32x4 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = ld(a.src[k] + g);
#pragma unroll
    for (int k = 0; k < 6; ++k) st(a.dst[k] + g, v[k]);
}

// deep, with dynamic LDS requested only to cap the number of resident blocks per CU
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void deep_capped_kernel(Arrays a)
{
    extern __shared__ float pad[];
    const unsigned long long g = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n4) return;
    f32x4 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = ld(a.src[k] + g);
    if (v[0].x == 12345.678f) pad[threadIdx.x] = v[1].x;   // keeps the allocation alive
#pragma unroll
    for (int k = 0; k < 6; ++k) st(a.dst[k] + g, v[k]);
}

// deep, loads as LDS-DMA (global_load_lds_dwordx4: no register destination), every lane then reads its own
// 16 bytes back from LDS and stores them; one tile per block, `pad` bytes of extra dynamic LDS cap the residency
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void deep_glds_kernel(Arrays a)
{
    extern __shared__ __attribute__((aligned(16))) f32x4 stage[];   // [6][BLOCK]
    const unsigned long long g = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n4) return;   // n4 is a multiple of BLOCK here: whole waves leave together
    const unsigned wave_base = threadIdx.x & ~63u;
#pragma unroll
    for (int k = 0; k < 6; ++k)
        __builtin_amdgcn_global_load_lds(a.src[k] + g, (__attribute__((address_space(3))) void*)(stage + k * BLOCK + wave_base), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 6; ++k) st(a.dst[k] + g, stage[k * BLOCK + threadIdx.x]);
}

// every lane: one group of ONE array; consecutive blocks take the six arrays of one tile in turn,
// so the chip sees the same twelve streams advance together, from waves that hold one access each
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void shallow_kernel(Arrays a)
{
    const unsigned k = blockIdx.x % 6u;
    const unsigned long long g = (unsigned long long)(blockIdx.x / 6u) * BLOCK + threadIdx.x;
    if (g >= a.n4) return;
    st(a.dst[k] + g, ld(a.src[k] + g));
}

// every WAVE one array: a block of 6 waves covers 64 groups (256 particles) of all six arrays
__global__ __launch_bounds__(384) void wave_per_array_kernel(Arrays a)
{
    const unsigned k = threadIdx.x >> 6;
    const unsigned long long g = (unsigned long long)blockIdx.x * 64u + (threadIdx.x & 63u);
    if (g >= a.n4) return;
    st(a.dst[k] + g, ld(a.src[k] + g));
}

// as above, values handed through LDS between the loading and the storing wave (rotated by one array),
// two barriers: what a step kernel built from one-access waves would have to pay
__global__ __launch_bounds__(384) void wave_per_array_lds_kernel(Arrays a)
{
    __shared__ f32x4 stage[6][64];
    const unsigned k = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned long long g = (unsigned long long)blockIdx.x * 64u + lane;
    const bool live = g < a.n4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (live) v = ld(a.src[k] + g);
    stage[k][lane] = v;
    __syncthreads();
    // every lane reads one float of each array (a particle's six values), adds them up and writes a
    // per-particle result back: the dependency pattern of the update
    float* flat = reinterpret_cast<float*>(&stage[0][0]);
    float s = 0.f;
    if (threadIdx.x < 256) {
#pragma unroll
        for (int j = 0; j < 6; ++j) s += flat[j * 256 + threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
#pragma unroll
        for (int j = 0; j < 6; ++j) flat[j * 256 + threadIdx.x] += s * 0.0f;
    }
    __syncthreads();
    if (live) st(a.dst[k] + g, stage[k][lane]);
}

// two or three arrays per lane (in between)
template <int PER, int BLOCK>
__global__ __launch_bounds__(BLOCK) void partial_kernel(Arrays a)
{
    constexpr unsigned parts = 6 / PER;
    const unsigned part = blockIdx.x % parts;
    const unsigned long long g = (unsigned long long)(blockIdx.x / parts) * BLOCK + threadIdx.x;
    if (g >= a.n4) return;
    f32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = ld(a.src[part * PER + k] + g);
#pragma unroll
    for (int k = 0; k < PER; ++k) st(a.dst[part * PER + k] + g, v[k]);
}

// deep, but the three position arrays are stored before the velocity arrays are loaded
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void deep_split_kernel(Arrays a)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n4) return;
    f32x4 v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = ld(a.src[k] + g);
#pragma unroll
    for (int k = 0; k < 3; ++k) st(a.dst[k] + g, v[k]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = ld(a.src[3 + k] + g);
#pragma unroll
    for (int k = 0; k < 3; ++k) st(a.dst[3 + k] + g, v[k]);
}

// deep, capped, on a tile-interleaved layout: the six components of TP4 consecutive groups lie next to each
// other (component c of tile t at ((t * 6 + c) * TP4) groups), so a block reads one contiguous piece of memory
// and writes one, instead of six pieces 1 GiB apart each
template <int BLOCK, int TP4>
__global__ __launch_bounds__(BLOCK) void deep_tiled_kernel(const f32x4* in, f32x4* out, unsigned long long n4)
{
    extern __shared__ float pad[];
    const unsigned long long g = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= n4) return;
    const unsigned long long base = (g / TP4) * 6ull * TP4 + (g % TP4);
    f32x4 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = ld(in + base + (unsigned long long)k * TP4);
    if (v[0].x == 12345.678f) pad[threadIdx.x] = v[1].x;
#pragma unroll
    for (int k = 0; k < 6; ++k) st(out + base + (unsigned long long)k * TP4, v[k]);
}


int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const int repeats = argc > 2 ? atoi(argv[2]) : 10;
    const unsigned long long n = 1ull << log2n;
    const size_t stagger = argc > 3 ? (size_t)atol(argv[3]) : 4096;   // bytes between array starts beyond a multiple of 2 MiB (the library: 4096)
    const bool capped_only = argc > 4;
    const size_t arr = n * 4 + (2u << 20) + stagger;
    char* slab = nullptr;
    CK(hipMalloc((void**)&slab, arr * 12));
    CK(hipMemset(slab, 0, arr * 12));
    Arrays a;
    for (int k = 0; k < 6; ++k) {
        a.src[k] = reinterpret_cast<const f32x4*>(slab + (size_t)k * arr);
        a.dst[k] = reinterpret_cast<f32x4*>(slab + (size_t)(6 + k) * arr);
    }
    a.n4 = n / 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = (double)n * 48.0;
    auto run = [&](const char* name, auto launch) {
        launch();
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e30f, total = 0.f;
        for (int r = 0; r < repeats; ++r) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            total += ms;
            if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-44s avg %7.4f ms  %6.3f TB/s   best %7.4f ms  %6.3f TB/s\n", name, total / repeats,
               bytes / (total / repeats * 1e-3) / 1e12, best, bytes / (best * 1e-3) / 1e12);
        fflush(stdout);
    };
    const unsigned long long n4 = a.n4;
    printf("n = 2^%d floats per array, 6 arrays in, 6 out (%.2f GB per pass), stagger %zu\n", log2n, bytes / 1e9, stagger);
    if (argc > 5) {   // tile-interleaved layout against the separate arrays, same caps
        const f32x4* in = reinterpret_cast<const f32x4*>(slab);
        f32x4* out = reinterpret_cast<f32x4*>(slab + 6 * arr);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_capped_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_tiled_kernel<256, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_tiled_kernel<256, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_tiled_kernel<256, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_tiled_kernel<256, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        char name[96];
        for (int round = 0; round < 2; ++round)
            for (int bpc : {2, 3, 4, 0}) {
                const size_t lds = bpc ? (size_t)160 * 1024 / bpc - 1024 : 0;
                snprintf(name, sizeof name, "separate arrays, 256 threads, <= %d per CU", bpc);
                run(name, [&] { hipLaunchKernelGGL(deep_capped_kernel<256>, dim3((unsigned)(n4 / 256)), dim3(256), lds, 0, a); });
                snprintf(name, sizeof name, "tiles of 64 groups (one wave), <= %d per CU", bpc);
                run(name, [&] { hipLaunchKernelGGL((deep_tiled_kernel<256, 64>), dim3((unsigned)(n4 / 256)), dim3(256), lds, 0, in, out, n4); });
                snprintf(name, sizeof name, "tiles of 256 groups (one block), <= %d per CU", bpc);
                run(name, [&] { hipLaunchKernelGGL((deep_tiled_kernel<256, 256>), dim3((unsigned)(n4 / 256)), dim3(256), lds, 0, in, out, n4); });
                snprintf(name, sizeof name, "tiles of 1024 groups, <= %d per CU", bpc);
                run(name, [&] { hipLaunchKernelGGL((deep_tiled_kernel<256, 1024>), dim3((unsigned)(n4 / 256)), dim3(256), lds, 0, in, out, n4); });
                snprintf(name, sizeof name, "tiles of 16 groups (256 B runs), <= %d per CU", bpc);
                run(name, [&] { hipLaunchKernelGGL((deep_tiled_kernel<256, 16>), dim3((unsigned)(n4 / 256)), dim3(256), lds, 0, in, out, n4); });
            }
        return 0;
    }
    if (capped_only) {   // the two shapes the step kernel uses, for sweeps over the stagger
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_capped_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        run("deep, 256-thread blocks, <= 2 per CU", [&] { hipLaunchKernelGGL(deep_capped_kernel<256>, dim3((unsigned)(n4 / 256)), dim3(256), 160 * 512 - 1024, 0, a); });
        run("deep, 256-thread blocks, <= 5 per CU", [&] { hipLaunchKernelGGL(deep_capped_kernel<256>, dim3((unsigned)(n4 / 256)), dim3(256), 160 * 1024 / 5 - 1024, 0, a); });
        run("shallow, 256-thread blocks", [&] { hipLaunchKernelGGL(shallow_kernel<256>, dim3((unsigned)(n4 / 256 * 6)), dim3(256), 0, 0, a); });
        return 0;
    }
    run("deep, 256-thread blocks", [&] { hipLaunchKernelGGL(deep_kernel<256>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, a); });
    run("deep, 512-thread blocks", [&] { hipLaunchKernelGGL(deep_kernel<512>, dim3((unsigned)(n4 / 512)), dim3(512), 0, 0, a); });
    run("deep, 64-thread blocks", [&] { hipLaunchKernelGGL(deep_kernel<64>, dim3((unsigned)(n4 / 64)), dim3(64), 0, 0, a); });
    run("shallow, 256-thread blocks", [&] { hipLaunchKernelGGL(shallow_kernel<256>, dim3((unsigned)(n4 / 256 * 6)), dim3(256), 0, 0, a); });
    run("shallow, 512-thread blocks", [&] { hipLaunchKernelGGL(shallow_kernel<512>, dim3((unsigned)(n4 / 512 * 6)), dim3(512), 0, 0, a); });
    run("shallow, 64-thread blocks", [&] { hipLaunchKernelGGL(shallow_kernel<64>, dim3((unsigned)(n4 / 64 * 6)), dim3(64), 0, 0, a); });
    run("one array per wave, 384-thread blocks", [&] { hipLaunchKernelGGL(wave_per_array_kernel, dim3((unsigned)(n4 / 64)), dim3(384), 0, 0, a); });
    run("one array per wave + LDS hand-over", [&] { hipLaunchKernelGGL(wave_per_array_lds_kernel, dim3((unsigned)(n4 / 64)), dim3(384), 0, 0, a); });
    run("two arrays per lane, 256", [&] { hipLaunchKernelGGL((partial_kernel<2, 256>), dim3((unsigned)(n4 / 256 * 3)), dim3(256), 0, 0, a); });
    run("three arrays per lane, 256", [&] { hipLaunchKernelGGL((partial_kernel<3, 256>), dim3((unsigned)(n4 / 256 * 2)), dim3(256), 0, 0, a); });
    run("deep split 3+3, 512", [&] { hipLaunchKernelGGL(deep_split_kernel<512>, dim3((unsigned)(n4 / 512)), dim3(512), 0, 0, a); });
    run("deep split 3+3, 256", [&] { hipLaunchKernelGGL(deep_split_kernel<256>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, a); });
    for (int bpc : {1, 2, 3, 4, 6, 8}) {
        const size_t lds = (160u * 1024u) / bpc - 1024u;
        char name[96];
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_capped_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_capped_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_capped_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        snprintf(name, sizeof name, "deep, 256-thread blocks, <= %d per CU", bpc);
        run(name, [&] { hipLaunchKernelGGL(deep_capped_kernel<256>, dim3((unsigned)(n4 / 256)), dim3(256), lds, 0, a); });
        snprintf(name, sizeof name, "deep, 512-thread blocks, <= %d per CU", bpc);
        run(name, [&] { hipLaunchKernelGGL(deep_capped_kernel<512>, dim3((unsigned)(n4 / 512)), dim3(512), lds, 0, a); });
    }
    for (int bpc : {1, 2, 3, 4, 5, 6}) {
        const size_t lds = (160u * 1024u) / bpc - 1024u;
        char name[96];
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_glds_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_glds_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        snprintf(name, sizeof name, "deep by LDS-DMA, 256-thread blocks, <= %d per CU", bpc);
        run(name, [&] { hipLaunchKernelGGL(deep_glds_kernel<256>, dim3((unsigned)(n4 / 256)), dim3(256), lds, 0, a); });
        if (bpc <= 3) {
            snprintf(name, sizeof name, "deep by LDS-DMA, 512-thread blocks, <= %d per CU", bpc);
            run(name, [&] { hipLaunchKernelGGL(deep_glds_kernel<512>, dim3((unsigned)(n4 / 512)), dim3(512), lds, 0, a); });
        }
    }
    for (int bpc : {4, 8, 12, 16, 24, 32}) {
        const size_t lds = (160u * 1024u) / bpc - 512u;
        char name[96];
        snprintf(name, sizeof name, "deep, 64-thread blocks, <= %d per CU", bpc);
        run(name, [&] { hipLaunchKernelGGL(deep_capped_kernel<64>, dim3((unsigned)(n4 / 64)), dim3(64), lds, 0, a); });
    }
    return 0;
}
