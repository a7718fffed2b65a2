// placement.hip — does the speed of the step kernel's twelve streams depend on where the
// allocation lands?  One process allocates the slab again and again behind dummy allocations of
// growing size and times the same capped copy (256-thread blocks, five per CU) on each.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/placement tools/placement.hip
//   tools/placement [log2 n = 28] [shift unit in MiB = 2] [shifts = 40] [log2 of the virtual alignment; 0 = hipMalloc; -1 = offsets inside one hipMalloc; -2 = inside one physically contiguous allocation; -3 = staggers between the arrays of a physically contiguous allocation] [unit in bytes]
// With an alignment the slab comes from the virtual-memory API: an address range reserved at that
// alignment, one physical allocation mapped into it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

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
    unsigned long long n4;
};

__global__ __launch_bounds__(256) void deep_capped_kernel(Arrays a)
{
    extern __shared__ float pad[];
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= a.n4) return;
    f32x4 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = __builtin_nontemporal_load(a.src[k] + g);
    if (v[0].x == 12345.678f) pad[threadIdx.x] = v[1].x;
#pragma unroll
    for (int k = 0; k < 6; ++k) __builtin_nontemporal_store(v[k], a.dst[k] + g);
}

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const size_t unit = argc > 5 ? (size_t)atol(argv[5]) : (size_t)(argc > 2 ? atol(argv[2]) : 2) << 20;   // argv[5]: unit in bytes
    const int shifts = argc > 3 ? atoi(argv[3]) : 40;
    const int align_log2 = argc > 4 ? atoi(argv[4]) : 0;
    const unsigned long long n = 1ull << log2n;
    const size_t arr = n * 4 + (2u << 20) + 4096;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    if (align_log2 == -3) {
        // one physically contiguous allocation; the arrays 4n + 2 MiB + stagger apart, stagger = s * unit
        const size_t stagger0 = argc > 6 ? (size_t)atol(argv[6]) : 0;   // first stagger of the sweep, bytes
        const size_t max_arr = n * 4 + (2u << 20) + stagger0 + unit * (size_t)shifts;
        char* slab = nullptr;
        CK(hipExtMallocWithFlags((void**)&slab, max_arr * 12, hipDeviceMallocContiguous));
        for (int s = 0; s < shifts; ++s) {
            const size_t arr_s = n * 4 + (2u << 20) + stagger0 + unit * (size_t)s;
            Arrays a;
            for (int k = 0; k < 6; ++k) {
                a.src[k] = reinterpret_cast<const f32x4*>(slab + (size_t)k * arr_s);
                a.dst[k] = reinterpret_cast<f32x4*>(slab + (size_t)(6 + k) * arr_s);
            }
            a.n4 = n / 4;
            float total = 0.f;
            const int reps = 10;
            for (int r = 0; r < reps + 1; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(deep_capped_kernel, dim3((unsigned)(a.n4 / 256)), dim3(256), 160 * 1024 / 5 - 1024, 0, a);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 1) total += ms;
            }
            printf("stagger %8zu B  avg %.4f ms  %.3f TB/s\n", stagger0 + unit * (size_t)s, total / reps, (double)n * 48.0 / (total / reps * 1e-3) / 1e12);
            fflush(stdout);
        }
        return 0;
    }
    if (align_log2 < 0) {
        // one allocation with room to spare: the arrays start `s` units into it
        char* slab = nullptr;
        if (align_log2 == -2) {   // physically contiguous
            CK(hipExtMallocWithFlags((void**)&slab, arr * 12 + unit * (size_t)shifts, hipDeviceMallocContiguous));
            printf("physically contiguous allocation at %p\n", (void*)slab);
        } else {
            CK(hipMalloc((void**)&slab, arr * 12 + unit * (size_t)shifts));
        }
        for (int s = 0; s < shifts; ++s) {
            Arrays a;
            for (int k = 0; k < 6; ++k) {
                a.src[k] = reinterpret_cast<const f32x4*>(slab + unit * (size_t)s + (size_t)k * arr);
                a.dst[k] = reinterpret_cast<f32x4*>(slab + unit * (size_t)s + (size_t)(6 + k) * arr);
            }
            a.n4 = n / 4;
            float total = 0.f;
            const int reps = 6;
            for (int r = 0; r < reps + 2; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(deep_capped_kernel, dim3((unsigned)(a.n4 / 256)), dim3(256), 160 * 1024 / 5 - 1024, 0, a);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) total += ms;
            }
            printf("offset %6zu KiB  avg %.4f ms  %.3f TB/s\n", unit * (size_t)s >> 10, total / reps, (double)n * 48.0 / (total / reps * 1e-3) / 1e12);
            fflush(stdout);
        }
        return 0;
    }
    for (int s = 0; s < shifts; ++s) {
        char* dummy = nullptr;
        if (s) CK(hipMalloc((void**)&dummy, unit * (size_t)s));
        char* slab = nullptr;
        hipMemGenericAllocationHandle_t handle{};
        size_t mapped = 0;
        char* reserved = nullptr;
        if (align_log2 == 0) {
            CK(hipMalloc((void**)&slab, arr * 12));
        } else {
            hipMemAllocationProp prop{};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = 0;
            size_t gran = 0;
            CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            mapped = (arr * 12 + gran - 1) / gran * gran;
            if (s == 0) printf("granularity %zu\n", gran);
            // the alignment argument of hipMemAddressReserve is not honoured here: reserve more and map at an aligned address inside
            const size_t align = (size_t)1 << align_log2;
            CK(hipMemAddressReserve((void**)&reserved, mapped + align, 0, nullptr, 0));
            slab = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(reserved) + align - 1) / align * align);
            CK(hipMemCreate(&handle, mapped, &prop, 0));
            CK(hipMemMap(slab, mapped, 0, handle, 0));
            hipMemAccessDesc acc{};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(slab, mapped, &acc, 1));
        }
        Arrays a;
        for (int k = 0; k < 6; ++k) {
            a.src[k] = reinterpret_cast<const f32x4*>(slab + (size_t)k * arr);
            a.dst[k] = reinterpret_cast<f32x4*>(slab + (size_t)(6 + k) * arr);
        }
        a.n4 = n / 4;
        float best = 1e30f, total = 0.f;
        const int reps = 6;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(deep_capped_kernel, dim3((unsigned)(a.n4 / 256)), dim3(256), 160 * 1024 / 5 - 1024, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { total += ms; if (ms < best) best = ms; }
        }
        printf("dummy %6zu MiB  slab at %p  avg %.4f ms  %.3f TB/s   best %.4f ms\n", unit * (size_t)s >> 20, (void*)slab,
               total / reps, (double)n * 48.0 / (total / reps * 1e-3) / 1e12, best);
        fflush(stdout);
        if (align_log2 == 0) {
            CK(hipFree(slab));
        } else {
            CK(hipMemUnmap(slab, mapped));
            CK(hipMemRelease(handle));
            CK(hipMemAddressFree(reserved, mapped + ((size_t)1 << align_log2)));
        }
        if (dummy) CK(hipFree(dummy));
    }
    return 0;
}
