// kbench.hip — standalone timing harness for variants of the streaming kernel
// (development tool; not part of the library).  Build:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Iparticles_amd/csrc tools/kbench.hip -o tools/kbench
// Run on the GPU box: tools/kbench [log2n] [field_side] [pos_extent]
#include "step_kernels.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace pcl;

// G = float4 groups per lane per array (particles per lane = 4*G)
// GPOL: cache-policy bits of the field fetch (buffer_load aux: 1 = sc0, 2 = nt, 16 = sc1); -1 = plain global_load
// SPOL: the same bits for the six particle-array loads (-1 = the library's non-temporal global_load)
template <int G, bool NT_LD, bool NT_ST, bool PERSIST, bool GATHER, int BLOCK, int GPOL = -1, bool XCD = false, int SPOL = -1>
__global__ __launch_bounds__(BLOCK) void variant_kernel(StreamArgs a, StepUniforms U)
{
    __amdgpu_buffer_rsrc_t srsrc[6];
    if (SPOL >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            srsrc[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src[k]), 0, (unsigned)(a.n * 4ull), 0x00020000);
    }
    auto sload = [&](int k, unsigned long long g) -> f32x4 {
        if (SPOL < 0) return load4<NT_LD>(a.src[k], g);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(srsrc[k], (int)(unsigned)(g * 16ull), 0, SPOL < 0 ? 0 : SPOL);
        return __builtin_bit_cast(f32x4, raw);
    };
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f32x4*>(a.field), 0, (U.last_cell + 1u) * 16u, 0x00020000);
    const unsigned long long n_groups = a.n / 4ull;  // n is a multiple of 4*G*BLOCK here
    const unsigned long long tile = (unsigned long long)BLOCK * G;
    const unsigned long long n_tiles = n_groups / tile;
    // XCD: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD one contiguous eighth of the tiles
    const unsigned long long first_tile = XCD ? (unsigned long long)(blockIdx.x & 7u) * (n_tiles >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    for (unsigned long long t = first_tile; t < n_tiles; t += PERSIST ? gridDim.x : n_tiles) {
        f32x4 X[G], Y[G], Z[G], VX[G], VY[G], VZ[G];
        uint32_t T[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const unsigned long long g = t * tile + (unsigned long long)j * BLOCK + threadIdx.x;
            X[j] = sload(0, g);
            Y[j] = sload(1, g);
            Z[j] = sload(2, g);
            VX[j] = sload(3, g);
            VY[j] = sload(4, g);
            VZ[j] = sload(5, g);
            T[j] = NT_LD ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.ty) + g)
                         : reinterpret_cast<const uint32_t*>(a.ty)[g];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            f32x4 F[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t c = cell_index(U, X[j][k], Y[j][k], Z[j][k]);
                const uint32_t cc = GATHER ? c : (c & 63u);
                if (GPOL < 0) F[k] = a.field[cc];
                else {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cc * 16u), 0, GPOL < 0 ? 0 : GPOL);
                    F[k] = __builtin_bit_cast(f32x4, raw);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float m = mass_of(U, (T[j] >> (8 * k)) & 0xFFu);
                float x = X[j][k], y = Y[j][k], z = Z[j][k], vx = VX[j][k], vy = VY[j][k], vz = VZ[j][k];
                integrate(U, F[k].x, F[k].y, F[k].z, m, 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
                X[j][k] = x; Y[j][k] = y; Z[j][k] = z; VX[j][k] = vx; VY[j][k] = vy; VZ[j][k] = vz;
            }
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const unsigned long long g = t * tile + (unsigned long long)j * BLOCK + threadIdx.x;
            store4<NT_ST>(a.dst[0], g, X[j]);
            store4<NT_ST>(a.dst[1], g, Y[j]);
            store4<NT_ST>(a.dst[2], g, Z[j]);
            store4<NT_ST>(a.dst[3], g, VX[j]);
            store4<NT_ST>(a.dst[4], g, VY[j]);
            store4<NT_ST>(a.dst[5], g, VZ[j]);
        }
    }
}

// Pure copy of the same byte mix (6 arrays in, 6 out, 1 B type in): the stream's ceiling.
template <bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void copy_kernel(StreamArgs a)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n / 4ull) return;
    f32x4 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = load4<NT>(a.src[k], g);
    const uint32_t t = reinterpret_cast<const uint32_t*>(a.ty)[g];
    v[0].x += __uint_as_float(t & 1u);
#pragma unroll
    for (int k = 0; k < 6; ++k) store4<NT>(a.dst[k], g, v[k]);
}

// Copy of the same byte mix in a chunked layout: chunks of CH particles, each
// [x CH][y CH][z CH][vx CH][vy CH][vz CH][type CH bytes], one block per chunk.
template <bool NT, int CH>
__global__ __launch_bounds__(CH / 4) void copy_chunked_kernel(const char* __restrict__ s, char* __restrict__ d)
{
    const size_t chunk = (size_t)CH * 25;
    const char* sb = s + (size_t)blockIdx.x * chunk;
    char* db = d + (size_t)blockIdx.x * chunk;
    f32x4 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = load4<NT>(reinterpret_cast<const float*>(sb + (size_t)k * CH * 4), threadIdx.x);
    const uint32_t t = reinterpret_cast<const uint32_t*>(sb + (size_t)CH * 24)[threadIdx.x];
    v[0].x += __uint_as_float(t & 1u);
#pragma unroll
    for (int k = 0; k < 6; ++k) store4<NT>(reinterpret_cast<float*>(db + (size_t)k * CH * 4), threadIdx.x, v[k]);
}

// Copy of K of the six arrays (float4 per lane per array): how the streaming rate depends on
// the number of loads and stores one lane issues.
template <int K, int BLOCK>
__global__ __launch_bounds__(BLOCK) void copyk_kernel(StreamArgs a)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n / 4ull) return;   // every array holds a.n floats
    f32x4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = load4<true>(a.src[k], g);
#pragma unroll
    for (int k = 0; k < K; ++k) store4<true>(a.dst[k], g, v[k]);
}

// Plain float4 copy a -> b
template <bool NT>
__global__ __launch_bounds__(256) void copy1_kernel(const f32x4* __restrict__ s, f32x4* __restrict__ d, unsigned long long n4)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (g < n4) {
        f32x4 v = NT ? __builtin_nontemporal_load(s + g) : s[g];
        if (NT) __builtin_nontemporal_store(v, d + g); else d[g] = v;
    }
}

// Persistent blocks with the next tile's loads in flight while the current tile is updated:
// per iteration [field fetches of tile i] [array loads of tile i + grid] wait for the fetches,
// update, [stores of tile i].  vmcnt retires in issue order, so the fetches are issued before the
// prefetch and waiting for them leaves the prefetch in flight.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void pipelined_kernel(StreamArgs a, StepUniforms U)
{
    const unsigned long long n_tiles = a.n / 4ull / BLOCK;   // n is a multiple of 4 * BLOCK here
    const uint32_t lane16 = threadIdx.x * 16u;
    unsigned long long t = blockIdx.x;
    if (t >= n_tiles) return;
    f32x4 C[6];
    uint32_t ct;
    {
        const unsigned long long g0 = t * BLOCK;
#pragma unroll
        for (int k = 0; k < 6; ++k) C[k] = tile_load4(tile_rsrc(a.src[k] + g0 * 4ull, BLOCK * 16u), lane16);
        ct = __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(a.ty + g0 * 4ull, BLOCK * 4u), (int)(threadIdx.x * 4u), 0, 2);
    }
    for (;;) {
        f32x4 F[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) F[k] = a.field[cell_index(U, C[0][k], C[1][k], C[2][k])];
        const unsigned long long tn = t + gridDim.x;
        const bool more = tn < n_tiles;
        f32x4 N[6];
        uint32_t nt = 0u;
        {
            // unconditional (the last iteration reloads its own tile) so that the wait counts stay exact
            const unsigned long long g0 = (more ? tn : t) * BLOCK;
#pragma unroll
            for (int k = 0; k < 6; ++k) N[k] = tile_load4(tile_rsrc(a.src[k] + g0 * 4ull, BLOCK * 16u), lane16);
            nt = __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(a.ty + g0 * 4ull, BLOCK * 4u), (int)(threadIdx.x * 4u), 0, 2);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float m = mass_of(U, (ct >> (8 * k)) & 0xFFu);
            float x = C[0][k], y = C[1][k], z = C[2][k], vx = C[3][k], vy = C[4][k], vz = C[5][k];
            integrate(U, F[k].x, F[k].y, F[k].z, m, 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
            C[0][k] = x; C[1][k] = y; C[2][k] = z; C[3][k] = vx; C[4][k] = vy; C[5][k] = vz;
        }
        {
            const unsigned long long g0 = t * BLOCK;
#pragma unroll
            for (int k = 0; k < 6; ++k) tile_store4(tile_rsrc(a.dst[k] + g0 * 4ull, BLOCK * 16u), lane16, C[k]);
        }
        if (!more) break;
#pragma unroll
        for (int k = 0; k < 6; ++k) C[k] = N[k];
        ct = nt;
        t = tn;
    }
}

// Two particles per lane (8-byte accesses): half the bytes in flight per wave.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void half_kernel(StreamArgs a, StepUniforms U)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const unsigned long long p0 = (unsigned long long)blockIdx.x * BLOCK * 2ull;   // n is a multiple of 2 * BLOCK here
    const uint32_t lane8 = threadIdx.x * 8u;
    f32x2 V[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        V[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(tile_rsrc(a.src[k] + p0, BLOCK * 8u), (int)lane8, 0, 2));
    }
    const uint32_t t2 = __builtin_amdgcn_raw_buffer_load_b16(tile_rsrc(a.ty + p0, BLOCK * 2u), (int)(threadIdx.x * 2u), 0, 2);
    f32x4 F[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) F[k] = a.field[cell_index(U, V[0][k], V[1][k], V[2][k])];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float m = mass_of(U, (t2 >> (8 * k)) & 0xFFu);
        float x = V[0][k], y = V[1][k], z = V[2][k], vx = V[3][k], vy = V[4][k], vz = V[5][k];
        integrate(U, F[k].x, F[k].y, F[k].z, m, 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
        V[0][k] = x; V[1][k] = y; V[2][k] = z; V[3][k] = vx; V[4][k] = vy; V[5][k] = vz;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, V[k]), tile_rsrc(a.dst[k] + p0, BLOCK * 8u), (int)lane8, 0, 2);
    }
}

// The library kernel's access shape with the cache-policy bits of the six stores (and loads) as
// template parameters (buffer aux: 1 = sc0, 2 = nt, 16 = sc1).
template <int LDPOL, int STPOL>
__global__ __launch_bounds__(512) void policy_kernel(StreamArgs a, StepUniforms U)
{
    const unsigned long long g0 = (unsigned long long)blockIdx.x * 512ull;   // n is a multiple of 2048 here
    const uint32_t lane16 = threadIdx.x * 16u;
    f32x4 V[6];
#pragma unroll
    for (int k = 0; k < 6; ++k)
        V[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tile_rsrc(a.src[k] + g0 * 4ull, 512u * 16u), (int)lane16, 0, LDPOL));
    const uint32_t t4 = __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(a.ty + g0 * 4ull, 512u * 4u), (int)(threadIdx.x * 4u), 0, LDPOL);
    f32x4 F[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) F[k] = a.field[cell_index(U, V[0][k], V[1][k], V[2][k])];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float m = mass_of(U, (t4 >> (8 * k)) & 0xFFu);
        float x = V[0][k], y = V[1][k], z = V[2][k], vx = V[3][k], vy = V[4][k], vz = V[5][k];
        integrate(U, F[k].x, F[k].y, F[k].z, m, 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
        V[0][k] = x; V[1][k] = y; V[2][k] = z; V[3][k] = vx; V[4][k] = vy; V[5][k] = vz;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, V[k]), tile_rsrc(a.dst[k] + g0 * 4ull, 512u * 16u), (int)lane16, 0, STPOL);
}

// The library kernel's work with the order of its memory requests and waits spelled out
// (sched_barrier keeps the compiler from moving instructions across the marked points):
//   MODE 0  as the compiler schedules it (the library kernel's shape)
//   MODE 1  all seven array loads returned before the first index is computed, four fetches back to back
//   MODE 2  positions returned -> four indices -> four fetches back to back (velocities still in flight)
//   MODE 3  positions loaded and returned, four fetches issued, THEN velocities and types loaded
//   MODE 4  as 2 with the masses pinned in scalar registers (no indexed load of the mass)
//   MODE 5  as 1 with pinned masses
// PIN (modes 4, 5) and the 16-byte fetch are the only code differences besides the order.
template <int MODE, int SLEEP_MID = 0, int SLEEP_END = 0>
__global__ __launch_bounds__(512) void order_kernel(StreamArgs a, StepUniforms U)
{
    const unsigned long long g0 = (unsigned long long)blockIdx.x * 512ull;   // n is a multiple of 2048 here
    const uint32_t lane16 = threadIdx.x * 16u;
    f32x4 V[6];
    uint32_t t4;
    auto ld = [&](int k) { V[k] = tile_load4(tile_rsrc(a.src[k] + g0 * 4ull, 512u * 16u), lane16); };
    auto ldt = [&]() { t4 = __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(a.ty + g0 * 4ull, 512u * 4u), (int)(threadIdx.x * 4u), 0, 2); };
    ld(0); ld(1); ld(2);
    if (MODE != 3) { ld(3); ld(4); ld(5); ldt(); }
    if (MODE == 1 || MODE == 5) { __builtin_amdgcn_s_waitcnt(0x0F70); __builtin_amdgcn_sched_barrier(0); }
    f32x4 F[4];
    if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) F[k] = a.field[cell_index(U, V[0][k], V[1][k], V[2][k])];
    } else {
        uint32_t c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = cell_index(U, V[0][k], V[1][k], V[2][k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) F[k] = a.field[c[k]];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 3) { ld(3); ld(4); ld(5); ldt(); __builtin_amdgcn_sched_barrier(0); }
    if (SLEEP_MID) { __builtin_amdgcn_s_sleep(SLEEP_MID); __builtin_amdgcn_sched_barrier(0); }
    float m0 = U.mass[0], m1 = U.mass[1], m2 = U.mass[2], m3 = U.mass[3], m4 = U.mass[4];
    if (MODE >= 4) asm("" : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3), "+s"(m4));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t ty = (t4 >> (8 * k)) & 0xFFu;
        float m = m4;
        m = ty == 3u ? m3 : m;
        m = ty == 2u ? m2 : m;
        m = ty == 1u ? m1 : m;
        m = ty == 0u ? m0 : m;
        float x = V[0][k], y = V[1][k], z = V[2][k], vx = V[3][k], vy = V[4][k], vz = V[5][k];
        integrate(U, F[k].x, F[k].y, F[k].z, m, 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
        V[0][k] = x; V[1][k] = y; V[2][k] = z; V[3][k] = vx; V[4][k] = vy; V[5][k] = vz;
    }
    if (SLEEP_END) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < SLEEP_END / 100; ++q) __builtin_amdgcn_s_sleep(100);
        if (SLEEP_END % 100) __builtin_amdgcn_s_sleep(SLEEP_END % 100);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) tile_store4(tile_rsrc(a.dst[k] + g0 * 4ull, 512u * 16u), lane16, V[k]);
}

// Particles whose memory order follows the field's tile order: particle i sits
// uniformly inside tile (i * n_tiles / n) of tile_cells x tile_cells cells,
// tiles enumerated x-major (emulates a cell-sorted particle set).
__global__ void generate_sorted_kernel(StateSoA s, uint8_t* ty, unsigned long long n, unsigned side, unsigned tile_cells, float radius)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned tiles_per_side = side / tile_cells;
    const unsigned long long n_tiles = (unsigned long long)tiles_per_side * tiles_per_side;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long t = (unsigned long long)((__uint128_t)i * n_tiles / n);
        const unsigned tx = (unsigned)(t / tiles_per_side), tyy = (unsigned)(t % tiles_per_side);
        const float w = 2.0f * radius / (float)tiles_per_side;
        s.a[0][i] = -radius + w * ((float)tx + synth_uniform(1, i, 0));
        s.a[1][i] = -radius + w * ((float)tyy + synth_uniform(1, i, 1));
        s.a[2][i] = 2.0f * radius * (synth_uniform(1, i, 2) - 0.5f);
        for (int c = 0; c < 3; ++c) s.a[3 + c][i] = 10.0f * (synth_uniform(1, i, 3 + c) - 0.5f);
        ty[i] = (uint8_t)(i % 5ull);
    }
}

struct Ctx {
    unsigned long long n;
    StreamArgs fwd{}, bwd{};   // zero: no packed field copy, no histogram
    StepUniforms U;
    hipEvent_t e0, e1;
    StateSoA s0;
    uint8_t* ty;
    float extent;
    void regen()
    {
        hipLaunchKernelGGL(generate_kernel, dim3(4096), dim3(256), 0, 0, s0, ty, n, 12345ull, 0ull, extent, 10.0f);
        CK(hipDeviceSynchronize());
    }
};

template <typename F>
static float time_ms(Ctx& c, F launch, int reps = 20)
{
    for (int i = 0; i < 3; ++i) launch(i & 1 ? c.bwd : c.fwd);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(c.e0));
    for (int i = 0; i < reps; ++i) launch(i & 1 ? c.bwd : c.fwd);
    CK(hipEventRecord(c.e1));
    CK(hipEventSynchronize(c.e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, c.e0, c.e1));
    CK(hipGetLastError());
    return ms / reps;
}

static void report(const char* name, Ctx& c, float ms, double bytes_per_particle = 49.0)
{
    printf("%-58s %8.4f ms  %8.1f GB/s  %7.2f Gp/s\n", name, ms, c.n * bytes_per_particle / ms / 1e6, c.n / ms / 1e6);
    fflush(stdout);
}

template <int G, bool NT_LD, bool NT_ST, bool PERSIST, bool GATHER, int BLOCK, int GPOL = -1>
static void run_variant(Ctx& c, const char* label, int blocks_per_cu = 8)
{
    c.regen();
    const unsigned long long n_tiles = c.n / 4ull / ((unsigned long long)BLOCK * G);
    const unsigned grid = PERSIST ? (unsigned)std::min<unsigned long long>(n_tiles, 256ull * blocks_per_cu) : (unsigned)n_tiles;
    float ms = time_ms(c, [&](StreamArgs& a) {
        hipLaunchKernelGGL((variant_kernel<G, NT_LD, NT_ST, PERSIST, GATHER, BLOCK, GPOL>), dim3(grid), dim3(BLOCK), 0, 0, a, c.U);
    });
    char name[128];
    snprintf(name, sizeof name, "%s G=%d ntld=%d ntst=%d persist=%d(%u) gather=%d block=%d gpol=%d", label, G, NT_LD, NT_ST, PERSIST, grid, GATHER, BLOCK, GPOL);
    report(name, c, ms);
}

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 26;
    const unsigned side = argc > 2 ? atoi(argv[2]) : 1024;
    const float extent = argc > 3 ? atof(argv[3]) : 20.0f;
    const size_t stagger = argc > 4 ? atol(argv[4]) : 0;  // bytes between consecutive array bases (mod 2 MiB)
    Ctx c;
    c.n = 1ull << log2n;
    CK(hipEventCreate(&c.e0));
    CK(hipEventCreate(&c.e1));
    const size_t n_cells = (size_t)side * side;

    // one slab, arrays optionally staggered
    const size_t arr_bytes = c.n * 4 + (2u << 20);
    char* slab = nullptr;
    CK(hipMalloc((void**)&slab, arr_bytes * 12 + c.n + (64u << 20)));
    float* st[2][6];
    for (int b = 0; b < 2; ++b)
        for (int k = 0; k < 6; ++k) st[b][k] = (float*)(slab + (size_t)(b * 6 + k) * arr_bytes + (size_t)(b * 6 + k) * stagger);
    uint8_t* ty = (uint8_t*)(slab + 12 * arr_bytes + 12 * stagger);
    f32x4* field = nullptr;
    CK(hipMalloc((void**)&field, n_cells * 16));
    std::vector<float> hf(n_cells * 4);
    for (size_t i = 0; i < hf.size(); ++i) hf[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f;
    CK(hipMemcpy(field, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));

    for (int k = 0; k < 6; ++k) c.s0.a[k] = st[0][k];
    c.ty = ty;
    c.extent = extent;
    c.regen();

    for (int k = 0; k < 6; ++k) {
        c.fwd.src[k] = st[0][k]; c.fwd.dst[k] = st[1][k];
        c.bwd.src[k] = st[1][k]; c.bwd.dst[k] = st[0][k];
    }
    c.fwd.ty = c.bwd.ty = ty;
    c.fwd.field = c.bwd.field = field;
    c.fwd.n = c.bwd.n = c.n;
    c.fwd.partials = c.bwd.partials = nullptr;
    c.fwd.lds_cells = c.bwd.lds_cells = 0;

    StepUniforms& U = c.U;
    const float r = 10.0f, dt = 1.0f / 60.0f;
    U.decay = expf(-dt); U.dt = dt; U.max_velocity = 100.0f; U.radius = r; U.two_radius = 2.0f * r;
    U.slh = (r * 2.0f) * 0.5f; U.neg_slh = -U.slh; U.span_a = U.slh - U.neg_slh;
    U.span_b[0] = (float)side; U.span_b[1] = (float)side; U.span_b[2] = 1.0f;
    U.dim_y = side; U.dim_z = 1; U.last_cell = (uint32_t)(n_cells - 1); U.degenerate = 0;
    const float masses[5] = {1.0f, 0.5f, 2.0f, 1.5f, 0.75f};
    for (int k = 0; k < 5; ++k) U.mass[k] = masses[k];

    printf("n=2^%d field=%u^2 extent=%.1f stagger=%zu\n", log2n, side, extent, stagger);

    // ceilings
    {
        const unsigned long long n4 = c.n;  // copy n*16 bytes: use 4 arrays' worth
        float ms = time_ms(c, [&](StreamArgs& a) {
            hipLaunchKernelGGL(copy1_kernel<false>, dim3((unsigned)((n4 / 4 + 255) / 256)), dim3(256), 0, 0,
                               (const f32x4*)a.src[0], (f32x4*)a.dst[0], n4 / 4);
        });
        report("float4 copy (1 array, plain)", c, ms, 8.0);
        ms = time_ms(c, [&](StreamArgs& a) {
            hipLaunchKernelGGL(copy1_kernel<true>, dim3((unsigned)((n4 / 4 + 255) / 256)), dim3(256), 0, 0,
                               (const f32x4*)a.src[0], (f32x4*)a.dst[0], n4 / 4);
        });
        report("float4 copy (1 array, nt)", c, ms, 8.0);
        ms = time_ms(c, [&](StreamArgs& a) {
            hipLaunchKernelGGL((copy_kernel<false, 256>), dim3((unsigned)(c.n / 4 / 256)), dim3(256), 0, 0, a);
        });
        report("13-stream copy (plain)", c, ms);
        ms = time_ms(c, [&](StreamArgs& a) {
            hipLaunchKernelGGL((copy_kernel<true, 256>), dim3((unsigned)(c.n / 4 / 256)), dim3(256), 0, 0, a);
        });
        report("13-stream copy (nt)", c, ms);
    }

    const int set = argc > 5 ? atoi(argv[5]) : 0;
    if (set == 0) {
        run_variant<1, true, true, false, true, 256>(c, "step");
        run_variant<1, true, true, false, false, 256>(c, "step");
        run_variant<1, true, true, false, true, 512>(c, "step");
        run_variant<1, true, true, false, false, 512>(c, "step");
        run_variant<1, true, true, false, true, 1024>(c, "step");
        run_variant<2, true, true, false, true, 256>(c, "step");
        run_variant<2, true, true, false, true, 512>(c, "step");
        run_variant<1, true, true, true, true, 256>(c, "step");
        run_variant<1, true, true, true, true, 512>(c, "step", 4);
    } else if (set == 1) {
        run_variant<1, true, true, false, true, 256, -1>(c, "step");
        run_variant<1, true, true, false, true, 256, 0>(c, "step");
        run_variant<1, true, true, false, true, 256, 1>(c, "step");
        run_variant<1, true, true, false, true, 256, 2>(c, "step");
        run_variant<1, true, true, false, true, 256, 16>(c, "step");
        run_variant<1, true, true, false, true, 256, 17>(c, "step");
        run_variant<1, true, true, false, true, 256, 18>(c, "step");
        run_variant<1, true, true, false, true, 256, 19>(c, "step");
        run_variant<1, true, true, false, true, 512, 0>(c, "step");
        run_variant<1, true, true, false, true, 512, 2>(c, "step");
        run_variant<1, true, true, false, true, 512, 16>(c, "step");
        run_variant<1, true, true, false, true, 512, 17>(c, "step");
    }
    if (set == 3) {
        // stagger sweep on the stream alone (field fetch confined to 1 KiB) and with the fetch
        const size_t staggers[] = {0, 256, 512, 1024, 2048, 4096, 4352, 8192, 8448, 16640, 33024, 65792, 131328, 262400, 1048832, 69888, 135424};
        for (size_t sg : staggers) {
            for (int b = 0; b < 2; ++b)
                for (int k = 0; k < 6; ++k) st[b][k] = (float*)(slab + (size_t)(b * 6 + k) * arr_bytes + (size_t)(b * 6 + k) * sg);
            ty = (uint8_t*)(slab + 12 * arr_bytes + 12 * sg);
            for (int k = 0; k < 6; ++k) {
                c.s0.a[k] = st[0][k];
                c.fwd.src[k] = st[0][k]; c.fwd.dst[k] = st[1][k];
                c.bwd.src[k] = st[1][k]; c.bwd.dst[k] = st[0][k];
            }
            c.ty = ty; c.fwd.ty = ty; c.bwd.ty = ty;
            printf("stagger %zu\n", sg);
            run_variant<1, true, true, false, false, 256>(c, "  step");
            run_variant<1, true, true, false, true, 512>(c, "  step");
        }
    }
    if (set == 8) {
        // sorted particles (16x16-cell tiles), block 512: cache-policy bits of the particle-array loads
        const unsigned tile_cells = 16;
        auto run_spol = [&](auto kernel, const char* name, unsigned groups_per_lane = 1) {
            hipLaunchKernelGGL(generate_sorted_kernel, dim3(4096), dim3(256), 0, 0, c.s0, c.ty, c.n, side, tile_cells, r);
            CK(hipDeviceSynchronize());
            const unsigned grid = (unsigned)(c.n / 4ull / 512 / groups_per_lane);
            float total = 0;
            for (int k = 0; k < 8; ++k) {
                StreamArgs& a = (k & 1) ? c.bwd : c.fwd;
                CK(hipEventRecord(c.e0));
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), 0, 0, a, c.U);
                CK(hipEventRecord(c.e1));
                CK(hipEventSynchronize(c.e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, c.e0, c.e1));
                if (k >= 2) total += ms;
            }
            CK(hipGetLastError());
            printf("%-28s steps 2..7 after the sort: %8.4f ms average\n", name, total / 6);
            fflush(stdout);
        };
        if (c.n * 4ull >= (1ull << 32)) { printf("arrays too large for 32-bit buffer offsets\n"); return 1; }
        run_spol(variant_kernel<1, true, true, false, true, 512, -1, false, -1>, "nt global_load (library)");
        run_spol(variant_kernel<1, true, true, false, true, 512, -1, false, 0>, "buffer_load plain");
        run_spol(variant_kernel<1, true, true, false, true, 512, -1, false, 2>, "buffer_load nt");
        run_spol(variant_kernel<1, true, true, false, true, 512, -1, false, 1>, "buffer_load sc0");
        run_spol(variant_kernel<1, true, true, false, true, 512, -1, false, 3>, "buffer_load sc0 nt");
        run_spol(variant_kernel<1, true, true, false, true, 512, -1, false, 16>, "buffer_load sc1");
        run_spol(variant_kernel<1, true, true, false, true, 512, -1, false, 18>, "buffer_load sc1 nt");
        run_spol(variant_kernel<1, true, true, false, true, 512, -1, false, 17>, "buffer_load sc0 sc1");
        run_spol(variant_kernel<1, true, true, false, true, 512, -1, false, 19>, "buffer_load sc0 sc1 nt");
        run_spol(variant_kernel<2, true, true, false, true, 512, -1, false, 2>, "buffer_load nt, 8 particles/lane", 2);
    }
    if (set == 9) {
        // sorted particles (16x16-cell tiles): the library kernel (one tile per block) against
        // persistent blocks that keep the next tile's loads in flight
        const unsigned tile_cells = side >= 16 ? 16 : side;
        auto run9 = [&](auto launch, const char* name) {
            hipLaunchKernelGGL(generate_sorted_kernel, dim3(4096), dim3(256), 0, 0, c.s0, c.ty, c.n, side, tile_cells, r);
            CK(hipDeviceSynchronize());
            float total = 0;
            for (int k = 0; k < 8; ++k) {
                StreamArgs& a = (k & 1) ? c.bwd : c.fwd;
                CK(hipEventRecord(c.e0));
                launch(a);
                CK(hipEventRecord(c.e1));
                CK(hipEventSynchronize(c.e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, c.e0, c.e1));
                if (k >= 2) total += ms;
            }
            CK(hipGetLastError());
            printf("%-44s steps 2..7 after the sort: %8.4f ms average  %7.1f GB/s\n", name, total / 6, c.n * 49.0 / (total / 6) / 1e6);
            fflush(stdout);
        };
        TileMap T{};
        c.fwd.hist = c.bwd.hist = nullptr; c.fwd.hist_groups = c.bwd.hist_groups = 1; c.fwd.runs = c.bwd.runs = nullptr;
        run9([&](StreamArgs& a) {
            hipLaunchKernelGGL((step_stream_kernel<false, true, false, false>), dim3((unsigned)(c.n / 4 / kStreamThreads)), dim3(kStreamThreads), 0, 0, a, c.U, T);
        }, "library kernel, one tile per block");
        for (unsigned per_cu : {2u, 3u, 4u}) {
            char name[96];
            snprintf(name, sizeof name, "pipelined, block 512, %u blocks per CU", per_cu);
            run9([&](StreamArgs& a) { hipLaunchKernelGGL((pipelined_kernel<512>), dim3(256 * per_cu), dim3(512), 0, 0, a, c.U); }, name);
        }
        for (unsigned per_cu : {4u, 6u, 8u}) {
            char name[96];
            snprintf(name, sizeof name, "pipelined, block 256, %u blocks per CU", per_cu);
            run9([&](StreamArgs& a) { hipLaunchKernelGGL((pipelined_kernel<256>), dim3(256 * per_cu), dim3(256), 0, 0, a, c.U); }, name);
        }
        run9([&](StreamArgs& a) { hipLaunchKernelGGL((half_kernel<512>), dim3((unsigned)(c.n / 2 / 512)), dim3(512), 0, 0, a, c.U); }, "two particles per lane, block 512");
        run9([&](StreamArgs& a) { hipLaunchKernelGGL((half_kernel<1024>), dim3((unsigned)(c.n / 2 / 1024)), dim3(1024), 0, 0, a, c.U); }, "two particles per lane, block 1024");
        run9([&](StreamArgs& a) { hipLaunchKernelGGL((half_kernel<256>), dim3((unsigned)(c.n / 2 / 256)), dim3(256), 0, 0, a, c.U); }, "two particles per lane, block 256");
        for (unsigned per_cu : {8u, 12u, 16u}) {
            char name[96];
            snprintf(name, sizeof name, "pipelined, block 128, %u blocks per CU", per_cu);
            run9([&](StreamArgs& a) { hipLaunchKernelGGL((pipelined_kernel<128>), dim3(256 * per_cu), dim3(128), 0, 0, a, c.U); }, name);
        }
    }
    if (set == 10) {
        const unsigned tile_cells = side >= 16 ? 16 : side;
        auto run10 = [&](auto kernel, const char* name) {
            hipLaunchKernelGGL(generate_sorted_kernel, dim3(4096), dim3(256), 0, 0, c.s0, c.ty, c.n, side, tile_cells, r);
            CK(hipDeviceSynchronize());
            float total = 0;
            for (int k = 0; k < 8; ++k) {
                StreamArgs& a = (k & 1) ? c.bwd : c.fwd;
                CK(hipEventRecord(c.e0));
                hipLaunchKernelGGL(kernel, dim3((unsigned)(c.n / 2048)), dim3(512), 0, 0, a, c.U);
                CK(hipEventRecord(c.e1));
                CK(hipEventSynchronize(c.e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, c.e0, c.e1));
                if (k >= 2) total += ms;
            }
            CK(hipGetLastError());
            printf("%-36s steps 2..7 after the sort: %8.4f ms average  %7.1f GB/s\n", name, total / 6, c.n * 49.0 / (total / 6) / 1e6);
            fflush(stdout);
        };
        run10(policy_kernel<2, 2>, "loads nt, stores nt (library)");
        run10(policy_kernel<2, 0>, "loads nt, stores plain");
        run10(policy_kernel<2, 1>, "loads nt, stores sc0");
        run10(policy_kernel<2, 3>, "loads nt, stores sc0 nt");
        run10(policy_kernel<2, 16>, "loads nt, stores sc1");
        run10(policy_kernel<2, 18>, "loads nt, stores sc1 nt");
        run10(policy_kernel<2, 17>, "loads nt, stores sc0 sc1");
        run10(policy_kernel<2, 19>, "loads nt, stores sc0 sc1 nt");
        run10(policy_kernel<0, 2>, "loads plain, stores nt");
        run10(policy_kernel<2, 2>, "loads nt, stores nt (again)");
    }
    if (set == 11) {
        const unsigned tile_cells = side >= 16 ? 16 : side;
        TileMap T{};
        auto run11 = [&](auto kernel, const char* name) {
            hipLaunchKernelGGL(generate_sorted_kernel, dim3(4096), dim3(256), 0, 0, c.s0, c.ty, c.n, side, tile_cells, r);
            CK(hipDeviceSynchronize());
            float total = 0;
            for (int k = 0; k < 8; ++k) {
                StreamArgs& a = (k & 1) ? c.bwd : c.fwd;
                CK(hipEventRecord(c.e0));
                if constexpr (std::is_same_v<decltype(kernel), decltype(&step_stream_kernel<false, true, false, false>)>)
                    hipLaunchKernelGGL(kernel, dim3((unsigned)(c.n / kStreamTile)), dim3(kStreamThreads), 0, 0, a, c.U, T);
                else
                    hipLaunchKernelGGL(kernel, dim3((unsigned)(c.n / 2048)), dim3(512), 0, 0, a, c.U);
                CK(hipEventRecord(c.e1));
                CK(hipEventSynchronize(c.e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, c.e0, c.e1));
                if (k < 6) total += ms;
            }
            CK(hipGetLastError());
            printf("%-44s steps 0..5 after the sort: %8.4f ms average  %7.1f GB/s\n", name, total / 6, c.n * 49.0 / (total / 6) / 1e6);
            fflush(stdout);
        };
        c.fwd.hist = c.bwd.hist = nullptr; c.fwd.hist_groups = c.bwd.hist_groups = 1; c.fwd.runs = c.bwd.runs = nullptr;
        for (int rep = 0; rep < 2; ++rep) {
            run11(step_stream_kernel<false, true, false, false>, "library kernel");
            run11(order_kernel<2, 0, 0>, "2 positions back, then 4 fetches");
            run11(order_kernel<2, 0, 50>, "2 + sleep 3200 clk before the stores");
            run11(order_kernel<2, 0, 100>, "2 + sleep 6400 clk before the stores");
            run11(order_kernel<2, 0, 200>, "2 + sleep 12800 clk before the stores");
            run11(order_kernel<2, 0, 400>, "2 + sleep 25600 clk before the stores");
            run11(order_kernel<2, 0, 800>, "2 + sleep 51200 clk before the stores");
            run11(order_kernel<2, 100, 0>, "2 + sleep 6400 clk after the fetches");
            run11(order_kernel<2, 100, 100>, "2 + sleep 6400 clk after the fetches and before the stores");
        }
    }
    if (set == 2) {
        const unsigned tile_cells = argc > 6 ? atoi(argv[6]) : 16;
        for (int variant = 0; variant < 3; ++variant) {
            hipLaunchKernelGGL(generate_sorted_kernel, dim3(4096), dim3(256), 0, 0, c.s0, c.ty, c.n, side, tile_cells, r);
            CK(hipDeviceSynchronize());
            const int block = variant == 0 ? 512 : (variant == 1 ? 512 : 1024);
            const unsigned grid = (unsigned)(c.n / 4ull / block);
            printf("sorted by %ux%u-cell tiles, block %d xcd-remap %d; per-step time as the order decays\n", tile_cells, tile_cells, block, variant >= 1);
            for (int k = 0; k < 17; ++k) {
                StreamArgs& a = (k & 1) ? c.bwd : c.fwd;
                CK(hipEventRecord(c.e0));
                if (variant == 0) hipLaunchKernelGGL((variant_kernel<1, true, true, false, true, 512, -1, false>), dim3(grid), dim3(512), 0, 0, a, c.U);
                else if (variant == 1) hipLaunchKernelGGL((variant_kernel<1, true, true, false, true, 512, -1, true>), dim3(grid), dim3(512), 0, 0, a, c.U);
                else hipLaunchKernelGGL((variant_kernel<1, true, true, false, true, 1024, -1, true>), dim3(grid), dim3(1024), 0, 0, a, c.U);
                CK(hipEventRecord(c.e1));
                CK(hipEventSynchronize(c.e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, c.e0, c.e1));
                if (k < 5 || k == 8 || k == 16) printf("step %3d: %8.4f ms  %8.1f GB/s %7.2f Gp/s\n", k, ms, c.n * 49.0 / ms / 1e6, c.n / ms / 1e6);
            }
        }
    }
    if (set == 4) {
        char* A = slab;
        char* B = slab + (size_t)6 * arr_bytes;   // second half of the slab
        for (int rep = 0; rep < 2; ++rep) {
            float ms = time_ms(c, [&](StreamArgs& a) {
                const bool f = &a == &c.fwd;
                hipLaunchKernelGGL((copy_chunked_kernel<true, 2048>), dim3((unsigned)(c.n / 2048)), dim3(512), 0, 0, f ? A : B, f ? B : A);
            });
            report("chunked copy CH=2048 (nt)", c, ms);
            ms = time_ms(c, [&](StreamArgs& a) {
                const bool f = &a == &c.fwd;
                hipLaunchKernelGGL((copy_chunked_kernel<true, 1024>), dim3((unsigned)(c.n / 1024)), dim3(256), 0, 0, f ? A : B, f ? B : A);
            });
            report("chunked copy CH=1024 (nt)", c, ms);
            ms = time_ms(c, [&](StreamArgs& a) {
                const bool f = &a == &c.fwd;
                hipLaunchKernelGGL((copy_chunked_kernel<false, 2048>), dim3((unsigned)(c.n / 2048)), dim3(512), 0, 0, f ? A : B, f ? B : A);
            });
            report("chunked copy CH=2048 (plain)", c, ms);
            ms = time_ms(c, [&](StreamArgs& a) {
                hipLaunchKernelGGL((copy_kernel<true, 512>), dim3((unsigned)(c.n / 4 / 512)), dim3(512), 0, 0, a);
            });
            report("13-stream copy block 512 (nt)", c, ms);
        }
    }
    if (set == 5) {
        // one contiguous float4 copy at growing footprints (first half of the slab -> second half)
        const f32x4* A = (const f32x4*)slab;
        f32x4* B = (f32x4*)(slab + (size_t)6 * arr_bytes);
        for (size_t mib = 256; mib <= 6144 && mib * (1u << 20) <= (size_t)6 * arr_bytes; mib += (mib < 1024 ? 256 : 1024)) {
            const unsigned long long n16 = mib * (1ull << 20) / 16;
            float ms = time_ms(c, [&](StreamArgs&) {
                hipLaunchKernelGGL(copy1_kernel<true>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, A, B, n16);
            });
            printf("contiguous nt copy %5zu MiB -> %5zu MiB: %8.4f ms %8.1f GB/s\n", mib, mib, ms, 2.0 * mib * 1048576.0 / ms / 1e6);
        }
    }
    if (set == 7) {
#define RUNK(K, BLOCK)                                                                                            \
        {                                                                                                         \
            const unsigned grid = (unsigned)((c.n / 4 + BLOCK - 1) / BLOCK);                                      \
            float ms = time_ms(c, [&](StreamArgs& a) {                                                            \
                hipLaunchKernelGGL((copyk_kernel<K, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, a);                   \
            });                                                                                                   \
            printf("copy of %d arrays, block %4d: %8.4f ms %8.1f GB/s\n", K, BLOCK, ms, c.n * 8.0 * K / ms / 1e6); \
            fflush(stdout);                                                                                       \
        }
        RUNK(1, 64) RUNK(6, 64) RUNK(1, 128) RUNK(6, 128)
        RUNK(1, 256) RUNK(2, 256) RUNK(3, 256) RUNK(4, 256) RUNK(6, 256)
        RUNK(1, 512) RUNK(2, 512) RUNK(3, 512) RUNK(6, 512)
        RUNK(1, 1024) RUNK(3, 1024) RUNK(6, 1024)
    }
    return 0;
}
