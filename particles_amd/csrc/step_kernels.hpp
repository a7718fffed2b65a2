// step_kernels.hpp — gfx950 device code of the particle update.
//
// The arithmetic follows src/compute.wgsl of the reference statement by
// statement (citations below are relative to the reference checkout) and uses
// only IEEE-754 correctly rounded fp32 operations (+, -, *, /, sqrt, compares,
// float->u32 truncation) with FMA contraction disabled, so that on finite
// inputs the streaming path is bit-for-bit reproducible on any IEEE machine.
// exp(-delta_t) is the same for every particle and is evaluated once on the
// host.  Layout: structure-of-arrays fp32 state (x,y,z,vx,vy,vz), one byte of
// type per particle, field cells as float4 (only xyz is read).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace pcl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlockThreads = 256;     // 4 waves of 64
constexpr int kParticlesPerLane = 4;   // one 16-byte access per array per lane
constexpr int kTileParticles = kBlockThreads * kParticlesPerLane;

// Bits of a handle's device error word: conditions a kernel can only detect
// while it runs; the next call that synchronises the stream reports them.
constexpr uint32_t kErrScatterOverflow = 1u;   // a tile reserved more slots of a bin than the histogram counted
constexpr uint32_t kErrBrushRecord = 2u;       // a cursor record read from device memory has an unknown mode or falloff

// Everything that is uniform over a launch; passed by value (lives in SGPRs).
struct StepUniforms {
    float decay;         // exp(-delta_t), evaluated on the host       wgsl:151
    float dt;            // delta_t
    float max_velocity;  //                                            wgsl:159
    float radius;        // bounding_volume_radius                     wgsl:170
    float two_radius;    // 2.0 * radius                               wgsl:60,63
    float slh;           // (radius * 2.0) * 0.5                       wgsl:83,147
    float neg_slh;       // -slh
    float span_a;        // slh - (-slh)                               wgsl:75
    float span_b[3];     // f32(dim) - 0.0                             wgsl:77
    uint32_t dim_y;
    uint32_t dim_z;
    uint32_t last_cell;  // n_cells - 1: index clamp for the fetch
    uint32_t degenerate; // start_a == end_a                           wgsl:69
    float mass[5];       // particle_type_masses[k].mass               wgsl:109
    // Operands of the refined-reciprocal divisions (integrate4): correctly rounded reciprocals,
    // evaluated on the host, of the launch-uniform divisors; fast_div says that all of them lie in
    // the exponent window in which those divisions are exact (uniforms_of in particles_hip.hip).
    float rcp_span_a;
    float rcp_mass[5];
    uint32_t fast_div;
};

struct DiagPartial {
    double kinetic_energy;
    unsigned long long count;
    float pos_max[3];
    float neg_pos_min[3];
    float pad[2];
};

struct StateSoA {
    float* a[6];  // x, y, z, vx, vy, vz
};

// ---------------------------------------------------------------- helpers

// wgsl:68-79 with start_b = 0; the uniform operands are precomputed.
__device__ __forceinline__ float map_range_axis(const StepUniforms& U, float val, float span_b)
{
    const float valc = fminf(fmaxf(val, U.neg_slh), U.slh);
    const float oz = __fdiv_rn(valc - U.neg_slh, U.span_a);
    return oz * span_b + 0.0f;
}

// wgsl:81-89 and the fetch of wgsl:147-148; an index past the end is clamped
// to the last cell.
__device__ __forceinline__ uint32_t cell_index(const StepUniforms& U, float x, float y, float z)
{
    uint32_t ixx = 0u, ixy = 0u, ixz = 0u;
    if (!U.degenerate) {
        ixx = (uint32_t)map_range_axis(U, x, U.span_b[0]);
        ixy = (uint32_t)map_range_axis(U, y, U.span_b[1]);
        ixz = (uint32_t)map_range_axis(U, z, U.span_b[2]);
    }
    const uint32_t fgi = ixx * U.dim_y * U.dim_z + ixy * U.dim_z + ixz;
    return fgi < U.last_cell ? fgi : U.last_cell;
}

// wgsl:58-66
__device__ __forceinline__ float wrap_symmetrically(float val, float max, float two_max)
{
    if (val > max) return val - two_max;
    if (val < -max) return val + two_max;
    return val;
}

__device__ __forceinline__ float mass_of(const StepUniforms& U, uint32_t ty)
{
    float m = U.mass[4];
    m = ty == 3u ? U.mass[3] : m;
    m = ty == 2u ? U.mass[2] : m;
    m = ty == 1u ? U.mass[1] : m;
    m = ty == 0u ? U.mass[0] : m;
    return m;
}

// wgsl:148-172 for one particle whose pair acceleration is (ax, ay, az) and
// whose field cell value is f.
__device__ __forceinline__ void integrate(const StepUniforms& U, float fx, float fy, float fz,
                                          float m, float ax, float ay, float az,
                                          float& x, float& y, float& z,
                                          float& vx, float& vy, float& vz)
{
    ax = ax + __fdiv_rn(10.0f * fx, m);
    ay = ay + __fdiv_rn(10.0f * fy, m);
    az = az + __fdiv_rn(10.0f * fz, m);

    vx = vx * U.decay;
    vy = vy * U.decay;
    vz = vz * U.decay;
    vx = vx + ax * U.dt;
    vy = vy + ay * U.dt;
    vz = vz + az * U.dt;

    const float vel = sqrtf((vx * vx + vy * vy) + vz * vz);
    if (vel > 0.001f) {
        const float capped = fminf(fmaxf(vel, 0.0f), U.max_velocity);
        vx = __fdiv_rn(vx, vel) * capped;
        vy = __fdiv_rn(vy, vel) * capped;
        vz = __fdiv_rn(vz, vel) * capped;
    }

    x = x + vx * U.dt;
    y = y + vy * U.dt;
    z = z + vz * U.dt;

    x = wrap_symmetrically(x, U.radius, U.two_radius);
    y = wrap_symmetrically(y, U.radius, U.two_radius);
    z = wrap_symmetrically(z, U.radius, U.two_radius);
}

// ------------------------------------------------ packed fast path (4 particles per lane)
//
// The same update, the same correctly rounded results, in about two thirds of the vector
// instructions:
//  * two particles per instruction through the packed fp32 operations of CDNA (v_pk_mul_f32,
//    v_pk_add_f32, v_pk_fma_f32): a lane holds 4 consecutive particles per array, i.e. two
//    register pairs;
//  * a / b for the six divisions per particle whose divisor is uniform over the launch (the span
//    of the bounding volume, wgsl:75; the five masses, wgsl:148) as the refinement steps of the
//    hardware's own division sequence
//        q0 = a * y,  r0 = fma(-b, q0, a),  q1 = fma(r0, y, q0),  r1 = fma(-b, q1, a),  q = fma(r1, y, q1)
//    on the divisor's correctly rounded reciprocal y, which the host evaluates once.  What the full
//    sequence adds, scaling of operands with extreme exponents and the fix-up of zeros, infinities
//    and NaNs, is replaced by preconditions under which no intermediate overflows, underflows or
//    loses a bit: divisors within [2^-40, 2^40] in magnitude (checked on the host: fast_div), field
//    components zero or within [2^-60, 2^60] (checked by the kernels that write the field: a device
//    flag, cell_in_window); a launch whose operands are outside takes the scalar path, kernel by
//    kernel.  tools/debug/divcheck.c holds the host-side check of the sequence (every significand
//    against 48 divisors, 2 * 10^9 random pairs inside the window: no mismatch);
//  * sqrt as v_sqrt_f32 and the hardware lowering's one-ulp correction without its scaling of
//    arguments below 2^-96: the root of such an argument is far below the 0.001 threshold of
//    wgsl:157, above which alone the value of the root is used.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

__device__ __forceinline__ f32x2 div_refined(f32x2 a, f32x2 neg_b, f32x2 y)
{
    f32x2 q = a * y;
    f32x2 r = fma2(neg_b, q, a);
    q = fma2(r, y, q);
    r = fma2(neg_b, q, a);
    return fma2(r, y, q);
}

// Correctly rounded square root for arguments that are +0, +inf or at least 2^-96; anything
// below 2^-96 yields a value below 2^-47.
__device__ __forceinline__ float sqrt_no_scaling(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    float r = r_dn <= 0.0f ? s_dn : s;
    r = r_up > 0.0f ? s_up : r;
    return r;
}

// wgsl:81-89 for two particles; not for the degenerate range (start_a == end_a).
__device__ __forceinline__ void cell_index2(const StepUniforms& U, f32x2 x, f32x2 y, f32x2 z, uint32_t out[2])
{
    const f32x2 neg_span = splat2(-U.span_a), rcp = splat2(U.rcp_span_a), lo = splat2(U.neg_slh);
    auto axis = [&](f32x2 v, float span_b) -> f32x2 {
        f32x2 c;
        c.x = fminf(fmaxf(v.x, U.neg_slh), U.slh);
        c.y = fminf(fmaxf(v.y, U.neg_slh), U.slh);
        // oz * span_b (the shader's "+ start_b" adds +0 to a value that is never negative)
        return div_refined(c - lo, neg_span, rcp) * splat2(span_b);
    };
    const f32x2 fx = axis(x, U.span_b[0]), fy = axis(y, U.span_b[1]), fz = axis(z, U.span_b[2]);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t fgi = (uint32_t)fx[k] * U.dim_y * U.dim_z + (uint32_t)fy[k] * U.dim_z + (uint32_t)fz[k];
        out[k] = fgi < U.last_cell ? fgi : U.last_cell;
    }
}

// wgsl:148-172 for two particles (fx, fy, fz: their field cells; m, rm: their masses and the
// masses' reciprocals); pair acceleration zero.
__device__ __forceinline__ void integrate2(const StepUniforms& U, f32x2 fx, f32x2 fy, f32x2 fz, f32x2 m, f32x2 rm,
                                           f32x2& x, f32x2& y, f32x2& z, f32x2& vx, f32x2& vy, f32x2& vz)
{
    const f32x2 ten = splat2(10.0f), zero = splat2(0.0f), neg_m = -m;
    const f32x2 ax = zero + div_refined(ten * fx, neg_m, rm);
    const f32x2 ay = zero + div_refined(ten * fy, neg_m, rm);
    const f32x2 az = zero + div_refined(ten * fz, neg_m, rm);
    const f32x2 decay = splat2(U.decay), dt = splat2(U.dt);
    f32x2 wx = vx * decay + ax * dt;
    f32x2 wy = vy * decay + ay * dt;
    f32x2 wz = vz * decay + az * dt;

    const f32x2 s = (wx * wx + wy * wy) + wz * wz;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float vel = sqrt_no_scaling(s[k]);
        const float capped = fminf(fmaxf(vel, 0.0f), U.max_velocity);
        const bool moving = vel > 0.001f;
        // The three divisions by the particle's own speed keep the full sequence: nothing bounds their
        // operands in advance (a refined variant with a per-particle window test and fall-back measured
        // slower: the test and the branch cost more than the seven instructions per division they save).
        const float nx = __fdiv_rn(wx[k], vel) * capped;
        const float ny = __fdiv_rn(wy[k], vel) * capped;
        const float nz = __fdiv_rn(wz[k], vel) * capped;
        wx[k] = moving ? nx : wx[k];
        wy[k] = moving ? ny : wy[k];
        wz[k] = moving ? nz : wz[k];
    }
    vx = wx; vy = wy; vz = wz;
    x = x + vx * dt;
    y = y + vy * dt;
    z = z + vz * dt;
    const f32x2 two_r = splat2(U.two_radius);
    auto wrap = [&](f32x2& p) {
        const f32x2 down = p - two_r, up = p + two_r;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float low = p[k] < -U.radius ? up[k] : p[k];
            p[k] = p[k] > U.radius ? down[k] : low;
        }
    };
    wrap(x); wrap(y); wrap(z);
}

// Masses and their reciprocals by particle type, one private copy per wave in LDS (a table
// look-up instead of a chain of selects per particle); types above 4 read the last mass, as
// mass_of does.  A wave's own LDS writes and reads execute in order, so no barrier is needed.
struct MassTable {
    float2* tab;   // this wave's 8 entries
    __device__ __forceinline__ void init(float2* wave_tab, const StepUniforms& U)
    {
        tab = wave_tab;
        // one lane writes all eight entries from the scalar registers the masses arrive in (an index
        // computed per lane would turn into a load from the kernel-argument block)
        if ((threadIdx.x & 63u) == 0u) {
            wave_tab[0] = make_float2(U.mass[0], U.rcp_mass[0]);
            wave_tab[1] = make_float2(U.mass[1], U.rcp_mass[1]);
            wave_tab[2] = make_float2(U.mass[2], U.rcp_mass[2]);
            wave_tab[3] = make_float2(U.mass[3], U.rcp_mass[3]);
            const float2 last = make_float2(U.mass[4], U.rcp_mass[4]);
            wave_tab[4] = last; wave_tab[5] = last; wave_tab[6] = last; wave_tab[7] = last;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
    __device__ __forceinline__ float2 of(uint32_t ty) const { return tab[ty < 7u ? ty : 7u]; }
};

// One update of a lane's 4 consecutive particles on the packed path (cell_index2, integrate2);
// `fetch(cell)` returns the field value, m[] receives the particles' masses.
template <typename Fetch>
__device__ __forceinline__ void step4_packed(const StepUniforms& U, const MassTable& M, uint32_t t4, Fetch fetch,
                                             f32x4& X, f32x4& Y, f32x4& Z, f32x4& VX, f32x4& VY, f32x4& VZ, float m[4])
{
    uint32_t ci[4];
    cell_index2(U, X.xy, Y.xy, Z.xy, ci);
    cell_index2(U, X.zw, Y.zw, Z.zw, ci + 2);
    f32x4 F[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) F[k] = fetch(ci[k]);
    // All four fetches leave before any arithmetic of the update: left alone, the scheduler moves the second
    // pair's cell indices and fetches behind the first pair's update to save registers, and the lane pays
    // two fetch round trips in a row (measured: 2.26 against 2.20 ms for 2^28 particles).
    __builtin_amdgcn_sched_barrier(0);
    float2 mt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mt[k] = M.of((t4 >> (8 * k)) & 0xFFu); m[k] = mt[k].x; }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int a = 2 * h, b = 2 * h + 1;
        f32x2 x = {X[a], X[b]}, y = {Y[a], Y[b]}, z = {Z[a], Z[b]};
        f32x2 vx = {VX[a], VX[b]}, vy = {VY[a], VY[b]}, vz = {VZ[a], VZ[b]};
        integrate2(U, f32x2{F[a].x, F[b].x}, f32x2{F[a].y, F[b].y}, f32x2{F[a].z, F[b].z},
                   f32x2{mt[a].x, mt[b].x}, f32x2{mt[a].y, mt[b].y}, x, y, z, vx, vy, vz);
        X[a] = x.x; X[b] = x.y; Y[a] = y.x; Y[b] = y.y; Z[a] = z.x; Z[b] = z.y;
        VX[a] = vx.x; VX[b] = vx.y; VY[a] = vy.x; VY[b] = vy.y; VZ[a] = vz.x; VZ[b] = vz.y;
    }
}

// The scalar path for a lane's four particles (full divisions, the library square root).
template <typename Fetch>
__device__ __forceinline__ void step4_scalar(const StepUniforms& U, uint32_t t4, Fetch fetch,
                                             f32x4& X, f32x4& Y, f32x4& Z, f32x4& VX, f32x4& VY, f32x4& VZ, float m[4])
{
    f32x4 F[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) F[k] = fetch(cell_index(U, X[k], Y[k], Z[k]));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[k] = mass_of(U, (t4 >> (8 * k)) & 0xFFu);
        float x = X[k], y = Y[k], z = Z[k], vx = VX[k], vy = VY[k], vz = VZ[k];
        integrate(U, F[k].x, F[k].y, F[k].z, m[k], 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
        X[k] = x; Y[k] = y; Z[k] = z; VX[k] = vx; VY[k] = vy; VZ[k] = vz;
    }
}

// Barrier for hand-offs that go through LDS only.  __syncthreads() also waits
// for the wave's outstanding global loads and stores (vmcnt(0)); where the
// hand-off goes through LDS only that wait is wasted: a block would drain its
// stores before it may reduce its diagnostics or stage the next array of a
// scatter.  Here only LDS traffic is ordered.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ------------------------------------------------------- diagnostics pieces

struct DiagAcc {
    double ke;
    unsigned long long count;
    float mx[3];
    float nmn[3];
    __device__ __forceinline__ void init()
    {
        ke = 0.0;
        count = 0ull;
        for (int c = 0; c < 3; ++c) { mx[c] = -INFINITY; nmn[c] = -INFINITY; }
    }
    __device__ __forceinline__ void add(float m, float x, float y, float z,
                                        float vx, float vy, float vz)
    {
        const double dx = vx, dy = vy, dz = vz;
        ke += 0.5 * (double)m * ((dx * dx + dy * dy) + dz * dz);
        count += 1ull;
        mx[0] = fmaxf(mx[0], x);  mx[1] = fmaxf(mx[1], y);  mx[2] = fmaxf(mx[2], z);
        nmn[0] = fmaxf(nmn[0], -x); nmn[1] = fmaxf(nmn[1], -y); nmn[2] = fmaxf(nmn[2], -z);
    }
};

// Wave-wide reductions on the data-parallel-primitive path of the VALU (no LDS traffic): an inclusive
// scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the row totals handed on with row_bcast:15
// (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3); lane 63 ends up with the reduction over the
// wave.  Lanes without a source lane take `identity`.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_take(uint32_t identity, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void diag_dpp_step(DiagAcc& acc)
{
    const uint32_t ninf = 0xFF800000u;   // -inf
    const unsigned long long ke_bits = __builtin_bit_cast(unsigned long long, acc.ke);
    const uint32_t ke_lo = dpp_take<CTRL, ROW_MASK>(0u, (uint32_t)ke_bits), ke_hi = dpp_take<CTRL, ROW_MASK>(0u, (uint32_t)(ke_bits >> 32));
    acc.ke += __builtin_bit_cast(double, ((unsigned long long)ke_hi << 32) | ke_lo);
    const uint32_t c_lo = dpp_take<CTRL, ROW_MASK>(0u, (uint32_t)acc.count), c_hi = dpp_take<CTRL, ROW_MASK>(0u, (uint32_t)(acc.count >> 32));
    acc.count += ((unsigned long long)c_hi << 32) | c_lo;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        acc.mx[c] = fmaxf(acc.mx[c], __uint_as_float(dpp_take<CTRL, ROW_MASK>(ninf, __float_as_uint(acc.mx[c]))));
        acc.nmn[c] = fmaxf(acc.nmn[c], __uint_as_float(dpp_take<CTRL, ROW_MASK>(ninf, __float_as_uint(acc.nmn[c]))));
    }
}

// Reduce one accumulator per thread to one DiagPartial per block, in a fixed
// order (the wave's scan above, then the waves in sequence): deterministic for a given
// launch geometry.
template <int THREADS = kBlockThreads>
__device__ __forceinline__ void block_reduce_store(DiagAcc acc, DiagPartial* out)
{
    __shared__ DiagPartial wave_part[THREADS / 64];
    diag_dpp_step<0x111, 0xf>(acc);   // row_shr:1
    diag_dpp_step<0x112, 0xf>(acc);   // row_shr:2
    diag_dpp_step<0x114, 0xf>(acc);   // row_shr:4
    diag_dpp_step<0x118, 0xf>(acc);   // row_shr:8
    diag_dpp_step<0x142, 0xa>(acc);   // row_bcast:15 into rows 1 and 3
    diag_dpp_step<0x143, 0xc>(acc);   // row_bcast:31 into rows 2 and 3
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) {
        DiagPartial p;
        p.kinetic_energy = acc.ke;
        p.count = acc.count;
        for (int c = 0; c < 3; ++c) { p.pos_max[c] = acc.mx[c]; p.neg_pos_min[c] = acc.nmn[c]; }
        p.pad[0] = p.pad[1] = 0.0f;
        wave_part[wave] = p;
    }
    lds_barrier();
    if (threadIdx.x == 0) {
        DiagPartial p = wave_part[0];
        for (int w = 1; w < THREADS / 64; ++w) {
            p.kinetic_energy += wave_part[w].kinetic_energy;
            p.count += wave_part[w].count;
            for (int c = 0; c < 3; ++c) {
                p.pos_max[c] = fmaxf(p.pos_max[c], wave_part[w].pos_max[c]);
                p.neg_pos_min[c] = fmaxf(p.neg_pos_min[c], wave_part[w].neg_pos_min[c]);
            }
        }
        *out = p;
    }
}

// Diagnostics record as the C ABI lays it out (pcl_diagnostics, 48 bytes).
struct DiagOut {
    double count;
    double kinetic_energy;
    float pos_max[3];
    float neg_pos_min[3];
    float pad[2];
};

// Fold block partials in a fixed order: block b of the grid folds the
// contiguous chunk [b*chunk, (b+1)*chunk) into out_partials[b]; the last level
// (one block) also writes the pcl_diagnostics record.
__global__ __launch_bounds__(kBlockThreads) void diag_fold_kernel(
    const DiagPartial* __restrict__ partials, uint32_t n_partials, uint32_t chunk,
    DiagPartial* __restrict__ out_partials, DiagOut* __restrict__ out)
{
    __shared__ DiagPartial folded;
    DiagAcc acc;
    acc.init();
    const uint32_t begin = blockIdx.x * chunk;
    const uint32_t end = begin + chunk < n_partials ? begin + chunk : n_partials;
    for (uint32_t i = begin + threadIdx.x; i < end; i += kBlockThreads) {
        const DiagPartial p = partials[i];
        acc.ke += p.kinetic_energy;
        acc.count += p.count;
        for (int c = 0; c < 3; ++c) {
            acc.mx[c] = fmaxf(acc.mx[c], p.pos_max[c]);
            acc.nmn[c] = fmaxf(acc.nmn[c], p.neg_pos_min[c]);
        }
    }
    block_reduce_store(acc, &folded);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (out_partials) out_partials[blockIdx.x] = folded;
        if (out) {
            DiagOut o;
            o.count = (double)folded.count;
            o.kinetic_energy = folded.kinetic_energy;
            for (int c = 0; c < 3; ++c) { o.pos_max[c] = folded.pos_max[c]; o.neg_pos_min[c] = folded.neg_pos_min[c]; }
            o.pad[0] = o.pad[1] = 0.0f;
            *out = o;
        }
    }
}

// ------------------------------------------------------ streaming kernel K1

// Field tiles of (1<<sx) x (1<<sy) x (1<<sz) cells, enumerated x-tile slowest
// like the cells themselves; the sort key of the field-tile particle order
// (reorder_kernels.hpp).
struct TileMap {
    uint32_t sx, sy, sz;     // log2 of the tile extent per axis
    uint32_t nty, ntz;       // tiles along y and z
    uint32_t n_bins;
    uint32_t max_ix[3];      // dim - 1 per axis (an index equal to dim is clamped)
    float lookahead;         // seconds: the key is the tile of pos + vel * lookahead
    float inv_cell[3];       // cells per unit length per axis (0 when the range is degenerate)
    uint32_t reach;          // pair cell maps: neighbour cells to sweep each way (1 or 2); 0 otherwise
    float cell_len;          // pair cell maps: cell edge length
};

// Sort key of a particle: the field tile of where it will be `lookahead`
// seconds from now (half a refresh interval), so that the order is at its
// best in the middle of the interval instead of at its start.  Only a
// locality heuristic — any key gives the same per-particle results — so it
// uses a reciprocal multiply (inv_cell = dim / (2 slh)) instead of the exact
// division of the field fetch; what matters is that every kernel computes it
// with this one function.
__device__ __forceinline__ uint32_t tile_key(const StepUniforms& U, const TileMap& T,
                                             float x, float y, float z, float vx, float vy, float vz)
{
    const float px = fminf(fmaxf(x + vx * T.lookahead, U.neg_slh), U.slh);
    const float py = fminf(fmaxf(y + vy * T.lookahead, U.neg_slh), U.slh);
    const float pz = fminf(fmaxf(z + vz * T.lookahead, U.neg_slh), U.slh);
    uint32_t ix = (uint32_t)((px - U.neg_slh) * T.inv_cell[0]);
    uint32_t iy = (uint32_t)((py - U.neg_slh) * T.inv_cell[1]);
    uint32_t iz = (uint32_t)((pz - U.neg_slh) * T.inv_cell[2]);
    ix = ix < T.max_ix[0] ? ix : T.max_ix[0];
    iy = iy < T.max_ix[1] ? iy : T.max_ix[1];
    iz = iz < T.max_ix[2] ? iz : T.max_ix[2];
    return ((ix >> T.sx) * T.nty + (iy >> T.sy)) * T.ntz + (iz >> T.sz);
}

// counters[key] += 1, returning this particle's rank within its counter.
// Plain LDS atomics: combining equal keys across the wave first was measured
// and bought nothing (the LDS serves same-address atomics quickly) while
// costing ~100 VALU instructions per call.
__device__ __forceinline__ uint32_t lds_rank_add(uint32_t* counters, uint32_t key, bool active)
{
    return active ? atomicAdd(&counters[key], 1u) : 0u;
}

// The streaming kernels fetch the field from a second copy packed to 12 bytes per cell (x, y, z;
// the shader reads only .xyz, wgsl:148) when the host supplies one: a quarter fewer cache lines
// under the same particles (measured on 2^28 particles, 1024^2 field: plain step 2.351 against
// 2.368 ms).  Null: the 16-byte cells.
struct Cell3 {
    float x, y, z;
};

// One launch-uniform choice of copy, made without a branch (a branch per fetch would end a basic
// block at every fetch and keep the four fetches of a lane from being in flight together): both
// copies are read as 12 bytes at cell * stride.
__device__ __forceinline__ f32x4 fetch_cell(const f32x4* __restrict__ field, const float* __restrict__ field3, uint32_t c)
{
    const char* base = field3 ? reinterpret_cast<const char*>(field3) : reinterpret_cast<const char*>(field);
    const uint32_t stride = field3 ? 12u : 16u;
    const Cell3 v = *reinterpret_cast<const Cell3*>(base + (unsigned long long)c * stride);
    return f32x4{v.x, v.y, v.z, 1.0f};
}

struct StreamArgs {
    const float* src[6];   // dst may equal src (in-place update): no __restrict__ on the state arrays
    float* dst[6];
    const uint8_t* __restrict__ ty;
    const f32x4* __restrict__ field;
    unsigned long long n;
    DiagPartial* partials;  // one per block when DIAG
    uint32_t lds_cells;     // cells staged in LDS when FIELD_LDS
    uint32_t* hist;         // global tile histogram when COUNT: counter of bin b for tile t is hist[(t % hist_groups) * n_bins + b]
    uint32_t hist_groups;
    uint32_t* runs;         // when COUNT: total over blocks of the bins each block touched (may be null)
    const float* __restrict__ field3;   // the field packed to 12 bytes per cell, or null (fetch_cell)
    const uint32_t* __restrict__ field_flag;   // non-zero: some field component is outside the packed path's window
    uint32_t xcd_chunk;     // k, a power of two, or 0: block b takes tile (b with bits [0,3) and [3,3+log2 k) swapped): see stream_tiles
};

// Whether this launch may take the packed path: the host's verdict on the uniform divisors and
// the device flag that the kernels writing the field keep.  Uniform over the launch.
__device__ __forceinline__ bool packed_path_ok(const StepUniforms& U, const uint32_t* field_flag)
{
#ifdef PCL_FORCE_PACKED   // analysis builds: the packed path alone
    return true;
#endif
    return U.fast_div != 0u && U.degenerate == 0u && field_flag != nullptr && *field_flag == 0u;
}

template <bool NT>
__device__ __forceinline__ f32x4 load4(const float* base, unsigned long long group)
{
    const f32x4* p = reinterpret_cast<const f32x4*>(base) + group;
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

template <bool NT>
__device__ __forceinline__ void store4(float* base, unsigned long long group, f32x4 v)
{
    f32x4* p = reinterpret_cast<f32x4*>(base) + group;
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// Non-temporal access to one block's tile of a particle array through a buffer
// resource whose base is the tile (wave-uniform, in SGPRs) and whose extent is
// the tile's valid bytes: the lane supplies a 32-bit offset only, so no 64-bit
// address arithmetic runs per lane, and the array may be of any size.
// (0x00020000: raw buffer, 32-bit data format; aux 2 = nt.)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* tile_base, uint32_t valid_bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(tile_base), 0, valid_bytes, 0x00020000);
}

__device__ __forceinline__ f32x4 tile_load4(__amdgpu_buffer_rsrc_t r, uint32_t lane_byte)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_byte, 0, 2));
}

__device__ __forceinline__ void tile_store4(__amdgpu_buffer_rsrc_t r, uint32_t lane_byte, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)lane_byte, 0, 2);
}

// Pair term provably zero (every polynomial coefficient +-0, or no distance
// can pass both cut-offs): wgsl:115-144 contributes exactly +0, so the update
// is a pure stream: 24 B state + 1 B type in, 24 B out per particle, plus one
// cache-resident 16-byte field fetch.  A block owns tiles of kStreamThreads*4
// consecutive particles (one 16-byte access per array per lane); the grid is
// one tile per block, or a smaller persistent grid striding over the tiles
// when the field is staged in LDS.
#ifndef PCL_STREAM_THREADS
#define PCL_STREAM_THREADS 512   // 8 waves
#endif
constexpr int kStreamThreads = PCL_STREAM_THREADS;
constexpr int kStreamTile = kStreamThreads * kParticlesPerLane;
constexpr uint32_t kRunSlots = 256;   // counters of StreamArgs::runs

// The tiles of one block (and, in block 0, the up to 3 particles past the last full group), either
// on the packed path (PACKED; see integrate2) or on the scalar one.  The two are separate
// instantiations so that neither's registers weigh on the other; the kernel picks one per launch.
template <bool DIAG, bool NT, bool FIELD_LDS, bool COUNT, int BLOCK, bool PACKED>
__device__ __forceinline__ void stream_tiles(const StreamArgs& a, const StepUniforms& U, const TileMap& T,
                                             const f32x4* lds_field, uint32_t* lds_hist, float2* wave_mass_tab, DiagAcc& acc)
{
    auto fetch = [&](uint32_t c) -> f32x4 { return FIELD_LDS ? lds_field[c] : fetch_cell(a.field, a.field3, c); };
    MassTable M;
    if (PACKED) M.init(wave_mass_tab, U);

    const unsigned long long n_groups = a.n / 4ull;
    const unsigned long long n_tiles = (n_groups + BLOCK - 1) / BLOCK;
    // Blocks are dealt round-robin over the 8 XCDs (b and b + 8 share one, with its L2).  With xcd_chunk = k
    // set, every run of 8k consecutive tiles is dealt so that each XCD takes k CONSECUTIVE tiles of it: the
    // blocks that run together still work on one stretch of the arrays, and the field cells of a stretch of
    // k tiles are fetched into one L2 instead of all eight.  A matter of speed only: any block may take any
    // tile (the grid is a multiple of 8k blocks; those past the last tile have nothing to do).
    unsigned long long first = blockIdx.x;
    if (a.xcd_chunk) {
        const uint32_t k = a.xcd_chunk, b = blockIdx.x;
        first = (unsigned long long)(b & ~(8u * k - 1u)) + (b & 7u) * k + ((b >> 3) & (k - 1u));
    }
    for (unsigned long long t = first; t < n_tiles; t += gridDim.x) {
        const unsigned long long g = t * BLOCK + threadIdx.x;
        if (g >= n_groups) break;
        // the tile's groups, and this lane's byte offset inside the tile
        const unsigned long long g0 = t * BLOCK;
        const uint32_t groups = (uint32_t)(n_groups - g0 < (unsigned long long)BLOCK ? n_groups - g0 : BLOCK);
        const uint32_t lane16 = threadIdx.x * 16u;
        f32x4 X, Y, Z, VX, VY, VZ;
        uint32_t t4;
        if (NT) {
            X = tile_load4(tile_rsrc(a.src[0] + g0 * 4ull, groups * 16u), lane16);
            Y = tile_load4(tile_rsrc(a.src[1] + g0 * 4ull, groups * 16u), lane16);
            Z = tile_load4(tile_rsrc(a.src[2] + g0 * 4ull, groups * 16u), lane16);
            VX = tile_load4(tile_rsrc(a.src[3] + g0 * 4ull, groups * 16u), lane16);
            VY = tile_load4(tile_rsrc(a.src[4] + g0 * 4ull, groups * 16u), lane16);
            VZ = tile_load4(tile_rsrc(a.src[5] + g0 * 4ull, groups * 16u), lane16);
            t4 = __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(a.ty + g0 * 4ull, groups * 4u), (int)(threadIdx.x * 4u), 0, 2);
        } else {
            X = load4<false>(a.src[0], g);
            Y = load4<false>(a.src[1], g);
            Z = load4<false>(a.src[2], g);
            VX = load4<false>(a.src[3], g);
            VY = load4<false>(a.src[4], g);
            VZ = load4<false>(a.src[5], g);
            t4 = reinterpret_cast<const uint32_t*>(a.ty)[g];
        }

        float m[4];
        if (PACKED) step4_packed(U, M, t4, fetch, X, Y, Z, VX, VY, VZ, m);
        else step4_scalar(U, t4, fetch, X, Y, Z, VX, VY, VZ, m);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (DIAG) acc.add(m[k], X[k], Y[k], Z[k], VX[k], VY[k], VZ[k]);
            if (COUNT) lds_rank_add(lds_hist, tile_key(U, T, X[k], Y[k], Z[k], VX[k], VY[k], VZ[k]), true);
        }

        if (NT) {
            tile_store4(tile_rsrc(a.dst[0] + g0 * 4ull, groups * 16u), lane16, X);
            tile_store4(tile_rsrc(a.dst[1] + g0 * 4ull, groups * 16u), lane16, Y);
            tile_store4(tile_rsrc(a.dst[2] + g0 * 4ull, groups * 16u), lane16, Z);
            tile_store4(tile_rsrc(a.dst[3] + g0 * 4ull, groups * 16u), lane16, VX);
            tile_store4(tile_rsrc(a.dst[4] + g0 * 4ull, groups * 16u), lane16, VY);
            tile_store4(tile_rsrc(a.dst[5] + g0 * 4ull, groups * 16u), lane16, VZ);
        } else {
            store4<false>(a.dst[0], g, X);
            store4<false>(a.dst[1], g, Y);
            store4<false>(a.dst[2], g, Z);
            store4<false>(a.dst[3], g, VX);
            store4<false>(a.dst[4], g, VY);
            store4<false>(a.dst[5], g, VZ);
        }
    }

    // Up to 3 particles past the last full group: block 0, lanes 0..2 (scalar path: identical results).
    const unsigned long long tail = n_groups * 4ull + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 3 && tail < a.n) {
        float x = a.src[0][tail], y = a.src[1][tail], z = a.src[2][tail];
        float vx = a.src[3][tail], vy = a.src[4][tail], vz = a.src[5][tail];
        const float m = mass_of(U, a.ty[tail]);
        const f32x4 f = fetch(cell_index(U, x, y, z));
        integrate(U, f.x, f.y, f.z, m, 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
        a.dst[0][tail] = x; a.dst[1][tail] = y; a.dst[2][tail] = z;
        a.dst[3][tail] = vx; a.dst[4][tail] = vy; a.dst[5][tail] = vz;
        if (DIAG) acc.add(m, x, y, z, vx, vy, vz);
        if (COUNT) atomicAdd(&lds_hist[tile_key(U, T, x, y, z, vx, vy, vz)], 1u);
    }
}

// COUNT: also histogram the field tile of every OUTPUT position into a.hist
// (LDS counters per persistent block, one global atomic per non-empty bin), so
// that the next step can be a step_scatter_kernel without a counting pass.
// FIELD_LDS and COUNT are exclusive (both use the dynamic LDS).
// The counting variant with fused diagnostics is held to the 80 registers that three resident blocks allow.
template <bool DIAG, bool NT, bool FIELD_LDS, bool COUNT = false, int BLOCK = kStreamThreads>
__global__ __launch_bounds__(BLOCK, (DIAG && COUNT) ? 6 : 1) void step_stream_kernel(StreamArgs a, StepUniforms U, TileMap T)
{
    // Small fields (<= 64 KiB; the host decides when it pays) are staged in LDS once per persistent block;
    // larger ones are fetched through L2 / Infinity Cache.
    extern __shared__ __attribute__((aligned(16))) f32x4 lds_field[];
    uint32_t* lds_hist = reinterpret_cast<uint32_t*>(lds_field);
    if (FIELD_LDS) {
        for (uint32_t c = threadIdx.x; c < a.lds_cells; c += BLOCK) lds_field[c] = a.field[c];
        __syncthreads();
    }
    __shared__ uint32_t touched_sh;
    if (COUNT) {
        for (uint32_t b = threadIdx.x; b < T.n_bins; b += BLOCK) lds_hist[b] = 0u;
        if (threadIdx.x == 0) touched_sh = 0u;
        __syncthreads();
    }
    __shared__ float2 mass_tab[BLOCK / 64][8];

    DiagAcc acc;
    if (DIAG) acc.init();

    if (packed_path_ok(U, a.field_flag))
        stream_tiles<DIAG, NT, FIELD_LDS, COUNT, BLOCK, true>(a, U, T, lds_field, lds_hist, mass_tab[threadIdx.x >> 6], acc);
    else
        stream_tiles<DIAG, NT, FIELD_LDS, COUNT, BLOCK, false>(a, U, T, lds_field, lds_hist, mass_tab[threadIdx.x >> 6], acc);

    if (COUNT) {
        lds_barrier();   // LDS counters only: no need to drain the stores first
        // one tile per block in a counting launch; the ragged tail (block 0) counts with tile 0
        // (the counter group is that of the 2048-particle tile of the scatter step the block lies in)
        const uint32_t count_group = (uint32_t)((unsigned long long)blockIdx.x * BLOCK / kStreamThreads) % a.hist_groups;
        uint32_t touched = 0u;
        for (uint32_t b = threadIdx.x; b < T.n_bins; b += BLOCK) {
            const uint32_t c = lds_hist[b];
            if (c) { atomicAdd(&a.hist[count_group * T.n_bins + b], c); ++touched; }
        }
        // How many bins this block's particles fall into: 1-2 while the order is intact, more
        // the further they have drifted.  The host reads the total to pace the refreshes
        // (kRunSlots counters, so that the blocks do not all hit one address).
        if (a.runs) {
            if (touched) atomicAdd(&touched_sh, touched);
            lds_barrier();
            if (threadIdx.x == 0) atomicAdd(&a.runs[blockIdx.x % kRunSlots], touched_sh);
        }
    }
    if (DIAG) block_reduce_store<BLOCK>(acc, a.partials + blockIdx.x);
}

// Several consecutive updates of every particle in one pass over the arrays:
// the state stays in registers between the sub-steps, so HBM traffic is that of
// ONE step (49 B per particle) however many are fused; each sub-step fetches
// its own field cell.  The arithmetic of every sub-step is that of
// step_stream_kernel, so k fused steps equal k single steps bit for bit.  Only
// for callers that do not need the intermediate states (no rendering, no field
// edit, no parameter change between the fused steps).
template <bool DIAG>
__global__ __launch_bounds__(kStreamThreads) void step_stream_fused_kernel(StreamArgs a, StepUniforms U,
                                                                          uint32_t sub_steps)
{
    __shared__ float2 mass_tab[kStreamThreads / 64][8];
    const bool packed = packed_path_ok(U, a.field_flag);
    MassTable M;
    if (packed) M.init(mass_tab[threadIdx.x >> 6], U);
    DiagAcc acc;
    if (DIAG) acc.init();
    const unsigned long long n_groups = a.n / 4ull;
    const unsigned long long g = (unsigned long long)blockIdx.x * kStreamThreads + threadIdx.x;
    if (g < n_groups) {
        const unsigned long long g0 = (unsigned long long)blockIdx.x * kStreamThreads;
        const uint32_t groups = (uint32_t)(n_groups - g0 < (unsigned long long)kStreamThreads ? n_groups - g0 : kStreamThreads);
        const uint32_t lane16 = threadIdx.x * 16u;
        f32x4 X = tile_load4(tile_rsrc(a.src[0] + g0 * 4ull, groups * 16u), lane16);
        f32x4 Y = tile_load4(tile_rsrc(a.src[1] + g0 * 4ull, groups * 16u), lane16);
        f32x4 Z = tile_load4(tile_rsrc(a.src[2] + g0 * 4ull, groups * 16u), lane16);
        f32x4 VX = tile_load4(tile_rsrc(a.src[3] + g0 * 4ull, groups * 16u), lane16);
        f32x4 VY = tile_load4(tile_rsrc(a.src[4] + g0 * 4ull, groups * 16u), lane16);
        f32x4 VZ = tile_load4(tile_rsrc(a.src[5] + g0 * 4ull, groups * 16u), lane16);
        const uint32_t t4 = __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(a.ty + g0 * 4ull, groups * 4u), (int)(threadIdx.x * 4u), 0, 2);
        float m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = mass_of(U, (t4 >> (8 * k)) & 0xFFu);
        if (packed) {
            auto fetch = [&](uint32_t c) -> f32x4 { return fetch_cell(a.field, a.field3, c); };
            for (uint32_t s = 0; s < sub_steps; ++s) step4_packed(U, M, t4, fetch, X, Y, Z, VX, VY, VZ, m);
        } else {
        for (uint32_t s = 0; s < sub_steps; ++s) {
            f32x4 F[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) F[k] = fetch_cell(a.field, a.field3, cell_index(U, X[k], Y[k], Z[k]));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float x = X[k], y = Y[k], z = Z[k], vx = VX[k], vy = VY[k], vz = VZ[k];
                integrate(U, F[k].x, F[k].y, F[k].z, m[k], 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
                X[k] = x; Y[k] = y; Z[k] = z; VX[k] = vx; VY[k] = vy; VZ[k] = vz;
            }
        }
        }
        if (DIAG) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc.add(m[k], X[k], Y[k], Z[k], VX[k], VY[k], VZ[k]);
        }
        tile_store4(tile_rsrc(a.dst[0] + g0 * 4ull, groups * 16u), lane16, X);
        tile_store4(tile_rsrc(a.dst[1] + g0 * 4ull, groups * 16u), lane16, Y);
        tile_store4(tile_rsrc(a.dst[2] + g0 * 4ull, groups * 16u), lane16, Z);
        tile_store4(tile_rsrc(a.dst[3] + g0 * 4ull, groups * 16u), lane16, VX);
        tile_store4(tile_rsrc(a.dst[4] + g0 * 4ull, groups * 16u), lane16, VY);
        tile_store4(tile_rsrc(a.dst[5] + g0 * 4ull, groups * 16u), lane16, VZ);
    }
    // Up to 3 particles past the last full group: block 0, lanes 0..2.
    const unsigned long long tail = n_groups * 4ull + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 3 && tail < a.n) {
        float x = a.src[0][tail], y = a.src[1][tail], z = a.src[2][tail];
        float vx = a.src[3][tail], vy = a.src[4][tail], vz = a.src[5][tail];
        const float m = mass_of(U, a.ty[tail]);
        for (uint32_t s = 0; s < sub_steps; ++s) {
            const f32x4 f = fetch_cell(a.field, a.field3, cell_index(U, x, y, z));
            integrate(U, f.x, f.y, f.z, m, 0.0f, 0.0f, 0.0f, x, y, z, vx, vy, vz);
        }
        a.dst[0][tail] = x; a.dst[1][tail] = y; a.dst[2][tail] = z;
        a.dst[3][tail] = vx; a.dst[4][tail] = vy; a.dst[5][tail] = vz;
        if (DIAG) acc.add(m, x, y, z, vx, vy, vz);
    }
    if (DIAG) block_reduce_store<kStreamThreads>(acc, a.partials + blockIdx.x);
}

// -------------------------------------------------------- general kernel K2

// Parameter block as the reference lays it out (912 bytes): wgsl:17-28.
struct ParamsBlock {
    float polys[25][8];
    float masses[5][4];
    uint32_t dims[3];
    float delta_t, max_velocity, radius, cut_off_distance, distance_exponent;
};

// wgsl:51-56, in the shader's left-to-right order (not Horner).
__device__ __forceinline__ float eval_poly7(float x, const float* c)
{
    const float x2 = x * x;
    const float x4 = x2 * x2;
    const float x3 = x2 * x;
    float r = (c[7] * x4) * x3;
    r = r + (c[6] * x3) * x3;
    r = r + (c[5] * x2) * x3;
    r = r + c[4] * x4;
    r = r + c[3] * x3;
    r = r + c[2] * x2;
    r = r + c[1] * x;
    r = r + c[0];
    return r;
}

struct PairArgs {
    const float* __restrict__ src[6];
    float* __restrict__ dst[6];
    const uint8_t* __restrict__ ty;
    const f32x4* __restrict__ field;
    const ParamsBlock* __restrict__ params;
    unsigned long long n;
    DiagPartial* partials;
    // Thresholds on the SQUARED distance s, found on the host by bisection over
    // the float bit patterns, such that with the correctly rounded square root
    //   sqrtf(s) < 0.001f          <=>  s <  s_near     (wgsl:126)
    //   sqrtf(s) > cut_off_distance <=>  s >= s_far      (wgsl:129)
    // so rejected pairs never pay for the square root.
    float s_near;
    float s_far;
    // Gathered variant only: the partners, as (x, y, z, type bits) records in
    // global particle order; records whose type is >= 5 are padding.
    const f32x4* __restrict__ sources;
    unsigned long long n_sources;
};

// One term of wgsl:132-139 for a partner that passed the skip tests.
__device__ __forceinline__ void pair_term(const f32x4 o, float x, float y, float z, float exponent,
                                          const StepUniforms& U, const float* my_polys, float& tx, float& ty, float& tz)
{
    const float dx = o.x - x, dy = o.y - y, dz = o.z - z;
    const float s = (dx * dx + dy * dy) + dz * dz;
    const float len = sqrtf(s);
    // pow(x, 0) is exactly 1 for every finite positive x
    const float df = exponent == 0.0f ? 1.0f : powf(len, exponent);
    const uint32_t oty = __float_as_uint(o.w);
    const float m2 = mass_of(U, oty);
    const float pv = eval_poly7(len, &my_polys[oty * 8u]);
    tx = ((__fdiv_rn(dx, len) * pv) * m2) * df;    // wgsl:139
    ty = ((__fdiv_rn(dy, len) * pv) * m2) * df;
    tz = ((__fdiv_rn(dz, len) * pv) * m2) * df;
}

// One particle (this lane) against a tile of 64 staged candidates (x, y, z, type) in LDS, in candidate
// order.  Two passes: a cheap pass applies the skip tests of wgsl:120-131 on the squared distance to all 64
// and records the survivors in two 32-bit lane masks (candidate k < 32 is bit 31 - k of the first, so that
// a mask grows by one shift-and-or per candidate and no shift amount depends on k); a second pass
// evaluates wgsl:132-139 for the survivors only, lowest candidate first.  A wave therefore runs the
// expensive part max-over-lanes(survivors) times instead of once per candidate whenever any lane survives,
// and each lane still adds its terms in candidate order.  The particle itself is among the candidates and
// drops out with s = 0 < s_near.  Entries past the end of the candidates hold no_candidate(): a position
// whose squared distance from anything overflows to +inf (also from a particle at +-inf, where an infinite
// sentinel would give NaN), which fails the far test s >= s_far for every s_far the host passes (with no far
// cut-off at all -- an infinite or NaN cut_off_distance -- the host passes +inf, so that the one pair this
// deviates on is a real pair whose squared distance overflows fp32).
__device__ __forceinline__ f32x4 no_candidate() { return f32x4{3.4028234663852886e38f, 0.0f, 0.0f, 0.0f}; }

__device__ __forceinline__ void sweep_candidates(const f32x4* tile, bool live, float x, float y, float z,
                                                 float s_near, float s_far, float exponent, const StepUniforms& U,
                                                 const float* my_polys, float& ax, float& ay, float& az)
{
    uint32_t mask[2] = {0u, 0u};
    if (live) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                const f32x4 o = tile[32 * h + k];
                const float dx = o.x - x, dy = o.y - y, dz = o.z - z;
                const float s = (dx * dx + dy * dy) + dz * dz;
                const bool keep = !(s < s_near) && !(s >= s_far);
                mask[h] = (mask[h] << 1) | (keep ? 1u : 0u);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t m = mask[h];
        while (m) {
            const int k = __builtin_clz(m);
            m &= ~(0x80000000u >> k);
            float tx, ty, tz;
            pair_term(tile[32 * h + k], x, y, z, exponent, U, my_polys, tx, ty, tz);
            ax = ax + tx; ay = ay + ty; az = az + tz;
        }
    }
}

// All-pairs attraction (wgsl:111-144) tiled through LDS: a block owns PB
// particles (one per lane), every tile of PB source particles is staged once
// per block as (x, y, z, type) and swept by all lanes in index order, so each
// particle's sum is accumulated in the shader's order j = 0..N-1 (bit-exact;
// the price is that the parallelism is N lanes, no split of the j range).
// PB = 64 (one wave per block) spreads small sets over more CUs.
//
// GATHERED: the partners come from a.sources (n_sources records of all ranks'
// particles in global order, padding marked by a type >= 5) instead of this
// handle's own arrays; the particle's own record is among them and drops out
// with a zero distance like any coincident partner (wgsl:120-127).  Each sum
// still runs over j = 0..N-1 of the global set, so a sharded run adds the same
// terms in the same order as one device holding everything.
template <bool DIAG, int PB, bool GATHERED = false>
__global__ __launch_bounds__(PB) void step_pairs_kernel(PairArgs a, StepUniforms U)
{
    __shared__ f32x4 tile[PB];
    __shared__ float polys[25 * 8];

    for (int k = threadIdx.x; k < 25 * 8; k += PB)
        polys[k] = a.params->polys[k / 8][k % 8];
    const float exponent = a.params->distance_exponent;

    const unsigned long long i = (unsigned long long)blockIdx.x * PB + threadIdx.x;
    const bool live = i < a.n;
    float x = 0.f, y = 0.f, z = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
    uint32_t my_ty = 0u;
    if (live) {
        x = a.src[0][i]; y = a.src[1][i]; z = a.src[2][i];
        vx = a.src[3][i]; vy = a.src[4][i]; vz = a.src[5][i];
        my_ty = a.ty[i];
    }
    const float* my_polys = &polys[my_ty * 5u * 8u];
    float ax = 0.0f, ay = 0.0f, az = 0.0f;

    const unsigned long long n_partners = GATHERED ? a.n_sources : a.n;
    for (unsigned long long base = 0; base < n_partners; base += PB) {
        __syncthreads();
        const unsigned long long j = base + threadIdx.x;
        f32x4 t = no_candidate();
        if (j < n_partners) {
            if (GATHERED) {
                t = a.sources[j];
                if (__float_as_uint(t.w) >= 5u) t = no_candidate();   // padding between the ranks' ranges
            } else {
                t.x = a.src[0][j]; t.y = a.src[1][j]; t.z = a.src[2][j];
                t.w = __uint_as_float((uint32_t)a.ty[j]);
            }
        }
        tile[threadIdx.x] = t;
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < PB; sub += 64)   // whole tiles of 64: the last ones end in no_candidate()
            if (base + sub < n_partners)
                sweep_candidates(tile + sub, live, x, y, z, a.s_near, a.s_far, exponent, U, my_polys, ax, ay, az);
    }

    DiagAcc acc;
    if (DIAG) acc.init();
    if (live) {
        const float m = mass_of(U, my_ty);
        const f32x4 f = a.field[cell_index(U, x, y, z)];
        integrate(U, f.x, f.y, f.z, m, ax, ay, az, x, y, z, vx, vy, vz);
        a.dst[0][i] = x; a.dst[1][i] = y; a.dst[2][i] = z;
        a.dst[3][i] = vx; a.dst[4][i] = vy; a.dst[5][i] = vz;
        if (DIAG) acc.add(m, x, y, z, vx, vy, vz);
    }
    if (DIAG) block_reduce_store<PB>(acc, a.partials + blockIdx.x);
}

// Cell-binned pair term.  With the particles sorted by a uniform grid of cells
// whose edge is at least cut_off_distance (key = tile_key with the cell map C,
// lookahead 0), every partner of particle i within the cut-off lies in the 27
// cells around i's cell, and the three cells of one (x, y) column are adjacent
// bins, i.e. one contiguous slot range [bin_end[b0-1], bin_end[b1]).  The
// skip tests and the force expression are those of wgsl:120-139; what changes
// against step_pairs_kernel is the ORDER in which a particle's terms are added
// (cell order instead of j = 0..N-1), so results agree with the shader up to
// fp32 summation order instead of bit for bit.  Work per particle drops from N
// to the population of 27 cells.
template <bool DIAG>
__global__ __launch_bounds__(kBlockThreads) void step_pairs_binned_kernel(
    PairArgs a, StepUniforms U, TileMap C, const uint32_t* __restrict__ bin_end)
{
    __shared__ float polys[25 * 8];
    for (int k = threadIdx.x; k < 25 * 8; k += kBlockThreads)
        polys[k] = a.params->polys[k / 8][k % 8];
    __syncthreads();
    const float exponent = a.params->distance_exponent;

    const unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x;
    const bool live = i < a.n;
    DiagAcc acc;
    if (DIAG) acc.init();
    if (live) {
        float x = a.src[0][i], y = a.src[1][i], z = a.src[2][i];
        float vx = a.src[3][i], vy = a.src[4][i], vz = a.src[5][i];
        const uint32_t my_ty = a.ty[i];
        const float* my_polys = &polys[my_ty * 5u * 8u];
        // this particle's cell, exactly as the sort keyed it (shifts are 0 in a cell map)
        const float px = fminf(fmaxf(x, U.neg_slh), U.slh);
        const float py = fminf(fmaxf(y, U.neg_slh), U.slh);
        const float pz = fminf(fmaxf(z, U.neg_slh), U.slh);
        uint32_t cx = (uint32_t)((px - U.neg_slh) * C.inv_cell[0]);
        uint32_t cy = (uint32_t)((py - U.neg_slh) * C.inv_cell[1]);
        uint32_t cz = (uint32_t)((pz - U.neg_slh) * C.inv_cell[2]);
        cx = cx < C.max_ix[0] ? cx : C.max_ix[0];
        cy = cy < C.max_ix[1] ? cy : C.max_ix[1];
        cz = cz < C.max_ix[2] ? cz : C.max_ix[2];
        // R cells each way reach every partner within the cut-off (R * cell_len >= cut-off).
        const int R = (int)C.reach;
        const uint32_t z0 = cz > (uint32_t)R ? cz - (uint32_t)R : 0u;
        const uint32_t z1 = cz + (uint32_t)R < C.max_ix[2] ? cz + (uint32_t)R : C.max_ix[2];
        // Distance from this particle to a column of cells along one axis; the outermost cells
        // also hold the particles beyond the box, so they extend to infinity.  A thousandth of a
        // cell is taken off for the rounding of the sort key.
        auto axis_gap = [&](float p, int cell, uint32_t last) -> float {
            const float lo = cell == 0 ? -INFINITY : U.neg_slh + (float)cell * C.cell_len;
            const float hi = cell == (int)last ? INFINITY : U.neg_slh + (float)(cell + 1) * C.cell_len;
            const float gap = fmaxf(fmaxf(lo - p, p - hi), 0.0f);
            return fmaxf(gap - 0.001f * C.cell_len, 0.0f);
        };
        float ax = 0.0f, ay = 0.0f, az = 0.0f;
        for (int dx = -R; dx <= R; ++dx) {
            const int X = (int)cx + dx;
            if (X < 0 || X > (int)C.max_ix[0]) continue;
            const float gx = axis_gap(x, X, C.max_ix[0]);
            for (int dy = -R; dy <= R; ++dy) {
                const int Y = (int)cy + dy;
                if (Y < 0 || Y > (int)C.max_ix[1]) continue;
                const float gy = axis_gap(y, Y, C.max_ix[1]);
                if (gx * gx + gy * gy >= a.s_far) continue;   // no cell of this column is within the cut-off
                const uint32_t b0 = ((uint32_t)X * C.nty + (uint32_t)Y) * C.ntz + z0;
                const uint32_t b1 = b0 + (z1 - z0);
                const uint32_t j0 = b0 ? bin_end[b0 - 1u] : 0u;
                const uint32_t j1 = bin_end[b1];
                for (uint32_t j = j0; j < j1; ++j) {
                    const float ddx = a.src[0][j] - x, ddy = a.src[1][j] - y, ddz = a.src[2][j] - z;
                    const float s = (ddx * ddx + ddy * ddy) + ddz * ddz;
                    if (j == (uint32_t)i || s < a.s_near || s >= a.s_far) continue;   // wgsl:120-131
                    const float len = sqrtf(s);
                    const float df = exponent == 0.0f ? 1.0f : powf(len, exponent);
                    const uint32_t oty = a.ty[j];
                    const float m2 = mass_of(U, oty);
                    const float pv = eval_poly7(len, &my_polys[oty * 8u]);
                    ax = ax + ((__fdiv_rn(ddx, len) * pv) * m2) * df;    // wgsl:139
                    ay = ay + ((__fdiv_rn(ddy, len) * pv) * m2) * df;
                    az = az + ((__fdiv_rn(ddz, len) * pv) * m2) * df;
                }
            }
        }
        const float m = mass_of(U, my_ty);
        const f32x4 f = a.field[cell_index(U, x, y, z)];
        integrate(U, f.x, f.y, f.z, m, ax, ay, az, x, y, z, vx, vy, vz);
        a.dst[0][i] = x; a.dst[1][i] = y; a.dst[2][i] = z;
        a.dst[3][i] = vx; a.dst[4][i] = vy; a.dst[5][i] = vz;
        if (DIAG) acc.add(m, x, y, z, vx, vy, vz);
    }
    if (DIAG) block_reduce_store(acc, a.partials + blockIdx.x);
}

// The same cell-binned pair term with the candidates shared by a whole wave.  One wave (a block of 64
// threads) owns 64 consecutive slots of the cell-sorted order, i.e. the particles of one or two cells of
// one (x, y) column (rarely of two columns: then the columns are taken one after the other with the other
// column's lanes idle).  For every neighbour column the wave sweeps ONE slot range, the union of its lanes'
// ranges (cells zmin - R .. zmax + R of that column), staged 64 candidates at a time in LDS and swept with
// sweep_candidates: the skip tests of wgsl:120-131 for all 64 candidates first, then the force expression
// for each lane's survivors only.  Where step_pairs_binned_kernel pays the force expression once per
// candidate whenever ANY lane accepts it (always, with 64 lanes), a wave here pays it
// max-over-lanes(survivors) times per 64 candidates.  A candidate outside a lane's own neighbourhood is
// further away than the cut-off and fails the skip test, so every lane adds the same terms in the same
// order (columns in (dx, dy) order, slots ascending) as in step_pairs_binned_kernel.
template <bool DIAG>
__global__ __launch_bounds__(64) void step_pairs_binned_wave_kernel(
    PairArgs a, StepUniforms U, TileMap C, const uint32_t* __restrict__ bin_end)
{
    __shared__ f32x4 tile[64];
    __shared__ float polys[25 * 8];
    for (int k = threadIdx.x; k < 25 * 8; k += 64)
        polys[k] = a.params->polys[k / 8][k % 8];
    __syncthreads();
    const float exponent = a.params->distance_exponent;

    const unsigned long long i = (unsigned long long)blockIdx.x * 64ull + threadIdx.x;
    const bool live = i < a.n;
    float x = 0.f, y = 0.f, z = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
    uint32_t my_ty = 0u;
    if (live) {
        x = a.src[0][i]; y = a.src[1][i]; z = a.src[2][i];
        vx = a.src[3][i]; vy = a.src[4][i]; vz = a.src[5][i];
        my_ty = a.ty[i];
    }
    const float* my_polys = &polys[my_ty * 5u * 8u];
    // this particle's cell, exactly as the sort keyed it (shifts are 0 in a cell map)
    const float px = fminf(fmaxf(x, U.neg_slh), U.slh);
    const float py = fminf(fmaxf(y, U.neg_slh), U.slh);
    const float pz = fminf(fmaxf(z, U.neg_slh), U.slh);
    uint32_t cx = (uint32_t)((px - U.neg_slh) * C.inv_cell[0]);
    uint32_t cy = (uint32_t)((py - U.neg_slh) * C.inv_cell[1]);
    uint32_t cz = (uint32_t)((pz - U.neg_slh) * C.inv_cell[2]);
    cx = cx < C.max_ix[0] ? cx : C.max_ix[0];
    cy = cy < C.max_ix[1] ? cy : C.max_ix[1];
    cz = cz < C.max_ix[2] ? cz : C.max_ix[2];
    const int R = (int)C.reach;
    auto axis_gap = [&](float p, int cell, uint32_t last) -> float {   // see step_pairs_binned_kernel
        const float lo = cell == 0 ? -INFINITY : U.neg_slh + (float)cell * C.cell_len;
        const float hi = cell == (int)last ? INFINITY : U.neg_slh + (float)(cell + 1) * C.cell_len;
        const float gap = fmaxf(fmaxf(lo - p, p - hi), 0.0f);
        return fmaxf(gap - 0.001f * C.cell_len, 0.0f);
    };
    float ax = 0.0f, ay = 0.0f, az = 0.0f;

    unsigned long long todo = __builtin_amdgcn_ballot_w64(live);
    while (todo) {   // wave-uniform: one trip per (x, y) column present among the lanes
        const int leader = __builtin_ctzll(todo);
        const uint32_t CX = (uint32_t)__builtin_amdgcn_readlane((int)cx, leader);
        const uint32_t CY = (uint32_t)__builtin_amdgcn_readlane((int)cy, leader);
        const bool mine = live && cx == CX && cy == CY;
        const unsigned long long members = __builtin_amdgcn_ballot_w64(mine);
        todo &= ~members;
        // the slots are sorted z-fastest: the first member holds the lowest cell of the column, the last the highest
        const uint32_t zmin = (uint32_t)__builtin_amdgcn_readlane((int)cz, __builtin_ctzll(members));
        const uint32_t zmax = (uint32_t)__builtin_amdgcn_readlane((int)cz, 63 - __builtin_clzll(members));
        const uint32_t z0 = zmin > (uint32_t)R ? zmin - (uint32_t)R : 0u;
        const uint32_t z1 = zmax + (uint32_t)R < C.max_ix[2] ? zmax + (uint32_t)R : C.max_ix[2];
        for (int dx = -R; dx <= R; ++dx) {
            const int X = (int)CX + dx;
            if (X < 0 || X > (int)C.max_ix[0]) continue;
            const float gx = axis_gap(x, X, C.max_ix[0]);
            for (int dy = -R; dy <= R; ++dy) {
                const int Y = (int)CY + dy;
                if (Y < 0 || Y > (int)C.max_ix[1]) continue;
                const float gy = axis_gap(y, Y, C.max_ix[1]);
                const bool reach = mine && !(gx * gx + gy * gy >= a.s_far);   // some cell of this column may be within the cut-off
                if (__builtin_amdgcn_ballot_w64(reach) == 0ull) continue;
                const uint32_t b0 = ((uint32_t)X * C.nty + (uint32_t)Y) * C.ntz + z0;
                const uint32_t b1 = b0 + (z1 - z0);
                const uint32_t j0 = b0 ? bin_end[b0 - 1u] : 0u;
                const uint32_t j1 = bin_end[b1];
                auto candidate = [&](uint32_t j) -> f32x4 {
                    f32x4 t = no_candidate();
                    if (j < j1) {
                        t.x = a.src[0][j]; t.y = a.src[1][j]; t.z = a.src[2][j];
                        t.w = __uint_as_float((uint32_t)a.ty[j]);
                    }
                    return t;
                };
                f32x4 next = candidate(j0 + threadIdx.x);
                for (uint32_t base = j0; base < j1; base += 64u) {
                    __syncthreads();   // the previous tile has been swept
                    tile[threadIdx.x] = next;
                    __syncthreads();
                    next = candidate(base + 64u + threadIdx.x);   // in flight during the sweep
                    sweep_candidates(tile, reach, x, y, z, a.s_near, a.s_far, exponent, U, my_polys, ax, ay, az);
                }
            }
        }
    }

    DiagAcc acc;
    if (DIAG) acc.init();
    if (live) {
        const float m = mass_of(U, my_ty);
        const f32x4 f = a.field[cell_index(U, x, y, z)];
        integrate(U, f.x, f.y, f.z, m, ax, ay, az, x, y, z, vx, vy, vz);
        a.dst[0][i] = x; a.dst[1][i] = y; a.dst[2][i] = z;
        a.dst[3][i] = vx; a.dst[4][i] = vy; a.dst[5][i] = vz;
        if (DIAG) acc.add(m, x, y, z, vx, vy, vz);
    }
    if (DIAG) block_reduce_store<64>(acc, a.partials + blockIdx.x);
}

// ------------------------------------------------- standalone diagnostics K3

__global__ __launch_bounds__(kBlockThreads) void diag_reduce_kernel(
    StateSoA s, const uint8_t* __restrict__ ty, unsigned long long n, StepUniforms U,
    DiagPartial* partials)
{
    DiagAcc acc;
    acc.init();
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x;
         i < n; i += stride)
        acc.add(mass_of(U, ty[i]), s.a[0][i], s.a[1][i], s.a[2][i], s.a[3][i], s.a[4][i], s.a[5][i]);
    block_reduce_store(acc, partials + blockIdx.x);
}

// -------------------------------------------------- layout conversion K4

// 48-byte record of the reference: pos vec4 @0, vel vec4 @16, ty u32 @32.
struct ParticleAoS {
    float pos[4];
    float vel[4];
    uint32_t ty;
    uint32_t pad[3];
};

__global__ __launch_bounds__(kBlockThreads) void unpack_aos_kernel(
    const ParticleAoS* __restrict__ in, unsigned long long count, StateSoA s,
    uint8_t* __restrict__ ty, unsigned long long first)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x;
         i < count; i += stride) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(in[i].pos);
        const f32x4 v = *reinterpret_cast<const f32x4*>(in[i].vel);
        const unsigned long long o = first + i;
        s.a[0][o] = p.x; s.a[1][o] = p.y; s.a[2][o] = p.z;
        s.a[3][o] = v.x; s.a[4][o] = v.y; s.a[5][o] = v.z;
        ty[o] = (uint8_t)in[i].ty;
    }
}

// Writes what wgsl:180 writes: w lanes 1.0, padding 0.
// `slot_of` (may be null) maps a caller index to the slot that holds it.
__global__ __launch_bounds__(kBlockThreads) void pack_aos_kernel(
    StateSoA s, const uint8_t* __restrict__ ty, const uint32_t* __restrict__ slot_of,
    unsigned long long first, unsigned long long count, ParticleAoS* __restrict__ out)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x;
         i < count; i += stride) {
        const unsigned long long o = slot_of ? slot_of[first + i] : first + i;
        const f32x4 p = {s.a[0][o], s.a[1][o], s.a[2][o], 1.0f};
        const f32x4 v = {s.a[3][o], s.a[4][o], s.a[5][o], 1.0f};
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = {(uint32_t)ty[o], 0u, 0u, 0u};
        *reinterpret_cast<f32x4*>(out[i].pos) = p;
        *reinterpret_cast<f32x4*>(out[i].vel) = v;
        *reinterpret_cast<u32x4*>(&out[i].ty) = t;
    }
}

// Partner records of the gathered pair step: (x, y, z, type bits) per particle
// in caller order, then `pad` records marked as padding (type 0xFFFFFFFF).
__global__ __launch_bounds__(kBlockThreads) void pack_pair_sources_kernel(
    StateSoA s, const uint8_t* __restrict__ ty, const uint32_t* __restrict__ slot_of,
    unsigned long long count, unsigned long long pad, f32x4* __restrict__ out)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x;
         i < count + pad; i += stride) {
        f32x4 r = {0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)};
        if (i < count) {
            const unsigned long long o = slot_of ? slot_of[i] : i;
            r.x = s.a[0][o]; r.y = s.a[1][o]; r.z = s.a[2][o];
            r.w = __uint_as_float((uint32_t)ty[o]);
        }
        out[i] = r;
    }
}

// The same records for particles kept in tile order, walking the slots (coalesced
// reads) and writing each record at its caller index: one scattered 48-byte write
// per particle instead of seven scattered reads.  `orig[i]` is the caller index of
// slot i; records outside [first, first + count) are skipped.
__global__ __launch_bounds__(kBlockThreads) void pack_aos_by_slot_kernel(
    StateSoA s, const uint8_t* __restrict__ ty, const uint32_t* __restrict__ orig, unsigned long long n,
    unsigned long long first, unsigned long long count, ParticleAoS* __restrict__ out)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x; i < n; i += stride) {
        const unsigned long long o = orig ? orig[i] : i;
        if (o < first || o - first >= count) continue;
        const f32x4 p = {s.a[0][i], s.a[1][i], s.a[2][i], 1.0f};
        const f32x4 v = {s.a[3][i], s.a[4][i], s.a[5][i], 1.0f};
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = {(uint32_t)ty[i], 0u, 0u, 0u};
        ParticleAoS* r = out + (o - first);
        *reinterpret_cast<f32x4*>(r->pos) = p;
        *reinterpret_cast<f32x4*>(r->vel) = v;
        *reinterpret_cast<u32x4*>(&r->ty) = t;
    }
}

__global__ __launch_bounds__(kBlockThreads) void cell_index_kernel(
    StateSoA s, const uint32_t* __restrict__ slot_of, unsigned long long first,
    unsigned long long count, StepUniforms U, uint32_t* __restrict__ out)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x;
         i < count; i += stride) {
        const unsigned long long o = slot_of ? slot_of[first + i] : first + i;
        out[i] = cell_index(U, s.a[0][o], s.a[1][o], s.a[2][o]);
    }
}

// Order-independent checksum of the whole state: every particle contributes a 64-bit mix of its
// caller index, its six state words and its type; the contributions are summed (mod 2^64) and
// xor-ed, both of which do not care in which order, or by whom, they are combined.  Two states give
// the same pair iff (up to collisions of the mix) every caller index holds the same bits.
// orig: caller index of every slot, or null while the slots are in caller order.
__device__ __forceinline__ unsigned long long checksum_mix(unsigned long long h, uint32_t w)
{
    h = (h ^ w) * 0xFF51AFD7ED558CCDull;
    return h ^ (h >> 32);
}

__global__ __launch_bounds__(kBlockThreads) void state_checksum_kernel(
    StateSoA s, const uint8_t* __restrict__ ty, const uint32_t* __restrict__ orig, unsigned long long n,
    unsigned long long first_index, unsigned long long* __restrict__ out /* sum, xor */)
{
    __shared__ unsigned long long block_sum, block_xor;
    if (threadIdx.x == 0) { block_sum = 0ull; block_xor = 0ull; }
    __syncthreads();
    unsigned long long sum = 0ull, x = 0ull;
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x; i < n; i += stride) {
        const unsigned long long caller = orig ? orig[i] : i;
        unsigned long long h = (first_index + caller + 1ull) * 0x9E3779B97F4A7C15ull;
#pragma unroll
        for (int c = 0; c < 6; ++c) h = checksum_mix(h, __float_as_uint(s.a[c][i]));
        h = checksum_mix(h, ty[i]);
        sum += h;
        x ^= h;
    }
    atomicAdd(&block_sum, sum);
    atomicXor(&block_xor, x);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], block_sum);
        atomicXor(&out[1], block_xor);
    }
}

// Whether a field cell lies inside the window of the packed path: every component zero or
// within [2^-60, 2^60] in magnitude (NaN and infinities are outside).
__device__ __forceinline__ bool cell_in_window(float x, float y, float z)
{
    auto ok = [](float v) { const float a = fabsf(v); return v == 0.0f || (a >= 0x1p-60f && a <= 0x1p60f); };
    return ok(x) && ok(y) && ok(z);
}

// Brings what the streaming kernels need up to date with field[first, first + count): the copy
// packed to 12 bytes per cell (field3[3c .. 3c+2] = field[c].xyz; null: not kept) and the flag
// that tells them about components outside the packed path's window (set, never cleared here).
__global__ __launch_bounds__(kBlockThreads) void prepare_field_kernel(
    const f32x4* __restrict__ field, unsigned long long first, unsigned long long count, float* __restrict__ field3,
    uint32_t* __restrict__ field_flag)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    bool bad = false;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x; i < count; i += stride) {
        const f32x4 c = field[first + i];
        if (field3) {
            float* o = field3 + 3ull * (first + i);
            o[0] = c.x; o[1] = c.y; o[2] = c.z;
        }
        bad = bad || !cell_in_window(c.x, c.y, c.z);
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(field_flag, 1u);
}

// ------------------------------------------------ indexed field patch K5

// field[index[i]] = cells[i]: brush edits rewrite a list of cells
// (src/cursor.rs:79-103 writes grid.grid[ix] for ix in modify_vector_indices).
// `field_flag` (may be null): see prepare_field_kernel; raised when a patched cell leaves the window.
// An index that occurs more than once: one of its cells wins, which one is not defined (the host entry
// point removes all but the last occurrence first, which is what the reference's sequential loop leaves).
// The packed copy is brought up to date by repack_cells_kernel in a second launch, from the cells as
// they ended up: written here from cells[i], two occurrences of an index could leave the two copies of
// the field with different winners.
__global__ __launch_bounds__(kBlockThreads) void patch_field_kernel(
    f32x4* __restrict__ field, uint32_t* __restrict__ field_flag, unsigned long long n_cells,
    const uint32_t* __restrict__ index, const f32x4* __restrict__ cells, unsigned long long count)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x;
         i < count; i += stride) {
        const uint32_t c = index[i];
        if (c < n_cells) {
            const f32x4 v = cells[i];
            field[c] = v;
            if (field_flag && !cell_in_window(v.x, v.y, v.z)) atomicOr(field_flag, 1u);
        }
    }
}

// field3[3c .. 3c+2] = field[c].xyz for the listed cells (after patch_field_kernel has finished).
__global__ __launch_bounds__(kBlockThreads) void repack_cells_kernel(
    const f32x4* __restrict__ field, float* __restrict__ field3, unsigned long long n_cells,
    const uint32_t* __restrict__ index, unsigned long long count)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x;
         i < count; i += stride) {
        const uint32_t c = index[i];
        if (c < n_cells) {
            const f32x4 v = field[c];
            field3[3ull * c] = v.x; field3[3ull * c + 1] = v.y; field3[3ull * c + 2] = v.z;
        }
    }
}

// --------------------------------------------- seeded synthetic scene

__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float synth_uniform(unsigned long long seed, unsigned long long g,
                                               uint32_t axis)
{
    const unsigned long long h = mix64(seed ^ mix64(g * 8ull + (unsigned long long)axis));
    return (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
}

// Distribution of src/main.rs:41-48,156-169: extent * (u - 0.5), ty = index % 5.
__global__ __launch_bounds__(kBlockThreads) void generate_kernel(
    StateSoA s, uint8_t* __restrict__ ty, unsigned long long n, unsigned long long seed,
    unsigned long long first_global, float pos_extent, float vel_extent)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlockThreads;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlockThreads + threadIdx.x;
         i < n; i += stride) {
        const unsigned long long g = first_global + i;
#pragma unroll
        for (uint32_t c = 0; c < 3; ++c) {
            s.a[c][i] = pos_extent * (synth_uniform(seed, g, c) - 0.5f);
            s.a[3 + c][i] = vel_extent * (synth_uniform(seed, g, 3u + c) - 0.5f);
        }
        ty[i] = (uint8_t)(g % 5ull);
    }
}

}  // namespace pcl
