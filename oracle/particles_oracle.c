/*
 * particles_oracle.c — CPU restatement of the reference's particle update.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library or its Python
 * host layer links, imports or calls this file; it is the checker used by
 * tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED.  The reference (sphereflow/particles) is Rust + WGSL with
 * no tests, fixtures or golden vectors, and it cannot be built or run in this
 * environment (no rustc/cargo, no WGSL compiler, no Vulkan loader).  All of
 * the path's arithmetic is in src/compute.wgsl, which this file transcribes
 * statement by statement; what wgpu 0.18.0 / naga 0.14.2 (Cargo.lock) add on
 * top — the precision of exp/pow/sqrt/normalize/divide, FMA contraction and
 * the out-of-bounds policy for the field fetch — is implementation-defined
 * there and is frozen here as follows:
 *
 *   length(v)    = sqrtf((v.x*v.x + v.y*v.y) + v.z*v.z)
 *   normalize(v) = v / length(v)                (three IEEE divides)
 *   pow, exp     = powf, expf (libm)
 *   clamp(x,a,b) = fminf(fmaxf(x, a), b)
 *   u32(f)       = truncation (f is always within [0, dim] here)
 *   no FMA contraction (build with -ffp-contract=off)
 *   field index past the end -> last cell (naga's "Restrict" bounds policy)
 *   inputs are finite and ty < 5
 *
 * Function-by-function citations are relative to the reference checkout.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- layouts: src/main.rs:76-83, src/main.rs:196-201, src/poly7.rs:6-10,
 *      src/sim_params.rs:6-17 (mirrored by src/compute.wgsl:1-28,40-49) ---- */
typedef struct {
    float    pos[4];
    float    vel[4];
    uint32_t ty;
    uint32_t pad[3];
} orc_particle;

typedef struct { float coeffs[8]; } orc_poly7;
typedef struct { float mass; float pad[3]; } orc_mass_wrap;

typedef struct {
    orc_poly7     attraction_force[25];
    orc_mass_wrap particle_type_masses[5];
    uint32_t      force_grid_dimensions[3];
    float         delta_t;
    float         max_velocity;
    float         bounding_volume_radius;
    float         cut_off_distance;
    float         distance_exponent;
} orc_sim_params;

typedef struct {
    double count;
    double kinetic_energy;
    float  pos_max[3];
    float  neg_pos_min[3];
    float  pad[2];
} orc_diagnostics;

_Static_assert(sizeof(orc_particle) == 48, "Particle is 48 bytes");
_Static_assert(offsetof(orc_particle, vel) == 16, "vel @16");
_Static_assert(offsetof(orc_particle, ty) == 32, "ty @32");
_Static_assert(sizeof(orc_poly7) == 32, "Poly7 is 32 bytes");
_Static_assert(sizeof(orc_mass_wrap) == 16, "MassWrap is 16 bytes");
_Static_assert(sizeof(orc_sim_params) == 912, "SimParams is 912 bytes");
_Static_assert(offsetof(orc_sim_params, particle_type_masses) == 800, "masses @800");
_Static_assert(offsetof(orc_sim_params, force_grid_dimensions) == 880, "dims @880");
_Static_assert(offsetof(orc_sim_params, delta_t) == 892, "delta_t @892");
_Static_assert(offsetof(orc_sim_params, max_velocity) == 896, "max_velocity @896");
_Static_assert(offsetof(orc_sim_params, bounding_volume_radius) == 900, "radius @900");
_Static_assert(offsetof(orc_sim_params, cut_off_distance) == 904, "cutoff @904");
_Static_assert(offsetof(orc_sim_params, distance_exponent) == 908, "exponent @908");
_Static_assert(sizeof(orc_diagnostics) == 48, "diagnostics is 48 bytes");

size_t orc_sizeof_particle(void) { return sizeof(orc_particle); }
size_t orc_sizeof_sim_params(void) { return sizeof(orc_sim_params); }

/* ---- src/sim_params.rs:20-31 ---- */
void orc_sim_params_default(orc_sim_params* p)
{
    memset(p, 0, sizeof *p);
    for (int i = 0; i < 5; ++i) p->particle_type_masses[i].mass = 1.0f;
    p->force_grid_dimensions[0] = 10;
    p->force_grid_dimensions[1] = 10;
    p->force_grid_dimensions[2] = 10;
    p->delta_t = 0.0f;
    p->max_velocity = 100.0f;
    p->bounding_volume_radius = 10.0f;
    p->cut_off_distance = 1.0f;
    p->distance_exponent = 0.0f;
}

/* ---- src/compute.wgsl:51-56 ---- */
float orc_eval_poly7(float x, const float* coeffs)
{
    /* fields h,g,f,e,d,c,b,a are coeffs[0..8]: src/compute.wgsl:40-49 */
    const float h = coeffs[0], g = coeffs[1], f = coeffs[2], e = coeffs[3];
    const float d = coeffs[4], c = coeffs[5], b = coeffs[6], a = coeffs[7];
    const float x2 = x * x;
    const float x4 = x2 * x2;
    const float x3 = x2 * x;
    float r = (a * x4) * x3;
    r = r + (b * x3) * x3;
    r = r + (c * x2) * x3;
    r = r + d * x4;
    r = r + e * x3;
    r = r + f * x2;
    r = r + g * x;
    r = r + h;
    return r;
}

/* ---- src/compute.wgsl:58-66 ---- */
float orc_wrap_symmetrically(float val, float max)
{
    if (val > max) return val - 2.0f * max;
    if (val < -max) return val + 2.0f * max;
    return val;
}

/* ---- src/compute.wgsl:68-79 ---- */
float orc_map_range(float val, float start_a, float end_a, float start_b, float end_b)
{
    if (start_a == end_a) return start_b;
    const float valc = fminf(fmaxf(val, start_a), end_a);
    const float oz = (valc - start_a) / (end_a - start_a);
    const float valb = oz * (end_b - start_b) + start_b;
    return valb;
}

static uint32_t to_u32(float f)
{
    /* WGSL u32(f32) truncates; saturate outside the range so that the C
     * conversion is never undefined. */
    if (!(f >= 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

/* ---- src/compute.wgsl:81-89 ---- */
uint32_t orc_force_grid_index(float bounding_cube_side_length, const uint32_t* dims,
                              const float* v)
{
    const float slh = bounding_cube_side_length * 0.5f;
    const uint32_t ixx = to_u32(orc_map_range(v[0], -slh, slh, 0.0f, (float)dims[0]));
    const uint32_t ixy = to_u32(orc_map_range(v[1], -slh, slh, 0.0f, (float)dims[1]));
    const uint32_t ixz = to_u32(orc_map_range(v[2], -slh, slh, 0.0f, (float)dims[2]));
    return ixx * dims[1] * dims[2] + ixy * dims[2] + ixz;
}

static uint32_t clamp_cell(uint32_t fgi, size_t n_cells)
{
    /* frozen out-of-bounds policy: clamp to the last cell */
    if ((size_t)fgi >= n_cells) return (uint32_t)(n_cells - 1);
    return fgi;
}

static float length3(const float* v)
{
    return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
}

/*
 * The pair loop contributes exactly +0 when every coefficient of every
 * polynomial is +-0 (finite inputs: dN*0*m*df = +-0 and +0 + +-0 = +0), or
 * when no distance can pass both cut-offs (cut_off_distance < 0.001):
 * src/compute.wgsl:126-131,139.
 */
int orc_pair_term_inert(const orc_sim_params* p)
{
    if (p->cut_off_distance < 0.001f) return 1;
    for (int k = 0; k < 25; ++k)
        for (int c = 0; c < 8; ++c)
            if (p->attraction_force[k].coeffs[c] != 0.0f) return 0;
    /* 0 * mass * pow(len, exponent) is +-0 only while both factors are finite */
    for (int k = 0; k < 5; ++k)
        if (!isfinite(p->particle_type_masses[k].mass)) return 0;
    const double e = (double)p->distance_exponent;
    if (e == 0.0) return 1;
    if (!isfinite(e)) return 0;
    const double hi = isfinite(p->cut_off_distance) ? (double)p->cut_off_distance : 3.4028234663852886e38;
    const double reach = fmax(fabs(log(0.001)), fabs(log(hi)));
    return fabs(e) * reach <= 69.0;   /* pow(len, e) <= 1e30 for every len in [0.001, cut_off] */
}

/* ---- one invocation of src/compute.wgsl:97-181 for particle `index` ---- */
static void step_one(const orc_sim_params* params, const orc_particle* src,
                     orc_particle* dst, size_t total, size_t index,
                     const float (*grid)[4], size_t n_cells, int skip_pairs, float decay)
{
    float vPos[3] = { src[index].pos[0], src[index].pos[1], src[index].pos[2] };
    float vVel[3] = { src[index].vel[0], src[index].vel[1], src[index].vel[2] };
    const uint32_t vParticleType = src[index].ty;
    const float vMass = params->particle_type_masses[vParticleType].mass;
    const float deltaT = params->delta_t;

    float cAcc[3] = { 0.0f, 0.0f, 0.0f };

    /* src/compute.wgsl:115-144; `continue` runs the continuing block, so i
     * always advances. */
    if (!skip_pairs) {
        for (size_t i = 0; i < total; ++i) {
            if (i == index) continue;
            float direction[3] = { src[i].pos[0] - vPos[0], src[i].pos[1] - vPos[1],
                                   src[i].pos[2] - vPos[2] };
            const float direction_length = length3(direction);
            if (direction_length < 0.001f) continue;
            if (direction_length > params->cut_off_distance) continue;
            const float distance_factor = powf(direction_length, params->distance_exponent);
            const float len = length3(direction); /* normalize() recomputes the length */
            const float directionN[3] = { direction[0] / len, direction[1] / len,
                                          direction[2] / len };
            const uint32_t attractionForceIndex = src[i].ty + vParticleType * 5u;
            const float m2 = params->particle_type_masses[src[i].ty].mass;
            const float pv = orc_eval_poly7(direction_length,
                                            params->attraction_force[attractionForceIndex].coeffs);
            for (int c = 0; c < 3; ++c) {
                float t = directionN[c] * pv;
                t = t * m2;
                t = t * distance_factor;
                cAcc[c] = cAcc[c] + t;
            }
        }
    }

    /* src/compute.wgsl:147-148 */
    const uint32_t fgi = clamp_cell(
        orc_force_grid_index(params->bounding_volume_radius * 2.0f,
                             params->force_grid_dimensions, vPos),
        n_cells);
    for (int c = 0; c < 3; ++c) {
        const float t = 10.0f * grid[fgi][c];
        cAcc[c] = cAcc[c] + t / vMass;
    }

    /* src/compute.wgsl:151,154; decay = exp(-deltaT) is the same for every
     * particle and is evaluated once per step by the callers. */
    for (int c = 0; c < 3; ++c) vVel[c] = vVel[c] * decay;
    for (int c = 0; c < 3; ++c) vVel[c] = vVel[c] + cAcc[c] * deltaT;

    /* src/compute.wgsl:157-160 */
    const float vel = length3(vVel);
    if (vel > 0.001f) {
        const float nlen = length3(vVel);
        const float capped = fminf(fmaxf(vel, 0.0f), params->max_velocity);
        for (int c = 0; c < 3; ++c) vVel[c] = (vVel[c] / nlen) * capped;
    }

    /* src/compute.wgsl:163 */
    for (int c = 0; c < 3; ++c) vPos[c] = vPos[c] + vVel[c] * deltaT;

    /* src/compute.wgsl:170-172 */
    for (int c = 0; c < 3; ++c)
        vPos[c] = orc_wrap_symmetrically(vPos[c], params->bounding_volume_radius);

    /* src/compute.wgsl:180 */
    orc_particle out;
    out.pos[0] = vPos[0]; out.pos[1] = vPos[1]; out.pos[2] = vPos[2]; out.pos[3] = 1.0f;
    out.vel[0] = vVel[0]; out.vel[1] = vVel[1]; out.vel[2] = vVel[2]; out.vel[3] = 1.0f;
    out.ty = src[index].ty;
    out.pad[0] = out.pad[1] = out.pad[2] = 0;
    dst[index] = out;
}

/* What bounds the effect of the ORDER in which a particle's pair terms are added (the cell-binned
 * kernel of the library adds them cell by cell, src/compute.wgsl:115-144 in index order): for
 * particles [first, first + count), per component the sum of the terms' magnitudes (in double) and the
 * number of terms that passed the skip tests of wgsl:126-131.  Recursive fp32 summation of m terms in
 * any order stays within (m - 1) * 2^-24 * sum|t| of the exact sum. */
void orc_pair_term_magnitudes(const orc_sim_params* params, const orc_particle* src, size_t total,
                              size_t first, size_t count, double (*abs_sum)[3], uint32_t* terms)
{
    for (size_t k = 0; k < count; ++k) {
        const size_t index = first + k;
        const float vPos[3] = { src[index].pos[0], src[index].pos[1], src[index].pos[2] };
        const uint32_t vParticleType = src[index].ty;
        double acc[3] = { 0.0, 0.0, 0.0 };
        uint32_t m = 0;
        for (size_t i = 0; i < total; ++i) {
            if (i == index) continue;
            float direction[3] = { src[i].pos[0] - vPos[0], src[i].pos[1] - vPos[1], src[i].pos[2] - vPos[2] };
            const float direction_length = length3(direction);
            if (direction_length < 0.001f) continue;
            if (direction_length > params->cut_off_distance) continue;
            const float distance_factor = powf(direction_length, params->distance_exponent);
            const float len = length3(direction);
            const float m2 = params->particle_type_masses[src[i].ty].mass;
            const float pv = orc_eval_poly7(direction_length,
                                            params->attraction_force[src[i].ty + vParticleType * 5u].coeffs);
            for (int c = 0; c < 3; ++c) {
                float t = (direction[c] / len) * pv;
                t = t * m2;
                t = t * distance_factor;
                acc[c] += fabs((double)t);
            }
            ++m;
        }
        abs_sum[k][0] = acc[0]; abs_sum[k][1] = acc[1]; abs_sum[k][2] = acc[2];
        terms[k] = m;
    }
}

/* ---- cut-off cells of the library's cell-binned pair term -------------------------------------
 * No counterpart in the reference (its loop visits every particle); restated here because the
 * cells are index work that must match bit for bit.  The library sorts the particles by a uniform
 * grid of n x n x n cells over the bounding volume whose edge is at least 1.0001 x the cut-off
 * (swept one cell each way), or at least half of that (swept two cells each way) where that sweeps
 * the smaller volume; n is capped (20, or 64 for large sets).  A particle's cell along an axis is
 * u32((clamp(p, -slh, slh) - (-slh)) * inv_cell) clamped to n - 1, with inv_cell = (float)(n / side),
 * and its key (cx * n + cy) * n + cz. */
void orc_pair_cell_map(const orc_sim_params* p, int allow_reach2, uint32_t max_axis,
                       uint32_t* cells_per_axis, uint32_t* reach, float* inv_cell)
{
    const double side = 2.0 * (double)p->bounding_volume_radius;
    const double edge = (double)p->cut_off_distance * 1.0001;
    uint32_t n = 1, r = 1;
    if (side > 0.0 && edge > 0.0 && isfinite(side) && isfinite(edge)) {
        const double cap = (double)max_axis;
        const double n1 = fmin(cap, fmax(1.0, floor(side / edge)));
        const double n2 = fmin(cap, fmax(1.0, floor(2.0 * side / edge)));
        if (allow_reach2 && 5.0 / n2 < 3.0 / n1) { n = (uint32_t)n2; r = 2; }
        else { n = (uint32_t)n1; r = 1; }
    }
    *cells_per_axis = n;
    *reach = r;
    *inv_cell = (side > 0.0 && isfinite(side)) ? (float)((double)n / side) : 0.0f;
}

void orc_pair_cell_keys(const orc_sim_params* p, const orc_particle* particles, size_t count,
                        uint32_t cells_per_axis, float inv_cell, uint32_t* keys)
{
    const float side = p->bounding_volume_radius * 2.0f;
    const float slh = side * 0.5f, neg_slh = -slh;
    const uint32_t last = cells_per_axis - 1u;
    for (size_t k = 0; k < count; ++k) {
        uint32_t c[3];
        for (int a = 0; a < 3; ++a) {
            const float v = fminf(fmaxf(particles[k].pos[a], neg_slh), slh);
            const float t = (v - neg_slh) * inv_cell;
            uint32_t i = to_u32(t);
            c[a] = i < last ? i : last;
        }
        keys[k] = (c[0] * cells_per_axis + c[1]) * cells_per_axis + c[2];
    }
}

/*
 * Update particles [first, first+count) of a set of `total`.  `mode`:
 * 0 = always run the pair loop, as the shader does;
 * 1 = skip it when orc_pair_term_inert() proves it contributes nothing.
 */
void orc_step_range(const orc_sim_params* params, const orc_particle* src, orc_particle* dst,
                    size_t total, size_t first, size_t count,
                    const float (*grid)[4], size_t n_cells, int mode)
{
    const int skip = (mode == 1) && orc_pair_term_inert(params);
    const float decay = expf(-params->delta_t);
    for (size_t i = first; i < first + count; ++i)
        step_one(params, src, dst, total, i, grid, n_cells, skip, decay);
}

/* Whole-set update, optionally split by index range over `threads` threads. */
void orc_step(const orc_sim_params* params, const orc_particle* src, orc_particle* dst,
              size_t total, const float (*grid)[4], size_t n_cells, int mode, int threads)
{
    const int skip = (mode == 1) && orc_pair_term_inert(params);
    const float decay = expf(-params->delta_t);
#ifdef _OPENMP
    if (threads > 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
        for (long long i = 0; i < (long long)total; ++i)
            step_one(params, src, dst, total, (size_t)i, grid, n_cells, skip, decay);
        return;
    }
#endif
    (void)threads;
    for (size_t i = 0; i < total; ++i)
        step_one(params, src, dst, total, i, grid, n_cells, skip, decay);
}

/*
 * `steps` updates with the reference's ping-pong (src/compute.rs:147-180,
 * 221-227): step k reads buffer (k & 1) and writes the other.  Returns the
 * index (0/1) of the buffer holding the latest state.
 */
int orc_run(const orc_sim_params* params, orc_particle* buf0, orc_particle* buf1,
            size_t total, const float (*grid)[4], size_t n_cells,
            int steps, int mode, int threads)
{
    orc_particle* bufs[2] = { buf0, buf1 };
    int swap = 0;
    for (int s = 0; s < steps; ++s) {
        orc_step(params, bufs[swap], bufs[swap ^ 1], total, grid, n_cells, mode, threads);
        swap ^= 1;
    }
    return swap;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Flat cell index each particle would fetch: src/compute.wgsl:147. */
void orc_cell_indices(const orc_sim_params* params, const orc_particle* particles, size_t n,
                      size_t n_cells, uint32_t* out)
{
    for (size_t i = 0; i < n; ++i)
        out[i] = clamp_cell(orc_force_grid_index(params->bounding_volume_radius * 2.0f,
                                                 params->force_grid_dimensions,
                                                 particles[i].pos),
                            n_cells);
}

/* Diagnostics (not in the reference): count, sum 0.5*m*|v|^2 in f64, bounds. */
void orc_diagnostics_of(const orc_sim_params* params, const orc_particle* particles, size_t n,
                        orc_diagnostics* out)
{
    double ke = 0.0;
    float mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    float nmn[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < n; ++i) {
        const double m = (double)params->particle_type_masses[particles[i].ty].mass;
        const double vx = particles[i].vel[0], vy = particles[i].vel[1], vz = particles[i].vel[2];
        ke += 0.5 * m * ((vx * vx + vy * vy) + vz * vz);
        for (int c = 0; c < 3; ++c) {
            mx[c] = fmaxf(mx[c], particles[i].pos[c]);
            nmn[c] = fmaxf(nmn[c], -particles[i].pos[c]);
        }
    }
    memset(out, 0, sizeof *out);
    out->count = (double)n;
    out->kinetic_energy = ke;
    for (int c = 0; c < 3; ++c) { out->pos_max[c] = mx[c]; out->neg_pos_min[c] = nmn[c]; }
}

/* ---- scene generators ---- */

/* Lattice of src/main.rs:117-154: pos = (ix/nx*max.x, iy/ny*max.y, iz/nz*max.z, 1),
 * vel = 0 (all four lanes), ty = flat index % 5. */
void orc_lattice_particles(orc_particle* out, float max_x, float max_y, float max_z,
                           size_t nx, size_t ny, size_t nz)
{
    for (size_t ix = 0; ix < nx; ++ix)
        for (size_t iy = 0; iy < ny; ++iy)
            for (size_t iz = 0; iz < nz; ++iz) {
                const size_t index = ix * ny * nz + iy * nz + iz;
                orc_particle p;
                memset(&p, 0, sizeof p);
                p.pos[0] = ((float)ix / (float)nx) * max_x;
                p.pos[1] = ((float)iy / (float)ny) * max_y;
                p.pos[2] = ((float)iz / (float)nz) * max_z;
                p.pos[3] = 1.0f;
                p.ty = (uint32_t)(index % 5);
                out[index] = p;
            }
}

/* Grid::new_centered + get_force_vectors: src/grid.rs:68-103.
 * v = (nx,ny,nz)*0.5 - (ix,iy,iz), w = 1. */
void orc_grid_centered(float (*out)[4], size_t nx, size_t ny, size_t nz)
{
    const float cx = (float)nx * 0.5f, cy = (float)ny * 0.5f, cz = (float)nz * 0.5f;
    size_t k = 0;
    for (size_t ix = 0; ix < nx; ++ix)
        for (size_t iy = 0; iy < ny; ++iy)
            for (size_t iz = 0; iz < nz; ++iz) {
                out[k][0] = cx - (float)ix;
                out[k][1] = cy - (float)iy;
                out[k][2] = cz - (float)iz;
                out[k][3] = 1.0f;
                ++k;
            }
}

/* Grid::new_uniform + get_force_vectors: src/grid.rs:49-64,97-103. */
void orc_grid_uniform(float (*out)[4], size_t nx, size_t ny, size_t nz, float vx, float vy, float vz)
{
    for (size_t k = 0; k < nx * ny * nz; ++k) {
        out[k][0] = vx; out[k][1] = vy; out[k][2] = vz; out[k][3] = 1.0f;
    }
}

/* f32::powi for a non-negative exponent.  Rust lowers it to llvm.powi, which
 * is compiler-rt's __powisf2 (or, for a constant exponent, LLVM's in-line
 * expansion of the same square-and-multiply sequence, low bit first). */
static float orc_powi(float a, int b)
{
    float r = 1.0f;
    while (1) {
        if (b & 1) r = r * a;
        b = b / 2;
        if (b == 0) break;
        a = a * a;
    }
    return r;
}

/* Poly7::from_points: src/poly7.rs:37-58, with inverse() of src/poly7.rs:120-147
 * (elimination without pivot search; every row operation applied to the matrix
 * and to the identity beside it, element by element, left to right) and
 * multiply_vector of src/poly7.rs:110-118.  Returns 0 where the reference
 * returns None (a non-finite entry in the inverse), 1 otherwise. */
int orc_poly7_from_points(const float (*pts)[2], float coeffs[8])
{
    float m[8][8], res[8][8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            m[i][j] = orc_powi(pts[i][0], j);
            res[i][j] = (i == j) ? 1.0f : 0.0f;
        }
    for (int i = 0; i < 8; ++i) {
        for (int j = i + 1; j < 8; ++j) {
            const float pivot = -m[j][i] / m[i][i];
            for (int k = 0; k < 8; ++k) m[j][k] = m[j][k] + pivot * m[i][k];
            for (int k = 0; k < 8; ++k) res[j][k] = res[j][k] + pivot * res[i][k];
        }
        const float div = m[i][i];
        for (int k = 0; k < 8; ++k) m[i][k] = m[i][k] / div;
        for (int k = 0; k < 8; ++k) res[i][k] = res[i][k] / div;
    }
    for (int i = 7; i >= 0; --i)
        for (int j = i - 1; j >= 0; --j) {
            const float pivot = -m[j][i] / m[i][i];
            for (int k = 0; k < 8; ++k) m[j][k] = m[j][k] + pivot * m[i][k];
            for (int k = 0; k < 8; ++k) res[j][k] = res[j][k] + pivot * res[i][k];
        }
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j)
            if (!isfinite(res[i][j])) return 0;
    for (int row = 0; row < 8; ++row) {
        float acc = 0.0f;
        for (int i = 0; i < 8; ++i) acc = acc + res[row][i] * pts[i][1];
        coeffs[row] = acc;
    }
    return 1;
}

/* Poly7::eval: src/poly7.rs:60-68 (ascending powers, running product). */
float orc_poly7_eval_host(const float coeffs[8], float x)
{
    float res = 0.0f, x_to_the_i = 1.0f;
    for (int i = 0; i < 8; ++i) {
        res = res + x_to_the_i * coeffs[i];
        x_to_the_i = x_to_the_i * x;
    }
    return res;
}

/* Grid::get_positions: src/grid.rs:119-138 with bounds pos = -r, dir = 2r
 * (src/sim_params.rs:33-45). */
void orc_grid_positions(float (*out)[4], size_t nx, size_t ny, size_t nz, float radius)
{
    const float left = -radius, dir = 2.0f * radius;
    size_t k = 0;
    for (size_t ix = 0; ix < nx; ++ix)
        for (size_t iy = 0; iy < ny; ++iy)
            for (size_t iz = 0; iz < nz; ++iz) {
                out[k][0] = left + dir * (((float)ix + 0.5f) / (float)nx);
                out[k][1] = left + dir * (((float)iy + 0.5f) / (float)ny);
                out[k][2] = left + dir * (((float)iz + 0.5f) / (float)nz);
                out[k][3] = 1.0f;
                ++k;
            }
}

/* Counter-based hash for the seeded synthetic scene (splitmix64 finaliser).
 * The reference's generator (src/main.rs:41-48,156-169) is unseeded thread
 * RNG; only its distribution max*(u - 0.5), u in [0,1), is reproduced. */
static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

float orc_synth_uniform(uint64_t seed, uint64_t global_index, uint32_t axis)
{
    const uint64_t h = mix64(seed ^ mix64(global_index * 8ULL + (uint64_t)axis));
    return (float)(h >> 40) * (1.0f / 16777216.0f); /* top 24 bits -> [0,1) */
}

void orc_synth_particles(orc_particle* out, size_t n, uint64_t seed, uint64_t first_global_index,
                         float pos_extent, float vel_extent)
{
    for (size_t i = 0; i < n; ++i) {
        const uint64_t g = first_global_index + (uint64_t)i;
        orc_particle p;
        memset(&p, 0, sizeof p);
        for (uint32_t c = 0; c < 3; ++c) {
            p.pos[c] = pos_extent * (orc_synth_uniform(seed, g, c) - 0.5f);
            p.vel[c] = vel_extent * (orc_synth_uniform(seed, g, 3u + c) - 0.5f);
        }
        p.pos[3] = 1.0f;
        p.vel[3] = 1.0f;
        p.ty = (uint32_t)(g % 5ULL);
        out[i] = p;
    }
}

/* ------------------------------------------------------------------------
 * Field brush: src/cursor.rs:79-116,158-203 (host-side Rust in the reference;
 * f32 arithmetic without contraction).  cgmath 0.18 (Cargo.lock) supplies the
 * vector operations, restated here from its published definitions:
 *   dot(a,b)      = a.x*b.x + a.y*b.y + a.z*b.z      (element-wise product, summed x+y+z)
 *   magnitude(v)  = sqrt(dot(v,v)),  distance(a,b) = magnitude(b - a)
 *   normalize(v)  = v * (1.0 / magnitude(v))          (normalize_to(1.0))
 *   cross(a,b)    = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
 *   s * v, v * s  = component-wise
 * cam_rot is given as its three columns: right (Matrix3::x), up (::y),
 * dir (::z); src/cursor.rs:167-168 reads .z and .x.
 * ------------------------------------------------------------------------ */
typedef struct {
    int32_t mode;      /* 0 Centered, 1 Shift, 2 Rotate: src/cursor.rs:124-129 */
    int32_t relative;  /* 1 Relative, 0 Absolute:        src/cursor.rs:131-135 */
    int32_t falloff;   /* 0 Abrupt, 1 Linear, 2 InverseDistance: src/cursor.rs:137-142 */
    float   falloff_dist;
    float   strength;
} orc_edit_mode;

/* EditMode::default: src/cursor.rs:206-216 */
void orc_edit_mode_default(orc_edit_mode* m)
{
    m->mode = 1; m->relative = 1; m->falloff = 0; m->falloff_dist = 1.0f; m->strength = 1.0f;
}

static float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static float magnitude3(const float* v) { return sqrtf(dot3(v, v)); }
static void normalize3(const float* v, float* out)
{
    const float s = 1.0f / magnitude3(v);
    out[0] = v[0] * s; out[1] = v[1] * s; out[2] = v[2] * s;
}

/* EditMode::get_vector: src/cursor.rs:158-203 */
void orc_brush_get_vector(const orc_edit_mode* m, const float* v, const float* v_pos, const float* md_pos,
                          const float* cursor_pos, const float* cam_right, const float* cam_dir, float* out)
{
    float res[3];
    if (m->mode == 0) {                       /* Centered: :170-173 */
        const float d[3] = {cursor_pos[0] - v_pos[0], cursor_pos[1] - v_pos[1], cursor_pos[2] - v_pos[2]};
        normalize3(d, res);
    } else if (m->mode == 1) {                /* Shift: :174 */
        for (int k = 0; k < 3; ++k) res[k] = cursor_pos[k] - md_pos[k];
    } else {                                  /* Rotate: :175-190 */
        if (cursor_pos[0] == md_pos[0] && cursor_pos[1] == md_pos[1] && cursor_pos[2] == md_pos[2]) {
            for (int k = 0; k < 3; ++k) res[k] = v[k];
        } else {
            const float dc[3] = {cursor_pos[0] - md_pos[0], cursor_pos[1] - md_pos[1], cursor_pos[2] - md_pos[2]};
            const float d[3] = {v_pos[0] - md_pos[0], v_pos[1] - md_pos[1], v_pos[2] - md_pos[2]};
            float dv[3];
            normalize3(d, dv);
            const float dir[3] = {dv[1] * cam_dir[2] - dv[2] * cam_dir[1],
                                  dv[2] * cam_dir[0] - dv[0] * cam_dir[2],
                                  dv[0] * cam_dir[1] - dv[1] * cam_dir[0]};
            const float s = dot3(dc, cam_right);
            for (int k = 0; k < 3; ++k) res[k] = dir[k] * s;
        }
    }
    const float res_len = magnitude3(res);    /* :192 */
    if (m->falloff == 1) {                    /* Linear: :195-199 */
        const float d[3] = {md_pos[0] - v_pos[0], md_pos[1] - v_pos[1], md_pos[2] - v_pos[2]};
        const float factor = magnitude3(d) / m->falloff_dist;
        const float s = 1.0f - factor;
        for (int k = 0; k < 3; ++k) res[k] = s * res[k];
    } else if (m->falloff == 2) {             /* InverseDistance: :200 */
        const float s = m->falloff_dist / (res_len + 1.0f);
        for (int k = 0; k < 3; ++k) res[k] = s * res[k];
    }
    for (int k = 0; k < 3; ++k) out[k] = res[k] * m->strength;   /* :202 */
}

static void cell_position(const orc_sim_params* p, size_t flat, float* out)
{
    const size_t ny = p->force_grid_dimensions[1], nz = p->force_grid_dimensions[2];
    const size_t ix = flat / (ny * nz), iy = (flat / nz) % ny, iz = flat % nz;
    const float left = -p->bounding_volume_radius, dir = 2.0f * p->bounding_volume_radius;
    out[0] = left + dir * (((float)ix + 0.5f) / (float)p->force_grid_dimensions[0]);   /* src/grid.rs:124-131 */
    out[1] = left + dir * (((float)iy + 0.5f) / (float)ny);
    out[2] = left + dir * (((float)iz + 0.5f) / (float)nz);
}

/* Cursor::mouse_down: src/cursor.rs:105-116.  Writes the selected cell indices
 * (ascending) and their current vectors; returns how many (never more than
 * `capacity` are written). */
size_t orc_brush_mouse_down(const orc_sim_params* p, const float (*grid)[4], const float* cursor_pos,
                            float falloff_dist, uint32_t* out_indices, float (*out_vectors)[3], size_t capacity)
{
    const size_t n = (size_t)p->force_grid_dimensions[0] * p->force_grid_dimensions[1] * p->force_grid_dimensions[2];
    size_t count = 0;
    for (size_t ix = 0; ix < n; ++ix) {
        float vpos[3];
        cell_position(p, ix, vpos);
        const float d[3] = {vpos[0] - cursor_pos[0], vpos[1] - cursor_pos[1], vpos[2] - cursor_pos[2]};   /* pos.distance(vpos) */
        if (magnitude3(d) < falloff_dist) {
            if (count < capacity) {
                out_indices[count] = (uint32_t)ix;
                for (int k = 0; k < 3; ++k) out_vectors[count][k] = grid[ix][k];
            }
            ++count;
        }
    }
    return count;
}

/* Cursor::mouse_moved: src/cursor.rs:79-103; rewrites the selected cells of
 * `grid` in place (w stays 1.0: src/grid.rs:97-103). */
void orc_brush_mouse_moved(const orc_sim_params* p, float (*grid)[4], const orc_edit_mode* m,
                           const uint32_t* indices, const float (*md_vectors)[3], size_t count,
                           const float* md_pos, const float* cursor_pos, const float* cam_right, const float* cam_dir)
{
    for (size_t k = 0; k < count; ++k) {
        const size_t ix = indices[k];
        float vpos[3], disp[3];
        cell_position(p, ix, vpos);
        orc_brush_get_vector(m, grid[ix], vpos, md_pos, cursor_pos, cam_right, cam_dir, disp);
        for (int c = 0; c < 3; ++c)
            grid[ix][c] = m->relative ? disp[c] + md_vectors[k][c] : disp[c];   /* :97-100 */
        grid[ix][3] = 1.0f;
    }
}
