// authoring.cpp — host-side helpers with which the reference's GUI builds the
// inputs of the update: force curves through eight sample points
// (src/poly7.rs:37-58,120-147) and the initial fields (src/grid.rs:49-103,119-138,
// src/sim_params.rs:33-60).  Plain host C++: these run once per edit, not per
// frame.  Every fp32 operation is done in the reference's order with no
// contraction (-ffp-contract=off), so the results equal the Rust ones bit for
// bit wherever Rust's are defined by IEEE arithmetic; f32::powi is
// compiler-rt's __powisf2 (square-and-multiply, low bit first), which is also
// what LLVM expands a constant exponent to.
#include "particles_hip.h"

#include <cmath>
#include <cstddef>


namespace {

// compiler-rt lib/builtins/powisf2.c for b >= 0
float powi(float a, int b)
{
    float r = 1.0f;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return r;
}

constexpr int N = 8;

// src/poly7.rs:78-91: m[into] += mul * m[from], element by element
void row_mul_add(float mul, float m[N][N], int from, int into)
{
    for (int i = 0; i < N; ++i) {
        const float fr = m[from][i];
        m[into][i] += mul * fr;
    }
}

// src/poly7.rs:93-97
void row_div(float div, float m[N][N], int row)
{
    for (int i = 0; i < N; ++i) m[row][i] /= div;
}

// src/poly7.rs:120-147: Gauss-Jordan without pivot search, on m and the identity alongside
void inverse(float m[N][N], float res[N][N])
{
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) res[i][j] = i == j ? 1.0f : 0.0f;
    for (int i = 0; i < N; ++i) {
        for (int j = i + 1; j < N; ++j) {
            const float pivot = -m[j][i] / m[i][i];
            row_mul_add(pivot, m, i, j);
            row_mul_add(pivot, res, i, j);
        }
        const float div = m[i][i];
        row_div(div, m, i);
        row_div(div, res, i);
    }
    for (int i = N - 1; i >= 0; --i)
        for (int j = i - 1; j >= 0; --j) {
            const float pivot = -m[j][i] / m[i][i];
            row_mul_add(pivot, m, i, j);
            row_mul_add(pivot, res, i, j);
        }
}

}  // namespace

extern "C" {

int pcl_poly7_from_points(const float (*points)[2], pcl_poly7* out)
{
    if (!points || !out) return PCL_ERR_INVALID;
    float m[N][N], inv[N][N];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) m[i][j] = powi(points[i][0], j);      // src/poly7.rs:40-43
    inverse(m, inv);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            if (!std::isfinite(inv[i][j])) return PCL_ERR_SINGULAR;       // src/poly7.rs:46-52: None
    for (int row = 0; row < N; ++row) {                                   // src/poly7.rs:110-118
        float acc = 0.0f;
        for (int i = 0; i < N; ++i) acc += inv[row][i] * points[i][1];
        out->coeffs[row] = acc;
    }
    return PCL_OK;
}

float pcl_poly7_eval(const pcl_poly7* p, float x)
{
    float res = 0.0f, x_to_the_i = 1.0f;                                  // src/poly7.rs:60-68
    if (!p) return 0.0f;
    for (int i = 0; i < N; ++i) {
        res += x_to_the_i * p->coeffs[i];
        x_to_the_i *= x;
    }
    return res;
}

void pcl_grid_centered(size_t nx, size_t ny, size_t nz, float (*out)[4])
{
    if (!out) return;
    const float cx = (float)nx * 0.5f, cy = (float)ny * 0.5f, cz = (float)nz * 0.5f;   // src/grid.rs:71
    size_t k = 0;
    for (size_t ix = 0; ix < nx; ++ix)
        for (size_t iy = 0; iy < ny; ++iy)
            for (size_t iz = 0; iz < nz; ++iz, ++k) {
                out[k][0] = cx - (float)ix;
                out[k][1] = cy - (float)iy;
                out[k][2] = cz - (float)iz;
                out[k][3] = 1.0f;                                         // src/grid.rs:97-103
            }
}

void pcl_grid_uniform(size_t nx, size_t ny, size_t nz, const float value[3], float (*out)[4])
{
    if (!out || !value) return;
    const size_t cap = nx * ny * nz;                                      // src/grid.rs:49-54
    for (size_t k = 0; k < cap; ++k) {
        out[k][0] = value[0]; out[k][1] = value[1]; out[k][2] = value[2];
        out[k][3] = 1.0f;
    }
}

void pcl_grid_positions(size_t nx, size_t ny, size_t nz, const float bounds_pos[3], const float bounds_dir[3],
                        float (*out)[4])
{
    if (!out || !bounds_pos || !bounds_dir) return;
    size_t k = 0;
    for (size_t ix = 0; ix < nx; ++ix)
        for (size_t iy = 0; iy < ny; ++iy)
            for (size_t iz = 0; iz < nz; ++iz, ++k) {                      // src/grid.rs:119-138
                out[k][0] = bounds_pos[0] + bounds_dir[0] * (((float)ix + 0.5f) / (float)nx);
                out[k][1] = bounds_pos[1] + bounds_dir[1] * (((float)iy + 0.5f) / (float)ny);
                out[k][2] = bounds_pos[2] + bounds_dir[2] * (((float)iz + 0.5f) / (float)nz);
                out[k][3] = 1.0f;
            }
}

int pcl_sim_params_force_grid(const pcl_sim_params* params, int centered, float (*out)[4], size_t capacity_cells)
{
    if (!params || !out) return PCL_ERR_INVALID;
    const size_t nx = params->force_grid_dimensions[0], ny = params->force_grid_dimensions[1],
                 nz = params->force_grid_dimensions[2];
    if (nx * ny * nz > capacity_cells) return PCL_ERR_OUT_OF_RANGE;
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    if (centered) pcl_grid_centered(nx, ny, nz, out);                     // src/sim_params.rs:33-45
    else pcl_grid_uniform(nx, ny, nz, zero, out);                         // src/sim_params.rs:47-60
    return PCL_OK;
}

}  // extern "C"
