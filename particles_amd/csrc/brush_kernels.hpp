// brush_kernels.hpp — the field brush of the reference's cursor on the device.
//
// Reference: src/cursor.rs:79-116 (mouse_down selects the cells within
// falloff_dist of the cursor and remembers their vectors; mouse_moved rewrites
// exactly those cells) and src/cursor.rs:158-203 (EditMode::get_vector).  There
// the brush runs on the host over the whole field and the whole field is
// uploaded again every frame (src/main.rs:371-380); here the selection and the
// rewrite are two small kernels over the field already resident in HBM, so an
// edit moves 64 bytes (the cursor record) instead of 16 bytes per cell.
//
// Arithmetic follows the host code's f32 operations one for one (cgmath 0.18:
// dot = x*x + y*y + z*z summed in that order, normalize = v * (1 / |v|)),
// with correctly rounded sqrt and divide and no contraction, so results equal a
// CPU evaluation bit for bit.
#pragma once

#include "step_kernels.hpp"

namespace pcl {

// Mirrors pcl_brush_input (64 bytes).
struct BrushInput {
    float cursor_pos[3];
    float falloff_dist;
    float cam_right[3];   // Matrix3::x of the cursor's rotation: src/cursor.rs:168
    float strength;
    float cam_dir[3];     // Matrix3::z: src/cursor.rs:167
    int32_t mode;         // 0 Centered, 1 Shift, 2 Rotate
    int32_t relative;     // 1 Relative, 0 Absolute
    int32_t falloff;      // 0 Abrupt, 1 Linear, 2 InverseDistance
    int32_t reserved[2];
};
static_assert(sizeof(BrushInput) == 64, "brush record layout");

// What mouse_down leaves behind for mouse_moved: src/cursor.rs:12-14.
struct BrushState {
    float md_pos[3];
    uint32_t count;
};

struct FieldGeometry {
    uint32_t nx, ny, nz;
    float left;   // bounds.pos = -radius on every axis: src/sim_params.rs:33-45
    float dir;    // bounds.dir = 2 * radius
};

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 scale3(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float magnitude3(V3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ V3 normalize3(V3 a) { return scale3(a, __fdiv_rn(1.0f, magnitude3(a))); }
__device__ __forceinline__ V3 load3(const float* p) { return {p[0], p[1], p[2]}; }

// Grid::get_positions: src/grid.rs:119-138 (x slowest, z fastest).
__device__ __forceinline__ V3 cell_position(const FieldGeometry& G, uint32_t flat)
{
    const uint32_t iz = flat % G.nz, iy = (flat / G.nz) % G.ny, ix = flat / (G.nz * G.ny);
    return {G.left + G.dir * __fdiv_rn((float)ix + 0.5f, (float)G.nx),
            G.left + G.dir * __fdiv_rn((float)iy + 0.5f, (float)G.ny),
            G.left + G.dir * __fdiv_rn((float)iz + 0.5f, (float)G.nz)};
}

// EditMode::get_vector: src/cursor.rs:158-203.
__device__ __forceinline__ V3 brush_get_vector(const BrushInput& in, V3 v, V3 v_pos, V3 md_pos)
{
    const V3 cursor = load3(in.cursor_pos);
    V3 res;
    if (in.mode == 0) {
        res = normalize3(sub3(cursor, v_pos));                                  // :170-173
    } else if (in.mode == 1) {
        res = sub3(cursor, md_pos);                                             // :174
    } else {
        if (cursor.x == md_pos.x && cursor.y == md_pos.y && cursor.z == md_pos.z) {
            res = v;                                                            // :176-177
        } else {
            const V3 dc = sub3(cursor, md_pos);
            const V3 dv = normalize3(sub3(v_pos, md_pos));
            const V3 cd = load3(in.cam_dir);
            const V3 dir = {dv.y * cd.z - dv.z * cd.y, dv.z * cd.x - dv.x * cd.z, dv.x * cd.y - dv.y * cd.x};
            res = scale3(dir, dot3(dc, load3(in.cam_right)));                   // :183-188
        }
    }
    const float res_len = magnitude3(res);                                      // :192
    if (in.falloff == 1) {
        const float factor = __fdiv_rn(magnitude3(sub3(md_pos, v_pos)), in.falloff_dist);
        res = scale3(res, 1.0f - factor);                                       // :195-199
    } else if (in.falloff == 2) {
        res = scale3(res, __fdiv_rn(in.falloff_dist, res_len + 1.0f));          // :200
    }
    return scale3(res, in.strength);                                            // :202
}

// A record whose mode or falloff is none of the enumerated values (possible only
// for records read from device memory; the host variants reject them before the
// launch) edits nothing and sets kErrBrushRecord in the error word.
__device__ __forceinline__ bool brush_record_valid(const BrushInput& in)
{
    return in.mode >= 0 && in.mode <= 2 && in.falloff >= 0 && in.falloff <= 2;
}

// Cursor::mouse_down: src/cursor.rs:105-116.  `state` must be zeroed before the
// launch.  The selection is appended in arbitrary order (each selected cell is
// rewritten independently, so the order never reaches the field).
__global__ __launch_bounds__(kBlockThreads) void brush_select_kernel(
    BrushInput by_value, const BrushInput* __restrict__ from_device, FieldGeometry G, uint32_t n_cells,
    const f32x4* __restrict__ field, BrushState* __restrict__ state, uint32_t* __restrict__ sel_index,
    f32x4* __restrict__ sel_vector, uint32_t* error)
{
    const BrushInput in = from_device ? *from_device : by_value;
    if (!brush_record_valid(in)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(error, kErrBrushRecord);
        return;   // state->count stays 0: an empty selection
    }
    const V3 cursor = load3(in.cursor_pos);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state->md_pos[0] = cursor.x; state->md_pos[1] = cursor.y; state->md_pos[2] = cursor.z;
    }
    const uint32_t stride = gridDim.x * kBlockThreads;
    for (uint32_t c = blockIdx.x * kBlockThreads + threadIdx.x; c < n_cells; c += stride) {
        if (magnitude3(sub3(cell_position(G, c), cursor)) < in.falloff_dist) {
            const uint32_t slot = atomicAdd(&state->count, 1u);
            sel_index[slot] = c;
            sel_vector[slot] = field[c];
        }
    }
}

// Cursor::mouse_moved: src/cursor.rs:79-103.  `field3` (may be null): the copy packed to 12 bytes per
// cell, kept equal; `field_flag` (may be null): raised when a rewritten cell leaves the window of the
// streaming kernels' packed path (cell_in_window).
__global__ __launch_bounds__(kBlockThreads) void brush_apply_kernel(
    BrushInput by_value, const BrushInput* __restrict__ from_device, FieldGeometry G,
    f32x4* __restrict__ field, float* __restrict__ field3, uint32_t* __restrict__ field_flag,
    const BrushState* __restrict__ state, const uint32_t* __restrict__ sel_index,
    const f32x4* __restrict__ sel_vector, uint32_t* error)
{
    const BrushInput in = from_device ? *from_device : by_value;
    if (!brush_record_valid(in)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(error, kErrBrushRecord);
        return;
    }
    const V3 md_pos = load3(state->md_pos);
    const uint32_t count = state->count;
    const uint32_t stride = gridDim.x * kBlockThreads;
    for (uint32_t k = blockIdx.x * kBlockThreads + threadIdx.x; k < count; k += stride) {
        const uint32_t c = sel_index[k];
        const f32x4 cur = field[c];
        const V3 d = brush_get_vector(in, {cur.x, cur.y, cur.z}, cell_position(G, c), md_pos);
        const f32x4 md_v = sel_vector[k];
        f32x4 out;
        if (in.relative) {
            out.x = d.x + md_v.x; out.y = d.y + md_v.y; out.z = d.z + md_v.z;   // :98
        } else {
            out.x = d.x; out.y = d.y; out.z = d.z;                              // :99
        }
        out.w = 1.0f;                                                           // src/grid.rs:97-103
        field[c] = out;
        if (field3) { field3[3ull * c] = out.x; field3[3ull * c + 1u] = out.y; field3[3ull * c + 2u] = out.z; }
        if (field_flag && !cell_in_window(out.x, out.y, out.z)) atomicOr(field_flag, 1u);
    }
}

}  // namespace pcl
