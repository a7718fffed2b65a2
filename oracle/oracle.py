"""ctypes wrapper of oracle/particles_oracle.c (TEST INFRASTRUCTURE ONLY).

The C file is the oracle: a scalar restatement of the reference's
src/compute.wgsl with its numeric choices frozen in its header (PARITY
UNPINNED: the reference ships no golden vectors and cannot be run here).
This module only marshals numpy arrays in and out of it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libparticles_oracle.so")

PARTICLE_DTYPE = np.dtype(
    [("pos", np.float32, (4,)), ("vel", np.float32, (4,)), ("ty", np.uint32), ("_padd", np.uint32, (3,))])
SIM_PARAMS_BYTES = 912

MODE_ALWAYS_PAIRS = 0   # run the all-pairs loop as the shader always does
MODE_SKIP_INERT = 1     # skip it when it provably adds +0


class Diagnostics(C.Structure):
    _fields_ = [("count", C.c_double), ("kinetic_energy", C.c_double),
                ("pos_max", C.c_float * 3), ("neg_pos_min", C.c_float * 3), ("pad", C.c_float * 2)]


_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, sz, i, f, u64, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_uint64, C.c_uint32
    lib.orc_sizeof_particle.restype = sz
    lib.orc_sizeof_sim_params.restype = sz
    lib.orc_sim_params_default.argtypes = [vp]
    lib.orc_eval_poly7.restype = f
    lib.orc_eval_poly7.argtypes = [f, vp]
    lib.orc_wrap_symmetrically.restype = f
    lib.orc_wrap_symmetrically.argtypes = [f, f]
    lib.orc_map_range.restype = f
    lib.orc_map_range.argtypes = [f, f, f, f, f]
    lib.orc_force_grid_index.restype = u32
    lib.orc_force_grid_index.argtypes = [f, vp, vp]
    lib.orc_pair_term_inert.restype = i
    lib.orc_pair_term_inert.argtypes = [vp]
    lib.orc_step_range.argtypes = [vp, vp, vp, sz, sz, sz, vp, sz, i]
    lib.orc_step.argtypes = [vp, vp, vp, sz, vp, sz, i, i]
    lib.orc_run.restype = i
    lib.orc_run.argtypes = [vp, vp, vp, sz, vp, sz, i, i, i]
    lib.orc_max_threads.restype = i
    lib.orc_cell_indices.argtypes = [vp, vp, sz, sz, vp]
    lib.orc_pair_term_magnitudes.argtypes = [vp, vp, sz, sz, sz, vp, vp]
    lib.orc_pair_cell_map.argtypes = [vp, i, u32, vp, vp, vp]
    lib.orc_pair_cell_keys.argtypes = [vp, vp, sz, u32, f, vp]
    lib.orc_diagnostics_of.argtypes = [vp, vp, sz, C.POINTER(Diagnostics)]
    lib.orc_lattice_particles.argtypes = [vp, f, f, f, sz, sz, sz]
    lib.orc_grid_centered.argtypes = [vp, sz, sz, sz]
    lib.orc_grid_positions.argtypes = [vp, sz, sz, sz, f]
    lib.orc_grid_uniform.argtypes = [vp, sz, sz, sz, f, f, f]
    lib.orc_poly7_from_points.restype = i
    lib.orc_poly7_from_points.argtypes = [vp, vp]
    lib.orc_poly7_eval_host.restype = f
    lib.orc_poly7_eval_host.argtypes = [vp, f]
    lib.orc_synth_uniform.restype = f
    lib.orc_synth_uniform.argtypes = [u64, u64, u32]
    lib.orc_synth_particles.argtypes = [vp, sz, u64, u64, f, f]
    lib.orc_edit_mode_default.argtypes = [vp]
    lib.orc_brush_get_vector.argtypes = [vp] * 8
    lib.orc_brush_mouse_down.restype = sz
    lib.orc_brush_mouse_down.argtypes = [vp, vp, vp, f, vp, vp, sz]
    lib.orc_brush_mouse_moved.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]
    assert lib.orc_sizeof_particle() == 48 and lib.orc_sizeof_sim_params() == SIM_PARAMS_BYTES
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _params_buf(params):
    """Accepts bytes, a numpy record, or any object with .tobytes()."""
    b = params if isinstance(params, (bytes, bytearray)) else params.tobytes()
    assert len(b) == SIM_PARAMS_BYTES
    return np.frombuffer(bytearray(b), dtype=np.uint8)


def _particles(a):
    a = np.ascontiguousarray(a)
    assert a.dtype.itemsize == 48
    return a.view(PARTICLE_DTYPE).reshape(-1)


def _grid(g):
    g = np.ascontiguousarray(g, dtype=np.float32)
    assert g.ndim == 2 and g.shape[1] == 4
    return g


def default_params_bytes():
    buf = np.zeros(SIM_PARAMS_BYTES, dtype=np.uint8)
    load().orc_sim_params_default(_p(buf))
    return buf.tobytes()


def eval_poly7(x, coeffs):
    c = np.ascontiguousarray(coeffs, dtype=np.float32)
    return float(load().orc_eval_poly7(float(x), _p(c)))


def wrap_symmetrically(val, mx):
    return float(load().orc_wrap_symmetrically(float(val), float(mx)))


def map_range(val, a0, a1, b0, b1):
    return float(load().orc_map_range(float(val), float(a0), float(a1), float(b0), float(b1)))


def force_grid_index(side, dims, v):
    d = np.ascontiguousarray(dims, dtype=np.uint32)
    vv = np.ascontiguousarray(v, dtype=np.float32)
    return int(load().orc_force_grid_index(float(side), _p(d), _p(vv)))


def pair_term_inert(params):
    return bool(load().orc_pair_term_inert(_p(_params_buf(params))))


def step(params, particles, grid, mode=MODE_ALWAYS_PAIRS, threads=1):
    """One update of every particle; returns the new array."""
    src = _particles(particles)
    dst = np.zeros_like(src)
    g = _grid(grid)
    load().orc_step(_p(_params_buf(params)), _p(src), _p(dst), src.shape[0], _p(g), g.shape[0], mode, threads)
    return dst


def step_range(params, particles, grid, first, count, mode=MODE_ALWAYS_PAIRS):
    """Update particles [first, first+count) of the set; returns those records."""
    src = _particles(particles)
    dst = np.zeros_like(src)
    g = _grid(grid)
    load().orc_step_range(_p(_params_buf(params)), _p(src), _p(dst), src.shape[0], first, count,
                          _p(g), g.shape[0], mode)
    return dst[first:first + count]


def run(params, particles, grid, steps, mode=MODE_ALWAYS_PAIRS, threads=1):
    """`steps` ping-pong updates; returns the latest state."""
    a = _particles(particles).copy()
    b = a.copy()  # both buffers start equal: src/compute.rs:52-61
    g = _grid(grid)
    which = load().orc_run(_p(_params_buf(params)), _p(a), _p(b), a.shape[0], _p(g), g.shape[0],
                           int(steps), mode, threads)
    return b if which else a


def cell_indices(params, particles, n_cells):
    src = _particles(particles)
    out = np.zeros(src.shape[0], dtype=np.uint32)
    load().orc_cell_indices(_p(_params_buf(params)), _p(src), src.shape[0], int(n_cells), _p(out))
    return out


def pair_term_magnitudes(params, particles, first=0, count=None):
    """Per particle of [first, first + count): sum of |pair term| per component (float64, shape (count, 3))
    and the number of terms the skip tests let through (what bounds the effect of the summation order)."""
    p = _particles(particles)
    count = p.shape[0] - first if count is None else count
    abs_sum = np.zeros((count, 3), dtype=np.float64)
    terms = np.zeros(count, dtype=np.uint32)
    load().orc_pair_term_magnitudes(_p(_params_buf(params)), _p(p), p.shape[0], first, count, _p(abs_sum), _p(terms))
    return abs_sum, terms


def pair_cell_map(params, allow_reach2=True, max_axis=20):
    """(cells per axis, reach, inv_cell) of the cut-off cells the library bins the pair term by."""
    n, r, inv = C.c_uint32(), C.c_uint32(), C.c_float()
    load().orc_pair_cell_map(_p(_params_buf(params)), int(allow_reach2), int(max_axis), C.byref(n), C.byref(r), C.byref(inv))
    return n.value, r.value, np.float32(inv.value)


def pair_cell_keys(params, particles, cells_per_axis, inv_cell):
    p = _particles(particles)
    keys = np.zeros(p.shape[0], dtype=np.uint32)
    load().orc_pair_cell_keys(_p(_params_buf(params)), _p(p), p.shape[0], int(cells_per_axis), float(inv_cell), _p(keys))
    return keys


def diagnostics(params, particles):
    src = _particles(particles)
    d = Diagnostics()
    load().orc_diagnostics_of(_p(_params_buf(params)), _p(src), src.shape[0], C.byref(d))
    return {"count": int(d.count), "kinetic_energy": float(d.kinetic_energy),
            "pos_max": np.array(d.pos_max[:], dtype=np.float32),
            "pos_min": -np.array(d.neg_pos_min[:], dtype=np.float32)}


def lattice_particles(max_xyz, nx, ny, nz):
    out = np.zeros(nx * ny * nz, dtype=PARTICLE_DTYPE)
    load().orc_lattice_particles(_p(out), float(max_xyz[0]), float(max_xyz[1]), float(max_xyz[2]), nx, ny, nz)
    return out


def grid_centered(nx, ny, nz):
    out = np.zeros((nx * ny * nz, 4), dtype=np.float32)
    load().orc_grid_centered(_p(out), nx, ny, nz)
    return out


def grid_uniform(nx, ny, nz, value=(0.0, 0.0, 0.0)):
    out = np.zeros((nx * ny * nz, 4), dtype=np.float32)
    load().orc_grid_uniform(_p(out), nx, ny, nz, float(value[0]), float(value[1]), float(value[2]))
    return out


def poly7_from_points(points):
    """Coefficients (float32[8]) of the polynomial through 8 (x, y) points, or None."""
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(8, 2)
    out = np.zeros(8, dtype=np.float32)
    return out if load().orc_poly7_from_points(_p(pts), _p(out)) else None


def poly7_eval_host(coeffs, x):
    c = np.ascontiguousarray(coeffs, dtype=np.float32).reshape(8)
    return np.float32(load().orc_poly7_eval_host(_p(c), float(x)))


def grid_positions(nx, ny, nz, radius):
    out = np.zeros((nx * ny * nz, 4), dtype=np.float32)
    load().orc_grid_positions(_p(out), nx, ny, nz, float(radius))
    return out


def synth_particles(n, seed, first_global_index=0, pos_extent=2.0, vel_extent=10.0):
    out = np.zeros(int(n), dtype=PARTICLE_DTYPE)
    load().orc_synth_particles(_p(out), int(n), int(seed), int(first_global_index), float(pos_extent), float(vel_extent))
    return out


def max_threads():
    return int(load().orc_max_threads())


# ---- field brush (src/cursor.rs) -------------------------------------------
EDIT_MODE_DTYPE = np.dtype([("mode", np.int32), ("relative", np.int32), ("falloff", np.int32),
                            ("falloff_dist", np.float32), ("strength", np.float32)])
MODES = {"centered": 0, "shift": 1, "rotate": 2}
FALLOFFS = {"abrupt": 0, "linear": 1, "inverse_distance": 2}


def edit_mode(mode="shift", relative=True, falloff="abrupt", falloff_dist=1.0, strength=1.0):
    m = np.zeros(1, dtype=EDIT_MODE_DTYPE)
    m["mode"], m["relative"], m["falloff"] = MODES[mode], int(relative), FALLOFFS[falloff]
    m["falloff_dist"], m["strength"] = falloff_dist, strength
    return m


def _f3(v):
    return np.ascontiguousarray(v, dtype=np.float32).reshape(3)


def brush_get_vector(m, v, v_pos, md_pos, cursor_pos, cam_right, cam_dir):
    out = np.zeros(3, dtype=np.float32)
    load().orc_brush_get_vector(_p(m), _p(_f3(v)), _p(_f3(v_pos)), _p(_f3(md_pos)), _p(_f3(cursor_pos)),
                                _p(_f3(cam_right)), _p(_f3(cam_dir)), _p(out))
    return out


def brush_mouse_down(params, grid, cursor_pos, falloff_dist):
    """Selected cell indices (ascending) and their vectors at mouse-down."""
    g = _grid(grid)
    idx = np.zeros(g.shape[0], dtype=np.uint32)
    vec = np.zeros((g.shape[0], 3), dtype=np.float32)
    n = load().orc_brush_mouse_down(_p(_params_buf(params)), _p(g), _p(_f3(cursor_pos)), float(falloff_dist),
                                    _p(idx), _p(vec), g.shape[0])
    return idx[:n].copy(), vec[:n].copy()


def brush_mouse_moved(params, grid, m, indices, md_vectors, md_pos, cursor_pos, cam_right=(1, 0, 0), cam_dir=(0, 0, 1)):
    """Returns the field after the edit (input untouched)."""
    g = _grid(grid).copy()
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    vec = np.ascontiguousarray(md_vectors, dtype=np.float32).reshape(-1, 3)
    load().orc_brush_mouse_moved(_p(_params_buf(params)), _p(g), _p(m), _p(idx), _p(vec), idx.shape[0],
                                 _p(_f3(md_pos)), _p(_f3(cursor_pos)), _p(_f3(cam_right)), _p(_f3(cam_dir)))
    return g
