"""Host-side mirror of the reference's ``Compute`` (src/compute.rs:11-228)
over ALL GPUs of a node driven by one process: ``pcl_group_*`` of the C ABI
(include/particles_hip.h), which shards the particles by index range and keeps
the replicas of the field equal with RCCL inside the library."""
import ctypes as C

import numpy as np

from . import _lib
from .compute import _as_cells, _as_particles, _ptr
from .particle import PARTICLE_DTYPE
from .sim_params import SIM_PARAMS_DTYPE, SimParams


class Group:
    def __init__(self, particles, force_grid, devices=None):
        self._lib = _lib.load()
        if devices is None:
            devices = list(range(max(1, int(self._lib.pcl_device_count()))))
        self.devices = [int(d) for d in devices]
        dev = (C.c_int * len(self.devices))(*self.devices)
        p, g = _as_particles(particles), _as_cells(force_grid)
        self._g = C.c_void_p()
        _lib.check(self._lib.pcl_group_create(dev, len(self.devices), _ptr(p), p.shape[0], _ptr(g), g.shape[0],
                                              C.byref(self._g)))

    @classmethod
    def new(cls, particles, force_grid, devices=None):
        """``Compute::new`` (src/compute.rs:25) over the given device ordinals (default: all)."""
        return cls(particles, force_grid, devices)

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            self._lib.pcl_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def size(self):
        return int(self._lib.pcl_group_size(self._g))

    @property
    def num_particles(self):
        return int(self._lib.pcl_group_num_particles(self._g))

    def handle(self, i):
        """Raw ``pcl_compute*`` of GPU i (owned by the group)."""
        return C.c_void_p(self._lib.pcl_group_handle(self._g, int(i)))

    def upload_particles(self, particles):
        p = _as_particles(particles)
        _lib.check(self._lib.pcl_group_upload_particles(self._g, _ptr(p), p.shape[0]))

    def generate_particles(self, n_total, seed, pos_extent=2.0, vel_extent=10.0):
        _lib.check(self._lib.pcl_group_generate_particles(self._g, int(n_total), int(seed), float(pos_extent), float(vel_extent)))

    def update_force_grid(self, force_grid):
        g = _as_cells(force_grid)
        _lib.check(self._lib.pcl_group_update_force_grid(self._g, _ptr(g), g.shape[0]))

    def patch_force_grid(self, first_cell, cells):
        g = _as_cells(cells)
        _lib.check(self._lib.pcl_group_patch_force_grid(self._g, int(first_cell), _ptr(g), g.shape[0]))

    def patch_force_grid_indexed(self, cell_indices, cells):
        idx = np.ascontiguousarray(cell_indices, dtype=np.uint32).reshape(-1)
        g = _as_cells(cells)
        if idx.shape[0] != g.shape[0]:
            raise ValueError("cell_indices and cells must have the same length")
        _lib.check(self._lib.pcl_group_patch_force_grid_indexed(self._g, _ptr(idx), _ptr(g), idx.shape[0]))

    def update_sim_params(self, sim_params):
        raw = sim_params.raw if isinstance(sim_params, SimParams) else np.asarray(sim_params)
        if raw.dtype != SIM_PARAMS_DTYPE:
            raise TypeError("sim_params must be a SimParams or a SIM_PARAMS_DTYPE record (912 bytes)")
        raw = np.ascontiguousarray(raw)
        _lib.check(self._lib.pcl_group_update_sim_params(self._g, raw.ctypes.data_as(C.c_void_p)))

    def enable_step_diagnostics(self, enable=True):
        _lib.check(self._lib.pcl_group_enable_step_diagnostics(self._g, int(bool(enable))))

    def compute(self):
        """One update of every range (src/compute.rs:221-227)."""
        _lib.check(self._lib.pcl_group_step(self._g))

    def brush_mouse_down(self, record):
        r = np.ascontiguousarray(record)
        _lib.check(self._lib.pcl_group_brush_mouse_down(self._g, _ptr(r)))

    def brush_mouse_moved(self, record):
        r = np.ascontiguousarray(record)
        _lib.check(self._lib.pcl_group_brush_mouse_moved(self._g, _ptr(r)))

    def brush_mouse_up(self):
        _lib.check(self._lib.pcl_group_brush_mouse_up(self._g))

    def diagnostics(self):
        d = _lib.Diagnostics()
        _lib.check(self._lib.pcl_group_diagnostics_read(self._g, C.byref(d)))
        return {"count": int(d.count), "kinetic_energy": float(d.kinetic_energy),
                "pos_max": np.array(d.pos_max[:], dtype=np.float32),
                "pos_min": -np.array(d.neg_pos_min[:], dtype=np.float32)}

    def read_particles(self, first=0, count=None):
        count = self.num_particles - first if count is None else int(count)
        out = np.zeros(count, dtype=PARTICLE_DTYPE)
        _lib.check(self._lib.pcl_group_read_particles(self._g, _ptr(out), int(first), count))
        return out

    def read_force_grid(self, replica=0, first_cell=0, count=None):
        h = self.handle(replica)
        count = int(self._lib.pcl_num_cells(h)) - first_cell if count is None else int(count)
        out = np.zeros((count, 4), dtype=np.float32)
        _lib.check(self._lib.pcl_group_read_force_grid(self._g, int(replica), _ptr(out), int(first_cell), count))
        return out

    def synchronize(self):
        _lib.check(self._lib.pcl_group_synchronize(self._g))
