"""Host-side mirror of the reference's ``Compute`` (src/compute.rs:11-228)
over the gfx950 C ABI (include/particles_hip.h).

Method names and argument meaning follow the reference: ``Compute.new``
(compute.rs:25), ``upload_particles`` (:182), ``update_force_grid`` (:198),
``update_sim_params`` (:206), ``compute`` (:221), ``num_particles`` (:19).
The wgpu ``device``/``ComputePass`` arguments become the HIP device ordinal
given at construction.  The reference has no error returns (wgpu validation
failures panic); here invalid arguments raise ``PclError``.
"""
import ctypes as C

import numpy as np

from . import _lib
from .particle import PARTICLE_DTYPE
from .cursor import BRUSH_INPUT_DTYPE
from .sim_params import SIM_PARAMS_DTYPE, SimParams


def _as_particles(particles):
    a = np.ascontiguousarray(particles)
    if a.dtype != PARTICLE_DTYPE:
        raise TypeError("particles must be an array of PARTICLE_DTYPE records (48 bytes each)")
    return a.reshape(-1)


def _as_cells(force_grid):
    a = np.ascontiguousarray(force_grid, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 4:
        raise TypeError("force_grid must have shape (n_cells, 4): [x, y, z, 1.0] per cell")
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Compute:
    """Owns the device copy of the particle state, field and parameters."""

    def __init__(self, particles, force_grid, device=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        p = _as_particles(particles)
        g = _as_cells(force_grid)
        _lib.check(self._lib.pcl_create(_ptr(p), p.shape[0], _ptr(g), g.shape[0], int(device), C.byref(self._h)))
        self.device = int(device)

    @classmethod
    def new(cls, particles, force_grid, device=0):
        """``Compute::new(device, particles, force_grid)`` (compute.rs:25):
        copies both inputs; parameters start as ``SimParams::new()``
        (delta_t = 0) until ``update_sim_params`` is called."""
        return cls(particles, force_grid, device)

    # -- lifetime -------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.pcl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the reference's surface ---------------------------------------
    @property
    def num_particles(self):
        return int(self._lib.pcl_num_particles(self._h))

    @property
    def num_cells(self):
        return int(self._lib.pcl_num_cells(self._h))

    def upload_particles(self, particles):
        p = _as_particles(particles)
        _lib.check(self._lib.pcl_upload_particles(self._h, _ptr(p), p.shape[0]))

    def update_force_grid(self, force_grid):
        g = _as_cells(force_grid)
        _lib.check(self._lib.pcl_update_force_grid(self._h, _ptr(g), g.shape[0]))

    def update_sim_params(self, sim_params):
        raw = sim_params.raw if isinstance(sim_params, SimParams) else np.asarray(sim_params)
        if raw.dtype != SIM_PARAMS_DTYPE:
            raise TypeError("sim_params must be a SimParams or a SIM_PARAMS_DTYPE record (912 bytes)")
        raw = np.ascontiguousarray(raw)
        _lib.check(self._lib.pcl_update_sim_params(self._h, raw.ctypes.data_as(C.c_void_p)))

    def compute(self):
        """One update src -> dst, then flip (compute.rs:221-227). Asynchronous."""
        _lib.check(self._lib.pcl_step(self._h))

    # -- north-star conveniences ---------------------------------------
    def step(self, dt=None):
        """``step(dt)``: set delta_t (when given) and run one update."""
        if dt is None:
            _lib.check(self._lib.pcl_step(self._h))
        else:
            _lib.check(self._lib.pcl_step_dt(self._h, float(dt)))

    # -- field brush (src/cursor.rs) --------------------------------------
    @staticmethod
    def _brush_record(record):
        r = np.ascontiguousarray(record)
        if r.dtype != BRUSH_INPUT_DTYPE or r.size != 1:
            raise TypeError("expected one BRUSH_INPUT_DTYPE record (EditMode.record)")
        return r

    def brush_mouse_down(self, record):
        """Cursor::mouse_down (src/cursor.rs:105-116) on the device field."""
        r = self._brush_record(record)
        _lib.check(self._lib.pcl_brush_mouse_down(self._h, _ptr(r)))

    def brush_mouse_moved(self, record):
        """Cursor::mouse_moved (src/cursor.rs:79-103) on the device field."""
        r = self._brush_record(record)
        _lib.check(self._lib.pcl_brush_mouse_moved(self._h, _ptr(r)))

    def brush_mouse_down_device(self, dev_record):
        _lib.check(self._lib.pcl_brush_mouse_down_device(self._h, C.c_void_p(int(dev_record))))

    def brush_mouse_moved_device(self, dev_record):
        _lib.check(self._lib.pcl_brush_mouse_moved_device(self._h, C.c_void_p(int(dev_record))))

    def brush_mouse_up(self):
        _lib.check(self._lib.pcl_brush_mouse_up(self._h))

    def brush_selection(self):
        """Selected cell indices, ascending (Cursor::modify_vector_indices)."""
        n = C.c_size_t(0)
        _lib.check(self._lib.pcl_brush_selection(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            _lib.check(self._lib.pcl_brush_selection(self._h, _ptr(out), out.shape[0], C.byref(n)))
        return out

    def read_force_grid(self, first_cell=0, count=None):
        """Cells of the device field as (count, 4) float32."""
        count = self.num_cells - first_cell if count is None else int(count)
        out = np.zeros((count, 4), dtype=np.float32)
        _lib.check(self._lib.pcl_read_force_grid(self._h, _ptr(out), int(first_cell), count))
        return out

    def pack_pair_sources_device(self, dev_out, capacity_records):
        """Latest (x, y, z, type) records in caller order, padded to
        ``capacity_records``, into device memory (for the all-gather of the
        sharded pair step)."""
        _lib.check(self._lib.pcl_pack_pair_sources_device(self._h, C.c_void_p(int(dev_out)), int(capacity_records)))

    def step_pairs_gathered(self, dev_sources, n_sources):
        """One update whose pair partners are the gathered records of all ranges."""
        _lib.check(self._lib.pcl_step_pairs_gathered(self._h, C.c_void_p(int(dev_sources)), int(n_sources)))

    # -- several GPUs, one process per GPU: RCCL inside the library ------------
    @staticmethod
    def comm_unique_id():
        """128 bytes that rank 0 creates and the host carries to every rank."""
        uid = (C.c_char * _lib.UNIQUE_ID_BYTES)()
        _lib.check(_lib.load().pcl_comm_unique_id(uid))
        return bytes(uid)

    def comm_attach(self, unique_id, rank, world):
        """Join the communicator (ncclCommInitRank on this handle's GPU); collective."""
        if len(unique_id) != _lib.UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        _lib.check(self._lib.pcl_comm_attach(self._h, C.c_char_p(bytes(unique_id)), int(rank), int(world)))

    def comm_detach(self):
        _lib.check(self._lib.pcl_comm_detach(self._h))

    def comm_update_force_grid(self, force_grid, n_cells, root=0):
        """Whole-field replace on every rank from the root's cells (others pass ``None``)."""
        g = _as_cells(force_grid) if force_grid is not None else None
        _lib.check(self._lib.pcl_comm_update_force_grid(self._h, _ptr(g), int(n_cells), int(root)))

    def comm_patch_force_grid(self, first_cell, cells, count, root=0):
        g = _as_cells(cells) if cells is not None else None
        _lib.check(self._lib.pcl_comm_patch_force_grid(self._h, int(first_cell), _ptr(g), int(count), int(root)))

    def comm_patch_force_grid_indexed(self, cell_indices, cells, count, root=0):
        idx = np.ascontiguousarray(cell_indices, dtype=np.uint32).reshape(-1) if cell_indices is not None else None
        g = _as_cells(cells) if cells is not None else None
        _lib.check(self._lib.pcl_comm_patch_force_grid_indexed(self._h, _ptr(idx), _ptr(g), int(count), int(root)))

    def comm_brush_mouse_down(self, record, root=0):
        r = self._brush_record(record) if record is not None else None
        _lib.check(self._lib.pcl_comm_brush_mouse_down(self._h, _ptr(r), int(root)))

    def comm_brush_mouse_moved(self, record, root=0):
        r = self._brush_record(record) if record is not None else None
        _lib.check(self._lib.pcl_comm_brush_mouse_moved(self._h, _ptr(r), int(root)))

    def comm_step(self):
        """One update of this rank's range; all-gather first when the pair term is
        active, all-reduce of the diagnostics when they are enabled."""
        _lib.check(self._lib.pcl_comm_step(self._h))

    def comm_diagnostics(self):
        """Diagnostics of all ranges (identical on every rank); collective."""
        d = _lib.Diagnostics()
        _lib.check(self._lib.pcl_comm_diagnostics_read(self._h, C.byref(d)))
        return {"count": int(d.count), "kinetic_energy": float(d.kinetic_energy),
                "pos_max": np.array(d.pos_max[:], dtype=np.float32),
                "pos_min": -np.array(d.neg_pos_min[:], dtype=np.float32)}

    def step_n(self, n_steps):
        """``n_steps`` updates with the result of ``n_steps`` calls of
        ``compute()``; consecutive streaming steps are fused into single
        launches (the state stays in registers between them)."""
        _lib.check(self._lib.pcl_step_n(self._h, int(n_steps)))

    def patch_force_grid(self, first_cell, cells):
        g = _as_cells(cells)
        _lib.check(self._lib.pcl_patch_force_grid(self._h, int(first_cell), _ptr(g), g.shape[0]))

    def patch_force_grid_indexed(self, cell_indices, cells):
        """field[cell_indices[i]] = cells[i] (brush edit of a list of cells)."""
        idx = np.ascontiguousarray(cell_indices, dtype=np.uint32).reshape(-1)
        g = _as_cells(cells)
        if idx.shape[0] != g.shape[0]:
            raise ValueError("cell_indices and cells must have the same length")
        _lib.check(self._lib.pcl_patch_force_grid_indexed(self._h, _ptr(idx), _ptr(g), idx.shape[0]))

    def patch_force_grid_indexed_device(self, dev_indices, dev_cells, count):
        _lib.check(self._lib.pcl_patch_force_grid_indexed_device(
            self._h, C.c_void_p(int(dev_indices)), C.c_void_p(int(dev_cells)), int(count)))

    def update_force_grid_device(self, dev_ptr, n_cells):
        _lib.check(self._lib.pcl_update_force_grid_device(self._h, C.c_void_p(int(dev_ptr)), int(n_cells)))

    def patch_force_grid_device(self, first_cell, dev_ptr, count):
        _lib.check(self._lib.pcl_patch_force_grid_device(self._h, int(first_cell), C.c_void_p(int(dev_ptr)), int(count)))

    def upload_particles_soa(self, x, y, z, vx, vy, vz, ty):
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (x, y, z, vx, vy, vz)]
        t = np.ascontiguousarray(ty, dtype=np.uint8)
        n = t.shape[0]
        if any(a.shape != (n,) for a in arrs):
            raise ValueError("all SoA arrays must have the same length")
        _lib.check(self._lib.pcl_upload_particles_soa(self._h, n, *[_ptr(a) for a in arrs], _ptr(t)))

    def generate_particles(self, n, seed, first_global_index=0, pos_extent=2.0, vel_extent=10.0):
        _lib.check(self._lib.pcl_generate_particles(self._h, int(n), int(seed), int(first_global_index),
                                                    float(pos_extent), float(vel_extent)))

    def read_particles(self, first=0, count=None):
        """AoS records of the latest state (pos.w = vel.w = 1.0)."""
        if count is None:
            count = self.num_particles - first
        out = np.zeros(int(count), dtype=PARTICLE_DTYPE)
        _lib.check(self._lib.pcl_read_particles(self._h, _ptr(out), int(first), int(count)))
        return out

    def read_soa(self, first=0, count=None):
        if count is None:
            count = self.num_particles - first
        count = int(count)
        arrs = [np.zeros(count, dtype=np.float32) for _ in range(6)]
        ty = np.zeros(count, dtype=np.uint8)
        _lib.check(self._lib.pcl_read_soa(self._h, int(first), count, *[_ptr(a) for a in arrs], _ptr(ty)))
        return dict(zip(("x", "y", "z", "vx", "vy", "vz"), arrs), ty=ty)

    def pack_particles_device(self, dev_ptr, first=0, count=None):
        if count is None:
            count = self.num_particles - first
        _lib.check(self._lib.pcl_pack_particles_device(self._h, C.c_void_p(int(dev_ptr)), int(first), int(count)))

    def pack_particles_device_unordered(self, dev_ptr):
        """All records in current memory order (every particle once): what an
        instance stream needs, as one streaming pass."""
        _lib.check(self._lib.pcl_pack_particles_device_unordered(self._h, C.c_void_p(int(dev_ptr))))

    def diagnostics(self):
        d = _lib.Diagnostics()
        _lib.check(self._lib.pcl_diagnostics_read(self._h, C.byref(d)))
        return {
            "count": int(d.count),
            "kinetic_energy": float(d.kinetic_energy),
            "pos_max": np.array(d.pos_max[:], dtype=np.float32),
            "pos_min": -np.array(d.neg_pos_min[:], dtype=np.float32),
        }

    def diagnostics_device(self, dev_ptr):
        _lib.check(self._lib.pcl_diagnostics_device(self._h, C.c_void_p(int(dev_ptr))))

    def diagnostics_device_noalloc(self, dev_ptr):
        """``diagnostics_device`` for callers that must not allocate (at most 2^32 particles)."""
        _lib.check(self._lib.pcl_diagnostics_device_noalloc(self._h, C.c_void_p(int(dev_ptr))))

    def enable_step_diagnostics(self, enable=True):
        _lib.check(self._lib.pcl_enable_step_diagnostics(self._h, int(bool(enable))))

    def debug_cell_indices(self, first=0, count=None):
        if count is None:
            count = self.num_particles - first
        out = np.zeros(int(count), dtype=np.uint32)
        _lib.check(self._lib.pcl_debug_cell_indices(self._h, _ptr(out), int(first), int(count)))
        return out

    def debug_pair_cells(self):
        """Cut-off cells of the binned pair term: (cell key per particle in caller order, bin_end per cell,
        cells per axis, reach)."""
        n = self.num_particles
        cells = np.zeros(n, dtype=np.uint32)
        bin_end = np.zeros(64 ** 3, dtype=np.uint32)
        per_axis, reach = C.c_uint32(), C.c_uint32()
        _lib.check(self._lib.pcl_debug_pair_cells(self._h, _ptr(cells), _ptr(bin_end), bin_end.shape[0],
                                                  C.byref(per_axis), C.byref(reach)))
        return cells, bin_end[: per_axis.value ** 3], per_axis.value, reach.value

    def pair_sweep_info(self):
        """Which sweep the last cell-binned pair step ran, and the occupied cut-off cells it was chosen from."""
        shared, cells = C.c_uint32(), C.c_uint32()
        _lib.check(self._lib.pcl_pair_sweep_info(self._h, C.byref(shared), C.byref(cells)))
        return {"shared_sweep": bool(shared.value), "cells_occupied": int(cells.value)}

    def debug_state_checksum(self, first_global_index=0):
        """(sum, xor) over all particles of a 64-bit mix of (global index, state bits, type): independent of
        the order of the particles in memory; the words of index ranges combine (sum mod 2^64, xor) to
        those of the whole set (see pcl_debug_state_checksum)."""
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.pcl_debug_state_checksum(self._h, int(first_global_index), C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def debug_slot_order(self):
        """Caller index of the particle in every memory slot (the identity until the set is reordered)."""
        out = np.zeros(self.num_particles, dtype=np.uint32)
        _lib.check(self._lib.pcl_debug_slot_order(self._h, _ptr(out), out.shape[0]))
        return out

    def set_pair_mode(self, mode):
        """0 automatic, 1 all-pairs sweep in index order (bit-exact), 2 cell-binned."""
        _lib.check(self._lib.pcl_set_pair_mode(self._h, int(mode)))

    def force_pair_kernel(self, enable=True):
        _lib.check(self._lib.pcl_force_pair_kernel(self._h, int(bool(enable))))

    @property
    def last_step_kernel(self):
        return int(self._lib.pcl_last_step_kernel(self._h))

    @property
    def step_count(self):
        return int(self._lib.pcl_step_count(self._h))

    def set_stream(self, hip_stream):
        """Run on the caller's stream: a ``hipStream_t`` handle value (e.g.
        ``torch.cuda.current_stream().cuda_stream``).  The value 0 is the
        device's null stream — PyTorch's default stream — and is adopted as
        such; ``None`` restores the handle's internal stream."""
        if hip_stream is None:
            _lib.check(self._lib.pcl_set_stream(self._h, None))
        elif int(hip_stream) == 0:
            _lib.check(self._lib.pcl_use_null_stream(self._h))
        else:
            _lib.check(self._lib.pcl_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def synchronize(self):
        _lib.check(self._lib.pcl_synchronize(self._h))

    def timer_start(self):
        _lib.check(self._lib.pcl_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        _lib.check(self._lib.pcl_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    def step_timing_begin(self, max_steps):
        """Bracket each of the next ``max_steps`` steps with a HIP event pair."""
        _lib.check(self._lib.pcl_step_timing_begin(self._h, int(max_steps)))
        self._timing_cap = int(max_steps)

    def step_timing_read(self):
        """(elapsed_ms float32[n], kinds int32[n]) of the recorded steps."""
        cap = getattr(self, "_timing_cap", 0)
        ms = np.zeros(cap, dtype=np.float32)
        kinds = np.zeros(cap, dtype=np.int32)
        n = C.c_size_t()
        _lib.check(self._lib.pcl_step_timing_read(self._h, _ptr(ms) if cap else None, _ptr(kinds) if cap else None,
                                                  cap, C.byref(n)))
        return ms[: n.value], kinds[: n.value]

    def launch_geometry(self):
        g, b, p = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.check(self._lib.pcl_launch_geometry(self._h, C.byref(g), C.byref(b), C.byref(p)))
        return {"grid_blocks": g.value, "block_threads": b.value, "particles_per_lane": p.value}

    @property
    def pci_bus_id(self):
        buf = C.create_string_buffer(64)
        _lib.check(self._lib.pcl_device_pci_bus_id(self._h, buf, 64))
        return buf.value.decode()

    @property
    def comm_count(self):
        """Ranks of the attached communicator as RCCL itself counts them (0: none attached)."""
        return int(self._lib.pcl_comm_count(self._h))

    def launch_residency(self):
        """Blocks per CU the plain streaming step's last launch asked for (0: no limit) and what the
        runtime's occupancy calculation grants it."""
        a, g = C.c_uint32(), C.c_uint32()
        _lib.check(self._lib.pcl_launch_residency(self._h, C.byref(a), C.byref(g)))
        return {"intended": a.value, "granted": g.value}

    def set_reorder_interval(self, interval):
        """-1 automatic, 0 never, k > 0: refresh the field-tile order every k steps."""
        _lib.check(self._lib.pcl_set_reorder_interval(self._h, int(interval)))

    @property
    def reorder_interval_in_use(self):
        """Steps between refreshes of the tile order currently in force (0: none)."""
        return int(self._lib.pcl_reorder_interval_in_use(self._h))

    def reorder_now(self):
        _lib.check(self._lib.pcl_reorder_now(self._h))

    @property
    def reorder_count(self):
        return int(self._lib.pcl_reorder_count(self._h))

    def set_tuning(self, grid_blocks=0, variant=0):
        _lib.check(self._lib.pcl_set_tuning(self._h, int(grid_blocks), int(variant)))
