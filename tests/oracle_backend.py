"""Test-only stand-in for particles_amd.Compute backed by the CPU oracle.

Lets the host-side sharding logic (index ranges, field broadcast, diagnostics
all-reduce) run under gloo on a machine without a GPU.  Lives under tests/
on purpose: the product package never falls back to the CPU."""
import numpy as np

import particles_amd as pa
from oracle import oracle as O


class OracleCompute:
    def __init__(self, particles, force_grid, device=0):
        self.p = np.ascontiguousarray(particles).view(O.PARTICLE_DTYPE).reshape(-1).copy()
        self.g = np.ascontiguousarray(force_grid, dtype=np.float32).copy()
        self.sp = pa.SimParams()          # src/compute.rs:33: starts as SimParams::new()
        self.stepped = False

    @classmethod
    def new(cls, particles, force_grid, device=0):
        return cls(particles, force_grid, device)

    @property
    def num_particles(self):
        return self.p.shape[0]

    def upload_particles(self, particles):
        self.p = np.ascontiguousarray(particles).view(O.PARTICLE_DTYPE).reshape(-1).copy()

    def generate_particles(self, n, seed, first_global_index=0, pos_extent=2.0, vel_extent=10.0):
        self.p = O.synth_particles(n, seed, first_global_index, pos_extent, vel_extent)

    def update_force_grid(self, force_grid):
        self.g = np.ascontiguousarray(force_grid, dtype=np.float32).copy()

    def patch_force_grid(self, first_cell, cells):
        cells = np.ascontiguousarray(cells, dtype=np.float32)
        if first_cell + cells.shape[0] > self.g.shape[0]:
            raise pa.PclError(-4, "range outside [0, n)")
        self.g[first_cell:first_cell + cells.shape[0]] = cells

    def patch_force_grid_indexed(self, cell_indices, cells):
        idx = np.ascontiguousarray(cell_indices, dtype=np.uint32)
        if idx.size and idx.max() >= self.g.shape[0]:
            raise pa.PclError(-4, "cell index outside the field")
        self.g[idx] = np.ascontiguousarray(cells, dtype=np.float32)

    def update_sim_params(self, sp):
        self.sp = sp.copy()

    def compute(self):
        if self.p.shape[0]:
            self.p = O.step(self.sp, self.p, self.g, mode=O.MODE_SKIP_INERT)

    def step(self, dt=None):
        if dt is not None:
            self.sp.delta_t = dt
        self.compute()

    def step_n(self, n_steps):
        for _ in range(int(n_steps)):
            self.compute()

    def pack_pair_sources(self, capacity_records):
        """(x, y, z, type bits) per particle, padded with type 0xFFFFFFFF records."""
        rec = np.zeros((capacity_records, 4), dtype=np.float32)
        rec[:, 3].view(np.uint32)[:] = 0xFFFFFFFF
        n = self.p.shape[0]
        rec[:n, :3] = self.p["pos"][:, :3]
        rec[:n, 3].view(np.uint32)[:] = self.p["ty"]
        return rec

    def step_pairs_gathered(self, records, own_block_start):
        """Update this range against the gathered records of all ranges;
        ``own_block_start`` is the record index where this range's block begins."""
        records = np.ascontiguousarray(records, dtype=np.float32)
        ty = records[:, 3].view(np.uint32)
        keep = ty < 5
        first = int(np.count_nonzero(keep[:own_block_start]))
        n = self.p.shape[0]
        full = np.zeros(int(np.count_nonzero(keep)), dtype=O.PARTICLE_DTYPE)
        full["pos"][:, :3] = records[keep, :3]
        full["pos"][:, 3] = 1.0
        full["ty"] = ty[keep]
        assert np.array_equal(full["pos"][first:first + n, :3], self.p["pos"][:, :3])
        full[first:first + n] = self.p
        if n:
            self.p = O.step_range(self.sp, full, self.g, first, n).copy()

    def enable_step_diagnostics(self, enable=True):
        pass

    def diagnostics(self):
        return O.diagnostics(self.sp, self.p)

    def read_particles(self, first=0, count=None):
        count = self.p.shape[0] - first if count is None else count
        return self.p[first:first + count].view(pa.PARTICLE_DTYPE).copy()

    def synchronize(self):
        pass

    def close(self):
        pass

    # -- field brush: the oracle's restatement of src/cursor.rs ------------
    def brush_mouse_down(self, record):
        r = np.ascontiguousarray(record).reshape(1)
        self._md_pos = r["cursor_pos"][0].copy()
        self._sel, self._md_vec = O.brush_mouse_down(self.sp, self.g, self._md_pos, float(r["falloff_dist"][0]))
        self._brush_down = True

    def brush_mouse_moved(self, record):
        if not getattr(self, "_brush_down", False):
            return
        r = np.ascontiguousarray(record).reshape(1)
        m = np.zeros(1, dtype=O.EDIT_MODE_DTYPE)
        for k in ("mode", "relative", "falloff", "falloff_dist", "strength"):
            m[k] = r[k]
        self.g = O.brush_mouse_moved(self.sp, self.g, m, self._sel, self._md_vec, self._md_pos,
                                     r["cursor_pos"][0], r["cam_right"][0], r["cam_dir"][0])

    def brush_mouse_up(self):
        self._brush_down = False

    def read_force_grid(self, first_cell=0, count=None):
        count = self.g.shape[0] - first_cell if count is None else count
        return self.g[first_cell:first_cell + count].copy()
