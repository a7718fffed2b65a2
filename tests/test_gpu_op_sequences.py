"""Random sequences of API calls on one handle, mirrored call by call on the
CPU oracle (tests/oracle_backend.py), with the two states compared after every
call.  The single-feature parity tests exercise each entry point from a fresh
handle; these sequences exercise what a handle carries from one call to the
next: ping-pong side, tile order and its caller-index maps, a pending tile
histogram, brush selection, fused-step requests, capacity changes on upload.

Bit-exact throughout: streaming steps, all-pairs steps with exponent 0
(summation in index order), field edits and brush strokes are all computed
with the same correctly rounded operations on both sides."""
import numpy as np
import pytest

import particles_amd as pa
from conftest import assert_particles_bit_equal
from oracle_backend import OracleCompute

pytestmark = pytest.mark.gpu

DTS = (0.0, 1.0 / 144, 1.0 / 60, 1.0 / 60, 1.0 / 30, 0.05)
SIZES = (0, 1, 3, 100, 4095, 4096, 5003, 8191, 9000)
FIELDS = ((96, 96, 1), (24, 20, 18), (128, 40, 1), (10, 10, 10))


def random_field(rng, dims):
    n_cells = dims[0] * dims[1] * dims[2]
    g = np.ones((n_cells, 4), dtype=np.float32)
    g[:, :3] = rng.uniform(-3, 3, size=(n_cells, 3)).astype(np.float32)
    return g


def brush_record(rng, radius):
    mode = pa.EditMode(mode=str(rng.choice(["centered", "shift", "rotate"])),
                       relative=bool(rng.integers(2)),
                       falloff=str(rng.choice(["abrupt", "linear", "inverse_distance"])),
                       falloff_dist=float(rng.uniform(0.3, 0.4 * radius)), strength=float(rng.uniform(-2, 2)))
    return mode


class Pair:
    """The device handle and its model, driven together."""

    def __init__(self, oracle, rng):
        self.O, self.rng = oracle, rng
        self.sp = pa.SimParams()
        self.dims = FIELDS[int(rng.integers(len(FIELDS)))]
        self.sp.force_grid_dimensions = self.dims
        self.sp.delta_t = np.float32(1.0 / 60)
        self.sp.masses = rng.uniform(0.2, 4.0, size=5).astype(np.float32)
        self.radius = 10.0
        g = random_field(rng, self.dims)
        p = self.particles(int(rng.choice([4096, 5003, 9000])))
        self.dev = pa.Compute.new(p, g)
        self.model = OracleCompute.new(p, g)
        self.push_params()
        self.log = []

    def particles(self, n):
        return self.O.synth_particles(n, seed=int(self.rng.integers(1 << 30)), pos_extent=2.0 * self.radius,
                                      vel_extent=float(self.rng.choice([0.0, 2.0, 10.0, 60.0])))

    def push_params(self):
        self.dev.update_sim_params(self.sp)
        self.model.update_sim_params(self.sp)

    def check(self, what):
        got, want = self.dev.read_particles(), self.model.read_particles()
        assert_particles_bit_equal(got, want, f"after {what}; calls so far: {self.log}")
        assert np.array_equal(self.dev.read_force_grid(), self.model.read_force_grid()), f"field after {what}: {self.log}"

    # -- the calls ----------------------------------------------------------
    def op_step(self):
        self.dev.compute(); self.model.compute()

    def op_step_dt(self):
        dt = float(self.rng.choice(DTS))
        self.sp.delta_t = np.float32(dt)
        self.dev.step(dt); self.model.step(np.float32(dt))

    def op_step_n(self):
        k = int(self.rng.integers(1, 21))
        self.dev.step_n(k); self.model.step_n(k)

    def op_interval(self):
        self.dev.set_reorder_interval(int(self.rng.choice([-1, 0, 1, 2, 3, 4, 7])))

    def op_reorder_now(self):
        self.dev.reorder_now()

    def op_launch_shape(self):
        """Block size and residency cap of the plain step (what large sets pick automatically), the packed
        field copy on or off, plain or non-temporal loads: results never depend on them."""
        block_sel, per_cu = int(self.rng.integers(4)), int(self.rng.choice([0, 1, 2, 5, 12, 15]))
        variant = (block_sel << 28) | (per_cu << 24)
        if self.rng.random() < 0.3:
            variant |= 0x400000
        if self.rng.random() < 0.3:
            variant |= 0x1
        self.dev.set_tuning(0, variant)

    def op_params(self):
        self.sp.masses = self.rng.uniform(0.2, 4.0, size=5).astype(np.float32)
        self.sp.max_velocity = np.float32(self.rng.choice([0.5, 20.0, 100.0]))
        self.sp.delta_t = np.float32(self.rng.choice(DTS))
        self.push_params()

    def op_pair_steps(self):
        if self.model.num_particles > 6000:
            return
        self.sp.cut_off_distance = np.float32(self.rng.uniform(0.3, 1.5))
        self.sp.distance_exponent = np.float32(0.0)
        self.sp.polys[:] = self.rng.uniform(-2, 2, size=(25, 8)).astype(np.float32)
        self.push_params()
        self.dev.set_pair_mode(1)          # the all-pairs sweep: summation in index order, bit-exact
        for _ in range(int(self.rng.integers(1, 3))):
            self.dev.compute(); self.model.compute()
        if self.model.num_particles:
            assert self.dev.last_step_kernel == pa.KERNEL_PAIRS
        self.sp.polys[:] = 0.0
        self.push_params()

    def op_patch(self):
        n_cells = self.model.g.shape[0]
        first = int(self.rng.integers(n_cells))
        count = int(self.rng.integers(1, min(500, n_cells - first) + 1))
        cells = random_field(self.rng, (count, 1, 1))
        self.dev.patch_force_grid(first, cells); self.model.patch_force_grid(first, cells)

    def op_patch_indexed(self):
        n_cells = self.model.g.shape[0]
        idx = self.rng.choice(n_cells, size=int(self.rng.integers(1, 300)), replace=False).astype(np.uint32)
        cells = random_field(self.rng, (idx.shape[0], 1, 1))
        self.dev.patch_force_grid_indexed(idx, cells); self.model.patch_force_grid_indexed(idx, cells)

    def op_field_replace(self):
        if self.rng.random() < 0.5:      # other dimensions: parameters first, then the matching field
            self.dims = FIELDS[int(self.rng.integers(len(FIELDS)))]
            self.sp.force_grid_dimensions = self.dims
        g = random_field(self.rng, self.dims)
        self.dev.update_force_grid(g); self.model.update_force_grid(g)
        self.push_params()

    def op_brush(self):
        mode = brush_record(self.rng, self.radius)
        down = self.rng.uniform(-0.8 * self.radius, 0.8 * self.radius, size=3)
        rec = mode.record(down)
        self.dev.brush_mouse_down(rec); self.model.brush_mouse_down(rec)
        for _ in range(int(self.rng.integers(1, 4))):
            rec = mode.record(down + self.rng.uniform(-1, 1, size=3))
            self.dev.brush_mouse_moved(rec); self.model.brush_mouse_moved(rec)
            if self.rng.random() < 0.5:   # frames go on while the button is held
                self.dev.compute(); self.model.compute()
        self.dev.brush_mouse_up(); self.model.brush_mouse_up()

    def op_upload(self):
        p = self.particles(int(self.rng.choice(SIZES)))
        if self.rng.random() < 0.5:
            self.dev.upload_particles(p)
        else:
            self.dev.upload_particles_soa(p["pos"][:, 0], p["pos"][:, 1], p["pos"][:, 2],
                                          p["vel"][:, 0], p["vel"][:, 1], p["vel"][:, 2], p["ty"].astype(np.uint8))
        self.model.upload_particles(p)

    def op_read_ranges(self):
        n = self.model.num_particles
        if n == 0:
            return
        first = int(self.rng.integers(n))
        count = int(self.rng.integers(1, n - first + 1))
        want = self.model.read_particles(first, count)
        assert_particles_bit_equal(self.dev.read_particles(first, count), want, f"range read: {self.log}")
        soa = self.dev.read_soa(first, count)
        for k, (f, c) in {"x": ("pos", 0), "y": ("pos", 1), "z": ("pos", 2), "vx": ("vel", 0), "vy": ("vel", 1), "vz": ("vel", 2)}.items():
            assert np.array_equal(soa[k].view(np.uint32), np.ascontiguousarray(want[f][:, c]).view(np.uint32)), (k, self.log)
        assert np.array_equal(soa["ty"], want["ty"].astype(np.uint8))
        cells = self.O.cell_indices(self.sp, self.model.read_particles(), self.model.g.shape[0])
        assert np.array_equal(self.dev.debug_cell_indices(first, count), cells[first:first + count]), self.log

    def op_pack(self):
        import torch
        n = self.model.num_particles
        if n == 0:
            return
        want = self.model.read_particles().view(np.uint8).reshape(n, 48)
        buf = torch.zeros((n, 48), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        self.dev.pack_particles_device(buf.data_ptr(), 0, n)
        self.dev.synchronize()
        assert buf.cpu().numpy().tobytes() == want.tobytes(), self.log
        buf.zero_()
        torch.cuda.synchronize()
        self.dev.pack_particles_device_unordered(buf.data_ptr())
        self.dev.synchronize()
        key = lambda a: a[np.lexsort(a.T[::-1])]
        assert key(buf.cpu().numpy()).tobytes() == key(want).tobytes(), self.log

    def op_diag(self):
        self.dev.enable_step_diagnostics(bool(self.rng.integers(2)))
        d, want = self.dev.diagnostics(), self.model.diagnostics()
        assert d["count"] == want["count"]
        assert abs(d["kinetic_energy"] - want["kinetic_energy"]) <= 1e-12 * max(want["kinetic_energy"], 1e-300), self.log
        if want["count"]:
            assert np.array_equal(d["pos_max"], want["pos_max"]) and np.array_equal(d["pos_min"], want["pos_min"]), self.log

    OPS = (("step", 10), ("step_dt", 3), ("step_n", 4), ("interval", 4), ("reorder_now", 2), ("launch_shape", 2), ("params", 2),
           ("pair_steps", 1), ("patch", 2), ("patch_indexed", 2), ("field_replace", 1), ("brush", 2), ("upload", 2),
           ("read_ranges", 2), ("pack", 1), ("diag", 2))

    def run(self, n_calls):
        names = [n for n, _ in self.OPS]
        weights = np.array([w for _, w in self.OPS], dtype=np.float64)
        weights /= weights.sum()
        self.check("construction")
        for _ in range(n_calls):
            name = str(self.rng.choice(names, p=weights))
            self.log.append(name)
            getattr(self, "op_" + name)()
            self.check(name)

    def close(self):
        self.dev.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_call_sequences_match_the_model(oracle, seed):
    rng = np.random.default_rng(77000 + seed)
    pair = Pair(oracle, rng)
    try:
        if seed % 2 == 0:
            pair.dev.set_reorder_interval(int(rng.choice([1, 2, 3])))   # tile order from the start
        pair.run(80)
        assert pair.dev.step_count > 0
    finally:
        pair.close()
