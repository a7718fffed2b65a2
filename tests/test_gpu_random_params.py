"""Randomised parity sweep over the parameter domain the reference's GUI
offers (src/gui.rs:130-318: max_velocity 0.1..100, distance_exponent -5..5,
bounding radius 0.25..5 plus the default 10, masses 0.01..10, cut_off 0.1..5,
force-curve samples -10..10) and over field shapes, time steps and particle
counts.  Streaming path and pair path with exponent 0: bit-exact; pair path
with powf: 1e-5 relative."""
import numpy as np
import pytest

import particles_amd as pa
from conftest import assert_particles_bit_equal

pytestmark = pytest.mark.gpu


def random_case(rng, pairs):
    sp = pa.SimParams()
    dims = tuple(int(v) for v in rng.choice([1, 2, 3, 7, 10, 16, 33, 64], size=3))
    sp.force_grid_dimensions = dims
    sp.delta_t = np.float32(rng.choice([0.0, 1e-4, 1.0 / 144, 1.0 / 60, 0.05, 0.19]))
    sp.max_velocity = np.float32(rng.uniform(0.1, 100.0))
    sp.bounding_volume_radius = np.float32(rng.choice([0.25, 0.5, 1.0, 2.5, 5.0, 10.0]))
    sp.masses = rng.uniform(0.01, 10.0, size=5).astype(np.float32)
    n_cells = dims[0] * dims[1] * dims[2]
    g = np.ones((n_cells, 4), dtype=np.float32)
    g[:, :3] = rng.uniform(-3, 3, size=(n_cells, 3)).astype(np.float32)
    if pairs:
        sp.cut_off_distance = np.float32(rng.uniform(0.1, 5.0))
        sp.polys[:] = rng.uniform(-10, 10, size=(25, 8)).astype(np.float32) * (rng.random((25, 8)) < 0.5)
        sp.polys[0, 0] = 1.0   # never all-zero
    return sp, g


@pytest.mark.parametrize("seed", range(24))
def test_stream_path_random_parameters(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    sp, g = random_case(rng, pairs=False)
    n = int(rng.choice([1, 6, 100, 1023, 4097, 20000]))
    r = float(sp.bounding_volume_radius)
    p = oracle.synth_particles(n, seed=seed, pos_extent=2.0 * r, vel_extent=float(rng.choice([0.0, 1.0, 10.0, 400.0])))
    steps = 6
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        if n >= 4096:
            c.set_reorder_interval(int(rng.integers(1, 4)))
        assert np.array_equal(c.debug_cell_indices(), oracle.cell_indices(sp, p, g.shape[0]))
        for _ in range(steps):
            c.compute()
        want = oracle.run(sp, p, g, steps, mode=oracle.MODE_SKIP_INERT)
        assert_particles_bit_equal(c.read_particles(), want, f"seed {seed}")
        assert np.array_equal(c.debug_cell_indices(), oracle.cell_indices(sp, want, g.shape[0]))


@pytest.mark.parametrize("seed", range(12))
def test_pair_path_random_parameters(oracle, seed):
    rng = np.random.default_rng(5000 + seed)
    sp, g = random_case(rng, pairs=True)
    exact = seed % 2 == 0
    sp.distance_exponent = np.float32(0.0 if exact else rng.uniform(-5.0, 5.0))
    n = int(rng.choice([2, 65, 700, 1500]))
    r = float(sp.bounding_volume_radius)
    p = oracle.synth_particles(n, seed=seed, pos_extent=min(2.0 * r, 6.0), vel_extent=5.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.compute()
        assert c.last_step_kernel == pa.KERNEL_PAIRS
        got = c.read_particles()
    want = oracle.run(sp, p, g, 1)
    if exact:
        assert_particles_bit_equal(got, want, f"pairs seed {seed}")
    else:
        two_r = 2.0 * r
        for f in ("pos", "vel"):
            a, b = got[f][:, :3].astype(np.float64), want[f][:, :3].astype(np.float64)
            d = np.abs(a - b)
            if f == "pos":
                d = np.minimum(d, np.abs(two_r - d))
            assert np.all(d <= 1e-5 * np.maximum(np.abs(a), np.abs(b)) + 1e-6), f"pairs seed {seed} {f}"


@pytest.mark.parametrize("case", ["denormal_field", "denormal_velocity", "tiny_dt", "tiny_radius", "huge_radius",
                                  "zero_speed_cap", "negative_zero_velocity", "extreme_masses"])
def test_stream_path_extreme_but_finite_values(oracle, case):
    """Denormals, signed zeros and extreme magnitudes: the device keeps fp32
    denormals and IEEE division, so these match the CPU bit for bit too."""
    rng = np.random.default_rng(77)
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.force_grid_dimensions = (8, 5, 3)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = np.ones((120, 4), dtype=np.float32)
    g[:, :3] = rng.uniform(-3, 3, size=(120, 3)).astype(np.float32)
    vel_extent = 10.0
    if case == "denormal_field":
        g[:, :3] = (rng.uniform(-1, 1, size=(120, 3)) * 1e-39).astype(np.float32)
        vel_extent = 0.0
    elif case == "denormal_velocity":
        vel_extent = 1e-41
        g[:, :3] = 0.0
    elif case == "tiny_dt":
        sp.delta_t = np.float32(1e-30)
    elif case == "tiny_radius":
        sp.bounding_volume_radius = np.float32(1e-3)
    elif case == "huge_radius":
        sp.bounding_volume_radius = np.float32(1e6)
    elif case == "zero_speed_cap":
        sp.max_velocity = np.float32(0.0)
    elif case == "extreme_masses":
        sp.masses = (1e-6, 1e6, 1e-3, 1e3, 1.0)
    r = float(sp.bounding_volume_radius)
    p = oracle.synth_particles(3001, seed=5, pos_extent=2.0 * r, vel_extent=vel_extent)
    if case == "negative_zero_velocity":
        p["vel"][:, :3] = np.float32(-0.0)
        g[::2, :3] = 0.0                      # half the cells exert no force: -0 must survive the update
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        for _ in range(4):
            c.compute()
        got = c.read_particles()
    want = oracle.run(sp, p, g, 4, mode=oracle.MODE_SKIP_INERT)
    assert np.isfinite(want["pos"]).all() and np.isfinite(want["vel"]).all()
    assert_particles_bit_equal(got, want, case)
