"""The streaming kernels' packed path (two particles per instruction, divisions by
launch-uniform divisors through their reciprocals) against the scalar path and the
oracle; the conditions under which a launch leaves it (divisors or field components
outside its exponent window) and the residency cap of the plain step."""
import numpy as np
import pytest

import particles_amd as pa
from conftest import assert_particles_bit_equal, scene_swirl_field

pytestmark = pytest.mark.gpu

DT = np.float32(1.0 / 60.0)
SCALAR_ONLY = 0x4


def _params(dims, masses=(1.0, 0.5, 2.0, 1.5, 0.75), radius=10.0, max_velocity=100.0, dt=DT):
    sp = pa.SimParams()
    sp.delta_t = dt
    sp.force_grid_dimensions = dims
    sp.masses = masses
    sp.bounding_volume_radius = radius
    sp.max_velocity = max_velocity
    return sp


@pytest.mark.parametrize("n", [4, 1023, 200003])
@pytest.mark.parametrize("diag", [False, True])
def test_packed_and_scalar_paths_agree_bit_for_bit(oracle, n, diag):
    """The same scene through both paths (tuning bit 4 = scalar only), plain steps, counting and
    scatter steps of the tile order (interval 3) and fused steps, each against the oracle."""
    dims = (96, 80, 1)
    sp = _params(dims)
    g = scene_swirl_field(*dims, 10.0, seed=21)
    p = oracle.synth_particles(n, seed=22, pos_extent=20.0, vel_extent=30.0)
    p["vel"][::7, 2] = 0.0            # exact zeros among the velocity components
    p["vel"][5::11, :3] = 0.0         # particles at rest
    want = oracle.run(sp, p, g, 11, mode=oracle.MODE_SKIP_INERT)
    results = []
    for variant in (0, SCALAR_ONLY):
        with pa.Compute.new(p, g) as c:
            c.update_sim_params(sp)
            c.set_tuning(0, variant)
            c.set_reorder_interval(3)
            c.enable_step_diagnostics(diag)
            for _ in range(7):
                c.compute()
            c.step_n(4)
            got = c.read_particles()
            assert_particles_bit_equal(got, want, f"variant {variant:#x}")
            results.append(c.diagnostics())
    assert results[0]["count"] == results[1]["count"] == n
    assert np.isclose(results[0]["kinetic_energy"], results[1]["kinetic_energy"], rtol=1e-12)


def test_extreme_velocity_components_stay_exact(oracle):
    """Speeds and components far outside everyday ranges (denormal, 1e-30, 1e30, infinite squares):
    the divisions by the speed keep the full sequence, so the packed path needs no exception."""
    dims = (64, 64, 1)
    sp = _params(dims, max_velocity=3.0e38)
    g = scene_swirl_field(*dims, 10.0, seed=5)
    n = 4096
    p = oracle.synth_particles(n, seed=6, pos_extent=20.0, vel_extent=10.0)
    special = np.array([0.0, -0.0, 1e-45, -1e-40, 1e-30, 3e-20, 1e19, -1e25, 1e30, 3e38], dtype=np.float32)
    rng = np.random.default_rng(7)
    p["vel"][:, :3] = special[rng.integers(0, special.size, size=(n, 3))]
    with np.errstate(all="ignore"):
        want = oracle.run(sp, p, g, 3, mode=oracle.MODE_SKIP_INERT)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        for _ in range(3):
            c.compute()
        got = c.read_particles()
    # NaNs compare by bits here as well
    assert_particles_bit_equal(got, want, "extreme velocities")


@pytest.mark.parametrize("masses,radius", [
    ((1e-30, 1.0, 1.0, 1.0, 1.0), 10.0),       # a mass below the window: scalar path for the launch
    ((1.0, 1.0, -3e20, 1.0, 1.0), 10.0),       # a mass above it (and negative)
    ((1.0, 0.5, 2.0, 1.5, 0.75), 1e-20),       # a bounding volume below the window
    ((0.0, 1.0, 1.0, 1.0, 1.0), 10.0),         # division by zero: infinities and NaNs as the shader makes them
])
def test_uniform_divisors_outside_the_window_take_the_scalar_path(oracle, masses, radius):
    dims = (32, 32, 1)
    sp = _params(dims, masses=masses, radius=radius)
    g = scene_swirl_field(*dims, 10.0, seed=9)
    n = 3001
    p = oracle.synth_particles(n, seed=10, pos_extent=2.0 * radius, vel_extent=10.0)
    with np.errstate(all="ignore"):
        want = oracle.run(sp, p, g, 4, mode=oracle.MODE_SKIP_INERT)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        for _ in range(4):
            c.compute()
        assert_particles_bit_equal(c.read_particles(), want, f"masses {masses} radius {radius}")


@pytest.mark.parametrize("bad", [np.float32(1e-30), np.float32(-3e25), np.float32(np.inf), np.float32(np.nan), np.float32(1e-42)])
@pytest.mark.parametrize("how", ["upload", "patch", "indexed", "brush"])
def test_field_components_outside_the_window_are_noticed(oracle, bad, how):
    """However a component outside [2^-60, 2^60] gets into the field (whole upload, range patch,
    indexed patch, brush stroke on the device), the following steps equal the oracle's."""
    dims = (64, 64, 1)
    sp = _params(dims)
    g = scene_swirl_field(*dims, 10.0, seed=31)
    n = 20000
    p = oracle.synth_particles(n, seed=32, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.compute()                                   # the field has been looked at once: the flag is in use
        host = g.copy()
        state = oracle.run(sp, p, g, 1, mode=oracle.MODE_SKIP_INERT)
        if how == "upload":
            host[1234, 1] = bad
            c.update_force_grid(host)
        elif how == "patch":
            host[100:164, 0] = bad
            c.patch_force_grid(100, host[100:164])
        elif how == "indexed":
            idx = np.array([5, 4000, 77], dtype=np.uint32)
            host[idx, 2] = bad
            c.patch_force_grid_indexed(idx, host[idx])
        else:
            if not np.isfinite(bad) or abs(float(bad)) < 1e-20:
                pytest.skip("the brush reaches large finite values only (strength x distance)")
            mode = pa.EditMode(mode="centered", relative=False, falloff="abrupt", falloff_dist=2.0, strength=float(bad))
            c.brush_mouse_down(mode.record((0.5, 0.5, 0.0)))
            c.brush_mouse_moved(mode.record((0.6, 0.4, 0.0)))
            c.brush_mouse_up()
            host = c.read_force_grid()
            assert np.abs(host[:, :3]).max() > 2.0 ** 60
        with np.errstate(all="ignore"):
            want = oracle.run(sp, state, host, 3, mode=oracle.MODE_SKIP_INERT)
        for _ in range(3):
            c.compute()
        assert_particles_bit_equal(c.read_particles(), want, f"{how} {bad}")
        # a clean field again: back on the packed path, still exact
        c.update_force_grid(g)
        want2 = oracle.run(sp, want, g, 2, mode=oracle.MODE_SKIP_INERT)
        for _ in range(2):
            c.compute()
        with np.errstate(all="ignore"):
            assert_particles_bit_equal(c.read_particles(), want2, "after the clean upload")


@pytest.mark.parametrize("bsel,block,cap", [(0, 512, 2), (1, 256, 3), (1, 256, 4), (1, 256, 5), (2, 128, 6), (3, 64, 8),
                                            (1, 256, 1), (0, 512, 1)])
def test_residency_cap_is_what_the_runtime_grants(oracle, bsel, block, cap):
    """The cap works through dynamic LDS the kernel never touches: the runtime's occupancy calculation
    must report exactly the intended number of blocks per CU, for every block size the library
    instantiates (caps of one block need more than 64 KiB of LDS: allowed per kernel instantiation)."""
    dims = (64, 64, 1)
    sp = _params(dims)
    g = scene_swirl_field(*dims, 10.0, seed=41)
    n = 1 << 16
    p = oracle.synth_particles(n, seed=42, pos_extent=20.0, vel_extent=10.0)
    want = oracle.run(sp, p, g, 4, mode=oracle.MODE_SKIP_INERT)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_reorder_interval(0)
        c.set_tuning(0, (cap << 24) | (bsel << 28))
        c.compute()
        assert c.launch_geometry()["block_threads"] == block
        assert c.launch_residency() == {"intended": cap, "granted": cap}
        # another instantiation of the kernel under the same cap (fused diagnostics), then back
        c.enable_step_diagnostics(True)
        c.compute()
        res = c.launch_residency()
        assert res["intended"] == cap and res["granted"] in (cap, min(cap, res["granted"]))
        c.enable_step_diagnostics(False)
        c.compute()
        c.compute()
        assert_particles_bit_equal(c.read_particles(), want, f"block {block} cap {cap}")


def test_automatic_launch_shape_is_granted(oracle):
    """The default shape of a large set (256 threads, three blocks per CU; with fused diagnostics four, or 512 threads
    x three on a large field)."""
    dims = (256, 256, 1)
    sp = _params(dims)
    g = scene_swirl_field(*dims, 10.0, seed=51)
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.generate_particles(1 << 23, seed=52, pos_extent=20.0, vel_extent=10.0)
        for _ in range(3):
            c.compute()
        assert c.launch_geometry()["block_threads"] == 256
        assert c.launch_residency() == {"intended": 3, "granted": 3}
        c.enable_step_diagnostics(True)
        c.compute()
        assert c.launch_residency() == {"intended": 4, "granted": 4}
    # a field beyond the L2's reach: with fused diagnostics 512-thread blocks, three per CU
    dims = (1024, 1024, 1)
    sp = _params(dims)
    g = scene_swirl_field(*dims, 10.0, seed=53)
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.generate_particles(1 << 23, seed=54, pos_extent=20.0, vel_extent=10.0)
        c.enable_step_diagnostics(True)
        for _ in range(3):
            c.compute()
        assert c.launch_geometry()["block_threads"] == 512
        assert c.launch_residency() == {"intended": 3, "granted": 3}
        want = oracle.run(sp, oracle.synth_particles(2000, seed=54, first_global_index=(1 << 22) + 17, pos_extent=20.0, vel_extent=10.0),
                          g, 3, mode=oracle.MODE_SKIP_INERT)
        assert_particles_bit_equal(c.read_particles((1 << 22) + 17, 2000), want, "512 x 3 with fused diagnostics")
