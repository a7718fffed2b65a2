"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Bit-exact for positions, velocities and cell indices on
the streaming path and on the pair path with distance_exponent = 0; relative
tolerance 1e-5 (north-star bar) where powf is involved."""
import numpy as np
import pytest

import particles_amd as pa
from conftest import assert_particles_bit_equal, scene_swirl_field, binned_single_step_check

pytestmark = pytest.mark.gpu

DT = np.float32(1.0 / 60.0)


def default_scene():
    sp = pa.SimParams()
    p = pa.lattice_particles((5.0, 2.0, 2.0), 10, 10, 10)          # src/main.rs:225-231
    g = sp.new_force_grid_centered().get_force_vectors()             # src/main.rs:142-147,235
    return sp, p, g


def test_default_scene_16_steps_bit_exact(oracle):
    sp, p, g = default_scene()
    sp.delta_t = DT
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        for k in range(1, 17):
            c.compute()
            if k in (1, 2, 15, 16):  # odd and even ping-pong positions
                want = oracle.run(sp, p, g, k, mode=oracle.MODE_ALWAYS_PAIRS)
                assert_particles_bit_equal(c.read_particles(), want, f"default scene, step {k}")
        assert c.last_step_kernel == pa.KERNEL_STREAM
        assert c.step_count == 16


def test_new_starts_with_default_params_dt0(oracle):
    # src/compute.rs:33: Compute::new uploads its own SimParams::new() (delta_t = 0)
    sp, p, g = default_scene()
    with pa.Compute.new(p, g) as c:
        c.compute()
        want = oracle.run(pa.SimParams(), p, g, 1)
        assert_particles_bit_equal(c.read_particles(), want, "dt=0 step")
        # positions unchanged by a dt = 0 step
        assert np.array_equal(c.read_particles()["pos"][:, :3], p["pos"][:, :3])


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 63, 64, 255, 256, 1023, 1024, 1025, 4099, 100003])
def test_stream_kernel_ragged_sizes(oracle, n):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (16, 8, 4)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(16, 8, 4, 10.0, seed=7)
    p = oracle.synth_particles(n, seed=1000 + n, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        assert np.array_equal(c.debug_cell_indices(), oracle.cell_indices(sp, p, g.shape[0]))
        for _ in range(5):
            c.compute()
        want = oracle.run(sp, p, g, 5, mode=oracle.MODE_SKIP_INERT)
        assert_particles_bit_equal(c.read_particles(), want, f"n={n}")
        assert np.array_equal(c.debug_cell_indices(), oracle.cell_indices(sp, want, g.shape[0]))


@pytest.mark.parametrize("block_sel,per_cu", [(0, 2), (1, 5), (2, 10), (3, 12), (1, 15), (0, 1)])
@pytest.mark.parametrize("n,diag", [(100003, False), (4099, False), (1025, True)])
def test_plain_step_launch_shapes_give_identical_results(oracle, block_sel, per_cu, n, diag):
    """The plain streaming step in blocks of 512/256/128/64 threads with a cap on the resident
    blocks per CU (what large sets use automatically): ragged sizes, bit-equal to the oracle;
    with fused diagnostics only the 512-thread shape applies (the others fall back to it)."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (16, 8, 4)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(16, 8, 4, 10.0, seed=7)
    p = oracle.synth_particles(n, seed=77 + n, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_tuning(0, (block_sel << 28) | (per_cu << 24))
        c.enable_step_diagnostics(diag)
        for _ in range(3):
            c.compute()
        want = oracle.run(sp, p, g, 3, mode=oracle.MODE_SKIP_INERT)
        assert_particles_bit_equal(c.read_particles(), want, f"block_sel={block_sel} per_cu={per_cu} n={n}")
        if diag:
            d = c.diagnostics()      # with step diagnostics on: the record the last step accumulated
            dw = oracle.diagnostics(sp, want)
            assert d["count"] == dw["count"] == n
            assert abs(d["kinetic_energy"] - dw["kinetic_energy"]) <= 1e-12 * dw["kinetic_energy"]
            assert np.array_equal(d["pos_max"], dw["pos_max"]) and np.array_equal(d["pos_min"], dw["pos_min"])


def test_automatic_launch_shape_of_large_sets_is_bit_exact(oracle):
    """From 3 * 2^24 particles on the plain step runs in 256-thread blocks, five to a CU; a ragged
    count above the switch, field fetched through the caches in caller order (no tile order),
    slices against the oracle."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (64, 64, 1)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(64, 64, 1, 10.0, seed=11)
    n = (1 << 26) + 1027
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.set_reorder_interval(0)
        c.generate_particles(n, seed=99, pos_extent=20.0, vel_extent=10.0)
        for _ in range(3):
            c.compute()
        for first in (0, 33554432 - 500, n - 3000):
            p = oracle.synth_particles(3000, seed=99, first_global_index=first, pos_extent=20.0, vel_extent=10.0)
            want = oracle.run(sp, p, g, 3, mode=oracle.MODE_SKIP_INERT)
            assert_particles_bit_equal(c.read_particles(first, 3000), want, f"slice at {first}")


@pytest.mark.parametrize("n,diag,block_bits", [(600003, False, 0x13000000), (1100001, True, 0x03000000),
                                               (262144 + 5, False, 0x13000000), (2 ** 21 + 2049, False, 0x22000000)])
def test_plain_step_with_tiles_dealt_per_xcd_is_bit_exact(oracle, n, diag, block_bits):
    """Tuning bit 0x100: the blocks of the plain step take the tiles in another order (runs of 128
    tiles, each XCD 16 consecutive ones; the grid is rounded up to a multiple of 128 blocks and
    the surplus blocks own no tile).  Ragged counts whose last run is partly empty, with and without
    fused diagnostics (one partial per BLOCK, the empty ones included): bit-equal to the oracle."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (64, 32, 2)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(64, 32, 2, 10.0, seed=17)
    p = oracle.synth_particles(n, seed=5 + n, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_tuning(0, 0x100 | block_bits)
        c.enable_step_diagnostics(diag)
        for _ in range(3):
            c.compute()
        want = oracle.run(sp, p, g, 3, mode=oracle.MODE_SKIP_INERT)
        assert_particles_bit_equal(c.read_particles(), want, f"n={n}")
        if diag:
            d = c.diagnostics()
            dw = oracle.diagnostics(sp, want)
            assert d["count"] == dw["count"] == n
            assert abs(d["kinetic_energy"] - dw["kinetic_energy"]) <= 1e-12 * dw["kinetic_energy"]
            assert np.array_equal(d["pos_max"], dw["pos_max"]) and np.array_equal(d["pos_min"], dw["pos_min"])


def test_automatic_xcd_dealing_in_its_window_is_bit_exact(oracle):
    """2^25 + 1027 particles in tile order on a 1024^2 field (32 per cell): inside the window in which
    the plain step deals its tiles per XCD by itself; 26 steps = a refresh cycle and two steps of the
    next, slices against the oracle."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (1024, 1024, 1)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(1024, 1024, 1, 10.0, seed=11)
    n = (1 << 25) + 1027
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.generate_particles(n, seed=98, pos_extent=20.0, vel_extent=10.0)
        for _ in range(26):
            c.compute()
        assert c.reorder_count >= 2
        for first in (0, 16777216 - 500, n - 2000):
            p = oracle.synth_particles(2000, seed=98, first_global_index=first, pos_extent=20.0, vel_extent=10.0)
            want = oracle.run(sp, p, g, 26, mode=oracle.MODE_SKIP_INERT)
            assert_particles_bit_equal(c.read_particles(first, 2000), want, f"slice at {first}")


def test_empty_particle_set():
    sp, p, g = default_scene()
    with pa.Compute.new(p[:0], g) as c:
        assert c.num_particles == 0
        c.compute()
        assert c.read_particles().shape == (0,)
        assert c.diagnostics()["count"] == 0


def test_device_generator_matches_oracle(oracle):
    sp, p, g = default_scene()
    with pa.Compute.new(p, g) as c:
        c.generate_particles(5000, seed=42, first_global_index=123456789012, pos_extent=20.0, vel_extent=10.0)
        want = oracle.synth_particles(5000, seed=42, first_global_index=123456789012, pos_extent=20.0, vel_extent=10.0)
        assert_particles_bit_equal(c.read_particles(), want, "synthetic scene")


def pair_params(exponent):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.cut_off_distance = 1.5
    sp.distance_exponent = exponent
    rng = np.random.default_rng(99)
    sp.polys[:] = rng.uniform(-2.0, 2.0, size=(25, 8)).astype(np.float32)
    return sp


@pytest.mark.parametrize("n", [2, 257, 1000, 3001])
def test_pair_kernel_exponent0_bit_exact(oracle, n):
    sp = pair_params(0.0)
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(n, seed=5, pos_extent=4.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        for _ in range(3):
            c.compute()
        assert c.last_step_kernel == pa.KERNEL_PAIRS
        want = oracle.run(sp, p, g, 3)
        assert_particles_bit_equal(c.read_particles(), want, f"pairs n={n}")


@pytest.mark.parametrize("exponent", [-2.0, 1.5])
def test_pair_kernel_powf_within_tolerance(oracle, exponent):
    sp = pair_params(exponent)
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(1500, seed=6, pos_extent=4.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.compute()
        got = c.read_particles()
    want = oracle.run(sp, p, g, 1)
    # single step, identical inputs: |a-b| <= 1e-5*max(|a|,|b|) + 1e-6, positions
    # compared modulo the wrap (2r)
    two_r = 2.0 * sp.bounding_volume_radius
    dpos = np.abs(got["pos"][:, :3] - want["pos"][:, :3])
    dpos = np.minimum(dpos, np.abs(two_r - dpos))
    tol_pos = 1e-5 * np.maximum(np.abs(got["pos"][:, :3]), np.abs(want["pos"][:, :3])) + 1e-6
    tol_vel = 1e-5 * np.maximum(np.abs(got["vel"][:, :3]), np.abs(want["vel"][:, :3])) + 1e-6
    assert np.all(dpos <= tol_pos)
    assert np.all(np.abs(got["vel"][:, :3] - want["vel"][:, :3]) <= tol_vel)


def test_stream_equals_pair_kernel_with_zero_polys(oracle):
    sp = pa.SimParams()
    sp.delta_t = DT
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(2000, seed=8, pos_extent=2.0, vel_extent=10.0)
    res = []
    for forced in (False, True):
        with pa.Compute.new(p, g) as c:
            c.update_sim_params(sp)
            c.force_pair_kernel(forced)
            for _ in range(4):
                c.compute()
            assert c.last_step_kernel == (pa.KERNEL_PAIRS if forced else pa.KERNEL_STREAM)
            res.append(c.read_particles())
    assert_particles_bit_equal(res[0], res[1], "stream vs pairs")


def test_diagnostics_match_oracle(oracle):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(50001, seed=9, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        for fused in (False, True):
            c.enable_step_diagnostics(fused)
            c.compute()
            d = c.diagnostics()
            want = oracle.diagnostics(sp, c.read_particles())
            assert d["count"] == want["count"] == 50001
            assert abs(d["kinetic_energy"] - want["kinetic_energy"]) <= 1e-12 * want["kinetic_energy"]
            assert np.array_equal(d["pos_max"], want["pos_max"])
            assert np.array_equal(d["pos_min"], want["pos_min"])


def test_field_patch_and_readback_ranges(oracle):
    sp, p, g = default_scene()
    sp.delta_t = DT
    g2 = g.copy()
    g2[100:200, :3] = np.float32(3.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.patch_force_grid(100, g2[100:200])
        c.compute()
        want = oracle.run(sp, p, g2, 1)
        assert_particles_bit_equal(c.read_particles(), want, "patched field")
        assert_particles_bit_equal(c.read_particles(17, 300), want[17:317], "sub-range")
        soa = c.read_soa(5, 10)
        assert np.array_equal(soa["x"], want["pos"][5:15, 0]) and np.array_equal(soa["ty"], want["ty"][5:15])
        with pytest.raises(pa.PclError):
            c.read_particles(990, 20)
        with pytest.raises(pa.PclError):
            c.patch_force_grid(990, g2[:20])


def test_invalid_type_rejected():
    sp, p, g = default_scene()
    bad = p.copy()
    bad["ty"][3] = 5
    with pytest.raises(pa.PclError):
        pa.Compute.new(bad, g)


def test_lds_field_path_large_n(oracle):
    # 10^3 cells = 16 KB: staged in LDS by a persistent grid from 2^22 particles on (8 tiles per block)
    sp = pa.SimParams()
    sp.delta_t = DT
    g = sp.new_force_grid_centered().get_force_vectors()
    n = (1 << 22) + 3
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.generate_particles(n, seed=77, pos_extent=20.0, vel_extent=10.0)
        c.compute()
        c.compute()
        assert c.launch_geometry()["grid_blocks"] == 512 and c.reorder_interval_in_use == 0   # the persistent grid
        got = c.read_particles(n - 5000, 5000)
    p = oracle.synth_particles(5000, seed=77, first_global_index=n - 5000, pos_extent=20.0, vel_extent=10.0)
    want = oracle.run(sp, p, g, 2, mode=oracle.MODE_SKIP_INERT)
    assert_particles_bit_equal(got, want, "LDS-staged field")


# ---- field-tile order: results and read-back are independent of the memory order

@pytest.mark.parametrize("n,interval", [(4096, 1), (4099, 2), (100003, 3), (300001, 5)])
def test_reorder_keeps_results_and_caller_order(oracle, n, interval):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (128, 64, 2)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(128, 64, 2, 10.0, seed=3)
    p = oracle.synth_particles(n, seed=n, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_reorder_interval(interval)
        c.enable_step_diagnostics(True)
        for _ in range(7):
            c.compute()
        assert c.reorder_count >= 7 // interval
        want = oracle.run(sp, p, g, 7, mode=oracle.MODE_SKIP_INERT)
        assert_particles_bit_equal(c.read_particles(), want, f"tile order n={n}")
        assert_particles_bit_equal(c.read_particles(1000, 2000), want[1000:3000], "sub-range in caller order")
        soa = c.read_soa(17, 500)
        assert np.array_equal(soa["vy"], want["vel"][17:517, 1]) and np.array_equal(soa["ty"], want["ty"][17:517])
        assert np.array_equal(c.debug_cell_indices(), oracle.cell_indices(sp, want, g.shape[0]))
        d, wd = c.diagnostics(), oracle.diagnostics(sp, want)
        assert d["count"] == n and abs(d["kinetic_energy"] - wd["kinetic_energy"]) <= 1e-12 * wd["kinetic_energy"]
        assert np.array_equal(d["pos_max"], wd["pos_max"]) and np.array_equal(d["pos_min"], wd["pos_min"])
        # a fresh upload is back in caller order
        c.upload_particles(p)
        c.compute()
        assert_particles_bit_equal(c.read_particles(), oracle.run(sp, p, g, 1, mode=oracle.MODE_SKIP_INERT), "after upload")


def test_pair_kernel_after_reorder_runs_in_caller_order(oracle):
    sp = pa.SimParams()
    sp.delta_t = DT
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(5000, seed=12, pos_extent=4.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_reorder_interval(1)
        c.compute()
        c.compute()
        assert c.reorder_count == 2
        mid = oracle.run(sp, p, g, 2, mode=oracle.MODE_SKIP_INERT)
        sp2 = pair_params(0.0)
        c.update_sim_params(sp2)
        c.compute()
        assert c.last_step_kernel == pa.KERNEL_PAIRS
        assert_particles_bit_equal(c.read_particles(), oracle.run(sp2, mid, g, 1), "pairs after tile order")


@pytest.mark.parametrize("log2n", [26, 28])
def test_full_size_slices_match_oracle(oracle, log2n):
    """2^26 and 2^28 particles (2^28 on this field is BASELINE configuration 3, the workload the
    bandwidth target is quoted on) generated on the device, 1024^2 field, everything automatic: the
    default launch shape (256-thread blocks, three per CU), the tile order refreshed every 24 steps,
    52 steps = two whole refresh cycles after the first sort.  Particles are independent, so slices
    of the big run are checked against the oracle stepping the same global indices."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (1024, 1024, 1)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(1024, 1024, 1, 10.0, seed=5)
    n = 1 << log2n
    steps = 52
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.generate_particles(n, seed=2025, pos_extent=20.0, vel_extent=10.0)
        assert c.reorder_interval_in_use == 24
        for _ in range(steps):
            c.compute()
        assert c.reorder_count >= 4          # the first sort in two passes, then one refresh per cycle
        assert c.launch_geometry()["block_threads"] == 256
        assert c.launch_residency() == {"intended": 3, "granted": 3}
        for first in (0, 12345678, n // 2 - 2000, n - 4000):
            p = oracle.synth_particles(4000, seed=2025, first_global_index=first, pos_extent=20.0, vel_extent=10.0)
            want = oracle.run(sp, p, g, steps, mode=oracle.MODE_SKIP_INERT)
            assert_particles_bit_equal(c.read_particles(first, 4000), want, f"2^{log2n} slice at {first}")
            assert np.array_equal(c.debug_cell_indices(first, 4000), oracle.cell_indices(sp, want, g.shape[0]))
        assert c.diagnostics()["count"] == n


def _state_checksum(p):
    """numpy twin of pcl_debug_state_checksum over records in caller order."""
    M = np.uint64(0xFF51AFD7ED558CCD)
    with np.errstate(over="ignore"):
        h = (np.arange(p.shape[0], dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        words = [p["pos"][:, k].view(np.uint32) for k in range(3)] + [p["vel"][:, k].view(np.uint32) for k in range(3)] + [p["ty"]]
        for w in words:
            h = (h ^ w.astype(np.uint64)) * M
            h = h ^ (h >> np.uint64(32))
        return int(h.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(h))


@pytest.mark.parametrize("n", [1, 1000, 300007])
def test_state_checksum_matches_its_numpy_twin_in_any_memory_order(oracle, n):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (64, 64, 1)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(64, 64, 1, 10.0, seed=3)
    p = oracle.synth_particles(n, seed=n, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        q = p.copy(); q["pos"][:, 3] = 1.0; q["vel"][:, 3] = 1.0
        assert c.debug_state_checksum() == _state_checksum(q)
        c.set_reorder_interval(2)
        for _ in range(5):
            c.compute()
        if n >= 4096:
            assert c.reorder_count >= 2 and not np.array_equal(c.debug_slot_order(), np.arange(n, dtype=np.uint32))
        got = c.read_particles()
        assert c.debug_state_checksum() == _state_checksum(got)
        # one bit of one particle changes both words
        got["vel"][n // 2, 1] = np.nextafter(got["vel"][n // 2, 1], np.float32(np.inf))
        a, b = _state_checksum(got)
        assert a != c.debug_state_checksum()[0] and b != c.debug_state_checksum()[1]


@pytest.mark.parametrize("log2n", [28, 30])
def test_every_particle_of_a_full_size_run_agrees_between_independent_kernel_paths(oracle, log2n):
    """BASELINE's full sizes, EVERY particle: the headline workload (2^28 particles, 1024^2 field) and 2^30
    particles stepped (a) as shipped (packed arithmetic, tile order refreshed by the counting and scatter
    steps, per-XCD launch shape), (b) on the scalar path with full divisions, the particles never moved
    from caller order, (c) through pcl_step_n's fused launches.  The three states must agree in an
    order-independent checksum over all particles (caller index, six state words, type); slices of (a)
    are held to the oracle in test_full_size_slices_match_oracle, and the checksum itself to its numpy twin."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (1024, 1024, 1)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(1024, 1024, 1, 10.0, seed=5)
    n = 1 << log2n
    steps = 52 if log2n == 28 else 27
    sums = {}
    for name, variant, interval in (("shipped", 0, -1), ("scalar, caller order", 0x4, 0), ("fused", 0, -1)):
        with pa.Compute.new(pa.empty_particles(0), g) as c:
            c.update_sim_params(sp)
            c.set_tuning(0, variant)
            c.set_reorder_interval(interval)
            c.generate_particles(n, seed=2025, pos_extent=20.0, vel_extent=10.0)
            if name == "fused":
                c.step_n(steps)
            else:
                for _ in range(steps):
                    c.compute()
            if interval == 0:
                assert c.reorder_count == 0
            else:
                assert c.reorder_count >= 3
            sums[name] = c.debug_state_checksum()
            if name == "shipped":     # anchor: one slice against the oracle
                first = n // 3
                p = oracle.synth_particles(1000, seed=2025, first_global_index=first, pos_extent=20.0, vel_extent=10.0)
                want = oracle.run(sp, p, g, steps, mode=oracle.MODE_SKIP_INERT)
                assert_particles_bit_equal(c.read_particles(first, 1000), want, f"2^{log2n} slice at {first}")
    assert sums["shipped"] == sums["scalar, caller order"] == sums["fused"], sums


@pytest.mark.parametrize("shards", [2, 8])
def test_index_range_shards_of_the_headline_workload_add_up_to_the_unsharded_run(oracle, shards):
    """BASELINE configuration 3 under strong scaling, every particle: 2^28 particles on the 1024^2 field split
    by index range into 2 and 8 handles (2^27 and 2^25 particles each: the shares of 2 and 8 GPUs; the latter
    inside the window where the plain step deals its tiles per XCD), each generating and stepping its own range
    for 52 steps with everything automatic, exactly as a rank does.  With the pair term inert no rank needs
    another, so the handles run one after the other on this GPU.  The ranges' checksums (each with its first
    global index) must combine -- sums mod 2^64, xors -- to the words of ONE handle holding all 2^28."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (1024, 1024, 1)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(1024, 1024, 1, 10.0, seed=5)
    n, steps = 1 << 28, 52

    def run(first, count):
        with pa.Compute.new(pa.empty_particles(0), g) as c:
            c.update_sim_params(sp)
            c.generate_particles(count, seed=2025, first_global_index=first, pos_extent=20.0, vel_extent=10.0)
            for _ in range(steps):
                c.compute()
            assert c.reorder_count >= 4
            return c.debug_state_checksum(first)

    whole = run(0, n)
    total, xored = 0, 0
    for rank in range(shards):
        first, last = pa.shard_range(n, rank, shards)
        a, b = run(first, last - first)
        total = (total + a) % (1 << 64)
        xored ^= b
    assert (total, xored) == whole


def test_config4_share_with_diagnostics_every_step_through_the_library_communicator(oracle):
    """One GPU's share of BASELINE configuration 4 (2^30 particles over 8 GPUs = 2^27 each, 1024^2 field,
    diagnostics all-reduced every step), stepped through pcl_comm_step on a one-rank RCCL communicator
    attached to the handle: after every step the all-reduced record against the oracle's over ALL 2^27
    particles (count and bounds exact, kinetic energy to 1e-11: f64 sums of 1.3e8 terms in different orders), at the end slices bit for bit.  Ten
    steps at a refresh interval of 4, so that the plain, the counting and the scatter step with fused
    diagnostics all run at this size."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (1024, 1024, 1)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(1024, 1024, 1, 10.0, seed=5)
    n = 1 << 27
    steps = 10
    # the oracle's diagnostics of every step, chunk by chunk (particles are independent; sums in f64)
    chunk = 1 << 22
    threads = max(1, min(oracle.max_threads(), 16))
    ke = np.zeros(steps)
    mx = np.full((steps, 3), -np.inf, dtype=np.float32)
    mn = np.full((steps, 3), np.inf, dtype=np.float32)
    for first in range(0, n, chunk):
        p = oracle.synth_particles(chunk, seed=77, first_global_index=first, pos_extent=20.0, vel_extent=10.0)
        for k in range(steps):
            p = oracle.step(sp, p, g, mode=oracle.MODE_SKIP_INERT, threads=threads)
            d = oracle.diagnostics(sp, p)
            ke[k] += d["kinetic_energy"]
            mx[k] = np.maximum(mx[k], d["pos_max"])
            mn[k] = np.minimum(mn[k], d["pos_min"])
    with pa.Compute.new(pa.empty_particles(0), np.zeros((1, 4), dtype=np.float32)) as c:
        c.comm_attach(pa.Compute.comm_unique_id(), 0, 1)
        assert c.comm_count == 1
        c.comm_update_force_grid(g, g.shape[0])
        c.update_sim_params(sp)
        c.generate_particles(n, seed=77, pos_extent=20.0, vel_extent=10.0)
        c.set_reorder_interval(4)
        c.enable_step_diagnostics(True)
        for k in range(steps):
            c.comm_step()
            d = c.comm_diagnostics()
            assert d["count"] == n, f"step {k + 1}"
            assert abs(d["kinetic_energy"] - ke[k]) <= 1e-11 * ke[k], f"step {k + 1}: {d['kinetic_energy']} vs {ke[k]}"
            assert np.array_equal(d["pos_max"], mx[k]) and np.array_equal(d["pos_min"], mn[k]), f"step {k + 1}"
        assert c.reorder_count >= 3
        for first in (0, n // 2 - 1000, n - 3000):
            p = oracle.synth_particles(3000, seed=77, first_global_index=first, pos_extent=20.0, vel_extent=10.0)
            want = oracle.run(sp, p, g, steps, mode=oracle.MODE_SKIP_INERT)
            assert_particles_bit_equal(c.read_particles(first, 3000), want, f"slice at {first}")


def test_one_billion_particles_slices_match_oracle(oracle):
    """2^30 particles on one GPU (4 GiB per array: byte offsets beyond 32 bits),
    one full refresh cycle of the tile order; 2^31 passes the same check when
    run by hand (tests/big_sizes_check.py 31)."""
    import big_sizes_check
    big_sizes_check.check([30])


def test_cpp_host_mirror_matches_oracle(oracle, tmp_path):
    """include/particles_compute.hpp driven like the reference's App
    (per frame: update_force_grid, update_sim_params, compute)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "compute_wrapper_test")
    out = tmp_path / "cpp_out.bin"
    subprocess.run([exe, str(out), "6"], check=True, timeout=120)
    got = np.fromfile(out, dtype=pa.PARTICLE_DTYPE)
    sp, p, g = default_scene()
    sp.delta_t = DT
    assert_particles_bit_equal(got, oracle.run(sp, p, g, 6), "C++ host mirror")


@pytest.mark.parametrize("n", [1, 5, 7, 2051, 4100])
def test_reorder_now_tiny_and_ragged_sets(oracle, n):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (32, 32, 4)
    g = scene_swirl_field(32, 32, 4, 10.0, seed=1)
    p = oracle.synth_particles(n, seed=3 * n, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.reorder_now()
        assert c.reorder_count == 1
        assert_particles_bit_equal(c.read_particles(), _with_unit_w(p), "reorder alone changes nothing")
        c.compute()
        c.reorder_now()
        c.compute()
        assert_particles_bit_equal(c.read_particles(), oracle.run(sp, p, g, 2, mode=oracle.MODE_SKIP_INERT), f"n={n}")


def _with_unit_w(p):
    q = np.array(p, copy=True).view(pa.PARTICLE_DTYPE)
    q["pos"][:, 3] = 1.0
    q["vel"][:, 3] = 1.0
    q["_padd"] = 0
    return q


def test_reorder_3d_field_many_bins(oracle):
    """128 x 128 x 64 cells: more tiles of 256 cells than bins, so tiles grow."""
    dims = (128, 128, 64)
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = dims
    rng = np.random.default_rng(11)
    g = np.ones((dims[0] * dims[1] * dims[2], 4), dtype=np.float32)
    g[:, :3] = rng.uniform(-1, 1, size=(g.shape[0], 3)).astype(np.float32)
    n = 200001
    p = oracle.synth_particles(n, seed=17, pos_extent=20.0, vel_extent=40.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_reorder_interval(2)
        for _ in range(5):
            c.compute()
        assert c.reorder_count >= 2
        assert_particles_bit_equal(c.read_particles(), oracle.run(sp, p, g, 5, mode=oracle.MODE_SKIP_INERT), "3-D field")


def test_changing_dt_between_counting_and_scatter_steps(oracle):
    """delta_t changes every frame in the reference (wall clock, src/main.rs:335-342);
    the sort key's lookahead must not make a pending histogram inconsistent."""
    sp = pa.SimParams()
    sp.force_grid_dimensions = (64, 64, 1)
    g = scene_swirl_field(64, 64, 1, 10.0, seed=2)
    n = 50000
    p = oracle.synth_particles(n, seed=23, pos_extent=20.0, vel_extent=10.0)
    cur = p
    with pa.Compute.new(p, g) as c:
        c.set_reorder_interval(3)
        for k in range(10):
            sp.delta_t = np.float32(0.004 + 0.003 * k)
            c.update_sim_params(sp)
            c.compute()
            cur = oracle.step(sp, cur, g, mode=oracle.MODE_SKIP_INERT)
        assert c.reorder_count >= 3
        assert_particles_bit_equal(c.read_particles(), cur, "per-frame dt")


def test_step_timing_reports_each_launch_and_its_kind():
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (64, 64, 1)
    g = scene_swirl_field(64, 64, 1, 10.0, seed=2)
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.generate_particles(1 << 16, seed=1, pos_extent=20.0)
        c.set_reorder_interval(4)
        c.step_timing_begin(9)
        for _ in range(9):
            c.compute()
        ms, kinds = c.step_timing_read()
        assert ms.shape == (9,) and np.all(ms > 0)
        # first step sorts (scatter), then every 4th; the step before a sort also counts
        assert kinds.tolist() == [4, 1, 1, 3, 4, 1, 1, 3, 4]
        c.compute()
        assert c.step_timing_read()[0].size == 9             # recording stopped: nothing new was appended


def test_step_timing_of_a_one_kernel_step_is_the_kernels_own_time():
    """From 2^23 particles on a plain step is one shaped launch, and step timing hands its two events to that launch
    instead of recording them around it (tuning bit 0x800: recorded around every step).  Both ways: every launch has
    a positive time and the right kind, the launch times add up to no more than the stopwatch around the region, and
    the two ways agree on the plain kernel to a tenth."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (256, 256, 1)
    g = scene_swirl_field(256, 256, 1, 10.0, seed=2)
    plain = {}
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        for variant in (0, 0x800):
            c.set_tuning(0, variant)
            c.generate_particles(1 << 23, seed=1, pos_extent=20.0)
            c.set_reorder_interval(6)
            for _ in range(3):
                c.compute()                                   # the first sort and the clocks
            c.step_timing_begin(12)
            c.timer_start()
            for _ in range(12):
                c.compute()
            region = c.timer_stop()
            ms, kinds = c.step_timing_read()
            assert ms.shape == (12,) and np.all(ms > 0), (variant, ms)
            assert sorted(set(kinds.tolist())) == [1, 3, 4] and (kinds == 1).sum() == 8, kinds
            assert ms.sum() <= region * 1.02, (variant, ms.sum(), region)
            plain[variant] = float(ms[kinds == 1].mean())
        assert c.launch_residency()["intended"] == 3          # the shaped launch it is about
    assert abs(plain[0] - plain[0x800]) <= 0.1 * plain[0x800], plain


# ---- cell-binned pair term: same terms, different summation order

def _close(got, want, two_r, rtol=1e-5, atol=1e-6):
    """|a-b| <= rtol*max(|a|,|b|) + atol, positions modulo the wrap."""
    for f in ("pos", "vel"):
        a, b = got[f][:, :3].astype(np.float64), want[f][:, :3].astype(np.float64)
        d = np.abs(a - b)
        if f == "pos":
            d = np.minimum(d, np.abs(two_r - d))       # modulo the wrap
        bad = d > rtol * np.maximum(np.abs(a), np.abs(b)) + atol
        assert not bad.any(), f"{f}: {bad.sum()} components out of tolerance, worst {d.max()}"


@pytest.mark.parametrize("n,cutoff,extent,max_axis", [
    (5000, 1.5, 20.0, 20),        # one-cell reach, 13 cells per axis
    (5000, 2.5, 44.0, 20),        # two-cell reach (16 cells of half the cut-off), particles beyond the box
    (3000, 30.0, 20.0, 20),       # cut-off larger than the box: a single cell
    (1 << 18, 0.45, 22.0, 64),    # from 2^18 particles on up to 64 cells per axis
])
def test_binned_pair_cells_match_the_oracle_bit_for_bit(oracle, n, cutoff, extent, max_axis):
    """The bin work of the cell-binned pair term: cell key of every particle and the end of every
    cell's slot range, against the oracle's restatement (orc_pair_cell_map / orc_pair_cell_keys), with
    particles exactly on cell faces, on the faces of the bounding volume and beyond it."""
    sp = pair_params(0.0)
    sp.cut_off_distance = cutoff
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(n, seed=n + 5, pos_extent=extent, vel_extent=10.0)
    per_axis, reach, inv_cell = oracle.pair_cell_map(sp, True, max_axis)
    r = np.float32(sp.bounding_volume_radius)
    cell_len = np.float32(2.0 * float(r) / per_axis)
    faces = (-r + np.arange(per_axis + 1, dtype=np.float32) * cell_len).astype(np.float32)
    rng = np.random.default_rng(n)
    k = min(n // 2, 2000)
    for axis in range(3):                                   # a cell face on one axis, anything on the others
        p["pos"][axis * k // 3:(axis + 1) * k // 3, axis] = faces[rng.integers(0, faces.size, size=(axis + 1) * k // 3 - axis * k // 3)]
    p["pos"][k:k + 8, :3] = [[r, r, r], [-r, -r, -r], [r, -r, 0], [np.nextafter(r, np.float32(0)), 0, 0],
                             [np.nextafter(r, np.float32(100)), 0, 0], [3 * r, -3 * r, 0], [0, 0, 0], [-0.0, 0.0, -0.0]]
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        cells, bin_end, got_axis, got_reach = c.debug_pair_cells()
        assert (got_axis, got_reach) == (per_axis, reach)
        want = oracle.pair_cell_keys(sp, p, per_axis, inv_cell)
        bad = np.nonzero(cells != want)[0]
        assert bad.size == 0, f"{bad.size} cell keys differ; first {bad[0]}: got {cells[bad[0]]} want {want[bad[0]]} at {p['pos'][bad[0]]}"
        assert np.array_equal(bin_end, np.cumsum(np.bincount(want, minlength=per_axis ** 3)).astype(np.uint32))
        # the order itself: sorted by cell, inside a cell by ascending caller index (what a stable sort of
        # (cell, caller index) pairs leaves; the pair sums run in this order)
        assert np.array_equal(c.debug_slot_order(), np.argsort(want, kind="stable").astype(np.uint32))
        # the sort left the particles in cell order; reads still return caller order
        assert_particles_bit_equal(c.read_particles(), _with_w1(p), "after the cell sort")


@pytest.mark.parametrize("n,per_axis_cutoff", [(1, 1.0), (2047, 1.0), (2048, 1.0), (2049, 0.4), (300001, 0.32), (2 ** 21 + 77, 0.32)])
def test_cell_sort_is_the_stable_sort_by_cell(oracle, n, per_axis_cutoff):
    """The library's own radix sort (pair_sort.hip) on sizes around its chunk of 2048 elements and
    over many blocks, with 64 cells per axis (18-bit keys: three passes) and with few cells (one pass):
    slot order == numpy's stable argsort of the oracle's cell keys."""
    sp = pair_params(0.0)
    sp.cut_off_distance = per_axis_cutoff
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(n, seed=n + 11, pos_extent=24.0, vel_extent=1.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        assert np.array_equal(c.debug_slot_order(), np.arange(n, dtype=np.uint32))
        cells, bin_end, per_axis, reach = c.debug_pair_cells()
        want_axis, want_reach, inv_cell = oracle.pair_cell_map(sp, True, 64 if n >= 2 ** 18 else 20)
        assert (per_axis, reach) == (want_axis, want_reach)
        want = oracle.pair_cell_keys(sp, p, per_axis, inv_cell)
        assert np.array_equal(cells, want)
        assert np.array_equal(c.debug_slot_order(), np.argsort(want, kind="stable").astype(np.uint32))
        assert np.array_equal(bin_end, np.cumsum(np.bincount(want, minlength=per_axis ** 3)).astype(np.uint32))


@pytest.mark.parametrize("n,cutoff,extent,exponent", [
    (40000, 0.3, 2.0, 0.0), (40000, 1.0, 2.0, 0.0), (70001, 1.0, 20.0, 0.0), (9000, 2.5, 24.0, -1.0), (300000, 0.9, 20.0, 0.0), (129, 30.0, 20.0, 0.5)])
def test_the_two_sweeps_of_the_binned_pair_term_agree_bit_for_bit(oracle, n, cutoff, extent, exponent):
    """The sweep with one particle per lane and the sweep with the candidates shared by a wave add the same terms in
    the same order: a start-up cluster, uniform sets with one- and two-cell reach, particles beyond the box, one
    cell, ragged last waves; three steps each, every bit equal."""
    sp = pair_params(exponent)
    sp.cut_off_distance = cutoff
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(n, seed=n + 3, pos_extent=extent, vel_extent=4.0)
    states = []
    for variant in (0x800000, 0x80000000):
        with pa.Compute.new(p, g) as c:
            c.update_sim_params(sp)
            c.set_pair_mode(2)
            c.set_tuning(0, variant)
            for _ in range(3):
                c.compute()
            assert c.pair_sweep_info()["shared_sweep"] == (variant == 0x80000000)
            states.append(c.read_particles())
    assert_particles_bit_equal(states[1], states[0], "shared sweep against per-lane sweep")


def test_the_binned_sweep_follows_the_population_of_the_cells(oracle):
    """A start-up cluster (40000 particles in a 2-unit cube of the 20-unit box: a few dozen crowded cells) moves to
    the shared sweep as soon as the count of occupied cells of the first sort is in (step 3), a uniform set of the
    same size (5 particles per cell) stays with one particle per lane."""
    sp = pair_params(0.0)
    sp.cut_off_distance = 0.5
    g = sp.new_force_grid_centered().get_force_vectors()
    for extent, crowded in ((2.0, True), (20.0, False)):
        p = oracle.synth_particles(40000, seed=8, pos_extent=extent, vel_extent=1.0)
        with pa.Compute.new(p, g) as c:
            c.update_sim_params(sp)
            c.set_pair_mode(2)
            used = []
            for _ in range(5):
                c.compute()
                used.append(c.pair_sweep_info()["shared_sweep"])
            info = c.pair_sweep_info()
            assert used[:2] == [False, False] and used[2:] == [crowded] * 3, (extent, used, info)
            assert 0 < info["cells_occupied"] <= 20 ** 3
            assert (40000 // info["cells_occupied"] >= 512) == crowded


def _with_w1(p):
    q = p.copy()
    q["pos"][:, 3] = 1.0
    q["vel"][:, 3] = 1.0
    return q


@pytest.mark.parametrize("n,cutoff,extent,exponent", [
    (3000, 1.5, 20.0, 0.0), (3001, 0.7, 6.0, 0.0), (8192, 2.5, 20.0, -1.0), (257, 30.0, 20.0, 0.0), (5000, 1.0, 40.0, 0.5),
    (4000, 3.0, 40.0, 0.0), (6000, 4.9, 24.0, 0.0)])   # two-cell reach, particles beyond the box
def test_binned_pair_kernel_matches_oracle(oracle, n, cutoff, extent, exponent):
    sp = pair_params(exponent)
    sp.cut_off_distance = cutoff
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(n, seed=n, pos_extent=extent, vel_extent=10.0)   # extent 40: particles outside the box
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_pair_mode(2)
        c.compute()
        got = c.read_particles()
        assert c.last_step_kernel == pa.KERNEL_PAIRS
    want = oracle.run(sp, p, g, 1)
    binned_single_step_check(oracle, sp, p, g, got, want, 0, f"n={n} cut-off {cutoff} extent {extent} exponent {exponent}")


def test_binned_pairs_many_cells_sorted_with_global_counters(oracle):
    """From 2^18 particles on the cut-off cells may number up to 64 per axis
    (sorted with counters in global memory): slices against the oracle, which
    sums over all particles."""
    sp = pair_params(0.0)
    sp.cut_off_distance = 0.45
    sp.polys[:] *= np.float32(0.2)
    g = sp.new_force_grid_centered().get_force_vectors()
    n = 1 << 18
    p = oracle.synth_particles(n, seed=31, pos_extent=22.0, vel_extent=3.0)   # some particles beyond the box
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.compute()                              # automatic mode: binned, 64^3 cells, two-cell reach
        assert c.last_step_kernel == pa.KERNEL_PAIRS
        got = np.concatenate([c.read_particles(0, 300), c.read_particles(n - 300, 300)])
    binned_single_step_check(oracle, sp, p, g, got[:300], oracle.step_range(sp, p, g, 0, 300), 0, "2^18, first 300")
    binned_single_step_check(oracle, sp, p, g, got[300:], oracle.step_range(sp, p, g, n - 300, 300), n - 300, "2^18, last 300")


def test_binned_and_exact_pair_kernels_agree_over_steps(oracle):
    sp = pair_params(0.0)
    sp.polys[:] *= np.float32(0.05)          # gentle forces: keep the 3-step horizon short of chaos
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(20000, seed=77, pos_extent=20.0, vel_extent=2.0)
    out = []
    for mode in (1, 2):
        with pa.Compute.new(p, g) as c:
            c.update_sim_params(sp)
            c.set_pair_mode(mode)
            for _ in range(3):
                c.compute()
            out.append(c.read_particles())
            d = c.diagnostics()
            assert d["count"] == 20000
    _close(out[1], out[0], 2.0 * sp.bounding_volume_radius, rtol=1e-4, atol=1e-5)
    want = oracle.run(sp, p, g, 3)
    assert_particles_bit_equal(out[0], want, "exact sweep, 3 steps")


def test_binned_pairs_then_stream_then_exact(oracle):
    """Order bookkeeping across kernel families: binned pairs (cell order) ->
    streaming (field-tile order) -> exact pairs (caller order)."""
    sp = pair_params(0.0)
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(6000, seed=5, pos_extent=20.0, vel_extent=5.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_pair_mode(2)
        c.compute()
        mid1 = c.read_particles()
        _close(mid1, oracle.run(sp, p, g, 1), 20.0)
        sp0 = pa.SimParams()
        sp0.delta_t = DT
        c.update_sim_params(sp0)                 # zero force curves: streaming kernel
        c.set_reorder_interval(1)
        c.compute()
        assert c.last_step_kernel == pa.KERNEL_STREAM
        mid2 = c.read_particles()
        assert_particles_bit_equal(mid2, oracle.run(sp0, mid1, g, 1, mode=oracle.MODE_SKIP_INERT), "stream after binned")
        c.update_sim_params(sp)
        c.set_pair_mode(1)
        c.compute()
        assert_particles_bit_equal(c.read_particles(), oracle.run(sp, mid2, g, 1), "exact pairs after stream")


@pytest.mark.parametrize("n,interval,steps", [(1003, 0, 7), (5000, 5, 23), (70001, 12, 40), (4, 0, 2), (3, 0, 5), (50000, 3, 17)])
def test_fused_steps_equal_single_steps(oracle, n, interval, steps):
    """pcl_step_n: runs of streaming steps fused into one launch (state kept
    in registers) give the bits of the same number of single steps, across the
    counting and scatter steps of the tile order."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.force_grid_dimensions = (96, 80, 1)
    g = scene_swirl_field(96, 80, 1, 10.0, seed=8)
    p = oracle.synth_particles(n, seed=n, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_reorder_interval(interval)
        c.enable_step_diagnostics(True)
        c.step_timing_begin(steps)
        c.step_n(steps)
        ms, kinds = c.step_timing_read()
        got = c.read_particles()
        d = c.diagnostics()
        assert c.step_count == steps
    want = oracle.run(sp, p, g, steps, mode=oracle.MODE_SKIP_INERT)
    assert_particles_bit_equal(got, want, f"{steps} steps through pcl_step_n")
    wd = oracle.diagnostics(sp, want)
    assert d["count"] == n and abs(d["kinetic_energy"] - wd["kinetic_energy"]) <= 1e-12 * max(1.0, wd["kinetic_energy"])
    if interval == 0 or (n >= 4096 and interval >= 5):   # interval 3 leaves one plain step per cycle: nothing to fuse
        assert (kinds == pa.STEP_STREAM_FUSED).any() and len(kinds) < steps   # fewer launches than steps


def test_step_n_with_pair_term_falls_back_to_single_steps(oracle):
    sp = pair_params(0.0)
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(500, seed=3, pos_extent=6.0, vel_extent=4.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.step_n(3)
        assert c.step_count == 3 and c.last_step_kernel == pa.KERNEL_PAIRS
        assert_particles_bit_equal(c.read_particles(), oracle.run(sp, p, g, 3), "pair steps through pcl_step_n")


def test_rejected_upload_leaves_a_tile_ordered_set_intact(oracle):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (96, 80, 1)
    g = scene_swirl_field(96, 80, 1, 10.0, seed=8)
    p = oracle.synth_particles(6000, seed=1, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_reorder_interval(2)
        for _ in range(3):
            c.compute()
        assert c.reorder_count >= 1          # slots are no longer in caller order
        bad = p.copy().view(pa.PARTICLE_DTYPE)
        bad["ty"][17] = 9
        with pytest.raises(pa.PclError):
            c.upload_particles(bad)
        want = oracle.run(sp, p, g, 3, mode=oracle.MODE_SKIP_INERT)
        assert_particles_bit_equal(c.read_particles(), want, "state after a rejected upload")


@pytest.mark.parametrize("vel_extent,expect_long", [(0.0, True), (10.0, False)])
def test_automatic_refresh_pacing_follows_the_drift(oracle, vel_extent, expect_long):
    """Automatic mode: particles that stay put stretch the refresh interval
    (24 -> 48 -> 96, each change one cycle after its measurement), fast ones keep it at 24; results are unaffected."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (128, 128, 1)          # 256 KiB: too large for LDS staging
    n = 1 << 20
    g = np.zeros((128 * 128, 4), dtype=np.float32) if expect_long else scene_swirl_field(128, 128, 1, 10.0, seed=3)
    g[:, 3] = 1.0
    steps = 330
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.generate_particles(n, seed=6, pos_extent=20.0, vel_extent=vel_extent)
        assert c.reorder_interval_in_use == 24
        for _ in range(steps):
            c.compute()
        c.synchronize()
        for first in (0, n - 2000):
            p = oracle.synth_particles(2000, seed=6, first_global_index=first, pos_extent=20.0, vel_extent=vel_extent)
            want = oracle.run(sp, p, g, steps, mode=oracle.MODE_SKIP_INERT)
            assert_particles_bit_equal(c.read_particles(first, 2000), want, f"slice at {first}")
        if expect_long:
            # The measurement reaches the host asynchronously, and the host enqueues steps ahead of the GPU: how many
            # refreshes still go out at the old interval depends on how far ahead it runs (6 with no lag, 9-10 seen).
            assert c.reorder_interval_in_use >= 48 and c.reorder_count < 2 + steps // 24      # 15 at a fixed 24
        else:
            assert c.reorder_interval_in_use == 24 and c.reorder_count >= steps // 24 - 1


# ---- BASELINE configuration 2 and the switch between the LDS-staged field and the tile order

@pytest.mark.parametrize("n,dims,expect", [
    (1 << 24, (256, 256, 1), "tile-order"),          # configuration 2: 16M particles, 256 x 256 static field
    ((1 << 24) - 1, (64, 64, 1), "lds"),             # 64 KiB field, one particle short of the switch: staged in LDS
    (1 << 24, (64, 64, 1), "tile-order"),            # the same field from 2^24 particles on: tile order
])
def test_config2_16m_particles_slices_match_oracle(oracle, n, dims, expect):
    """Particles generated on the device, swirl field, automatic settings, 26
    steps = more than one full refresh cycle of the tile order (scatter at steps 1
    and 25, counting steps before them); slices of the result against the
    oracle stepping the same global indices, and the cell indices bit-exact."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = dims
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(*dims, 10.0, seed=12345)
    steps = 26
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.generate_particles(n, seed=12345 + 2, pos_extent=20.0, vel_extent=10.0)
        if expect == "lds":
            assert c.reorder_interval_in_use == 0 and c.launch_geometry()["grid_blocks"] == 512
        else:
            assert c.reorder_interval_in_use == 24
        for _ in range(steps):
            c.compute()
        assert c.last_step_kernel == pa.KERNEL_STREAM
        # 2 refreshes at a fixed 24 (the first one in two passes); the automatic pacing may stretch the interval after the first cycle
        assert (c.reorder_count == 0) if expect == "lds" else (c.reorder_count >= 2)
        for first in (0, 7654321, n // 2 - 2000, n - 4000):
            p = oracle.synth_particles(4000, seed=12345 + 2, first_global_index=first, pos_extent=20.0, vel_extent=10.0)
            want = oracle.run(sp, p, g, steps, mode=oracle.MODE_SKIP_INERT)
            assert_particles_bit_equal(c.read_particles(first, 4000), want, f"n={n} {dims} slice at {first}")
            assert np.array_equal(c.debug_cell_indices(first, 4000), oracle.cell_indices(sp, want, g.shape[0]))
        d = c.diagnostics()
        assert d["count"] == n


def test_config5_64m_particles_with_per_frame_brush_edits(oracle):
    """BASELINE configuration 5 at full size on one GPU: 2^26 particles, 1024^2
    field, every frame a brush stroke on the device field (src/cursor.rs:79-116,
    Shift mode) and one update with fused diagnostics, 18 frames (one refresh
    cycle of the tile order); the field bytes and slices of the particles
    against the oracle applying the same strokes and stepping the same global
    indices."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (1024, 1024, 1)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(1024, 1024, 1, 10.0, seed=5)
    n = 1 << 26
    frames = 18
    em = pa.EditMode(mode="shift", falloff="linear", falloff_dist=1.0, strength=1.0)
    m = oracle.edit_mode("shift", True, "linear", 1.0, 1.0)
    firsts = (0, 33333333, n - 3000)
    slices = [oracle.synth_particles(3000, seed=77, first_global_index=f, pos_extent=20.0, vel_extent=10.0) for f in firsts]
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.enable_step_diagnostics(True)
        c.generate_particles(n, seed=77, pos_extent=20.0, vel_extent=10.0)
        brush = pa.DeviceBrush(c, em)
        brush.mouse_down((0.0, 0.0, 0.0))
        idx, vec = oracle.brush_mouse_down(sp, g, (0.0, 0.0, 0.0), 1.0)
        assert idx.size > 5000                                   # a stroke of several thousand cells
        want_g = g
        for k in range(frames):
            cursor = (0.3 * np.cos(0.05 * k), 0.3 * np.sin(0.05 * k), 0.0)
            brush.mouse_moved(cursor)
            c.compute()
            want_g = oracle.brush_mouse_moved(sp, want_g, m, idx, vec, (0.0, 0.0, 0.0), cursor)
            slices = [oracle.step(sp, s, want_g, mode=oracle.MODE_SKIP_INERT) for s in slices]
        assert c.reorder_count >= 2
        assert c.read_force_grid().tobytes() == want_g.tobytes()
        for f, want in zip(firsts, slices):
            assert_particles_bit_equal(c.read_particles(f, 3000), want, f"slice at {f}")
        d = c.diagnostics()
        assert d["count"] == n
