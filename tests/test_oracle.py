"""CPU tests of the oracle: layouts, hand-derived known answers (each derived
from the text of src/compute.wgsl, cited per test) and agreement with an
independent numpy-float32 restatement.  The reference ships no golden vectors
(no tests, fixtures or recorded outputs), so these are the oracle's pins."""
import math

import numpy as np
import pytest

import particles_amd as pa
import np_restatement as NP
from conftest import assert_particles_bit_equal

F = np.float32
DT = F(1.0 / 60.0)


def default_scene():
    sp = pa.SimParams()
    p = pa.lattice_particles((5.0, 2.0, 2.0), 10, 10, 10)
    g = sp.new_force_grid_centered().get_force_vectors()
    return sp, p, g


def exp_neg(dt):
    """exp(-dt) rounded to fp32 from a double evaluation."""
    return F(math.exp(-float(F(dt))))


# ---- layouts (src/main.rs:76-83,196-201; src/poly7.rs:6-10; src/sim_params.rs:6-17)

def test_layouts(oracle):
    lib = oracle.load()
    assert lib.orc_sizeof_particle() == 48 == pa.PARTICLE_DTYPE.itemsize
    assert lib.orc_sizeof_sim_params() == 912 == pa.SIM_PARAMS_DTYPE.itemsize
    f = pa.SIM_PARAMS_DTYPE.fields
    assert [f[k][1] for k in ("attraction_force", "particle_type_masses", "force_grid_dimensions", "delta_t",
                              "max_velocity", "bounding_volume_radius", "cut_off_distance",
                              "distance_exponent")] == [0, 800, 880, 892, 896, 900, 904, 908]
    assert pa.PARTICLE_DTYPE.fields["pos"][1] == 0
    assert pa.PARTICLE_DTYPE.fields["vel"][1] == 16
    assert pa.PARTICLE_DTYPE.fields["ty"][1] == 32
    assert pa.MASS_WRAP_DTYPE.itemsize == 16 and pa.POLY7_DTYPE.itemsize == 32


def test_default_params_and_scene_generators(oracle):
    sp, p, g = default_scene()
    assert sp.tobytes() == oracle.default_params_bytes()            # src/sim_params.rs:20-31
    assert p.tobytes() == oracle.lattice_particles((5, 2, 2), 10, 10, 10).tobytes()   # src/main.rs:117-154
    assert g.tobytes() == oracle.grid_centered(10, 10, 10).tobytes()                   # src/grid.rs:68-103
    grid = sp.new_force_grid_centered()
    assert grid.get_positions().tobytes() == oracle.grid_positions(10, 10, 10, 10.0).tobytes()  # src/grid.rs:119-138
    assert p.shape == (1000,) and g.shape == (1000, 4)
    assert np.all(p["vel"] == 0) and np.all(p["pos"][:, 3] == 1) and np.array_equal(p["ty"], np.arange(1000) % 5)


# ---- known answers -------------------------------------------------------

def test_kat_rest_particle_and_cell_755(oracle):
    """Lattice particle (0,0,0): cell (5,5,5), field 0 -> stays at rest.
    Lattice particle (4.5,1.8,1.8): ixx = u32(7.25) = 7, ixy = u32(5.9) = 5,
    ixz = 5 -> flat 755, field (5-7,0,0) -> acceleration (-20,0,0)
    (wgsl:81-89,147-148; grid.rs:68-95)."""
    sp, p, g = default_scene()
    sp.delta_t = DT
    ci = oracle.cell_indices(sp, p, 1000)
    assert ci[0] == 555 and tuple(g[555][:3]) == (0.0, 0.0, 0.0)
    i = 9 * 100 + 9 * 10 + 9
    assert np.allclose(p["pos"][i][:3], (4.5, 1.8, 1.8)) and ci[i] == 755
    assert tuple(g[755][:3]) == (-2.0, 0.0, 0.0)
    out = oracle.step(sp, p, g)
    assert tuple(out["pos"][0]) == (0.0, 0.0, 0.0, 1.0) and tuple(out["vel"][0]) == (0.0, 0.0, 0.0, 1.0)
    vx = F(F(0.0) * exp_neg(DT) + F(-20.0) * DT)                 # wgsl:151,154
    assert out["vel"][i][0] == vx and out["vel"][i][1] == 0 and out["vel"][i][2] == 0
    assert out["pos"][i][0] == F(F(4.5) + vx * DT)               # wgsl:163


@pytest.mark.parametrize("k", range(8))
def test_kat_poly7_unit_coefficient_is_power(oracle, k):
    """coeffs = e_k evaluates to x^k in the shader's product order (wgsl:51-56)."""
    c = np.zeros(8, dtype=F)
    c[k] = 1.0
    x = F(1.37)
    x2 = F(x * x); x4 = F(x2 * x2); x3 = F(x2 * x)
    want = [F(1.0), x, x2, x3, x4, F(x2 * x3), F(x3 * x3), F(x4 * x3)][k]
    assert F(oracle.eval_poly7(x, c)) == want


def test_kat_poly7_is_not_horner(oracle):
    c = np.array([0.3, -1.1, 0.7, 2.0, -0.4, 0.9, -1.3, 0.6], dtype=F)
    x = F(0.83)
    x2 = F(x * x); x4 = F(x2 * x2); x3 = F(x2 * x)
    r = F(F(c[7] * x4) * x3)
    for t in (F(F(c[6] * x3) * x3), F(F(c[5] * x2) * x3), F(c[4] * x4), F(c[3] * x3), F(c[2] * x2), F(c[1] * x), c[0]):
        r = F(r + t)
    assert F(oracle.eval_poly7(x, c)) == r


def test_kat_map_range(oracle):
    """wgsl:68-79: degenerate interval -> start_b; clamp below/above -> 0 / dim."""
    assert oracle.map_range(3.0, 2.0, 2.0, 7.0, 9.0) == 7.0
    assert oracle.map_range(-50.0, -10.0, 10.0, 0.0, 10.0) == 0.0
    assert oracle.map_range(50.0, -10.0, 10.0, 0.0, 10.0) == 10.0
    assert oracle.map_range(0.0, -10.0, 10.0, 0.0, 10.0) == 5.0


def test_kat_cell_index_edges(oracle):
    """v == +slh maps to ix == dim, one past the end (wgsl:85-88): on z it
    aliases into the next row; at the last cell it is clamped to n_cells-1."""
    dims = (10, 10, 10)
    assert oracle.force_grid_index(20.0, dims, (-10.0, -10.0, -10.0)) == 0
    assert oracle.force_grid_index(20.0, dims, (-10.0, -10.0, 10.0)) == 10        # aliases to (0,1,0)
    assert oracle.force_grid_index(20.0, dims, (10.0, 10.0, 10.0)) == 1110        # past the end
    sp = pa.SimParams()
    p = pa.empty_particles(1)
    p["pos"][0] = (10.0, 10.0, 10.0, 1.0)
    assert oracle.cell_indices(sp, p, 1000)[0] == 999
    assert oracle.force_grid_index(0.0, dims, (1.0, 2.0, 3.0)) == 0               # degenerate range


def test_kat_wrap(oracle):
    """wgsl:58-66: beyond +-max by eps wraps by 2*max once; exactly +-max does not."""
    r = F(10.0)
    up, dn = np.nextafter(r, F(np.inf)), np.nextafter(-r, F(-np.inf))
    assert F(oracle.wrap_symmetrically(up, r)) == F(up - F(2.0) * r)
    assert F(oracle.wrap_symmetrically(dn, r)) == F(dn + F(2.0) * r)
    assert oracle.wrap_symmetrically(10.0, 10.0) == 10.0 and oracle.wrap_symmetrically(-10.0, 10.0) == -10.0
    assert oracle.wrap_symmetrically(35.0, 10.0) == 15.0   # a single wrap only


def test_kat_speed_clamp(oracle):
    """wgsl:157-160: |v| = 250 with cap 100 keeps direction, length 100;
    |v| <= 0.001 is left untouched."""
    sp = pa.SimParams()          # dt = 0: decay 1, no acceleration applied
    g = np.zeros((1000, 4), dtype=F)
    p = pa.empty_particles(2)
    p["vel"][0][:3] = (150.0, 200.0, 0.0)
    p["vel"][1][:3] = (0.0005, 0.0, 0.0)
    out = oracle.step(sp, p, g)
    assert np.allclose(out["vel"][0][:3], (60.0, 80.0, 0.0), rtol=1e-6)
    assert abs(np.linalg.norm(out["vel"][0][:3].astype(np.float64)) - 100.0) < 1e-4
    assert out["vel"][1][0] == F(0.0005)
    assert np.array_equal(out["pos"][:, :3], p["pos"][:, :3])   # dt = 0: positions unchanged


def test_kat_two_particle_pair_force(oracle):
    """Constant force curve c (coeffs[0]): equal and opposite directions, each
    acceleration scaled by the OTHER particle's mass (wgsl:134-139)."""
    sp = pa.SimParams()
    sp.delta_t = F(0.5)
    sp.masses = (2.0, 3.0, 1.0, 1.0, 1.0)
    sp.polys[:] = 0
    sp.set_poly(0, 1, pa.Poly7.const_val(0.25))
    sp.set_poly(1, 0, pa.Poly7.const_val(0.25))
    g = np.zeros((1000, 4), dtype=F)
    p = pa.empty_particles(2)
    p["pos"][0][:3] = (0.0, 0.0, 0.0); p["ty"][0] = 0
    p["pos"][1][:3] = (0.5, 0.0, 0.0); p["ty"][1] = 1
    out = oracle.step(sp, p, g)
    a0 = F(F(F(1.0) * F(0.25)) * F(3.0))      # dirN.x * poly * m2 * pow(len, 0)
    a1 = F(F(F(-1.0) * F(0.25)) * F(2.0))
    assert out["vel"][0][0] == F(a0 * F(0.5)) and out["vel"][1][0] == F(a1 * F(0.5))
    assert out["vel"][0][1] == 0 and out["vel"][1][2] == 0


def test_kat_pair_cutoffs(oracle):
    """Pairs closer than 0.001 or farther than cut_off_distance are skipped (wgsl:126-131)."""
    sp = pa.SimParams()
    sp.delta_t = F(0.5)
    sp.polys[:] = 0
    for a in range(5):
        for b in range(5):
            sp.set_poly(a, b, pa.Poly7.const_val(1.0))
    g = np.zeros((1000, 4), dtype=F)
    p = pa.empty_particles(3)
    p["pos"][1][:3] = (0.0005, 0.0, 0.0)     # too close to particle 0
    p["pos"][2][:3] = (1.5, 0.0, 0.0)        # beyond cut_off 1.0 of both
    out = oracle.step(sp, p, g)
    assert np.all(out["vel"][:, :3] == 0)
    assert not oracle.pair_term_inert(sp)
    sp.cut_off_distance = 0.0005
    assert oracle.pair_term_inert(sp) and sp.pair_term_inert()


def test_kat_zero_polys_pair_loop_is_inert(oracle):
    """With every coefficient zero the pair loop adds exactly +0, so running it
    and skipping it agree bit for bit (wgsl:139)."""
    sp = pa.SimParams()
    sp.delta_t = DT
    assert oracle.pair_term_inert(sp) and sp.pair_term_inert()
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(400, seed=1, pos_extent=2.0)
    a = oracle.run(sp, p, g, 3, mode=oracle.MODE_ALWAYS_PAIRS)
    b = oracle.run(sp, p, g, 3, mode=oracle.MODE_SKIP_INERT)
    assert a.tobytes() == b.tobytes()
    sp.polys[3, 2] = -0.0
    assert sp.pair_term_inert()
    sp.polys[3, 2] = 1e-30
    assert not sp.pair_term_inert()


def test_kat_ping_pong_parity(oracle):
    """Odd and even step counts land in the right buffer (compute.rs:147-180,221-227)."""
    sp, p, g = default_scene()
    sp.delta_t = DT
    s1 = oracle.step(sp, p, g)
    s2 = oracle.step(sp, s1, g)
    s3 = oracle.step(sp, s2, g)
    assert oracle.run(sp, p, g, 1).tobytes() == s1.tobytes()
    assert oracle.run(sp, p, g, 2).tobytes() == s2.tobytes()
    assert oracle.run(sp, p, g, 3).tobytes() == s3.tobytes()


def test_threads_and_ranges_agree(oracle):
    sp = pa.SimParams()
    sp.delta_t = DT
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(5000, seed=2, pos_extent=20.0)
    one = oracle.step(sp, p, g, mode=oracle.MODE_SKIP_INERT, threads=1)
    many = oracle.step(sp, p, g, mode=oracle.MODE_SKIP_INERT, threads=4)
    assert one.tobytes() == many.tobytes()
    assert oracle.step_range(sp, p, g, 1234, 777, mode=oracle.MODE_SKIP_INERT).tobytes() == one[1234:2011].tobytes()


# ---- independent restatement --------------------------------------------

@pytest.mark.parametrize("seed,dims", [(1, (10, 10, 10)), (2, (16, 8, 4)), (3, (64, 64, 1))])
def test_numpy_restatement_agrees_stream(oracle, seed, dims):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = dims
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    rng = np.random.default_rng(seed)
    g = np.ones((dims[0] * dims[1] * dims[2], 4), dtype=F)
    g[:, :3] = rng.uniform(-2, 2, size=(g.shape[0], 3)).astype(F)
    p = oracle.synth_particles(3000, seed=seed, pos_extent=20.0, vel_extent=10.0).view(pa.PARTICLE_DTYPE)
    cur = p
    for _ in range(4):
        got, fgi = NP.step(sp, cur, g, exp_neg(DT), pairs=False)
        assert np.array_equal(fgi, oracle.cell_indices(sp, cur, g.shape[0]))
        want = oracle.step(sp, cur, g, mode=oracle.MODE_SKIP_INERT).view(pa.PARTICLE_DTYPE)
        assert_particles_bit_equal(got, want, "numpy restatement vs C oracle")
        cur = want


def test_numpy_restatement_agrees_pairs(oracle):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.cut_off_distance = 1.5
    rng = np.random.default_rng(5)
    sp.polys[:] = rng.uniform(-2, 2, size=(25, 8)).astype(F)
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(300, seed=9, pos_extent=4.0, vel_extent=10.0).view(pa.PARTICLE_DTYPE)
    got, _ = NP.step(sp, p, g, exp_neg(DT), pairs=True)
    want = oracle.step(sp, p, g).view(pa.PARTICLE_DTYPE)
    assert_particles_bit_equal(got, want, "pair term, exponent 0")
    sp.distance_exponent = -1.5          # powf: numpy and libm may differ in the last place
    got, _ = NP.step(sp, p, g, exp_neg(DT), pairs=True)
    want = oracle.step(sp, p, g).view(pa.PARTICLE_DTYPE)
    for f in ("pos", "vel"):
        assert np.allclose(got[f], want[f], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("seed", range(48))
def test_numpy_restatement_agrees_over_the_gui_parameter_ranges(oracle, seed):
    """The parameter domain of the reference's GUI (src/gui.rs:130-318; the generator of the GPU sweep in
    test_gpu_random_params.py), field shapes from one cell to 64 per axis, time steps from 0 to 0.19, particles
    beyond the bounding cube and speeds on both sides of the clamp: three steps, every one bit for bit
    (every fourth seed with the pair term, exponent 0)."""
    from test_gpu_random_params import random_case
    rng = np.random.default_rng(7000 + seed)
    pairs = seed % 4 == 3
    sp, g = random_case(rng, pairs)
    r = float(sp.bounding_volume_radius)
    n = 300 if pairs else 2000
    cur = oracle.synth_particles(n, seed=seed, pos_extent=(min(2.0 * r, 6.0) if pairs else 2.4 * r),
                                 vel_extent=float(rng.choice([0.0, 1.0, 10.0, 400.0]))).view(pa.PARTICLE_DTYPE)
    for step in range(3):
        got, fgi = NP.step(sp, cur, g, exp_neg(sp.delta_t), pairs=pairs)
        assert np.array_equal(fgi, oracle.cell_indices(sp, cur, g.shape[0]))
        want = oracle.step(sp, cur, g, mode=oracle.MODE_ALWAYS_PAIRS if pairs else oracle.MODE_SKIP_INERT).view(pa.PARTICLE_DTYPE)
        assert_particles_bit_equal(got, want, f"seed {seed} step {step}")
        cur = want


def test_synthetic_generator_distribution(oracle):
    """pos_extent*(u-0.5) with u in [0,1): the distribution of src/main.rs:41-48."""
    p = oracle.synth_particles(200000, seed=4, pos_extent=2.0, vel_extent=10.0)
    assert p["pos"][:, :3].min() >= -1.0 and p["pos"][:, :3].max() < 1.0
    assert p["vel"][:, :3].min() >= -5.0 and p["vel"][:, :3].max() < 5.0
    assert abs(p["pos"][:, :3].mean()) < 0.01 and abs(p["pos"][:, :3].std() - 2 / math.sqrt(12)) < 0.01
    assert np.array_equal(p["ty"], np.arange(200000) % 5)
    q = oracle.synth_particles(100, seed=4, first_global_index=1000, pos_extent=2.0, vel_extent=10.0)
    assert q.tobytes() == p[1000:1100].tobytes()
