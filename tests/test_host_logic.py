"""Host-side data model: Poly7 (src/poly7.rs), Grid (src/grid.rs), SimParams
(src/sim_params.rs), particle generators (src/main.rs)."""
import numpy as np
import pytest

import particles_amd as pa

F = np.float32


def test_poly7_basics():
    assert pa.Poly7.COEFF_NAMES == ("h", "g", "f", "e", "d", "c", "b", "a")
    p = pa.Poly7.const_val(2.5)
    assert p.coeffs[0] == F(2.5) and not p.coeffs[1:].any()
    p.invert()
    assert p.coeffs[0] == F(-2.5)
    q = pa.Poly7([1, 2, 3, 0, 0, 0, 0, 0])
    assert q.eval(2.0) == F(1 + 4 + 12)
    assert not pa.Poly7.zero().coeffs.any()


def test_poly7_from_points_interpolates():
    """src/poly7.rs:37-58: the polynomial passes through the 8 points."""
    xs = np.linspace(0.0, 1.0, 8, dtype=F)
    ys = np.array([0.0, 1.0, -0.5, 2.0, 0.3, -1.0, 0.7, 0.1], dtype=F)
    p = pa.Poly7.from_points(np.stack([xs, ys], axis=1))
    assert p is not None
    for x, y in zip(xs, ys):
        # the reference inverts the fp32 Vandermonde matrix without pivoting: ill-conditioned
        assert abs(float(p.eval(x)) - float(y)) < 0.5
    # points on a low-degree polynomial are recovered closely: y = x^2 - 0.5 x
    ys2 = (xs * xs - F(0.5) * xs).astype(F)
    q = pa.Poly7.from_points(np.stack([xs, ys2], axis=1))
    for x, y in zip(xs, ys2):
        assert abs(float(q.eval(x)) - float(y)) < 1e-3


def test_poly7_from_points_rejects_repeated_x():
    pts = np.zeros((8, 2), dtype=F)
    pts[:, 0] = [0, 0.1, 0.2, 0.2, 0.4, 0.5, 0.6, 0.7]
    assert pa.Poly7.from_points(pts) is None


def test_grid_constructors_and_layout():
    sp = pa.SimParams()
    g = sp.new_force_grid_centered()
    assert g.size == (10, 10, 10) and g.num_instances() == 1000
    # x slowest, z fastest; v = (n/2) - index  (src/grid.rs:68-95)
    assert tuple(g.grid[0]) == (5.0, 5.0, 5.0)
    assert tuple(g.grid[1]) == (5.0, 5.0, 4.0)
    assert tuple(g.grid[755]) == (-2.0, 0.0, 0.0)
    fv = g.get_force_vectors()
    assert fv.shape == (1000, 4) and np.all(fv[:, 3] == 1.0)
    z = sp.new_force_grid_zero()
    assert not z.grid.any()
    pos = g.get_positions()
    assert tuple(pos[0]) == (-9.0, -9.0, -9.0, 1.0) and tuple(pos[-1]) == (9.0, 9.0, 9.0, 1.0)
    idx = g.get_indices((0.0, 0.0, 0.0), 1.8)     # the 8 cell centres at (+-1,+-1,+-1), distance 1.73
    assert len(idx) == 8
    with pytest.raises(ValueError):
        pa.Grid(np.zeros((5, 3)), (2, 2, 2), g.bounds)


def test_sim_params_fields_roundtrip():
    sp = pa.SimParams()
    assert sp.delta_t == 0.0 and sp.max_velocity == 100.0 and sp.bounding_volume_radius == 10.0
    assert sp.cut_off_distance == 1.0 and sp.distance_exponent == 0.0
    assert sp.force_grid_dimensions == (10, 10, 10) and np.all(sp.masses == 1.0)
    sp.force_grid_dimensions = (256, 256, 1)
    sp.masses = (1, 2, 3, 4, 5)
    sp.set_poly(2, 3, pa.Poly7.const_val(1.0))
    raw = np.frombuffer(sp.tobytes(), dtype=np.uint8)
    assert raw[880:892].view(np.uint32).tolist() == [256, 256, 1]
    assert raw[800:880].view(np.float32)[::4].tolist() == [1, 2, 3, 4, 5]
    assert raw[:800].view(np.float32)[(3 + 5 * 2) * 8] == 1.0        # table index ty_other + 5*ty_self
    c = sp.copy()
    c.delta_t = 0.5
    assert sp.delta_t == 0.0


def test_random_particles_distribution():
    p = pa.random_particles(50000, np.random.default_rng(0), first_index=3)
    assert p["pos"][:, :3].min() >= -1.0 and p["pos"][:, :3].max() < 1.0      # rand_v4(2.0)
    assert p["vel"][:, :3].min() >= -5.0 and p["vel"][:, :3].max() < 5.0      # rand_v4(10.0)
    assert np.all(p["pos"][:, 3] == 1.0) and np.all(p["vel"][:, 3] == 1.0)
    assert p["ty"][0] == 3 and p["ty"][2] == 0
