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


def _abi_from_points(pts):
    import ctypes as C
    from particles_amd import _lib
    pts = np.ascontiguousarray(pts, dtype=F).reshape(8, 2)
    out = np.zeros(8, dtype=F)
    rc = _lib.load().pcl_poly7_from_points(pts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return rc, out


def test_poly7_from_points_equals_the_oracle_bit_for_bit(oracle):
    """src/poly7.rs:37-58,120-147 restated twice (host C++ behind the C ABI, C in
    the oracle): same powi, same elimination order, same bits, over sample sets
    the GUI can produce (x in [0, 1], values in [-10, 10], src/gui.rs:236-288)
    and over hostile ones (clustered, negative, large x)."""
    from particles_amd import _lib
    rng = np.random.default_rng(2)
    cases = []
    for k in range(200):
        xs = np.sort(rng.uniform(0.0, 1.0, 8)).astype(F)
        cases.append(np.stack([xs, rng.uniform(-10, 10, 8).astype(F)], axis=1))
    for k in range(50):
        xs = rng.uniform(-3.0, 3.0, 8).astype(F)
        cases.append(np.stack([xs, rng.uniform(-10, 10, 8).astype(F)], axis=1))
    cases.append(np.stack([np.linspace(0.0, 1.0, 8, dtype=F), np.zeros(8, dtype=F)], axis=1))
    cases.append(np.stack([F(0.5) + np.arange(8, dtype=F) * F(1e-3), np.ones(8, dtype=F)], axis=1))   # nearly singular
    cases.append(np.stack([np.arange(8, dtype=F) * F(50.0), np.arange(8, dtype=F)], axis=1))          # x^7 up to 6e17
    n_some = 0
    for pts in cases:
        rc, got = _abi_from_points(pts)
        want = oracle.poly7_from_points(pts)
        if want is None:
            assert rc == _lib.PCL_ERR_SINGULAR
            assert pa.Poly7.from_points(pts) is None
        else:
            assert rc == 0 and got.tobytes() == want.tobytes(), (pts, got, want)
            assert pa.Poly7.from_points(pts).coeffs.tobytes() == want.tobytes()
            n_some += 1
    assert n_some >= 250


def test_poly7_from_points_known_answers(oracle):
    """Hand-checkable cases of src/poly7.rs:37-58.  With x = 0, 1, ..., 7 every
    entry of the Vandermonde matrix and every intermediate of the elimination
    is a small rational; points on y = 3 are recovered as the constant 3 up to
    fp32 round-off of that elimination, and the polynomial reproduces the
    sample values it was built from."""
    xs = np.arange(8, dtype=F)
    rc, c = _abi_from_points(np.stack([xs, np.full(8, 3.0, dtype=F)], axis=1))
    assert rc == 0 and abs(float(c[0]) - 3.0) < 1e-4 and np.all(np.abs(c[1:]) < 1e-3)
    # the first row of the inverse is e_0 exactly (x_0 = 0 makes row 0 of the matrix e_0): c[0] = y_0 bit for bit
    ys = np.array([0.25, 1.0, -0.5, 2.0, 0.3, -1.0, 0.7, 0.1], dtype=F)
    rc, c = _abi_from_points(np.stack([np.linspace(0.0, 1.0, 8, dtype=F), ys], axis=1))
    assert rc == 0 and c[0] == ys[0]
    # the GUI's own use: 8 samples on [0, 1]; the curve goes through them to the accuracy the
    # unpivoted fp32 elimination allows, which the oracle's restatement defines exactly
    p = pa.Poly7(c)
    want = oracle.poly7_from_points(np.stack([np.linspace(0.0, 1.0, 8, dtype=F), ys], axis=1))
    assert p.coeffs.tobytes() == want.tobytes()
    for x in np.linspace(0.0, 1.0, 8, dtype=F):
        assert p.eval(x) == oracle.poly7_eval_host(want, x)          # Poly7::eval, src/poly7.rs:60-68
    # low-degree data is recovered closely: y = x^2 - 0.5 x
    xs = np.linspace(0.0, 1.0, 8, dtype=F)
    ys2 = (xs * xs - F(0.5) * xs).astype(F)
    q = pa.Poly7.from_points(np.stack([xs, ys2], axis=1))
    for x, y in zip(xs, ys2):
        assert abs(float(q.eval(x)) - float(y)) < 1e-3


def test_abi_grid_constructors_equal_the_oracle(oracle):
    """pcl_grid_centered / _uniform / _positions and pcl_sim_params_force_grid
    (src/grid.rs:49-103,119-138, src/sim_params.rs:33-60) against the oracle's
    generators and the numpy host mirror, byte for byte."""
    import ctypes as C
    from particles_amd import _lib
    lib = _lib.load()
    for dims in ((10, 10, 10), (7, 3, 2), (256, 256, 1), (1, 1, 1)):
        n = dims[0] * dims[1] * dims[2]
        out = np.zeros((n, 4), dtype=F)
        lib.pcl_grid_centered(*dims, out.ctypes.data_as(C.c_void_p))
        assert out.tobytes() == oracle.grid_centered(*dims).tobytes()
        sp = pa.SimParams()
        sp.force_grid_dimensions = dims
        sp.bounding_volume_radius = 2.5
        assert out.tobytes() == sp.new_force_grid_centered().get_force_vectors().tobytes()
        out2 = np.zeros((n, 4), dtype=F)
        raw = np.ascontiguousarray(sp.raw)
        assert lib.pcl_sim_params_force_grid(raw.ctypes.data_as(C.c_void_p), 1, out2.ctypes.data_as(C.c_void_p), n) == 0
        assert out2.tobytes() == out.tobytes()
        assert lib.pcl_sim_params_force_grid(raw.ctypes.data_as(C.c_void_p), 0, out2.ctypes.data_as(C.c_void_p), n) == 0
        assert out2.tobytes() == sp.new_force_grid_zero().get_force_vectors().tobytes() == oracle.grid_uniform(*dims).tobytes()
        if n > 1:
            assert lib.pcl_sim_params_force_grid(raw.ctypes.data_as(C.c_void_p), 1, out2.ctypes.data_as(C.c_void_p), n - 1) == _lib.PCL_ERR_OUT_OF_RANGE
        v = np.array([0.5, -1.0, 2.0], dtype=F)
        lib.pcl_grid_uniform(*dims, v.ctypes.data_as(C.c_void_p), out2.ctypes.data_as(C.c_void_p))
        assert out2.tobytes() == oracle.grid_uniform(*dims, value=v).tobytes()
        pos, dirv = np.full(3, -2.5, dtype=F), np.full(3, 5.0, dtype=F)
        lib.pcl_grid_positions(*dims, pos.ctypes.data_as(C.c_void_p), dirv.ctypes.data_as(C.c_void_p), out2.ctypes.data_as(C.c_void_p))
        assert out2.tobytes() == oracle.grid_positions(*dims, 2.5).tobytes() == sp.new_force_grid_centered().get_positions().tobytes()
    p = pa.Poly7([1, 2, 3, 0, 0.5, 0, 0, -1])
    pc = np.ascontiguousarray(p.coeffs)
    for x in (0.0, 0.3, 1.0, -2.0):
        assert F(lib.pcl_poly7_eval(pc.ctypes.data_as(C.c_void_p), x)) == p.eval(x) == oracle.poly7_eval_host(pc, x)


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
