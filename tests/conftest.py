import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/particles_oracle.c); builds it if needed."""
    from oracle import oracle as O
    O.load()
    return O


def bits(a):
    """Raw bytes of a float/record array, for bit-exact comparison."""
    return np.ascontiguousarray(a).view(np.uint8)


def assert_particles_bit_equal(got, want, what=""):
    """pos.xyzw, vel.xyzw and ty identical bit for bit."""
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    for f in ("pos", "vel"):
        g = np.ascontiguousarray(got[f]).view(np.uint32)
        w = np.ascontiguousarray(want[f]).view(np.uint32)
        bad = np.nonzero((g != w).any(axis=1))[0]
        assert bad.size == 0, (f"{what}: {bad.size} particles differ in {f}; first {bad[0]}: "
                               f"got {got[f][bad[0]]} want {want[f][bad[0]]}")
    assert np.array_equal(got["ty"], want["ty"]), f"{what}: ty differs"


def scene_swirl_field(nx, ny, nz, radius, seed):
    """Synthetic field of the benchmark configs: the swirl (-y, x, 0)/r sampled
    at the cell centres (centres as in src/grid.rs:119-138) plus a seeded
    uniform [-1, 1) perturbation; w = 1."""
    import particles_amd as pa
    r = np.float32(radius)
    g = pa.Grid.new_uniform(nx, ny, nz, pa.Bounds((-r, -r, -r), (2 * r, 2 * r, 2 * r)))
    pos = g.get_positions()
    rng = np.random.default_rng(seed)
    out = np.empty((nx * ny * nz, 4), dtype=np.float32)
    out[:, 0] = -pos[:, 1] / r
    out[:, 1] = pos[:, 0] / r
    out[:, 2] = 0.0
    out[:, :3] += rng.uniform(-1.0, 1.0, size=(nx * ny * nz, 3)).astype(np.float32)
    out[:, 3] = 1.0
    return out
