"""Committed fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py
with the CPU oracle).  CPU: the oracle still reproduces them.  GPU: the HIP
path reproduces them — bit for bit, except the one fixture that exercises
powf, which is held to the 1e-5 relative tolerance."""
import glob
import os

import numpy as np
import pytest

import particles_amd as pa
from conftest import assert_particles_bit_equal

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(path):
    z = np.load(path)
    sp = pa.SimParams()
    sp.raw = np.frombuffer(z["params"].tobytes(), dtype=pa.SIM_PARAMS_DTYPE)[0].copy()
    p0 = np.frombuffer(z["particles0"].tobytes(), dtype=pa.PARTICLE_DTYPE).copy()
    want = np.frombuffer(z["expected"].tobytes(), dtype=pa.PARTICLE_DTYPE).copy()
    return sp, p0, z["grid"], int(z["steps"]), int(z["mode"]), want, z["cells0"]


def test_fixtures_exist():
    assert len(FIXTURES) >= 4


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_fixture(oracle, path):
    sp, p0, g, steps, mode, want, cells0 = load(path)
    got = oracle.run(sp, p0, g, steps, mode=mode)
    assert got.tobytes() == want.tobytes()
    assert np.array_equal(oracle.cell_indices(sp, p0, g.shape[0]), cells0)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_reproduces_fixture(path):
    sp, p0, g, steps, mode, want, cells0 = load(path)
    with pa.Compute.new(p0, g) as c:
        c.update_sim_params(sp)
        assert np.array_equal(c.debug_cell_indices(), cells0)          # index work: bit-exact
        for _ in range(steps):
            c.compute()
        got = c.read_particles()
    if float(sp.distance_exponent) == 0.0 or sp.pair_term_inert():
        assert_particles_bit_equal(got, want, os.path.basename(path))
    else:
        two_r = 2.0 * sp.bounding_volume_radius
        for f in ("pos", "vel"):
            a, b = got[f][:, :3].astype(np.float64), want[f][:, :3].astype(np.float64)
            d = np.abs(a - b)
            if f == "pos":
                d = np.minimum(d, np.abs(two_r - d))                    # modulo the wrap
            assert np.all(d <= 1e-5 * np.maximum(np.abs(a), np.abs(b)) + 1e-6)
