"""The K-step comparison policy of conftest.py (tolerance + straddler
accounting) exercised on the CPU: the C oracle against the independent numpy
restatement, whose float32 `power` rounds differently from glibc's powf in the
last place — the same kind of difference the GPU's powf has.  Strong forces
and a short cut-off make particles flip across the pair cut-off within 16
steps; the policy must explain every particle out of tolerance, and must
refuse a difference that no discontinuity explains."""
import math

import numpy as np
import pytest

import particles_amd as pa
import np_restatement as R
from conftest import compare_k_steps, straddlers_on_trajectory


def strong_pair_scene():
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.cut_off_distance = 1.5
    sp.distance_exponent = -2.0
    rng = np.random.default_rng(99)
    sp.polys[:] = rng.uniform(-2.0, 2.0, size=(25, 8)).astype(np.float32)
    return sp


@pytest.fixture(scope="module")
def two_trajectories(oracle):
    sp = strong_pair_scene()
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(600, seed=6, pos_extent=6.0, vel_extent=10.0)
    want, direct, tainted = straddlers_on_trajectory(oracle, sp, p, g, 16)
    decay = np.float32(math.exp(-float(np.float32(sp.delta_t))))
    cur = p
    for _ in range(16):
        cur, _ = R.step(sp, cur, g, decay)
    return sp, cur, want, direct, tainted


def test_policy_explains_every_flip(two_trajectories):
    sp, got, want, direct, tainted = two_trajectories
    rep = compare_k_steps(got, want, 2.0 * float(sp.bounding_volume_radius), direct, tainted,
                          budget=got.shape[0], what="oracle vs numpy restatement, 16 steps, strong forces")
    assert rep["out_of_tolerance"] > 0            # the scene does flip particles: the policy is exercised
    assert rep["out_of_tolerance_unexplained"] == 0


def test_policy_refuses_an_unexplained_difference(two_trajectories):
    sp, got, want, direct, tainted = two_trajectories
    clean = np.nonzero(~direct & ~tainted)[0]
    assert clean.size > 0
    bad = got.copy()
    bad["vel"][clean[0], 0] += np.float32(0.01)
    with pytest.raises(AssertionError, match="straddle nothing"):
        compare_k_steps(bad, want, 2.0 * float(sp.bounding_volume_radius), direct, tainted, budget=got.shape[0])


def test_policy_enforces_the_budget(two_trajectories):
    sp, got, want, direct, tainted = two_trajectories
    with pytest.raises(AssertionError, match="budget"):
        compare_k_steps(got, want, 2.0 * float(sp.bounding_volume_radius), direct, tainted, budget=0)
