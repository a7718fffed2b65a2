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


# ---------------------------------------------------------------------------
# K-step comparison policy for paths whose arithmetic is not bit-reproducible
# (powf; the cell-binned pair term, whose summation order differs).
#
# Bar: |a - b| <= 1e-5 * max(|a|, |b|) + 1e-6 per component, positions compared
# modulo 2r (circular distance).  The update has discontinuities: the fetched
# field cell (src/compute.wgsl:81-89,147), the wrap plane (:58-66), the speed
# threshold 0.001 (:157) and the two pair cut-offs (:126-131).  A state that is
# within the bar of the oracle's may sit on the other side of one of them, and
# then differs by O(1) afterwards.  Such particles are STRADDLERS: they are
# found on the oracle's own trajectory (a discontinuity closer than the error
# the bar admits), counted and held to a stated budget; they are never dropped
# silently, and a particle out of tolerance that is not one fails the test.
# ---------------------------------------------------------------------------

RTOL, ATOL = 1e-5, 1e-6
STRADDLE_MARGIN = 4.0      # a discontinuity within 4x the admitted error counts as straddled


def _admitted(x):
    """State error the bar admits at value(s) x."""
    return RTOL * np.abs(x) + ATOL


def straddlers_on_trajectory(oracle, sp, p0, grid, steps, mode=None, pair_block=512):
    """Steps the oracle `steps` times from p0 and marks, per particle,
      direct[i]   at some step a discontinuity of i's own update lay within
                  STRADDLE_MARGIN x the admitted error of its trajectory;
      tainted[i]  (pair term only) at some step i was within the cut-off of a
                  particle marked `direct` at an earlier step, so one of its
                  force terms may have come from a state that is O(1) off
                  (one level only: partners of tainted particles are not
                  marked, their error is second order in delta_t).
    Returns (final state, direct, tainted)."""
    mode = oracle.MODE_ALWAYS_PAIRS if mode is None else mode
    n = p0.shape[0]
    r = float(sp.bounding_volume_radius)
    dims = [int(d) for d in sp.force_grid_dimensions]
    cut = float(sp.cut_off_distance)
    pairs = not sp.pair_term_inert()
    direct = np.zeros(n, dtype=bool)
    tainted = np.zeros(n, dtype=bool)
    cur = np.array(p0, copy=True)
    for _ in range(steps):
        pos = cur["pos"][:, :3].astype(np.float64)
        m = STRADDLE_MARGIN * _admitted(pos)
        new = np.zeros(n, dtype=bool)
        # the cell this step fetches: distance to the nearest cell face per axis
        for a in range(3):
            if dims[a] > 1 and r > 0.0:
                cell = 2.0 * r / dims[a]
                t = (np.clip(pos[:, a], -r, r) + r) / cell
                new |= np.abs(t - np.rint(t)) * cell <= m[:, a]
        off_before = direct.copy()
        if pairs and n > 1:
            mp = m.max(axis=1)
            for b0 in range(0, n, pair_block):
                d = pos[None, :, :] - pos[b0:b0 + pair_block, None, :]
                ln = np.sqrt((d * d).sum(axis=2))
                ln[np.arange(ln.shape[0]), np.arange(b0, b0 + ln.shape[0])] = np.inf   # j == i
                mm = mp[b0:b0 + pair_block, None] + mp[None, :]
                near = (np.abs(ln - cut) <= mm) | (np.abs(ln - 0.001) <= mm)
                new[b0:b0 + pair_block] |= near.any(axis=1)
                within = ln <= cut + mm
                tainted[b0:b0 + pair_block] |= (within & off_before[None, :]).any(axis=1)
        nxt = oracle.step(sp, cur, grid, mode=mode)
        npos = nxt["pos"][:, :3].astype(np.float64)
        nvel = nxt["vel"][:, :3].astype(np.float64)
        # the wrap plane: a position that ends within the margin of +-r may have wrapped or not
        new |= ((r - np.abs(npos)) <= STRADDLE_MARGIN * _admitted(npos)).any(axis=1)
        # the speed threshold of the renormalisation
        speed = np.sqrt((nvel * nvel).sum(axis=1))
        new |= np.abs(speed - 0.001) <= STRADDLE_MARGIN * _admitted(speed)
        direct |= new
        cur = nxt
    return cur, direct, tainted & ~direct


def compare_k_steps(got, want, two_r, direct, tainted, budget, what=""):
    """The K-step bar.  Returns a dict of counts (also printed).  Fails when a
    particle is out of tolerance without being a straddler (or tainted by one),
    or when more than `budget` particles are out of tolerance at all."""
    n = got.shape[0]
    bad = np.zeros(n, dtype=bool)
    bad_components = 0
    worst = 0.0
    for f in ("pos", "vel"):
        a, b = got[f][:, :3].astype(np.float64), want[f][:, :3].astype(np.float64)
        d = np.abs(a - b)
        if f == "pos":
            d = np.minimum(d, np.abs(two_r - d))
        over = ~(d <= RTOL * np.maximum(np.abs(a), np.abs(b)) + ATOL)     # NaN counts as over
        bad_components += int(over.sum())
        bad |= over.any(axis=1)
        if over.any():
            worst = max(worst, float(np.nanmax(np.where(over, d, 0.0))))
    rep = {"particles": n, "straddlers": int(direct.sum()), "tainted": int(tainted.sum()),
           "out_of_tolerance": int(bad.sum()), "components_out_of_tolerance": bad_components,
           "out_of_tolerance_straddlers": int((bad & direct).sum()),
           "out_of_tolerance_tainted": int((bad & tainted).sum()),
           "out_of_tolerance_unexplained": int((bad & ~direct & ~tainted).sum()),
           "budget": int(budget), "worst_abs_diff": worst}
    print(f"[k-step parity] {what}: {rep}")
    assert np.array_equal(got["ty"], want["ty"]), f"{what}: ty differs"
    assert rep["out_of_tolerance_unexplained"] == 0, f"{what}: particles out of tolerance that straddle nothing: {rep}"
    assert rep["out_of_tolerance"] <= budget, f"{what}: straddler budget exceeded: {rep}"
    return rep


def binned_single_step_check(oracle, sp, p_all, g, got, want, first, what):
    """One step of the binned kernel against the oracle.  Every pair term is computed exactly as the
    oracle computes it; what differs is the order in which a particle's terms are added.  The bar is on
    the VECTORS, per particle and with no allowance:
        |dv| <= 1e-5 max(|v_got|, |v_want|) + 1e-6
        |dx| <= 1e-5 max(|x_got|, |x_want|) + 1e-6,   x modulo 2r per axis
    (a per-component bar fails for small components of a velocity that the speed clamp of wgsl:157-160
    holds at max_velocity: a reordered sum moves the direction of a long vector by eps of its length).
    For information the number of terms per particle and the size of the summation-order effect
    (2 (m - 1) 2^-24 sum|t| per component, from the oracle's own loop) are printed."""
    two_r = 2.0 * float(sp.bounding_volume_radius)
    n = got.shape[0]
    abs_sum, terms = oracle.pair_term_magnitudes(sp, p_all, first, n)
    da = 2.0 * np.maximum(terms.astype(np.float64) - 1.0, 0.0)[:, None] * 2.0 ** -24 * abs_sum

    def vec(f):
        a, b = got[f][:, :3].astype(np.float64), want[f][:, :3].astype(np.float64)
        d = np.abs(a - b)
        if f == "pos":
            d = np.minimum(d, np.abs(two_r - d))       # modulo the wrap
        return np.linalg.norm(d, axis=1), np.maximum(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1))

    dv, vmax = vec("vel")
    dx, xmax = vec("pos")
    print(f"[binned, one step] {what}: worst |dv| / bar {np.max(dv / (RTOL * vmax + ATOL)):.3g}, worst |dx| / bar "
          f"{np.max(dx / (RTOL * xmax + ATOL)):.3g}; up to {int(terms.max())} terms per particle, bound on the "
          f"acceleration's summation-order difference up to {np.linalg.norm(da, axis=1).max():.3g}")
    bad_v, bad_x = np.nonzero(dv > RTOL * vmax + ATOL)[0], np.nonzero(dx > RTOL * xmax + ATOL)[0]
    assert bad_v.size == 0, (f"{what}: {bad_v.size} velocities beyond 1e-5 max|v| + 1e-6; first {first + bad_v[0]}: "
                             f"|dv| {dv[bad_v[0]]:.3g} against |v| {vmax[bad_v[0]]:.3g}")
    assert bad_x.size == 0, (f"{what}: {bad_x.size} positions beyond 1e-5 max|x| + 1e-6; first {first + bad_x[0]}: "
                             f"|dx| {dx[bad_x[0]]:.3g} against |x| {xmax[bad_x[0]]:.3g}")
    assert np.array_equal(got["ty"], want["ty"])
