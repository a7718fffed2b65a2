"""Field brush (src/cursor.rs:79-122,158-203) on the host, and the dirty-list
patch of the device field it feeds."""
import numpy as np
import pytest

import particles_amd as pa
from conftest import assert_particles_bit_equal

F = np.float32


def test_default_edit_mode_matches_reference_defaults():
    m = pa.EditMode()                       # src/cursor.rs:206-216
    assert (m.mode, m.relative, m.falloff, float(m.falloff_dist), float(m.strength)) == ("shift", True, "abrupt", 1.0, 1.0)


def test_mouse_down_selects_cells_within_falloff_distance():
    sp = pa.SimParams()
    grid = sp.new_force_grid_centered()
    b = pa.Brush(pa.EditMode(falloff_dist=1.8))
    b.mouse_down(grid, (0.0, 0.0, 0.0))
    pos = grid.get_positions()[:, :3]
    want = np.nonzero(np.linalg.norm(pos, axis=1) < 1.8)[0]     # src/cursor.rs:109-114
    assert np.array_equal(b.modify_vector_indices, want) and len(want) == 8
    assert np.array_equal(b.mouse_down_vectors, grid.grid[want])


def test_shift_relative_adds_cursor_displacement():
    """Shift/Relative/Abrupt: cell = cell_at_mouse_down + (cursor - md_pos)*strength
    (src/cursor.rs:98,174,194,202)."""
    sp = pa.SimParams()
    grid = sp.new_force_grid_centered()
    before = grid.grid.copy()
    b = pa.Brush(pa.EditMode(falloff_dist=2.5, strength=0.5))
    b.mouse_down(grid, (1.0, 1.0, 1.0))
    idx, cells = b.mouse_moved(grid, (1.5, 1.0, 0.0))
    assert idx.dtype == np.uint32 and cells.shape == (len(idx), 4) and np.all(cells[:, 3] == 1.0)
    want = before[idx] + F(0.5) * np.array([0.5, 0.0, -1.0], dtype=F)
    assert np.array_equal(cells[:, :3], want) and np.array_equal(grid.grid[idx], want)
    untouched = np.setdiff1d(np.arange(1000), idx)
    assert np.array_equal(grid.grid[untouched], before[untouched])
    # a second move is relative to the mouse-down vectors, not cumulative
    idx2, cells2 = b.mouse_moved(grid, (1.0, 1.0, 1.0))
    assert np.array_equal(cells2[:, :3], before[idx2])
    b.mouse_up()
    assert b.mouse_moved(grid, (2.0, 2.0, 2.0))[0].size == 0


def test_centered_absolute_points_at_cursor_and_falloffs():
    sp = pa.SimParams()
    grid = sp.new_force_grid_zero()
    pos = grid.get_positions()[:, :3]
    for falloff in ("abrupt", "linear", "inverse_distance"):
        b = pa.Brush(pa.EditMode(mode="centered", relative=False, falloff=falloff, falloff_dist=3.0, strength=2.0))
        b.mouse_down(grid, (0.3, 0.2, 0.1))
        idx, cells = b.mouse_moved(grid, (0.4, -0.2, 0.6))
        d = np.array([0.4, -0.2, 0.6], dtype=F)[None, :] - pos[idx]
        unit = d / np.linalg.norm(d, axis=1, keepdims=True)
        if falloff == "abrupt":
            want = unit * 2.0
        elif falloff == "linear":
            mag = np.linalg.norm(np.array([0.3, 0.2, 0.1], dtype=F)[None, :] - pos[idx], axis=1)
            want = (1.0 - mag / 3.0)[:, None] * unit * 2.0
        else:
            want = (3.0 / (1.0 + 1.0)) * unit * 2.0            # |unit| = 1
        assert np.allclose(cells[:, :3], want, rtol=1e-5, atol=1e-6)


def test_rotate_mode():
    sp = pa.SimParams()
    grid = sp.new_force_grid_centered()
    b = pa.Brush(pa.EditMode(mode="rotate", relative=False, falloff_dist=2.5))
    b.mouse_down(grid, (0.0, 0.0, 0.0))
    before = grid.grid[b.modify_vector_indices].copy()
    idx, cells = b.mouse_moved(grid, (0.0, 0.0, 0.0))          # cursor == md_pos: vectors kept
    assert np.array_equal(cells[:, :3], before)
    idx, cells = b.mouse_moved(grid, (0.7, 0.0, 0.0))
    pos = grid.get_positions()[idx, :3]
    dv = pos / np.linalg.norm(pos, axis=1, keepdims=True)
    want = np.cross(dv, np.array([0.0, 0.0, 1.0])) * 0.7        # dir = dv x cam_dir, times dc . cam_right
    assert np.allclose(cells[:, :3], want, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_brush_patch_reaches_the_device(oracle):
    sp = pa.SimParams()
    sp.delta_t = F(1.0 / 60.0)
    p = pa.lattice_particles((5.0, 2.0, 2.0), 10, 10, 10)
    grid = sp.new_force_grid_centered()
    with pa.Compute.new(p, grid.get_force_vectors()) as c:
        c.update_sim_params(sp)
        b = pa.Brush(pa.EditMode(falloff_dist=3.0))
        b.mouse_down(grid, (2.0, 1.0, 1.0))
        cur = p
        for k in range(4):
            idx, cells = b.mouse_moved(grid, (2.0 + 0.3 * k, 1.0, 1.0 - 0.2 * k))
            c.patch_force_grid_indexed(idx, cells)
            c.compute()
            cur = oracle.step(sp, cur, grid.get_force_vectors())
        assert_particles_bit_equal(c.read_particles(), cur, "brush edits")
        with pytest.raises(pa.PclError):
            c.patch_force_grid_indexed([1000], np.ones((1, 4), dtype=F))
