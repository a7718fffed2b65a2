"""Field brush: the editing half of the reference's cursor (src/cursor.rs).

``Brush.mouse_down`` selects the cells whose centre lies within
``falloff_dist`` of the cursor and remembers their vectors
(src/cursor.rs:105-116); ``Brush.mouse_moved`` rewrites exactly those cells
with ``EditMode.get_vector`` (src/cursor.rs:79-103,158-203) and returns the
dirty list ``(cell_indices, cells)`` so that only the edited cells travel to
the device (``Compute.patch_force_grid_indexed``) instead of the whole field
every frame (src/main.rs:371-380).  Camera picking (src/cursor.rs:36-57) is
windowing code and is not reproduced: the 3-D cursor position is an input.
"""
import numpy as np

F = np.float32

# The 64-byte record the device brush takes (pcl_brush_input).
BRUSH_INPUT_DTYPE = np.dtype([
    ("cursor_pos", np.float32, 3), ("falloff_dist", np.float32),
    ("cam_right", np.float32, 3), ("strength", np.float32),
    ("cam_dir", np.float32, 3), ("mode", np.int32),
    ("relative", np.int32), ("falloff", np.int32), ("reserved", np.int32, 2)])
assert BRUSH_INPUT_DTYPE.itemsize == 64


def _norm(v):
    """cgmath magnitude: sqrt(x*x + y*y + z*z), summed in that order."""
    v = np.asarray(v, dtype=F)
    return np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]).astype(F)


def _normalize(v):
    """cgmath normalize: v * (1 / magnitude(v))."""
    return (v * (F(1.0) / _norm(v))[..., None]).astype(F)


class EditMode:
    """src/cursor.rs:124-216; defaults of ``EditMode::default`` (:206-216)."""

    MODES = ("centered", "shift", "rotate")
    FALLOFFS = ("abrupt", "linear", "inverse_distance")

    def __init__(self, mode="shift", relative=True, falloff="abrupt", falloff_dist=1.0, strength=1.0):
        if mode not in self.MODES or falloff not in self.FALLOFFS:
            raise ValueError("unknown edit mode or falloff")
        self.mode, self.relative, self.falloff = mode, bool(relative), falloff
        self.falloff_dist, self.strength = F(falloff_dist), F(strength)

    def record(self, cursor_pos, cam_rot=None):
        """The device brush's input record for this mode, a cursor position and
        a rotation matrix whose columns are the camera's right/up/direction."""
        rot = np.eye(3, dtype=F) if cam_rot is None else np.asarray(cam_rot, dtype=F)
        r = np.zeros(1, dtype=BRUSH_INPUT_DTYPE)
        r["cursor_pos"], r["cam_right"], r["cam_dir"] = np.asarray(cursor_pos, dtype=F), rot[:, 0], rot[:, 2]
        r["falloff_dist"], r["strength"] = self.falloff_dist, self.strength
        r["mode"], r["falloff"] = self.MODES.index(self.mode), self.FALLOFFS.index(self.falloff)
        r["relative"] = int(self.relative)
        return r

    def get_vector(self, v, v_pos, md_pos, cursor_pos, cam_rot):
        """Vectorised over cells: ``v`` current vectors (n,3), ``v_pos`` their
        positions (n,3); ``md_pos`` the cursor at mouse-down, ``cursor_pos`` the
        cursor now, ``cam_rot`` a 3x3 matrix whose columns are the camera's
        right/up/direction axes (src/cursor.rs:158-203)."""
        v = np.asarray(v, dtype=F).reshape(-1, 3)
        v_pos = np.asarray(v_pos, dtype=F).reshape(-1, 3)
        md_pos = np.asarray(md_pos, dtype=F)
        cursor_pos = np.asarray(cursor_pos, dtype=F)
        cam_rot = np.asarray(cam_rot, dtype=F)
        cam_dir, cam_right = cam_rot[:, 2], cam_rot[:, 0]
        with np.errstate(all="ignore"):
            if self.mode == "centered":
                res = _normalize(cursor_pos[None, :] - v_pos)
            elif self.mode == "shift":
                res = np.broadcast_to(cursor_pos - md_pos, v.shape).astype(F)
            else:  # rotate
                if np.array_equal(cursor_pos, md_pos):
                    res = v.copy()
                else:
                    dc = cursor_pos - md_pos
                    dv = _normalize(v_pos - md_pos[None, :])
                    cross = np.stack([dv[:, 1] * cam_dir[2] - dv[:, 2] * cam_dir[1],
                                      dv[:, 2] * cam_dir[0] - dv[:, 0] * cam_dir[2],
                                      dv[:, 0] * cam_dir[1] - dv[:, 1] * cam_dir[0]], axis=1).astype(F)
                    res = cross * F((dc[0] * cam_right[0] + dc[1] * cam_right[1]) + dc[2] * cam_right[2])
            res_len = _norm(res)
            if self.falloff == "linear":
                mag = _norm(md_pos[None, :] - v_pos)
                res = (F(1.0) - mag / self.falloff_dist)[:, None] * res
            elif self.falloff == "inverse_distance":
                res = (self.falloff_dist / (res_len + F(1.0)))[:, None] * res
        return (res * self.strength).astype(F)


class Brush:
    """Selection and rewrite of field cells (src/cursor.rs:79-122)."""

    def __init__(self, edit_mode=None):
        self.edit_mode = edit_mode or EditMode()
        self.pos = np.zeros(3, dtype=F)
        self.rot = np.eye(3, dtype=F)
        self.mouse_down_on = None
        self.modify_vector_indices = np.zeros(0, dtype=np.int64)
        self.mouse_down_vectors = np.zeros((0, 3), dtype=F)
        self._positions = None

    def mouse_down(self, grid, pos, rot=None):
        """src/cursor.rs:105-116: select cells with |pos - centre| < falloff_dist."""
        self.pos = np.asarray(pos, dtype=F)
        if rot is not None:
            self.rot = np.asarray(rot, dtype=F)
        self.mouse_down_on = (self.pos.copy(), self.rot.copy())
        self._positions = grid.get_positions()[:, :3]
        self.modify_vector_indices = grid.get_indices(self.pos, self.edit_mode.falloff_dist)
        self.mouse_down_vectors = grid.grid[self.modify_vector_indices].copy()

    def mouse_moved(self, grid, pos):
        """src/cursor.rs:79-103: rewrite the selected cells in ``grid`` and
        return ``(cell_indices uint32, cells float32 (n,4))`` for the device."""
        self.pos = np.asarray(pos, dtype=F)
        if self.mouse_down_on is None or self.modify_vector_indices.size == 0:
            return np.zeros(0, dtype=np.uint32), np.zeros((0, 4), dtype=F)
        md_pos, _ = self.mouse_down_on
        ix = self.modify_vector_indices
        disp = self.edit_mode.get_vector(grid.grid[ix], self._positions[ix], md_pos, self.pos, self.rot)
        new = disp + self.mouse_down_vectors if self.edit_mode.relative else disp
        grid.grid[ix] = new
        cells = np.ones((ix.shape[0], 4), dtype=F)
        cells[:, :3] = new
        return ix.astype(np.uint32), cells

    def mouse_up(self):
        """src/cursor.rs:118-121."""
        self.mouse_down_on = None
        self.modify_vector_indices = np.zeros(0, dtype=np.int64)


class DeviceBrush:
    """The same three events as ``Brush``, executed on the device field of a
    ``Compute`` or ``ShardedCompute`` (src/cursor.rs:79-122): nothing but the
    64-byte cursor record leaves the host."""

    def __init__(self, compute, edit_mode=None):
        self.compute = compute
        self.edit_mode = edit_mode or EditMode()
        self.rot = np.eye(3, dtype=F)

    def mouse_down(self, pos, rot=None):
        if rot is not None:
            self.rot = np.asarray(rot, dtype=F)
        self.compute.brush_mouse_down(self.edit_mode.record(pos, self.rot))

    def mouse_moved(self, pos, rot=None):
        if rot is not None:
            self.rot = np.asarray(rot, dtype=F)
        self.compute.brush_mouse_moved(self.edit_mode.record(pos, self.rot))

    def mouse_up(self):
        self.compute.brush_mouse_up()
