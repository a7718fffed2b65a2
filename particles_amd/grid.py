"""Host copy of the vector-force field (reference: src/grid.rs).

Cells are stored x slowest, z fastest (src/grid.rs:72-84), the same flattening
the shader uses (src/compute.wgsl:88).
"""
import numpy as np


class Bounds:
    """Axis-aligned box: corner ``pos`` and extent ``dir`` (src/grid.rs:6-40)."""

    def __init__(self, pos, dir):
        self.pos = np.asarray(pos, dtype=np.float32)
        self.dir = np.asarray(dir, dtype=np.float32)

    def center(self):
        return self.pos + np.float32(0.5) * self.dir


class Grid:
    """``Grid<Vector3<f32>>`` of the reference (src/grid.rs:42-171)."""

    def __init__(self, vectors, size, bounds):
        self.size = tuple(int(s) for s in size)
        self.grid = np.ascontiguousarray(vectors, dtype=np.float32).reshape(-1, 3)
        if self.grid.shape[0] != self.size[0] * self.size[1] * self.size[2]:
            raise ValueError("grid length does not match its dimensions")
        self.bounds = bounds

    @classmethod
    def new_uniform(cls, n_x, n_y, n_z, bounds, value=(0.0, 0.0, 0.0)):  # src/grid.rs:49-64
        v = np.empty((n_x * n_y * n_z, 3), dtype=np.float32)
        v[:] = np.asarray(value, dtype=np.float32)
        return cls(v, (n_x, n_y, n_z), bounds)

    @classmethod
    def new_centered(cls, n_x, n_y, n_z, bounds):
        """Every vector points from its cell index to the grid centre:
        v = (nx,ny,nz)*0.5 - (ix,iy,iz)  (src/grid.rs:68-95)."""
        center = np.array([n_x, n_y, n_z], dtype=np.float32) * np.float32(0.5)
        ix, iy, iz = np.meshgrid(
            np.arange(n_x, dtype=np.float32), np.arange(n_y, dtype=np.float32),
            np.arange(n_z, dtype=np.float32), indexing="ij")
        idx = np.stack([ix.ravel(), iy.ravel(), iz.ravel()], axis=1)
        return cls(center[None, :] - idx, (n_x, n_y, n_z), bounds)

    def num_instances(self):  # src/grid.rs:105-107
        return self.size[0] * self.size[1] * self.size[2]

    def get_force_vectors(self):
        """[x, y, z, 1.0] per cell, the upload format (src/grid.rs:97-103)."""
        out = np.empty((self.grid.shape[0], 4), dtype=np.float32)
        out[:, :3] = self.grid
        out[:, 3] = 1.0
        return out

    def get_positions(self):
        """Cell-centre positions [x, y, z, 1.0] (src/grid.rs:119-138)."""
        f32 = np.float32
        n_x, n_y, n_z = self.size
        ax = self.bounds.pos[0] + self.bounds.dir[0] * ((np.arange(n_x, dtype=f32) + f32(0.5)) / f32(n_x))
        ay = self.bounds.pos[1] + self.bounds.dir[1] * ((np.arange(n_y, dtype=f32) + f32(0.5)) / f32(n_y))
        az = self.bounds.pos[2] + self.bounds.dir[2] * ((np.arange(n_z, dtype=f32) + f32(0.5)) / f32(n_z))
        px, py, pz = np.meshgrid(ax.astype(f32), ay.astype(f32), az.astype(f32), indexing="ij")
        out = np.empty((self.num_instances(), 4), dtype=f32)
        out[:, 0], out[:, 1], out[:, 2], out[:, 3] = px.ravel(), py.ravel(), pz.ravel(), 1.0
        return out

    def get_indices(self, center, radius):
        """Cells whose centre lies strictly within ``radius`` of ``center``
        (src/grid.rs:109-117), ascending."""
        pos = self.get_positions()[:, :3]
        d = pos - np.asarray(center, dtype=np.float32)[None, :]
        dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        return np.nonzero(dist < np.float32(radius))[0]
