"""Independent numpy-float32 restatement of src/compute.wgsl (vectorised over
particles), used to cross-check the C oracle.  Every operation is a numpy
float32 ufunc (IEEE add/sub/mul/div/sqrt), written from the shader text and
not from the C file: a disagreement means one of the two transcriptions is
wrong."""
import numpy as np

F = np.float32


def eval_poly7(x, c):
    # wgsl:51-56; c[..., k] multiplies x^k; the shader's order, not Horner
    x2 = x * x
    x4 = x2 * x2
    x3 = x2 * x
    return (((((((c[..., 7] * x4) * x3 + (c[..., 6] * x3) * x3) + (c[..., 5] * x2) * x3) + c[..., 4] * x4)
              + c[..., 3] * x3) + c[..., 2] * x2) + c[..., 1] * x) + c[..., 0]


def wrap_symmetrically(v, mx):
    # wgsl:58-66
    return np.where(v > mx, v - F(2.0) * mx, np.where(v < -mx, v + F(2.0) * mx, v)).astype(F)


def map_range(v, a0, a1, b0, b1):
    # wgsl:68-79
    if a0 == a1:
        return np.full_like(v, b0)
    valc = np.minimum(np.maximum(v, a0), a1)
    oz = (valc - a0) / (a1 - a0)
    return (oz * (b1 - b0) + b0).astype(F)


def force_grid_index(side, dims, pos):
    # wgsl:81-89 with u32 wrap-around arithmetic
    slh = F(side) * F(0.5)
    ix = [map_range(pos[:, k], -slh, slh, F(0.0), F(dims[k])).astype(np.uint32) for k in range(3)]
    dy, dz = np.uint32(dims[1]), np.uint32(dims[2])
    with np.errstate(over="ignore"):
        return (ix[0] * dy * dz + ix[1] * dz + ix[2]).astype(np.uint32)


def length(v):
    return np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]).astype(F)


def step(sp, particles, grid, decay, pairs=True):
    """One update (wgsl:97-181).  `sp` is a particles_amd.SimParams; `decay`
    is exp(-delta_t) as an np.float32 (the transcendental is supplied by the
    caller)."""
    raw = sp.raw
    pos = particles["pos"][:, :3].astype(F)
    vel = particles["vel"][:, :3].astype(F)
    ty = particles["ty"].astype(np.int64)
    masses = raw["particle_type_masses"]["mass"].astype(F)
    polys = raw["attraction_force"]["coeffs"].astype(F)
    dt = F(raw["delta_t"])
    n = pos.shape[0]
    acc = np.zeros((n, 3), dtype=F)
    if pairs and n > 1:
        cut, expo = F(raw["cut_off_distance"]), F(raw["distance_exponent"])
        for j in range(n):  # the shader's accumulation order: j = 0..n-1
            d = (pos[j][None, :] - pos).astype(F)
            ln = length(d)
            ok = ~((np.arange(n) == j) | (ln < F(0.001)) | (ln > cut))
            if not ok.any():
                continue
            with np.errstate(all="ignore"):
                df = np.power(ln, expo, dtype=F)
                dn = (d / ln[:, None]).astype(F)
                pv = eval_poly7(ln, polys[ty[j] + 5 * ty]).astype(F)
                term = ((dn * pv[:, None]) * masses[ty[j]]) * df[:, None]
            acc = np.where(ok[:, None], acc + term.astype(F), acc).astype(F)
    dims = raw["force_grid_dimensions"]
    fgi = force_grid_index(F(raw["bounding_volume_radius"]) * F(2.0), dims, pos)
    fgi = np.minimum(fgi, np.uint32(grid.shape[0] - 1))
    acc = (acc + (F(10.0) * grid[fgi, :3].astype(F)) / masses[ty][:, None]).astype(F)
    vel = (vel * decay).astype(F)
    vel = (vel + acc * dt).astype(F)
    speed = length(vel)
    with np.errstate(all="ignore"):
        capped = np.minimum(np.maximum(speed, F(0.0)), F(raw["max_velocity"]))
        renorm = ((vel / speed[:, None]) * capped[:, None]).astype(F)
    vel = np.where((speed > F(0.001))[:, None], renorm, vel).astype(F)
    pos = (pos + vel * dt).astype(F)
    r = F(raw["bounding_volume_radius"])
    pos = np.stack([wrap_symmetrically(pos[:, k], r) for k in range(3)], axis=1)
    out = np.zeros_like(particles)
    out["pos"][:, :3], out["pos"][:, 3] = pos, 1.0
    out["vel"][:, :3], out["vel"][:, 3] = vel, 1.0
    out["ty"] = particles["ty"]
    return out, fgi
