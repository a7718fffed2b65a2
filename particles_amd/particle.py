"""Particle record and mass wrapper, bit-compatible with the reference.

Reference: ``struct Particle`` (src/main.rs:76-83, src/compute.wgsl:1-5) is 48
bytes: pos vec4 @0, vel vec4 @16, ty u32 @32, 12 bytes of padding.
``struct MassWrap`` (src/main.rs:196-201) is a 16-byte padded f32.
"""
import numpy as np

PARTICLE_DTYPE = np.dtype(
    [("pos", np.float32, (4,)), ("vel", np.float32, (4,)), ("ty", np.uint32), ("_padd", np.uint32, (3,))],
    align=False,
)
assert PARTICLE_DTYPE.itemsize == 48
assert PARTICLE_DTYPE.fields["vel"][1] == 16 and PARTICLE_DTYPE.fields["ty"][1] == 32

MASS_WRAP_DTYPE = np.dtype([("mass", np.float32), ("_pad", np.float32, (3,))])
assert MASS_WRAP_DTYPE.itemsize == 16

NUM_TYPES = 5


def empty_particles(n):
    """Zero-initialised array of ``n`` particle records."""
    return np.zeros(int(n), dtype=PARTICLE_DTYPE)


def lattice_particles(max_xyz, num_x, num_y, num_z):
    """The start-up lattice of ``ParticleSystem::new`` (src/main.rs:117-154).

    pos = (ix/nx*max.x, iy/ny*max.y, iz/nz*max.z, 1), vel = 0 (all four lanes),
    ty = flat index % 5, x slowest and z fastest.
    """
    mx, my, mz = (np.float32(v) for v in max_xyz)
    ix, iy, iz = np.meshgrid(
        np.arange(num_x, dtype=np.float32),
        np.arange(num_y, dtype=np.float32),
        np.arange(num_z, dtype=np.float32),
        indexing="ij",
    )
    p = empty_particles(num_x * num_y * num_z)
    p["pos"][:, 0] = ((ix / np.float32(num_x)) * mx).ravel()
    p["pos"][:, 1] = ((iy / np.float32(num_y)) * my).ravel()
    p["pos"][:, 2] = ((iz / np.float32(num_z)) * mz).ravel()
    p["pos"][:, 3] = 1.0
    p["ty"] = (np.arange(p.shape[0]) % NUM_TYPES).astype(np.uint32)
    return p


def random_particles(n, rng, first_index=0, pos_max=2.0, vel_max=10.0):
    """Distribution of ``set_num_particles`` (src/main.rs:41-48,156-169).

    pos = pos_max*(u-0.5), vel = vel_max*(u-0.5) per axis with u uniform in
    [0,1); w lanes 1.0; ty = index % 5.  The reference draws from an unseeded
    thread RNG, so only the distribution is reproduced; ``rng`` is a
    ``numpy.random.Generator``.
    """
    p = empty_particles(n)
    u = rng.random((n, 6), dtype=np.float32)
    p["pos"][:, :3] = np.float32(pos_max) * (u[:, :3] - np.float32(0.5))
    p["vel"][:, :3] = np.float32(vel_max) * (u[:, 3:] - np.float32(0.5))
    p["pos"][:, 3] = 1.0
    p["vel"][:, 3] = 1.0
    p["ty"] = ((np.arange(n) + first_index) % NUM_TYPES).astype(np.uint32)
    return p
