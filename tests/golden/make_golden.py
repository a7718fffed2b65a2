#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ with the CPU oracle
(oracle/particles_oracle.c).  The reference ships no golden vectors and cannot
be run here, so these files pin the ORACLE (and through it the HIP path)
against regressions; they are not outputs of the reference itself.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import particles_amd as pa  # noqa: E402
from oracle import oracle as O  # noqa: E402

F = np.float32
DT = F(1.0 / 60.0)


def save(name, sp, p0, g, steps, mode, extra=None):
    out = O.run(sp, p0, g, steps, mode=mode)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        params=np.frombuffer(sp.tobytes(), dtype=np.uint8), particles0=p0.view(np.uint8).reshape(-1, 48),
        grid=g, steps=np.int32(steps), mode=np.int32(mode), expected=out.view(np.uint8).reshape(-1, 48),
        cells0=O.cell_indices(sp, p0, g.shape[0]), **(extra or {}))
    print(name, out.shape[0], "particles", steps, "steps")


def main():
    # 1. the reference's start-up scene, dt fixed at 1/60, 16 steps
    sp = pa.SimParams()
    sp.delta_t = DT
    p = pa.lattice_particles((5.0, 2.0, 2.0), 10, 10, 10)
    g = sp.new_force_grid_centered().get_force_vectors()
    save("default_scene_16", sp, p, g, 16, O.MODE_ALWAYS_PAIRS)

    # 2. seeded random scene, non-zero force curves, exponent 0 (bit-exact on the GPU)
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.cut_off_distance = 1.5
    sp.polys[:] = np.random.default_rng(2024).uniform(-2, 2, size=(25, 8)).astype(F)
    p = O.synth_particles(600, seed=21, pos_extent=4.0, vel_extent=10.0).view(pa.PARTICLE_DTYPE)
    g = sp.new_force_grid_centered().get_force_vectors()
    save("pairs_exp0_3", sp, p, g, 3, O.MODE_ALWAYS_PAIRS)

    # 3. the same with a distance exponent (powf: tolerance on the GPU)
    sp.distance_exponent = -1.5
    save("pairs_expm1p5_1", sp, p, g, 1, O.MODE_ALWAYS_PAIRS)

    # 4. boundary cases on a flat 64x64x1 field: particles on and next to the
    #    wrap planes and cell faces, speeds around the clamp and the 0.001 gate
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (64, 64, 1)
    sp.max_velocity = 7.5
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    rng = np.random.default_rng(7)
    g = np.ones((64 * 64, 4), dtype=F)
    g[:, :3] = rng.uniform(-1, 1, size=(64 * 64, 3)).astype(F)
    p = pa.empty_particles(4096)
    r = F(10.0)
    faces = (np.arange(65, dtype=F) / F(64.0) * F(20.0) - r).astype(F)
    specials = np.concatenate([faces, np.nextafter(faces, F(np.inf)), np.nextafter(faces, F(-np.inf)),
                               np.array([r, -r, 0.0, -0.0], dtype=F)])
    p["pos"][:, 0] = rng.choice(specials, size=4096)
    p["pos"][:, 1] = rng.choice(specials, size=4096)
    p["pos"][:, 2] = rng.choice(specials, size=4096)
    speeds = rng.choice(np.array([0.0, 0.0005, 0.001, 0.0011, 7.4, 7.5, 7.6, 250.0], dtype=F), size=4096)
    d = rng.normal(size=(4096, 3)).astype(F)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(F)
    p["vel"][:, :3] = d * speeds[:, None]
    p["pos"][:, 3] = 1.0
    p["ty"] = rng.integers(0, 5, size=4096).astype(np.uint32)
    save("edges_64x64_4", sp, p, g, 4, O.MODE_SKIP_INERT)


if __name__ == "__main__":
    main()
