#!/usr/bin/env python3
"""The reference's default scene through the Python mirror of ``Compute``
(src/compute.rs:11-228): 1000 lattice particles (src/main.rs:117-154) in the
centred 10 x 10 x 10 field (src/grid.rs:68-95), stepped at a fixed dt.

    python examples/default_scene.py [steps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    params = pa.SimParams()                                   # SimParams::new()
    particles = pa.lattice_particles((5.0, 2.0, 2.0), 10, 10, 10)
    field = params.new_force_grid_centered()                  # src/sim_params.rs:33-45
    with pa.Compute.new(particles, field.get_force_vectors()) as compute:
        params.delta_t = np.float32(1.0 / 60.0)
        compute.update_sim_params(params)
        for _ in range(steps):
            compute.compute()
        out = compute.read_particles()
        d = compute.diagnostics()
    print(f"steps {steps} particles {d['count']} kinetic_energy {d['kinetic_energy']:.9g}")
    for i in (0, 999):
        print(f"particle {i}: pos", *(f"{v:.9g}" for v in out["pos"][i, :3]), "vel", *(f"{v:.9g}" for v in out["vel"][i, :3]))


if __name__ == "__main__":
    main()
