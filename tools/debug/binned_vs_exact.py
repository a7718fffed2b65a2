#!/usr/bin/env python3
"""Step-by-step difference between the binned and the exact pair kernel (both on the GPU) on one scene."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import particles_amd as pa
from oracle import oracle as O

def scene(exponent, cutoff, scale):
    sp = pa.SimParams()
    sp.delta_t = np.float32(1 / 60)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.cut_off_distance = cutoff
    sp.distance_exponent = exponent
    rng = np.random.default_rng(99)
    sp.polys[:] = rng.uniform(-2.0, 2.0, size=(25, 8)).astype(np.float32) * np.float32(scale)
    return sp

for cutoff, n, scale, variant in [(2.5, 4000, 0.0015, 0), (2.5, 4000, 0.0015, 0x10000), (1.5, 6000, 0.05, 0)]:
    sp = scene(0.0, cutoff, scale)
    g = sp.new_force_grid_centered().get_force_vectors()
    p = O.synth_particles(n, seed=n + 1, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as a, pa.Compute.new(p, g) as b:
        for c, mode in ((a, 1), (b, 2)):
            c.update_sim_params(sp); c.set_pair_mode(mode)
        b.set_tuning(0, variant)
        for k in range(1, 17):
            a.compute(); b.compute()
            x, y = a.read_particles(), b.read_particles()
            dv = np.abs(x["vel"][:, :3].astype(np.float64) - y["vel"][:, :3]).max(axis=1)
            dp = np.abs(x["pos"][:, :3].astype(np.float64) - y["pos"][:, :3]).max(axis=1)
            print(f"cut {cutoff} variant {variant:#x} step {k}: vel differs in {(dv > 0).sum()} particles, max {dv.max():.3g}, "
                  f"> 1e-4: {(dv > 1e-4).sum()}; pos max {dp.max():.3g}", flush=True)
