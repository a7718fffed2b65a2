#!/usr/bin/env python3
"""A few steps of the cell-binned pair term, for rocprofv3 --kernel-trace --stats: which share of a step the sort
before the sweep takes (restore order, keys, radix sort, cell ends, gather) at a given size.
    python tools/debug/binned_breakdown.py [log2 n = 20] [cut-off = 1.0] [steps = 6]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import particles_amd as pa

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sp = pa.SimParams()
sp.delta_t = np.float32(1 / 60)
sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
sp.polys[:] = np.random.default_rng(1).uniform(-1, 1, size=(25, 8)).astype(np.float32)
sp.cut_off_distance = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
g = sp.new_force_grid_centered().get_force_vectors()
with pa.Compute.new(pa.empty_particles(0), g) as c:
    c.update_sim_params(sp)
    c.set_pair_mode(2)
    c.generate_particles(1 << lg, 7, 0, 20.0, 10.0)
    for _ in range(steps):
        c.compute()
    c.synchronize()
    print("kernel", c.last_step_kernel, "particles", c.num_particles)
