#!/usr/bin/env python3
"""Maximum-size check (run by hand on the GPU box: ``python tests/big_sizes_check.py 30 31``):
2^k particles generated on the device, 1024^2 field, automatic tile order, 26 steps (one full
count + scatter refresh cycle at the interval of 24); slices of the result against the oracle stepping the same global
indices.  Arrays of 2^30 floats and more put byte offsets beyond 32 bits."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import particles_amd as pa
from conftest import assert_particles_bit_equal, scene_swirl_field
from oracle import oracle as O


def check(sizes, steps=26):
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.force_grid_dimensions = (1024, 1024, 1)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(1024, 1024, 1, 10.0, seed=5)
    for log2n in sizes:
        n = (1 << log2n) + (5 if log2n >= 32 else 0)   # from 2^32 on: ragged, element indices beyond 32 bits, no tile order
        t0 = time.time()
        with pa.Compute.new(pa.empty_particles(0), g) as c:
            c.update_sim_params(sp)
            c.generate_particles(n, seed=99, pos_extent=20.0, vel_extent=10.0)
            for _ in range(steps):
                c.compute()
            c.synchronize()
            print(f"2^{log2n}: {steps} steps done in {time.time() - t0:.1f} s, reorders {c.reorder_count}", flush=True)
            assert c.reorder_count >= 2 or log2n >= 32   # caller indices are 32-bit: no tile order from 2^32 - 1 particles on
            for first in (0, n // 2 - 1000, n // 2 + 123456789 % (n // 4), n - 3000):
                p = O.synth_particles(3000, seed=99, first_global_index=first, pos_extent=20.0, vel_extent=10.0)
                want = O.run(sp, p, g, steps, mode=O.MODE_SKIP_INERT)
                assert_particles_bit_equal(c.read_particles(first, 3000), want, f"2^{log2n} slice at {first}")
            d = c.diagnostics()
            assert d["count"] == n, d
            print(f"2^{log2n}: slices bit-exact, count {d['count']}, kinetic energy {d['kinetic_energy']:.6e}", flush=True)


if __name__ == "__main__":
    # usage: big_sizes_check.py [--steps K] log2n...
    argv = sys.argv[1:]
    k = 26
    if argv and argv[0] == "--steps":
        k = int(argv[1])
        argv = argv[2:]
    check([int(a) for a in argv], steps=k)
