#!/usr/bin/env python3
"""The cell-binned pair kernel alone, for rocprofv3: uniform particles in the 20-unit box, cut-off 1,
force curves of the usual random set; one warm-up and two timed steps per size.
    python3 tools/binnedprof.py [log2n ...]          (default: 20 24)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [20, 24]
    sp = pa.SimParams()
    sp.delta_t = np.float32(1 / 60)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.polys[:] = np.random.default_rng(1).uniform(-1, 1, size=(25, 8)).astype(np.float32)
    sp.cut_off_distance = float(os.environ.get("PAIRBENCH_CUTOFF", "1.0"))
    g = sp.new_force_grid_centered().get_force_vectors()
    for log2n in sizes:
        n = 1 << log2n
        with pa.Compute.new(pa.empty_particles(0), g) as c:
            c.update_sim_params(sp)
            c.set_pair_mode(pa.PAIRS_BINNED)
            c.set_tuning(0, int(os.environ.get("PAIRBENCH_VARIANT", "0"), 0))
            c.generate_particles(n, 7, 0, 20.0, 10.0)
            c.compute()
            c.synchronize()
            c.timer_start()
            for _ in range(2):
                c.compute()
            ms = c.timer_stop() / 2
            # candidates swept per particle: density x volume of the cells within reach (upper bound: no column skipped)
            print(f"n 2^{log2n}: {ms:.3f} ms/step (sort + sweep)", flush=True)


if __name__ == "__main__":
    main()
