#!/usr/bin/env python3
"""Per-launch step times over refresh cycles of the tile order on the headline workload (GPU only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa
from bench import swirl_field, make_params


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    dims = (1024, 1024, 1)
    with pa.Compute.new(pa.empty_particles(0), swirl_field(*dims, 10.0, 1)) as c:
        c.update_sim_params(make_params(pa, dims))
        c.generate_particles(1 << log2n, 12345, 0, 20.0, 10.0)
        for _ in range(25):
            c.compute()
        c.synchronize()
        c.step_timing_begin(36)
        for _ in range(36):
            c.compute()
        ms, kinds = c.step_timing_read()
        names = {1: "plain", 3: "count", 4: "scatter"}
        for i, (t, k) in enumerate(zip(ms, kinds)):
            print(f"step {i:2d} {names.get(int(k), k):8s} {t:7.4f} ms")


if __name__ == "__main__":
    main()
