#!/usr/bin/env python3
"""Timing of the general (all-pairs) kernel at the reference's operating sizes (GPU only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa


def main():
    sp = pa.SimParams()
    sp.delta_t = np.float32(1 / 60)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.polys[:] = np.random.default_rng(1).uniform(-1, 1, size=(25, 8)).astype(np.float32)
    g = sp.new_force_grid_centered().get_force_vectors()
    variants = {1: (0,), 2: (0,)}
    cutoff = float(os.environ.get("PAIRBENCH_CUTOFF", "1.0"))
    sp.cut_off_distance = cutoff
    if "PAIRBENCH_VARIANT" in os.environ:
        variants = {2: tuple(int(v, 0) for v in os.environ["PAIRBENCH_VARIANT"].split(","))}
    for mode, variant in ((m, v) for m in sorted(variants) for v in variants[m]):
      for exponent in (0.0,):
        sp.distance_exponent = exponent
        for extent in (2.0, 20.0):
            for n in (1000, 16384, 50000, 131072, 1 << 20, 1 << 24):
                if mode == 1 and n > 131072:
                    continue
                if extent == 2.0 and n > (1 << 20):
                    continue
                with pa.Compute.new(pa.empty_particles(0), g) as c:
                    c.update_sim_params(sp)
                    c.set_pair_mode(mode)
                    c.set_tuning(0, variant)
                    c.generate_particles(n, 7, 0, extent, 10.0)
                    for _ in range(2):
                        c.compute()
                    c.synchronize()
                    steps = 10 if n <= 50000 else 2
                    c.timer_start()
                    for _ in range(steps):
                        c.compute()
                    ms = c.timer_stop() / steps
                    assert c.last_step_kernel == pa.KERNEL_PAIRS
                    print(f"cutoff {cutoff} mode {mode} variant {variant:#x} exponent {exponent:4.1f} extent {extent:4.1f} n {n:7d}: {ms:9.4f} ms/step  "
                          f"{n * n / ms / 1e6:9.2f} G pair visits/s", flush=True)


if __name__ == "__main__":
    main()
