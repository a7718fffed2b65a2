#!/usr/bin/env python3
"""Timing sweep of the streaming kernel over tuning knobs and scenes (GPU only).
Prints one line per configuration: GB/s of algorithmic traffic (49 B/particle-step)."""
import argparse
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa
from bench import swirl_field, make_params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=26)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--fields", type=str, default="1024,256")
    ap.add_argument("--extents", type=str, default="20,2")
    ap.add_argument("--grids", type=str, default="0")
    ap.add_argument("--variants", type=str, default="0")
    ap.add_argument("--reorder", type=str, default="-1", help="comma list of reorder intervals (-1 auto, 0 off)")
    args = ap.parse_args()
    n = 1 << args.log2n
    c = pa.Compute.new(pa.empty_particles(0), np.zeros((1, 4), dtype=np.float32))
    for fd in [int(v) for v in args.fields.split(",")]:
        dims = (fd, fd, 1)
        c.update_force_grid(swirl_field(*dims, 10.0, 1))
        c.update_sim_params(make_params(pa, dims))
        for ext in [float(v) for v in args.extents.split(",")]:
            for grid, var, ro in itertools.product([int(v) for v in args.grids.split(",")],
                                                   [int(v) for v in args.variants.split(",")],
                                                   [int(v) for v in args.reorder.split(",")]):
                c.set_tuning(grid, var)
                c.set_reorder_interval(ro)
                c.generate_particles(n, 12345, 0, ext, 10.0)
                r0 = c.reorder_count
                for _ in range(3):
                    c.compute()
                c.synchronize()
                c.timer_start()
                for _ in range(args.steps):
                    c.compute()
                ms = c.timer_stop() / args.steps
                print(f"field {fd:5d}^2 extent {ext:5.1f} grid {c.launch_geometry()['grid_blocks']:6d} variant {var:3d} "
                      f"reorder {ro:3d} ({c.reorder_count - r0:3d} sorts): "
                      f"{ms:8.4f} ms/step  {n * 49 / ms / 1e6:8.1f} GB/s  {n / ms / 1e6:8.2f} Gparticle-steps/s",
                      flush=True)


if __name__ == "__main__":
    main()
