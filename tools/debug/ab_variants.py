"""A/B of tuning variants inside one process (one box): for every variant given on the command line the
headline workload is generated afresh and stepped with one event pair per launch.  Prints per launch kind
the average duration, the average over steps 6..25 (the window `bench.py --warmup 5 --steps 20` times) and
over the whole cycles after the first.   python tools/debug/ab_variants.py [--log2n 28] [--steps 70] 0 0x100 ..."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("PCL_DEBUG_LIB"):      # a development build of the library instead of the shipped one
    from particles_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["PCL_DEBUG_LIB"])
import particles_amd as pa
from bench import swirl_field, make_params, RADIUS, MASSES

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=28)
ap.add_argument("--steps", type=int, default=70)
ap.add_argument("--rounds", type=int, default=1)
ap.add_argument("--interval", type=int, default=-1)
ap.add_argument("--sequence", action="store_true")
ap.add_argument("variants", nargs="+", type=lambda s: int(s, 0))
args = ap.parse_args()
dims = (1024, 1024, 1)
g = swirl_field(*dims, RADIUS, 12345)
sp = make_params(pa, dims, MASSES)
with pa.Compute.new(pa.empty_particles(0), g) as c:
    c.update_sim_params(sp)
    for rnd in range(args.rounds):
        for v in args.variants:
            c.set_tuning(0, v)
            c.set_reorder_interval(args.interval)
            c.generate_particles(1 << args.log2n, seed=12345, pos_extent=20.0, vel_extent=10.0)
            c.step_timing_begin(args.steps)
            for _ in range(args.steps):
                c.compute()
            c.synchronize()
            ms, kind = c.step_timing_read()
            ms, kind = np.asarray(ms), np.asarray(kind)
            k = lambda q, sel=slice(None): (ms[sel][kind[sel] == q].mean() if (kind[sel] == q).any() else float("nan"))
            later = slice(17, None)
            print(f"variant {v:#x}: first step {ms[0]:.2f} ms | steps 6..25 avg {ms[5:25].mean():.4f} "
                  f"(plain {k(1, slice(5, 25)):.4f}) | after first cycle: plain {k(1, later):.4f} count {k(3, later):.4f} "
                  f"scatter {k(4, later):.4f} whole {ms[later].mean():.4f}", flush=True)
            if args.sequence:
                print(' '.join(f"{q}:{m:.3f}" for m, q in zip(ms, kind)), flush=True)
