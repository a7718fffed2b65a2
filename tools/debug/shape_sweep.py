"""Launch shape of the streaming kernels by problem size: for every size and every tuning variant given, the
average duration of the plain, counting and scatter launches over whole refresh cycles (first cycle skipped).
   python tools/debug/shape_sweep.py [--diag] [--field 1024] --log2n 24 26 28 -- 0x0 0x13000000 ..."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("PCL_DEBUG_LIB"):
    from particles_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["PCL_DEBUG_LIB"])
import particles_amd as pa
from bench import swirl_field, make_params, RADIUS, MASSES

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, nargs="+", default=[28])
ap.add_argument("--field", type=int, default=1024)
ap.add_argument("--diag", action="store_true")
ap.add_argument("--steps", type=int, default=66)
ap.add_argument("--interval", type=int, default=-1)
ap.add_argument("variants", nargs="+", type=lambda s: int(s, 0))
args = ap.parse_args()
dims = (args.field, args.field, 1)
g = swirl_field(*dims, RADIUS, 12345)
sp = make_params(pa, dims, MASSES)
with pa.Compute.new(pa.empty_particles(0), g) as c:
    c.update_sim_params(sp)
    c.enable_step_diagnostics(args.diag)
    for log2n in args.log2n:
        for v in args.variants:
            c.set_tuning(0, v)
            c.set_reorder_interval(args.interval)
            c.generate_particles(1 << log2n, seed=12345, pos_extent=20.0, vel_extent=10.0)
            c.step_timing_begin(args.steps)
            for _ in range(args.steps):
                c.compute()
            c.synchronize()
            ms, kind = c.step_timing_read()
            ms, kind = np.asarray(ms)[17:], np.asarray(kind)[17:]
            k = lambda q: (ms[kind == q].mean() if (kind == q).any() else float("nan"))
            print(f"2^{log2n} field {args.field}^2 diag {int(args.diag)} variant {v:#010x}: plain {k(1):.4f} count {k(3):.4f} "
                  f"scatter {k(4):.4f} whole {ms.mean():.4f} ms  ({(1 << log2n) * 49 / ms.mean() / 8e9 * 100:.1f} % of 8 TB/s)", flush=True)
