"""Kind and duration of every launch over the first 70 steps of the headline workload (kind 1 plain step,
3 counting step, 4 scatter step): how the plain step ages between refreshes, and how the first cycle after the
sort of unordered input differs from the later ones.  python tools/debug/step_sequence.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import particles_amd as pa
from bench import swirl_field, make_params, RADIUS, MASSES
dims = (1024, 1024, 1)
g = swirl_field(*dims, RADIUS, 12345)
sp = make_params(pa, dims, MASSES)
with pa.Compute.new(pa.empty_particles(0), g) as c:
    c.update_sim_params(sp)
    c.generate_particles(1 << 28, seed=12345, pos_extent=20.0, vel_extent=10.0)
    n = 70
    c.step_timing_begin(n)
    for _ in range(n):
        c.compute()
    c.synchronize()
    ms, kind = c.step_timing_read()
    print(' '.join(f"{k}:{m:.3f}" for m, k in zip(ms, kind)))
