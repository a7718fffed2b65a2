#!/usr/bin/env python3
"""Does the plain step have fast and slow placements of the library's own allocation?  One process; before every
handle a dummy allocation of D MiB shifts where the particle arrays land.
    python tools/debug/placement_sweep.py [log2n = 28] [first D = 0] [last D = 80] [step = 2]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import particles_amd as pa
from bench import swirl_field, make_params, RADIUS, MASSES

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
d0, d1, dd = (int(sys.argv[k]) if len(sys.argv) > k else v for k, v in ((2, 0), (3, 80), (4, 2)))
dims = (1024, 1024, 1)
g = swirl_field(*dims, RADIUS, 12345)
sp = make_params(pa, dims, MASSES)
for D in range(d0, d1 + 1, dd):
    dummy = torch.empty(max(D, 0) << 20, dtype=torch.uint8, device="cuda") if D else None
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.generate_particles(1 << lg, seed=12345, pos_extent=20.0, vel_extent=10.0)
        steps = 22
        c.step_timing_begin(steps)
        for _ in range(steps):
            c.compute()
        c.synchronize()
        ms, kind = c.step_timing_read()
        ms, kind = np.asarray(ms), np.asarray(kind)
        plain = ms[2:][kind[2:] == 1]
        print(f"dummy {D:5d} MiB  plain step avg {plain.mean():.4f} ms  min {plain.min():.4f}  max {plain.max():.4f}", flush=True)
    del dummy
    torch.cuda.empty_cache()
