#!/usr/bin/env python3
"""The cell sort of the binned pair term on its own (pcl_debug_pair_cells runs exactly that sort);
meant to run under rocprofv3 --kernel-trace --stats, which gives the time of every kernel of it.
    python tools/debug/sort_timing.py [log2 n = 24] [cut-off = 1.0]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import particles_amd as pa
from particles_amd import _lib
import ctypes as C


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    sp = pa.SimParams()
    sp.delta_t = np.float32(1 / 60)
    sp.polys[:] = np.random.default_rng(1).uniform(-1, 1, size=(25, 8)).astype(np.float32)
    sp.cut_off_distance = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    g = sp.new_force_grid_centered().get_force_vectors()
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        c.generate_particles(1 << lg, 7, 0, 20.0, 10.0)
        per_axis, reach = C.c_uint32(), C.c_uint32()
        for _ in range(5):   # the sort alone: no outputs asked for beyond the two scalars
            _lib.check(c._lib.pcl_debug_pair_cells(c._h, None, None, 0, C.byref(per_axis), C.byref(reach)))
        c.synchronize()
        print("cells per axis", per_axis.value, "reach", reach.value)


main()
