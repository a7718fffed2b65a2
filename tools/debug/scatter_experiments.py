"""Timing-only take-apart of the scatter step (development build abtmp/libparticles_phase.so, results of
modes > 0 are wrong): mode 0 as shipped, 1 dword stores to tile-local slots, 2 16-byte stores to tile-local
slots, 3 no state stores.  The scatter launch of step 33 is timed in each mode on the same input."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from particles_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "abtmp", "libparticles_phase.so")
import particles_amd as pa
from bench import swirl_field, make_params, RADIUS, MASSES

dims = (1024, 1024, 1)
g = swirl_field(*dims, RADIUS, 12345)
sp = make_params(pa, dims, MASSES)
with pa.Compute.new(pa.empty_particles(0), g) as c:
    c.update_sim_params(sp)
    lib = _lib.load()
    lib.pcl_debug_experiment.argtypes = [C.c_void_p, C.c_int]
    for variant in [int(v, 0) for v in sys.argv[1:]] or [0]:
        for mode in (0, 1, 2, 3, 0):
            c.set_tuning(0, variant)
            _lib.check(lib.pcl_debug_experiment(c._h, 0))
            c.generate_particles(1 << 28, seed=12345, pos_extent=20.0, vel_extent=10.0)
            for _ in range(32):
                c.compute()
            c.synchronize()
            _lib.check(lib.pcl_debug_experiment(c._h, mode))
            c.step_timing_begin(1)
            c.compute()
            c.synchronize()
            ms, kind = c.step_timing_read()
            print(f"variant {variant:#x} mode {mode}: kind {kind[0]} {ms[0]:.3f} ms", flush=True)
