"""Where a block of the scatter step spends its time (development build with -DPCL_PHASE_CLOCKS at
abtmp/libparticles_phase.so): thread 0 of every 1024th block stamps the 100 MHz wall clock at its phase
boundaries; prints the median duration of every phase in microseconds for a steady-state refresh.
   python tools/debug/phase_clocks.py [variant ...]"""
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

NAMES = {1: "loads issued, counters cleared, barrier", 2: "keys + ranks (waits for the loads)", 3: "rank barrier + reservations issued",
         4: "field fetch + update", 5: "plan (sweep, reservations consumed)", 6: "dest + n_local barrier",
         7: "x", 8: "y", 9: "z", 10: "vx", 11: "vy", 12: "vz", 13: "orig", 16: "type + end"}
dims = (1024, 1024, 1)
g = swirl_field(*dims, RADIUS, 12345)
sp = make_params(pa, dims, MASSES)
variants = [int(v, 0) for v in sys.argv[1:] if not v.startswith("mode=")] or [0]
modes = [int(v[5:]) for v in sys.argv[1:] if v.startswith("mode=")] or [0]
with pa.Compute.new(pa.empty_particles(0), g) as c:
    c.update_sim_params(sp)
    lib = _lib.load()
    lib.pcl_debug_phase_clocks.argtypes = [C.c_void_p, C.c_void_p]
    lib.pcl_debug_experiment.argtypes = [C.c_void_p, C.c_int]
    for v, mode in [(v, m) for v in variants for m in modes]:
        c.set_tuning(0, v)
        _lib.check(lib.pcl_debug_experiment(c._h, 0))
        c.generate_particles(1 << 28, seed=12345, pos_extent=20.0, vel_extent=10.0)
        for _ in range(32):          # the launch of step 33 is the second steady-state scatter step (interval 16)
            c.compute()
        c.synchronize()
        _lib.check(lib.pcl_debug_experiment(c._h, mode))   # timing experiments (wrong results) on that launch only
        c.compute()
        c.synchronize()
        buf = np.zeros((256, 24), dtype=np.uint64)
        _lib.check(lib.pcl_debug_phase_clocks(c._h, buf.ctypes.data))
        t = buf.astype(np.int64)
        t = t[:128]                  # 2^28 particles = 131072 blocks = 128 sampled
        stamps = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16]
        print(f"variant {v:#x} mode {mode}: sampled blocks {len(t)}, kernel span {(t[:, 16].max() - t[:, 0].min()) / 100:.0f} us, "
              f"median block lifetime {np.median(t[:, 16] - t[:, 0]) / 100:.2f} us")
        for a, b in zip(stamps[:-1], stamps[1:]):
            d = (t[:, b] - t[:, a]) / 100.0
            print(f"   {NAMES[b]:45s} median {np.median(d):7.2f} us   p90 {np.percentile(d, 90):7.2f}")
