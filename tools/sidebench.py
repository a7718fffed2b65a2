#!/usr/bin/env python3
"""Timing of the operations either side of the step (GPU only): instance-stream pack for the
renderer, brush events on the device field, host read-back."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa
from bench import swirl_field, make_params


def timed(c, fn, reps=10):
    fn()
    c.synchronize()
    c.timer_start()
    for _ in range(reps):
        fn()
    return c.timer_stop() / reps


def main():
    import torch
    dims = (1024, 1024, 1)
    n = 1 << 26
    with pa.Compute.new(pa.empty_particles(0), swirl_field(*dims, 10.0, 1)) as c:
        c.update_sim_params(make_params(pa, dims))
        c.generate_particles(n, 12345, 0, 20.0, 10.0)
        out = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        ms = timed(c, lambda: c.pack_particles_device(out.data_ptr(), 0, n))
        print(f"pack 2^26 particles into 48-byte instance records (caller order = memory order): {ms:.3f} ms, "
              f"{n * 73 / ms / 1e6:.0f} GB/s (25 B read + 48 B written per particle)")
        for _ in range(14):
            c.compute()
        ms = timed(c, lambda: c.pack_particles_device(out.data_ptr(), 0, n))
        print(f"same after the tile order took over (slots walked, records written at their caller index): {ms:.3f} ms")
        ms = timed(c, lambda: c.pack_particles_device_unordered(out.data_ptr()))
        print(f"tile order, records in memory order (pcl_pack_particles_device_unordered): {ms:.3f} ms")
        ms = timed(c, lambda: c.pack_particles_device(out.data_ptr(), 1000, 5000))
        print(f"tile order, 5000 records in caller order (gather): {ms * 1e3:.1f} us")
        em = pa.EditMode(mode="shift", falloff_dist=1.0)
        b = pa.DeviceBrush(c, em)
        ms = timed(c, lambda: b.mouse_down((0.0, 0.0, 0.0)))
        print(f"brush mouse_down over 1024^2 cells ({c.brush_selection().size} selected): {ms * 1e3:.1f} us")
        ms = timed(c, lambda: b.mouse_moved((0.1, 0.2, 0.0)))
        print(f"brush mouse_moved: {ms * 1e3:.1f} us")
        t0 = time.perf_counter()
        c.read_particles(0, 1 << 22)
        dt = time.perf_counter() - t0
        print(f"read_particles of 2^22 records to pageable host memory: {dt * 1e3:.1f} ms, {(1 << 22) * 48 / dt / 1e9:.1f} GB/s")


if __name__ == "__main__":
    main()
