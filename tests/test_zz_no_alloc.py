"""pcl_diagnostics_device_noalloc (the entry point for callers that cannot
allocate) against pcl_diagnostics_device on the same state, through the C ABI:
two calls in a row, the second on a smaller particle set (the scratch array
keeps the first call's partials, which the second must not read)."""
import ctypes as C

import numpy as np
import pytest

import particles_amd as pa
from particles_amd import _lib
from conftest import scene_swirl_field

pytestmark = pytest.mark.gpu


def _record(t):
    host = t.cpu().numpy()
    return {"count": host[:16].view(np.float64)[0], "kinetic_energy": host[:16].view(np.float64)[1],
            "pos_max": host[16:28].view(np.float32).copy(), "neg_pos_min": host[28:40].view(np.float32).copy()}


def test_noalloc_diagnostics_equal_the_allocating_entry_point(oracle):
    import torch
    lib = _lib.load()
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.force_grid_dimensions = (64, 64, 1)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    g = scene_swirl_field(64, 64, 1, 10.0, seed=4)
    sizes = [(1 << 21) + 3, 5003]           # 4096 block partials, then 20
    free_before = None
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        out_new = torch.zeros(48, dtype=torch.uint8, device="cuda")
        out_old = torch.zeros(48, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for n in sizes:
            p = oracle.synth_particles(n, seed=n, pos_extent=20.0, vel_extent=10.0)
            c.upload_particles(p)
            c.compute()
            c.compute()
            c.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            rc = lib.pcl_diagnostics_device_noalloc(c._h, C.c_void_p(out_new.data_ptr()))
            assert rc == 0, lib.pcl_last_error()
            c.synchronize()
            assert torch.cuda.mem_get_info()[0] == free_before      # nothing was allocated
            rc = lib.pcl_diagnostics_device(c._h, C.c_void_p(out_old.data_ptr()))
            assert rc == 0, lib.pcl_last_error()
            c.synchronize()
            a, b = _record(out_new), _record(out_old)
            want = oracle.diagnostics(sp, c.read_particles())
            assert a["count"] == b["count"] == n == want["count"]
            assert np.array_equal(a["pos_max"], b["pos_max"]) and np.array_equal(a["pos_max"], want["pos_max"])
            assert np.array_equal(a["neg_pos_min"], b["neg_pos_min"]) and np.array_equal(-a["neg_pos_min"], want["pos_min"])
            # f64 sums in block order: equal to the allocating entry point's, and to the oracle's to 1e-12
            assert abs(a["kinetic_energy"] - b["kinetic_energy"]) <= 1e-12 * b["kinetic_energy"]
            assert abs(a["kinetic_energy"] - want["kinetic_energy"]) <= 1e-12 * want["kinetic_energy"]


def test_noalloc_diagnostics_rejects_bad_arguments():
    lib = _lib.load()
    sp = pa.SimParams()
    g = sp.new_force_grid_centered().get_force_vectors()
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        assert lib.pcl_diagnostics_device_noalloc(c._h, None) == _lib.PCL_ERR_INVALID
        assert b"dev_out" in lib.pcl_last_error()
