"""Conditions only a running kernel can see are reported, not hidden: the
refresh of the particle order checks every reservation against what its
histogram counted, the device brush checks the record it reads from device
memory, and the next synchronising call returns PCL_ERR_DEVICE once."""
import ctypes as C

import numpy as np
import pytest

import particles_amd as pa
from particles_amd import _lib
from conftest import assert_particles_bit_equal, scene_swirl_field

pytestmark = pytest.mark.gpu

DT = np.float32(1.0 / 60.0)


def test_scatter_with_keys_the_histogram_did_not_count_is_reported(oracle):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (128, 128, 1)
    g = scene_swirl_field(128, 128, 1, 10.0, seed=3)
    n = 200000
    p = oracle.synth_particles(n, seed=4, pos_extent=20.0, vel_extent=40.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_reorder_interval(3)
        for _ in range(3):                 # scatter, plain, counting step: a histogram is pending
            c.compute()
        c.synchronize()                    # nothing to report so far
        c.set_tuning(0, 0x40000)           # the scatter step keys with another lookahead than the histogram
        c.compute()
        with pytest.raises(pa.PclError) as e:
            c.synchronize()
        assert e.value.code == _lib.PCL_ERR_DEVICE and "histogram" in str(e.value)
        c.synchronize()                    # reported once
        # a fresh upload makes the handle usable again
        c.set_tuning(0, 0)
        c.upload_particles(p)
        for _ in range(4):
            c.compute()
        assert_particles_bit_equal(c.read_particles(), oracle.run(sp, p, g, 4, mode=oracle.MODE_SKIP_INERT), "after recovery")


def test_device_brush_record_with_unknown_mode_edits_nothing_and_is_reported():
    import torch
    sp = pa.SimParams()
    g = sp.new_force_grid_centered().get_force_vectors()
    with pa.Compute.new(pa.empty_particles(0), g) as c:
        c.update_sim_params(sp)
        rec = pa.EditMode(mode="shift", falloff_dist=3.0, strength=1.0).record((0.0, 0.0, 0.0))
        bad = rec.copy()
        bad["mode"] = 7
        with pytest.raises(pa.PclError) as e:        # the host variant looks at the record before the launch
            c.brush_mouse_down(bad)
        assert e.value.code == _lib.PCL_ERR_INVALID
        t = torch.from_numpy(bad.view(np.uint8).reshape(64).copy()).cuda()
        torch.cuda.synchronize()
        c.brush_mouse_down_device(t.data_ptr())
        c.brush_mouse_moved_device(t.data_ptr())
        with pytest.raises(pa.PclError) as e:
            c.synchronize()
        assert e.value.code == _lib.PCL_ERR_DEVICE and "cursor record" in str(e.value)
        assert c.brush_selection().size == 0
        assert np.array_equal(c.read_force_grid(), g)


@pytest.mark.parametrize("exponent", [60.0, -60.0])
def test_zero_curves_with_overflowing_pow_take_the_pair_kernel(oracle, exponent):
    """0 * pow(len, exponent) is NaN once pow overflows (src/compute.wgsl:139),
    so all-zero force curves are only a reason to skip the pair loop while pow
    stays finite; here it does not, and the result must be the full loop's."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.cut_off_distance = 5.0
    sp.distance_exponent = exponent
    assert not sp.pair_term_inert() and not oracle.pair_term_inert(sp)
    g = sp.new_force_grid_centered().get_force_vectors()
    p = oracle.synth_particles(700, seed=8, pos_extent=6.0, vel_extent=4.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.compute()
        assert c.last_step_kernel == pa.KERNEL_PAIRS
        got = c.read_particles()
    want = oracle.run(sp, p, g, 1, mode=oracle.MODE_ALWAYS_PAIRS)
    # NaN where the shader produces NaN, finite elsewhere, same places
    gn, wn = np.isnan(got["pos"][:, :3]).any(axis=1), np.isnan(want["pos"][:, :3]).any(axis=1)
    assert np.array_equal(gn, wn) and wn.any()
    assert_particles_bit_equal(got[~wn], want[~wn], f"exponent {exponent}, finite particles")
