"""GPU tests of the sharded host layer: the gathered pair step through the
C ABI with two handles on one card, and ShardedCompute over RCCL in a one-rank
group (``force_collectives`` runs every broadcast / all-gather / all-reduce)."""
import os
import socket

import numpy as np
import pytest

import particles_amd as pa
from conftest import assert_particles_bit_equal, scene_swirl_field

pytestmark = pytest.mark.gpu

DT = np.float32(1.0 / 60.0)


def pair_scene(oracle, n, exponent=0.0, seed=21):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.cut_off_distance = np.float32(1.5)
    sp.distance_exponent = np.float32(exponent)
    sp.polys[:] = np.random.default_rng(seed).uniform(-1, 1, size=(25, 8)).astype(np.float32)
    p = oracle.synth_particles(n, seed=seed, pos_extent=6.0, vel_extent=4.0)
    g = sp.new_force_grid_centered().get_force_vectors()
    return sp, p, g


@pytest.mark.parametrize("n,split,exponent", [(1000, 500, 0.0), (1537, 1, 0.0), (4097, 4096, 0.0), (3000, 1234, -1.5)])
def test_gathered_pair_step_two_ranges_equal_one_handle(oracle, n, split, exponent):
    """Two handles hold [0, split) and [split, n); each packs its records,
    the blocks are concatenated (with padding, as an all-gather of equal
    blocks would) and each range is updated against all of them.  The result
    must equal one handle holding everything, bit for bit, over several steps."""
    import torch
    sp, p, g = pair_scene(oracle, n, exponent)
    cap = max(split, n - split) + 3            # a few padding records per block
    with pa.Compute.new(p, g) as whole, pa.Compute.new(p[:split], g) as lo, pa.Compute.new(p[split:], g) as hi:
        for c in (whole, lo, hi):
            c.update_sim_params(sp)
            c.set_pair_mode(1)
        records = torch.empty((2 * cap, 4), dtype=torch.float32, device="cuda")
        for _ in range(3):
            whole.compute()
            lo.pack_pair_sources_device(records[:cap].data_ptr(), cap)
            hi.pack_pair_sources_device(records[cap:].data_ptr(), cap)
            lo.synchronize()
            hi.synchronize()                 # different streams: both blocks complete before either step
            lo.step_pairs_gathered(records.data_ptr(), 2 * cap)
            hi.step_pairs_gathered(records.data_ptr(), 2 * cap)
            lo.synchronize()
            hi.synchronize()
        got = np.concatenate([lo.read_particles(), hi.read_particles()])
        want = whole.read_particles()
        assert lo.last_step_kernel == pa.KERNEL_PAIRS
    assert_particles_bit_equal(got, want, "two gathered ranges vs one handle")
    if exponent == 0.0:
        assert_particles_bit_equal(got, oracle.run(sp, p, g, 3), "gathered ranges vs oracle")


def test_gathered_pair_step_rejects_bad_arguments(oracle):
    import torch
    sp, p, g = pair_scene(oracle, 100)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        buf = torch.empty((100, 4), dtype=torch.float32, device="cuda")
        with pytest.raises(pa.PclError):
            c.pack_pair_sources_device(buf.data_ptr(), 99)          # capacity below the particle count
        with pytest.raises(pa.PclError):
            c.pack_pair_sources_device(buf.data_ptr() + 4, 100)     # misaligned
        with pytest.raises(pa.PclError):
            c.step_pairs_gathered(0, 100)                           # null with a non-zero count
        c.step_pairs_gathered(0, 0)                                 # no partners at all: field term only
        got = c.read_particles()
    sp0 = sp.copy()
    sp0.polys[:] = 0
    assert_particles_bit_equal(got, oracle.run(sp0, p, g, 1), "empty partner list")


@pytest.fixture(scope="module")
def one_rank_rccl():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_sharded_stream_path_over_rccl(oracle, one_rank_rccl):
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.force_grid_dimensions = (64, 64, 1)
    g = scene_swirl_field(64, 64, 1, 10.0, seed=4)
    p = oracle.synth_particles(50001, seed=8, pos_extent=20.0, vel_extent=10.0)
    sc = pa.ShardedCompute.new(p.view(pa.PARTICLE_DTYPE), g, force_collectives=True)
    try:
        sc.update_sim_params(sp)
        sc.enable_step_diagnostics(True)
        for _ in range(4):
            sc.compute()
        patch = np.full((100, 4), 1.25, dtype=np.float32)
        sc.patch_force_grid(1000, patch)
        sc.compute()
        got = sc.gather_particles()
        diag = sc.diagnostics()
    finally:
        sc.close()
    want = oracle.run(sp, p, g, 4, mode=oracle.MODE_SKIP_INERT)
    g2 = g.copy()
    g2[1000:1100] = 1.25
    want = oracle.run(sp, want, g2, 1, mode=oracle.MODE_SKIP_INERT)
    assert_particles_bit_equal(got, want, "sharded stream path over RCCL")
    wd = oracle.diagnostics(sp, want)
    assert diag["count"] == 50001
    assert abs(diag["kinetic_energy"] - wd["kinetic_energy"]) <= 1e-9 * wd["kinetic_energy"]
    assert np.array_equal(diag["pos_max"], wd["pos_max"]) and np.array_equal(diag["pos_min"], wd["pos_min"])


def test_sharded_pair_path_all_gather_over_rccl(oracle, one_rank_rccl):
    sp, p, g = pair_scene(oracle, 2001)
    sc = pa.ShardedCompute.new(p.view(pa.PARTICLE_DTYPE), g, force_collectives=True)
    try:
        sc.update_sim_params(sp)
        for _ in range(2):
            sc.compute()
        sc.step(1.0 / 120.0)
        assert sc.local.last_step_kernel == pa.KERNEL_PAIRS
        got = sc.gather_particles()
    finally:
        sc.close()
    want = oracle.run(sp, p, g, 2)
    sp.delta_t = np.float32(1.0 / 120.0)
    want = oracle.run(sp, want, g, 1)
    assert_particles_bit_equal(got, want, "sharded pair path (all-gather) over RCCL")


def test_sharded_device_brush_over_rccl(oracle, one_rank_rccl):
    """Config-5 frame through the sharded layer: the cursor record is broadcast
    as device bytes and consumed by the brush kernels, then one update."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.force_grid_dimensions = (64, 64, 1)
    g = scene_swirl_field(64, 64, 1, 10.0, seed=6)
    p = oracle.synth_particles(20000, seed=12, pos_extent=20.0, vel_extent=10.0)
    em = pa.EditMode(mode="rotate", falloff="inverse_distance", falloff_dist=2.5, strength=0.8)
    m = oracle.edit_mode("rotate", True, "inverse_distance", 2.5, 0.8)
    sc = pa.ShardedCompute.new(p.view(pa.PARTICLE_DTYPE), g, force_collectives=True)
    try:
        sc.update_sim_params(sp)
        brush = pa.DeviceBrush(sc, em)
        brush.mouse_down((0.5, -0.5, 0.0))
        idx, vec = oracle.brush_mouse_down(sp, g, (0.5, -0.5, 0.0), 2.5)
        want_p, want_g = p, g
        for k in range(4):
            cursor = (0.5 + 0.2 * k, -0.5 + 0.1 * k, 0.0)
            brush.mouse_moved(cursor)
            sc.compute()
            want_g = oracle.brush_mouse_moved(sp, want_g, m, idx, vec, (0.5, -0.5, 0.0), cursor)
            want_p = oracle.step(sp, want_p, want_g, mode=oracle.MODE_SKIP_INERT)
        got_p, got_g = sc.gather_particles(), sc.read_force_grid()
    finally:
        sc.close()
    assert got_g.tobytes() == want_g.tobytes()
    assert_particles_bit_equal(got_p, want_p, "sharded device brush over RCCL")


def test_sharded_native_collectives(oracle, one_rank_rccl):
    """ShardedCompute with the collectives inside the library (pcl_comm_*): the
    process group only carries the unique id; brush, patches with and without a
    size header, diagnostics and the pair term's all-gather all run as RCCL
    calls on the handle's own stream."""
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.force_grid_dimensions = (64, 64, 1)
    g = scene_swirl_field(64, 64, 1, 10.0, seed=6)
    p = oracle.synth_particles(30001, seed=12, pos_extent=20.0, vel_extent=10.0)
    em = pa.EditMode(mode="shift", falloff="linear", falloff_dist=2.0, strength=0.5)
    m = oracle.edit_mode("shift", True, "linear", 2.0, 0.5)
    sc = pa.ShardedCompute.new(p.view(pa.PARTICLE_DTYPE), g, collectives="native", n_cells=64 * 64)
    try:
        assert sc.local._lib.pcl_comm_world(sc.local._h) == 1
        sc.update_sim_params(sp)
        sc.enable_step_diagnostics(True)
        brush = pa.DeviceBrush(sc, em)
        brush.mouse_down((0.5, -0.5, 0.0))
        idx, vec = oracle.brush_mouse_down(sp, g, (0.5, -0.5, 0.0), 2.0)
        brush.mouse_moved((1.0, 0.0, 0.0))
        want_g = oracle.brush_mouse_moved(sp, g, m, idx, vec, (0.5, -0.5, 0.0), (1.0, 0.0, 0.0))
        patch = np.full((50, 4), 0.75, dtype=np.float32)
        sc.patch_force_grid(2000, patch, count=50)           # every rank knows the size: no header
        sc.patch_force_grid(3000, patch[:7])                 # size carried by a header
        sc.patch_force_grid_indexed(np.array([5, 4095], dtype=np.uint32), patch[:2])
        want_g[2000:2050] = 0.75
        want_g[3000:3007] = 0.75
        want_g[[5, 4095]] = 0.75
        for _ in range(3):
            sc.compute()
        sc.compute_n(4)
        want = oracle.run(sp, p, want_g, 7, mode=oracle.MODE_SKIP_INERT)
        assert sc.read_force_grid().tobytes() == want_g.tobytes()
        assert_particles_bit_equal(sc.gather_particles(), want, "native collectives, streaming")
        d, wd = sc.diagnostics(), oracle.diagnostics(sp, want)
        assert d["count"] == 30001 and abs(d["kinetic_energy"] - wd["kinetic_energy"]) <= 1e-12 * wd["kinetic_energy"]
        assert np.array_equal(d["pos_max"], wd["pos_max"]) and np.array_equal(d["pos_min"], wd["pos_min"])
        # pair term: the all-gather inside pcl_comm_step
        sp2, p2, g2 = pair_scene(oracle, 1501)
        sc.upload_particles(p2.view(pa.PARTICLE_DTYPE))
        sc.update_force_grid(g2)
        sc.update_sim_params(sp2)
        sc.local.set_pair_mode(pa.PAIRS_ALL)
        sc.compute()
        sc.step(1.0 / 120.0)
        assert sc.local.last_step_kernel == pa.KERNEL_PAIRS
        want = oracle.run(sp2, p2, g2, 1)
        sp2.delta_t = np.float32(1.0 / 120.0)
        want = oracle.run(sp2, want, g2, 1)
        assert_particles_bit_equal(sc.gather_particles(), want, "native collectives, pair steps")
    finally:
        sc.close()


@pytest.mark.parametrize("tile_order", [False, True])
def test_instance_stream_pack_on_the_device(oracle, tile_order):
    """pcl_pack_particles_device[_unordered]: the 48-byte records the renderer
    reads (src/main.rs:86-107), written into device memory, in caller order
    (whole set, large and small ranges) and in memory order."""
    import torch
    sp = pa.SimParams()
    sp.delta_t = DT
    sp.force_grid_dimensions = (64, 64, 1)
    g = scene_swirl_field(64, 64, 1, 10.0, seed=6)
    n = 20003
    p = oracle.synth_particles(n, seed=14, pos_extent=20.0, vel_extent=10.0)
    with pa.Compute.new(p, g) as c:
        c.update_sim_params(sp)
        c.set_reorder_interval(2 if tile_order else 0)
        for _ in range(3):
            c.compute()
        assert (c.reorder_count > 0) == tile_order
        want = oracle.run(sp, p, g, 3, mode=oracle.MODE_SKIP_INERT).view(np.uint8).reshape(n, 48)
        buf = torch.zeros((n, 48), dtype=torch.uint8, device="cuda")
        for first, count in ((0, n), (5, n - 9), (1000, 700), (n - 1, 1)):
            buf.zero_()
            torch.cuda.synchronize()
            c.pack_particles_device(buf.data_ptr(), first, count)
            c.synchronize()
            got = buf.cpu().numpy()
            assert got[:count].tobytes() == want[first:first + count].tobytes(), (first, count)
            assert not got[count:].any()                    # nothing written past the range
        buf.zero_()
        torch.cuda.synchronize()
        c.pack_particles_device_unordered(buf.data_ptr())
        c.synchronize()
        got = buf.cpu().numpy()
        key = lambda a: a[np.lexsort(a.T[::-1])]            # every record exactly once, any order
        assert key(got).tobytes() == key(want).tobytes()
        if not tile_order:
            assert got.tobytes() == want.tobytes()
