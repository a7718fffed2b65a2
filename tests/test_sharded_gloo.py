"""World-size-2 test of the particle-sharded host layer over gloo (CPU).

The single-GPU backend is replaced by the oracle-backed stand-in of
tests/oracle_backend.py; what is under test is the sharding itself: index
ranges, field broadcast from rank 0, dirty-range patch broadcast, diagnostics
all-reduce, and that N-way sharding reproduces the unsharded state exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import particles_amd as pa
    from oracle import oracle as O
    from oracle_backend import OracleCompute

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sp = pa.SimParams()
        sp.delta_t = np.float32(1.0 / 60.0)
        sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
        n = 1001  # not divisible by the world size
        particles = O.synth_particles(n, seed=3, pos_extent=20.0, vel_extent=10.0).view(pa.PARTICLE_DTYPE)
        grid0 = sp.new_force_grid_centered().get_force_vectors()
        # only rank 0 holds the real field; the others must receive it
        sc = pa.ShardedCompute.new(particles, grid0 if rank == 0 else None, compute_cls=OracleCompute)
        assert (sc.first, sc.last) == pa.shard_range(n, rank, world)
        sc.update_sim_params(sp)
        sc.enable_step_diagnostics(True)
        for _ in range(3):
            sc.compute()
        # brush edit on rank 0 only, dirty range broadcast
        patch = np.full((50, 4), 2.5, dtype=np.float32)
        sc.patch_force_grid(400, patch if rank == 0 else None)
        sc.compute()
        # brush-style list of cells, also from rank 0 only
        idx = np.array([7, 999, 123], dtype=np.uint32)
        vals = np.array([[1, 2, 3, 1], [4, 5, 6, 1], [7, 8, 9, 1]], dtype=np.float32)
        sc.patch_force_grid_indexed(idx if rank == 0 else None, vals if rank == 0 else None)
        sc.compute()
        full = sc.gather_particles()
        diag = sc.diagnostics()
        # a whole-field replace and a device-generated shard
        sc.update_force_grid(grid0 if rank == 0 else None)
        sc.compute()
        full2 = sc.gather_particles()
        sc.compute_n(3)                      # three updates without looking in between
        full3 = sc.gather_particles()
        sc2 = pa.ShardedCompute.new(None, grid0 if rank == 0 else None, compute_cls=OracleCompute,
                                    n_total=777, seed=11, pos_extent=20.0)
        gen = sc2.gather_particles()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), full=full, full2=full2, full3=full3, gen=gen,
                 count=diag["count"], ke=diag["kinetic_energy"], pos_max=diag["pos_max"], pos_min=diag["pos_min"])
        sc.close()
        sc2.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_unsharded(tmp_path, oracle):
    import particles_amd as pa
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)

    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    p = oracle.synth_particles(1001, seed=3, pos_extent=20.0, vel_extent=10.0)
    g = sp.new_force_grid_centered().get_force_vectors()
    want = oracle.run(sp, p, g, 3, mode=oracle.MODE_SKIP_INERT)
    g2 = g.copy()
    g2[400:450] = 2.5
    want = oracle.run(sp, want, g2, 1, mode=oracle.MODE_SKIP_INERT)
    g2[[7, 999, 123]] = np.array([[1, 2, 3, 1], [4, 5, 6, 1], [7, 8, 9, 1]], dtype=np.float32)
    want = oracle.run(sp, want, g2, 1, mode=oracle.MODE_SKIP_INERT)
    want_diag = oracle.diagnostics(sp, want)
    want2 = oracle.run(sp, want, g, 1, mode=oracle.MODE_SKIP_INERT)
    want3 = oracle.run(sp, want2, g, 3, mode=oracle.MODE_SKIP_INERT)
    want_gen = oracle.synth_particles(777, seed=11, pos_extent=20.0)

    for rank in range(world):
        r = np.load(tmp_path / f"rank{rank}.npz")
        assert r["full"].tobytes() == want.tobytes()          # byte-identical to the unsharded run
        assert r["full2"].tobytes() == want2.tobytes()
        assert r["full3"].tobytes() == want3.tobytes()
        assert r["gen"].tobytes() == want_gen.tobytes()       # global indices seed the generator
        assert int(r["count"]) == 1001
        assert abs(float(r["ke"]) - want_diag["kinetic_energy"]) <= 1e-12 * want_diag["kinetic_energy"]
        assert np.array_equal(r["pos_max"], want_diag["pos_max"])
        assert np.array_equal(r["pos_min"], want_diag["pos_min"])


def _pair_scene(pa, O, n):
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.cut_off_distance = np.float32(1.5)
    sp.polys[:] = np.random.default_rng(5).uniform(-1, 1, size=(25, 8)).astype(np.float32)
    particles = O.synth_particles(n, seed=9, pos_extent=4.0, vel_extent=2.0)
    grid = sp.new_force_grid_centered().get_force_vectors()
    return sp, particles, grid


def _pair_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import particles_amd as pa
    from oracle import oracle as O
    from oracle_backend import OracleCompute

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sp, particles, grid = _pair_scene(pa, O, 301)   # ranges of 150 and 151: one padding record travels
        sc = pa.ShardedCompute.new(particles.view(pa.PARTICLE_DTYPE), grid if rank == 0 else None,
                                   compute_cls=OracleCompute)
        sc.update_sim_params(sp)
        for _ in range(2):
            sc.compute()
        sc.step(1.0 / 120.0)
        full = sc.gather_particles()
        np.savez(os.path.join(out_dir, f"pair_rank{rank}.npz"), full=full)
        sc.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_pair_term_matches_unsharded(tmp_path, oracle):
    """Pair term active: ranks exchange positions and types every step and
    reproduce the single-range result byte for byte."""
    import particles_amd as pa
    world = 2
    mp.spawn(_pair_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sp, p, g = _pair_scene(pa, oracle, 301)
    assert not sp.pair_term_inert()
    want = oracle.run(sp, p, g, 2)
    sp.delta_t = np.float32(1.0 / 120.0)
    want = oracle.run(sp, want, g, 1)
    for rank in range(world):
        r = np.load(tmp_path / f"pair_rank{rank}.npz")
        assert r["full"].tobytes() == want.tobytes()


def _brush_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import particles_amd as pa
    from oracle import oracle as O
    from oracle_backend import OracleCompute

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sp = pa.SimParams()
        sp.delta_t = np.float32(1.0 / 60.0)
        particles = O.synth_particles(500, seed=4, pos_extent=20.0, vel_extent=10.0).view(pa.PARTICLE_DTYPE)
        grid0 = sp.new_force_grid_centered().get_force_vectors()
        sc = pa.ShardedCompute.new(particles, grid0 if rank == 0 else None, compute_cls=OracleCompute)
        sc.update_sim_params(sp)
        em = pa.EditMode(mode="centered", falloff="linear", falloff_dist=4.0, strength=2.0)
        # only rank 0 knows where the cursor is
        sc.brush_mouse_down(em.record((1.0, 2.0, 3.0)) if rank == 0 else None)
        for k in range(3):
            sc.brush_mouse_moved(em.record((1.0 + k, 2.0, 3.0 - k)) if rank == 0 else None)
            sc.compute()
        sc.brush_mouse_up()
        sc.brush_mouse_moved(em.record((0.0, 0.0, 0.0)) if rank == 0 else None)   # no effect
        np.savez(os.path.join(out_dir, f"brush_rank{rank}.npz"), full=sc.gather_particles(), field=sc.read_force_grid())
        sc.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_brush_record_broadcast(tmp_path, oracle):
    """The brush runs on every replica of the field from rank 0's 64-byte record."""
    import particles_amd as pa
    world = 2
    mp.spawn(_brush_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    p = oracle.synth_particles(500, seed=4, pos_extent=20.0, vel_extent=10.0)
    g = sp.new_force_grid_centered().get_force_vectors()
    m = oracle.edit_mode("centered", True, "linear", 4.0, 2.0)
    idx, vec = oracle.brush_mouse_down(sp, g, (1.0, 2.0, 3.0), 4.0)
    assert idx.size == 28
    for k in range(3):
        g = oracle.brush_mouse_moved(sp, g, m, idx, vec, (1.0, 2.0, 3.0), (1.0 + k, 2.0, 3.0 - k))
        p = oracle.step(sp, p, g, mode=oracle.MODE_SKIP_INERT)
    for rank in range(world):
        r = np.load(tmp_path / f"brush_rank{rank}.npz")
        assert r["field"].tobytes() == g.tobytes()
        assert r["full"].tobytes() == p.tobytes()


@pytest.mark.parametrize("n,world", [(0, 1), (1, 2), (10, 3), (1001, 2), (1 << 30, 8), (7, 8)])
def test_shard_ranges_partition_the_set(n, world):
    import particles_amd as pa
    edges = [pa.shard_range(n, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n
    for a, b in zip(edges, edges[1:]):
        assert a[1] == b[0]
    sizes = [b - a for a, b in edges]
    assert max(sizes) - min(sizes) <= 1


def _sparse_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import particles_amd as pa
    from oracle import oracle as O
    from oracle_backend import OracleCompute

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sp, particles, grid = _pair_scene(pa, O, 2)     # two particles over three ranks: rank 0 owns none
        sc = pa.ShardedCompute.new(particles.view(pa.PARTICLE_DTYPE), grid if rank == 0 else None,
                                   compute_cls=OracleCompute)
        assert sc.num_local_particles == (0, 1, 1)[rank]
        sc.enable_step_diagnostics(True)
        inert = pa.SimParams()
        inert.delta_t = sp.delta_t
        sc.update_sim_params(inert)
        sc.compute()                                     # streaming path with an empty range
        sc.update_sim_params(sp)
        sc.compute()                                     # pair path: rank 0 contributes padding only
        d = sc.diagnostics()
        np.savez(os.path.join(out_dir, f"sparse_rank{rank}.npz"), full=sc.gather_particles(), count=d["count"],
                 ke=d["kinetic_energy"])
        sc.close()
    finally:
        dist.destroy_process_group()


def test_more_ranks_than_particles(tmp_path, oracle):
    """Ranks whose index range is empty take part in every collective."""
    import particles_amd as pa
    world = 3
    mp.spawn(_sparse_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sp, p, g = _pair_scene(pa, oracle, 2)
    inert = pa.SimParams()
    inert.delta_t = sp.delta_t
    want = oracle.run(sp, oracle.run(inert, p, g, 1, mode=oracle.MODE_SKIP_INERT), g, 1)
    ke = oracle.diagnostics(sp, want)["kinetic_energy"]
    for rank in range(world):
        r = np.load(tmp_path / f"sparse_rank{rank}.npz")
        assert r["full"].tobytes() == want.tobytes()
        assert int(r["count"]) == 2 and abs(float(r["ke"]) - ke) <= 1e-12 * max(ke, 1e-300)
