"""What the ranks of a communicator agree on before they step together (pcl_comm_step, one
process per rank): asked once -- one all-reduce of five 64-bit words, read back on the host --
and again only after a call that replaced the particle set or changed what a step exchanges, not
on every step; ranks found to be about to issue different collectives when they ask get an error
on EVERY rank, not a hang.
Two processes on one GPU, tests/fake_rccl standing in for RCCL (which does not admit two ranks on
one device); its per-process call counter is what the counts below read."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FAKE_RCCL = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
NCCL_UINT64, NCCL_FLOAT, NCCL_DOUBLE = 5, 7, 8     # ncclDataType_t
N = 3001


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["PCL_RCCL_LIBRARY"] = FAKE_RCCL
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import particles_amd as pa
    from particles_amd import _lib
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fake = ctypes.CDLL(FAKE_RCCL)          # the same mapping the library resolves its entry points from
        fake.fake_rccl_allreduce_calls.restype = ctypes.c_ulonglong
        agreements = lambda: int(fake.fake_rccl_allreduce_calls(NCCL_UINT64))
        records = lambda: int(fake.fake_rccl_allreduce_calls(NCCL_DOUBLE))

        sp = pa.SimParams()
        sp.delta_t = np.float32(1.0 / 60.0)
        sp.force_grid_dimensions = (8, 8, 8)
        g = pa.Grid.new_centered(8, 8, 8, 10.0).get_force_vectors()
        p = O.synth_particles(N, seed=5, pos_extent=6.0, vel_extent=2.0).view(pa.PARTICLE_DTYPE)
        first, last = pa.shard_range(N, rank, world)
        c = pa.Compute.new(p[first:last], g)
        ids = [pa.Compute.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        c.comm_attach(ids[0], rank, world)
        c.update_sim_params(sp)
        seen = {}

        # pair term inert, diagnostics off: the first step asks, the next ones do not
        for _ in range(4):
            c.comm_step()
        seen["inert"] = agreements()
        # per-frame parameter updates that leave the exchange as it is do not ask again
        for k in range(3):
            sp.delta_t = np.float32(1.0 / (61.0 + k))
            c.update_sim_params(sp)
            c.comm_step()
        seen["same_exchange"] = agreements()
        # diagnostics on: one more question, then one grouped pair of all-reduces per step
        c.enable_step_diagnostics(True)
        for _ in range(3):
            c.comm_step()
        seen["diag"] = agreements()
        seen["records"] = records()
        # pair term on: one more; five steps, no further question
        sp.cut_off_distance = np.float32(1.5)
        sp.polys[:] = np.random.default_rng(3).uniform(-1, 1, size=(25, 8)).astype(np.float32)
        c.update_sim_params(sp)
        c.set_pair_mode(1)
        for _ in range(5):
            c.comm_step()
        seen["pairs"] = agreements()
        # a new particle set of the same size: asked again (the other rank's count may have changed)
        c.upload_particles(p[first:last])
        c.comm_step()
        c.comm_step()
        seen["upload"] = agreements()

        # the ranks disagree at a point where they all ask (after an upload): diagnostics off on rank 1
        # only -> both return an error, nobody waits for a collective the other never issues
        if rank == 1:
            c.enable_step_diagnostics(False)
        c.upload_particles(p[first:last])
        try:
            c.comm_step()
            seen["error"] = ""
        except _lib.PclError as e:
            seen["error"] = str(e)
        # and after the ranks are alike again the next step works
        c.enable_step_diagnostics(False)
        c.comm_step()
        seen["after"] = agreements()
        c.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **seen)
        c.close()
    finally:
        dist.destroy_process_group()


def test_ranks_agree_once_and_a_disagreement_is_an_error_on_every_rank(tmp_path):
    assert os.path.exists(FAKE_RCCL), "make builds tests/fake_rccl/libfake_rccl.so"
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        r = np.load(tmp_path / f"rank{rank}.npz")
        assert int(r["inert"]) == 1, "one question for the first step, none for the three after it"
        assert int(r["same_exchange"]) == 1, "a new dt does not change what a step exchanges"
        assert int(r["diag"]) == 2 and int(r["records"]) == 3
        assert int(r["pairs"]) == 3, "the pair term costs one question, not one per step"
        assert int(r["upload"]) == 4
        assert "step diagnostics are enabled on some ranks" in str(r["error"]), f"rank {rank}: {r['error']}"
        assert int(r["after"]) == 6    # the refused step asked, the step after it asked again


def _worker_beside(rank, world, port, out_dir):
    """Two ranks with step diagnostics on: after every pcl_comm_step the record that
    pcl_comm_diagnostics_device points at, fetched through the handle's stream with no host
    synchronisation in between, must be the record of ALL particles after that step."""
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["PCL_RCCL_LIBRARY"] = FAKE_RCCL
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import particles_amd as pa
    from particles_amd import _lib
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = _lib.load()
        n, dims = 200_001, (32, 32, 1)
        cells = dims[0] * dims[1]
        g = np.ones((cells, 4), dtype=np.float32)
        g[:, :3] = np.random.default_rng(8).uniform(-1, 1, size=(cells, 3)).astype(np.float32)
        sp = pa.SimParams()
        sp.delta_t = np.float32(1.0 / 60.0)
        sp.force_grid_dimensions = dims
        sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
        state = O.synth_particles(n, seed=5, pos_extent=20.0, vel_extent=10.0)
        first, last = pa.shard_range(n, rank, world)
        c = pa.Compute.new(state[first:last], g)
        ids = [pa.Compute.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        c.comm_attach(ids[0], rank, world)
        c.update_sim_params(sp)
        c.set_reorder_interval(5)
        c.enable_step_diagnostics(True)
        keep = np.ascontiguousarray(g[cells - 3:])
        worst = 0.0
        for step in range(12):
            c.comm_step()
            rec = ctypes.c_void_p()
            _lib.check(lib.pcl_comm_diagnostics_device(c._h, ctypes.byref(rec)))
            _lib.check(lib.pcl_patch_force_grid_device(c._h, cells - 3, rec, 3))
            d = _lib.Diagnostics.from_buffer_copy(c.read_force_grid(cells - 3, 3).tobytes())
            _lib.check(lib.pcl_patch_force_grid(c._h, cells - 3, keep.ctypes.data_as(ctypes.c_void_p), 3))
            state = O.step(sp, state, g, mode=O.MODE_SKIP_INERT)
            want = O.diagnostics(sp, state)
            assert int(d.count) == n, (rank, step, d.count)
            assert np.array_equal(np.array(d.pos_max[:], dtype=np.float32), want["pos_max"]), (rank, step)
            assert np.array_equal(-np.array(d.neg_pos_min[:], dtype=np.float32), want["pos_min"]), (rank, step)
            worst = max(worst, abs(d.kinetic_energy - want["kinetic_energy"]) / want["kinetic_energy"])
        assert worst <= 1e-11, worst
        got = c.read_particles()
        assert got.tobytes() == state[first:last].tobytes()
        np.savez(os.path.join(out_dir, f"beside{rank}.npz"), worst=worst, steps=12)
        c.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_the_all_reduce_beside_the_step_holds_the_global_record_of_that_step(tmp_path):
    assert os.path.exists(FAKE_RCCL), "make builds tests/fake_rccl/libfake_rccl.so"
    mp.spawn(_worker_beside, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        r = np.load(tmp_path / f"beside{rank}.npz")
        assert int(r["steps"]) == 12 and float(r["worst"]) <= 1e-11
