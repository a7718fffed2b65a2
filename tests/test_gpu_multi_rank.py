"""Three ranks on ONE GPU: three processes, each with its own handle of the HIP
library on device 0 and its own index range, talking over a host-memory
process group (gloo; RCCL does not admit several ranks on one device).  The
whole sharded path with the real backend and more than one rank: field
replication from rank 0, range ownership, device-generated shards seeded by
global index, the brush replicated on every rank's field, field patches, diagnostics
all-reduce, and the pair term's all-gather — each compared with the unsharded
CPU oracle, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORLD = 3
N = 30011            # ranges of 10003, 10004, 10004
N_PAIRS = 2501
PATCH = np.arange(40 * 4, dtype=np.float32).reshape(40, 4) / np.float32(16.0) - np.float32(5.0)
PATCH_IDX = np.array([9215, 0, 4607], dtype=np.uint32)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _scene(pa, O):
    sys.path.insert(0, HERE)
    from conftest import scene_swirl_field
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    sp.force_grid_dimensions = (96, 96, 1)
    g = scene_swirl_field(96, 96, 1, 10.0, seed=4)
    return sp, g


def _pair_params(sp):
    sp = sp.copy()
    sp.cut_off_distance = np.float32(1.25)
    sp.polys[:] = np.random.default_rng(8).uniform(-1, 1, size=(25, 8)).astype(np.float32)
    return sp


def _brush(pa):
    return pa.EditMode(mode="rotate", relative=False, falloff="linear", falloff_dist=2.0, strength=1.5)


FAKE_RCCL = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")


def _worker(rank, world, port, out_dir, backend="gloo"):
    sys.path.insert(0, os.path.dirname(HERE))
    collectives = "torch"
    if backend == "native-stand-in":
        # the library's own collectives (pcl_comm_*), RCCL replaced by the host-staged stand-in so that the
        # ranks can share the GPU; the gloo group only carries the unique id and the test's gathers
        os.environ["PCL_RCCL_LIBRARY"] = FAKE_RCCL
        backend, collectives = "gloo", "native"
    import torch
    import torch.distributed as dist
    import particles_amd as pa
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = 0
    if backend == "nccl":            # RCCL: one GPU per rank
        device = rank
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sp, g = _scene(pa, O)
        # every rank generates its own range on the device; the field comes from rank 0
        sc = pa.ShardedCompute.new(None, g if rank == 0 else None, n_total=N, seed=31, pos_extent=20.0,
                                   vel_extent=10.0, device=device, collectives=collectives)
        assert type(sc.local) is pa.Compute and (sc.first, sc.last) == pa.shard_range(N, rank, world)
        sc.local.set_reorder_interval(3)            # tile order inside every range
        sc.update_sim_params(sp)
        sc.enable_step_diagnostics(True)
        for _ in range(4):
            sc.compute()
        mode = _brush(pa)
        sc.brush_mouse_down(mode.record((1.0, -2.0, 0.0)) if rank == 0 else None)
        sc.brush_mouse_moved(mode.record((1.5, -1.0, 0.0)) if rank == 0 else None)
        sc.brush_mouse_up()
        # dirty range and list of cells from rank 0 only
        sc.patch_force_grid(4000, PATCH if rank == 0 else None)
        sc.patch_force_grid_indexed(PATCH_IDX if rank == 0 else None, PATCH[:3] if rank == 0 else None)
        sc.compute_n(5)
        stream = sc.gather_particles()
        diag = sc.diagnostics()
        field = sc.read_force_grid()
        reorders = sc.local.reorder_count

        # the pair term across ranks: a smaller set, uploaded from the host
        p = O.synth_particles(N_PAIRS, seed=32, pos_extent=6.0, vel_extent=4.0).view(pa.PARTICLE_DTYPE)
        sc.upload_particles(p)
        sc.local.set_pair_mode(1)
        sc.update_sim_params(_pair_params(sp))
        for _ in range(3):
            sc.compute()
        pairs = sc.gather_particles()
        # two particles over three ranks: rank 0's range is empty, it still takes part in everything
        sc.upload_particles(p[:2])
        first, last = pa.shard_range(2, rank, world)
        assert sc.num_local_particles == last - first and (world != 3 or sc.num_local_particles == (0, 1, 1)[rank])
        sc.compute()
        sparse = sc.gather_particles()
        sparse_count = sc.diagnostics()["count"]
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), stream=stream, pairs=pairs, field=field, sparse=sparse,
                 sparse_count=sparse_count,
                 count=diag["count"], ke=diag["kinetic_energy"], pos_max=diag["pos_max"], pos_min=diag["pos_min"],
                 reorders=reorders, kernel=sc.local.last_step_kernel)
        sc.close()
    finally:
        dist.destroy_process_group()


def test_three_ranks_on_one_gpu_match_the_unsharded_oracle(tmp_path, oracle):
    _run_and_compare(tmp_path, oracle, WORLD, "gloo")


def test_three_ranks_through_the_librarys_collectives_over_the_rccl_stand_in(tmp_path, oracle):
    """The same script with `collectives="native"`: every broadcast, all-reduce and all-gather is a
    pcl_comm_* call inside the library, in a world of three (ncclCommInitRank with a shared id), with
    tests/fake_rccl standing in for RCCL, which does not admit several ranks on one GPU."""
    assert os.path.exists(FAKE_RCCL), "make builds tests/fake_rccl/libfake_rccl.so"
    _run_and_compare(tmp_path, oracle, WORLD, "native-stand-in")


def test_two_ranks_over_rccl_match_the_unsharded_oracle(tmp_path, oracle):
    """The same script with one GPU per rank and every collective over RCCL
    (field and cursor-record broadcasts, diagnostics all-reduces, the pair
    term's all-gather, all ordered against the kernels on the device's null
    stream).  Needs two GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank; this box has one")
    _run_and_compare(tmp_path, oracle, 2, "nccl")


def _run_and_compare(tmp_path, oracle, world, backend):
    import particles_amd as pa
    sys.path.insert(0, HERE)
    from oracle_backend import OracleCompute
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), backend), nprocs=world, join=True)

    sp, g = _scene(pa, oracle)
    model = OracleCompute.new(oracle.synth_particles(N, seed=31, pos_extent=20.0, vel_extent=10.0), g)
    model.update_sim_params(sp)
    for _ in range(4):
        model.compute()
    mode = _brush(pa)
    model.brush_mouse_down(mode.record((1.0, -2.0, 0.0)))
    model.brush_mouse_moved(mode.record((1.5, -1.0, 0.0)))
    model.brush_mouse_up()
    model.patch_force_grid(4000, PATCH)
    model.patch_force_grid_indexed(PATCH_IDX, PATCH[:3])
    model.step_n(5)
    want_stream = model.read_particles()
    want_diag = model.diagnostics()
    want_field = model.read_force_grid()
    model.upload_particles(oracle.synth_particles(N_PAIRS, seed=32, pos_extent=6.0, vel_extent=4.0))
    model.update_sim_params(_pair_params(sp))
    for _ in range(3):
        model.compute()
    want_pairs = model.read_particles()
    model.upload_particles(oracle.synth_particles(N_PAIRS, seed=32, pos_extent=6.0, vel_extent=4.0)[:2])
    model.compute()
    want_sparse = model.read_particles()
    assert not np.array_equal(want_field, g)                      # the brush did change cells

    for rank in range(world):
        r = np.load(tmp_path / f"rank{rank}.npz")
        assert r["stream"].tobytes() == want_stream.tobytes(), f"rank {rank}: streaming path"
        assert r["field"].tobytes() == want_field.tobytes(), f"rank {rank}: field after the brush"
        assert r["pairs"].tobytes() == want_pairs.tobytes(), f"rank {rank}: pair path"
        assert r["sparse"].tobytes() == want_sparse.tobytes() and int(r["sparse_count"]) == 2, f"rank {rank}: empty range"
        assert int(r["kernel"]) == pa.KERNEL_PAIRS and int(r["reorders"]) >= 2
        assert int(r["count"]) == N
        assert abs(float(r["ke"]) - want_diag["kinetic_energy"]) <= 1e-12 * want_diag["kinetic_energy"]
        assert np.array_equal(r["pos_max"], want_diag["pos_max"]) and np.array_equal(r["pos_min"], want_diag["pos_min"])
