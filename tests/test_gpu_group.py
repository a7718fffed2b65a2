"""RCCL inside the library: pcl_group_* (one process, all visible GPUs) and
pcl_comm_* (one handle per process) driven through every collective — field
broadcast, cursor-record broadcast, dirty-range and indexed patches, the
diagnostics all-reduce, the pair term's all-gather — and compared with a CPU
model of ONE device (the oracle), byte for byte.  On a one-GPU box the
communicator has one rank; every collective still runs through RCCL."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import particles_amd as pa
from particles_amd import _lib
from conftest import assert_particles_bit_equal
from oracle_backend import OracleCompute

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NX, NY, NZ, N, N_PAIRS = 24, 16, 2, 3001, 601
CELLS = NX * NY * NZ


def build(tmp_path):
    exe = tmp_path / "group_scene"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "group_scene.c"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "particles_amd"), "-lparticles_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "particles_amd"), "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def test_group_example_builds_as_strict_c11(tmp_path):
    assert build(tmp_path).exists()


class Lcg:
    """The generator of examples/group_scene.c."""
    def __init__(self, seed):
        self.s = seed

    def unit(self):
        self.s = (self.s * 1664525 + 1013904223) & 0xFFFFFFFF
        return np.float32(self.s >> 8) * np.float32(1.0 / 16777216.0)


def scene_inputs():
    r = Lcg(2024)
    p = pa.empty_particles(N)
    for i in range(N):
        for c in range(3):
            p["pos"][i, c] = np.float32(20.0) * (r.unit() - np.float32(0.5))
            p["vel"][i, c] = np.float32(10.0) * (r.unit() - np.float32(0.5))
        p["pos"][i, 3] = p["vel"][i, 3] = 1.0
        p["ty"][i] = i % 5
    g = np.ones((CELLS, 4), dtype=np.float32)
    for i in range(CELLS):
        for c in range(3):
            g[i, c] = np.float32(2.0) * (r.unit() - np.float32(0.5))
    return p, g, r


def model(p, g, r):
    """The same script on the CPU oracle, one device."""
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.force_grid_dimensions = (NX, NY, NZ)
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    m = OracleCompute.new(p, g)
    m.update_sim_params(sp)
    for _ in range(3):
        m.compute()
    mode = pa.EditMode(mode="rotate", relative=False, falloff="linear", falloff_dist=4.0, strength=1.5)
    m.brush_mouse_down(mode.record((1.0, -2.0, 0.5)))
    m.brush_mouse_moved(mode.record((1.5, -1.0, 0.5)))
    m.brush_mouse_up()
    patch = (np.arange(160, dtype=np.float32).reshape(40, 4) / np.float32(16.0) - np.float32(5.0)).astype(np.float32)
    m.patch_force_grid(100, patch)
    m.patch_force_grid_indexed(np.array([CELLS - 1, 0, 333], dtype=np.uint32), patch[:3])
    for _ in range(4):
        m.compute()
    stream, field, sdiag = m.read_particles(), m.read_force_grid(), m.diagnostics()
    m.upload_particles(p[:N_PAIRS])
    sp.cut_off_distance = 6.0
    for k in range(25):
        for c in range(3):
            sp.polys[k, c] = np.float32(0.05) * (r.unit() - np.float32(0.5))
    m.update_sim_params(sp)
    for _ in range(3):
        m.compute()
    return sp, stream, field, sdiag, m.read_particles(), m.diagnostics()


def check_diag(got, want):
    assert got["count"] == want["count"]
    assert abs(got["kinetic_energy"] - want["kinetic_energy"]) <= 1e-12 * want["kinetic_energy"]
    assert np.array_equal(got["pos_max"], want["pos_max"]) and np.array_equal(got["pos_min"], want["pos_min"])


def diag_of(buf):
    d = _lib.Diagnostics.from_buffer_copy(buf)
    return {"count": int(d.count), "kinetic_energy": float(d.kinetic_energy),
            "pos_max": np.array(d.pos_max[:], dtype=np.float32), "pos_min": -np.array(d.neg_pos_min[:], dtype=np.float32)}


def check_program_output(raw, what):
    p, g, r = scene_inputs()
    sp, stream, field, sdiag, pairs, pdiag = model(p, g, r)
    o = 0
    assert raw[o:o + 912] == sp.tobytes(); o += 912
    got = np.frombuffer(raw[o:o + 48 * N], dtype=pa.PARTICLE_DTYPE); o += 48 * N
    assert_particles_bit_equal(got, stream, f"streaming steps + brush + patches over {what}")
    got_field = np.frombuffer(raw[o:o + 16 * CELLS], dtype=np.float32).reshape(CELLS, 4); o += 16 * CELLS
    assert got_field.tobytes() == field.tobytes() and not np.array_equal(field, g)
    check_diag(diag_of(raw[o:o + 48]), sdiag); o += 48
    got = np.frombuffer(raw[o:o + 48 * N_PAIRS], dtype=pa.PARTICLE_DTYPE); o += 48 * N_PAIRS
    assert_particles_bit_equal(got, pairs, f"pair steps (all-gather) over {what}")
    check_diag(diag_of(raw[o:o + 48]), pdiag); o += 48
    assert o == len(raw)


@pytest.mark.gpu
def test_plain_c_program_drives_a_group_through_every_collective(tmp_path, oracle):
    out = tmp_path / "group.bin"
    text = subprocess.check_output([str(build(tmp_path)), str(out)], text=True, timeout=300)
    assert "group of " in text          # RCCL prints its version banner on stdout first
    check_program_output(out.read_bytes(), "the group")


FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3, 5])
def test_group_of_several_ranks_on_one_gpu_over_the_rccl_stand_in(tmp_path, oracle, ranks):
    """The same program with 2, 3 and 5 ranks, all on GPU 0: RCCL itself refuses that, so the
    library is pointed (PCL_RCCL_LIBRARY) at tests/fake_rccl, which carries out the same ten
    entry points through host memory.  What this covers that a world of one cannot: roots and
    non-roots, index ranges, the padded all-gather of ranges of unequal length, the grouped
    all-reduces, all equal to the model of ONE device.  RCCL itself is not exercised."""
    assert os.path.exists(FAKE_RCCL), "make builds tests/fake_rccl/libfake_rccl.so"
    out = tmp_path / "group.bin"
    env = dict(os.environ, PCL_RCCL_LIBRARY=FAKE_RCCL)
    text = subprocess.check_output([str(build(tmp_path)), str(out), str(ranks), "0"], text=True, timeout=300, env=env)
    assert f"group of {ranks}" in text
    check_program_output(out.read_bytes(), f"{ranks} ranks on one GPU")


@pytest.mark.gpu
def test_python_group_matches_the_model(oracle):
    p, g, r = scene_inputs()
    sp, stream, field, sdiag, pairs, pdiag = model(p, g, r)
    r2 = scene_inputs()[2]
    with pa.Group.new(p, g) as grp:
        sp0 = pa.SimParams()
        sp0.delta_t = np.float32(1.0 / 60.0)
        sp0.force_grid_dimensions = (NX, NY, NZ)
        sp0.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
        grp.update_sim_params(sp0)
        grp.enable_step_diagnostics(True)
        for _ in range(3):
            grp.compute()
        mode = pa.EditMode(mode="rotate", relative=False, falloff="linear", falloff_dist=4.0, strength=1.5)
        grp.brush_mouse_down(mode.record((1.0, -2.0, 0.5)))
        grp.brush_mouse_moved(mode.record((1.5, -1.0, 0.5)))
        grp.brush_mouse_up()
        patch = (np.arange(160, dtype=np.float32).reshape(40, 4) / np.float32(16.0) - np.float32(5.0)).astype(np.float32)
        grp.patch_force_grid(100, patch)
        grp.patch_force_grid_indexed(np.array([CELLS - 1, 0, 333], dtype=np.uint32), patch[:3])
        for _ in range(4):
            grp.compute()
        assert_particles_bit_equal(grp.read_particles(), stream, "group, streaming")
        assert_particles_bit_equal(grp.read_particles(1000, 1500), stream[1000:2500], "group, range read")
        assert grp.read_force_grid(grp.size - 1).tobytes() == field.tobytes()
        check_diag(grp.diagnostics(), sdiag)
        # generated on the device: every GPU fills its own range of the global index space
        grp.generate_particles(5000, seed=9, pos_extent=20.0, vel_extent=10.0)
        assert_particles_bit_equal(grp.read_particles(), oracle.synth_particles(5000, 9, 0, 20.0, 10.0), "group, generated")
        with pytest.raises(pa.PclError):
            grp.read_particles(4000, 2000)


@pytest.mark.gpu
def test_comm_attach_one_rank_runs_every_collective(oracle):
    """The one-process-per-GPU form: a unique id, ncclCommInitRank on the handle's
    device, and the collective calls on the handle (here a world of one)."""
    lib = _lib.load()
    p, g, r = scene_inputs()
    sp, stream, field, sdiag, pairs, pdiag = model(p, g, r)
    uid = (C.c_char * _lib.UNIQUE_ID_BYTES)()
    _lib.check(lib.pcl_comm_unique_id(uid))
    with pa.Compute.new(p, np.zeros((1, 4), dtype=np.float32)) as c:
        h = c._h
        assert lib.pcl_comm_rank(h) == -1 and lib.pcl_comm_world(h) == 0
        assert lib.pcl_comm_step(h) == _lib.PCL_ERR_INVALID          # no communicator yet
        _lib.check(lib.pcl_comm_attach(h, uid, 0, 1))
        assert lib.pcl_comm_rank(h) == 0 and lib.pcl_comm_world(h) == 1
        assert lib.pcl_comm_attach(h, uid, 0, 1) == _lib.PCL_ERR_INVALID   # already attached
        gg = np.ascontiguousarray(g)
        _lib.check(lib.pcl_comm_update_force_grid(h, gg.ctypes.data_as(C.c_void_p), CELLS, 0))
        sp0 = pa.SimParams()
        sp0.delta_t = np.float32(1.0 / 60.0)
        sp0.force_grid_dimensions = (NX, NY, NZ)
        sp0.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
        c.update_sim_params(sp0)
        c.enable_step_diagnostics(True)
        for _ in range(3):
            _lib.check(lib.pcl_comm_step(h))
        mode = pa.EditMode(mode="rotate", relative=False, falloff="linear", falloff_dist=4.0, strength=1.5)
        rec = np.ascontiguousarray(mode.record((1.0, -2.0, 0.5)))
        _lib.check(lib.pcl_comm_brush_mouse_down(h, rec.ctypes.data_as(C.c_void_p), 0))
        rec = np.ascontiguousarray(mode.record((1.5, -1.0, 0.5)))
        _lib.check(lib.pcl_comm_brush_mouse_moved(h, rec.ctypes.data_as(C.c_void_p), 0))
        c.brush_mouse_up()
        patch = (np.arange(160, dtype=np.float32).reshape(40, 4) / np.float32(16.0) - np.float32(5.0)).astype(np.float32)
        _lib.check(lib.pcl_comm_patch_force_grid(h, 100, patch.ctypes.data_as(C.c_void_p), 40, 0))
        idx = np.array([CELLS - 1, 0, 333], dtype=np.uint32)
        _lib.check(lib.pcl_comm_patch_force_grid_indexed(h, idx.ctypes.data_as(C.c_void_p), patch.ctypes.data_as(C.c_void_p), 3, 0))
        for _ in range(4):
            _lib.check(lib.pcl_comm_step(h))
        assert_particles_bit_equal(c.read_particles(), stream, "comm, streaming")
        assert c.read_force_grid().tobytes() == field.tobytes()
        d = _lib.Diagnostics()
        _lib.check(lib.pcl_comm_diagnostics_read(h, C.byref(d)))
        check_diag(diag_of(bytes(d)), sdiag)
        c.upload_particles(p[:N_PAIRS])
        c.update_sim_params(sp)
        c.set_pair_mode(pa.PAIRS_ALL)
        for _ in range(3):
            _lib.check(lib.pcl_comm_step(h))
        assert c.last_step_kernel == pa.KERNEL_PAIRS
        assert_particles_bit_equal(c.read_particles(), pairs, "comm, pair steps through the all-gather")
        _lib.check(lib.pcl_comm_diagnostics_read(h, C.byref(d)))
        check_diag(diag_of(bytes(d)), pdiag)
        _lib.check(lib.pcl_comm_detach(h))
        assert lib.pcl_comm_world(h) == 0


@pytest.mark.gpu
def test_the_all_reduce_of_a_step_runs_beside_the_next_step_and_holds_that_steps_record(oracle):
    """With step diagnostics on, pcl_comm_step starts the all-reduce of the step's record on a stream of
    the communicator's own and the handle's stream goes on.  What pcl_comm_diagnostics_device points at
    must, for work enqueued on the handle's stream after that call, be the record of the step just
    issued: here the library's own stream-ordered copy (pcl_patch_force_grid_device) carries the 48 bytes
    into three field cells with no host synchronisation in between, after every one of 26 steps (one
    refresh of the particle order among them), and the oracle says what they must be."""
    lib = _lib.load()
    n, dims = 300_000, (64, 64, 1)
    cells = dims[0] * dims[1]
    rng = np.random.default_rng(31)
    g = np.ones((cells, 4), dtype=np.float32)
    g[:, :3] = rng.uniform(-1, 1, size=(cells, 3)).astype(np.float32)
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.force_grid_dimensions = dims
    sp.masses = (1.0, 0.5, 2.0, 1.5, 0.75)
    state = oracle.synth_particles(n, 77, 0, 20.0, 10.0)
    uid = (C.c_char * _lib.UNIQUE_ID_BYTES)()
    _lib.check(lib.pcl_comm_unique_id(uid))
    with pa.Compute.new(state, g) as c:
        h = c._h
        _lib.check(lib.pcl_comm_attach(h, uid, 0, 1))
        c.update_sim_params(sp)
        c.set_reorder_interval(12)
        c.enable_step_diagnostics(True)
        keep = np.ascontiguousarray(g[cells - 3:])
        for step in range(26):
            _lib.check(lib.pcl_comm_step(h))
            rec = C.c_void_p()
            _lib.check(lib.pcl_comm_diagnostics_device(h, C.byref(rec)))
            _lib.check(lib.pcl_patch_force_grid_device(h, cells - 3, rec, 3))
            got = diag_of(c.read_force_grid(cells - 3, 3).tobytes())
            _lib.check(lib.pcl_patch_force_grid(h, cells - 3, keep.ctypes.data_as(C.c_void_p), 3))
            state = oracle.step(sp, state, g, mode=oracle.MODE_SKIP_INERT)
            want = oracle.diagnostics(sp, state)
            assert got["count"] == want["count"] == n, step
            assert np.array_equal(got["pos_max"], want["pos_max"]) and np.array_equal(got["pos_min"], want["pos_min"]), step
            assert abs(got["kinetic_energy"] - want["kinetic_energy"]) <= 1e-11 * want["kinetic_energy"], step
        assert c.reorder_count >= 2
        assert_particles_bit_equal(c.read_particles(), state, "after 26 steps with the all-reduce beside them")
        _lib.check(lib.pcl_comm_detach(h))


@pytest.mark.gpu
def test_a_long_stroke_through_the_communicator_equals_the_oracles():
    """Twenty cursor records in a row through pcl_comm_brush_mouse_moved with a step after each (the root's
    records queue in a ring of pinned slots, never making the host wait for the stream; twenty wrap the ring
    twice), mirrored call by call on the oracle: same field bytes, same particles."""
    lib = _lib.load()
    n, dims = 20_000, (32, 32, 4)
    cells = dims[0] * dims[1] * dims[2]
    rng = np.random.default_rng(5)
    g = np.ones((cells, 4), dtype=np.float32)
    g[:, :3] = rng.uniform(-1, 1, size=(cells, 3)).astype(np.float32)
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.force_grid_dimensions = dims
    p = pa.random_particles(n, np.random.default_rng(6), pos_max=20.0)
    model = OracleCompute(p, g)
    model.update_sim_params(sp)
    mode = pa.EditMode(mode="shift", relative=True, falloff="linear", falloff_dist=3.0, strength=0.7)
    uid = (C.c_char * _lib.UNIQUE_ID_BYTES)()
    _lib.check(lib.pcl_comm_unique_id(uid))
    with pa.Compute.new(p, g) as c:
        h = c._h
        _lib.check(lib.pcl_comm_attach(h, uid, 0, 1))
        c.update_sim_params(sp)
        rec = np.ascontiguousarray(mode.record((0.0, 0.0, 0.0)))
        _lib.check(lib.pcl_comm_brush_mouse_down(h, rec.ctypes.data_as(C.c_void_p), 0))
        model.brush_mouse_down(rec)
        for k in range(20):
            rec = np.ascontiguousarray(mode.record((0.2 * k, 0.1 * k, -0.05 * k)))
            _lib.check(lib.pcl_comm_brush_mouse_moved(h, rec.ctypes.data_as(C.c_void_p), 0))
            rec.view(np.uint8)[:] = 0                   # the record is the caller's again as soon as the call returns
            model.brush_mouse_moved(np.ascontiguousarray(mode.record((0.2 * k, 0.1 * k, -0.05 * k))))
            _lib.check(lib.pcl_comm_step(h))
            model.compute()
        c.brush_mouse_up()
        model.brush_mouse_up()
        assert c.read_force_grid().tobytes() == model.read_force_grid().tobytes()
        assert not np.array_equal(model.read_force_grid(), g)
        assert_particles_bit_equal(c.read_particles(), model.read_particles(), "a stroke of 20 records through the communicator")
        _lib.check(lib.pcl_comm_detach(h))


@pytest.mark.gpu
def test_indexed_patches_in_a_row_through_the_communicator_equal_the_oracles():
    """Seven indexed patches of different sizes through pcl_comm_patch_force_grid_indexed with steps between
    them; the caller's arrays are overwritten as soon as each call returns (the root's copy of them waits in
    pinned memory for the stream, the host does not wait)."""
    lib = _lib.load()
    n, dims = 50_000, (48, 48, 1)
    cells = dims[0] * dims[1]
    rng = np.random.default_rng(15)
    g = np.ones((cells, 4), dtype=np.float32)
    g[:, :3] = rng.uniform(-1, 1, size=(cells, 3)).astype(np.float32)
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    sp.force_grid_dimensions = dims
    p = pa.random_particles(n, np.random.default_rng(16), pos_max=20.0)
    model = OracleCompute(p, g)
    model.update_sim_params(sp)
    uid = (C.c_char * _lib.UNIQUE_ID_BYTES)()
    _lib.check(lib.pcl_comm_unique_id(uid))
    with pa.Compute.new(p, g) as c:
        h = c._h
        _lib.check(lib.pcl_comm_attach(h, uid, 0, 1))
        c.update_sim_params(sp)
        for count in (5, 900, 1, 2304, 64, 1500, 3):
            idx = rng.choice(cells, size=count, replace=False).astype(np.uint32)
            val = np.ones((count, 4), dtype=np.float32)
            val[:, :3] = rng.uniform(-3, 3, size=(count, 3)).astype(np.float32)
            model.patch_force_grid_indexed(idx.copy(), val.copy())
            _lib.check(lib.pcl_comm_patch_force_grid_indexed(h, idx.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p), count, 0))
            idx[:] = 0
            val[:] = np.float32(777.0)
            for _ in range(2):
                _lib.check(lib.pcl_comm_step(h))
                model.compute()
        assert c.read_force_grid().tobytes() == model.read_force_grid().tobytes()
        assert_particles_bit_equal(c.read_particles(), model.read_particles(), "indexed patches through the communicator")
        _lib.check(lib.pcl_comm_detach(h))
