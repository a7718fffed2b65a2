"""Particle-sharded update over the GPUs of one node (one process per GPU).

With the pair term inert every particle depends only on its own state, the
replicated field and the replicated 912-byte parameter block, so the particle
set is split into contiguous index ranges, one per rank, with no data-path
collective.  With the pair term active the loop of src/compute.wgsl:115-144
runs over ALL particles: every frame the ranks all-gather 16 bytes per particle
(position and type) and each updates its range against the gathered records in
global index order, which reproduces the single-device sums bit for bit.  What
travels, over ``torch.distributed`` (backend "nccl" is RCCL over xGMI on ROCm):

* the field — built on rank 0, replicated with one broadcast at construction,
  after ``update_force_grid`` and (dirty range only) after ``patch_force_grid``
  (the reference re-uploads the whole field every frame: src/main.rs:371-380);
* with the pair term active, per step one all-gather of 16 B x N;
* per-step diagnostics — two tiny all-reduces (sum: count, kinetic energy;
  max: max xyz, -min xyz), issued asynchronously so they overlap the next step.

Under a host-memory process group (backend "gloo") the same class drives the
GPU backend with every collective staged through the host: slower, but it
needs no RCCL, and several ranks may then share one GPU.

``collectives="native"`` moves every collective into the library
(``pcl_comm_*`` of the C ABI: RCCL calls on the handle's own stream): the
process group is then only used once, to carry rank 0's 128-byte unique id to
the other ranks, and this class becomes a thin caller.

The class mirrors ``Compute`` (src/compute.rs:11-228); ``compute_cls`` is the
single-GPU backend each rank instantiates for its range.
"""
import numpy as np

from .compute import Compute
from .cursor import BRUSH_INPUT_DTYPE
from .particle import PARTICLE_DTYPE
from .sim_params import SimParams


def shard_range(n_total, rank, world_size):
    """Contiguous index range [first, last) owned by ``rank``: g*N/G .. (g+1)*N/G."""
    first = (rank * n_total) // world_size
    last = ((rank + 1) * n_total) // world_size
    return first, last


class ShardedCompute:
    def __init__(self, particles, force_grid, *, process_group=None, device=None, compute_cls=Compute,
                 n_total=None, seed=None, pos_extent=2.0, vel_extent=10.0, force_collectives=False,
                 collectives="torch", n_cells=None):
        """Every rank passes the full ``particles`` array and keeps its own
        range, or passes ``particles=None`` with ``n_total`` and ``seed`` to
        fill its range with the seeded synthetic scene on the device.
        ``force_grid`` is taken from rank 0 (other ranks may pass ``None``).
        ``force_collectives`` runs the broadcast / all-reduce code path even
        in a one-rank group (to exercise RCCL on a single GPU).
        ``collectives``: "torch" (``torch.distributed`` calls) or "native"
        (RCCL inside the library; GPU only).  ``n_cells``: the field's cell
        count when every rank knows it, which saves the header round trip.
        """
        import torch
        import torch.distributed as dist

        self._torch, self._dist = torch, dist
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self.on_gpu = backend == "nccl" or (backend == "none" and torch.cuda.is_available())
        if device is None:
            device = torch.cuda.current_device() if self.on_gpu else 0
        self.device = int(device)
        self._comm_device = torch.device("cuda", self.device) if self.on_gpu else torch.device("cpu")
        # collectives are skipped only when there is nobody to talk to
        self._collectives = dist.is_initialized() and (self.world_size > 1 or force_collectives)

        if particles is not None:
            particles = np.ascontiguousarray(particles).reshape(-1)
            self.n_total = particles.shape[0]
        else:
            if n_total is None or seed is None:
                raise ValueError("pass either particles, or n_total and seed")
            self.n_total = int(n_total)
        self.first, self.last = shard_range(self.n_total, self.rank, self.world_size)

        if collectives not in ("torch", "native"):
            raise ValueError("collectives must be 'torch' or 'native'")
        self._native = collectives == "native"
        local = particles[self.first:self.last] if particles is not None else np.zeros(0, dtype=PARTICLE_DTYPE)
        if self._native:
            if not self.on_gpu and dist.is_initialized() and dist.get_backend(process_group) != "gloo":
                raise ValueError("native collectives need the GPU backend")
            self.on_gpu = True
            self._collectives = True
            # the handle keeps its own stream: the library's RCCL calls are ordered on it
            self.local = compute_cls.new(local, np.zeros((1, 4), dtype=np.float32), device=self.device)
            uid = [Compute.comm_unique_id() if self.rank == 0 else None]
            if dist.is_initialized():
                dist.broadcast_object_list(uid, src=0, group=self.group)
            self.local.comm_attach(uid[0], self.rank, self.world_size)
            self.update_force_grid(force_grid, n_cells=n_cells)
            grid = None
        else:
            grid = self._replicate_field(force_grid, 0, n_cells)
        if self._native:
            pass
        elif self.on_gpu and self._collectives:
            # the replicated field is already on this device: adopt it without a host round trip
            self.local = compute_cls.new(local, np.zeros((1, 4), dtype=np.float32), device=self.device)
            self.local.set_stream(torch.cuda.current_stream().cuda_stream)
            self.local.update_force_grid_device(grid.data_ptr(), grid.shape[0])
        else:
            self.local = compute_cls.new(local, grid, device=self.device)
            if self.on_gpu:
                self.local.set_stream(torch.cuda.current_stream().cuda_stream)
        if particles is None:
            self.local.generate_particles(self.last - self.first, seed, first_global_index=self.first,
                                          pos_extent=pos_extent, vel_extent=vel_extent)
        self._brush_ring = None
        self._brush_slot = 0
        self._params = SimParams()     # the handle starts with SimParams::new(): all curves zero
        self._pair_active = False
        self._pair_bufs = None
        self._pair_stage = None
        self._diag_enabled = False
        # Ring of diagnostics records: an all-reduce in flight owns its record
        # until waited for, so consecutive steps use different ones.
        self._diag_ring = None
        self._diag_slot = 0
        self._pending = []

    @classmethod
    def new(cls, particles, force_grid, **kw):
        return cls(particles, force_grid, **kw)

    # -- field replication ------------------------------------------------
    def _agree(self, value):
        """Rank 0's (small, picklable) value on every rank: a host-synchronous
        round trip, used only where the other ranks cannot know a size."""
        box = [value if self.rank == 0 else None]
        if self._dist.is_initialized() and self.world_size > 1:
            self._dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    def _replicate_field(self, cells, first_cell, count):
        """Broadcast rank 0's cells [first_cell, first_cell+count) and return
        them on every rank.  ``count=None``: the other ranks learn the size
        from a header that rank 0 sends first (a host-synchronous round trip);
        when every rank passes ``count`` the payload goes out at once."""
        torch, dist = self._torch, self._dist
        if not self._collectives:
            return np.ascontiguousarray(cells, dtype=np.float32)
        if self.rank == 0:
            cells = np.ascontiguousarray(cells, dtype=np.float32)
        if count is None:
            first_cell, n = self._agree((int(first_cell), int(cells.shape[0])) if self.rank == 0 else None)
        else:
            n = int(count)
            if self.rank == 0 and cells.shape[0] != n:
                raise ValueError("count does not match the cells rank 0 passed")
        t = torch.empty((n, 4), dtype=torch.float32, device=self._comm_device)
        if self.rank == 0:
            t.copy_(torch.from_numpy(cells))
        dist.broadcast(t, src=0, group=self.group)
        self._last_field_tensor = t          # keeps the device buffer alive for *_device uploads
        self._last_first_cell = first_cell
        return t.cpu().numpy() if not self.on_gpu else t

    def update_force_grid(self, force_grid, n_cells=None):
        """Whole-field replace (src/compute.rs:198-204) from rank 0's array."""
        if self._native:
            if n_cells is None:
                n_cells = self._agree(int(np.asarray(force_grid).shape[0]) if self.rank == 0 else None)
            self.local.comm_update_force_grid(force_grid if self.rank == 0 else None, n_cells)
            return
        g = self._replicate_field(force_grid, 0, n_cells)
        if self.on_gpu and self._collectives:
            self.local.update_force_grid_device(g.data_ptr(), g.shape[0])
        else:
            self.local.update_force_grid(g)

    def patch_force_grid(self, first_cell, cells, count=None):
        """Brush edit: only cells [first_cell, first_cell+len) travel.  Ranks
        other than 0 may pass ``cells=None``; when every rank passes
        ``first_cell`` and ``count`` no header precedes the payload."""
        if self._native:
            if count is None:
                first_cell, count = self._agree((int(first_cell), int(np.asarray(cells).shape[0])) if self.rank == 0 else None)
            self.local.comm_patch_force_grid(first_cell, cells if self.rank == 0 else None, count)
            return
        g = self._replicate_field(cells, first_cell, count)
        first_cell = first_cell if not self._collectives else self._last_first_cell
        if self.on_gpu and self._collectives:
            self.local.patch_force_grid_device(first_cell, g.data_ptr(), g.shape[0])
        else:
            self.local.patch_force_grid(first_cell, g)

    def patch_force_grid_indexed(self, cell_indices, cells, count=None):
        """Brush edit of a list of cells (src/cursor.rs:84-101): rank 0's
        ``(cell_indices, cells)`` lists are broadcast and applied everywhere.
        ``count`` passed by every rank saves the header round trip."""
        torch, dist = self._torch, self._dist
        if not self._collectives:
            self.local.patch_force_grid_indexed(cell_indices, cells)
            return
        if self.rank == 0:
            cell_indices = np.ascontiguousarray(cell_indices, dtype=np.uint32).reshape(-1)
            cells = np.ascontiguousarray(cells, dtype=np.float32).reshape(-1, 4)
        n = int(count) if count is not None else self._agree(int(cell_indices.shape[0]) if self.rank == 0 else None)
        if n == 0:
            return
        if self._native:
            self.local.comm_patch_force_grid_indexed(cell_indices if self.rank == 0 else None,
                                                     cells if self.rank == 0 else None, n)
            return
        # one message: n cells of 16 bytes followed by n 4-byte indices
        buf = torch.empty(n * 5, dtype=torch.int32, device=self._comm_device)
        if self.rank == 0:
            packed = np.concatenate([cells.view(np.int32).reshape(-1), cell_indices.view(np.int32)])
            buf.copy_(torch.from_numpy(packed))
        dist.broadcast(buf, src=0, group=self.group)
        if self.on_gpu:
            self._last_field_tensor = buf
            self.local.patch_force_grid_indexed_device(buf[n * 4:].data_ptr(), buf.data_ptr(), n)
        else:
            host = buf.numpy()
            self.local.patch_force_grid_indexed(host[n * 4:].view(np.uint32), host[:n * 4].view(np.float32).reshape(n, 4))

    # -- field brush ------------------------------------------------------
    def _replicate_brush_record(self, record):
        """Rank 0's 64-byte cursor record on every rank: a device tensor on
        the GPU path (consumed by the kernels without a host round trip), a
        host record otherwise."""
        torch, dist = self._torch, self._dist
        if not self._collectives:
            return record
        if self._brush_ring is None:
            self._brush_ring = [torch.zeros(64, dtype=torch.uint8, device=self._comm_device) for _ in range(self._RING)]
        self._brush_slot = (self._brush_slot + 1) % self._RING
        t = self._brush_ring[self._brush_slot]
        if self.rank == 0:
            t.copy_(torch.from_numpy(np.ascontiguousarray(record).view(np.uint8).reshape(64).copy()))
        dist.broadcast(t, src=0, group=self.group)
        return t if self.on_gpu else t.numpy().view(BRUSH_INPUT_DTYPE)

    def brush_mouse_down(self, record):
        """Cursor::mouse_down (src/cursor.rs:105-116) on every replica of the
        field; ``record`` is taken from rank 0 (others may pass ``None``)."""
        if self._native:
            self.local.comm_brush_mouse_down(record if self.rank == 0 else None)
            return
        r = self._replicate_brush_record(record)
        if self.on_gpu and self._collectives:
            self.local.brush_mouse_down_device(r.data_ptr())
        else:
            self.local.brush_mouse_down(r)

    def brush_mouse_moved(self, record):
        """Cursor::mouse_moved (src/cursor.rs:79-103): every rank rewrites the
        same cells of its replica, so only the 64-byte record travels."""
        if self._native:
            self.local.comm_brush_mouse_moved(record if self.rank == 0 else None)
            return
        r = self._replicate_brush_record(record)
        if self.on_gpu and self._collectives:
            self.local.brush_mouse_moved_device(r.data_ptr())
        else:
            self.local.brush_mouse_moved(r)

    def brush_mouse_up(self):
        self.local.brush_mouse_up()

    def read_force_grid(self, first_cell=0, count=None):
        return self.local.read_force_grid(first_cell, count)

    # -- parameters and stepping -----------------------------------------
    def update_sim_params(self, sim_params):
        """Replicated by value on every rank (src/compute.rs:206-219)."""
        self.local.update_sim_params(sim_params)
        if not isinstance(sim_params, SimParams):
            wrapped = SimParams()
            wrapped.raw = np.asarray(sim_params)
            sim_params = wrapped
        self._params = sim_params.copy()
        self._pair_active = not sim_params.pair_term_inert()

    def _pair_capacity(self):
        """Records per rank in the all-gather: the largest range, ceil(N / G)."""
        return -(-self.n_total // self.world_size)

    def _pair_step(self):
        """One update with the pair term active across ranks: pack (x, y, z,
        type) of the local range, all-gather, update against all records."""
        torch, dist = self._torch, self._dist
        cap = self._pair_capacity()
        if self._pair_bufs is None or self._pair_bufs[0].shape[0] != cap:
            self._pair_bufs = (torch.empty((cap, 4), dtype=torch.float32, device=self._comm_device),
                               torch.empty((cap * self.world_size, 4), dtype=torch.float32, device=self._comm_device))
        send, recv = self._pair_bufs
        if self.on_gpu:
            self.local.pack_pair_sources_device(send.data_ptr(), cap)
            dist.all_gather_into_tensor(recv, send, group=self.group)
            self.local.step_pairs_gathered(recv.data_ptr(), recv.shape[0])
        elif hasattr(self.local, "pack_pair_sources"):
            # host-resident backend (the tests' CPU stand-in)
            send.copy_(torch.from_numpy(self.local.pack_pair_sources(cap)))
            dist.all_gather_into_tensor(recv, send, group=self.group)
            self.local.step_pairs_gathered(recv.numpy(), self.rank * cap)
        else:
            # GPU backend under a host-memory process group (gloo): the records are staged
            # through the host on their way to the collective and back
            dev = torch.device("cuda", self.device)
            if self._pair_stage is None or self._pair_stage[0].shape[0] != cap:
                self._pair_stage = (torch.empty((cap, 4), dtype=torch.float32, device=dev),
                                    torch.empty((cap * self.world_size, 4), dtype=torch.float32, device=dev))
            send_dev, recv_dev = self._pair_stage
            self.local.synchronize()             # the previous step may still read recv_dev
            self.local.pack_pair_sources_device(send_dev.data_ptr(), cap)
            self.local.synchronize()
            send.copy_(send_dev)
            dist.all_gather_into_tensor(recv, send, group=self.group)
            recv_dev.copy_(recv)
            torch.cuda.synchronize(dev)
            self.local.step_pairs_gathered(recv_dev.data_ptr(), recv_dev.shape[0])

    def _advance(self, dt=None):
        if self._native:
            if dt is not None:      # step(dt) = set delta_t, then one update
                self._params.delta_t = dt
                self.local.update_sim_params(self._params)
            self.local.comm_step()  # all-gather (pair term) and diagnostics all-reduce happen inside
        elif self._pair_active and self._collectives and self.n_total > 0:
            if dt is not None:      # step(dt) = set delta_t, then one update
                self._params.delta_t = dt
                self.local.update_sim_params(self._params)
            self._pair_step()
        elif dt is None:
            self.local.compute()
        else:
            self.local.step(dt)

    def enable_step_diagnostics(self, enable=True):
        self._diag_enabled = bool(enable)
        self.local.enable_step_diagnostics(enable)

    def compute(self):
        """One update of the local range (src/compute.rs:221-227); with step
        diagnostics enabled, also starts their all-reduce."""
        self._advance()
        if self._diag_enabled:
            self._start_diag_allreduce()

    def compute_n(self, n_steps):
        """``n_steps`` updates of the local range (fused launches where the
        path allows); step diagnostics describe the final state."""
        if self._native and self._pair_active:
            for _ in range(int(n_steps)):
                self.local.comm_step()
            return
        if self._pair_active and self._collectives and self.n_total > 0:
            for _ in range(int(n_steps)):
                self._pair_step()
        else:
            self.local.step_n(n_steps)
        if self._diag_enabled:
            self._start_diag_allreduce()

    def step(self, dt=None):
        self._advance(dt)
        if self._diag_enabled:
            self._start_diag_allreduce()

    _RING = 8

    @property
    def _diag_buf(self):
        if self._diag_ring is None:
            self._diag_ring = [self._torch.zeros(48, dtype=self._torch.uint8, device=self._comm_device)
                               for _ in range(self._RING)]
        return self._diag_ring[self._diag_slot]

    def _fill_diag_buf(self):
        torch = self._torch
        self._diag_slot = (self._diag_slot + 1) % self._RING
        if self.on_gpu:
            self.local.diagnostics_device(self._diag_buf.data_ptr())
        else:
            d = self.local.diagnostics()
            host = np.zeros(48, dtype=np.uint8)
            host[:16].view(np.float64)[:] = (float(d["count"]), d["kinetic_energy"])
            host[16:28].view(np.float32)[:] = d["pos_max"]
            host[28:40].view(np.float32)[:] = -np.asarray(d["pos_min"], dtype=np.float32)
            self._diag_buf.copy_(torch.from_numpy(host))

    def _start_diag_allreduce(self):
        torch, dist = self._torch, self._dist
        if self._native:
            return      # pcl_comm_step starts it on the handle's stream; diagnostics() reads it
        self._fill_diag_buf()
        if self._collectives:
            sums = self._diag_buf[:16].view(torch.float64)     # count, kinetic energy
            maxs = self._diag_buf[16:40].view(torch.float32)   # max xyz, -min xyz
            self._pending.append(dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._pending.append(dist.all_reduce(maxs, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
            while len(self._pending) > self._RING:   # two works per step: at most RING/2 steps in flight
                self._pending.pop(0).wait()

    def wait(self):
        """Complete outstanding diagnostics all-reduces and the local stream."""
        while self._pending:
            self._pending.pop(0).wait()
        self.local.synchronize()

    def diagnostics(self):
        """Global diagnostics of the latest state (identical on every rank)."""
        if self._native:
            return self.local.comm_diagnostics()
        if not self._diag_enabled:
            self._start_diag_allreduce()
        self.wait()
        if self.on_gpu:
            self._torch.cuda.current_stream().synchronize()
        host = self._diag_buf.cpu().numpy()
        count, ke = host[:16].view(np.float64)
        return {"count": int(count), "kinetic_energy": float(ke),
                "pos_max": host[16:28].view(np.float32).copy(),
                "pos_min": -host[28:40].view(np.float32)}

    # -- data access -------------------------------------------------------
    @property
    def num_particles(self):
        """Global particle count (src/compute.rs:19)."""
        return self.n_total

    @property
    def num_local_particles(self):
        return self.last - self.first

    def upload_particles(self, particles):
        """Replace the whole set (src/compute.rs:182-196); every rank passes
        the full array and keeps its range."""
        particles = np.ascontiguousarray(particles).reshape(-1)
        self.n_total = particles.shape[0]
        self.first, self.last = shard_range(self.n_total, self.rank, self.world_size)
        self.local.upload_particles(particles[self.first:self.last])

    def read_local(self):
        """This rank's records, global indices [self.first, self.last)."""
        return self.local.read_particles()

    def gather_particles(self):
        """The full set in caller order on every rank (small sets / tests)."""
        mine = self.read_local()
        if not self._collectives:
            return mine
        parts = [None] * self.world_size
        self._dist.all_gather_object(parts, mine, group=self.group)
        return np.concatenate(parts)

    def close(self):
        self.wait()
        self.local.close()
