#!/usr/bin/env python3
"""Benchmark of the per-frame particle update on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one launch of the update over every resident particle
(``Compute::compute`` of the reference, src/compute.rs:221-227).  Workload:
BASELINE configuration 3, 2^28 particles on a 1024 x 1024 x 1 field, the
256M-particle configuration the HBM target is quoted on, particle-sharded over
the N GPUs (strong scaling: the 2^28 are split); seeded synthetic inputs
generated on the device: pos uniform over the whole bounding cube (worst case
for the field fetch), vel = 10*(u-0.5), ty = i % 5, masses (1, .5, 2, 1.5, .75),
zero force curves (the reference's default, so the streaming kernel runs),
dt = 1/60.  For N > 1 the script runs once per GPU: started by
torch.distributed.run (RANK/WORLD_SIZE in the environment), or, when started
plainly as ``python bench.py --gpus N``, it starts torch.distributed.run itself
as a child process (before anything here touches the GPU) and exits with its
status.  Ranks own disjoint index ranges; the field is created on rank 0 and
replicated with one RCCL broadcast issued INSIDE the library (pcl_comm_*: the
calls a Rust or C host would make; ``--collectives torch`` routes them through
torch.distributed instead), and with diagnostics on the per-step record is
all-reduced.  The process group only carries the communicator's unique id and
this script's own timing exchange.  Rank 0 prints one JSON line.

With N > 1 and no explicit workload flags the same invocation also times, after
the headline, the weak-scaling case (2^28 particles on EVERY GPU) and, at
N = 8, BASELINE configuration 4 (2^30 particles, diagnostics all-reduced every
step); they appear under "also" in the line, never as "value", and cannot lose
it: if they are not through after --also-timeout seconds the line is printed
with the headline alone.
``--scaling weak|strong``, ``--particles-per-gpu``, ``--total-particles`` select
one explicit case instead.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak
BYTES_PER_PARTICLE_STEP = 49     # 24 B state in + 1 B type in + 24 B state out (SoA fp32, u8 type)
MASSES = (1.0, 0.5, 2.0, 1.5, 0.75)
RADIUS = 10.0
DT = 1.0 / 60.0


def swirl_field(nx, ny, nz, radius, seed):
    """(-y, x, 0)/r at the cell centres plus a seeded uniform [-1,1) perturbation."""
    r = np.float32(radius)
    cx = -r + (2 * r) * ((np.arange(nx, dtype=np.float32) + np.float32(0.5)) / np.float32(nx))
    cy = -r + (2 * r) * ((np.arange(ny, dtype=np.float32) + np.float32(0.5)) / np.float32(ny))
    px, py, _ = np.meshgrid(cx, cy, np.zeros(nz, dtype=np.float32), indexing="ij")
    out = np.empty((nx * ny * nz, 4), dtype=np.float32)
    out[:, 0] = (-py / r).ravel()
    out[:, 1] = (px / r).ravel()
    out[:, 2] = 0.0
    rng = np.random.default_rng(seed)
    out[:, :3] += rng.uniform(-1.0, 1.0, size=(nx * ny * nz, 3)).astype(np.float32)
    out[:, 3] = 1.0
    return out


def make_params(pa, dims, masses=MASSES):
    sp = pa.SimParams()
    sp.delta_t = np.float32(DT)
    sp.force_grid_dimensions = dims
    sp.masses = tuple(masses)
    sp.bounding_volume_radius = RADIUS
    sp.max_velocity = 100.0
    return sp


def measured_traffic(n_local, dims, pos_extent):
    """HBM bytes per launch of the step kernels: STORED figures from the committed rocprofv3 PMC passes
    (profiles/traffic.json, which names the pass files), looked up by workload; {} when this workload was
    not profiled.  Keys: hbm_bytes (the plain step kernel), kernels (per launch kind), source.  Not measured
    in this run: counters need rocprofv3 around the process."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            table = json.load(f)
        return table.get(f"{n_local}|{dims[0]}x{dims[1]}x{dims[2]}|{float(pos_extent)}") or {}
    except (OSError, ValueError, KeyError):
        return {}


def cpu_baseline(sp, field, seed, pos_extent, budget_s=12.0):
    """Time the CPU oracle (scalar C restatement of the shader, same pair-skip
    rule, index-range split over host threads) on a bounded sample of the same
    workload: 2^24 particles for about `budget_s` seconds on every core this
    process may use, plus quick points at 2^20 particles and on one thread."""
    from oracle import oracle as O
    nproc = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = nproc
    # OpenMP's own limit (OMP_NUM_THREADS, which the GPU boxes set to their CPU share) and the affinity mask
    threads = max(1, min(O.max_threads(), usable))

    def rate(n, thr, budget, max_steps):
        p = O.synth_particles(n, seed, 0, pos_extent, 10.0)
        p = O.step(sp, p, field, mode=O.MODE_SKIP_INERT, threads=thr)   # warm-up
        steps, t0 = 0, time.perf_counter()
        while True:
            p = O.step(sp, p, field, mode=O.MODE_SKIP_INERT, threads=thr)
            steps += 1
            el = time.perf_counter() - t0
            if el >= budget or steps >= max_steps:
                return n * steps / el, steps, el

    n = 1 << 24
    value, steps, el = rate(n, threads, budget_s, 256)
    small, _, _ = rate(1 << 20, threads, 1.5, 64)
    single, _, _ = rate(1 << 20, 1, 1.5, 8)
    return {
        "value": value,
        "unit": "particle-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{steps} steps of {n} particles (same field, params and distributions), "
                  f"C oracle -O3, {threads} OpenMP threads, {el:.1f} s; 2^20 particles on {threads} threads: "
                  f"{small:.3g} particle-steps/s; 2^20 on one thread: {single:.3g}; "
                  f"nproc {nproc}, usable by this process {usable}, OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS', 'unset')}",
        "nproc": nproc,
        "value_2p20": small,
        "single_thread_value": single,
    }


def self_launch(n_ranks, argv):
    """``python bench.py --gpus N`` without a launcher: start N ranks through
    torch.distributed.run as a child process.  This process has not imported
    torch or touched the GPU; it only waits and passes the child's status on."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--particles-per-gpu", type=int, default=None,
                    help="weak scaling: this many particles on every rank (default 2^28)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="weak: --particles-per-gpu on every rank; strong: --total-particles split over the ranks "
                         "(default: strong, configuration 3; giving --particles-per-gpu alone selects weak)")
    ap.add_argument("--total-particles", type=int, default=None,
                    help="strong scaling: the whole set (default 2^28 = configuration 3; 2^30 over 8 GPUs = configuration 4)")
    ap.add_argument("--field", type=int, nargs=3, default=(1024, 1024, 1), metavar=("NX", "NY", "NZ"))
    ap.add_argument("--pos-extent", type=float, default=2.0 * RADIUS,
                    help="positions are pos_extent*(u-0.5); 20 = whole cube, 2 = the reference's cluster")
    ap.add_argument("--vel-extent", type=float, default=10.0,
                    help="initial velocities are vel_extent*(u-0.5) per axis (10: about 4 field cells per step)")
    ap.add_argument("--diagnostics", choices=("on", "off"), default="off",
                    help="fuse count/kinetic-energy/bounds into the step and all-reduce them per step")
    ap.add_argument("--masses", type=float, nargs=5, default=MASSES, metavar="M",
                    help="masses of the five particle types (the reference's default is 1 for all)")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--reorder-interval", type=int, default=-1,
                    help="steps between refreshes of the field-tile particle order (-1 automatic, 0 never)")
    ap.add_argument("--grid-blocks", type=int, default=0)
    ap.add_argument("--variant", type=lambda s: int(s, 0), default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch-timing", choices=("on", "off"), default="on",
                    help="off: no per-launch event pairs (the roofline then uses the whole region's time per step)")
    ap.add_argument("--clock-warmup-ms", type=float, default=150.0,
                    help="before the warm-up steps, keep the GPU busy for this long with a memory-bound torch kernel "
                         "on a scratch buffer (no simulation step): a short run (20 steps of 2^25 particles are 7 ms) "
                         "otherwise ends before the GPU has reached its clocks.  0: skip")
    ap.add_argument("--fused-steps", type=int, default=96,
                    help="after the timed region, also time this many steps through pcl_step_n (0: skip)")
    ap.add_argument("--edits", action="store_true",
                    help="every step a brush (src/cursor.rs, Shift mode) rewrites the cells within falloff_dist of a "
                         "moving cursor on the device field; rank 0's cursor record is broadcast before the step")
    ap.add_argument("--collectives", choices=("native", "torch"), default="native",
                    help="native: RCCL called inside the library on the handle's own stream (pcl_comm_*), the process "
                         "group only carries the unique id; torch: torch.distributed calls (backend nccl is RCCL)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default=None,
                    help="process-group backend (default: gloo with native collectives, where it is control plane only; "
                         "nccl with torch collectives).  torch collectives over gloo stage everything through the host "
                         "and admit several ranks on one GPU (a rehearsal of the multi-rank path on a one-GPU box)")
    ap.add_argument("--also-timeout", type=float, default=240.0,
                    help="N > 1 without workload flags: seconds the companion cases (weak scaling, configuration 4) may "
                         "take after the headline before the line is printed without them")
    ap.add_argument("--force-collectives", action="store_true",
                    help="use the RCCL broadcast/all-reduce path even with one rank (when started by torch.distributed.run)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    # stdout carries exactly one JSON line: anything libraries print there (RCCL's version banner)
    # goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import particles_amd as pa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus disagree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.backend is None:
        args.backend = "gloo" if args.collectives == "native" else "nccl"
    device_index = local_rank % torch.cuda.device_count()   # one GPU per rank; wraps only in a rehearsal on fewer GPUs
    if args.backend == "nccl" and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: RCCL needs one GPU per rank ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(device_index)
    if world > 1 or (args.force_collectives and "RANK" in os.environ):
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo")
    ctx = {"torch": torch, "dist": dist, "pa": pa, "world": world, "rank": rank, "device_index": device_index,
           "reduce_device": "cuda" if args.backend == "nccl" else "cpu"}

    # Which cases run: an explicit one (any workload flag given), or the headline and its companions.
    explicit = args.scaling is not None or args.particles_per_gpu is not None or args.total_particles is not None
    if args.scaling is None:
        args.scaling = "weak" if (args.particles_per_gpu is not None and args.total_particles is None) else "strong"
    per_gpu = args.particles_per_gpu if args.particles_per_gpu is not None else 1 << 28
    total = args.total_particles if args.total_particles is not None else 1 << 28
    headline = {"scaling": args.scaling, "n_total": total if args.scaling == "strong" else per_gpu * world,
                "diag_on": args.diagnostics == "on", "with_cpu_baseline": world == 1 and not args.no_cpu_baseline}
    out = run_case(args, ctx, **headline)
    if world > 1 and not explicit and not args.edits:
        # The headline is measured; the companion cases must not be able to lose it.  If they are not through after
        # --also-timeout seconds (a rank stuck in a collective), rank 0 prints the line without them and every rank leaves.
        def give_up():
            if rank == 0:
                out["also"] = {"error": f"the companion cases did not finish within {args.also_timeout:.0f} s; "
                                        "the headline values of this line were complete before they started"}
                os.write(json_fd, (json.dumps(out) + "\n").encode())
            os._exit(0)
        watchdog = threading.Timer(args.also_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        also = {"weak_scaling": run_case(args, ctx, scaling="weak", n_total=(1 << 28) * world,
                                         diag_on=args.diagnostics == "on", with_cpu_baseline=False)}
        if world == 8:
            also["configuration_4"] = run_case(args, ctx, scaling="strong", n_total=1 << 30, diag_on=True, with_cpu_baseline=False)
        watchdog.cancel()
        if rank == 0:
            out["also"] = {k: {f: v[f] for f in ("value", "unit", "ms_per_step", "scaling", "rccl_ranks",
                                                 "kernel_ms_per_rank", "config", "roofline", "final_diagnostics")}
                           for k, v in also.items()}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


def run_case(args, ctx, *, scaling, n_total, diag_on, with_cpu_baseline):
    """One timed case: builds the sharded scene, warms up, times args.steps steps between fences and returns the
    result record on rank 0 (None elsewhere)."""
    torch, dist, pa = ctx["torch"], ctx["dist"], ctx["pa"]
    world, rank, device_index, reduce_device = ctx["world"], ctx["rank"], ctx["device_index"], ctx["reduce_device"]
    dims = tuple(args.field)
    n_cells = dims[0] * dims[1] * dims[2]
    sp = make_params(pa, dims, args.masses)

    # Field: built on rank 0, replicated by ShardedCompute with one broadcast over RCCL;
    # every rank fills its own index range with the seeded scene on the device.
    field_host = swirl_field(*dims, RADIUS, args.seed) if rank == 0 else None
    # one process and nobody to talk to: no communicator is formed at all (the field goes straight to the handle)
    collectives = args.collectives if dist.is_initialized() else "torch"
    try:
        sc = pa.ShardedCompute.new(None, field_host, n_total=n_total, seed=args.seed,
                                   pos_extent=args.pos_extent, vel_extent=args.vel_extent, device=device_index,
                                   force_collectives=args.force_collectives, collectives=collectives, n_cells=n_cells)
    except pa.PclError as e:
        # no fallback to another collective path: what the library's RCCL said is the result
        raise SystemExit(f"rank {rank}: setting up the sharded scene with --collectives {collectives} failed: {e}")
    n_local = sc.num_local_particles        # this rank's index range [g*N/G, (g+1)*N/G)
    c = sc.local
    # which physical device this rank runs on, and how many ranks the library's communicator counts
    print(f"bench.py rank {rank}/{world}: device {device_index} pci {c.pci_bus_id} particles {n_local} "
          f"library communicator ranks {c.comm_count}", file=sys.stderr, flush=True)
    c.set_tuning(args.grid_blocks, args.variant)
    c.set_reorder_interval(args.reorder_interval)
    sc.update_sim_params(sp)
    sc.enable_step_diagnostics(diag_on)

    brush = None
    if args.edits:
        # the brush runs on every replica of the device field; rank 0 owns the cursor, like the
        # reference's App (src/main.rs:287-316), and only its 64-byte record travels
        brush = pa.DeviceBrush(sc, pa.EditMode(mode="shift", falloff_dist=1.0, strength=1.0))
        brush.mouse_down((0.0, 0.0, 0.0))
    frame = [0]

    def one_step():
        if args.edits:
            a = 0.05 * frame[0]
            brush.mouse_moved((0.3 * np.cos(a), 0.3 * np.sin(a), 0.0))
            frame[0] += 1
        sc.compute()          # one update of the local range (+ the diagnostics all-reduce when enabled)

    def fence():
        sc.wait()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # The GPU takes some tens of milliseconds of continuous work to reach its clocks; the warm-up steps of a small
    # range are over long before that.  Untimed, touches no simulation state, advances nothing.
    if args.clock_warmup_ms > 0:
        scratch = torch.zeros(1 << 26, dtype=torch.float32, device=f"cuda:{device_index}")
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < args.clock_warmup_ms:
            for _ in range(16):
                scratch.add_(1.0)
            torch.cuda.synchronize()
        del scratch
    for _ in range(args.warmup):
        one_step()
    fence()
    reorders0 = c.reorder_count
    c.step_timing_begin(args.steps if args.launch_timing == "on" else 0)   # one HIP event pair per launch, on the kernel's stream, no syncs
    t0 = time.perf_counter()
    c.timer_start()
    for _ in range(args.steps):
        one_step()
    region_ms = c.timer_stop() / max(1, args.steps)   # HIP events around the whole timed region
    # This rank's clock stops when its own GPU is idle again; the closing barrier comes after it, and the
    # MAX over ranks taken below is the job's time (every rank started from the same barrier).  A clock
    # stopped behind the barrier would charge the steps a host round trip between N processes, which at
    # 8 ranks is a tenth of a 20-step window of 2^25-particle steps.
    sc.wait()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    launch_ms, launch_kind = c.step_timing_read()
    reorders_timed = c.reorder_count - reorders0
    # the dominant kernel: the plain step kernel (streaming or pair), without the launches that
    # also count / scatter for the order refresh
    plain = launch_ms[(launch_kind == 1) | (launch_kind == 2)]
    kernel_ms = float(plain.mean()) if plain.size else region_ms
    by_kind = {name: {"launches": int((launch_kind == k).sum()),
                      "avg_ms": float(launch_ms[launch_kind == k].mean()) if (launch_kind == k).any() else None}
               for k, name in ((1, "step_stream_kernel"), (2, "step_pairs_kernel"),
                               (3, "step_stream_kernel+count"), (4, "step_scatter_kernel"))}

    kernel_ms_local = kernel_ms
    per_rank_kernel_ms = [kernel_ms]
    if dist.is_initialized():
        t = torch.tensor([elapsed, kernel_ms, region_ms], dtype=torch.float64, device=reduce_device)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_kernel_ms = [float(e[1]) for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms, region_ms = float(t[0]), float(t[1]), float(t[2])
    if collectives == "native":
        rccl_ranks = c.comm_count                            # ncclCommCount of the library's own communicator
    else:
        rccl_ranks = dist.get_world_size() if (dist.is_initialized() and dist.get_backend() == "nccl") else 0

    # Beside the headline (one launch per step): the same update through pcl_step_n, which fuses
    # runs of steps into single launches for callers that do not need the intermediate states.
    fused = None
    if args.fused_steps > 0 and not args.edits:
        fence()
        c.timer_start()
        sc.compute_n(args.fused_steps)
        fused_ms = c.timer_stop() / args.fused_steps
        fence()
        if dist.is_initialized():
            t = torch.tensor([fused_ms], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fused_ms = float(t[0])
        fused = {"steps": args.fused_steps, "ms_per_step": fused_ms,
                 "value": n_total / (fused_ms * 1e-3), "unit": "particle-steps/s",
                 "note": "NOT the headline: pcl_step_n keeps the particles in registers over runs of up to 16 "
                         "steps (49 B of HBM traffic per particle per launch instead of per step); results are "
                         "bit-identical to single steps, intermediate states are not materialised"}

    diag = sc.diagnostics()      # collective: global count / kinetic energy / bounds
    geometry = c.launch_geometry()
    geometry["resident_blocks_per_cu"] = c.launch_residency()
    kernel = c.last_step_kernel

    out = None
    if rank == 0:
        total = n_total
        value = total * args.steps / elapsed
        # rank 0's own launches over rank 0's own range
        achieved = n_local * BYTES_PER_PARTICLE_STEP / (kernel_ms_local * 1e-3) / 1e9
        whole = n_local * BYTES_PER_PARTICLE_STEP / (region_ms * 1e-3) / 1e9
        if scaling == "strong" and total == 1 << 28 and dims == (1024, 1024, 1) and world > 1:
            which = f"BASELINE configuration 3 (256M particles, 1024x1024 field, particle-sharded over {world} GPUs, strong scaling)"
        elif scaling == "strong" and total == 1 << 30 and dims == (1024, 1024, 1) and diag_on:
            which = "BASELINE configuration 4 (1B particles, diagnostics all-reduce per step)"
        elif world == 1 and total == 1 << 28 and dims == (1024, 1024, 1):
            which = "BASELINE configuration 3 at one GPU (256M particles, 1024x1024 field)"
        elif world == 1 and total == 1 << 24 and dims == (256, 256, 1):
            which = "BASELINE configuration 2 (16M particles, 256x256 field)"
        elif total == 1 << 26 and args.edits:
            which = "BASELINE configuration 5 (64M particles, per-frame brush edits)"
        elif scaling == "weak" and n_local == 1 << 28 and dims == (1024, 1024, 1):
            which = "configuration 3's one-GPU load on every GPU (weak scaling: 2^28 particles per GPU)"
        else:
            which = "custom"
        # the launches that also refresh the particle order move more bytes per particle:
        # counting step 49 B (same traffic, extra LDS histogram), scatter step 49 B + 4 B index read + 4 B index write + 1 B type write
        traffic_table = measured_traffic(n_local, dims, args.pos_extent) if args.vel_extent == 10.0 else {}
        kernel_rooflines = []
        for k, name, nbytes in ((1, "step_stream_kernel", 49), (2, "step_pairs_kernel", 49),
                                (3, "step_stream_kernel+count", 49), (4, "step_scatter_kernel", 58)):
            ms = by_kind[name]["avg_ms"]
            if ms:
                a = n_local * nbytes / (ms * 1e-3) / 1e9
                kernel_rooflines.append({"kernel": name, "launches": by_kind[name]["launches"], "avg_ms": ms,
                                         "bytes_per_particle": nbytes, "achieved": a, "frac": a / HBM_PEAK_GBS,
                                         "traffic": (traffic_table.get("kernels") or {}).get(name)})
        traffic, traffic_source = traffic_table.get("hbm_bytes"), traffic_table.get("source")
        out = {
            "metric": "particle-steps/sec",
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_warmup_ms": args.clock_warmup_ms,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "rccl_ranks": rccl_ranks,
            "kernel_ms_per_rank": {"min": min(per_rank_kernel_ms), "max": max(per_rank_kernel_ms)},
            "config": {
                "workload": f"{which}: {n_local} particles on rank 0's GPU ({total} total over {world}), "
                            f"{dims[0]}x{dims[1]}x{dims[2]} static vector field, fp32 SoA, pair term inert",
                "particles_per_gpu": n_local,
                "total_particles": total,
                "field_dims": list(dims),
                "pos_extent": args.pos_extent,
                "vel_extent": args.vel_extent,
                "diagnostics_per_step": diag_on,
                "field_edits_per_step": bool(args.edits),
                "reorder_interval": args.reorder_interval,
                "reorder_passes_in_timed_region": reorders_timed,
                "parallelism": f"particle-sharded x{world}, field replicated by "
                               + ("no collective (one process)" if not dist.is_initialized() else
                                  "RCCL broadcast inside the library (pcl_comm_*)" if collectives == "native" else
                                  "RCCL broadcast through torch.distributed" if args.backend == "nccl" else
                                  "gloo broadcast (host-staged rehearsal)"),
                "collectives": collectives if dist.is_initialized() else "none",
                "kernel": {1: "step_stream_kernel", 2: "step_pairs_kernel"}.get(kernel, "none"),
                "launch": geometry,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": (f"stored figure from rocprofv3 PMC passes, {traffic_source or 'profiles/traffic.json'}; "
                                   "not measured in this run") if traffic else None,
                "whole_step_frac": whole / HBM_PEAK_GBS,
                "bytes_per_particle_step": BYTES_PER_PARTICLE_STEP,
                "kernel": {1: "step_stream_kernel", 2: "step_pairs_kernel"}.get(kernel, "none"),
                "kernel_ms": kernel_ms_local,
                "kernels": kernel_rooflines,
                "launches_by_kind": by_kind,
                "whole_region": {"ms_per_step": region_ms, "achieved": whole, "frac": whole / HBM_PEAK_GBS},
                "note": "achieved/frac: algorithmic bytes per launch / average launch duration of the dominant kernel "
                        "(HIP event pair per launch on the kernel's stream, rank 0). whole_step_frac / whole_region: "
                        "49 B per particle over the HIP-event time of the whole timed region / steps, i.e. including "
                        "the counting and scatter launches that refresh the particle order. kernels: every launch "
                        "kind of the timed region with its own byte count and, where profiled, its counter-derived "
                        "HBM bytes per launch (stored figures)",
            },
            "final_diagnostics": {"count": diag["count"], "kinetic_energy": diag["kinetic_energy"]},
        }
        if fused is not None:
            out["fused_stepping"] = fused
        if with_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sp, field_host, args.seed, args.pos_extent)
            out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    sc.close()
    return out


if __name__ == "__main__":
    main()
