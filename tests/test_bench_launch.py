"""``python bench.py --gpus N`` started plainly (no torch.distributed.run, no
RANK/WORLD_SIZE in the environment) starts its own ranks.  On the CPU the ranks
come up and stop at bench.py's "needs a GPU" check (there is no CPU fallback);
on a GPU box two ranks share the one GPU over gloo and rank 0 prints exactly one
JSON line; with two or more GPUs the same runs over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _plain_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_plain_start_with_two_gpus_launches_two_ranks_cpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU check of the launcher; the GPU tests below run the whole thing")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--particles-per-gpu", "4096", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=_plain_env())
    # both ranks were started (by torch.distributed.run as a child) and both refused to run without a GPU
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]
    assert r.stdout.strip() == ""


def _run_and_parse(extra, timeout=600, env=None):
    r = subprocess.run([sys.executable, BENCH] + extra, capture_output=True, text=True, timeout=timeout,
                       env=dict(_plain_env(), **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_plain_start_two_ranks_one_gpu_over_gloo():
    out = _run_and_parse(["--gpus", "2", "--collectives", "torch", "--backend", "gloo", "--steps", "6", "--warmup", "2",
                          "--particles-per-gpu", str(1 << 20), "--field", "256", "256", "1",
                          "--diagnostics", "on", "--no-cpu-baseline", "--fused-steps", "0"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["rccl_ranks"] == 0
    assert out["config"]["total_particles"] == 2 << 20
    assert out["final_diagnostics"]["count"] == 2 << 20          # the all-reduce saw both ranks
    assert out["kernel_ms_per_rank"]["min"] > 0


@pytest.mark.gpu
def test_strong_scaling_splits_the_total():
    out = _run_and_parse(["--gpus", "2", "--collectives", "torch", "--backend", "gloo", "--steps", "4", "--warmup", "1", "--scaling", "strong",
                          "--total-particles", str((1 << 21) + 1), "--field", "256", "256", "1",
                          "--no-cpu-baseline", "--fused-steps", "0"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["particles_per_gpu"] == 1 << 20           # rank 0's range [0, N/2)
    assert out["final_diagnostics"]["count"] == (1 << 21) + 1


@pytest.mark.gpu
def test_native_collectives_with_two_ranks_on_one_gpu_over_the_rccl_stand_in():
    """`--collectives native`: field broadcast, brush record and the per-step diagnostics all-reduce are
    pcl_comm_* calls inside the library; tests/fake_rccl stands in for RCCL (which does not admit two
    ranks on one GPU), the gloo group only carries the unique id and the benchmark's own timing exchange."""
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "libfake_rccl.so")
    assert os.path.exists(fake), "make builds tests/fake_rccl/libfake_rccl.so"
    out = _run_and_parse(["--gpus", "2", "--steps", "6", "--warmup", "2",
                          "--particles-per-gpu", str(1 << 20), "--field", "256", "256", "1", "--edits",
                          "--diagnostics", "on", "--no-cpu-baseline", "--fused-steps", "0"],
                         env={"PCL_RCCL_LIBRARY": fake})
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2           # ncclCommCount of the library's communicator
    assert out["config"]["collectives"] == "native"                # the default
    assert out["final_diagnostics"]["count"] == 2 << 20


@pytest.mark.gpu
def test_default_command_with_two_ranks_is_configuration_3_strong_with_its_companions():
    """What the driver runs for N > 1 (`bench.py --gpus N --steps K --warmup W`, nothing else): BASELINE
    configuration 3 split over the ranks through the library's own collectives, the weak-scaling case beside it.
    Rehearsed with two ranks on one GPU over the RCCL stand-in."""
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "libfake_rccl.so")
    out = _run_and_parse(["--gpus", "2", "--steps", "3", "--warmup", "2", "--fused-steps", "0"],
                         env={"PCL_RCCL_LIBRARY": fake}, timeout=900)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["rccl_ranks"] == 2
    assert "configuration 3" in out["config"]["workload"] and "strong" in out["config"]["workload"]
    assert out["config"]["total_particles"] == 1 << 28 and out["config"]["particles_per_gpu"] == 1 << 27
    assert out["final_diagnostics"]["count"] == 1 << 28
    weak = out["also"]["weak_scaling"]
    assert weak["scaling"] == "weak" and weak["config"]["particles_per_gpu"] == 1 << 28 and weak["rccl_ranks"] == 2
    assert "configuration_4" not in out["also"]                    # eight ranks only


@pytest.mark.gpu
def test_companion_cases_that_do_not_finish_cannot_lose_the_headline():
    """The same command with a time limit the companion cases cannot meet: the one JSON line still comes, with the
    headline's values and an error where the companions would be, and the launcher ends with status 0."""
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "libfake_rccl.so")
    out = _run_and_parse(["--gpus", "2", "--steps", "3", "--warmup", "2", "--fused-steps", "0", "--also-timeout", "0.2"],
                         env={"PCL_RCCL_LIBRARY": fake}, timeout=900)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["final_diagnostics"]["count"] == 1 << 28
    assert "did not finish" in out["also"]["error"]


@pytest.mark.gpu
def test_plain_start_two_ranks_over_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank; this box has one")
    out = _run_and_parse(["--gpus", "2", "--steps", "6", "--warmup", "2", "--particles-per-gpu", str(1 << 22),
                          "--diagnostics", "on", "--no-cpu-baseline", "--fused-steps", "0"])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["collectives"] == "native"
    assert out["final_diagnostics"]["count"] == 2 << 22
