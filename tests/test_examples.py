"""The programs under examples/: the C one must build as strict C11 against the
library (CPU check); on a GPU both must print the oracle's default scene."""
import os
import subprocess
import sys

import numpy as np
import pytest

import particles_amd as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_c_example(tmp_path):
    exe = tmp_path / "default_scene"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "default_scene.c"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "particles_amd"), "-lparticles_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "particles_amd"), "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def test_c_example_builds_as_strict_c11(tmp_path):
    assert build_c_example(tmp_path).exists()


def parse(text):
    lines = text.strip().splitlines()
    head = lines[0].split()
    rows = [np.array([float(v) for v in (l.split()[3:6] + l.split()[7:10])], dtype=np.float32) for l in lines[1:3]]
    return int(float(head[3])), float(head[5]), rows


@pytest.mark.gpu
def test_examples_print_the_oracles_default_scene(tmp_path, oracle):
    sp = pa.SimParams()
    sp.delta_t = np.float32(1.0 / 60.0)
    p = pa.lattice_particles((5.0, 2.0, 2.0), 10, 10, 10)
    g = sp.new_force_grid_centered().get_force_vectors()
    want = oracle.run(sp, p, g, 16, mode=oracle.MODE_ALWAYS_PAIRS)
    want_ke = oracle.diagnostics(sp, want)["kinetic_energy"]
    outputs = [subprocess.check_output([str(build_c_example(tmp_path)), "16"], text=True),
               subprocess.check_output([sys.executable, os.path.join(ROOT, "examples", "default_scene.py"), "16"], text=True)]
    for text in outputs:
        count, ke, rows = parse(text)
        assert count == 1000 and abs(ke - want_ke) <= 1e-8 * want_ke
        for row, i in zip(rows, (0, 999)):     # %.9g round-trips fp32 exactly
            assert np.array_equal(row[:3], want["pos"][i, :3]) and np.array_equal(row[3:], want["vel"][i, :3]), text


@pytest.mark.gpu
def test_native_walk_over_the_whole_c_abi():
    """tests/cpp/api_walk (built by `make`): every entry point of include/particles_hip.h called from
    a native program, status codes checked; `make sanitize` builds the copy whose host code runs under
    AddressSanitizer and UndefinedBehaviorSanitizer."""
    exe = os.path.join(ROOT, "tests", "cpp", "api_walk")
    assert os.path.exists(exe), "tests/cpp/api_walk is not built (make)"
    done = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and "api_walk ok" in done.stdout, done.stdout + done.stderr
