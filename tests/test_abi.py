"""The C-ABI library loads without a GPU and exports every symbol that
include/particles_hip.h declares; calls that need a device fail with a status
code and a message instead of crashing."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import particles_amd as pa
from particles_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "particles_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("pcl_create", "pcl_destroy", "pcl_upload_particles", "pcl_update_force_grid",
                 "pcl_patch_force_grid", "pcl_update_sim_params", "pcl_step", "pcl_step_dt",
                 "pcl_read_particles", "pcl_read_soa", "pcl_diagnostics_read", "pcl_debug_cell_indices",
                 "pcl_num_particles", "pcl_synchronize", "pcl_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in particles_hip.h but not exported"


def test_python_binding_covers_every_declared_symbol():
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()


def test_default_params_match_reference_defaults():
    lib = _lib.load()
    buf = np.zeros((), dtype=pa.SIM_PARAMS_DTYPE)
    lib.pcl_sim_params_default(buf.ctypes.data_as(C.c_void_p))
    assert buf.tobytes() == pa.SimParams().tobytes()     # src/sim_params.rs:20-31


def test_null_and_invalid_arguments_return_status_codes():
    lib = _lib.load()
    assert lib.pcl_step(None) == _lib.PCL_ERR_INVALID
    assert b"null handle" in lib.pcl_last_error()
    assert lib.pcl_update_sim_params(None, None) == _lib.PCL_ERR_INVALID
    out = C.c_void_p()
    g = np.zeros((1, 4), dtype=np.float32)
    assert lib.pcl_create(None, 5, g.ctypes.data_as(C.c_void_p), 1, 0, C.byref(out)) == _lib.PCL_ERR_INVALID
    assert lib.pcl_create(None, 0, None, 0, 0, C.byref(out)) == _lib.PCL_ERR_INVALID
    assert lib.pcl_num_particles(None) == 0 and lib.pcl_destroy(None) == _lib.PCL_OK


def test_no_cpu_fallback_without_a_device():
    """On a machine without a GPU construction fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pa.PclError):
        pa.Compute.new(pa.empty_particles(4), np.zeros((1, 4), dtype=np.float32))


def test_header_is_valid_c11_and_links_from_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as strict C11 and a
    plain C program must link against the library (no C++ in the interface)."""
    import subprocess
    src = tmp_path / "abi_probe.c"
    src.write_text(
        '#include "particles_hip.h"\n'
        '#include <stdio.h>\n'
        'int main(void) {\n'
        '    pcl_sim_params p; pcl_brush_input b; pcl_compute* h = 0;\n'
        '    pcl_sim_params_default(&p); pcl_brush_input_default(&b);\n'
        '    if (sizeof(pcl_particle) != 48 || sizeof(p) != 912 || sizeof(b) != 64) return 1;\n'
        '    if (pcl_step(h) != PCL_ERR_INVALID) return 2;          /* null handle: status code, no crash */\n'
        '    printf("%g %g %s\\n", (double)p.max_velocity, (double)b.falloff_dist, pcl_last_error());\n'
        '    return 0;\n'
        '}\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "abi_probe"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                           str(src), "-o", str(exe), "-L", os.path.join(root, "particles_amd"), "-lparticles_hip",
                           "-Wl,-rpath," + os.path.join(root, "particles_amd"), "-Wl,-rpath-link,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True)
    assert out.split()[:2] == ["100", "1"]
