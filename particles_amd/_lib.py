"""ctypes binding of the C ABI declared in include/particles_hip.h.

The HIP library is the only compute backend: when it is missing or cannot be
loaded this module raises, there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libparticles_hip.so")

PCL_OK = 0
PCL_ERR_INVALID = -1
PCL_ERR_HIP = -2
PCL_ERR_NO_DEVICE = -3
PCL_ERR_OUT_OF_RANGE = -4
PCL_ERR_DEVICE = -5
PCL_ERR_COMM = -6
PCL_ERR_SINGULAR = -7
UNIQUE_ID_BYTES = 128

KERNEL_NONE, KERNEL_STREAM, KERNEL_PAIRS = 0, 1, 2
STEP_NONE, STEP_STREAM, STEP_PAIRS, STEP_STREAM_COUNT, STEP_STREAM_SCATTER, STEP_PAIRS_BINNED = 0, 1, 2, 3, 4, 5
STEP_PAIRS_GATHERED, STEP_STREAM_FUSED = 6, 7
PAIRS_AUTO, PAIRS_ALL, PAIRS_BINNED = 0, 1, 2


class PclError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"particles_hip error {code}: {message}")
        self.code = code


class Diagnostics(C.Structure):
    """pcl_diagnostics (48 bytes)."""
    _fields_ = [
        ("count", C.c_double),
        ("kinetic_energy", C.c_double),
        ("pos_max", C.c_float * 3),
        ("neg_pos_min", C.c_float * 3),
        ("pad", C.c_float * 2),
    ]


assert C.sizeof(Diagnostics) == 48

_vp, _sz, _u64, _u32, _i, _f = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int, C.c_float

# name -> (restype, argtypes); every symbol the header declares.
SIGNATURES = {
    "pcl_last_error": (C.c_char_p, []),
    "pcl_device_count": (_i, []),
    "pcl_sim_params_default": (None, [_vp]),
    "pcl_create": (_i, [_vp, _sz, _vp, _sz, _i, C.POINTER(_vp)]),
    "pcl_destroy": (_i, [_vp]),
    "pcl_upload_particles": (_i, [_vp, _vp, _sz]),
    "pcl_upload_particles_soa": (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcl_generate_particles": (_i, [_vp, _sz, _u64, _u64, _f, _f]),
    "pcl_update_force_grid": (_i, [_vp, _vp, _sz]),
    "pcl_update_force_grid_device": (_i, [_vp, _vp, _sz]),
    "pcl_patch_force_grid": (_i, [_vp, _sz, _vp, _sz]),
    "pcl_patch_force_grid_device": (_i, [_vp, _sz, _vp, _sz]),
    "pcl_patch_force_grid_indexed": (_i, [_vp, _vp, _vp, _sz]),
    "pcl_patch_force_grid_indexed_device": (_i, [_vp, _vp, _vp, _sz]),
    "pcl_brush_input_default": (None, [_vp]),
    "pcl_brush_mouse_down": (_i, [_vp, _vp]),
    "pcl_brush_mouse_down_device": (_i, [_vp, _vp]),
    "pcl_brush_mouse_moved": (_i, [_vp, _vp]),
    "pcl_brush_mouse_moved_device": (_i, [_vp, _vp]),
    "pcl_brush_mouse_up": (_i, [_vp]),
    "pcl_brush_selection": (_i, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "pcl_read_force_grid": (_i, [_vp, _vp, _sz, _sz]),
    "pcl_num_cells": (_sz, [_vp]),
    "pcl_update_sim_params": (_i, [_vp, _vp]),
    "pcl_step": (_i, [_vp]),
    "pcl_step_n": (_i, [_vp, _u64]),
    "pcl_step_dt": (_i, [_vp, _f]),
    "pcl_pack_pair_sources_device": (_i, [_vp, _vp, _sz]),
    "pcl_step_pairs_gathered": (_i, [_vp, _vp, _sz]),
    "pcl_force_pair_kernel": (_i, [_vp, _i]),
    "pcl_set_pair_mode": (_i, [_vp, _i]),
    "pcl_last_step_kernel": (_i, [_vp]),
    "pcl_enable_step_diagnostics": (_i, [_vp, _i]),
    "pcl_num_particles": (_sz, [_vp]),
    "pcl_step_count": (_u64, [_vp]),
    "pcl_read_particles": (_i, [_vp, _vp, _sz, _sz]),
    "pcl_read_soa": (_i, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcl_pack_particles_device": (_i, [_vp, _vp, _sz, _sz]),
    "pcl_pack_particles_device_unordered": (_i, [_vp, _vp]),
    "pcl_diagnostics_read": (_i, [_vp, C.POINTER(Diagnostics)]),
    "pcl_diagnostics_device": (_i, [_vp, _vp]),
    "pcl_diagnostics_device_noalloc": (_i, [_vp, _vp]),
    "pcl_debug_cell_indices": (_i, [_vp, _vp, _sz, _sz]),
    "pcl_debug_pair_cells": (_i, [_vp, _vp, _vp, _sz, C.POINTER(_u32), C.POINTER(_u32)]),
    "pcl_pair_sweep_info": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "pcl_debug_slot_order": (_i, [_vp, _vp, _sz]),
    "pcl_debug_state_checksum": (_i, [_vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pcl_set_stream": (_i, [_vp, _vp]),
    "pcl_use_null_stream": (_i, [_vp]),
    "pcl_synchronize": (_i, [_vp]),
    "pcl_timer_start": (_i, [_vp]),
    "pcl_timer_stop": (_i, [_vp, C.POINTER(_f)]),
    "pcl_step_timing_begin": (_i, [_vp, _sz]),
    "pcl_step_timing_read": (_i, [_vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    "pcl_launch_geometry": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "pcl_device_pci_bus_id": (_i, [_vp, C.c_char_p, _sz]),
    "pcl_launch_residency": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "pcl_set_reorder_interval": (_i, [_vp, _i]),
    "pcl_reorder_interval_in_use": (_i, [_vp]),
    "pcl_reorder_now": (_i, [_vp]),
    "pcl_reorder_count": (_u64, [_vp]),
    "pcl_set_tuning": (_i, [_vp, _u32, _u32]),
    # authoring helpers (host only)
    "pcl_poly7_from_points": (_i, [_vp, _vp]),
    "pcl_poly7_eval": (_f, [_vp, _f]),
    "pcl_grid_centered": (None, [_sz, _sz, _sz, _vp]),
    "pcl_grid_uniform": (None, [_sz, _sz, _sz, _vp, _vp]),
    "pcl_grid_positions": (None, [_sz, _sz, _sz, _vp, _vp, _vp]),
    "pcl_sim_params_force_grid": (_i, [_vp, _i, _vp, _sz]),
    # several GPUs: RCCL inside the library
    "pcl_comm_unique_id": (_i, [_vp]),
    "pcl_comm_attach": (_i, [_vp, _vp, _i, _i]),
    "pcl_comm_detach": (_i, [_vp]),
    "pcl_comm_rank": (_i, [_vp]),
    "pcl_comm_world": (_i, [_vp]),
    "pcl_comm_count": (_i, [_vp]),
    "pcl_comm_update_force_grid": (_i, [_vp, _vp, _sz, _i]),
    "pcl_comm_patch_force_grid": (_i, [_vp, _sz, _vp, _sz, _i]),
    "pcl_comm_patch_force_grid_indexed": (_i, [_vp, _vp, _vp, _sz, _i]),
    "pcl_comm_brush_mouse_down": (_i, [_vp, _vp, _i]),
    "pcl_comm_brush_mouse_moved": (_i, [_vp, _vp, _i]),
    "pcl_comm_step": (_i, [_vp]),
    "pcl_comm_diagnostics_read": (_i, [_vp, C.POINTER(Diagnostics)]),
    "pcl_comm_diagnostics_device": (_i, [_vp, C.POINTER(_vp)]),
    "pcl_group_create": (_i, [_vp, _i, _vp, _sz, _vp, _sz, C.POINTER(_vp)]),
    "pcl_group_destroy": (_i, [_vp]),
    "pcl_group_size": (_i, [_vp]),
    "pcl_group_handle": (_vp, [_vp, _i]),
    "pcl_group_num_particles": (_sz, [_vp]),
    "pcl_group_upload_particles": (_i, [_vp, _vp, _sz]),
    "pcl_group_generate_particles": (_i, [_vp, _sz, _u64, _f, _f]),
    "pcl_group_update_force_grid": (_i, [_vp, _vp, _sz]),
    "pcl_group_patch_force_grid": (_i, [_vp, _sz, _vp, _sz]),
    "pcl_group_patch_force_grid_indexed": (_i, [_vp, _vp, _vp, _sz]),
    "pcl_group_update_sim_params": (_i, [_vp, _vp]),
    "pcl_group_enable_step_diagnostics": (_i, [_vp, _i]),
    "pcl_group_step": (_i, [_vp]),
    "pcl_group_brush_mouse_down": (_i, [_vp, _vp]),
    "pcl_group_brush_mouse_moved": (_i, [_vp, _vp]),
    "pcl_group_brush_mouse_up": (_i, [_vp]),
    "pcl_group_diagnostics_read": (_i, [_vp, C.POINTER(Diagnostics)]),
    "pcl_group_read_particles": (_i, [_vp, _vp, _sz, _sz]),
    "pcl_group_read_force_grid": (_i, [_vp, _i, _vp, _sz, _sz]),
    "pcl_group_synchronize": (_i, [_vp]),
}

_lib = None


def _share_torch_hip_runtime():
    """Keep the process on ONE HIP runtime.  PyTorch-ROCm ships its own copy of
    libamdhip64 and asks for it by file name; this library asks for the soname
    libamdhip64.so.7.  Loaded in the order "this library, then torch" the two
    requests resolve to two different files, and the second runtime to
    initialise finds no GPU.  Loading torch's copy first (when torch is
    installed; torch itself is not imported) makes both requests resolve to it."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    path = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        # The library loads RCCL at first use; the copy built against that same HIP runtime is
        # PyTorch's, so name it unless the caller has chosen one.
        rccl = os.path.join(libdir, "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("PCL_RCCL_LIBRARY", rccl)


def load():
    """Load libparticles_hip.so (built in-tree by ``make``); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make` at the repository root "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != PCL_OK:
        msg = load().pcl_last_error()
        raise PclError(code, msg.decode("utf-8", "replace") if msg else "")
    return code
